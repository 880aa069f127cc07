"""bindings/rust/src/ffi.rs against include/birdnet_hip.h, textually (the image has no rustc, so nothing compiles the crate here;
VERDICT r4 item 7).  The public surface the crate serves is the reference's src/lib.rs:93-108 and the context getters of
src/batch_context.rs:135-165.

* ffi.rs is exactly what tools/gen_rust_ffi.py writes for the committed header (a header edit without a regenerated binding fails);
* independently of the generator's type mapper: every header entry point is bound, with the same arity, the same pointer depth
  and constness per argument, and the same scalar width;
* struct fields (names, order, widths, array extents) and the status / model-type constants agree;
* the safe shim (classifier_hip.rs) only calls entry points that exist, with the number of arguments they take."""
from __future__ import annotations

import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "birdnet_hip.h")
FFI = os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")
SHIM = os.path.join(ROOT, "bindings", "rust", "src", "classifier_hip.rs")

WIDTH = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64",
         "char": "c_char", "void": "c_void", "bn_status": "i32"}


def _gen():
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _header_text():
    t = open(HEADER).read()
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//.*", "", t)


def _c_functions():
    out = {}
    for m in re.finditer(r"^\s*([A-Za-z_][\w\s\*]*?)\b(bn_\w+)\s*\(([^;{]*?)\)\s*;", _header_text(), flags=re.M | re.S):
        args = " ".join(m.group(3).split())
        out[m.group(2)] = (" ".join(m.group(1).split()), [] if args == "void" else [a.strip() for a in args.split(",")])
    return out


def _rust_functions():
    body = re.search(r'extern "C" \{(.*?)\n\}', open(FFI).read(), flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (bn_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", body, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = ((m.group(3) or "()").strip(), [a.split(":", 1)[1].strip() for a in args])
    return out


def _c_shape(decl: str):
    """(scalar, [constness of each pointer level's POINTEE, outermost pointer first]) of a C parameter / return declarator"""
    decl = decl.replace("volatile", " ")
    am = re.match(r"^\s*(\w+)\s*\(\*\s*\w*\)\s*\[(\w+)\]\s*$", decl)
    if am:  # pointer to array: char (*names)[N]
        return am.group(1) + f"[{am.group(2)}]", [False]
    toks = re.findall(r"\w+|\*", decl)
    if toks and toks[-1] != "*" and len([t for t in toks if t not in ("const", "*")]) == 2:
        toks = toks[:-1]  # drop the parameter name
    const_base = toks[0] == "const"
    base = toks[1] if const_base else toks[0]
    rest = toks[2:] if const_base else toks[1:]
    levels, pointee_const = [], const_base
    i = 0
    while i < len(rest):
        assert rest[i] == "*", decl
        levels.append(pointee_const)
        pointee_const = i + 1 < len(rest) and rest[i + 1] == "const"
        i += 2 if pointee_const else 1
    return base, list(reversed(levels))


def _rust_shape(ty: str):
    levels = []
    while True:
        m = re.match(r"^\*(const|mut)\s+(.*)$", ty)
        if not m:
            break
        levels.append(m.group(1) == "const")
        ty = m.group(2)
    am = re.match(r"^\[(\w+); (\w+)\]$", ty)
    if am:
        return am.group(1) + f"[{am.group(2)}]", levels
    return ty, levels


def test_ffi_rs_is_what_the_generator_writes_for_this_header():
    assert open(FFI).read() == _gen().generate(), "include/birdnet_hip.h changed: run `python tools/gen_rust_ffi.py`"


def test_every_entry_point_is_bound_with_the_header_arity_pointerness_and_widths():
    cf, rf = _c_functions(), _rust_functions()
    assert len(cf) >= 54 and set(cf) == set(rf), (sorted(set(cf) - set(rf)), sorted(set(rf) - set(cf)))
    # the entry points rows f3 / f4 / a10 need from Rust and round 4 lacked
    for need in ("bn_ctx_read_output", "bn_recording_create_resampled", "bn_recording_wait", "bn_model_io_info", "bn_group_get_stats"):
        assert need in rf
    for name, (cret, cargs) in cf.items():
        rret, rargs = rf[name]
        assert len(cargs) == len(rargs), (name, cargs, rargs)
        for ca, ra in zip(cargs, rargs):
            cb, cl = _c_shape(ca)
            rb, rl = _rust_shape(ra)
            cb = re.sub(r"^(\w+)", lambda m: WIDTH.get(m.group(1), m.group(1)), cb)
            assert (cb, cl) == (rb, rl), (name, ca, ra)
        if cret == "void":
            assert rret == "()", name
        else:
            cb, cl = _c_shape(cret + " x" if "*" not in cret else cret)
            rb, rl = _rust_shape(rret)
            assert (WIDTH.get(cb, cb), cl) == (rb, rl), (name, cret, rret)


def test_structs_and_constants_agree():
    gen = _gen()
    text, rust = _header_text(), open(FFI).read()
    for sname, fields in gen.structs(text):
        body = re.search(r"pub struct %s \{(.*?)\}" % sname, rust, flags=re.S).group(1)
        got = [(m.group(1), m.group(2)) for m in re.finditer(r"pub (?:r#)?(\w+): ([^,]+),", body)]
        want = []
        for fname, cty, dims in fields:
            ty = WIDTH[cty]
            for d in reversed(dims):
                ty = f"[{ty}; {d}]"
            want.append((fname, ty))
        assert got == want, sname
        assert "#[repr(C)]\n#[derive(Clone, Copy)]\npub struct %s {" % sname in rust
    consts = dict(re.findall(r"pub const (BN_\w+): \w+ = (-?\d+);", rust))
    for _, items in gen.enums(text):
        for k, v in items:
            assert consts[k] == str(v), k
    for k, (v, _) in gen.defines(open(HEADER).read()).items():
        assert consts[k] == str(v), k
    # the opaque handles exist and nothing else of the header's typedefs is missing
    for o in re.findall(r"typedef struct (\w+) \1;", text):
        assert "pub struct %s { _p: [u8; 0] }" % o in rust, o


def test_safe_shim_calls_only_bound_entry_points_with_their_arity():
    rf = _rust_functions()
    src = re.sub(r"//.*", "", open(SHIM).read())
    calls = 0
    for m in re.finditer(r"\b(bn_[a-z_0-9]+)\s*\(", src):
        name = m.group(1)
        if name not in rf:
            assert re.search(r"\b(struct|fn|type)\s+%s\b" % name, src) or name in ("bn_model_config", "bn_ctx_stats"), name
            continue
        depth, i, n, seen = 1, m.end(), 0, False
        while depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                n += 1
            if depth and not ch.isspace():
                seen = True
            i += 1
        tail = src[m.end():i - 1].rstrip()
        n_args = 0 if not seen else n + (0 if tail.endswith(",") else 1)
        assert n_args == len(rf[name][1]), (name, n_args, len(rf[name][1]))
        calls += 1
    assert calls >= 8
