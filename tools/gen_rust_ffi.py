#!/usr/bin/env python3
"""Writes bindings/rust/src/ffi.rs from include/birdnet_hip.h: every struct, status / model-type constant and entry point
of the C ABI, in header order.  The build image has no rustc, so nothing can compile the crate here; this generator is what
keeps it honest instead -- tests/test_rust_binding.py re-runs it on the committed header and fails when ffi.rs differs,
and checks names, arity, pointer-ness and integer widths of every `extern "C"` item against its own parse of the header.

    python tools/gen_rust_ffi.py            # rewrite bindings/rust/src/ffi.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if the committed file is stale

The public surface these bindings serve is the reference's lib.rs:93-108 (Classifier, BatchInferenceContext, ...)."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "birdnet_hip.h")
OUT = os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64",
           "char": "c_char", "void": "c_void", "bn_status": "i32", "int": "i32"}
OPAQUE = ["bn_model", "bn_ctx", "bn_recording", "bn_group"]
RUST_KEYWORDS = {"type", "in", "ref", "box", "fn", "loop", "match", "move", "mod", "impl", "use", "where", "as"}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//.*", "", text)


def defines(text: str) -> dict:
    """{name: (value, rust type)}: `0u`-suffixed values are the u32 context flags, BN_PCM_* / BN_SHARING_* i32 arguments, the ABI version is
    compared with bn_abi_version()'s i32, everything else sizes an array"""
    out = {}
    for m in re.finditer(r"^#define\s+(BN_[A-Z_0-9]+)\s+\(?(-?\d+)\)?(u?)(?=\s|$)", text, flags=re.M):
        k = m.group(1)
        out[k] = (int(m.group(2)), "u32" if m.group(3) else "i32" if k == "BN_ABI_VERSION" or k.startswith(("BN_PCM_", "BN_SHARING_")) else "usize")
    return out


def enums(text: str):
    """[(enum name, [(constant, value)])]"""
    out = []
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        items = []
        for it in m.group(2).split(","):
            it = it.strip()
            if it:
                k, v = it.split("=")
                items.append((k.strip(), int(v.strip(), 0)))
        out.append((m.group(1), items))
    return out


def structs(text: str):
    """[(struct name, [(field, c type, [array dims as written])])] for the typedef structs with a body"""
    out = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            f = " ".join(f.split())
            if not f:
                continue
            fm = re.match(r"^([\w ]+?)\s*(\w+)((?:\[\w+\])*)$", f)
            assert fm, f
            fields.append((fm.group(2), fm.group(1).strip(), re.findall(r"\[(\w+)\]", fm.group(3))))
        out.append((m.group(1), fields))
    return out


def functions(text: str):
    """[(name, return c type, [(param name, c declarator text)])] in header order"""
    out = []
    for m in re.finditer(r"^\s*([A-Za-z_][\w\s\*]*?)\b(bn_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"^(.*?)(\w+)$", a) if "(*" not in a else re.match(r"^(.*?\(\*)(\w+)(\)\[\w+\])$", a)
                assert am, a
                if "(*" in a:
                    params.append((am.group(2), am.group(1)[:-2].strip() + " (*)" + am.group(3)[1:]))
                else:
                    params.append((am.group(2), am.group(1).strip()))
        out.append((name, ret, params))
    return out


def rust_type(c: str, consts: dict) -> str:
    """C declarator without the name -> Rust type.  Handles `const T *`, `T **`, `const T *const *`, `T *const *`,
    `const volatile int32_t *` (the cancellation flag: an AtomicI32's address on the Rust side) and `char (*)[N]`."""
    c = c.replace("volatile ", "").strip()
    am = re.match(r"^(\w+) \(\*\)\[(\w+)\]$", c)
    if am:
        return f"*mut [{rust_type(am.group(1), consts)}; {am.group(2)}]"
    toks = re.findall(r"\w+|\*", c)
    # base type = leading [const] name; then a sequence of `*` each optionally followed by `const`
    i, base_const = 0, False
    if toks[i] == "const":
        base_const, i = True, 1
    base = toks[i]
    i += 1
    ty = SCALARS.get(base, base)
    pointee_const = base_const
    while i < len(toks):
        assert toks[i] == "*", c
        i += 1
        ty = ("*const " if pointee_const else "*mut ") + ty
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const, i = True, i + 1
    return ty


def ident(name: str) -> str:
    return "r#" + name if name in RUST_KEYWORDS else name


def generate() -> str:
    raw = open(HEADER).read()
    text = strip_comments(raw)
    consts = defines(raw)
    lines = ["//! Raw bindings of include/birdnet_hip.h -- GENERATED by tools/gen_rust_ffi.py, do not edit (tests/test_rust_binding.py",
             "//! fails when this file and the header disagree).  Every entry point, struct and constant of the C ABI, in header order.",
             "//! Source only -- see ../README.md.",
             "#![allow(non_camel_case_types)]",
             "use core::ffi::c_void;",
             "use std::os::raw::c_char;",
             ""]
    for k, (v, ty) in consts.items():
        lines.append(f"pub const {k}: {ty} = {v};")
    lines.append("")
    lines += ["/// Call once before anything else: a library built from another header revision is refused instead of misread.",
              "pub fn assert_abi() {",
              "    let got = unsafe { bn_abi_version() };",
              '    assert_eq!(got, BN_ABI_VERSION, "libbirdnet_hip speaks ABI {got}, this crate was written for ABI {BN_ABI_VERSION}");',
              "}", ""]
    for ename, items in enums(text):
        lines.append(f"// {ename}")
        for k, v in items:
            lines.append(f"pub const {k}: i32 = {v};")
        lines.append("")
    for o in OPAQUE:
        lines += ["#[repr(C)]", f"pub struct {o} {{ _p: [u8; 0] }}"]
    lines.append("")
    for sname, fields in structs(text):
        lines += ["#[repr(C)]", "#[derive(Clone, Copy)]", f"pub struct {sname} {{"]
        for fname, cty, dims in fields:
            ty = SCALARS[cty]
            for d in reversed(dims):
                ty = f"[{ty}; {d}]"
            lines.append(f"    pub {ident(fname)}: {ty},")
        # plain numbers and arrays of them: all-zero bytes are a valid value (arrays longer than 32 have no derived Default)
        lines += ["}", f"impl Default for {sname} {{", "    fn default() -> Self { unsafe { core::mem::zeroed() } }", "}", ""]
    lines += ['#[link(name = "birdnet_hip")]', 'extern "C" {']
    for name, ret, params in functions(text):
        ps = ", ".join(f"{ident(p)}: {rust_type(t, consts)}" for p, t in params)
        r = "" if ret == "void" else f" -> {rust_type(ret, consts)}"
        lines.append(f"    pub fn {name}({ps}){r};")
    lines += ["}", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    want = generate()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == want else 1)
    open(OUT, "w").write(want)
    print(f"wrote {OUT}: {want.count('pub fn bn_')} entry points")
