// ONNX graph -> launch plan.  See engine.h.
//
// Conventions
//  * Every activation has the batch as its outermost dimension; `dims` /
//    `strides` below describe ONE sample (batch stripped).  Ops that would mix
//    the batch with other dimensions are refused.
//  * Activations are strided views over per-sample storages, so Transpose,
//    Reshape, Squeeze, Unsqueeze, Slice (incl. negative steps) and Identity
//    cost nothing.  Convolutions produce channels-last (NHWC / NWC) storage,
//    which is what the MFMA GEMM and the depthwise kernel want; an ONNX NCHW
//    tensor is simply the view dims=[C,H,W], strides=[1,W*C,C] over it.
//  * Conv+BatchNorm+activation(+residual Add) and MatMul/Gemm+bias+activation
//    chains are folded into one launch at import time.
#include "engine.h"
#include "plan_rules.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <numeric>
#include <set>

namespace bn {
namespace {

constexpr int64_t BATCH_SENTINEL = -1000000007;  // "the batch size" inside folded shape tensors

using Dims = std::vector<int64_t>;

int64_t prod(const Dims &d) {
    int64_t n = 1;
    for (auto v : d) n *= v;
    return n;
}
Dims row_major(const Dims &d) {
    Dims s(d.size());
    int64_t acc = 1;
    for (int i = (int)d.size() - 1; i >= 0; i--) {
        s[i] = acc;
        acc *= d[i];
    }
    return s;
}
std::string dims_str(const Dims &d) {
    std::string s = "[";
    for (size_t i = 0; i < d.size(); i++) s += (i ? "," : "") + std::to_string(d[i]);
    return s + "]";
}

struct Val {
    bool is_const = false;
    bool is_int = false;
    Dims dims;     // const: full dims; activation: per-sample dims
    Dims strides;  // activation only (elements)
    Space space = Space::NONE;
    int32_t storage = -1;
    int64_t offset = 0;
    std::vector<float> f;
    std::vector<int64_t> i;
    // squeeze-excite: `is_gate` marks the per-channel gate vector produced by the fused SE kernel;
    // `gate_storage >= 0` marks "x * gate" not yet applied (the consuming 1x1 conv applies it on load)
    bool is_gate = false;
    int32_t gate_storage = -1;
    // (round 5) frames * window not yet applied: the constant `win_name` (vals_ key) multiplies along the `win_nu`-th NON-UNIT
    // dimension; the DFT node that consumes the value folds it into its basis, any other consumer gets apply_window() in get()
    std::string win_name;
    int32_t win_nu = -1;
    int64_t numel() const { return prod(dims); }
    bool contiguous() const { return !is_const && strides_equal(row_major(dims)); }
    bool strides_equal(const Dims &s) const {
        for (size_t k = 0; k < dims.size(); k++)
            if (dims[k] != 1 && strides[k] != s[k]) return false;
        return true;
    }
};

[[noreturn]] void unsupported(const OnnxNode &n, const std::string &why) {
    throw UnsupportedModel("node '" + n.name + "' (" + n.op_type + "): " + why);
}

struct ActSpec {
    int32_t act = ACT_NONE;
    float p0 = 0, p1 = 0;
};

class Builder {
   public:
    Builder(const OnnxModel &m, Plan &p) : m_(m), plan_(p) {}

    void run(const std::vector<int> &wanted) {
        plan_.io = read_io_meta(m_);
        if (m_.inputs.size() != 1)
            throw UnsupportedModel("expected exactly one graph input, found " + std::to_string(m_.inputs.size()));
        nodes_ = m_.nodes;  // mutable copy (pruning edits attributes / initializers)
        for (auto &t : m_.initializers) {
            Val v;
            v.is_const = true;
            v.dims = t.dims;
            if (t.is_float()) v.f = t.f;
            else { v.is_int = true; v.i = t.i; }
            vals_[t.name] = std::move(v);
        }
        canonicalize_spectrogram_dialects();
        // graph input
        const auto &in = m_.inputs[0];
        if (!in.has_shape || in.shape.size() < 2)
            throw UnsupportedModel("graph input must have a [batch, samples] or [batch, 1, samples] shape");
        {
            Val v;
            v.dims.assign(in.shape.begin() + 1, in.shape.end());
            for (auto d : v.dims)
                if (d <= 0) throw UnsupportedModel("graph input has a dynamic non-batch dimension");
            v.strides = row_major(v.dims);
            v.space = Space::INPUT;
            vals_[in.name] = v;
            plan_.sample_count = v.numel();
        }
        // liveness from wanted outputs
        std::set<std::string> want;
        for (int w : wanted) {
            if (w < 0 || w >= (int)m_.outputs.size()) throw UnsupportedModel("wanted output index out of range");
            want.insert(m_.outputs[w].name);
        }
        for (auto &o : m_.outputs) graph_outputs_.insert(o.name);
        live_.assign(nodes_.size(), false);
        {
            std::map<std::string, int> producer;
            for (size_t k = 0; k < nodes_.size(); k++)
                for (auto &o : nodes_[k].outputs) producer[o] = (int)k;
            std::vector<std::string> stack(want.begin(), want.end());
            while (!stack.empty()) {
                std::string t = stack.back();
                stack.pop_back();
                auto it = producer.find(t);
                if (it == producer.end() || live_[it->second]) continue;
                live_[it->second] = true;
                for (auto &i : nodes_[it->second].inputs)
                    if (!i.empty()) stack.push_back(i);
            }
        }
        for (size_t k = 0; k < nodes_.size(); k++)
            if (live_[k])
                for (auto &i : nodes_[k].inputs)
                    if (!i.empty()) consumers_[i].push_back((int)k);
        wanted_names_ = want;
        absorbed_.assign(nodes_.size(), false);

        // Arity of every node up front: the lowering (and its look-ahead fusion) indexes inputs / outputs directly,
        // so a malformed file must be refused here, not found by an out-of-range read later.
        for (const auto &nd : nodes_) {
            static const std::map<std::string, size_t> min_inputs = {
                {"Add", 2}, {"Sub", 2}, {"Mul", 2}, {"Div", 2}, {"Pow", 2}, {"Max", 2}, {"Min", 2}, {"Conv", 2}, {"MatMul", 2}, {"Gemm", 2},
                {"BatchNormalization", 5}, {"Reshape", 2}, {"Gather", 2}, {"Concat", 1}};
            if (nd.outputs.empty() || nd.outputs[0].empty()) throw UnsupportedModel("node '" + nd.name + "' (" + nd.op_type + ") has no output");
            auto it = min_inputs.find(nd.op_type);
            const size_t need = it != min_inputs.end() ? it->second : (nd.op_type == "Constant" || nd.op_type == "ConstantOfShape" || nd.op_type == "Range" ? 0 : 1);
            if (nd.inputs.size() < need) throw UnsupportedModel("node '" + nd.name + "' (" + nd.op_type + ") has " + std::to_string(nd.inputs.size()) + " inputs, needs " + std::to_string(need));
            for (size_t k = 0; k < need; k++)
                if (nd.inputs[k].empty()) throw UnsupportedModel("node '" + nd.name + "' (" + nd.op_type + "): required input " + std::to_string(k) + " is missing");
        }

        prune_zero_rows();
        merge_framing_products();

        for (size_t k = 0; k < nodes_.size(); k++) {
            if (!live_[k] || absorbed_[k]) continue;
            cur_ = (int)k;
            lower(nodes_[k]);
        }
        // outputs
        plan_.outputs.resize(m_.outputs.size());
        for (size_t k = 0; k < m_.outputs.size(); k++) {
            auto &oi = plan_.outputs[k];
            oi.name = m_.outputs[k].name;
            if (!want.count(oi.name)) continue;
            auto it = vals_.find(oi.name);
            if (it == vals_.end()) throw UnsupportedModel("graph output '" + oi.name + "' was never produced");
            Val v = it->second;
            if (v.is_const) v = upload_as_activation(v);
            if (v.space == Space::INPUT || !v.contiguous()) v = materialize(v, row_major(v.dims), "output:" + oi.name);
            oi.dims = v.dims;
            oi.row_elems = v.numel();
            oi.ref = Ref{v.space, v.storage, v.offset};
            oi.computed = true;
            plan_.storages[v.storage].pinned = true;
        }
        fuse_elementwise_chains();
        pack_bf16x3_weights();
        plan_memory();
        for (auto &op : plan_.ops) {
            (op.mfma ? plan_.macs_mfma : plan_.macs_valu) += op.macs;
            plan_.macs_mfma += op.macs_mfma_extra;
            plan_.recompute_macs += op.macs_recompute;
            plan_.macs_valu += op.macs_valu_extra;
            plan_.act_bytes += op.bytes;
            plan_.weight_bytes += op.weight_bytes;
            plan_.fft_flops += op.flops_fft;
        }
        plan_.dft_gemm_macs = dft_gemm_macs_;
        plan_.dft_performed_macs = dft_performed_macs_;
        plan_.dft_fft_equiv_flops = dft_fft_equiv_flops_;
    }

   private:
    const OnnxModel &m_;
    Plan &plan_;
    std::vector<OnnxNode> nodes_;
    std::map<std::string, Val> vals_;
    std::map<std::string, std::vector<int>> consumers_;
    std::set<std::string> graph_outputs_, wanted_names_;
    std::vector<bool> live_, absorbed_;
    int cur_ = 0;
    double dft_gemm_macs_ = 0, dft_performed_macs_ = 0, dft_fft_equiv_flops_ = 0;

    // ------------------------------------------------------------------ utils
    const Val &get(const OnnxNode &n, size_t idx) {
        if (idx >= n.inputs.size() || n.inputs[idx].empty()) unsupported(n, "missing input " + std::to_string(idx));
        auto it = vals_.find(n.inputs[idx]);
        if (it == vals_.end()) unsupported(n, "input '" + n.inputs[idx] + "' is not defined (graph not topologically sorted?)");
        if (it->second.gate_storage >= 0 && !(n.op_type == "Conv" && idx == 0)) it->second = apply_gate(it->second, n.inputs[idx]);
        if (!it->second.win_name.empty() && !(idx == 0 && (n.op_type == "DFT" || n.op_type == "Unsqueeze" || n.op_type == "Squeeze" || n.op_type == "Reshape" ||
                                                          n.op_type == "Identity")))
            it->second = apply_window(it->second, n.inputs[idx]);
        return it->second;
    }
    // frames * window as an explicit elementwise launch (fallback when the consumer is not a DFT after all)
    Val apply_window(const Val &x, const std::string &why) {
        Val plain = x;
        plain.win_name.clear();
        plain.win_nu = -1;
        const Val &w = vals_.at(x.win_name);
        int ax = -1, nu = 0;
        for (size_t k = 0; k < x.dims.size(); k++)
            if (x.dims[k] != 1 && nu++ == x.win_nu) ax = (int)k;
        if (ax < 0 || w.numel() != x.dims[ax]) throw UnsupportedModel("internal: pending window of '" + why + "' lost its axis");
        Val out = new_act(x.dims, strides_for_order(x.dims, phys_order(x)));
        Dims ws(x.dims.size(), 0);
        ws[ax] = 1;
        emit_elt("window.mul:" + why, out, ref_of(plain), plain.strides, batch_stride(plain), Ref{Space::CONSTS, add_const(w.f), 0}, ws, 0, BIN_MUL, ActSpec{});
        return out;
    }
    // x * gate as an explicit elementwise launch (fallback when the consumer cannot fold the gate)
    Val apply_gate(const Val &x, const std::string &why) {
        Val plain = x;
        plain.gate_storage = -1;
        Val out = new_act(x.dims, x.strides_equal(row_major(x.dims)) ? row_major(x.dims) : strides_for_order(x.dims, phys_order(x)));
        Dims gs(x.dims.size(), 0);
        gs[0] = 1;  // gate is [C] along logical dim 0
        emit_elt("se.mul:" + why, out, ref_of(plain), plain.strides, batch_stride(plain), Ref{Space::ARENA, x.gate_storage, 0}, gs,
                 plan_.storages[x.gate_storage].elems, BIN_MUL, ActSpec{});
        return out;
    }
    bool has_input(const OnnxNode &n, size_t idx) const { return idx < n.inputs.size() && !n.inputs[idx].empty(); }
    const Val *opt(const OnnxNode &n, size_t idx) {
        if (!has_input(n, idx)) return nullptr;
        return &get(n, idx);
    }
    // Sole live consumer of tensor `t` (and t is not itself a wanted graph output), else -1.
    int sole_consumer(const std::string &t) {
        if (wanted_names_.count(t)) return -1;
        auto it = consumers_.find(t);
        if (it == consumers_.end()) return -1;
        int found = -1, cnt = 0;
        for (int k : it->second)
            if (!absorbed_[k]) { found = k; cnt++; }
        return cnt == 1 ? found : -1;
    }
    std::vector<int> live_consumers(const std::string &t) {
        std::vector<int> r;
        auto it = consumers_.find(t);
        if (it != consumers_.end())
            for (int k : it->second)
                if (!absorbed_[k]) r.push_back(k);
        return r;
    }
    bool const_scalar(const Val &v, float &out) {
        if (!v.is_const || v.numel() != 1) return false;
        out = v.is_int ? (float)v.i[0] : v.f[0];
        return true;
    }
    std::vector<int64_t> const_ints(const OnnxNode &n, const Val &v) {
        if (!v.is_const) unsupported(n, "expected a constant integer tensor");
        if (v.is_int) return v.i;
        std::vector<int64_t> r(v.f.size());
        for (size_t k = 0; k < r.size(); k++) r[k] = (int64_t)v.f[k];
        return r;
    }

    int32_t new_storage(int64_t elems) {
        Storage s;
        s.elems = elems;
        plan_.storages.push_back(s);
        return (int32_t)plan_.storages.size() - 1;
    }
    Val new_act(const Dims &dims, const Dims &strides) {
        Val v;
        v.dims = dims;
        v.strides = strides;
        v.space = Space::ARENA;
        // storage extent: max offset + 1
        int64_t ext = 1;
        for (size_t k = 0; k < dims.size(); k++) ext += (dims[k] - 1) * std::abs(strides[k]);
        v.storage = new_storage(ext);
        return v;
    }
    int32_t add_const(const std::vector<float> &data) {
        plan_.consts.push_back(data);
        return (int32_t)plan_.consts.size() - 1;
    }
    Ref ref_of(const Val &v) const { return Ref{v.space, v.storage, v.offset}; }
    void touch(const Ref &r, int op_index) {
        if (r.space != Space::ARENA) return;
        auto &s = plan_.storages[r.id];
        if (s.first < 0) s.first = op_index;
        s.last = std::max(s.last, op_index);
    }
    void push_op(PlanOp &&op) {
        int idx = (int)plan_.ops.size();
        touch(op.out, idx);
        touch(op.a, idx);
        touch(op.b, idx);
        touch(op.res, idx);
        touch(op.scale, idx);
        touch(op.w2, idx);
        touch(op.bias2, idx);
        for (auto &r : op.eb) touch(r, idx);
        for (auto &r : op.x) touch(r, idx);
        plan_.ops.push_back(std::move(op));
    }

    // Physical order of logical dims (outermost first) for an activation.
    static std::vector<int> phys_order(const Val &v) {
        std::vector<int> ord(v.dims.size());
        std::iota(ord.begin(), ord.end(), 0);
        // size-1 dims carry no layout information: treat them as outermost
        auto key = [&](int a) { return v.dims[a] == 1 ? INT64_MAX : std::abs(v.strides[a]); };
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key(a) > key(b); });
        return ord;
    }
    // Contiguous strides for `dims` laid out in the given physical order.
    static Dims strides_for_order(const Dims &dims, const std::vector<int> &ord) {
        Dims s(dims.size(), 0);
        int64_t acc = 1;
        for (int k = (int)ord.size() - 1; k >= 0; k--) {
            s[ord[k]] = acc;
            acc *= dims[ord[k]];
        }
        return s;
    }

    // Broadcast strides of `v` against out dims (right aligned); const operands are row-major.
    Dims bcast_strides(const OnnxNode *n, const Dims &vdims, const Dims &vstrides, const Dims &out) {
        Dims s(out.size(), 0);
        int off = (int)out.size() - (int)vdims.size();
        if (off < 0) {
            if (n) unsupported(*n, "operand rank exceeds result rank");
            throw UnsupportedModel("broadcast rank mismatch");
        }
        for (size_t k = 0; k < vdims.size(); k++) {
            if (vdims[k] == out[k + off]) s[k + off] = vdims[k] == 1 ? 0 : vstrides[k];
            else if (vdims[k] == 1) s[k + off] = 0;
            else {
                if (n) unsupported(*n, "shapes " + dims_str(vdims) + " and " + dims_str(out) + " do not broadcast");
                throw UnsupportedModel("broadcast mismatch");
            }
        }
        return s;
    }

    // Emit an elementwise op.  sb/b may be empty for unary ops.
    void emit_elt(const std::string &name, const Val &out, const Ref &a, const Dims &sa, int64_t a_bs,
                  const Ref &b, const Dims &sb, int64_t b_bs, int bin, ActSpec act) {
        // iterate in out's physical order
        std::vector<int> ord = phys_order(out);
        struct L { int64_t n, so, sa, sb; };
        std::vector<L> loops;
        for (int d : ord) {
            if (out.dims[d] == 1) continue;
            L l{out.dims[d], out.strides[d], sa[d], sb.empty() ? 0 : sb[d]};
            if (!loops.empty()) {
                L &p = loops.back();
                if (p.so == l.so * l.n && p.sa == l.sa * l.n && p.sb == l.sb * l.n) {
                    p.n *= l.n; p.so = l.so; p.sa = l.sa; p.sb = l.sb;
                    continue;
                }
            }
            loops.push_back(l);
        }
        if (loops.empty()) loops.push_back(L{1, 1, 0, 0});
        if ((int)loops.size() > ELT_MAX_DIMS) throw UnsupportedModel("elementwise op '" + name + "' needs more than 5 loop dims");
        PlanOp op;
        op.kind = OpKind::ELT;
        op.name = name;
        op.out = ref_of(out);
        op.a = a;
        op.eb[0] = b;
        EltDesc &d = op.elt;
        d.nd = (int)loops.size();
        d.per_sample = 1;
        d.nstages = 1;
        EltStage &st = d.st[0];
        for (int k = 0; k < d.nd; k++) {
            d.size[k] = loops[k].n; d.so[k] = loops[k].so; d.sa[k] = loops[k].sa; st.sb[k] = loops[k].sb;
            d.per_sample *= loops[k].n;
        }
        int64_t out_ext = plan_.storages[out.storage].elems;
        d.bo = out_ext; d.ba = a_bs; st.bb = b_bs;
        st.bin = bin; st.act = act.act; st.p0 = act.p0; st.p1 = act.p1;
        op.bytes = 4.0 * (double)d.per_sample * (bin == BIN_NONE ? 2 : 3);
        push_op(std::move(op));
    }
    int64_t batch_stride(const Val &v) const {
        if (v.space == Space::INPUT) return plan_.sample_count;
        if (v.space == Space::ARENA) return plan_.storages[v.storage].elems;
        return 0;
    }

    // Copy `v` into fresh storage with the requested strides.
    Val materialize(const Val &v, const Dims &strides, const std::string &why) {
        Val out = new_act(v.dims, strides);
        emit_elt("copy(" + why + ")", out, ref_of(v), v.strides, batch_stride(v), Ref{}, {}, 0, BIN_NONE, ActSpec{});
        return out;
    }
    Val upload_as_activation(const Val &c) {
        // a constant graph output / concat input: broadcast over the batch via stride 0
        std::vector<float> data = c.is_int ? std::vector<float>(c.i.begin(), c.i.end()) : c.f;
        Dims d = c.dims;
        if (!d.empty() && d[0] == 1) d.erase(d.begin());
        Val out = new_act(d, row_major(d));
        Ref src{Space::CONSTS, add_const(data), 0};
        emit_elt("copy(const)", out, src, row_major(d), 0, Ref{}, {}, 0, BIN_NONE, ActSpec{});
        return out;
    }
    void define(const std::string &name, Val v) { vals_[name] = std::move(v); }

    // --------------------------------------------------------- constant folding
    static bool all_const(const std::vector<const Val *> &vs) {
        for (auto v : vs)
            if (v && !v->is_const) return false;
        return true;
    }
    Val make_const_f(const Dims &d, std::vector<float> f) {
        Val v; v.is_const = true; v.dims = d; v.f = std::move(f); return v;
    }
    Val make_const_i(const Dims &d, std::vector<int64_t> i) {
        Val v; v.is_const = true; v.is_int = true; v.dims = d; v.i = std::move(i); return v;
    }
    static Dims bcast_dims(const Dims &a, const Dims &b) {
        size_t r = std::max(a.size(), b.size());
        Dims o(r);
        for (size_t k = 0; k < r; k++) {
            int64_t da = k + a.size() >= r ? a[k + a.size() - r] : 1;
            int64_t db = k + b.size() >= r ? b[k + b.size() - r] : 1;
            if (da != db && da != 1 && db != 1) throw UnsupportedModel("constant broadcast mismatch");
            o[k] = std::max(da, db);
        }
        return o;
    }
    template <class T, class F>
    static std::vector<T> bcast_apply(const Dims &ad, const std::vector<T> &a, const Dims &bd, const std::vector<T> &b,
                                      const Dims &od, F fn) {
        int64_t n = prod(od);
        std::vector<T> o((size_t)n);
        Dims as = row_major(ad), bs = row_major(bd), os = row_major(od);
        for (int64_t lin = 0; lin < n; lin++) {
            int64_t ia = 0, ib = 0, rem = lin;
            for (size_t k = 0; k < od.size(); k++) {
                int64_t idx = rem / os[k];
                rem %= os[k];
                int ka = (int)k - (int)(od.size() - ad.size()), kb = (int)k - (int)(od.size() - bd.size());
                if (ka >= 0 && ad[ka] != 1) ia += idx * as[ka];
                if (kb >= 0 && bd[kb] != 1) ib += idx * bs[kb];
            }
            o[lin] = fn(a[ia], b[ib]);
        }
        return o;
    }
    // (round 5) comparison / boolean operators: 0 / 1 integers on the constant side, f32 0.0 / 1.0 on the device side
    static int compare_code(const std::string &t) {
        return t == "Greater" ? BIN_GT : t == "Less" ? BIN_LT : t == "GreaterOrEqual" ? BIN_GE : t == "LessOrEqual" ? BIN_LE : t == "Equal" ? BIN_EQ
             : t == "Xor" ? BIN_NE : t == "bn.SelA" ? BIN_SELA : t == "bn.SelB" ? BIN_SELB : BIN_NONE;
    }
    bool fold_compare(const OnnxNode &n, const Val &a, const Val &b) {
        const std::string &t = n.op_type;
        const int code = t == "And" ? BIN_MUL : t == "Or" ? BIN_MAX : compare_code(t);
        if (code == BIN_NONE) return false;
        Dims od = bcast_dims(a.dims, b.dims);
        const bool ints = a.is_int && b.is_int;
        auto pick = [code](auto x, auto y) -> decltype(x) {
            using T = decltype(x);
            switch (code) {
                case BIN_GT: return (T)(x > y);
                case BIN_LT: return (T)(x < y);
                case BIN_GE: return (T)(x >= y);
                case BIN_LE: return (T)(x <= y);
                case BIN_EQ: return (T)(x == y);
                case BIN_NE: return (T)(x != y);
                case BIN_MUL: return (T)((x != 0) && (y != 0));
                case BIN_MAX: return (T)((x != 0) || (y != 0));
                case BIN_SELA: return y != 0 ? x : (T)0;
                default: return y != 0 ? (T)0 : x;  // BIN_SELB
            }
        };
        const bool keeps_type = code == BIN_SELA || code == BIN_SELB;  // a select keeps its data operand's type, a comparison yields 0 / 1
        if (ints) {
            define(n.outputs[0], make_const_i(od, bcast_apply<int64_t>(a.dims, a.i, b.dims, b.i, od, [&](int64_t x, int64_t y) { return pick(x, y); })));
            return true;
        }
        std::vector<float> af = a.is_int ? std::vector<float>(a.i.begin(), a.i.end()) : a.f;
        std::vector<float> bf = b.is_int ? std::vector<float>(b.i.begin(), b.i.end()) : b.f;
        std::vector<float> of = bcast_apply<float>(a.dims, af, b.dims, bf, od, [&](float x, float y) { return pick(x, y); });
        if (keeps_type && !a.is_int) define(n.outputs[0], make_const_f(od, of));
        else define(n.outputs[0], make_const_i(od, std::vector<int64_t>(of.begin(), of.end())));
        return true;
    }
    bool fold_binary(const OnnxNode &n, const Val &a, const Val &b) {
        const std::string &t = n.op_type;
        if (fold_compare(n, a, b)) return true;
        Dims od = bcast_dims(a.dims, b.dims);
        if (a.is_int && b.is_int) {
            std::function<int64_t(int64_t, int64_t)> fn;
            if (t == "Add") fn = [](int64_t x, int64_t y) { return x + y; };
            else if (t == "Sub") fn = [](int64_t x, int64_t y) { return x - y; };
            else if (t == "Mul") fn = [](int64_t x, int64_t y) { return (x == BATCH_SENTINEL || y == BATCH_SENTINEL) ? (x == 1 ? y : (y == 1 ? x : BATCH_SENTINEL)) : x * y; };
            else if (t == "Div") fn = [](int64_t x, int64_t y) { return y ? x / y : 0; };
            else if (t == "Max") fn = [](int64_t x, int64_t y) { return std::max(x, y); };
            else if (t == "Min") fn = [](int64_t x, int64_t y) { return std::min(x, y); };
            else return false;
            define(n.outputs[0], make_const_i(od, bcast_apply<int64_t>(a.dims, a.i, b.dims, b.i, od, fn)));
            return true;
        }
        std::vector<float> af = a.is_int ? std::vector<float>(a.i.begin(), a.i.end()) : a.f;
        std::vector<float> bf = b.is_int ? std::vector<float>(b.i.begin(), b.i.end()) : b.f;
        std::function<float(float, float)> fn;
        if (t == "Add") fn = [](float x, float y) { return x + y; };
        else if (t == "Sub") fn = [](float x, float y) { return x - y; };
        else if (t == "Mul") fn = [](float x, float y) { return x * y; };
        else if (t == "Div") fn = [](float x, float y) { return x / y; };
        else if (t == "Pow") fn = [](float x, float y) { return std::pow(x, y); };
        else if (t == "Max") fn = [](float x, float y) { return std::max(x, y); };
        else if (t == "Min") fn = [](float x, float y) { return std::min(x, y); };
        else return false;
        define(n.outputs[0], make_const_f(od, bcast_apply<float>(a.dims, af, b.dims, bf, od, fn)));
        return true;
    }
    bool fold_unary(const OnnxNode &n, const Val &a) {
        const std::string &t = n.op_type;
        std::function<float(float)> fn;
        if (t == "Sqrt") fn = [](float x) { return std::sqrt(x); };
        else if (t == "Exp") fn = [](float x) { return std::exp(x); };
        else if (t == "Log") fn = [](float x) { return std::log(x); };
        else if (t == "Neg") fn = [](float x) { return -x; };
        else if (t == "Abs") fn = [](float x) { return std::fabs(x); };
        else if (t == "Reciprocal") fn = [](float x) { return 1.0f / x; };
        else if (t == "Floor") fn = [](float x) { return std::floor(x); };
        else if (t == "Ceil") fn = [](float x) { return std::ceil(x); };
        else if (t == "Sigmoid") fn = [](float x) { return 1.0f / (1.0f + std::exp(-x)); };
        else if (t == "Relu") fn = [](float x) { return x > 0 ? x : 0.0f; };
        else if (t == "Tanh") fn = [](float x) { return std::tanh(x); };
        else return false;
        if (a.is_int) {
            if (t == "Neg") { std::vector<int64_t> o = a.i; for (auto &x : o) x = -x; define(n.outputs[0], make_const_i(a.dims, o)); return true; }
            if (t == "Abs") { std::vector<int64_t> o = a.i; for (auto &x : o) x = std::llabs(x); define(n.outputs[0], make_const_i(a.dims, o)); return true; }
        }
        std::vector<float> af = a.is_int ? std::vector<float>(a.i.begin(), a.i.end()) : a.f;
        for (auto &x : af) x = fn(x);
        define(n.outputs[0], make_const_f(a.dims, af));
        return true;
    }
    // generic strided gather of a constant: out[idx] = src[offset + sum idx*stride]
    Val const_view(const Val &c, const Dims &dims, const Dims &strides, int64_t offset) {
        Val o; o.is_const = true; o.is_int = c.is_int; o.dims = dims;
        int64_t n = prod(dims);
        Dims os = row_major(dims);
        if (c.is_int) o.i.resize(n); else o.f.resize(n);
        for (int64_t lin = 0; lin < n; lin++) {
            int64_t src = offset, rem = lin;
            for (size_t k = 0; k < dims.size(); k++) { src += (rem / os[k]) * strides[k]; rem %= os[k]; }
            if (c.is_int) o.i[lin] = c.i[src]; else o.f[lin] = c.f[src];
        }
        return o;
    }

    // ------------------------------------------------------------- one spelling for the real / imaginary part of a spectrogram (round 5)
    // Three exporter dialects write "the real part of an STFT" (what BirdNET v2.4 feeds its mel banks):
    //   (a) Conv1D with the windowed cosine basis as weights -> Transpose                                  (the form every node-level pass below knows)
    //   (b) STFT(signal, step, window) -> Gather(index c, last axis)                                        (torch.stft exports, opset 17)
    //   (c) Reshape -> Gather(affine selector) -> Reshape  [tf.signal.frame]  -> Mul(window) -> Unsqueeze -> DFT(onesided) -> Gather(index c)
    // (b) and (c) are rewritten into (a) here, before liveness and the passes that prune mel-dead bins, merge the mel product and pick the
    // fold, so that all three spellings reach the same plan.  Only the exact patterns are rewritten (every intermediate has ONE consumer and
    // is no graph output); anything else keeps its nodes and goes through lower_stft / lower_dft, which map the general case.
    void canonicalize_spectrogram_dialects() {
        if (env_int("BN_CANON_SPECTRO", 1) == 0) return;
        std::map<std::string, int> uses;
        for (auto &nd : nodes_)
            for (auto &i : nd.inputs)
                if (!i.empty()) uses[i]++;
        for (auto &o : m_.outputs) uses[o.name] += 2;  // a graph output is never an intermediate
        std::map<std::string, int> producer;
        for (size_t k = 0; k < nodes_.size(); k++)
            for (auto &o : nodes_[k].outputs) producer[o] = (int)k;
        auto cval = [&](const std::string &name) -> const Val * {
            auto it = vals_.find(name);
            return it != vals_.end() && it->second.is_const ? &it->second : nullptr;
        };
        auto ints = [&](const std::string &name, std::vector<int64_t> &out) {
            const Val *v = cval(name);
            if (!v) return false;
            out = v->is_int ? v->i : std::vector<int64_t>(v->f.begin(), v->f.end());
            return true;
        };
        auto sole_prod = [&](const std::string &t, const char *op) -> int {  // producer of t if it is `op` and t has one use
            auto it = producer.find(t);
            if (it == producer.end() || nodes_[it->second].op_type != op || uses[t] != 1) return -1;
            return it->second;
        };
        std::vector<char> dead(nodes_.size(), 0);
        std::vector<std::vector<OnnxNode>> repl(nodes_.size());
        int serial = 0;
        for (size_t gk = 0; gk < nodes_.size(); gk++) {
            const OnnxNode &pick = nodes_[gk];
            if (pick.op_type != "Gather" || pick.inputs.size() != 2) continue;
            std::vector<int64_t> ci;
            const Val *civ = cval(pick.inputs[1]);
            if (!civ || civ->numel() != 1 || !civ->dims.empty() || !ints(pick.inputs[1], ci) || (ci[0] != 0 && ci[0] != 1)) continue;
            const int64_t gax = pick.attr_i("axis", 0);
            if (gax != 3 && gax != -1) continue;
            std::string signal;
            std::vector<float> window;
            int64_t N = 0, hop = 0;
            std::vector<int> chain;
            int k1 = sole_prod(pick.inputs[0], "STFT");
            if (k1 >= 0) {                                                      // ---- (b)
                const OnnxNode &st = nodes_[k1];
                std::vector<int64_t> stepv, lenv;
                if (st.attr_i("onesided", 1) == 0 || st.inputs.size() < 2 || !ints(st.inputs[1], stepv) || stepv.size() != 1) continue;
                const Val *w = st.inputs.size() > 2 && !st.inputs[2].empty() ? cval(st.inputs[2]) : nullptr;
                if (st.inputs.size() > 2 && !st.inputs[2].empty() && (!w || w->is_int)) continue;
                if (st.inputs.size() > 3 && !st.inputs[3].empty()) { if (!ints(st.inputs[3], lenv) || lenv.size() != 1) continue; N = lenv[0]; }
                else if (w) N = w->numel();
                if (N < 2 || (w && w->numel() != N)) continue;
                if (w) window = w->f;
                hop = stepv[0];
                signal = st.inputs[0];
                chain = {k1};
            } else if ((k1 = sole_prod(pick.inputs[0], "DFT")) >= 0) {            // ---- (c)
                const OnnxNode &df = nodes_[k1];
                if (df.attr_i("onesided", 0) == 0 || df.attr_i("inverse", 0) != 0 || df.inputs.size() != 1) continue;
                const int64_t dax = df.has("axis") ? df.attr_i("axis", 1) : (m_.opset >= 20 ? -2 : 1);
                if (dax != 2 && dax != -2) continue;
                const int ku = sole_prod(df.inputs[0], "Unsqueeze");
                if (ku < 0) continue;
                std::vector<int64_t> ax = nodes_[ku].attr_ints("axes");
                if (ax.empty() && (nodes_[ku].inputs.size() < 2 || !ints(nodes_[ku].inputs[1], ax))) continue;
                if (ax.size() != 1 || (ax[0] != 3 && ax[0] != -1)) continue;
                std::string fr = nodes_[ku].inputs[0];
                chain = {k1, ku};
                const int km = sole_prod(fr, "Mul");
                if (km >= 0) {
                    const OnnxNode &mu = nodes_[km];
                    const Val *w0 = cval(mu.inputs[0]), *w1 = cval(mu.inputs[1]);
                    const Val *w = w0 ? w0 : w1;
                    if (!w || (w0 && w1) || w->is_int) continue;
                    int nu = 0;
                    for (auto d : w->dims) nu += d != 1;
                    if (nu != 1 || w->dims.empty() || w->dims.back() != w->numel()) continue;  // along the frames' last axis
                    window = w->f;
                    fr = mu.inputs[w0 ? 1 : 0];
                    chain.push_back(km);
                }
                // tf.signal.frame: Reshape [0|-1, S / sub, sub] -> Gather(axis 1, selector[f][j] = f * h + j) -> Reshape [0|-1, F, L]
                const int kr2 = sole_prod(fr, "Reshape");
                if (kr2 < 0) continue;
                std::vector<int64_t> shp2, shp0;
                if (!ints(nodes_[kr2].inputs[1], shp2) || shp2.size() != 3) continue;
                const int kg = sole_prod(nodes_[kr2].inputs[0], "Gather");
                if (kg < 0 || nodes_[kg].attr_i("axis", 0) != 1) continue;
                const Val *sel = cval(nodes_[kg].inputs[1]);
                if (!sel || !sel->is_int || sel->dims.size() != 2) continue;
                const int kr0 = sole_prod(nodes_[kg].inputs[0], "Reshape");
                if (kr0 < 0 || !ints(nodes_[kr0].inputs[1], shp0) || shp0.size() != 3) continue;
                const int64_t F = sel->dims[0], J = sel->dims[1], sub = shp0[2];
                if (sub < 1 || F < 1 || J < 1 || shp2[1] != F || shp2[2] != J * sub) continue;
                const int64_t h = F > 1 ? sel->i[(size_t)J] - sel->i[0] : 1;
                bool affine = sel->i[0] == 0 && h >= 1;
                for (int64_t f = 0; f < F && affine; f++)
                    for (int64_t j = 0; j < J && affine; j++) affine = sel->i[(size_t)(f * J + j)] == f * h + j;
                // (the Conv frames the WHOLE signal: the selector must take every frame that fits)
                if (!affine || shp0[1] < (F - 1) * h + J || F != (shp0[1] - J) / h + 1) continue;
                N = J * sub;
                hop = h * sub;
                if (!window.empty() && (int64_t)window.size() != N) continue;
                signal = nodes_[kr0].inputs[0];
                chain.insert(chain.end(), {kr2, kg, kr0});
            } else continue;
            if (hop < 1 || N > 8192) continue;
            const int64_t bins = N / 2 + 1;
            Val basis;
            basis.is_const = true;
            basis.dims = {bins, 1, N};
            basis.f.resize((size_t)(bins * N));
            for (int64_t k = 0; k < bins; k++)
                for (int64_t t = 0; t < N; t++) {
                    const double wv = window.empty() ? 1.0 : (double)window[(size_t)t];
                    const double th = 2.0 * M_PI * (double)((k * t) % N) / (double)N;
                    basis.f[(size_t)(k * N + t)] = (float)(ci[0] == 0 ? wv * std::cos(th) : -wv * std::sin(th));
                }
            const std::string base = "spectro" + std::to_string(serial++) + ":" + pick.outputs[0];
            vals_[base + "/basis"] = std::move(basis);
            vals_[base + "/shape"] = make_const_i({3}, {0, 1, -1});
            OnnxNode rs = synth_node("Reshape", base + "/signal", {signal, base + "/shape"}, base + "/signal");
            OnnxNode conv = synth_node("Conv", (pick.name.empty() ? pick.outputs[0] : pick.name) + (ci[0] == 0 ? "~re" : "~im"), {base + "/signal", base + "/basis"}, base + "/bank");
            set_ints(conv, "strides", {hop});
            set_ints(conv, "kernel_shape", {N});
            OnnxNode tr = synth_node("Transpose", base + "/t", {base + "/bank"}, pick.outputs[0]);
            set_ints(tr, "perm", {0, 2, 1});
            repl[gk] = {rs, conv, tr};
            dead[gk] = 1;
            for (int c : chain) dead[(size_t)c] = 1;
        }
        if (!serial) return;
        std::vector<OnnxNode> out;
        for (size_t k = 0; k < nodes_.size(); k++) {
            for (auto &r : repl[k]) out.push_back(r);
            if (!dead[k]) out.push_back(nodes_[k]);
        }
        nodes_ = std::move(out);
    }

    // ------------------------------------------------------------- pruning
    // Conv(out channels c) -> [Transpose] -> MatMul(const W[c, :]): output channels whose W row is
    // all zero never influence the result, drop them (mel filterbanks cover a fraction of the bins).
    void prune_zero_rows() {
        for (size_t k = 0; k < nodes_.size(); k++) {
            if (!live_[k] || nodes_[k].op_type != "Conv") continue;
            OnnxNode &conv = nodes_[k];
            int c1 = sole_consumer(conv.outputs[0]);
            if (c1 < 0) continue;
            int mm = -1;
            if (nodes_[c1].op_type == "Transpose") {
                auto perm = nodes_[c1].attr_ints("perm");
                if (perm != std::vector<int64_t>{0, 2, 1}) continue;
                mm = sole_consumer(nodes_[c1].outputs[0]);
            } else continue;
            if (mm < 0 || nodes_[mm].op_type != "MatMul") continue;
            auto wi = vals_.find(nodes_[mm].inputs[1]);
            auto cw = vals_.find(conv.inputs[1]);
            if (wi == vals_.end() || cw == vals_.end() || !wi->second.is_const || !cw->second.is_const) continue;
            Val &W = wi->second;
            Val &CW = cw->second;
            if (W.dims.size() != 2 || CW.dims.empty() || W.dims[0] != CW.dims[0] || W.is_int) continue;
            // weights must not be shared with other nodes
            if (live_consumers(nodes_[mm].inputs[1]).size() != 1 || live_consumers(conv.inputs[1]).size() != 1) continue;
            int64_t C = W.dims[0], N = W.dims[1];
            std::vector<int64_t> keep;
            for (int64_t c = 0; c < C; c++) {
                bool nz = false;
                for (int64_t j = 0; j < N && !nz; j++) nz = W.f[c * N + j] != 0.0f;
                if (nz) keep.push_back(c);
            }
            if ((int64_t)keep.size() == C || keep.empty()) continue;
            // keep a multiple of 4 channels (a few dead neighbours stay): the consumer GEMM's K is then a whole number
            // of float4 loads for both operands instead of scalar loads (K = 309 -> 312, 127 -> 128)
            {
                std::vector<char> kept((size_t)C, 0);
                for (int64_t c : keep) kept[(size_t)c] = 1;
                for (int64_t c = keep.back() + 1; c < C && keep.size() % 4; c++) { kept[(size_t)c] = 1; keep.push_back(c); }
                for (int64_t c = keep.front() - 1; c >= 0 && keep.size() % 4; c--) { kept[(size_t)c] = 1; keep.push_back(c); }
                std::sort(keep.begin(), keep.end());
                if ((int64_t)keep.size() == C) continue;
            }
            int64_t per = CW.numel() / C;
            std::vector<float> nw(keep.size() * per), nW(keep.size() * N);
            for (size_t r = 0; r < keep.size(); r++) {
                std::copy(CW.f.begin() + keep[r] * per, CW.f.begin() + (keep[r] + 1) * per, nw.begin() + r * per);
                std::copy(W.f.begin() + keep[r] * N, W.f.begin() + (keep[r] + 1) * N, nW.begin() + r * N);
            }
            CW.f = nw; CW.dims[0] = (int64_t)keep.size();
            W.f = nW; W.dims[0] = (int64_t)keep.size();
            if (conv.inputs.size() > 2 && !conv.inputs[2].empty()) {
                auto bi = vals_.find(conv.inputs[2]);
                if (bi != vals_.end() && bi->second.is_const && live_consumers(conv.inputs[2]).size() == 1) {
                    std::vector<float> nb(keep.size());
                    for (size_t r = 0; r < keep.size(); r++) nb[r] = bi->second.f[keep[r]];
                    bi->second.f = nb; bi->second.dims[0] = (int64_t)keep.size();
                }
            }
        }
    }

    // ------------------------------------------------------------- framing conv x constant product -> one framing conv (round 4)
    // Conv1D(one input channel, C filters of length L) -> Transpose -> MatMul(const W[C, N]) is ONE linear map of the frame when nothing
    // sits between the two (v2.4 takes the REAL PART of its STFT straight into the mel bank): N filters  sum_c W[c][n] w_c  of length L,
    // bias  sum_c W[c][n] b_c.  Worth it when N filters cost less than the C filters (as an FFT or as folded matrix products) plus the
    // product: v2.4's 312 mel-live bins of the 1024-point branch -> 96 mel filters of 512 folded taps (the rows stay symmetric about the
    // frame centre), against an FFT of every frame plus the mel phase; NOT its 127 bins of the 2048-point branch, which the quarter fold
    // evaluates in 512 taps each where 96 merged filters would need 1024.  Costs in f32-MFMA SIMD-cycles per frame, the scale emit_stft
    // calibrated.  The filters are summed in double and rounded once.  BN_CONVMERGE=0 disables, =1 merges wherever the pattern matches.
    void merge_framing_products() {
        const int mode = env_int("BN_CONVMERGE", -1);
        if (mode == 0) return;
        const std::string stft_mode = getenv("BN_STFT") ? getenv("BN_STFT") : "auto";
        if (stft_mode == "1" && mode != 1) return;  // every recognised bank as an FFT: the banks stay banks
        for (size_t k = 0; k < nodes_.size(); k++) {
            if (!live_[k] || nodes_[k].op_type != "Conv") continue;
            OnnxNode &conv = nodes_[k];
            const int c1 = sole_consumer(conv.outputs[0]);
            if (c1 < 0 || nodes_[c1].op_type != "Transpose" || nodes_[c1].attr_ints("perm") != std::vector<int64_t>{0, 2, 1}) continue;
            const int mm = sole_consumer(nodes_[c1].outputs[0]);
            if (mm < 0 || nodes_[mm].op_type != "MatMul" || nodes_[mm].inputs[0] != nodes_[c1].outputs[0]) continue;
            auto wi = vals_.find(nodes_[mm].inputs[1]);
            auto cw = vals_.find(conv.inputs[1]);
            if (wi == vals_.end() || cw == vals_.end() || !wi->second.is_const || !cw->second.is_const || wi->second.is_int || cw->second.is_int) continue;
            Val &W = wi->second;
            Val &CW = cw->second;
            if (W.dims.size() != 2 || CW.dims.size() != 3 || CW.dims[1] != 1 || W.dims[0] != CW.dims[0]) continue;  // [C, 1, L] x [C, N]
            if (live_consumers(nodes_[mm].inputs[1]).size() != 1 || live_consumers(conv.inputs[1]).size() != 1) continue;
            if (conv.attr_i("group", 1) != 1) continue;
            const int64_t C = CW.dims[0], L = CW.dims[2], N = W.dims[1];
            if (L < 128 || N >= C) continue;
            Val *bias = nullptr;
            if (conv.inputs.size() > 2 && !conv.inputs[2].empty()) {
                auto bi = vals_.find(conv.inputs[2]);
                if (bi == vals_.end() || !bi->second.is_const || bi->second.is_int || live_consumers(conv.inputs[2]).size() != 1) continue;
                bias = &bi->second;
            }
            // symmetry of the rows about the frame centre (the fold rule's test, its tolerance)
            float maxabs = 0.0f;
            for (float v : CW.f) maxabs = std::max(maxabs, std::fabs(v));
            const float eps = 1.1920929e-7f * maxabs;
            bool all_sym = L % 64 == 0, all_folded = L % 64 == 0;
            for (int64_t c = 0; c < C && all_folded; c++) {
                const float *w = &CW.f[(size_t)(c * L)];
                bool sym = std::fabs(w[0]) <= eps, anti = sym && std::fabs(w[L / 2]) <= eps;
                for (int64_t t = 1; t < L / 2 && (sym || anti); t++) {
                    if (!(std::fabs(w[t] - w[L - t]) <= eps)) sym = false;
                    if (!(std::fabs(w[t] + w[L - t]) <= eps)) anti = false;
                }
                all_sym = all_sym && sym;
                all_folded = all_folded && (sym || anti);
            }
            // separate: the cheapest form the bank may take + the product; merged: N filters, folded if every row is symmetric
            const bool fft_size = L <= 2048 && ((L & (L - 1)) == 0 || fft_five_pow2(L));
            double bank = (double)C * (double)(all_folded ? L / 2 : L) / 32.0;
            bool product_inside = false;
            if (all_sym && L % 128 == 0 && C <= 160) bank = std::min(bank, (double)C * (double)(L / 4 + 1) / 32.0);  // quarter fold
            if (all_folded && fft_size && stft_mode != "0") {
                const double fft = 0.18 * (double)L * std::log2((double)L);  // (with its mel phase: emit_stft's calibration)
                if (fft < bank) { bank = fft; product_inside = true; }
            }
            const double separate = bank + (product_inside ? 0.0 : (double)C * (double)N / 32.0);
            const double merged = (double)N * (double)(all_sym ? L / 2 : L) / 32.0;
            if (mode != 1 && !(merged < 0.9 * separate)) continue;
            std::vector<float> nw((size_t)(N * L));
            std::vector<double> acc((size_t)L);
            for (int64_t n2 = 0; n2 < N; n2++) {
                std::fill(acc.begin(), acc.end(), 0.0);
                for (int64_t c = 0; c < C; c++) {
                    const double m = (double)W.f[(size_t)(c * N + n2)];
                    if (m == 0.0) continue;
                    const float *w = &CW.f[(size_t)(c * L)];
                    for (int64_t t = 0; t < L; t++) acc[(size_t)t] += m * (double)w[t];
                }
                for (int64_t t = 0; t < L; t++) nw[(size_t)(n2 * L + t)] = (float)acc[(size_t)t];
                if (all_sym) {  // exactly symmetric again after the rounding (the fold keeps the first half)
                    nw[(size_t)(n2 * L)] = 0.0f;
                    for (int64_t t = 1; t < L / 2; t++) nw[(size_t)(n2 * L + L - t)] = nw[(size_t)(n2 * L + t)];
                }
            }
            if (bias) {
                std::vector<float> nb((size_t)N);
                for (int64_t n2 = 0; n2 < N; n2++) {
                    double a = 0.0;
                    for (int64_t c = 0; c < C; c++) a += (double)W.f[(size_t)(c * N + n2)] * (double)bias->f[(size_t)c];
                    nb[(size_t)n2] = (float)a;
                }
                bias->f = nb;
                bias->dims[0] = N;
            }
            CW.f = nw;
            CW.dims[0] = N;
            // the product is gone: its node hands the transposed conv result on
            OnnxNode &prod = nodes_[mm];
            prod.op_type = "Identity";
            prod.inputs.resize(1);
        }
    }

    // ------------------------------------------------------------- windowed-DFT filter banks -> real FFT
    // A single-channel 1-D filter bank whose every row is  a_c * w[n] * cos(2 pi k_c n / L)  (symmetric rows) or
    // b_c * w[n] * sin(2 pi k_c n / L)  (antisymmetric rows) for ONE common window w and integer bins k_c is a
    // short-time Fourier transform: out[c] = a_c Re Y[k_c] - b_c Im Y[k_c] with Y = FFT(w . frame).  The bank is
    // recognised from the numbers alone (the exporter's node names say nothing): bins by the peak of each row's
    // spectrum, window and amplitudes by alternating least squares over all rows, and the model is accepted only if
    // every tap of every row agrees with it within f32 rounding of the taps (BN_STFT_TOL x max|row|, default 4e-7).
    // Then ONE launch of stft_kernel (kernels.h, FftDesc) computes all rows -- cos and sin blocks together -- instead
    // of one folded GEMM per symmetry run.  L must be a power of two in 128..2048, or five times a power of two (Perch's L = 640:
    // mixed radix 5 x 4 x 16).  Whether a recognised bank runs as an FFT: emit_stft.
    struct DftBank {
        int64_t L = 0;
        std::vector<float> window;
        std::vector<int> k;
        std::vector<double> a, b;
    };
    static void host_fft(std::vector<double> &re, std::vector<double> &im) {  // in-place radix-2, n a power of two
        const size_t n = re.size();
        for (size_t i = 1, j = 0; i < n; i++) {
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            const double ang = -2.0 * M_PI / (double)len;
            for (size_t i = 0; i < n; i += len)
                for (size_t k = 0; k < len / 2; k++) {
                    const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
                    const size_t u = i + k, v = i + k + len / 2;
                    const double xr = re[v] * wr - im[v] * wi, xi = re[v] * wi + im[v] * wr;
                    re[v] = re[u] - xr; im[v] = im[u] - xi;
                    re[u] += xr; im[u] += xi;
                }
        }
    }
    // L = 10 * 4^j * 16 ... in the kernel's terms: M = L / 2 = 5 q with q = 4^j * 16 a power of FOUR times 16 (radix 5 first, then
    // radix-4 passes down to the 16-point register blocks; a radix-2 pass exists only as a FIRST pass) -- 640 is the one in use
    static bool fft_five_pow2(int64_t L) {
        if (L % 10) return false;
        int64_t q = L / 10;
        if (q < 64 || (q & (q - 1))) return false;
        int lg = 0;
        while (((int64_t)1 << lg) < q) lg++;
        return lg % 2 == 0;  // q = 16 * 4^j
    }
    // bins 0 .. n/2 of the DFT of a real row by the definition (any n; the peak search below reads no other bin)
    static void host_dft_half(std::vector<double> &re, std::vector<double> &im) {
        const size_t n = re.size();
        std::vector<double> c(n), s_(n), xr(re);
        for (size_t t = 0; t < n; t++) { c[t] = std::cos(2.0 * M_PI * (double)t / (double)n); s_[t] = std::sin(2.0 * M_PI * (double)t / (double)n); }
        for (size_t k = 0; k <= n / 2; k++) {
            double ar = 0, ai = 0;
            for (size_t t = 0; t < n; t++) {
                const size_t ph = (k * t) % n;
                ar += xr[t] * c[ph];
                ai -= xr[t] * s_[ph];
            }
            re[k] = ar; im[k] = ai;
        }
    }
    bool detect_dft_bank(const std::vector<float> &wf, int64_t Cout, int64_t L, const std::vector<int> &cls, DftBank &bank) {
        const bool pow2 = (L & (L - 1)) == 0;
        if (L < 128 || L > 2048 || !(pow2 || fft_five_pow2(L))) return false;
        const double tol = getenv("BN_STFT_TOL") ? atof(getenv("BN_STFT_TOL")) : 4e-7;
        bank.L = L;
        bank.k.assign((size_t)Cout, 0);
        bank.a.assign((size_t)Cout, 0.0);
        bank.b.assign((size_t)Cout, 0.0);
        std::vector<double> rowmax((size_t)Cout, 0.0);
        std::vector<char> is_sin((size_t)Cout, 0), live((size_t)Cout, 0);
        double gmax = 0;
        for (float v : wf) gmax = std::max(gmax, (double)std::fabs(v));
        for (int64_t c = 0; c < Cout; c++) {
            const float *w = &wf[(size_t)(c * L)];
            double mx = 0;
            for (int64_t n = 0; n < L; n++) mx = std::max(mx, (double)std::fabs(w[n]));
            rowmax[(size_t)c] = mx;
            // a row whose largest tap is below f32 resolution of the bank's largest tap (sin(pi n) of the Nyquist bin
            // evaluated in floating point: ~1e-13) counts as the zero row it stands for
            if (!(mx > 1e-7 * gmax)) continue;
            live[(size_t)c] = 1;
            is_sin[(size_t)c] = cls[(size_t)c] < 0;
            std::vector<double> re(w, w + L), im((size_t)L, 0.0);
            if (pow2) host_fft(re, im);
            else host_dft_half(re, im);
            int best = 0;
            double bm = -1;
            for (int64_t q = 0; q <= L / 2; q++) {
                const double m2 = re[(size_t)q] * re[(size_t)q] + im[(size_t)q] * im[(size_t)q];
                if (m2 > bm) { bm = m2; best = (int)q; }
            }
            bank.k[(size_t)c] = best;
        }
        // unit sinusoids of every live row, from one table of cos(2 pi t / L)
        std::vector<double> ctab((size_t)L);
        for (int64_t t = 0; t < L; t++) ctab[(size_t)t] = std::cos(2.0 * M_PI * (double)t / (double)L);
        auto trig = [&](int64_t c, int64_t n) {
            const int64_t ph = ((int64_t)bank.k[(size_t)c] * n) % L;  // exact phase reduction
            return is_sin[(size_t)c] ? ctab[(size_t)((ph + 3 * L / 4) % L)] : ctab[(size_t)ph];  // sin x = cos(x - pi/2)
        };
        std::vector<double> amp((size_t)Cout, 0.0), win((size_t)L, 0.0), energy((size_t)Cout, 0.0);
        for (int64_t c = 0; c < Cout; c++) {
            if (!live[(size_t)c]) continue;
            double e = 0;
            for (int64_t n = 0; n < L; n++) e += (double)wf[(size_t)(c * L + n)] * wf[(size_t)(c * L + n)];
            energy[(size_t)c] = e;
        }
        // start from the row energy: sum (w t)^2 is ~ sum w^2 / 2 for a sinusoid, sum w^2 for the DC / Nyquist rows
        // (the sign settles in the first amplitude step)
        auto init_amp = [&](int64_t c) {
            const int k = bank.k[(size_t)c];
            amp[(size_t)c] = std::sqrt(energy[(size_t)c] * ((k == 0 || k == L / 2) ? 1.0 : 2.0));
        };
        std::vector<char> use((size_t)Cout, 0);
        auto als = [&](int max_it) {
            for (int it = 0; it < max_it; it++) {
                for (int64_t n = 0; n < L; n++) {
                    double num = 0, den = 0;
                    for (int64_t c = 0; c < Cout; c++) {
                        if (!use[(size_t)c]) continue;
                        const double t = amp[(size_t)c] * trig(c, n);
                        num += t * (double)wf[(size_t)(c * L + n)];
                        den += t * t;
                    }
                    win[(size_t)n] = den > 0 ? num / den : 0.0;
                }
                double wmax = 0;
                for (double v : win) wmax = std::max(wmax, std::fabs(v));
                if (!(wmax > 0)) return false;
                for (double &v : win) v /= wmax;
                double change = 0;
                for (int64_t c = 0; c < Cout; c++) {
                    if (!use[(size_t)c]) continue;
                    double num = 0, den = 0;
                    for (int64_t n = 0; n < L; n++) {
                        const double t = win[(size_t)n] * trig(c, n);
                        num += t * (double)wf[(size_t)(c * L + n)];
                        den += t * t;
                    }
                    if (!(den > 0)) return false;  // e.g. a sine row at bin 0: identically zero model, non-zero row
                    const double na = num / den;
                    change = std::max(change, std::fabs(na - amp[(size_t)c]) / std::max(std::fabs(na), 1e-300));
                    amp[(size_t)c] = na;
                }
                if (it > 0 && change < 1e-11) break;
            }
            return true;
        };
        // stage 1: rows whose spectral peak is unambiguous (bins 3 .. L/2 - 3: near DC and Nyquist the two mirror
        // lobes of a windowed sinusoid overlap and the peak can sit one bin off) give a first window estimate
        int64_t n_sure = 0;
        for (int64_t c = 0; c < Cout; c++) {
            use[(size_t)c] = live[(size_t)c] && bank.k[(size_t)c] >= 3 && bank.k[(size_t)c] <= L / 2 - 3;
            n_sure += use[(size_t)c];
            if (live[(size_t)c]) init_amp(c);
        }
        if (n_sure == 0)
            for (int64_t c = 0; c < Cout; c++) use[(size_t)c] = live[(size_t)c];
        // The amplitude SIGNS are unknown at this point (exporters write -sin rows next to +cos rows: with all-positive
        // starting amplitudes the cos^2 and sin^2 contributions to a least-squares window cancel), so the first window
        // comes from energies, which do not see signs: sum_c R_c[n]^2 = w[n]^2 sum_c a_c^2 t_c[n]^2, w >= 0 as every
        // analysis window in use is.  The first amplitude step then settles the signs.
        {
            double wmax = 0;
            for (int64_t n = 0; n < L; n++) {
                double num = 0, den = 0;
                for (int64_t c = 0; c < Cout; c++) {
                    if (!use[(size_t)c]) continue;
                    const double t = amp[(size_t)c] * trig(c, n), r = (double)wf[(size_t)(c * L + n)];
                    num += r * r;
                    den += t * t;
                }
                win[(size_t)n] = den > 0 ? std::sqrt(num / den) : 0.0;
                wmax = std::max(wmax, win[(size_t)n]);
            }
            if (!(wmax > 0)) return false;
            for (double &v : win) v /= wmax;
            for (int64_t c = 0; c < Cout; c++) {
                if (!use[(size_t)c]) continue;
                double num = 0, den = 0;
                for (int64_t n = 0; n < L; n++) {
                    const double t = win[(size_t)n] * trig(c, n);
                    num += t * (double)wf[(size_t)(c * L + n)];
                    den += t * t;
                }
                if (!(den > 0)) return false;
                amp[(size_t)c] = num / den;
            }
        }
        if (!als(3)) return false;
        // stage 2: every row's bin re-decided among the neighbours of its peak by the least-squares residual against
        // that window
        for (int64_t c = 0; c < Cout; c++) {
            if (!live[(size_t)c]) continue;
            const int k0 = bank.k[(size_t)c];
            int best = k0;
            double best_res = INFINITY;
            for (int k = std::max(0, k0 - 2); k <= std::min<int>((int)(L / 2), k0 + 2); k++) {
                bank.k[(size_t)c] = k;
                double num = 0, den = 0;
                for (int64_t n = 0; n < L; n++) {
                    const double t = win[(size_t)n] * trig(c, n);
                    num += t * (double)wf[(size_t)(c * L + n)];
                    den += t * t;
                }
                const double res = den > 0 ? energy[(size_t)c] - num * num / den : INFINITY;
                if (res < best_res) { best_res = res; best = k; }
            }
            if (getenv("BN_STFT_DEBUG") && (c < 4 || best != k0))
                fprintf(stderr, "stft: row %lld %s peak bin %d -> %d (residual %.3g of energy %.3g)\n", (long long)c, is_sin[(size_t)c] ? "sin" : "cos", k0, best,
                        best_res, energy[(size_t)c]);
            bank.k[(size_t)c] = best;
            {  // amplitude (with its sign) against the stage-1 window
                double num = 0, den = 0;
                for (int64_t n = 0; n < L; n++) {
                    const double t = win[(size_t)n] * trig(c, n);
                    num += t * (double)wf[(size_t)(c * L + n)];
                    den += t * t;
                }
                if (!(den > 0)) return false;
                amp[(size_t)c] = num / den;
            }
            use[(size_t)c] = 1;
        }
        // stage 3: all rows
        if (!als(40)) return false;
        // the window in f32 (what the kernel multiplies by); amplitudes refitted against the rounded window, then verify
        bank.window.resize((size_t)L);
        for (int64_t n = 0; n < L; n++) bank.window[(size_t)n] = (float)win[(size_t)n];
        for (int64_t c = 0; c < Cout; c++) {
            if (!live[(size_t)c]) continue;
            for (int64_t n = 0; n < L; n++) {
                const double model = amp[(size_t)c] * win[(size_t)n] * trig(c, n);
                if (!(std::fabs(model - (double)wf[(size_t)(c * L + n)]) <= tol * rowmax[(size_t)c])) {
                    if (getenv("BN_STFT_DEBUG"))
                        fprintf(stderr, "stft: row %lld (bin %d, %s) tap %lld: model %.9g vs %.9g (row max %.3g)\n", (long long)c, bank.k[(size_t)c],
                                is_sin[(size_t)c] ? "sin" : "cos", (long long)n, model, (double)wf[(size_t)(c * L + n)], rowmax[(size_t)c]);
                    return false;
                }
            }
            if (is_sin[(size_t)c]) bank.b[(size_t)c] = amp[(size_t)c];
            else bank.a[(size_t)c] = amp[(size_t)c];
        }
        return true;
    }
    // digit-reversed position of output k of an in-place DIF transform with the given radix sequence
    static int dif_position(int k, int n, const std::vector<int> &radix, size_t idx) {
        if (n == 1) return 0;
        const int r = radix[idx];
        return (k % r) * (n / r) + dif_position(k / r, n / r, radix, idx + 1);
    }
    bool emit_stft(const OnnxNode &n, const PlanOp &base, const DftBank &bank, int64_t Cout, int64_t OW, bool has_bias) {
        // Which banks run as FFTs.  The alternative is the pruned + folded matrix product on the exact-f32 MFMA path, whose
        // cost follows the number of LIVE bins: Cout * L/2 multiply-adds at 32 per SIMD-cycle, against roughly
        // 0.18 * L * log2 L cycles of vector butterflies, address arithmetic and LDS traffic for the FFT plus its untangle
        // and mel phases (both calibrated on the v2.4 banks, DESIGN.md section 4.11: L = 2048 with 127 live bins -- matrix
        // product 69 us, FFT 76-93 us; L = 1024 with 309 live bins -- 85 us vs 62-75 us).  Default ("auto"): FFT when the
        // matrix product is estimated at more than 1.5x the FFT.  BN_STFT=1: every recognised bank; BN_STFT=0: none;
        // BN_STFT_MINBINS=<n> additionally keeps banks with fewer live bins on the matrix path.
        const char *env = getenv("BN_STFT");
        const std::string mode = env ? env : "auto";
        if (mode == "0") return false;
        if (getenv("BN_STFT_MINBINS") && Cout < atoll(getenv("BN_STFT_MINBINS"))) return false;
        if (mode != "1") {
            const double lg = std::log2((double)bank.L);
            const double gemm_cycles = (double)Cout * (double)(bank.L / 2) / 32.0, fft_cycles = 0.18 * (double)bank.L * lg;
            if (!(gemm_cycles > 1.5 * fft_cycles)) return false;
        }
        const GemmDesc &g = base.gemm;
        if (g.act != ACT_NONE || g.has_res || g.has_scale || g.lda <= 0 || g.lda > 4096 || g.ldc != Cout) return false;
        const int L = (int)bank.L, M = L / 2;
        PlanOp op;
        op.kind = OpKind::FFT;
        op.name = "stft:" + n.name;
        op.out = base.out;
        op.a = base.a;
        op.bias = base.bias;
        FftDesc &d = op.fft;
        d.L = L; d.M = M; d.hop = (int32_t)g.lda; d.frames = (int32_t)OW; d.nout = (int32_t)Cout;
        d.logM = 0;
        while ((1 << d.logM) < M) d.logM++;
        d.F = 1024 / M;
        const bool five = M % 5 == 0;  // M = 5 q (detect_dft_bank: q = 16 * 4^j): three frames of 320 slots per wave pass
        // frames per tile: as many as keep the tile's signal span within the kernel's staging registers (8192 floats)
        if (five) {
            // whole wave passes, one per wave of the block where the span allows: 8 waves x F frames
            d.tpb = 8 * d.F;
            while (d.tpb > d.F && (int64_t)(d.tpb - 1) * g.lda + L > 8192) d.tpb -= d.F;
        } else {
            d.tpb = 16;
            while (d.tpb > d.F && (int64_t)(d.tpb - 1) * g.lda + L > 8192) d.tpb /= 2;
        }
        if ((int64_t)(d.tpb - 1) * g.lda + L > 8192 || d.tpb % d.F) return false;
        d.a_bs = g.a_bs; d.ldc = g.ldc; d.c_bs = g.c_bs; d.has_bias = has_bias ? 1 : 0;
        d.out_rs = g.ldc; d.out_cs = 1;
        d.power = 0; d.otab_stride = 8;
        // pass structure: radix 5 first for M = 5 q, radix 2 first when log2 M is odd, radix 4 down to 16-point blocks (registers)
        std::vector<int> radix;
        std::vector<float> tw;
        int nn = M;
        d.npass = 0;
        while (nn > 16) {
            const int r = (d.npass == 0 && five) ? 5 : (d.npass == 0 && (d.logM & 1)) ? 2 : 4;
            if (nn % r) return false;
            const int q = nn / r;
            d.pass_n[d.npass] = nn; d.pass_r[d.npass] = r; d.pass_tw[d.npass] = (int32_t)(tw.size() / 2);
            for (int pw = 1; pw < r; pw++)
                for (int j = 0; j < q; j++) {
                    const double ang = -2.0 * M_PI * (double)((int64_t)j * pw % nn) / (double)nn;
                    tw.push_back((float)std::cos(ang));
                    tw.push_back((float)std::sin(ang));
                }
            radix.push_back(r);
            nn = q;
            d.npass++;
        }
        if (nn != 16 || d.npass > 4) return false;
        radix.push_back(4);
        radix.push_back(4);
        d.tw_count = (int32_t)(tw.size() / 2);
        if (stft_lds_bytes(d, 8) > 156 * 1024) return false;
        // per-output table
        std::vector<float> otab((size_t)Cout * 8, 0.0f);  // per output: positions of Z[k], Z[M-k] (as floats), 4 coefficients, 2 pad
        auto physpos = [&](int k) {
            const int pp = dif_position(k, M, radix, 0);
            return pp + 2 * (pp >> 4);
        };
        for (int64_t c = 0; c < Cout; c++) {
            const int k = bank.k[(size_t)c];
            const double a = bank.a[(size_t)c], b = bank.b[(size_t)c];
            const double th = 2.0 * M_PI * (double)k / (double)L, cs = std::cos(th), sn = std::sin(th);
            float *e = &otab[(size_t)c * 8];
            e[0] = (float)physpos(k % M);
            e[1] = (float)physpos((M - k % M) % M);
            e[2] = (float)(0.5 * (a * (1.0 - sn) + b * cs));
            e[3] = (float)(0.5 * (a * cs - b * (1.0 - sn)));
            e[4] = (float)(0.5 * (a * (1.0 + sn) - b * cs));
            e[5] = (float)(0.5 * (a * cs + b * (1.0 + sn)));
        }
        op.w = Ref{Space::CONSTS, add_const(bank.window), 0};
        op.w2 = Ref{Space::CONSTS, add_const(tw), 0};
        op.bias2 = Ref{Space::CONSTS, add_const(otab), 0};
        const double log2L = std::log2((double)L);
        op.flops_fft = (double)OW * 2.5 * L * log2L;
        op.macs = op.flops_fft / 2;  // multiply-add equivalents actually performed (vector ALU)
        op.mfma = false;
        op.weight_bytes = 4.0 * (bank.window.size() + tw.size() + otab.size() + (has_bias ? Cout : 0));
        op.bytes = base.bytes;
        dft_gemm_macs_ += (double)OW * Cout * L;  // what the matrix-product evaluation of this bank multiplies
        dft_fft_equiv_flops_ += op.flops_fft;
        dft_performed_macs_ += op.flops_fft / 2;
        push_op(std::move(op));
        return true;
    }

    // ------------------------------------------------------------- quarter-folded cosine banks (round 4)
    // A recognised bank of windowed COSINES only (the real part of an STFT) with a window symmetric about the frame centre: the even
    // bins see the frame through S[n] = ye[n] + ye[L/2 - n], the odd ones through D[n] = ye[n] - ye[L/2 - n], n <= L/4 (kernels.h,
    // GemmDesc::fold == 2; kernels.hip, frame_fold2_kernel) -- L/4 + 1 taps per output instead of L/2.  The filter rows become pure
    // cosines x the fitted row amplitude (double -> f32), the window moves to the signal (two tables), the columns are grouped (even bins,
    // padded to 32, then odd bins) and a column map sends every result to its output channel.  v2.4's 127 mel-live bins of the 2048-point
    // branch: 17 K steps instead of 32.  BN_CONVFOLD2=0 keeps the half fold.
    bool emit_quarter_fold(const OnnxNode &n, const PlanOp &base, const DftBank &bank, const std::vector<int> &cls, int64_t Cout, int64_t OW, bool has_bias) {
        if (env_int("BN_CONVFOLD2", 1) == 0) return false;
        if (env_int("BN_FRAMEPAIR", 0) == 1) return false;  // (opt-in rule J fuses the mel product behind the HALF fold's launch)
        const int64_t L = bank.L;
        if (L % 128 != 0) return false;
        for (int64_t c = 0; c < Cout; c++)
            if (cls[(size_t)c] < 0 || bank.b[(size_t)c] != 0.0) return false;  // a sine row: the half fold (or the FFT) keeps the bank
        // the window: symmetric about the centre, nothing at tap 0 (the fold check saw that on the rows; here on the fitted window itself)
        // (a tap at which every row's cosine vanishes -- n = L/4 for a bank of odd bins -- is invisible in the rows: the fitted window holds
        // noise there, and nothing it multiplies reaches an output.  Such taps count as zero.)
        std::vector<float> w = bank.window;
        {
            std::vector<double> seen((size_t)L, 0.0);
            double smax = 0;
            for (int64_t t = 0; t < L; t++) {
                for (int64_t c = 0; c < Cout; c++) {
                    const double v = bank.a[(size_t)c] * std::cos(2.0 * M_PI * (double)((bank.k[(size_t)c] * t) % L) / (double)L);
                    seen[(size_t)t] += v * v;
                }
                smax = std::max(smax, seen[(size_t)t]);
            }
            for (int64_t t = 0; t < L; t++)
                if (!(seen[(size_t)t] > 1e-20 * smax)) w[(size_t)t] = 0.0f;
        }
        float wmax = 0.0f;
        for (float v : w) wmax = std::max(wmax, std::fabs(v));
        const float eps = 4e-7f * wmax;
        const bool dbg = getenv("BN_STFT_DEBUG") != nullptr;
        if (!(std::fabs(w[0]) <= eps)) {
            if (dbg) fprintf(stderr, "quarter fold: window tap 0 = %.3g\n", (double)w[0]);
            return false;
        }
        for (int64_t t = 1; t < L / 2; t++)
            if (!(std::fabs(w[(size_t)t] - w[(size_t)(L - t)]) <= eps)) {
                if (dbg) fprintf(stderr, "quarter fold: window not symmetric at tap %lld: %.9g vs %.9g\n", (long long)t, (double)w[(size_t)t], (double)w[(size_t)(L - t)]);
                return false;
            }
        std::vector<int32_t> even, odd, dead;
        for (int64_t c = 0; c < Cout; c++) (bank.a[(size_t)c] == 0.0 ? dead : (bank.k[(size_t)c] & 1) ? odd : even).push_back((int32_t)c);
        for (int32_t c : dead) {  // an all-zero row (its output is the bias) joins the group with room in its last wave column
            const size_t se = (32 - even.size() % 32) % 32, so = (32 - odd.size() % 32) % 32;
            (so > se ? odd : even).push_back(c);
        }
        const int64_t ne = ((int64_t)even.size() + 31) / 32 * 32, no = ((int64_t)odd.size() + 31) / 32 * 32, N2 = ne + no;
        const int64_t Q = L / 4, K2 = Q + 32;
        PlanOp f = base;
        f.gemm.N = (int32_t)N2;
        f.gemm.K = (int32_t)K2;
        f.gemm.fold = 2;
        f.gemm.fold_n = (int32_t)L;
        f.gemm.fold_ne = (int32_t)ne;
        if (!frame_fold2_shape_ok(f.gemm, nullptr)) {
            if (dbg) fprintf(stderr, "quarter fold: shape outside the kernel (N = %lld + %lld columns, %zu bytes of LDS)\n", (long long)ne, (long long)no, frame_fold2_lds_bytes(f.gemm));
            return false;
        }
        std::vector<float> colmap_bits((size_t)N2), wk((size_t)(N2 * K2), 0.0f), tabs((size_t)(2 * K2), 0.0f);
        std::vector<int32_t> colmap((size_t)N2, -1);
        for (size_t i = 0; i < even.size(); i++) colmap[i] = even[i];
        for (size_t i = 0; i < odd.size(); i++) colmap[(size_t)ne + i] = odd[i];
        static_assert(sizeof(int32_t) == sizeof(float), "the column map travels in a constant of floats");
        std::memcpy(colmap_bits.data(), colmap.data(), (size_t)N2 * sizeof(float));
        std::vector<double> ctab((size_t)L);
        for (int64_t t = 0; t < L; t++) ctab[(size_t)t] = std::cos(2.0 * M_PI * (double)t / (double)L);
        for (int64_t col = 0; col < N2; col++) {
            const int32_t c = colmap[(size_t)col];
            if (c < 0) continue;
            const int64_t k = bank.k[(size_t)c];
            const double a = bank.a[(size_t)c];
            float *row = &wk[(size_t)(col * K2)];
            for (int64_t t = 0; t < Q; t++) row[t] = (float)(a * ctab[(size_t)((k * t) % L)]);
            row[Q] = (k & 1) ? 0.0f : (float)(a * ctab[(size_t)((k * Q) % L)]);  // cos(pi k / 2) = +-1 for even k
        }
        // S[n] = wa[n] (x[n] + x[L-n]) + wb[n] (x[L/2-n] + x[L/2+n]); n = 0: ye[0] = 0 and ye[L/2] = y[L/2] once; n = L/4: ye[L/4] once
        for (int64_t t = 1; t < Q; t++) {
            tabs[(size_t)t] = w[(size_t)t];
            tabs[(size_t)(K2 + t)] = w[(size_t)(L / 2 - t)];
        }
        tabs[0] = 0.0f;
        tabs[(size_t)K2] = 0.5f * w[(size_t)(L / 2)];
        tabs[(size_t)Q] = w[(size_t)Q];
        tabs[(size_t)(K2 + Q)] = 0.0f;
        f.name = "Conv:" + n.name + "~quarter";
        if (frame_fold2q_ok(f.gemm)) {  // (round 5) half-height blocks on the bf16 matrix pipe: three exact bf16 planes per filter element, fragment order
            const int64_t steps = K2 / 32;
            std::vector<uint16_t> img((size_t)(N2 * K2 * 3), 0);  // [column group][step][16-wide k group][plane][lane][8]
            for (int64_t wn = 0; wn < N2 / 32; wn++)
                for (int64_t ks = 0; ks < steps; ks++)
                    for (int64_t g = 0; g < 2; g++)
                        for (int64_t lane = 0; lane < 64; lane++) {
                            const int64_t lr = lane & 31, lh = lane >> 5;
                            for (int64_t j = 0; j < 8; j++) {
                                uint16_t h, m, l;
                                split_bf16x3(wk[(size_t)((wn * 32 + lr) * K2 + ks * 32 + 16 * g + 8 * lh + j)], h, m, l);
                                const size_t at = (size_t)(((((wn * steps + ks) * 2 + g) * 3) * 64 + lane) * 8 + j);
                                img[at] = h;
                                img[at + 64 * 8] = m;
                                img[at + 2 * 64 * 8] = l;
                            }
                        }
            std::vector<float> pk(img.size() / 2);
            std::memcpy(pk.data(), img.data(), img.size() * sizeof(uint16_t));
            wk.swap(pk);
            f.gemm.fold_wpk = 2;
        } else if (frame_fold2p_ok(f.gemm)) {  // half-height blocks: the filter fragments in the order the waves load them
            const int64_t steps = K2 / 32;
            std::vector<float> pk((size_t)(N2 * K2));
            for (int64_t wn = 0; wn < N2 / 32; wn++)
                for (int64_t ks = 0; ks < steps; ks++)
                    for (int64_t g = 0; g < 4; g++)
                        for (int64_t lane = 0; lane < 64; lane++) {
                            const int64_t lr = lane & 31, lh = lane >> 5;
                            for (int64_t j = 0; j < 4; j++)
                                pk[(size_t)((((wn * steps + ks) * 4 + g) * 64 + lane) * 4 + j)] = wk[(size_t)((wn * 32 + lr) * K2 + ks * 32 + 8 * g + 4 * lh + j)];
                        }
            wk.swap(pk);
            f.gemm.fold_wpk = 1;
        }
        f.w = Ref{Space::CONSTS, add_const(wk), 0};
        f.w2 = Ref{Space::CONSTS, add_const(tabs), 0};
        f.bias2 = Ref{Space::CONSTS, add_const(colmap_bits), 0};
        f.macs = (double)OW * (double)Cout * (double)(Q + 1);  // multiplications actually performed (padding columns excluded)
        f.weight_bytes = 4.0 * (wk.size() + tabs.size() + colmap.size() + (has_bias ? Cout : 0));
        dft_gemm_macs_ += (double)OW * Cout * L;
        dft_performed_macs_ += f.macs;
        dft_fft_equiv_flops_ += (double)OW * 2.5 * (double)L * std::log2((double)L);
        push_op(std::move(f));
        return true;
    }

    // ------------------------------------------------------------- folded framing convolutions
    // A long single-channel 1-D filter bank whose rows are symmetric (w[n] == w[L-n], windowed cosine bases) or
    // antisymmetric (w[n] == -w[L-n], windowed sine bases) about the frame centre, with w[0] == 0 (Hann-type
    // windows), needs only half the multiplications:
    //     sum_n w[n] x[n] = sum_{n=1}^{L/2-1} w[n] (x[n] +- x[L-n]) + w[L/2] x[L/2]
    // The GEMM stages the folded frame rows (kernels.h, GemmDesc::fold) and K drops from L to L/2.  Rows count as
    // (anti)symmetric when the two halves agree within BN_CONVFOLD_TOL x max|w| (default 2^-23: one unit in the
    // last place of the largest tap -- f32-rounded DFT bases differ from their mirror image by that much in
    // ~1 % of the taps); the first-half taps are the ones kept.  Maximal runs of equal symmetry become one launch
    // each (cos block, sin block), writing column slices of the same output.  BN_CONVFOLD=0 disables.
    bool fold_framing_conv(const OnnxNode &n, const PlanOp &base, const std::vector<float> &wf, int64_t Cout, int64_t L, int64_t OW,
                           bool has_bias) {
        const char *env = getenv("BN_CONVFOLD");
        if (env && std::string(env) == "0") return false;
        if (L < 128 || L % 64 != 0) return false;  // K = L/2 must be a whole number of 32-wide K steps
        const double tol = getenv("BN_CONVFOLD_TOL") ? atof(getenv("BN_CONVFOLD_TOL")) : 1.1920929e-7;
        float maxabs = 0.0f;
        for (float v : wf) maxabs = std::max(maxabs, std::fabs(v));
        if (!(maxabs > 0.0f) || !std::isfinite(maxabs)) return false;
        const float eps = (float)(tol * maxabs);
        std::vector<int> cls((size_t)Cout);
        for (int64_t o = 0; o < Cout; o++) {
            const float *w = &wf[(size_t)(o * L)];
            if (!(std::fabs(w[0]) <= eps)) return false;
            bool sym = true, anti = std::fabs(w[L / 2]) <= eps;
            for (int64_t k = 1; k < L / 2 && (sym || anti); k++) {
                if (!(std::fabs(w[k] - w[L - k]) <= eps)) sym = false;
                if (!(std::fabs(w[k] + w[L - k]) <= eps)) anti = false;
            }
            if (!sym && !anti) return false;
            cls[(size_t)o] = (sym && anti) ? 0 : (sym ? 1 : -1);  // 0: an all-zero row joins either neighbour
        }
        {
            DftBank bank;
            if (detect_dft_bank(wf, Cout, L, cls, bank)) {
                if (emit_stft(n, base, bank, Cout, OW, has_bias)) return true;
                if (emit_quarter_fold(n, base, bank, cls, Cout, OW, has_bias)) return true;
            }
        }
        struct Run { int64_t n0, n1; int sign; };
        std::vector<Run> runs;
        for (int64_t o = 0; o < Cout; o++) {
            const int c = cls[(size_t)o];
            if (!runs.empty() && (c == 0 || runs.back().sign == 0 || c == runs.back().sign)) {
                runs.back().n1 = o + 1;
                if (runs.back().sign == 0) runs.back().sign = c;
            } else runs.push_back(Run{o, o + 1, c});
        }
        if (runs.size() > 4) return false;
        dft_gemm_macs_ += (double)OW * Cout * L;
        dft_performed_macs_ += (double)OW * Cout * (L / 2);  // folded: half the taps
        dft_fft_equiv_flops_ += (double)OW * 2.5 * (double)L * std::log2((double)L);  // the same frames as real FFTs (SURVEY 8(d) pricing, any L)
        const int64_t K = L / 2;
        for (const Run &r : runs) {
            const int sign = r.sign == 0 ? 1 : r.sign;
            const int64_t nn = r.n1 - r.n0;
            std::vector<float> wk((size_t)(nn * K));
            for (int64_t o = 0; o < nn; o++) {
                const float *w = &wf[(size_t)((r.n0 + o) * L)];
                for (int64_t c = 0; c + 1 < K; c++) wk[(size_t)(o * K + c)] = w[c + 1];
                wk[(size_t)(o * K + K - 1)] = sign > 0 ? 0.5f * w[K] : 0.0f;  // centre tap: its "pair" is itself
            }
            PlanOp f = base;
            f.name = "Conv:" + n.name + (sign > 0 ? "~sym" : "~anti");
            f.out.offset += r.n0;
            if (has_bias) f.bias.offset += r.n0;
            f.w = Ref{Space::CONSTS, add_const(wk), 0};
            f.gemm.N = (int32_t)nn;
            f.gemm.K = (int32_t)K;
            f.gemm.fold = sign;
            f.gemm.fold_n = (int32_t)L;
            f.macs = (double)OW * nn * K;  // multiplications actually performed
            f.weight_bytes = 4.0 * (wk.size() + (has_bias ? nn : 0));
            f.bytes = base.bytes * (double)nn / (double)Cout;
            push_op(std::move(f));
        }
        return true;
    }

    // Fused MBConv blocks the row-streaming kernel takes (mbrow.hip: one wave per (band of output rows, strip of output
    // columns, 32 mid channels), nothing in LDS): tiles_x / tiles_y become strips / bands -- the squeeze partials follow --
    // and `halo` the pixels it expands per sample.  Per-sample shapes only, so a segment's bits do not depend on its batch.
    // BN_MBROW=0 keeps the tiled kernels.
    static void row_streaming(MbDesc &m, double &halo) {
        const char *env = getenv("BN_MBROW");
        if (env && std::string(env) == "0") return;
        if (!mbconv_row_supported(m)) return;
        const int outw = mbconv_row_outw(m.k, m.s);
        // a strip expands 32 halo columns for `outw` outputs whatever the map's width: on narrow maps (Perch: 32 or 16
        // columns against strips of 30 / 28 / 14) half of every strip is idle and the tiled kernel wins (measured: 0.53-0.57
        // utilisation -> 1.2-2.5x slower, 0.71 and above -> faster).  Round 3: such maps are usually TALL (Perch: 125 x 32,
        // 63 x 16) -- the kernel then streams along the map's rows' direction instead (MbDesc::row_tr: strips across the
        // height), if that fills its strips.  BN_MBROW=force takes the row kernel regardless (tests), BN_MBROW_TR=0 never
        // transposes, =1 always does where the kernel supports it
        const bool force = env && std::string(env) == "force";
        const char *tre = getenv("BN_MBROW_TR");
        const int strips = (m.OW + outw - 1) / outw, strips_t = (m.OH + outw - 1) / outw;
        const double util = (double)m.OW / (double)(strips * outw), util_t = (double)m.OH / (double)(strips_t * outw);
        bool tr = m.k1 == 0 && m.k == 3 && util_t >= 0.7 && util_t > util + 0.08;  // (5 x 5 instances: no register to spare for the second addressing form)
        if (tre && std::string(tre) == "0") tr = false;
        if (tre && std::string(tre) == "1") tr = m.k1 == 0 && m.k == 3;
        if (tr) {
            MbDesc probe = m;
            probe.row_tr = 1;
            if (!mbconv_row_supported(probe)) tr = false;
        }
        if (!force && !tr && util < 0.7) return;
        m.row_mode = 1;
        m.row_tr = tr ? 1 : 0;
        m.row_b3 = (env_int("BN_MBROW_B3", 1) != 0 && env_int("BN_GEMM3", 2) != 0 && m.k1 == 0 && m.Cin % 8 == 0) ? 1 : 0;
        const int rows_k = tr ? m.OW : m.OH, cols_k = tr ? m.OH : m.OW;  // the kernel's output rows / columns
        // band height: a band of toh output rows expands (toh - 1) s + k halo rows, so taller bands recompute less (12 rows of a
        // 5x5 block: 16 halo rows instead of 2 x 10) -- what several contexts sharing the chip pay for; a block with one or two
        // 32-channel chunks keeps 8 so that one context alone still has enough waves (stem: 38 us at 8, 48 us at 12)
        // (bands balanced: 16 rows are one band of 16, not 12 + 4)
        const int toh_target = (m.C + 31) / 32 >= 3 ? 12 : 8;
        const int nbands = std::max(1, (rows_k + toh_target / 2 - 1) / toh_target);
        const int toh_default = (rows_k + nbands - 1) / nbands;
        m.toh = std::min<int32_t>(rows_k, getenv("BN_MBROW_TOH") ? std::max(1, atoi(getenv("BN_MBROW_TOH"))) : toh_default);
        m.tiles_x = (cols_k + outw - 1) / outw;
        m.tiles_y = (rows_k + m.toh - 1) / m.toh;
        halo = 32.0 * m.tiles_x * (double)m.tiles_y * ((m.toh - 1) * m.s + m.k);
    }

    // ------------------------------------------------------------- lowering
    void lower(const OnnxNode &n) {
        const std::string &t = n.op_type;
        // constant folding first
        std::vector<const Val *> ins;
        for (size_t k = 0; k < n.inputs.size(); k++) ins.push_back(n.inputs[k].empty() ? nullptr : &get(n, k));
        if (t == "Constant") { lower_constant(n); return; }
        if (t == "Shape") { lower_shape(n); return; }
        if (t == "Size") {
            const Val &v = get(n, 0);
            if (!v.is_const) unsupported(n, "Size of an activation depends on the batch size, which a plan does not fix");
            define(n.outputs[0], make_const_i({}, {v.numel()}));
            return;
        }
        bool allc = !ins.empty() && all_const(ins);
        if (allc && try_fold(n, ins)) return;

        if (t == "Identity" || t == "Dropout") { define(n.outputs[0], get(n, 0)); return; }
        if (t == "Cast" && !get(n, 0).is_const) {
            // activations are f32 whatever the graph calls them: a cast between float types is the value itself, a cast to bool is
            // x != 0 -> 1, a cast to an integer type truncates toward zero (ONNX Cast) -- both stay f32 numbers
            const int64_t to = n.attr_i("to", 1);
            if (to == 1 || to == 10 || to == 11 || to == 16) { define(n.outputs[0], get(n, 0)); return; }
            if (to == 9) return lower_unary(n, ActSpec{ACT_NEZ, 0, 0});
            if (to == 2 || to == 3 || to == 4 || to == 5 || to == 6 || to == 7 || to == 12 || to == 13) return lower_unary(n, ActSpec{ACT_TRUNC, 0, 0});
            unsupported(n, "Cast to element type " + std::to_string(to) + " (strings / complex) is outside the native subset");
        }
        if (t == "Not") return lower_unary(n, ActSpec{ACT_AFFINE, -1.0f, 1.0f});  // on a 0 / 1 tensor
        if (t == "And" || t == "Or" || compare_code(t) != BIN_NONE) return lower_binary(n);
        if (t == "Where") return lower_where(n);
        if (t == "Gather" && !get(n, 0).is_const) return lower_gather(n);
        if (t == "DFT") return lower_dft(n);
        if (t == "Transpose") return lower_transpose(n);
        if (t == "Reshape" || t == "Flatten" || t == "Squeeze" || t == "Unsqueeze") return lower_reshape_like(n);
        if (t == "Slice") return lower_slice(n);
        if (t == "Concat") return lower_concat(n);
        if (t == "Pad") return lower_pad(n);
        if (t == "Softmax" || t == "LogSoftmax") return lower_softmax(n);
        if (t == "Split") return lower_split(n);
        if (t == "MaxPool" || t == "AveragePool") return lower_pool(n);
        if (t == "Conv") return lower_conv(n);
        if (t == "MatMul" || t == "Gemm") return lower_matmul(n);
        if (t == "BatchNormalization") return lower_batchnorm(n);
        if (t == "Add" || t == "Sub" || t == "Mul" || t == "Div" || t == "Pow" || t == "Max" || t == "Min") return lower_binary(n);
        if (lower_composite(n)) return;
        if (t == "GlobalAveragePool" || t == "GlobalMaxPool" || t.rfind("Reduce", 0) == 0) return lower_reduce(n);
        if (t == "STFT") return lower_stft(n);
        if (t == "Expand") return lower_expand(n);
        if (t == "Tile") return lower_tile(n);
        if (t == "PRelu") return lower_prelu(n);
        if (t == "InstanceNormalization") return lower_instance_norm(n);
        ActSpec a;
        if (unary_spec(n, a)) return lower_unary(n, a);
        // exporter dialects this path has met but does not map: say what the node is and what to export instead
        if (t == "Resize" || t == "Upsample") unsupported(n, "resampling of feature maps is outside the native subset (no BirdNET / Perch graph needs it)");
        if (t == "Gather" || t == "GatherND" || t == "GatherElements") unsupported(n, "gathers of activations are mapped only where the constant indices are an affine pattern (a strided view: tf.signal.frame's selector)");
        unsupported(n, "operator is outside the native subset");
    }

    // ------------------------------------------------------------- STFT (opset 17)
    // output[b, f, k, 0/1] = Re / Im of sum_n signal[b, f*step + n] * window[n] * exp(-2 pi i k n / N), no padding,
    // k = 0 .. N/2 (onesided, the default) or 0 .. N-1.  That is a 1-D framing convolution with 2*bins filters of N
    // taps and stride `step`: the filters are built here (double-precision cos / sin, rounded once) in the order
    // [all real rows | all imaginary rows], so that the mirror-symmetry fold and the FFT recognition of the framing
    // conv see one cos block and one sin block; the node's [frames, bins, 2] result is a strided VIEW of the conv's
    // channels-last output (no copy).
    void lower_stft(const OnnxNode &n) {
        const Val x = get(n, 0);
        if (x.is_const) unsupported(n, "STFT of a constant signal");
        const Val *stepv = opt(n, 1), *win = opt(n, 2), *flen = opt(n, 3);
        if (!stepv || !stepv->is_const) unsupported(n, "frame_step must be a constant");
        const int64_t step = const_ints(n, *stepv).at(0);
        if (win && !win->is_const) unsupported(n, "window must be a constant");
        if (flen && !flen->is_const) unsupported(n, "frame_length must be a constant");
        int64_t N = flen ? const_ints(n, *flen).at(0) : (win ? win->numel() : 0);
        if (N <= 0 || step <= 0) unsupported(n, "needs a window or a frame_length, and a positive frame_step");
        if (win && (win->is_int || win->numel() != N)) unsupported(n, "window length " + std::to_string(win ? win->numel() : 0) + " differs from frame_length " + std::to_string(N));
        // signal: [L], [L, 1] (real); [L, 2] would be complex
        int64_t L = 0, sL = 0;
        if (x.dims.size() == 1) { L = x.dims[0]; sL = x.strides[0]; }
        else if (x.dims.size() == 2 && x.dims[1] == 1) { L = x.dims[0]; sL = x.strides[0]; }
        else unsupported(n, "signal must be real: [batch, length] or [batch, length, 1], got per-sample dims " + dims_str(x.dims));
        if (L < N) unsupported(n, "signal shorter than one frame");
        const bool onesided = n.attr_i("onesided", 1) != 0;
        const int64_t bins = onesided ? N / 2 + 1 : N;
        if (N * 2 * bins > ((int64_t)1 << 28)) unsupported(n, "DFT basis too large");
        Val w;
        w.is_const = true;
        w.dims = {2 * bins, 1, N};
        w.f.resize((size_t)(2 * bins * N));
        for (int64_t k = 0; k < bins; k++)
            for (int64_t t = 0; t < N; t++) {
                const double wv = win ? (double)win->f[(size_t)t] : 1.0;
                const double th = 2.0 * M_PI * (double)((k * t) % N) / (double)N;
                w.f[(size_t)(k * N + t)] = (float)(wv * std::cos(th));
                w.f[(size_t)((bins + k) * N + t)] = (float)(-wv * std::sin(th));
            }
        const std::string base = "stft:" + (n.name.empty() ? n.outputs[0] : n.name);
        Val xv = x;
        xv.dims = {1, L};
        xv.strides = {L * sL, sL};
        vals_[base + "/signal"] = xv;
        vals_[base + "/basis"] = std::move(w);
        OnnxNode conv;
        conv.name = base;
        conv.op_type = "Conv";
        conv.inputs = {base + "/signal", base + "/basis"};
        conv.outputs = {base + "/frames"};
        OnnxAttr st;
        st.name = "strides";
        st.type = 7;
        st.ints = {step};
        conv.attrs["strides"] = st;
        lower_conv(conv);
        auto yit = vals_.find(base + "/frames");
        if (yit == vals_.end()) unsupported(n, "internal: the framing conv defined no output");
        const Val y = yit->second;
        if (y.is_const || y.dims.size() != 2 || y.dims[0] != 2 * bins) unsupported(n, "internal: framing conv produced " + dims_str(y.dims));
        Val out = y;
        out.dims = {y.dims[1], bins, 2};
        out.strides = {y.strides[1], y.strides[0], bins * y.strides[0]};
        define(n.outputs[0], out);
    }

    // ---- operators lowered as short sequences of operators the planner already has (each step is an ordinary node to
    // the lowering: constants fold, elementwise steps fuse into chains)
    OnnxNode synth_node(const std::string &op, const std::string &name, std::vector<std::string> ins, const std::string &out) {
        OnnxNode s_;
        s_.name = name;
        s_.op_type = op;
        s_.inputs = std::move(ins);
        s_.outputs = {out};
        return s_;
    }
    std::string synth_const(const std::string &name, Dims dims, std::vector<float> f) {
        Val c;
        c.is_const = true;
        c.dims = std::move(dims);
        c.f = std::move(f);
        vals_[name] = std::move(c);
        return name;
    }
    static void set_ints(OnnxNode &nd, const std::string &key, std::vector<int64_t> v) {
        OnnxAttr a;
        a.name = key;
        a.type = 7;
        a.ints = std::move(v);
        nd.attrs[key] = a;
    }
    static void set_int(OnnxNode &nd, const std::string &key, int64_t v) {
        OnnxAttr a;
        a.name = key;
        a.type = 2;
        a.i = v;
        nd.attrs[key] = a;
    }

    // Operators written out as the operators above (round 5): the pieces go through the elementwise / reduction lowering, so they fuse into
    // their neighbours like hand-written graphs of the same arithmetic.  ONNX defines each of these by exactly this formula.
    bool lower_composite(const OnnxNode &n) {
        const std::string &t = n.op_type;
        if (n.outputs.empty() || n.inputs.empty()) return false;
        const std::string base = "cmp:" + n.outputs[0];
        const std::string &x = n.inputs[0];
        int seq = 0;
        auto step = [&](const char *op, std::vector<std::string> ins, bool last = false) {
            const std::string out = last ? n.outputs[0] : base + "/" + std::to_string(seq);
            lower(synth_node(op, last ? n.name : n.name + "/" + std::to_string(seq), std::move(ins), out));
            seq++;
            return out;
        };
        auto k = [&](float v) { return synth_const(base + "/c" + std::to_string(seq++), {1}, {v}); };
        auto reduce = [&](const char *op, const std::string &in, bool last) {
            const std::string out = last ? n.outputs[0] : base + "/" + std::to_string(seq);
            OnnxNode r = synth_node(op, last ? n.name : n.name + "/" + std::to_string(seq), {in}, out);
            seq++;
            if (n.has("axes")) set_ints(r, "axes", n.attr_ints("axes"));
            else if (has_input(n, 1)) r.inputs.push_back(n.inputs[1]);
            set_int(r, "keepdims", n.attr_i("keepdims", 1));
            if (n.has("noop_with_empty_axes")) set_int(r, "noop_with_empty_axes", n.attr_i("noop_with_empty_axes", 0));
            lower(r);
            return out;
        };
        if (t == "Elu" || t == "Selu" || t == "Celu") {
            // max(x, 0) + alpha (exp(min(x, 0) / a) - 1), a = alpha for Celu and 1 otherwise; Selu scales the sum by gamma
            const float alpha = n.attr_f("alpha", t == "Selu" ? 1.67326319217681884765625f : 1.0f);
            const float gamma = t == "Selu" ? n.attr_f("gamma", 1.05070102214813232421875f) : 1.0f;
            std::string neg = step("Min", {x, k(0.0f)});
            if (t == "Celu") neg = step("Div", {neg, k(alpha)});
            neg = step("Exp", {neg});
            neg = step("Mul", {neg, k(alpha * gamma)});
            neg = step("Sub", {neg, k(alpha * gamma)});
            std::string pos = step("Relu", {x});
            if (gamma != 1.0f) pos = step("Mul", {pos, k(gamma)});
            step("Add", {pos, neg}, true);
            return true;
        }
        if (t == "ThresholdedRelu") {
            const std::string m = step("Greater", {x, k(n.attr_f("alpha", 1.0f))});
            step("Where", {m, x, k(0.0f)}, true);
            return true;
        }
        if (t == "Softsign") {
            std::string d = step("Abs", {x});
            d = step("Add", {d, k(1.0f)});
            step("Div", {x, d}, true);
            return true;
        }
        if (t == "Mish") {
            std::string d = step("Softplus", {x});
            d = step("Tanh", {d});
            step("Mul", {x, d}, true);
            return true;
        }
        if (t == "Gelu") {
            std::string g;
            if (n.attr_s("approximate", "none") == "tanh") {
                g = step("Mul", {x, x});
                g = step("Mul", {g, x});
                g = step("Mul", {g, k(0.044715f)});
                g = step("Add", {g, x});
                g = step("Mul", {g, k(0.797884583473205566406250f)});
                g = step("Tanh", {g});
            } else {
                g = step("Mul", {x, k(0.707106769084930419921875f)});
                g = step("Erf", {g});
            }
            g = step("Mul", {g, k(0.5f)});
            g = step("Add", {g, k(0.5f)});
            step("Mul", {x, g}, true);
            return true;
        }
        if (t == "Sign") {
            const std::string p = step("Greater", {x, k(0.0f)}), m = step("Less", {x, k(0.0f)});
            step("Sub", {p, m}, true);
            return true;
        }
        if (t == "Sum" || t == "Mean") {
            std::string acc = x;
            const size_t cnt = n.inputs.size();
            if (cnt == 1) { lower(synth_node("Identity", n.name, {x}, n.outputs[0])); return true; }
            for (size_t j = 1; j < cnt; j++) acc = step("Add", {acc, n.inputs[j]}, t == "Sum" && j + 1 == cnt);
            if (t == "Mean") step("Div", {acc, k((float)cnt)}, true);
            return true;
        }
        if (t == "ReduceL1") {
            reduce("ReduceSum", step("Abs", {x}), true);
            return true;
        }
        if (t == "ReduceLogSum") {
            step("Log", {reduce("ReduceSum", x, false)}, true);
            return true;
        }
        if (t == "ReduceLogSumExp") {
            // log(sum(exp(x - m))) + m with m the maximum over the same axes: the form that cannot overflow (what ORT computes)
            OnnxNode mx = synth_node("ReduceMax", n.name + "/max", {x}, base + "/max");
            if (n.has("axes")) set_ints(mx, "axes", n.attr_ints("axes"));
            else if (has_input(n, 1)) mx.inputs.push_back(n.inputs[1]);
            set_int(mx, "keepdims", 1);
            lower(mx);
            std::string e = step("Sub", {x, base + "/max"});
            e = step("Exp", {e});
            e = reduce("ReduceSum", e, false);
            e = step("Log", {e});
            std::string m = base + "/max";
            if (n.attr_i("keepdims", 1) == 0) {
                OnnxNode m2 = synth_node("ReduceMax", n.name + "/max0", {x}, base + "/max0");
                if (n.has("axes")) set_ints(m2, "axes", n.attr_ints("axes"));
                else if (has_input(n, 1)) m2.inputs.push_back(n.inputs[1]);
                set_int(m2, "keepdims", 0);
                lower(m2);
                m = base + "/max0";
            }
            step("Add", {e, m}, true);
            return true;
        }
        if (t == "LayerNormalization") {
            // over the dimensions from `axis` to the last: (x - mean) / sqrt(var + eps) * Scale + B; the optional Mean / InvStdDev outputs
            // are training-side and not produced here
            if (n.outputs.size() > 1 && (!n.outputs[1].empty() || (n.outputs.size() > 2 && !n.outputs[2].empty())))
                unsupported(n, "the Mean / InvStdDev outputs of LayerNormalization are not produced");
            const Val &xv = get(n, 0);
            if (xv.is_const) return false;
            const int64_t r = (int64_t)xv.dims.size() + 1;
            int64_t axis = n.attr_i("axis", -1);
            if (axis < 0) axis += r;
            if (axis <= 0 || axis >= r) unsupported(n, "LayerNormalization over the batch dimension");
            std::vector<int64_t> axes;
            for (int64_t a = axis; a < r; a++) axes.push_back(a);
            auto mean_of = [&](const std::string &in) {
                const std::string out = base + "/" + std::to_string(seq);
                OnnxNode m = synth_node("ReduceMean", n.name + "/" + std::to_string(seq), {in}, out);
                seq++;
                set_ints(m, "axes", axes);
                set_int(m, "keepdims", 1);
                lower(m);
                return out;
            };
            const std::string d = step("Sub", {x, mean_of(x)});
            std::string v = mean_of(step("Mul", {d, d}));
            v = step("Add", {v, k(n.attr_f("epsilon", 1e-5f))});
            v = step("Sqrt", {v});
            v = step("Div", {d, v});
            const bool has_b = has_input(n, 2);
            v = step("Mul", {v, n.inputs[1]}, !has_b);
            if (has_b) step("Add", {v, n.inputs[2]}, true);
            return true;
        }
        return false;
    }

    // PRelu(x, slope) = max(x, 0) + slope * min(x, 0); one slope for everything is LeakyRelu
    void lower_prelu(const OnnxNode &n) {
        const Val *sl = opt(n, 1);
        if (!sl || !sl->is_const || sl->is_int) unsupported(n, "PRelu needs a constant slope");
        const std::string base = "prelu:" + n.outputs[0];
        if (sl->numel() == 1) {
            OnnxNode lr = synth_node("LeakyRelu", n.name, {n.inputs[0]}, n.outputs[0]);
            OnnxAttr a;
            a.name = "alpha";
            a.type = 1;
            a.f = sl->f[0];
            lr.attrs["alpha"] = a;
            lower(lr);
            return;
        }
        lower(synth_node("Relu", n.name + "/pos", {n.inputs[0]}, base + "/pos"));
        lower(synth_node("Min", n.name + "/neg", {n.inputs[0], synth_const(base + "/zero", {1}, {0.0f})}, base + "/neg"));
        lower(synth_node("Mul", n.name + "/scaled", {base + "/neg", n.inputs[1]}, base + "/scaled"));
        lower(synth_node("Add", n.name, {base + "/pos", base + "/scaled"}, n.outputs[0]));
    }

    // Tile: only repeats of dimensions of size 1 (= Expand); anything else would need a gather
    void lower_tile(const OnnxNode &n) {
        const Val &x = get(n, 0);
        const Val *rp = opt(n, 1);
        if (!rp || !rp->is_const) unsupported(n, "Tile needs constant repeats");
        std::vector<int64_t> rep = const_ints(n, *rp);
        Dims full = x.dims;
        if (!x.is_const) full.insert(full.begin(), 1);  // the batch dimension: repeats[0] must be 1
        if (rep.size() != full.size()) unsupported(n, "repeats must have one entry per dimension");
        std::vector<int64_t> target(full.size());
        for (size_t k = 0; k < full.size(); k++) {
            if (rep[k] != 1 && full[k] != 1) unsupported(n, "Tile repeats a dimension of size " + std::to_string(full[k]) + " (only size-1 dimensions, i.e. broadcasts, are mapped)");
            if (k == 0 && !x.is_const && rep[k] != 1) unsupported(n, "Tile along the batch dimension");
            target[k] = full[k] * rep[k];
        }
        const std::string sname = "tile:" + n.outputs[0] + "/shape";
        Val sh = make_const_i({(int64_t)target.size()}, target);
        vals_[sname] = sh;
        lower_expand(synth_node("Expand", n.name, {n.inputs[0], sname}, n.outputs[0]));
    }

    // InstanceNormalization: per (sample, channel) over the spatial dimensions, (x - mean) / sqrt(var + eps) * scale + B
    void lower_instance_norm(const OnnxNode &n) {
        const Val &x = get(n, 0);
        const Val *sc = opt(n, 1), *bi = opt(n, 2);
        if (x.is_const || !sc || !bi || !sc->is_const || !bi->is_const) unsupported(n, "InstanceNormalization needs an activation input and constant scale / B");
        const size_t r = x.dims.size();  // per-sample rank: [C, spatial...]
        if (r < 2 || sc->numel() != x.dims[0] || bi->numel() != x.dims[0]) unsupported(n, "scale / B must have one entry per channel");
        const std::string base = "inorm:" + n.outputs[0];
        std::vector<int64_t> axes;
        for (size_t k = 2; k <= r; k++) axes.push_back((int64_t)k);
        Dims cdims(r, 1);
        cdims[0] = x.dims[0];
        auto mean_of = [&](const std::string &in, const std::string &out, const std::string &nm) {
            OnnxNode m = synth_node("ReduceMean", nm, {in}, out);
            set_ints(m, "axes", axes);
            set_int(m, "keepdims", 1);
            lower(m);
        };
        mean_of(n.inputs[0], base + "/mean", n.name + "/mean");
        lower(synth_node("Sub", n.name + "/centred", {n.inputs[0], base + "/mean"}, base + "/d"));
        lower(synth_node("Mul", n.name + "/sq", {base + "/d", base + "/d"}, base + "/sq"));
        mean_of(base + "/sq", base + "/var", n.name + "/var");
        lower(synth_node("Add", n.name + "/eps", {base + "/var", synth_const(base + "/epsc", {1}, {n.attr_f("epsilon", 1e-5f)})}, base + "/ve"));
        lower(synth_node("Sqrt", n.name + "/std", {base + "/ve"}, base + "/std"));
        lower(synth_node("Div", n.name + "/norm", {base + "/d", base + "/std"}, base + "/norm"));
        lower(synth_node("Mul", n.name + "/scale", {base + "/norm", synth_const(base + "/scalec", cdims, sc->f)}, base + "/scaled"));
        lower(synth_node("Add", n.name, {base + "/scaled", synth_const(base + "/biasc", cdims, bi->f)}, n.outputs[0]));
    }

    // Expand(x, shape): x * ones(broadcast shape) -- exact for every float including -0, infinities and NaN payloads
    // that survive a multiplication by 1; goes through the elementwise lowering, so it fuses into its neighbours.
    void lower_expand(const OnnxNode &n) {
        const Val &x = get(n, 0);
        const Val *shp = opt(n, 1);
        if (!shp || !shp->is_const) unsupported(n, "Expand needs a constant shape");
        std::vector<int64_t> target = const_ints(n, *shp);
        Dims full = x.dims;
        if (!x.is_const) full.insert(full.begin(), BATCH_SENTINEL);
        while (full.size() < target.size()) full.insert(full.begin(), 1);
        while (target.size() < full.size()) target.insert(target.begin(), 1);
        Dims ones_dims;
        for (size_t k = 0; k < full.size(); k++) {
            int64_t tdim = target[k], xdim = full[k];
            if (xdim == BATCH_SENTINEL || tdim == BATCH_SENTINEL) {
                if (k != 0 || (xdim != BATCH_SENTINEL && xdim != 1) || (tdim != BATCH_SENTINEL && tdim != 1))
                    unsupported(n, "Expand may only keep the batch dimension as it is");
                continue;  // the batch dimension is not part of the ones operand
            }
            if (tdim != 1 && xdim != 1 && tdim != xdim) unsupported(n, "shape " + dims_str(target) + " does not broadcast with " + dims_str(full));
            ones_dims.push_back(std::max(tdim, xdim));
        }
        if (!x.is_const && full.size() != x.dims.size() + 1) unsupported(n, "Expand would add dimensions in front of the batch");
        if (prod(ones_dims) > ((int64_t)1 << 28)) unsupported(n, "Expand target too large");
        Val ones;
        ones.is_const = true;
        ones.dims = ones_dims;
        ones.f.assign((size_t)prod(ones_dims), 1.0f);
        if (x.is_const && x.is_int) unsupported(n, "Expand of an integer constant");
        const std::string oname = "expand:" + n.outputs[0] + "/ones";
        vals_[oname] = std::move(ones);
        OnnxNode mul;
        mul.name = n.name.empty() ? "expand:" + n.outputs[0] : n.name;
        mul.op_type = "Mul";
        mul.inputs = {n.inputs[0], oname};
        mul.outputs = {n.outputs[0]};
        lower(mul);
    }

    void lower_constant(const OnnxNode &n) {
        auto it = n.attrs.find("value");
        if (it != n.attrs.end()) {
            const OnnxTensor &t = it->second.t;
            define(n.outputs[0], t.is_float() ? make_const_f(t.dims, t.f) : make_const_i(t.dims, t.i));
        } else if (n.has("value_float")) define(n.outputs[0], make_const_f({}, {n.attr_f("value_float", 0)}));
        else if (n.has("value_int")) define(n.outputs[0], make_const_i({}, {n.attr_i("value_int", 0)}));
        else if (n.has("value_ints")) { auto v = n.attr_ints("value_ints"); define(n.outputs[0], make_const_i({(int64_t)v.size()}, v)); }
        else if (n.has("value_floats")) { auto v = n.attrs.at("value_floats").floats; define(n.outputs[0], make_const_f({(int64_t)v.size()}, v)); }
        else unsupported(n, "Constant without a supported value attribute");
    }
    void lower_shape(const OnnxNode &n) {
        const Val &v = get(n, 0);
        std::vector<int64_t> s;
        if (v.is_const) s = v.dims;
        else { s.push_back(BATCH_SENTINEL); s.insert(s.end(), v.dims.begin(), v.dims.end()); }
        define(n.outputs[0], make_const_i({(int64_t)s.size()}, s));
    }

    bool try_fold(const OnnxNode &n, const std::vector<const Val *> &ins) {
        const std::string &t = n.op_type;
        const Val &a = *ins[0];
        if (ins.size() >= 2 && ins[1] && (t == "Add" || t == "Sub" || t == "Mul" || t == "Div" || t == "Pow" || t == "Max" || t == "Min" ||
                                          t == "And" || t == "Or" || compare_code(t) != BIN_NONE))
            return fold_binary(n, a, *ins[1]);
        if (t == "Not") {
            std::vector<int64_t> o(a.is_int ? a.i.size() : a.f.size());
            for (size_t k = 0; k < o.size(); k++) o[k] = a.is_int ? a.i[k] == 0 : a.f[k] == 0.0f;
            define(n.outputs[0], make_const_i(a.dims, o));
            return true;
        }
        if (t == "Where" && ins.size() == 3 && ins[1] && ins[2]) {
            // Where(cond, x, y) on constants (the exporters' shape arithmetic: Where(Equal(shape, -1), ...)): SelA(x, cond) + SelB(y, cond)
            const std::string base = "where:" + n.outputs[0];
            lower(synth_node("bn.SelA", n.name + "/x", {n.inputs[1], n.inputs[0]}, base + "/x"));
            lower(synth_node("bn.SelB", n.name + "/y", {n.inputs[2], n.inputs[0]}, base + "/y"));
            lower(synth_node("Add", n.name, {base + "/x", base + "/y"}, n.outputs[0]));
            return true;
        }
        if (t == "Identity") { define(n.outputs[0], a); return true; }
        if (t == "Cast") {
            int64_t to = n.attr_i("to", 1);
            Val o = a;
            if (to == 1 || to == 11 || to == 10) { if (a.is_int) { o.is_int = false; o.f.assign(a.i.begin(), a.i.end()); o.i.clear(); } }
            else { if (!a.is_int) { o.is_int = true; o.i.resize(a.f.size()); for (size_t k = 0; k < a.f.size(); k++) o.i[k] = (int64_t)a.f[k]; o.f.clear(); } }
            define(n.outputs[0], o);
            return true;
        }
        if (t == "Unsqueeze" || t == "Squeeze" || t == "Reshape" || t == "Flatten") {
            Dims nd = reshape_target(n, a.dims, /*is_const=*/true);
            Val o = a; o.dims = nd;
            define(n.outputs[0], o);
            return true;
        }
        if (t == "Transpose") {
            auto perm = n.attr_ints("perm");
            size_t r = a.dims.size();
            if (perm.empty()) for (size_t k = 0; k < r; k++) perm.push_back((int64_t)(r - 1 - k));
            Dims rs = row_major(a.dims), nd(r), ns(r);
            for (size_t k = 0; k < r; k++) { nd[k] = a.dims[perm[k]]; ns[k] = rs[perm[k]]; }
            define(n.outputs[0], const_view(a, nd, ns, 0));
            return true;
        }
        if (t == "Concat") {
            int64_t axis = n.attr_i("axis", 0);
            size_t r = a.dims.size();
            if (axis < 0) axis += (int64_t)r;
            Dims od = a.dims;
            od[axis] = 0;
            for (auto v : ins) od[axis] += v->dims[axis];
            int64_t outer = 1, inner = 1;
            for (int64_t k = 0; k < axis; k++) outer *= od[k];
            for (size_t k = axis + 1; k < r; k++) inner *= od[k];
            Val o; o.is_const = true; o.is_int = a.is_int; o.dims = od;
            for (int64_t ou = 0; ou < outer; ou++)
                for (auto v : ins) {
                    int64_t chunk = v->dims[axis] * inner;
                    for (int64_t e = 0; e < chunk; e++) {
                        if (o.is_int) o.i.push_back(v->is_int ? v->i[ou * chunk + e] : (int64_t)v->f[ou * chunk + e]);
                        else o.f.push_back(v->is_int ? (float)v->i[ou * chunk + e] : v->f[ou * chunk + e]);
                    }
                }
            define(n.outputs[0], o);
            return true;
        }
        if (t == "Gather") {
            int64_t axis = n.attr_i("axis", 0);
            const Val &idx = *ins[1];
            if (axis < 0) axis += (int64_t)a.dims.size();
            if (axis != 0 || a.dims.size() != 1) return false;
            auto ids = const_ints(n, idx);
            Val o; o.is_const = true; o.is_int = a.is_int; o.dims = idx.dims;
            for (auto id : ids) {
                if (id < 0) id += a.dims[0];
                if (a.is_int) o.i.push_back(a.i[id]); else o.f.push_back(a.f[id]);
            }
            define(n.outputs[0], o);
            return true;
        }
        if (t == "Slice") {
            Dims nd, ns; int64_t off;
            slice_params(n, a.dims, row_major(a.dims), nd, ns, off, /*batched=*/false);
            define(n.outputs[0], const_view(a, nd, ns, off));
            return true;
        }
        if (t == "ConstantOfShape") {
            auto shape = const_ints(n, a);
            float fv = 0; bool isint = false; int64_t iv = 0;
            auto it = n.attrs.find("value");
            if (it != n.attrs.end()) { if (it->second.t.is_float()) fv = it->second.t.f[0]; else { isint = true; iv = it->second.t.i[0]; } }
            for (auto d : shape) if (d < 0) return false;
            if (isint) define(n.outputs[0], make_const_i(shape, std::vector<int64_t>(prod(shape), iv)));
            else define(n.outputs[0], make_const_f(shape, std::vector<float>(prod(shape), fv)));
            return true;
        }
        if (t == "Range") {
            if (!a.is_int) return false;
            int64_t s = a.i[0], e = ins[1]->i[0], d = ins[2]->i[0];
            std::vector<int64_t> r;
            for (int64_t x = s; d > 0 ? x < e : x > e; x += d) r.push_back(x);
            define(n.outputs[0], make_const_i({(int64_t)r.size()}, r));
            return true;
        }
        return fold_unary(n, a);
    }

    // Target dims for Reshape/Flatten/Squeeze/Unsqueeze.  For activations `dims`
    // is per-sample and ONNX axes/shapes include the batch at position 0.
    Dims reshape_target(const OnnxNode &n, const Dims &dims, bool is_const) {
        const std::string &t = n.op_type;
        Dims full = dims;
        if (!is_const) full.insert(full.begin(), BATCH_SENTINEL);
        Dims out;
        auto axes_of = [&](size_t input_idx) {
            std::vector<int64_t> ax = n.attr_ints("axes");
            if (ax.empty() && has_input(n, input_idx)) ax = const_ints(n, get(n, input_idx));
            return ax;
        };
        if (t == "Flatten") {
            int64_t axis = n.attr_i("axis", 1);
            int64_t r = (int64_t)full.size();
            if (axis < 0) axis += r;
            if (!is_const && axis != 1) unsupported(n, "Flatten must keep the batch as the leading dimension (axis=1)");
            int64_t a = 1, b = 1;
            for (int64_t k = 0; k < axis; k++) a *= full[k];
            for (int64_t k = axis; k < r; k++) b *= full[k];
            out = is_const ? Dims{a, b} : Dims{BATCH_SENTINEL, b};
        } else if (t == "Squeeze") {
            auto ax = axes_of(1);
            int64_t r = (int64_t)full.size();
            std::set<int64_t> drop;
            if (ax.empty()) { for (int64_t k = 0; k < r; k++) if (full[k] == 1) drop.insert(k); }
            else for (auto a : ax) drop.insert(a < 0 ? a + r : a);
            for (int64_t k = 0; k < r; k++) {
                if (drop.count(k)) { if (full[k] != 1) unsupported(n, "cannot squeeze a dimension of size " + std::to_string(full[k])); }
                else out.push_back(full[k]);
            }
        } else if (t == "Unsqueeze") {
            auto ax = axes_of(1);
            int64_t r = (int64_t)full.size() + (int64_t)ax.size();
            std::set<int64_t> ins;
            for (auto a : ax) ins.insert(a < 0 ? a + r : a);
            size_t src = 0;
            for (int64_t k = 0; k < r; k++) {
                if (ins.count(k)) out.push_back(1);
                else out.push_back(full[src++]);
            }
        } else {  // Reshape
            auto shape = const_ints(n, get(n, 1));
            bool allowzero = n.attr_i("allowzero", 0) != 0;
            int64_t known = 1, total = 1;
            int infer = -1;
            bool batch_in_total = false;
            for (auto d : full) { if (d == BATCH_SENTINEL) batch_in_total = true; else total *= d; }
            bool batch_in_out = false;
            for (size_t k = 0; k < shape.size(); k++) {
                int64_t d = shape[k];
                if (d == 0 && !allowzero) d = k < full.size() ? full[k] : 0;
                if (d == -1) { infer = (int)k; out.push_back(-1); continue; }
                if (d == BATCH_SENTINEL) batch_in_out = true;
                else known *= d;
                out.push_back(d);
            }
            if (!is_const && batch_in_total && !batch_in_out) {
                // The exporter wrote the batch as a literal (1 for a batch-1 trace) or as -1.
                if (infer == 0) { out[0] = BATCH_SENTINEL; infer = -1; }
                else if (!out.empty() && out[0] == 1 && (infer >= 0 || known == total)) out[0] = BATCH_SENTINEL;
                else unsupported(n, "Reshape target " + dims_str(shape) + " does not keep the batch dimension leading");
            }
            if (infer >= 0) {
                if (known == 0 || total % known) unsupported(n, "Reshape cannot infer -1");
                out[infer] = total / known;
            }
        }
        if (!is_const) {
            if (out.empty() || out[0] != BATCH_SENTINEL) unsupported(n, "result does not keep the batch as the leading dimension");
            out.erase(out.begin());
            for (auto d : out) if (d == BATCH_SENTINEL || d < 0) unsupported(n, "batch dimension used in a non-leading position");
            if (prod(out) != prod(dims)) unsupported(n, "element count changes from " + dims_str(dims) + " to " + dims_str(out));
        }
        return out;
    }

    void lower_reshape_like(const OnnxNode &n) {
        const Val &v = get(n, 0);
        Dims nd = reshape_target(n, v.dims, false);
        // view if the non-unit dims are unchanged, else require row-major contiguity
        Dims a, as, b;
        for (size_t k = 0; k < v.dims.size(); k++) if (v.dims[k] != 1) { a.push_back(v.dims[k]); as.push_back(v.strides[k]); }
        for (auto d : nd) if (d != 1) b.push_back(d);
        Val o = v;
        o.dims = nd;
        Dims vs;
        if (a == b) {
            o.strides.assign(nd.size(), 0);
            size_t j = 0;
            for (size_t k = 0; k < nd.size(); k++) if (nd[k] != 1) o.strides[k] = as[j++];
        } else if (!v.win_name.empty()) {
            vals_[n.inputs[0]] = apply_window(v, n.inputs[0]);  // (the non-unit dims change: the window's axis would be lost)
            return lower_reshape_like(n);
        } else if (view_reshape(a, as, nd, vs)) {
            o.strides = vs;  // (round 5) dims that merge / split without moving an element: tf.signal.frame's [frames, L / sub, sub] -> [frames, L]
        } else {
            Val src = v.contiguous() ? v : materialize(v, row_major(v.dims), n.name);
            o = src;
            o.dims = nd;
            o.strides = row_major(nd);
        }
        define(n.outputs[0], o);
    }

    // Strides of `nd` over the same elements as the (non-unit) dims `a` with strides `as`, if one exists without a copy: the old dims are
    // cut into maximal runs that are contiguous among themselves (stride[k] == dims[k+1] * stride[k+1]); every new non-unit dim must
    // fall inside one run (row-major inside it).  The usual no-copy reshape rule.
    static bool view_reshape(const Dims &a, const Dims &as, const Dims &nd, Dims &out) {
        out.assign(nd.size(), 0);
        size_t i = 0, j = 0;
        while (j < nd.size() && nd[j] == 1) j++;
        while (i < a.size() && j < nd.size()) {
            // old run [i, e): contiguous among themselves
            size_t e = i + 1;
            int64_t run = a[i];
            // extend the new side until the products match, extending the old run when the new product overshoots
            size_t j0 = j;
            int64_t np = nd[j];
            size_t je = j + 1;
            while (np != run) {
                if (np < run) {
                    while (je < nd.size() && nd[je] == 1) je++;
                    if (je >= nd.size()) return false;
                    np *= nd[je++];
                } else {
                    if (e >= a.size() || as[e - 1] != a[e] * as[e]) return false;
                    run *= a[e++];
                }
            }
            // strides of the new dims inside the run: row-major, innermost = the run's last old stride
            int64_t st = as[e - 1];
            for (size_t k = je; k-- > j0;) {
                if (nd[k] == 1) continue;
                out[k] = st;
                st *= nd[k];
            }
            i = e;
            j = je;
            while (j < nd.size() && nd[j] == 1) j++;
        }
        return i == a.size() && j == nd.size();
    }

    // Gather of an activation with CONSTANT indices that form an affine pattern idx[i0, i1, ...] = base + sum_k i_k * step_k: a strided
    // view, no copy.  This is what framing looks like in a TensorFlow export (tf.signal.frame gathers sub-frames with the selector
    // frame * (hop / sub) + j); anything that is not affine is refused.
    void lower_gather(const OnnxNode &n) {
        const Val &x = get(n, 0);
        const Val &iv = get(n, 1);
        if (x.is_const || !iv.is_const) unsupported(n, "Gather is mapped for an activation with constant indices only");
        int64_t axis = n.attr_i("axis", 0);
        const int64_t r = (int64_t)x.dims.size() + 1;
        if (axis < 0) axis += r;
        if (axis <= 0 || axis >= r) unsupported(n, "Gather along the batch dimension");
        axis -= 1;
        const int64_t D = x.dims[axis];
        std::vector<int64_t> idx = const_ints(n, iv);
        for (auto &v : idx) { if (v < 0) v += D; if (v < 0 || v >= D) unsupported(n, "Gather index out of range"); }
        const Dims &id = iv.dims;
        Dims irs = row_major(id), step(id.size(), 0);
        const int64_t base = idx.empty() ? 0 : idx[0];
        for (size_t k = 0; k < id.size(); k++) step[k] = id[k] > 1 ? idx[(size_t)irs[k]] - base : 0;
        for (size_t lin = 0; lin < idx.size(); lin++) {
            int64_t want = base, rem = (int64_t)lin;
            for (size_t k = 0; k < id.size(); k++) { want += (rem / irs[k]) * step[k]; rem %= irs[k]; }
            if (idx[lin] != want) unsupported(n, "Gather indices are not an affine pattern (only strided views of an activation are mapped; tf.signal.frame's selector is one)");
        }
        Val o = x;
        o.dims.assign(x.dims.begin(), x.dims.begin() + axis);
        o.strides.assign(x.strides.begin(), x.strides.begin() + axis);
        for (size_t k = 0; k < id.size(); k++) { o.dims.push_back(id[k]); o.strides.push_back(step[k] * x.strides[axis]); }
        o.dims.insert(o.dims.end(), x.dims.begin() + axis + 1, x.dims.end());
        o.strides.insert(o.strides.end(), x.strides.begin() + axis + 1, x.strides.end());
        o.offset = x.offset + base * x.strides[axis];
        define(n.outputs[0], o);
    }

    // Mul(frames, window[L]) whose result reaches a DFT node through nothing but unit-dimension reshapes: leave the window pending on the
    // value -- lower_dft folds it into its basis, which is what lets the framing kernels (mirror / quarter fold, FFT) recognise the bank
    bool try_pending_window(const OnnxNode &n, const Val &a, const Val &b) {
        const Val &x = a.is_const ? b : a, &w = a.is_const ? a : b;
        if (x.is_const || !w.is_const || w.is_int || !x.win_name.empty() || x.gate_storage >= 0 || wanted_names_.count(n.outputs[0])) return false;
        const int64_t L = w.numel();
        if (L < 2) return false;
        // the window's one non-unit dim, aligned from the right against the full (batched) dims of x
        int wax = -1;
        for (size_t k = 0; k < w.dims.size(); k++)
            if (w.dims[k] != 1) { if (wax >= 0) return false; wax = (int)k; }
        const int ax = wax + (int)(x.dims.size() + 1) - (int)w.dims.size() - 1;  // per-sample axis of x
        if (ax < 0 || ax >= (int)x.dims.size() || x.dims[ax] != L) return false;
        std::string cur = n.outputs[0];
        for (int hop = 0; hop < 4; hop++) {
            int c = sole_consumer(cur);
            if (c < 0 || nodes_[c].inputs.empty() || nodes_[c].inputs[0] != cur) return false;
            const std::string &ct = nodes_[c].op_type;
            if (ct == "DFT") {
                Val o = x;
                o.win_name = n.inputs[a.is_const ? 0 : 1];
                o.win_nu = 0;
                for (int k = 0; k < ax; k++) o.win_nu += x.dims[k] != 1;
                define(n.outputs[0], o);
                return true;
            }
            if (ct != "Unsqueeze" && ct != "Squeeze" && ct != "Reshape" && ct != "Identity") return false;
            cur = nodes_[c].outputs[0];
        }
        return false;
    }

    // ONNX DFT (opset 17: attribute axis, default 1; opset 20: input axis, default -2): forward, real input [.., L, 1], optional constant
    // dft_length N (zero-padded or truncated signal), onesided -> N/2 + 1 bins, output [.., bins, 2].  The frames [F, L] along `axis` are
    // rows of a framing convolution over the flat view they come from (hop = frame stride / sample stride: overlapping views of a signal
    // -- tf.signal.frame through lower_gather -- or hop = L for a materialised tensor), exactly like lower_stft: one cos | -sin bank built in
    // double precision with a pending window folded in, so the mirror / quarter folds and the FFT kernel apply to it like to any bank.
    void lower_dft(const OnnxNode &n) {
        Val x = get(n, 0);
        if (x.is_const) unsupported(n, "DFT of a constant signal");
        if (n.attr_i("inverse", 0) != 0) unsupported(n, "inverse DFT is outside the native subset (no BirdNET / Perch front end needs it)");
        const int64_t r = (int64_t)x.dims.size() + 1;
        if (x.dims.empty() || x.dims.back() == 2) unsupported(n, "complex input (last dimension 2) is outside the native subset; the front ends transform a real signal");
        if (x.dims.back() != 1) unsupported(n, "input must end in a dimension of 1 (real) per the operator's definition, got per-sample dims " + dims_str(x.dims));
        int64_t axis = n.has("axis") ? n.attr_i("axis", 1) : (m_.opset >= 20 ? -2 : 1);
        if (!n.has("axis") && has_input(n, 2)) axis = const_ints(n, get(n, 2)).at(0);
        if (axis < 0) axis += r;
        if (axis <= 0 || axis >= r - 1) unsupported(n, "DFT axis must be a signal dimension (not the batch, not the trailing real / imaginary one)");
        axis -= 1;
        const int64_t L = x.dims[axis];
        int64_t N = L;
        if (has_input(n, 1)) { const Val &dl = get(n, 1); if (!dl.is_const) unsupported(n, "dft_length must be a constant"); N = const_ints(n, dl).at(0); }
        if (N <= 0) unsupported(n, "dft_length must be positive");
        const int64_t taps = std::min(L, N);  // a longer transform zero-pads the signal (= fewer taps per basis row), a shorter one truncates it
        const bool onesided = n.attr_i("onesided", 0) != 0;
        const int64_t bins = onesided ? N / 2 + 1 : N;
        if (taps * 2 * bins > ((int64_t)1 << 28)) unsupported(n, "DFT basis too large");
        // pending window: must lie along the transformed axis and have its length
        std::vector<float> win;
        if (!x.win_name.empty()) {
            int nu = 0;
            for (int64_t k = 0; k < axis; k++) nu += x.dims[k] != 1;
            const Val &w = vals_.at(x.win_name);
            if (nu == x.win_nu && w.numel() == L) win = w.f;
            else x = apply_window(x, n.inputs[0]);
            x.win_name.clear();
            x.win_nu = -1;
        }
        // frames: every non-unit dim except `axis` and the trailing 1, merged into one row index with ONE stride if the view allows it
        Dims lead, lead_s;
        for (int64_t k = 0; k + 1 < (int64_t)x.dims.size(); k++)
            if (k != axis && x.dims[k] != 1) { lead.push_back(x.dims[k]); lead_s.push_back(x.strides[k]); }
        int64_t F = 1;
        for (auto d : lead) F *= d;
        int64_t sL = x.strides[axis], sF = lead.empty() ? L * sL : lead_s.back();
        bool rows_ok = sL > 0 && sF > 0 && sF % sL == 0;
        for (size_t k = 0; k + 1 < lead.size() && rows_ok; k++) rows_ok = lead_s[k] == lead[k + 1] * lead_s[k + 1];
        if (!rows_ok) {  // arbitrary view: copy the frames into [lead..., L] row-major, then they are non-overlapping rows
            Dims keep = x.dims;
            Val c = materialize(x, row_major(keep), n.name + "/frames");
            x = c;
            lead_s.clear();
            for (int64_t k = 0; k + 1 < (int64_t)x.dims.size(); k++)
                if (k != axis && x.dims[k] != 1) lead_s.push_back(x.strides[k]);
            sL = x.strides[axis];
            sF = lead.empty() ? L * sL : lead_s.back();
            rows_ok = sL > 0 && sF % sL == 0;
            for (size_t k = 0; k + 1 < lead.size() && rows_ok; k++) rows_ok = lead_s[k] == lead[k + 1] * lead_s[k + 1];
            if (!rows_ok) unsupported(n, "DFT axis must be the innermost signal dimension of its frames (transpose the frames first)");
        }
        const int64_t hop = sF / sL;
        Val w;
        w.is_const = true;
        w.dims = {2 * bins, 1, taps};
        w.f.resize((size_t)(2 * bins * taps));
        for (int64_t k = 0; k < bins; k++)
            for (int64_t t = 0; t < taps; t++) {
                const double wv = win.empty() ? 1.0 : (double)win[(size_t)t];
                const double th = 2.0 * M_PI * (double)((k * t) % N) / (double)N;
                w.f[(size_t)(k * taps + t)] = (float)(wv * std::cos(th));
                w.f[(size_t)((bins + k) * taps + t)] = (float)(-wv * std::sin(th));
            }
        const std::string base = "dft:" + (n.name.empty() ? n.outputs[0] : n.name);
        const int64_t span = (F - 1) * hop + taps;
        Val xv = x;
        xv.dims = {1, span};
        xv.strides = {span * sL, sL};
        vals_[base + "/signal"] = xv;
        vals_[base + "/basis"] = std::move(w);
        OnnxNode conv;
        conv.name = base;
        conv.op_type = "Conv";
        conv.inputs = {base + "/signal", base + "/basis"};
        conv.outputs = {base + "/frames"};
        set_ints(conv, "strides", {hop});
        lower_conv(conv);
        auto yit = vals_.find(base + "/frames");
        if (yit == vals_.end()) unsupported(n, "internal: the framing conv defined no output");
        const Val y = yit->second;
        if (y.is_const || y.dims.size() != 2 || y.dims[0] != 2 * bins || y.dims[1] != F) unsupported(n, "internal: framing conv produced " + dims_str(y.dims));
        // the node's result: x's dims with `axis` -> bins and the trailing 1 -> 2, as a strided view of the conv's [2 bins, F] output
        Val out = y;
        out.dims = x.dims;
        out.strides.assign(x.dims.size(), 0);
        int64_t fs = y.strides[1];
        for (int64_t k = (int64_t)x.dims.size() - 2; k >= 0; k--) {
            if (k == axis || x.dims[k] == 1) continue;
            out.strides[k] = fs;
            fs *= x.dims[k];
        }
        out.dims[axis] = bins;
        out.strides[axis] = y.strides[0];
        out.dims.back() = 2;
        out.strides.back() = bins * y.strides[0];
        define(n.outputs[0], out);
    }

    void lower_transpose(const OnnxNode &n) {
        const Val &v = get(n, 0);
        auto perm = n.attr_ints("perm");
        size_t r = v.dims.size() + 1;
        if (perm.empty()) for (size_t k = 0; k < r; k++) perm.push_back((int64_t)(r - 1 - k));
        if (perm.size() != r || perm[0] != 0) unsupported(n, "Transpose must keep the batch as the leading dimension");
        Val o = v;
        for (size_t k = 1; k < r; k++) { o.dims[k - 1] = v.dims[perm[k] - 1]; o.strides[k - 1] = v.strides[perm[k] - 1]; }
        define(n.outputs[0], o);
    }

    void slice_params(const OnnxNode &n, const Dims &dims, const Dims &strides, Dims &nd, Dims &ns, int64_t &off, bool batched) {
        std::vector<int64_t> starts, ends, axes, steps;
        if (n.has("starts")) { starts = n.attr_ints("starts"); ends = n.attr_ints("ends"); axes = n.attr_ints("axes"); }
        else {
            starts = const_ints(n, get(n, 1));
            ends = const_ints(n, get(n, 2));
            if (has_input(n, 3)) axes = const_ints(n, get(n, 3));
            if (has_input(n, 4)) steps = const_ints(n, get(n, 4));
        }
        int64_t r = (int64_t)dims.size() + (batched ? 1 : 0);
        if (axes.empty()) for (size_t k = 0; k < starts.size(); k++) axes.push_back((int64_t)k);
        if (steps.empty()) steps.assign(starts.size(), 1);
        nd = dims; ns = strides; off = 0;
        for (size_t k = 0; k < starts.size(); k++) {
            int64_t ax = axes[k] < 0 ? axes[k] + r : axes[k];
            if (batched) { if (ax == 0) unsupported(n, "Slice along the batch dimension"); ax -= 1; }
            int64_t d = dims[ax], st = steps[k], s = starts[k], e = ends[k];
            if (st == 0) unsupported(n, "Slice step 0");
            if (s < 0) s += d;
            if (e < 0) e += d;
            int64_t cnt;
            if (st > 0) { s = std::clamp<int64_t>(s, 0, d); e = std::clamp<int64_t>(e, 0, d); cnt = e > s ? (e - s + st - 1) / st : 0; }
            else {
                // ONNX: clamp start to [0, d-1], end to [-1, d-1] for negative steps
                int64_t s0 = starts[k] < 0 ? starts[k] + d : starts[k];
                int64_t e0 = ends[k];
                if (e0 < -d - 1) e0 = -1; else if (e0 < 0) e0 += d;
                s = std::clamp<int64_t>(s0, 0, d - 1);
                e = std::clamp<int64_t>(e0, -1, d - 1);
                cnt = s > e ? (s - e + (-st) - 1) / (-st) : 0;
            }
            if (cnt <= 0) unsupported(n, "Slice produces an empty tensor");
            off += s * strides[ax];
            nd[ax] = cnt;
            ns[ax] = strides[ax] * st;
        }
    }
    void lower_slice(const OnnxNode &n) {
        const Val &v = get(n, 0);
        Val o = v;
        int64_t off;
        slice_params(n, v.dims, v.strides, o.dims, o.strides, off, true);
        o.offset = v.offset + off;
        define(n.outputs[0], o);
    }

    void lower_concat(const OnnxNode &n) {
        std::vector<Val> ins;
        for (size_t k = 0; k < n.inputs.size(); k++) {
            Val v = get(n, k);
            if (v.is_const) v = upload_as_activation(v);
            ins.push_back(v);
        }
        int64_t r = (int64_t)ins[0].dims.size() + 1;
        int64_t axis = n.attr_i("axis", 0);
        if (axis < 0) axis += r;
        if (axis == 0) unsupported(n, "Concat along the batch dimension");
        axis -= 1;
        Dims od = ins[0].dims;
        od[axis] = 0;
        for (auto &v : ins) {
            if (v.dims.size() != od.size()) unsupported(n, "rank mismatch");
            od[axis] += v.dims[axis];
        }
        // layout: channels-last when the result (through elementwise ops) feeds a convolution,
        // else follow the first input's physical order
        bool to_conv = false;
        {
            std::string cur = n.outputs[0];
            for (int hop = 0; hop < 6 && !to_conv; hop++) {
                int c = sole_consumer(cur);
                if (c < 0) break;
                const OnnxNode &cn = nodes_[c];
                if (cn.op_type == "Conv" && cn.inputs[0] == cur) to_conv = true;
                else if (cn.inputs[0] == cur && (cn.op_type == "BatchNormalization" || cn.op_type == "Relu" || cn.op_type == "Clip" ||
                                                 cn.op_type == "Mul" || cn.op_type == "Add" || cn.op_type == "Sub" || cn.op_type == "Div" ||
                                                 cn.op_type == "Sigmoid" || cn.op_type == "Pow" || cn.op_type == "Log" || cn.op_type == "Sqrt"))
                    cur = cn.outputs[0];
                else break;
            }
        }
        Dims ostr = strides_for_order(od, phys_order(ins[0]));
        if (to_conv && od.size() == 3) ostr = Dims{1, od[2] * od[0], od[0]};
        else if (to_conv && od.size() == 2) ostr = Dims{1, od[0]};
        Val out = new_act(od, ostr);
        // Concat of single-channel planes along the channel axis followed by BatchNormalization (the v2.4 spectrogram
        // image): every plane sees ONE scale and shift, so the normalisation rides on the copy of each plane (and, with
        // rule E, ends up in the epilogue of the GEMM that produced it) instead of one more pass over the image.
        std::vector<float> plane_s, plane_t;
        std::string result = n.outputs[0];
        if (axis == 0 && od[0] == (int64_t)ins.size()) {
            int c = sole_consumer(n.outputs[0]);
            if (c >= 0 && nodes_[c].op_type == "BatchNormalization" && nodes_[c].inputs.size() >= 5 && nodes_[c].inputs[0] == n.outputs[0]) {
                const OnnxNode &bn_ = nodes_[c];
                const Val &sc = get(bn_, 1), &bi = get(bn_, 2), &mu = get(bn_, 3), &var = get(bn_, 4);
                if (sc.is_const && bi.is_const && mu.is_const && var.is_const && sc.numel() == od[0] && bi.numel() == od[0] &&
                    mu.numel() == od[0] && var.numel() == od[0] && bn_.outputs.size() == 1) {
                    const float eps = bn_.attr_f("epsilon", 1e-5f);
                    for (int64_t k = 0; k < od[0]; k++) {
                        const float sk = sc.f[k] / std::sqrt(var.f[k] + eps);
                        plane_s.push_back(sk);
                        plane_t.push_back(bi.f[k] - mu.f[k] * sk);
                    }
                    absorbed_[c] = true;
                    result = bn_.outputs[0];
                }
            }
        }
        int64_t pos = 0;
        for (auto &v : ins) {
            Val slot = out;
            slot.dims = v.dims;
            slot.offset = out.offset + pos * out.strides[axis];
            ActSpec plane_act;
            std::string what = "concat:" + n.name;
            if (!plane_s.empty()) {
                plane_act.act = ACT_AFFINE; plane_act.p0 = plane_s[(size_t)pos]; plane_act.p1 = plane_t[(size_t)pos];
                what += "+bn";
            }
            emit_elt(what, slot, ref_of(v), v.strides, batch_stride(v), Ref{}, {}, 0, BIN_NONE, plane_act);
            pos += v.dims[axis];
        }
        define(result, out);
    }

    // Padded copy of an activation (per-sample dims): the whole result is filled with `value`, then
    // the interior is overwritten by the input.  Two strided elementwise launches, no new kernel.
    Val pad_copy(Val v, const Dims &lo, const Dims &hi, float value, const std::string &name) {
        if (v.is_const) v = upload_as_activation(v);
        if (v.gate_storage >= 0) v = apply_gate(v, name);
        Dims od = v.dims;
        for (size_t i = 0; i < od.size(); i++) od[i] += lo[i] + hi[i];
        Val out = new_act(od, strides_for_order(od, phys_order(v)));
        Ref fill{Space::CONSTS, add_const(std::vector<float>{value}), 0};
        emit_elt("pad.fill:" + name, out, fill, Dims(od.size(), 0), 0, Ref{}, {}, 0, BIN_NONE, ActSpec{});
        Val slot = out;
        slot.dims = v.dims;
        for (size_t i = 0; i < od.size(); i++) slot.offset += lo[i] * out.strides[i];
        emit_elt("pad.copy:" + name, slot, ref_of(v), v.strides, batch_stride(v), Ref{}, {}, 0, BIN_NONE, ActSpec{});
        return out;
    }

    // ONNX Pad, constant mode (pads as attribute for opset < 11, as input otherwise)
    void lower_pad(const OnnxNode &n) {
        Val x = get(n, 0);
        if (n.attr_s("mode", "constant") != "constant") unsupported(n, "only constant-mode Pad is supported");
        std::vector<int64_t> pads = n.attr_ints("pads");
        float value = n.attr_f("value", 0.0f);
        if (pads.empty()) {
            const Val *pv = opt(n, 1);
            if (!pv || !pv->is_const) unsupported(n, "Pad needs constant pads");
            pads = pv->i;
            const Val *cv = opt(n, 2);
            if (cv) { if (!const_scalar(*cv, value)) unsupported(n, "Pad needs a constant fill value"); }
        }
        const size_t r = x.dims.size() + 1;  // with the batch dimension
        std::vector<int64_t> axes;
        if (const Val *av = opt(n, 3)) {
            if (!av->is_const) unsupported(n, "Pad needs constant axes");
            axes = av->i;
        } else {
            for (size_t i = 0; i < r; i++) axes.push_back((int64_t)i);
        }
        if (pads.size() != 2 * axes.size()) unsupported(n, "pads/axes size mismatch");
        Dims lo(x.dims.size(), 0), hi(x.dims.size(), 0);
        for (size_t k = 0; k < axes.size(); k++) {
            int64_t ax = axes[k] < 0 ? axes[k] + (int64_t)r : axes[k];
            const int64_t p0 = pads[k], p1 = pads[axes.size() + k];
            if (p0 < 0 || p1 < 0) unsupported(n, "negative pads");
            if (ax == 0) { if (p0 || p1) unsupported(n, "Pad along the batch dimension"); continue; }
            lo[ax - 1] = p0; hi[ax - 1] = p1;
        }
        define(n.outputs[0], pad_copy(x, lo, hi, value, n.name));
    }

    // MaxPool / AveragePool (1-D or 2-D, no dilation, floor mode) on a channels-last tensor.
    void lower_pool(const OnnxNode &n) {
        Val x = get(n, 0);
        if (x.is_const) unsupported(n, "pooling of a constant");
        if (x.gate_storage >= 0) x = apply_gate(x, n.name);
        const size_t sp = x.dims.size() - 1;
        if (sp != 1 && sp != 2) unsupported(n, "only 1-D and 2-D pooling is supported");
        if (n.attr_i("ceil_mode", 0) != 0) unsupported(n, "ceil_mode pooling");
        for (auto dd : n.attr_ints("dilations")) if (dd != 1) unsupported(n, "dilated pooling");
        if (n.outputs.size() > 1 && !n.outputs[1].empty()) unsupported(n, "MaxPool indices output");
        auto ks = n.attr_ints("kernel_shape");
        if (ks.size() != sp) unsupported(n, "kernel_shape rank mismatch");
        auto st = n.attr_ints("strides");
        if (st.empty()) st.assign(sp, 1);
        auto pads = n.attr_ints("pads");
        if (pads.empty()) pads.assign(2 * sp, 0);
        if (st.size() != sp || pads.size() != 2 * sp) unsupported(n, "strides / pads do not match the spatial rank");
        for (auto e : ks) if (e <= 0) unsupported(n, "kernel_shape must be positive");
        for (auto e : st) if (e <= 0) unsupported(n, "strides must be positive");
        for (auto e : pads) if (e < 0) unsupported(n, "negative pads");
        const int64_t C = x.dims[0], H = sp == 2 ? x.dims[1] : 1, W = sp == 2 ? x.dims[2] : x.dims[1];
        const int64_t kh = sp == 2 ? ks[0] : 1, kw = sp == 2 ? ks[1] : ks[0];
        const int64_t sh = sp == 2 ? st[0] : 1, sw = sp == 2 ? st[1] : st[0];
        int64_t pt = sp == 2 ? pads[0] : 0, pl = sp == 2 ? pads[1] : pads[0], pb = sp == 2 ? pads[2] : 0, pr = sp == 2 ? pads[3] : pads[1];
        const std::string auto_pad = n.attr_s("auto_pad", "NOTSET");
        auto out_dim = [&](int64_t in, int64_t k, int64_t s_, int64_t &p0, int64_t &p1) {
            if (auto_pad == "SAME_UPPER" || auto_pad == "SAME_LOWER") {
                const int64_t o = (in + s_ - 1) / s_;
                const int64_t tot = std::max<int64_t>((o - 1) * s_ + k - in, 0);
                p0 = auto_pad == "SAME_UPPER" ? tot / 2 : tot - tot / 2;
                p1 = tot - p0;
                return o;
            }
            if (auto_pad == "VALID") p0 = p1 = 0;
            return (in + p0 + p1 - k) / s_ + 1;
        };
        const int64_t OH = out_dim(H, kh, sh, pt, pb), OW = out_dim(W, kw, sw, pl, pr);
        if (OH <= 0 || OW <= 0) unsupported(n, "empty pooling output");
        x = to_channels_last(x, n.name);
        Dims od = sp == 2 ? Dims{C, OH, OW} : Dims{C, OW};
        Dims ostr = sp == 2 ? Dims{1, OW * C, C} : Dims{1, C};
        Val out = new_act(od, ostr);
        PlanOp op;
        op.kind = OpKind::POOL;
        op.name = n.op_type + ":" + n.name;
        op.out = ref_of(out);
        op.a = ref_of(x);
        PoolDesc &d = op.pool;
        d.H = (int32_t)H; d.W = (int32_t)W; d.C = (int32_t)C; d.OH = (int32_t)OH; d.OW = (int32_t)OW;
        d.kh = (int32_t)kh; d.kw = (int32_t)kw; d.sh = (int32_t)sh; d.sw = (int32_t)sw; d.pt = (int32_t)pt; d.pl = (int32_t)pl;
        d.is_max = n.op_type == "MaxPool" ? 1 : 0;
        d.count_include_pad = (int32_t)n.attr_i("count_include_pad", 0);
        d.in_bs = batch_stride(x); d.out_bs = plan_.storages[out.storage].elems;
        op.macs = 0;
        op.bytes = 4.0 * ((double)C * H * W + (double)C * OH * OW);
        push_op(std::move(op));
        define(n.outputs[0], out);
    }

    // Softmax / LogSoftmax over one non-batch axis, expanded into the launches the engine already has
    // (max -> x - max -> exp -> sum -> divide, or ... -> log(sum) -> subtract); the elementwise-chain pass fuses
    // the neighbours.  Opset < 13 semantics (flatten from `axis`) coincide with this when `axis` is the last one.
    void lower_softmax(const OnnxNode &n) {
        const Val &x = get(n, 0);
        if (x.is_const) unsupported(n, "Softmax of a constant");
        const int64_t r = (int64_t)x.dims.size() + 1;
        int64_t axis = n.attr_i("axis", -1);
        if (axis < 0) axis += r;
        if (axis <= 0 || axis >= r) unsupported(n, "Softmax over the batch dimension");
        auto mk = [&](const char *op, const std::string &suffix, std::vector<std::string> ins, const std::string &out) {
            OnnxNode q;
            q.op_type = op;
            q.name = n.name + "/" + suffix;
            q.inputs = std::move(ins);
            q.outputs = {out};
            return q;
        };
        auto with_axes = [&](OnnxNode q) {
            OnnxAttr a;
            a.name = "axes"; a.type = 7; a.ints = {axis};
            q.attrs["axes"] = a;
            OnnxAttr k;
            k.name = "keepdims"; k.type = 2; k.i = 1;
            q.attrs["keepdims"] = k;
            return q;
        };
        const std::string b = n.outputs[0] + "/sm.";
        lower_reduce(with_axes(mk("ReduceMax", "max", {n.inputs[0]}, b + "max")));
        lower_binary(mk("Sub", "shift", {n.inputs[0], b + "max"}, b + "shift"));
        ActSpec e; e.act = ACT_EXP;
        lower_unary(mk("Exp", "exp", {b + "shift"}, b + "exp"), e);
        lower_reduce(with_axes(mk("ReduceSum", "sum", {b + "exp"}, b + "sum")));
        if (n.op_type == "Softmax") {
            lower_binary(mk("Div", "div", {b + "exp", b + "sum"}, n.outputs[0]));
        } else {
            ActSpec l; l.act = ACT_LOG;
            lower_unary(mk("Log", "log", {b + "sum"}, b + "lse"), l);
            lower_binary(mk("Sub", "sub", {b + "shift", b + "lse"}, n.outputs[0]));
        }
    }

    // Split along a non-batch axis: every output is a view (no launch).
    void lower_split(const OnnxNode &n) {
        const Val &v = get(n, 0);
        const int64_t r = (int64_t)v.dims.size() + 1;
        int64_t axis = n.attr_i("axis", 0);
        if (axis < 0) axis += r;
        if (axis <= 0 || axis >= r) unsupported(n, "Split along the batch dimension");
        axis -= 1;
        std::vector<int64_t> sizes = n.attr_ints("split");
        if (sizes.empty() && has_input(n, 1)) sizes = const_ints(n, get(n, 1));
        const int64_t nout = (int64_t)n.outputs.size();
        if (sizes.empty()) {
            const int64_t each = (v.dims[axis] + nout - 1) / nout;
            for (int64_t k = 0; k < nout; k++) sizes.push_back(std::min(each, v.dims[axis] - k * each));
        }
        if ((int64_t)sizes.size() != nout) unsupported(n, "split sizes do not match the outputs");
        int64_t pos = 0;
        for (int64_t k = 0; k < nout; k++) {
            if (sizes[k] < 0 || pos + sizes[k] > v.dims[axis]) unsupported(n, "split sizes exceed the dimension");
            if (v.is_const) unsupported(n, "Split of a constant");
            Val o = v;
            o.dims[axis] = sizes[k];
            o.offset = v.offset + pos * v.strides[axis];
            if (!n.outputs[k].empty()) define(n.outputs[k], o);
            pos += sizes[k];
        }
    }

    bool unary_spec(const OnnxNode &n, ActSpec &a) {
        const std::string &t = n.op_type;
        if (t == "Relu") a.act = ACT_RELU;
        else if (t == "Sigmoid") a.act = ACT_SIGMOID;
        else if (t == "Tanh") a.act = ACT_TANH;
        else if (t == "Exp") a.act = ACT_EXP;
        else if (t == "Log") a.act = ACT_LOG;
        else if (t == "Sqrt") a.act = ACT_SQRT;
        else if (t == "Abs") a.act = ACT_ABS;
        else if (t == "Neg") a.act = ACT_NEG;
        else if (t == "Reciprocal") a.act = ACT_RECIP;
        else if (t == "Floor") a.act = ACT_FLOOR;
        else if (t == "Ceil") a.act = ACT_CEIL;
        else if (t == "Round") a.act = ACT_ROUND;
        else if (t == "Erf") a.act = ACT_ERF;
        else if (t == "Softplus") a.act = ACT_SOFTPLUS;
        else if (t == "HardSwish") a.act = ACT_HSWISH;
        else if (t == "HardSigmoid") { a.act = ACT_HSIGMOID; a.p0 = n.attr_f("alpha", 0.2f); a.p1 = n.attr_f("beta", 0.5f); }
        else if (t == "LeakyRelu") { a.act = ACT_LEAKY; a.p0 = n.attr_f("alpha", 0.01f); }
        else if (t == "Clip") {
            float lo = -INFINITY, hi = INFINITY;
            if (n.has("min")) lo = n.attr_f("min", lo);
            if (n.has("max")) hi = n.attr_f("max", hi);
            if (has_input(n, 1)) { if (!const_scalar(get(n, 1), lo)) return false; }
            if (has_input(n, 2)) { if (!const_scalar(get(n, 2), hi)) return false; }
            a.act = ACT_CLIP; a.p0 = lo; a.p1 = hi;
        } else return false;
        return true;
    }
    // activation pattern starting at tensor `cur`: returns true and the final tensor name if absorbed
    bool absorb_activation(std::string &cur, ActSpec &a) {
        auto cons = live_consumers(cur);
        if (wanted_names_.count(cur)) return false;
        if (cons.size() == 1) {
            const OnnxNode &c = nodes_[cons[0]];
            if (c.inputs.empty() || c.inputs[0] != cur) return false;
            ActSpec s;
            if (c.op_type == "Sigmoid" || c.op_type == "Relu" || c.op_type == "Clip" || c.op_type == "HardSwish" ||
                c.op_type == "HardSigmoid" || c.op_type == "LeakyRelu" || c.op_type == "Tanh") {
                if (c.op_type == "Sigmoid") {
                    // keep a bare Sigmoid fusable too
                }
                if (!unary_spec(c, s)) return false;
                a = s;
                absorbed_[cons[0]] = true;
                cur = c.outputs[0];
                return true;
            }
            return false;
        }
        if (cons.size() == 2) {  // x * sigmoid(x)
            int si = -1, mi = -1;
            for (int k : cons) { if (nodes_[k].op_type == "Sigmoid") si = k; else if (nodes_[k].op_type == "Mul") mi = k; }
            if (si < 0 || mi < 0) return false;
            const OnnxNode &sg = nodes_[si], &ml = nodes_[mi];
            if (sole_consumer(sg.outputs[0]) != mi) return false;
            bool ok = (ml.inputs[0] == cur && ml.inputs[1] == sg.outputs[0]) || (ml.inputs[1] == cur && ml.inputs[0] == sg.outputs[0]);
            if (!ok) return false;
            a.act = ACT_SILU;
            absorbed_[si] = absorbed_[mi] = true;
            cur = ml.outputs[0];
            return true;
        }
        return false;
    }

    void lower_unary(const OnnxNode &n, ActSpec a) {
        const Val &v = get(n, 0);
        if (v.is_const) unsupported(n, "constant operand not foldable");
        Val out = new_act(v.dims, strides_for_order(v.dims, phys_order(v)));
        emit_elt(n.op_type + ":" + n.name, out, ref_of(v), v.strides, batch_stride(v), Ref{}, {}, 0, BIN_NONE, a);
        define(n.outputs[0], out);
    }

    void lower_binary(const OnnxNode &n) {
        const Val &a = get(n, 0), &b = get(n, 1);
        const std::string &t = n.op_type;
        if (t == "Mul" && (a.is_gate != b.is_gate)) {
            // x * se_gate: leave it pending on the value; the consuming 1x1 conv folds it into its
            // operand load, any other consumer triggers apply_gate() in get()
            const Val &x = a.is_gate ? b : a;
            const Val &g = a.is_gate ? a : b;
            if (!x.is_const && x.gate_storage < 0 && x.dims.size() == 3 && x.dims[0] == g.dims[0] && x.space == Space::ARENA) {
                Val o = x;
                o.gate_storage = g.storage;
                if (wanted_names_.count(n.outputs[0])) o = apply_gate(o, n.name);
                define(n.outputs[0], o);
                return;
            }
        }
        if (t == "Mul" && a.is_const != b.is_const && try_pending_window(n, a, b)) return;
        int bin = t == "Add" ? BIN_ADD : t == "Sub" ? BIN_SUB : t == "Mul" ? BIN_MUL : t == "Div" ? BIN_DIV : t == "Pow" ? BIN_POW : t == "Max" ? BIN_MAX
                : t == "And" ? BIN_MUL : t == "Or" ? BIN_MAX : compare_code(t) != BIN_NONE ? compare_code(t) : BIN_MIN;
        // scalar constants become parametrised unary ops
        float c;
        if (!a.is_const && const_scalar(b, c)) {
            ActSpec s;
            bool ok = true;
            switch (bin) {
                case BIN_ADD: s = {ACT_AFFINE, 1.0f, c}; break;
                case BIN_SUB: s = {ACT_AFFINE, 1.0f, -c}; break;
                case BIN_MUL: s = {ACT_AFFINE, c, 0.0f}; break;
                case BIN_MAX: s = {ACT_MAXC, c, 0}; break;
                case BIN_MIN: s = {ACT_MINC, c, 0}; break;
                case BIN_POW:
                    if (c == 2.0f) s = {ACT_SQUARE, 0, 0};
                    else if (c == 0.5f) s = {ACT_SQRT, 0, 0};
                    else if (c == 1.0f) s = {ACT_AFFINE, 1.0f, 0.0f};
                    else s = {ACT_POW, c, 0};
                    break;
                case BIN_GT: s = {ACT_GTC, c, 0}; break;
                case BIN_LT: s = {ACT_LTC, c, 0}; break;
                case BIN_GE: s = {ACT_GEC, c, 0}; break;
                case BIN_LE: s = {ACT_LEC, c, 0}; break;
                case BIN_EQ: s = {ACT_EQC, c, 0}; break;
                default: ok = false;
            }
            if (ok && (t == "And" || t == "Or")) ok = false;  // (logic on 0 / 1 tensors takes the general path)
            if (ok) return lower_unary(n, s);
        }
        if (!b.is_const && const_scalar(a, c)) {
            ActSpec s;
            bool ok = true;
            switch (bin) {
                case BIN_ADD: s = {ACT_AFFINE, 1.0f, c}; break;
                case BIN_MUL: s = {ACT_AFFINE, c, 0.0f}; break;
                case BIN_SUB: s = {ACT_RSUB, c, 0}; break;
                case BIN_DIV: s = {ACT_RDIV, c, 0}; break;
                case BIN_MAX: s = {ACT_MAXC, c, 0}; break;
                case BIN_MIN: s = {ACT_MINC, c, 0}; break;
                case BIN_GT: s = {ACT_LTC, c, 0}; break;  // c > x
                case BIN_LT: s = {ACT_GTC, c, 0}; break;
                case BIN_GE: s = {ACT_LEC, c, 0}; break;
                case BIN_LE: s = {ACT_GEC, c, 0}; break;
                case BIN_EQ: s = {ACT_EQC, c, 0}; break;
                default: ok = false;
            }
            if (ok && (t == "And" || t == "Or")) ok = false;
            if (ok) {
                OnnxNode sw = n;
                std::swap(sw.inputs[0], sw.inputs[1]);
                return lower_unary(sw, s);
            }
        }
        // general broadcast; result dims
        auto act_dims = [&](const Val &v) { Dims d = v.dims; if (v.is_const) { if (d.size() > 0 && (int64_t)d.size() > 0) {} } return d; };
        (void)act_dims;
        // Normalise both to per-sample dims: constants may carry a leading batch dim of 1.
        auto per_sample = [&](const Val &v, size_t rank_hint) {
            Dims d = v.dims;
            if (v.is_const && d.size() == rank_hint + 1) {
                if (d[0] != 1) unsupported(n, "constant operand has a non-unit batch dimension");
                d.erase(d.begin());
            }
            return d;
        };
        size_t ra = a.is_const ? 0 : a.dims.size(), rb = b.is_const ? 0 : b.dims.size();
        size_t rank = std::max(ra, rb);
        Dims da = per_sample(a, rank), db = per_sample(b, rank);
        if (da.size() > rank || db.size() > rank) {
            // constant with more dims than the activation (e.g. [1,C,1,1] against [B,C]) is not expected
            rank = std::max(da.size(), db.size());
        }
        Dims od = bcast_dims(da, db);
        // dominant operand decides layout
        const Val *dom = nullptr;
        if (!a.is_const && prod(da) == prod(od) && da.size() == od.size()) dom = &a;
        else if (!b.is_const && prod(db) == prod(od) && db.size() == od.size()) dom = &b;
        Val out = dom ? new_act(od, strides_for_order(od, phys_order(*dom))) : new_act(od, row_major(od));
        auto operand = [&](const Val &v, const Dims &d, Ref &r, Dims &s, int64_t &bs) {
            if (v.is_const) {
                std::vector<float> data = v.is_int ? std::vector<float>(v.i.begin(), v.i.end()) : v.f;
                r = Ref{Space::CONSTS, add_const(data), 0};
                s = bcast_strides(&n, d, row_major(d), od);
                bs = 0;
            } else {
                r = ref_of(v);
                s = bcast_strides(&n, d, v.strides, od);
                bs = batch_stride(v);
            }
        };
        Ref ra_, rb_; Dims sa, sb; int64_t ba, bb;
        operand(a, da, ra_, sa, ba);
        operand(b, db, rb_, sb, bb);
        emit_elt(t + ":" + n.name, out, ra_, sa, ba, rb_, sb, bb, bin, ActSpec{});
        define(n.outputs[0], out);
    }

    // Where(cond, x, y) with at least one activation among the three: SelA(x, cond) + SelB(y, cond) -- each half is exactly x or +0
    // (never x * 0: infinities and NaNs of the unselected branch do not leak), the sum adds a +0; a branch that is the constant 0
    // drops its half.  The one difference to a true select: a selected -0.0 comes out as +0.0.
    void lower_where(const OnnxNode &n) {
        if (n.inputs.size() != 3) unsupported(n, "Where takes three inputs");
        const Val &x = get(n, 1), &y = get(n, 2);
        float c;
        const bool x0 = const_scalar(x, c) && c == 0.0f, y0 = const_scalar(y, c) && c == 0.0f;
        const std::string base = "where:" + n.outputs[0];
        if (y0 && !x0) return lower(synth_node("bn.SelA", n.name, {n.inputs[1], n.inputs[0]}, n.outputs[0]));
        if (x0 && !y0) return lower(synth_node("bn.SelB", n.name, {n.inputs[2], n.inputs[0]}, n.outputs[0]));
        lower(synth_node("bn.SelA", n.name + "/x", {n.inputs[1], n.inputs[0]}, base + "/x"));
        lower(synth_node("bn.SelB", n.name + "/y", {n.inputs[2], n.inputs[0]}, base + "/y"));
        lower(synth_node("Add", n.name, {base + "/x", base + "/y"}, n.outputs[0]));
    }

    void lower_batchnorm(const OnnxNode &n) {
        const Val &x = get(n, 0);
        const Val &sc = get(n, 1), &bi = get(n, 2), &mu = get(n, 3), &var = get(n, 4);
        if (!sc.is_const || !bi.is_const || !mu.is_const || !var.is_const) unsupported(n, "BatchNormalization parameters must be constants");
        float eps = n.attr_f("epsilon", 1e-5f);
        int64_t C = sc.numel();
        if (x.dims.empty() || x.dims[0] != C) unsupported(n, "channel count mismatch");
        std::vector<float> s(C), t(C);
        for (int64_t c = 0; c < C; c++) { s[c] = sc.f[c] / std::sqrt(var.f[c] + eps); t[c] = bi.f[c] - mu.f[c] * s[c]; }
        Dims cd(x.dims.size(), 1);
        cd[0] = C;
        Val mid = new_act(x.dims, strides_for_order(x.dims, phys_order(x)));
        Ref rs{Space::CONSTS, add_const(s), 0}, rt{Space::CONSTS, add_const(t), 0};
        Dims cs = bcast_strides(&n, cd, row_major(cd), x.dims);
        emit_elt("bn.mul:" + n.name, mid, ref_of(x), x.strides, batch_stride(x), rs, cs, 0, BIN_MUL, ActSpec{});
        Val out = new_act(x.dims, mid.strides);
        emit_elt("bn.add:" + n.name, out, ref_of(mid), mid.strides, batch_stride(mid), rt, cs, 0, BIN_ADD, ActSpec{});
        define(n.outputs[0], out);
    }

    // GlobalAveragePool -> Conv1x1 -> act -> Conv1x1 -> Sigmoid/HardSigmoid -> Mul(x, .) : squeeze-excite.
    // Emits the two SE launches and defines the gate; returns false if the pattern does not match.
    bool try_squeeze_excite(const OnnxNode &n, const Val &x) {
        if (x.dims.size() != 3 || x.space != Space::ARENA || x.gate_storage >= 0) return false;
        const int64_t C = x.dims[0], H = x.dims[1], W = x.dims[2];
        if (C % 4 || C > 4096 || !x.strides_equal(Dims{1, W * C, C}) || x.offset != 0) return false;
        if (plan_.storages[x.storage].elems % 4) return false;
        // spatial mean?
        if (n.op_type == "ReduceMean") {
            auto axes = n.attr_ints("axes");
            if (axes.empty() && has_input(n, 1)) axes = const_ints(n, get(n, 1));
            std::set<int64_t> ax;
            for (auto a : axes) ax.insert(a < 0 ? a + 4 : a);
            if (ax != std::set<int64_t>{2, 3} || n.attr_i("keepdims", 1) == 0) return false;
        } else if (n.op_type != "GlobalAveragePool") return false;
        int c1 = sole_consumer(n.outputs[0]);
        if (c1 < 0 || nodes_[c1].op_type != "Conv") return false;
        auto conv_1x1 = [&](const OnnxNode &cv, int64_t cin, int64_t &cout, std::vector<float> &w, std::vector<float> &b) {
            auto wi = vals_.find(cv.inputs[1]);
            if (wi == vals_.end() || !wi->second.is_const || wi->second.dims.size() != 4) return false;
            const Val &wv = wi->second;
            if (wv.dims[1] != cin || wv.dims[2] != 1 || wv.dims[3] != 1 || cv.attr_i("group", 1) != 1) return false;
            for (auto p : cv.attr_ints("pads")) if (p) return false;
            for (auto p : cv.attr_ints("strides")) if (p != 1) return false;
            cout = wv.dims[0];
            w = wv.f;
            b.clear();
            if (cv.inputs.size() > 2 && !cv.inputs[2].empty()) {
                auto bi = vals_.find(cv.inputs[2]);
                if (bi == vals_.end() || !bi->second.is_const) return false;
                b = bi->second.f;
            }
            return true;
        };
        int64_t Cr = 0, C2 = 0;
        std::vector<float> w1, b1, w2, b2;
        if (nodes_[c1].inputs[0] != n.outputs[0] || !conv_1x1(nodes_[c1], C, Cr, w1, b1) || Cr > 1024) return false;
        // dry-run the chain before absorbing anything
        std::vector<bool> saved = absorbed_;
        auto bail = [&]() { absorbed_ = saved; return false; };
        absorbed_[c1] = true;
        std::string cur = nodes_[c1].outputs[0];
        ActSpec act1;
        absorb_activation(cur, act1);
        int c2 = sole_consumer(cur);
        if (c2 < 0 || nodes_[c2].op_type != "Conv" || nodes_[c2].inputs[0] != cur || !conv_1x1(nodes_[c2], Cr, C2, w2, b2) || C2 != C) return bail();
        absorbed_[c2] = true;
        cur = nodes_[c2].outputs[0];
        int sg = sole_consumer(cur);
        ActSpec act2;
        if (sg < 0 || nodes_[sg].inputs[0] != cur || (nodes_[sg].op_type != "Sigmoid" && nodes_[sg].op_type != "HardSigmoid") || !unary_spec(nodes_[sg], act2)) return bail();
        absorbed_[sg] = true;
        const std::string gate_name = nodes_[sg].outputs[0];
        if (wanted_names_.count(gate_name)) return bail();
        // the gate must only feed Mul(x, gate)
        for (int k : live_consumers(gate_name)) {
            const OnnxNode &m = nodes_[k];
            if (m.op_type != "Mul") return bail();
            const std::string &other = m.inputs[0] == gate_name ? m.inputs[1] : m.inputs[0];
            if (other != n.inputs[0]) return bail();
        }
        // ---- emit
        const int64_t HW = H * W;
        int32_t splits = (int32_t)std::min<int64_t>(64, std::max<int64_t>(1, (HW + 31) / 32));
        // produced by the tiled depthwise kernel just before? then it emits the channel sums itself
        PlanOp *dwp = nullptr;
        if (!plan_.ops.empty()) {
            PlanOp &last = plan_.ops.back();
            if (last.kind == OpKind::DWCONV && last.dw.tiled && last.out.space == Space::ARENA && last.out.id == x.storage && last.out.offset == 0)
                dwp = &last;
        }
        PlanOp *mbp = nullptr;
        if (!dwp && !plan_.ops.empty()) {
            PlanOp &last = plan_.ops.back();
            if (last.kind == OpKind::MBCONV && last.out.space == Space::ARENA && last.out.id == x.storage && last.out.offset == 0) mbp = &last;
        }
        if (dwp) splits = dwp->dw.nblk;
        if (mbp) splits = mbp->mb.tiles_x * mbp->mb.tiles_y;
        Val partial = new_act(Dims{(int64_t)splits, C}, Dims{C, 1});
        if (mbp) {
            mbp->mb.has_gap = 1;
            mbp->mb.gap_bs = plan_.storages[partial.storage].elems;
            mbp->b = ref_of(partial);
            touch(mbp->b, (int)plan_.ops.size() - 1);
        } else if (dwp) {
            dwp->dw.has_gap = 1;
            dwp->dw.gap_bs = plan_.storages[partial.storage].elems;
            dwp->b = ref_of(partial);
            touch(dwp->b, (int)plan_.ops.size() - 1);
        } else {
            PlanOp op;
            op.kind = OpKind::GAP;
            op.name = "se.squeeze:" + n.name;
            op.out = ref_of(partial);
            op.a = ref_of(x);
            op.gap.HW = HW; op.gap.C = (int32_t)C; op.gap.splits = splits;
            op.gap.in_bs = batch_stride(x); op.gap.out_bs = plan_.storages[partial.storage].elems;
            op.bytes = 4.0 * (double)(HW * C + splits * C);
            push_op(std::move(op));
        }
        Val gate = new_act(Dims{C, 1, 1}, Dims{1, 0, 0});
        plan_.storages[gate.storage].elems = (C + 3) / 4 * 4;
        std::vector<float> w2t(w2.size());  // [C][Cr] -> [Cr][C]: coalesced reads in the excite product
        for (int64_t c = 0; c < C; c++)
            for (int64_t j = 0; j < Cr; j++) w2t[j * C + c] = w2[c * Cr + j];
        SeFcDesc sd{};
        sd.C = (int32_t)C; sd.Cr = (int32_t)Cr; sd.splits = splits; sd.inv_hw = 1.0f / (float)HW;
        sd.act1 = act1.act; sd.p0_1 = act1.p0; sd.p1_1 = act1.p1;
        sd.act2 = act2.act; sd.p0_2 = act2.p0; sd.p1_2 = act2.p1;
        sd.in_bs = plan_.storages[partial.storage].elems; sd.out_bs = plan_.storages[gate.storage].elems;
        // Rule H: the launch that produced the squeeze sums finishes the squeeze-excite itself -- its last block per sample
        // (ticket counter, write-through partial sums: no fence, kernels.h SeTail) -- when that launch is a fused MBConv
        // or a whole-map depthwise conv.  One launch less per block (16 in the v2.4 plan).  BN_SEFUSE=0 disables.
        PlanOp *host = nullptr;
        {
            // Opt-in (BN_SEFUSE = dw | mb | 1): correct, but slower than the separate excite launch on every block of the
            // v2.4 plan -- the last block streams both excite matrices through ONE compute unit (442 KB at C = 1152:
            // +26 us per whole-map depthwise launch against 8-10 us for the multi-block excite kernel), and every block of
            // a fused MBConv launch pays a store drain + returning atomic (+13 us per launch); DESIGN.md section 4.12.
            const std::string mode = getenv("BN_SEFUSE") ? getenv("BN_SEFUSE") : "0";  // 0 | dw | mb | 1 (both)
            if (mbp && !mbp->mb.whole_map && !mbp->mb.row_mode && (mode == "1" || mode == "mb")) host = mbp;  // the row-streaming form has no block that could be "last"
            else if (dwp && dwp->dw.tiled == 2 && (mode == "1" || mode == "dw")) host = dwp;
        }
        if (host) {
            Val cnt = new_act(Dims{1}, Dims{1});
            plan_.storages[cnt.storage].pinned = true;      // never recycled ...
            plan_.storages[cnt.storage].persistent = true;  // ... and never shared with an earlier tensor either: the word must stay zero between launches
            const int hidx = (int)plan_.ops.size() - 1;
            host->se_fused = 1;
            host->se = sd;
            host->x[0] = Ref{Space::CONSTS, add_const(w1), 0};
            if (!b1.empty()) host->x[1] = Ref{Space::CONSTS, add_const(b1), 0};
            host->x[2] = Ref{Space::CONSTS, add_const(w2t), 0};
            if (!b2.empty()) host->x[3] = Ref{Space::CONSTS, add_const(b2), 0};
            host->res = ref_of(gate);
            host->scale = ref_of(cnt);
            touch(host->res, hidx);
            touch(host->scale, hidx);
            host->name += "+se";
            host->macs += 2.0 * (double)C * Cr;
            host->weight_bytes += 4.0 * (w1.size() + w2.size() + b1.size() + b2.size());
            host->bytes += 4.0 * (double)(splits * C + C);
        } else {
            PlanOp op;
            Val hidden = new_act(Dims{Cr}, Dims{1});
            op.kind = OpKind::SEFC;
            op.name = "se.excite:" + n.name;
            op.out = ref_of(gate);
            op.a = ref_of(partial);
            op.b = ref_of(hidden);  // scratch between the two excite launches
            op.w = Ref{Space::CONSTS, add_const(w1), 0};
            if (!b1.empty()) op.bias = Ref{Space::CONSTS, add_const(b1), 0};
            op.w2 = Ref{Space::CONSTS, add_const(w2t), 0};
            if (!b2.empty()) op.bias2 = Ref{Space::CONSTS, add_const(b2), 0};
            op.se = sd;
            op.macs = 2.0 * (double)C * Cr;
            op.weight_bytes = 4.0 * (w1.size() + w2.size() + b1.size() + b2.size());
            op.bytes = 4.0 * (double)(splits * C + C);
            push_op(std::move(op));
        }
        gate.is_gate = true;
        define(gate_name, gate);
        return true;
    }

    void lower_reduce(const OnnxNode &n) {
        const Val &v = get(n, 0);
        const std::string &t = n.op_type;
        if (v.is_const) unsupported(n, "reduction of a constant");
        if (try_squeeze_excite(n, v)) return;
        // max(x - s) == max(x) - s exactly when s is one number per reduced range (rounding is monotone), same for min:
        // the reduction reads x itself and the shifted signal is no longer needed for it -- in the min-max normalisation
        // of the v2.4 front end the Sub then has a single consumer left and fuses into the scaling chain (one pass less).
        if ((t == "ReduceMax" || t == "ReduceMin") && !(getenv("BN_REDUCE_SHIFT") && std::string(getenv("BN_REDUCE_SHIFT")) == "0")) {
            int prod = -1;
            for (size_t k = 0; k < nodes_.size() && prod < 0; k++)
                if (live_[k] && !absorbed_[k] && nodes_[k].op_type == "Sub" && nodes_[k].outputs.size() == 1 && nodes_[k].outputs[0] == n.inputs[0]) prod = (int)k;
            if (prod >= 0 && nodes_[prod].inputs.size() == 2 && n.attr_i("keepdims", 1) != 0) {
                const OnnxNode &sub = nodes_[prod];
                auto xi = vals_.find(sub.inputs[0]), si = vals_.find(sub.inputs[1]);
                if (xi != vals_.end() && si != vals_.end() && !xi->second.is_const && xi->second.dims == v.dims && live_consumers(n.inputs[0]).size() > 1) {
                    // s must be constant along every reduced axis: here, one element per sample
                    if (si->second.numel() == 1) {
                        OnnxNode r2 = n;
                        r2.name = n.name + "/unshifted";
                        r2.inputs[0] = sub.inputs[0];
                        r2.outputs = {n.outputs[0] + "/unshifted"};
                        lower_reduce(r2);
                        OnnxNode s2;
                        s2.op_type = "Sub";
                        s2.name = n.name + "/shift";
                        s2.inputs = {r2.outputs[0], sub.inputs[1]};
                        s2.outputs = {n.outputs[0]};
                        lower_binary(s2);
                        return;
                    }
                }
            }
        }
        int op;
        std::vector<int64_t> axes;
        bool keep = n.attr_i("keepdims", 1) != 0;
        int64_t r = (int64_t)v.dims.size() + 1;
        if (t == "GlobalAveragePool" || t == "GlobalMaxPool") {
            op = t == "GlobalAveragePool" ? RED_MEAN : RED_MAX;
            for (int64_t k = 2; k < r; k++) axes.push_back(k);
            keep = true;
        } else {
            if (t == "ReduceMean") op = RED_MEAN; else if (t == "ReduceSum") op = RED_SUM; else if (t == "ReduceMax") op = RED_MAX;
            else if (t == "ReduceMin") op = RED_MIN; else if (t == "ReduceProd") op = RED_PROD; else if (t == "ReduceL2") op = RED_L2;
            else if (t == "ReduceSumSquare") op = RED_SUMSQ; else unsupported(n, "unsupported reduction");
            axes = n.attr_ints("axes");
            if (axes.empty() && has_input(n, 1)) axes = const_ints(n, get(n, 1));
            if (axes.empty()) {
                if (n.attr_i("noop_with_empty_axes", 0)) { define(n.outputs[0], v); return; }
                unsupported(n, "reduction over all axes would include the batch");
            }
        }
        std::set<int> red;
        for (auto a : axes) { if (a < 0) a += r; if (a == 0) unsupported(n, "reduction over the batch dimension"); red.insert((int)a - 1); }
        Dims od;
        for (size_t k = 0; k < v.dims.size(); k++) { if (red.count((int)k)) { if (keep) od.push_back(1); } else od.push_back(v.dims[k]); }
        // kept dims in input physical order
        std::vector<int> ord = phys_order(v);
        PlanOp op_;
        op_.kind = OpKind::REDUCE;
        op_.name = t + ":" + n.name;
        ReduceDesc &d = op_.red;
        d.op = op;
        Dims kept_dims_phys;
        std::vector<int> kept_axes_phys;
        d.kept = d.red = 1;
        d.inner_kept = 0;
        for (int ax : ord) {
            if (v.dims[ax] == 1) continue;
            if (red.count(ax)) {
                if (d.nr >= 3) unsupported(n, "more than 3 reduced dimensions");
                d.rsize[d.nr] = v.dims[ax]; d.rin[d.nr] = v.strides[ax]; d.nr++; d.red *= v.dims[ax];
            } else {
                if (d.nk >= 4) unsupported(n, "more than 4 kept dimensions");
                d.ksize[d.nk] = v.dims[ax]; d.kin[d.nk] = v.strides[ax]; d.nk++; d.kept *= v.dims[ax];
                kept_axes_phys.push_back(ax);
                if (std::abs(v.strides[ax]) == 1) d.inner_kept = 1;
            }
        }
        // output: contiguous over kept dims in input physical order
        Dims ostr_phys(kept_axes_phys.size());
        { int64_t acc = 1; for (int k = (int)kept_axes_phys.size() - 1; k >= 0; k--) { ostr_phys[k] = acc; acc *= v.dims[kept_axes_phys[k]]; } }
        for (int k = 0; k < d.nk; k++) d.kout[k] = ostr_phys[k];
        // logical output strides
        Dims ostr(od.size(), 0);
        {
            size_t oi = 0;
            for (size_t k = 0; k < v.dims.size(); k++) {
                bool is_red = red.count((int)k) != 0;
                if (is_red && !keep) continue;
                if (!is_red && v.dims[k] != 1) {
                    for (size_t q = 0; q < kept_axes_phys.size(); q++) if (kept_axes_phys[q] == (int)k) ostr[oi] = ostr_phys[q];
                }
                oi++;
            }
        }
        Val out = new_act(od, ostr);
        plan_.storages[out.storage].elems = std::max<int64_t>(d.kept, 1);
        d.bi = batch_stride(v);
        d.bo = plan_.storages[out.storage].elems;
        op_.out = ref_of(out);
        op_.a = ref_of(v);
        op_.bytes = 4.0 * (double)(d.kept * d.red + d.kept);
        // A whole-segment min / max (144 000 samples -> one number) as ONE block per sample leaves the chip idle
        // (32 blocks at batch 32, 26 us).  Min and max do not depend on the order of evaluation, so the range is cut
        // into c chunks reduced by c blocks per sample and a second tiny launch reduces the c partials: bit-identical
        // result, ~4x less time.  Sums keep their single fixed-order pass.  BN_REDUCE_SPLIT=0 disables.
        if ((op == RED_MAX || op == RED_MIN) && d.nk == 0 && d.nr == 1 && d.rin[0] == 1 && d.red >= 32768 &&
            !(getenv("BN_REDUCE_SPLIT") && std::string(getenv("BN_REDUCE_SPLIT")) == "0")) {
            int64_t c = 0;
            for (int64_t q = 64; q >= 4 && !c; q--)
                if (d.red % q == 0 && (d.red / q) % 4 == 0 && d.red / q >= 8192) c = q;
            if (c) {
                Val part = new_act(Dims{c}, Dims{1});
                PlanOp s1 = op_, s2 = op_;
                s1.name += "/chunks";
                s1.out = ref_of(part);
                s1.red.nk = 1; s1.red.ksize[0] = c; s1.red.kin[0] = d.red / c; s1.red.kout[0] = 1; s1.red.kept = c;
                s1.red.rsize[0] = d.red / c; s1.red.red = d.red / c;
                s1.red.bo = plan_.storages[part.storage].elems;
                s1.bytes = 4.0 * (double)(d.red + c);
                s2.a = ref_of(part);
                s2.red.rsize[0] = c; s2.red.red = c;
                s2.red.bi = plan_.storages[part.storage].elems;
                s2.bytes = 4.0 * (double)(c + 1);
                push_op(std::move(s1));
                push_op(std::move(s2));
                define(n.outputs[0], out);
                return;
            }
        }
        push_op(std::move(op_));
        define(n.outputs[0], out);
    }

    // Ensure `v` ([C,H,W] or [C,L]) is channels-last contiguous; returns the (possibly copied) value.
    Val to_channels_last(const Val &v, const std::string &why) {
        Dims want(v.dims.size());
        if (v.dims.size() == 3) { want = {1, v.dims[2] * v.dims[0], v.dims[0]}; }
        else { want = {1, v.dims[0]}; }
        if (v.space != Space::INPUT && v.strides_equal(want)) return v;
        if (v.space == Space::INPUT && v.strides_equal(want)) return v;
        return materialize(v, want, why);
    }

    void lower_conv(const OnnxNode &n) {
        Val x = get(n, 0);
        int32_t gate_storage = x.gate_storage;
        x.gate_storage = -1;
        const Val &w = get(n, 1);
        const Val *bptr = opt(n, 2);
        if (x.is_const || !w.is_const || (bptr && !bptr->is_const)) unsupported(n, "Conv needs an activation input and constant weights");
        size_t sp = x.dims.size() - 1;  // spatial rank
        if (sp != 1 && sp != 2) unsupported(n, "only 1-D and 2-D convolutions are supported");
        if (w.dims.size() != sp + 2) unsupported(n, "weight rank mismatch");
        int64_t groups = n.attr_i("group", 1);
        int64_t Cin = x.dims[0], Cout = w.dims[0], cpg = w.dims[1];
        if (cpg * groups != Cin || Cout % groups) unsupported(n, "channel/group mismatch");
        auto get2 = [&](const char *key, int64_t dflt) {
            auto v = n.attr_ints(key);
            if (v.empty()) v.assign(sp, dflt);
            if (v.size() != sp) unsupported(n, std::string(key) + " does not match the spatial rank");
            for (auto e : v) if (e <= 0) unsupported(n, std::string(key) + " must be positive");
            if (sp == 1) v.insert(v.begin(), dflt == 0 ? 0 : 1);
            return v;
        };
        int64_t H = sp == 2 ? x.dims[1] : 1, W = sp == 2 ? x.dims[2] : x.dims[1];
        int64_t kh = sp == 2 ? w.dims[2] : 1, kw = sp == 2 ? w.dims[3] : w.dims[2];
        auto strides = get2("strides", 1), dil = get2("dilations", 1);
        std::vector<int64_t> pads = n.attr_ints("pads");
        if (pads.empty()) pads.assign(2 * sp, 0);
        if (pads.size() != 2 * sp) unsupported(n, "pads does not match the spatial rank");
        for (auto e : pads) if (e < 0) unsupported(n, "negative pads");
        int64_t pt, pl, pb, pr;
        if (sp == 2) { pt = pads[0]; pl = pads[1]; pb = pads[2]; pr = pads[3]; }
        else { pt = pb = 0; pl = pads[0]; pr = pads[1]; }
        std::string auto_pad = n.attr_s("auto_pad", "NOTSET");
        auto out_dim = [&](int64_t in, int64_t k, int64_t s, int64_t d, int64_t &p0, int64_t &p1) {
            int64_t ke = (k - 1) * d + 1;
            if (auto_pad == "SAME_UPPER" || auto_pad == "SAME_LOWER") {
                int64_t o = (in + s - 1) / s;
                int64_t tot = std::max<int64_t>((o - 1) * s + ke - in, 0);
                p0 = auto_pad == "SAME_UPPER" ? tot / 2 : tot - tot / 2;
                p1 = tot - p0;
                return o;
            }
            if (auto_pad == "VALID") p0 = p1 = 0;
            return (in + p0 + p1 - ke) / s + 1;
        };
        int64_t OH = out_dim(H, kh, strides[0], dil[0], pt, pb), OW = out_dim(W, kw, strides[1], dil[1], pl, pr);
        if (OH <= 0 || OW <= 0) unsupported(n, "empty convolution output");

        // ---- epilogue fusion: BatchNorm, activation, residual ----
        std::vector<float> wf = w.f;
        std::vector<float> bias(Cout, 0.0f);
        bool has_bias = bptr != nullptr;
        if (bptr) bias = bptr->f;
        std::string cur = n.outputs[0];
        int64_t per_out = wf.size() / Cout;
        {
            int c = sole_consumer(cur);
            if (c >= 0 && nodes_[c].op_type == "BatchNormalization" && nodes_[c].inputs[0] == cur) {
                const OnnxNode &bn_ = nodes_[c];
                const Val &sc = get(bn_, 1), &bi = get(bn_, 2), &mu = get(bn_, 3), &var = get(bn_, 4);
                if (sc.is_const && bi.is_const && mu.is_const && var.is_const && sc.numel() == Cout) {
                    float eps = bn_.attr_f("epsilon", 1e-5f);
                    for (int64_t o = 0; o < Cout; o++) {
                        float s = sc.f[o] / std::sqrt(var.f[o] + eps);
                        for (int64_t k = 0; k < per_out; k++) wf[o * per_out + k] *= s;
                        bias[o] = (bias[o] - mu.f[o]) * s + bi.f[o];
                    }
                    has_bias = true;
                    absorbed_[c] = true;
                    cur = bn_.outputs[0];
                }
            }
        }
        ActSpec act;
        absorb_activation(cur, act);
        // residual: Add(cur, other) with `other` an available activation of identical shape
        Val res;
        bool has_res = false;
        Dims out_dims = sp == 2 ? Dims{Cout, OH, OW} : Dims{Cout, OW};
        Dims out_strides = sp == 2 ? Dims{1, OW * Cout, Cout} : Dims{1, Cout};
        {
            int c = sole_consumer(cur);
            if (c >= 0 && nodes_[c].op_type == "Add") {
                const OnnxNode &ad = nodes_[c];
                const std::string &other = ad.inputs[0] == cur ? ad.inputs[1] : ad.inputs[0];
                auto it = vals_.find(other);
                if (it != vals_.end() && !it->second.is_const && it->second.dims == out_dims && it->second.space == Space::ARENA &&
                    it->second.strides_equal(out_strides)) {
                    res = it->second;
                    has_res = true;
                    absorbed_[c] = true;
                    cur = ad.outputs[0];
                }
            }
        }

        Val out = new_act(out_dims, out_strides);
        PlanOp op;
        op.name = "Conv:" + n.name;
        op.out = ref_of(out);
        op.macs = (double)OH * OW * Cout * kh * kw * cpg;
        op.weight_bytes = 4.0 * (wf.size() + (has_bias ? Cout : 0));
        op.bytes = 4.0 * ((double)Cin * H * W + (double)Cout * OH * OW * (has_res ? 2 : 1));
        if (has_bias) op.bias = Ref{Space::CONSTS, add_const(bias), 0};
        if (has_res) op.res = ref_of(res);

        bool unit_dil = dil[0] == 1 && dil[1] == 1;
        // a padded 1-D convolution with a long filter (STFT / learned front ends) is worth a padded copy
        // of its input: it then runs as the overlapping-rows GEMM on the matrix cores instead of the
        // direct kernel (Perch-style front end: 14 ms -> GEMM time at batch 128)
        if (groups == 1 && kh == 1 && H == 1 && unit_dil && pt == 0 && pb == 0 && (pl || pr) && kw * Cin >= 64) {
            if (gate_storage >= 0) { Val gx = x; gx.gate_storage = gate_storage; x = apply_gate(gx, n.name); gate_storage = -1; }
            Dims lo(x.dims.size(), 0), hi(x.dims.size(), 0);
            lo.back() = pl; hi.back() = pr;
            x = pad_copy(x, lo, hi, 0.0f, n.name);
            W += pl + pr;
            pl = pr = 0;
        }
        bool no_pad = pt == 0 && pl == 0 && pb == 0 && pr == 0;
        const bool gemm_path = groups == 1 && kh == 1 && unit_dil && no_pad && (H == 1 || (kw == 1 && strides[0] == 1 && strides[1] == 1));
        if (gate_storage >= 0 && !(gemm_path && kw == 1)) {  // only the 1x1 GEMM folds the SE gate
            Val gx = x;
            gx.gate_storage = gate_storage;
            x = apply_gate(gx, n.name);
            gate_storage = -1;
        }
        if (gemm_path) {
            // GEMM: 1x1 conv (rows = H*W) or 1-D conv as overlapping rows (rows = OW, K = kw*Cin)
            x = to_channels_last(x, n.name);
            if (gate_storage >= 0) {
                op.scale = Ref{Space::ARENA, gate_storage, 0};
                op.gemm.has_scale = 1;
                op.gemm.s_bs = plan_.storages[gate_storage].elems;
            }
            op.kind = OpKind::GEMM;
            op.mfma = true;
            GemmDesc &g = op.gemm;
            g.K = (int32_t)(kw * Cin);
            g.N = (int32_t)Cout;
            if (H == 1) { g.rows = OW; g.lda = strides[1] * Cin; }
            else { g.rows = H * W; g.lda = Cin; }
            g.a_bs = batch_stride(x);
            g.ldc = Cout; g.c_bs = plan_.storages[out.storage].elems;
            g.act = act.act; g.p0 = act.p0; g.p1 = act.p1;
            g.has_bias = has_bias; g.has_res = has_res;
            if (has_res) { g.ldr = Cout; g.r_bs = batch_stride(res); }
            // weights [Cout][Cin][kw] -> [Cout][kw][Cin]  (K index = k*Cin + c)
            std::vector<float> wp(wf.size());
            for (int64_t o = 0; o < Cout; o++)
                for (int64_t c = 0; c < Cin; c++)
                    for (int64_t k = 0; k < kw; k++) wp[(o * kw + k) * Cin + c] = wf[(o * Cin + c) * kw + k];
            op.a = ref_of(x);
            if (H == 1 && Cin == 1 && !has_res && gate_storage < 0 && fold_framing_conv(n, op, wf, Cout, kw, OW, has_bias)) {
                define(cur, out);
                return;
            }
            op.w = Ref{Space::CONSTS, add_const(wp), 0};
        } else if (groups == Cin && cpg == 1 && Cout == Cin) {
            x = to_channels_last(x, n.name);
            if (has_res) {  // depthwise kernel has no residual input: undo that fusion
                unsupported(n, "depthwise convolution followed by a fused residual is not expected");
            }
            op.kind = OpKind::DWCONV;
            DwDesc &d = op.dw;
            d.H = (int32_t)H; d.W = (int32_t)W; d.C = (int32_t)Cin; d.OH = (int32_t)OH; d.OW = (int32_t)OW;
            d.kh = (int32_t)kh; d.kw = (int32_t)kw; d.sh = (int32_t)strides[0]; d.sw = (int32_t)strides[1];
            d.pt = (int32_t)pt; d.pl = (int32_t)pl; d.dh = (int32_t)dil[0]; d.dw = (int32_t)dil[1];
            d.act = act.act; d.p0 = act.p0; d.p1 = act.p1; d.has_bias = has_bias;
            d.in_bs = batch_stride(x); d.out_bs = plan_.storages[out.storage].elems;
            if (Cin % 4 == 0 && Cin / 4 <= 1024 && (kw == 3 || kw == 5) && (strides[1] == 1 || strides[1] == 2) && unit_dil && x.offset % 4 == 0 &&
                d.in_bs % 4 == 0 && d.out_bs % 4 == 0) {
                d.tiled = 1;
                d.tw = 4;
                // pixel tiles per block: up to 1024 lanes, but keep >= 8 blocks per sample; fewer, larger
                // blocks also mean fewer squeeze partials for the SE excite kernel to add up
                const int64_t tiles = OH * ((OW + d.tw - 1) / d.tw);
                d.rpb = (int32_t)std::max<int64_t>(1, std::min<int64_t>(1024 / (Cin / 4), (tiles + 7) / 8));
                d.nblk = (int32_t)((tiles + d.rpb - 1) / d.rpb);
                // small feature maps: one block stages the whole map of 32 channels in LDS (one round of
                // coalesced loads instead of a load-use chain per kernel row) and emits the complete
                // squeeze sums, so the excite kernel adds nothing up
                const bool map_off = getenv("BN_DWMAP") && std::string(getenv("BN_DWMAP")) == "0";
                if (!map_off && kh == kw && strides[0] == strides[1] && H * W <= 768) {
                    d.tiled = 2;
                    d.mapt = env_int("BN_DWMAPT", 1) != 0 ? 1 : 0;
                    d.nblk = 1;
                }
            }
            std::vector<float> wp(wf.size());  // [C][1][kh][kw] -> [kh][kw][C]
            for (int64_t c = 0; c < Cin; c++)
                for (int64_t k = 0; k < kh * kw; k++) wp[k * Cin + c] = wf[c * kh * kw + k];
            op.w = Ref{Space::CONSTS, add_const(wp), 0};
            op.a = ref_of(x);
            // expand 1x1 conv immediately before, consumed only here?  Then both run in one launch with the
            // expanded tensor kept in LDS (mbconv_expand_dw_kernel): the 6x-expanded activation never
            // touches HBM.  Measured on MI355X (batch 32, per pair): 48x256x16->96 s2: 101 -> 79 us,
            // 24x128x24->144 s1: 80 -> 54 us, 5x5 s2: 60 -> 52 us, 12x64x40->240 5x5 s1: 54 -> 50 us; on
            // the 6x32 maps (2 tiles per sample) and for K >= 80 the separate launches win, so the rule
            // below keeps those unfused.  The rule only looks at per-sample shapes (batch invariance).
            // BN_MBFUSE=0 disables, BN_MBFUSE=force fuses every eligible pair (tests).
            const char *mbenv = getenv("BN_MBFUSE");
            const bool mb_off = mbenv && std::string(mbenv) == "0";
            // Stem variant: the producer is a dense k1 x k1 convolution with few input channels (k1*k1*Cin1 <= 48,
            // e.g. the 3x3 stride-2 conv over the 2-channel spectrogram image).  Its output tile is rebuilt from im2col
            // rows staged in LDS and never written to HBM either (same kernel, different staging).  BN_STEMFUSE=0 disables.
            if (d.tiled && kh == kw && strides[0] == strides[1] && !mb_off && !plan_.ops.empty() &&
                !(getenv("BN_STEMFUSE") && std::string(getenv("BN_STEMFUSE")) == "0")) {
                PlanOp &pe = plan_.ops.back();
                const ConvDesc &cd = pe.conv;
                const int a1 = cd.act;
                const bool act_zero = a1 == ACT_NONE || a1 == ACT_RELU || (a1 == ACT_CLIP && cd.p0 <= 0.f && cd.p1 >= 0.f) || a1 == ACT_SILU ||
                                      a1 == ACT_HSWISH || a1 == ACT_LEAKY || a1 == ACT_TANH;
                const bool stem = pe.kind == OpKind::CONV && pe.out.space == Space::ARENA && pe.out.id == x.storage && pe.out.offset == 0 && x.offset == 0 &&
                                  cd.groups == 1 && cd.kh == cd.kw && cd.sh == cd.sw && cd.dh == 1 && cd.dw == 1 && !cd.has_res && cd.Cin <= 4 &&
                                  cd.kh * cd.kw * cd.Cin <= 48 && cd.OH == H && cd.OW == W && cd.Cout == Cin && act_zero &&
                                  (H * W >= 3072 || (mbenv && std::string(mbenv) == "force"));
                MbDesc probe{};
                probe.k = (int32_t)kw; probe.s = (int32_t)strides[1]; probe.Cin = stem ? cd.kh * cd.kw * cd.Cin : 4; probe.C = (int32_t)Cin;
                if (stem && mbconv_lds_bytes(probe) <= 150 * 1024 && sole_consumer(n.inputs[0]) == cur_) {
                    PlanOp mb;
                    mb.kind = OpKind::MBCONV;
                    mb.name = "stem:" + pe.name.substr(pe.name.find(':') + 1) + "+" + n.name;
                    mb.out = op.out;
                    mb.a = pe.a;
                    // first-conv filters [kh][kw][Cin1][Cout] (as lowered for the direct kernel) -> [Cout][KW]: im2col column
                    // order (ky, kx, c), zero padded to 8-wide K groups
                    const int64_t K1 = (int64_t)cd.kh * cd.kw * cd.Cin, KW = (K1 + 7) / 8 * 8;
                    {
                        const std::vector<float> &w0 = plan_.consts[pe.w.id];
                        std::vector<float> wpk((size_t)(Cin * KW), 0.0f);
                        for (int64_t nn = 0; nn < Cin; nn++)
                            for (int64_t k = 0; k < K1; k++) wpk[nn * KW + k] = w0[pe.w.offset + k * Cin + nn];
                        mb.w = Ref{Space::CONSTS, add_const(wpk), 0};
                    }
                    mb.bias = pe.bias;
                    mb.w2 = op.w; mb.bias2 = op.bias;
                    MbDesc &m = mb.mb;
                    m.H = (int32_t)H; m.W = (int32_t)W; m.Cin = (int32_t)K1; m.C = (int32_t)Cin; m.OH = (int32_t)OH; m.OW = (int32_t)OW;
                    m.k = (int32_t)kw; m.s = (int32_t)strides[1]; m.pt = (int32_t)pt; m.pl = (int32_t)pl;
                    m.act1 = cd.act; m.p0_1 = cd.p0; m.p1_1 = cd.p1; m.has_bias1 = cd.has_bias;
                    m.act2 = act.act; m.p0_2 = act.p0; m.p1_2 = act.p1; m.has_bias2 = has_bias;
                    m.in_bs = cd.in_bs; m.out_bs = d.out_bs;
                    m.k1 = cd.kh; m.s1 = cd.sh; m.pt1 = cd.pt; m.pl1 = cd.pl; m.H1 = cd.H; m.W1 = cd.W; m.Cin1 = cd.Cin;
                    const int toh = m.s == 1 ? 8 : 4, tow = m.s == 1 ? 16 : 8;
                    m.tiles_x = (int32_t)((OW + tow - 1) / tow); m.tiles_y = (int32_t)((OH + toh - 1) / toh);
                    double halo = (double)((toh - 1) * m.s + m.k) * ((tow - 1) * m.s + m.k) * m.tiles_x * m.tiles_y;
                    row_streaming(m, halo);
                    mb.macs = op.macs;
                    mb.weight_bytes = op.weight_bytes + pe.weight_bytes;
                    mb.bytes = 4.0 * ((double)cd.H * cd.W * cd.Cin + (double)OH * OW * Cin);
                    mb.mfma = false;
                    mb.macs_mfma_extra = halo * (double)K1 * Cin;
                    mb.macs_recompute = std::max(0.0, mb.macs_mfma_extra - (double)H * W * (double)K1 * Cin);
                    plan_.ops.pop_back();
                    push_op(std::move(mb));
                    define(cur, out);
                    return;
                }
            }
            if (d.tiled && kh == kw && strides[0] == strides[1] && !mb_off && !plan_.ops.empty()) {
                PlanOp &pe = plan_.ops.back();
                const int maxk = std::min(48, getenv("BN_MBFUSE_MAXK") ? atoi(getenv("BN_MBFUSE_MAXK")) : 48);  // kernel: <= 6 K groups in registers
                // the kernel relies on act(0) == 0 for the expand activation (pixels outside the image)
                const int a1 = pe.gemm.act;
                const bool act_zero = a1 == ACT_NONE || a1 == ACT_RELU || (a1 == ACT_CLIP && pe.gemm.p0 <= 0.f && pe.gemm.p1 >= 0.f) ||
                                      a1 == ACT_SILU || a1 == ACT_HSWISH || a1 == ACT_LEAKY || a1 == ACT_TANH;
                const double maxhalo = getenv("BN_MBFUSE_HALO") ? atof(getenv("BN_MBFUSE_HALO")) : 3.0;
                // small feature maps can take the whole-map kernel (one block = 32 mid channels x the whole map:
                // no halo, any K up to 256, complete squeeze sums).  Opt-in (BN_MBMAP=1): measured on MI355X at
                // batch 32 it saves 10 launches and ~1% of single-stream latency, but its per-block fixed costs
                // (filter slab + A rows per 32-channel chunk, half the waves idle on 3x16 maps) make the
                // marginal cost per extra batch ~30% higher than GEMM + whole-map depthwise, and the three-
                // context throughput drops from 34.2k to 33.1k seg/s.
                const bool map_off = !(getenv("BN_MBMAP") && std::string(getenv("BN_MBMAP")) == "1");
                const int64_t map_maxhw = getenv("BN_MBMAP_MAXHW") ? atoll(getenv("BN_MBMAP_MAXHW")) : 512;
                const bool producer0 = pe.kind == OpKind::GEMM && pe.out.space == Space::ARENA && pe.out.id == x.storage && pe.out.offset == 0 &&
                                       x.offset == 0 && !pe.gemm.has_scale && !pe.gemm.has_res && pe.gemm.rows == H * W && pe.gemm.N == Cin &&
                                       pe.gemm.lda == pe.gemm.K && pe.gemm.K % 4 == 0 && pe.a.offset % 4 == 0 && !pe.gemm.fold && !pe.gemm.npost &&
                                       (pe.a.space != Space::ARENA || plan_.storages[pe.a.id].elems % 4 == 0);
                // Whole-map form with the INPUT resident in LDS (mbmap.hip, round 3): one block = one sample's map x a group of
                // mid channels, input fetched once per block by LDS-DMA, no staging on the vector ALU.  Default wherever a
                // configuration fits (192- and 48-pixel maps); BN_MBMAP2=0 disables.
                int map2 = 0, map_b3 = 0, map_ws = 0;
                MbmapShape mshape;
                if (producer0) {
                    MbDesc q{};
                    q.H = (int32_t)H; q.W = (int32_t)W; q.Cin = pe.gemm.K; q.C = (int32_t)Cin; q.OH = (int32_t)OH; q.OW = (int32_t)OW;
                    q.k = (int32_t)kw; q.s = (int32_t)strides[1]; q.pt = (int32_t)pt; q.pl = (int32_t)pl;
                    q.act1 = pe.gemm.act; q.act2 = act.act; q.in_bs = pe.gemm.a_bs;
                    mshape = mbmap_shape(q);
                    map2 = mshape.cfg;
                    map_b3 = mbmap_b3_steps(q, mshape);
                    map_ws = mbmap_ws_steps(q, mshape);
                }
                const bool whole_map = map2 != 0 || (!map_off && H * W <= map_maxhw);
                const bool producer = producer0 && (map2 != 0 || pe.gemm.K <= (whole_map ? 256 : maxk));
                // halo recompute factor of the expand conv: staged halo pixels / image pixels
                const int toh0 = strides[1] == 1 ? 8 : 4, tow0 = strides[1] == 1 ? 16 : 8;
                const int64_t hp = ((toh0 - 1) * strides[1] + kw) * ((tow0 - 1) * strides[1] + kw);
                const double halo_factor = (double)((hp + 31) / 32 * 32) * ((OW + tow0 - 1) / tow0) * ((OH + toh0 - 1) / toh0) / (double)(H * W);
                const bool force = mbenv && std::string(mbenv) == "force";  // tests: small feature maps too
                // (round 3: stride-2 blocks on 768-pixel maps too -- with the row-streaming kernel the 12x64x40->240 s2 pair
                // runs in 17.5 us instead of 16.5 + 16.4, marginal cost per extra batch 8.6 against 24 us)
                const bool big_enough = H * W >= 768;
                MbDesc probe{};
                probe.k = (int32_t)kw; probe.s = (int32_t)strides[1]; probe.Cin = producer ? pe.gemm.K : 4; probe.C = (int32_t)Cin;
                probe.H = (int32_t)H; probe.W = (int32_t)W; probe.whole_map = whole_map ? 1 : 0;
                const bool fits = map2 != 0 || mbconv_lds_bytes(probe) <= 150 * 1024;
                const bool tiled_ok = act_zero && ((halo_factor <= maxhalo && big_enough) || force);
                if (producer && fits && (whole_map || tiled_ok) && sole_consumer(n.inputs[0]) == cur_) {
                    PlanOp mb;
                    mb.kind = OpKind::MBCONV;
                    mb.name = "mbconv:" + pe.name.substr(pe.name.find(':') + 1) + "+" + n.name;
                    mb.out = op.out;
                    mb.a = pe.a;
                    // expand filters repacked for the kernel: [C][KW] rows = Cin weights | zeros, KW = Cin rounded up to
                    // 8-wide K groups (the bias starts the accumulators)
                    if (whole_map && map2 && map_ws) {
                        // mbmap_ws.hip: the filters as bf16 planes in fragment order (plan_rules.h)
                        mb.w = Ref{Space::CONSTS, add_const(pack_mbmap_w3p(plan_.consts[pe.w.id].data() + pe.w.offset, Cin, pe.gemm.K)), 0};
                    } else if (whole_map && map2 && mshape.cin_pad != pe.gemm.K) {
                        // mbmap.hip with padded k: the filter rows get the zeros the input rows get from the zero page
                        const int64_t Kc = pe.gemm.K, KP = mshape.cin_pad;
                        const std::vector<float> &w0 = plan_.consts[pe.w.id];
                        std::vector<float> wpk((size_t)(Cin * KP), 0.0f);
                        for (int64_t nn = 0; nn < Cin; nn++)
                            for (int64_t k = 0; k < Kc; k++) wpk[nn * KP + k] = w0[pe.w.offset + nn * Kc + k];
                        mb.w = Ref{Space::CONSTS, add_const(wpk), 0};
                    } else if (whole_map && map2 && map_b3) {
                        // mbmap.hip's bf16x3 form: the filters in the order of its LDS chunk image (plan_rules.h)
                        mb.w = Ref{Space::CONSTS, add_const(pack_mbmap_w3f(plan_.consts[pe.w.id].data() + pe.w.offset, Cin, pe.gemm.K)), 0};
                    } else if (whole_map) {
                        mb.w = pe.w;  // [C][Cin] as the GEMM had it
                    } else {
                        const int64_t Kc = pe.gemm.K, KW = (Kc + 7) / 8 * 8;
                        const std::vector<float> &w0 = plan_.consts[pe.w.id];
                        std::vector<float> wpk((size_t)(Cin * KW), 0.0f);
                        for (int64_t nn = 0; nn < Cin; nn++)
                            for (int64_t k = 0; k < Kc; k++) wpk[nn * KW + k] = w0[pe.w.offset + nn * Kc + k];
                        mb.w = Ref{Space::CONSTS, add_const(wpk), 0};
                    }
                    mb.bias = pe.bias;
                    mb.w2 = op.w; mb.bias2 = op.bias;
                    MbDesc &m = mb.mb;
                    m.H = (int32_t)H; m.W = (int32_t)W; m.Cin = pe.gemm.K; m.C = (int32_t)Cin; m.OH = (int32_t)OH; m.OW = (int32_t)OW;
                    m.k = (int32_t)kw; m.s = (int32_t)strides[1]; m.pt = (int32_t)pt; m.pl = (int32_t)pl;
                    m.act1 = pe.gemm.act; m.p0_1 = pe.gemm.p0; m.p1_1 = pe.gemm.p1; m.has_bias1 = pe.gemm.has_bias;
                    m.act2 = act.act; m.p0_2 = act.p0; m.p1_2 = act.p1; m.has_bias2 = has_bias;
                    m.in_bs = pe.gemm.a_bs; m.out_bs = d.out_bs;
                    const int toh = m.s == 1 ? 8 : 4, tow = m.s == 1 ? 16 : 8;
                    m.tiles_x = (int32_t)((OW + tow - 1) / tow); m.tiles_y = (int32_t)((OH + toh - 1) / toh);
                    double halo = (double)((toh - 1) * m.s + m.k) * ((tow - 1) * m.s + m.k) * m.tiles_x * m.tiles_y;
                    if (whole_map) {
                        m.whole_map = map2 ? 2 : 1;
                        m.tiles_x = m.tiles_y = 1;  // one squeeze partial per sample
                        halo = (double)H * W;
                        if (map2) {
                            m.map_bands = mshape.bands; m.map_tr = mshape.tr; m.cin_pad = mshape.cin_pad;
                            m.map_b3 = map_b3; m.map_ws = map_ws;
                            m.tiles_y = mshape.bands;  // ... per band
                            // expand work performed: every band expands the 6 rows it loads (the rows two bands share twice), over the padded k
                            if (mshape.bands > 1) halo = (double)mshape.bands * 6.0 * (double)(mshape.tr ? H : W);
                            halo *= (double)mshape.cin_pad / (double)pe.gemm.K;
                        }
                    } else {
                        row_streaming(m, halo);
                    }
                    mb.macs = op.macs;                                   // depthwise part (VALU)
                    mb.weight_bytes = op.weight_bytes + pe.weight_bytes;
                    mb.bytes = 4.0 * ((double)H * W * pe.gemm.K + (double)OH * OW * Cin);
                    mb.mfma = false;
                    mb.macs_mfma_extra = halo * pe.gemm.K * Cin;          // expand part incl. halo recompute
                    mb.macs_recompute = std::max(0.0, mb.macs_mfma_extra - (double)H * W * (double)pe.gemm.K * Cin);
                    plan_.ops.pop_back();
                    push_op(std::move(mb));
                    define(cur, out);
                    return;
                }
            }
        } else {
            x = to_channels_last(x, n.name);
            op.kind = OpKind::CONV;
            ConvDesc &d = op.conv;
            d.H = (int32_t)H; d.W = (int32_t)W; d.Cin = (int32_t)Cin; d.OH = (int32_t)OH; d.OW = (int32_t)OW; d.Cout = (int32_t)Cout;
            d.kh = (int32_t)kh; d.kw = (int32_t)kw; d.sh = (int32_t)strides[0]; d.sw = (int32_t)strides[1];
            d.pt = (int32_t)pt; d.pl = (int32_t)pl; d.dh = (int32_t)dil[0]; d.dw = (int32_t)dil[1]; d.groups = (int32_t)groups;
            d.act = act.act; d.p0 = act.p0; d.p1 = act.p1; d.has_bias = has_bias; d.has_res = has_res;
            d.in_bs = batch_stride(x); d.out_bs = plan_.storages[out.storage].elems;
            // [Cout][cpg][kh][kw] -> [kh][kw][cpg][Cout]
            std::vector<float> wp(wf.size());
            for (int64_t o = 0; o < Cout; o++)
                for (int64_t c = 0; c < cpg; c++)
                    for (int64_t k = 0; k < kh * kw; k++) wp[(k * cpg + c) * Cout + o] = wf[(o * cpg + c) * kh * kw + k];
            op.w = Ref{Space::CONSTS, add_const(wp), 0};
            op.a = ref_of(x);
        }
        push_op(std::move(op));
        define(cur, out);
    }

    void lower_matmul(const OnnxNode &n) {
        Val a = get(n, 0);
        const Val &b = get(n, 1);
        bool gemm = n.op_type == "Gemm";
        if (a.is_const || !b.is_const) unsupported(n, "needs an activation left operand and a constant right operand");
        if (b.dims.size() != 2) unsupported(n, "right operand must be a 2-D constant");
        bool transB = gemm && n.attr_i("transB", 0);
        if (gemm && n.attr_i("transA", 0)) unsupported(n, "transA is not supported");
        float alpha = gemm ? n.attr_f("alpha", 1.0f) : 1.0f, beta = gemm ? n.attr_f("beta", 1.0f) : 1.0f;
        int64_t K = transB ? b.dims[1] : b.dims[0], N = transB ? b.dims[0] : b.dims[1];
        if (a.dims.empty() || a.dims.back() != K) unsupported(n, "inner dimensions disagree: " + dims_str(a.dims) + " x " + dims_str(b.dims));
        // A must be row-major contiguous so rows collapse to (rows, K) with lda = K
        if (!a.contiguous()) a = materialize(a, row_major(a.dims), n.name);
        int64_t rows = a.numel() / K;
        std::vector<float> wp((size_t)(N * K));  // [N][K]
        for (int64_t nn = 0; nn < N; nn++)
            for (int64_t k = 0; k < K; k++) wp[nn * K + k] = alpha * (transB ? b.f[nn * K + k] : b.f[k * N + nn]);
        std::vector<float> bias;
        bool has_bias = false;
        if (gemm && has_input(n, 2)) {
            const Val &c = get(n, 2);
            if (!c.is_const) unsupported(n, "Gemm C must be constant");
            if (c.numel() == N) bias = c.f;
            else if (c.numel() == 1) bias.assign(N, c.f[0]);
            else unsupported(n, "Gemm C must broadcast along rows");
            for (auto &v : bias) v *= beta;
            has_bias = true;
        }
        std::string cur = n.outputs[0];
        if (!has_bias) {  // MatMul + Add(const [N])
            int c = sole_consumer(cur);
            if (c >= 0 && nodes_[c].op_type == "Add") {
                const OnnxNode &ad = nodes_[c];
                const std::string &other = ad.inputs[0] == cur ? ad.inputs[1] : ad.inputs[0];
                auto it = vals_.find(other);
                if (it != vals_.end() && it->second.is_const && !it->second.is_int && it->second.numel() == N &&
                    (it->second.dims.empty() || it->second.dims.back() == N)) {
                    bias = it->second.f;
                    has_bias = true;
                    absorbed_[c] = true;
                    cur = ad.outputs[0];
                }
            }
        }
        ActSpec act;
        absorb_activation(cur, act);
        Dims od = a.dims;
        od.back() = N;
        Val out = new_act(od, row_major(od));
        PlanOp op;
        op.kind = OpKind::GEMM;
        op.mfma = true;
        op.name = n.op_type + ":" + n.name;
        op.out = ref_of(out);
        op.a = ref_of(a);
        op.w = Ref{Space::CONSTS, add_const(wp), 0};
        if (has_bias) op.bias = Ref{Space::CONSTS, add_const(bias), 0};
        GemmDesc &g = op.gemm;
        g.rows = rows; g.K = (int32_t)K; g.N = (int32_t)N;
        g.lda = K; g.a_bs = batch_stride(a);
        g.ldc = N; g.c_bs = plan_.storages[out.storage].elems;
        g.act = act.act; g.p0 = act.p0; g.p1 = act.p1; g.has_bias = has_bias;
        op.macs = (double)rows * K * N;
        op.weight_bytes = 4.0 * (wp.size() + bias.size());
        op.bytes = 4.0 * ((double)rows * K + (double)rows * N);
        push_op(std::move(op));
        define(cur, out);
    }

    // --------------------------------------------------------- elementwise chain fusion
    void all_refs(PlanOp &op, std::vector<Ref *> &out) {
        out = {&op.out, &op.a, &op.b, &op.res, &op.scale, &op.w2, &op.bias2};
        for (auto &r : op.eb) out.push_back(&r);
        for (auto &r : op.x) out.push_back(&r);
    }
    void recompute_liveness() {
        for (auto &st : plan_.storages) { st.first = -1; st.last = -1; }
        std::vector<Ref *> refs;
        for (size_t k = 0; k < plan_.ops.size(); k++) {
            all_refs(plan_.ops[k], refs);
            for (Ref *r : refs) touch(*r, (int)k);
        }
    }
    // An ELT launch whose result is read by exactly one later ELT launch, as that launch's primary
    // operand over the very same index space, is folded into it (its stages are prepended).
    void fuse_elementwise_chains() {
        bool changed = true;
        while (changed) {
            changed = false;
            recompute_liveness();
            // users per storage
            std::vector<std::vector<int>> users(plan_.storages.size());
            std::vector<Ref *> refs;
            for (size_t k = 0; k < plan_.ops.size(); k++) {
                all_refs(plan_.ops[k], refs);
                for (Ref *r : refs)
                    if (r->space == Space::ARENA && (users[r->id].empty() || users[r->id].back() != (int)k)) users[r->id].push_back((int)k);
            }
            for (size_t j = 0; j < plan_.ops.size() && !changed; j++) {
                PlanOp &cons = plan_.ops[j];
                if (cons.kind != OpKind::ELT || cons.a.space != Space::ARENA) continue;
                const auto &u = users[cons.a.id];
                if (u.size() != 2 || u[1] != (int)j || plan_.storages[cons.a.id].pinned) continue;
                PlanOp &prod = plan_.ops[u[0]];
                if (prod.kind != OpKind::ELT || prod.out.space != Space::ARENA || prod.out.id != cons.a.id) continue;
                const int64_t delta = cons.a.offset - prod.out.offset;  // consumer may read a shifted / reversed view
                if (prod.elt.nstages + cons.elt.nstages > ELT_MAX_STAGES) continue;
                // (A) same index space: the consumer reads exactly what the producer wrote, or
                // (B) the producer is a flat map over a dense buffer (offset in == offset out), so the
                //     consumer's own strides address the producer's operands directly
                const EltDesc &pe = prod.elt, &ce = cons.elt;
                if (pe.bo != ce.ba) continue;
                bool same = pe.nd == ce.nd && delta == 0;
                for (int k = 0; same && k < pe.nd; k++) same = pe.size[k] == ce.size[k] && (pe.size[k] == 1 || pe.so[k] == ce.sa[k]);
                bool flat_prod = pe.nd == 1 && pe.so[0] == 1 && pe.sa[0] == 1 && pe.per_sample == ce.per_sample;
                for (int k = 0; flat_prod && k < pe.nstages; k++)
                    flat_prod = pe.st[k].bin == BIN_NONE || pe.st[k].sb[0] == 0 || pe.st[k].sb[0] == 1;
                // (C) the consumer is a flat map over the dense intermediate: it adopts the producer's index
                //     space (its own strides scale by the producer's row-major runs)
                bool flat_cons = !same && !flat_prod && ce.nd == 1 && ce.sa[0] == 1 && delta == 0 && pe.per_sample == ce.per_sample;
                int64_t runs[ELT_MAX_DIMS] = {0};
                if (flat_cons) {
                    int64_t run = 1;
                    for (int k = pe.nd - 1; k >= 0; k--) {
                        if (pe.size[k] != 1 && pe.so[k] != run) flat_cons = false;
                        runs[k] = run;
                        run *= pe.size[k];
                    }
                }
                if (!same && !flat_prod && !flat_cons) continue;
                // the consumer must not also use the intermediate as a stage operand
                {
                    bool clash = false;
                    for (int k = 0; k < ce.nstages; k++) clash = clash || (cons.eb[k].space == Space::ARENA && cons.eb[k].id == cons.a.id);
                    if (clash) continue;
                }
                // nothing between the two may overwrite the producer's inputs: storages are single-assignment
                // except concat targets, which are only read after all their writers ran -> safe.
                PlanOp fused = cons;
                fused.name = prod.name + "+" + cons.name;
                fused.a = prod.a;
                EltDesc &fe = fused.elt;
                fe.ba = pe.ba;
                fe.nstages = pe.nstages + ce.nstages;
                if (same) {
                    for (int k = 0; k < pe.nd; k++) fe.sa[k] = pe.sa[k];
                    for (int k = 0; k < pe.nstages; k++) { fe.st[k] = pe.st[k]; fused.eb[k] = prod.eb[k]; }
                } else if (flat_cons) {
                    fe.nd = pe.nd;
                    for (int k = 0; k < pe.nd; k++) { fe.size[k] = pe.size[k]; fe.sa[k] = pe.sa[k]; fe.so[k] = ce.so[0] * runs[k]; }
                    for (int k = 0; k < pe.nstages; k++) { fe.st[k] = pe.st[k]; fused.eb[k] = prod.eb[k]; }
                } else {
                    // flat producer: its primary operand and every full-size stage operand are addressed
                    // with the strides the consumer used for the intermediate (fe.sa is already ce.sa)
                    fused.a.offset += delta;
                    for (int k = 0; k < pe.nstages; k++) {
                        fe.st[k] = pe.st[k];
                        fused.eb[k] = prod.eb[k];
                        const bool full = pe.st[k].bin != BIN_NONE && pe.st[k].sb[0] == 1;
                        if (full) fused.eb[k].offset += delta;
                        for (int q = 0; q < ELT_MAX_DIMS; q++) fe.st[k].sb[q] = (q < ce.nd && full) ? ce.sa[q] : 0;
                    }
                }
                for (int k = 0; k < ce.nstages; k++) {
                    fe.st[pe.nstages + k] = ce.st[k];
                    fused.eb[pe.nstages + k] = cons.eb[k];
                    if (flat_cons)
                        for (int q = 0; q < ELT_MAX_DIMS; q++) fe.st[pe.nstages + k].sb[q] = q < pe.nd ? ce.st[k].sb[0] * runs[q] : 0;
                }
                fused.bytes = prod.bytes + cons.bytes - 8.0 * (double)pe.per_sample;  // the intermediate never touches memory
                plan_.ops[j] = fused;
                plan_.ops.erase(plan_.ops.begin() + u[0]);
                changed = true;
            }
            // (D) a stage operand that is the square x*x of some view, produced only for this stage, is
            //     read straight from x and squared in the consumer (|z|^2 = re*re + im*im in one launch)
            for (size_t j = 0; j < plan_.ops.size() && !changed; j++) {
                PlanOp &cons = plan_.ops[j];
                if (cons.kind != OpKind::ELT) continue;
                EltDesc &ce = cons.elt;
                for (int k = 0; k < ce.nstages && !changed; k++) {
                    EltStage &cs = ce.st[k];
                    const Ref &br = cons.eb[k];
                    if (cs.bin == BIN_NONE || cs.bsq || br.space != Space::ARENA || plan_.storages[br.id].pinned) continue;
                    const auto &u = users[br.id];
                    if (u.size() != 2 || u[1] != (int)j) continue;
                    const PlanOp &prod = plan_.ops[u[0]];
                    const EltDesc &pe = prod.elt;
                    if (prod.kind != OpKind::ELT || prod.out.space != Space::ARENA || prod.out.id != br.id || prod.out.offset != br.offset) continue;
                    if (pe.nstages != 1 || pe.st[0].bin != BIN_MUL || pe.st[0].act != ACT_NONE || pe.st[0].bsq) continue;
                    if (prod.eb[0].space != prod.a.space || prod.eb[0].id != prod.a.id || prod.eb[0].offset != prod.a.offset || pe.st[0].bb != pe.ba) continue;
                    if (cons.a.space == Space::ARENA && cons.a.id == br.id) continue;
                    bool match = pe.nd == ce.nd && pe.bo == cs.bb;
                    for (int q = 0; match && q < pe.nd; q++)
                        match = pe.size[q] == ce.size[q] && pe.so[q] == cs.sb[q] && pe.st[0].sb[q] == pe.sa[q];
                    if (!match) continue;
                    cons.eb[k] = prod.a;
                    for (int q = 0; q < ELT_MAX_DIMS; q++) cs.sb[q] = q < pe.nd ? pe.sa[q] : 0;
                    cs.bb = pe.ba;
                    cs.bsq = 1;
                    cons.name += "+sq(" + prod.name + ")";
                    cons.bytes += prod.bytes - 12.0 * (double)pe.per_sample + 4.0 * (double)pe.per_sample;
                    plan_.ops.erase(plan_.ops.begin() + u[0]);
                    changed = true;
                }
            }
        }
        absorb_chains_into_gemms();
        absorb_into_stft();
        planar_stft_tables();
        fuse_fold_pairs();
        pair_minmax_reductions();
        fuse_gap_into_gemm();
        absorb_se_into_gemms();
        recompute_liveness();
    }

    // (I) The two small squeeze-excite products move into the prologue of the GEMM that consumes their gate, when that GEMM
    // runs on the LDS-DMA kernel (gemm_dma.hip): every block of a sample recomputes the gate from the squeeze partial sums
    // while its first K steps are in flight.  One launch less per MBConv block -- 7 - 8 us of the serial chain of a step,
    // of which 3.6 us are the launch itself -- for C x Cr x 8 bytes of L2 reads per block; only up to BN_SEGEMM_MAXC
    // channels (default 768: at C = 1152 the 442 KB of excite weights per block cost more than the launch they save).
    // MEASURED (batch 32, v2.4): slower.  A block has 2 - 8 waves for two products the stand-alone excite kernel spreads
    // over 80 per sample; the prologue is a chain of load latencies (K = 672: project conv 20 -> 52 us against 20 + 8 for
    // the two launches), four contexts 52.7 k -> 47.2 k segments/s.  Opt-in (BN_SEGEMM=1), kept under test.
    void absorb_se_into_gemms() {
        if (!(getenv("BN_SEGEMM") && atoi(getenv("BN_SEGEMM")) == 1)) return;
        const int max_c = gemm_dma_se_max_channels();
        std::vector<Ref *> refs;
        for (size_t j = 0; j < plan_.ops.size(); j++) {
            if (plan_.ops[j].kind != OpKind::SEFC || plan_.ops[j].out.space != Space::ARENA) continue;
            const int gate_id = plan_.ops[j].out.id;
            if (plan_.storages[gate_id].pinned) continue;
            int cons = -1, users = 0;
            for (size_t k = 0; k < plan_.ops.size(); k++) {
                if (k == j) continue;
                all_refs(plan_.ops[k], refs);
                bool uses = false;
                for (Ref *r : refs) uses = uses || (r->space == Space::ARENA && r->id == gate_id);
                if (uses) { users++; cons = (int)k; }
            }
            if (users != 1 || cons < (int)j) continue;
            PlanOp &g = plan_.ops[cons];
            const PlanOp &se = plan_.ops[j];
            if (g.kind != OpKind::GEMM || !g.gemm.has_scale || g.scale.space != Space::ARENA || g.scale.id != gate_id || g.scale.offset != 0 || g.se_fused) continue;
            if (se.se.C != g.gemm.K || se.se.C > max_c || se.se.C % 4 || !gemm_dma_shape(g.gemm)) continue;
            g.se_fused = 1;
            g.se = se.se;
            g.gemm.se_inline = 1;
            g.b = se.a;       // squeeze partial sums
            g.x[0] = se.w;    // W1 [Cr][C]
            g.x[1] = se.bias;
            g.x[2] = se.w2;   // W2 transposed [Cr][C]
            g.x[3] = se.bias2;
            g.name = se.name + "+" + g.name;
            g.macs_valu_extra += se.macs;
            g.weight_bytes += se.weight_bytes;
            g.bytes += se.bytes;
            plan_.ops.erase(plan_.ops.begin() + (long)j);
            j--;
        }
    }

    // (G) Neighbours of an STFT launch move into it:
    //  * the mel filter bank -- a GEMM over the spectrum rows with a SPARSE constant matrix (triangular filters: a few
    //    bins per band) -- together with whatever rule E put into that GEMM's epilogue (compression chain, layout copy
    //    into the spectrogram image): the spectrum rows never leave LDS.  BN_STFT_MEL=0 disables.
    //  * the elementwise chain that produced the signal, when every stage operand is one number per sample (the
    //    min-max normalisation of the v2.4 graph) and every reader of its result is an STFT launch: the chain runs
    //    while the span is loaded, the normalised segment is never written.  BN_STFT_PRE=0 disables.
    void absorb_into_stft() {
        const bool mel_on = !(getenv("BN_STFT_MEL") && std::string(getenv("BN_STFT_MEL")) == "0");
        const bool pre_on = !(getenv("BN_STFT_PRE") && std::string(getenv("BN_STFT_PRE")) == "0");
        auto users_of = [&]() {
            std::vector<std::vector<int>> users(plan_.storages.size());
            std::vector<Ref *> refs;
            for (size_t k = 0; k < plan_.ops.size(); k++) {
                all_refs(plan_.ops[k], refs);
                for (Ref *r : refs)
                    if (r->space == Space::ARENA && (users[r->id].empty() || users[r->id].back() != (int)k)) users[r->id].push_back((int)k);
            }
            return users;
        };
        //  * (round 3) the power / magnitude pass behind a cos | sin bank: an elementwise launch  re*re + im*im [-> sqrt]  over
        //    the two halves of every spectrum row, whose only reader it is -- the launch computes both linear forms of a bin
        //    in one lane and stores f(u^2 + v^2): half the spectrum bytes written, none read back, one launch less, and the
        //    mel bank that follows becomes a direct neighbour (next rule).  BN_STFT_POWER=0 disables.
        if (!(getenv("BN_STFT_POWER") && std::string(getenv("BN_STFT_POWER")) == "0")) {
            bool again = true;
            while (again) {
                again = false;
                auto users = users_of();
                for (size_t i = 0; i < plan_.ops.size() && !again; i++) {
                    PlanOp &f = plan_.ops[i];
                    FftDesc &d = f.fft;
                    if (f.kind != OpKind::FFT || d.nmel || d.power || d.has_bias || d.npost || f.out.space != Space::ARENA || plan_.storages[f.out.id].pinned) continue;
                    if (d.nout % 2 || d.ldc != d.nout || d.out_rs != d.nout || d.out_cs != 1) continue;
                    const auto &u = users[f.out.id];
                    if (u.size() != 2 || u[0] != (int)i) continue;
                    PlanOp &el = plan_.ops[u[1]];
                    const EltDesc &e = el.elt;
                    const int64_t nb = d.nout / 2;
                    if (el.kind != OpKind::ELT || e.nd != 2 || e.size[0] != d.frames || e.size[1] != nb || e.sa[0] != d.nout || e.sa[1] != 1 || e.ba != d.c_bs) continue;
                    if (el.a.space != Space::ARENA || el.a.id != f.out.id || el.a.offset != f.out.offset) continue;
                    if (el.out.space != Space::ARENA || e.so[1] != 1 || e.so[0] < nb) continue;
                    // stage 0: x * x (the operand is the same view); stage 1: + y^2 with y the other half of the row; [stage 2: sqrt]
                    if (e.nstages < 2 || e.nstages > 3) continue;
                    const EltStage &s0 = e.st[0], &s1 = e.st[1];
                    const Ref &r0 = el.eb[0], &r1 = el.eb[1];
                    if (s0.bin != BIN_MUL || s0.bsq || s0.act != ACT_NONE || r0.space != Space::ARENA || r0.id != f.out.id || r0.offset != f.out.offset ||
                        s0.sb[0] != d.nout || s0.sb[1] != 1 || s0.bb != d.c_bs)
                        continue;
                    if (s1.bin != BIN_ADD || !s1.bsq || s1.act != ACT_NONE || r1.space != Space::ARENA || r1.id != f.out.id || r1.offset != f.out.offset + nb ||
                        s1.sb[0] != d.nout || s1.sb[1] != 1 || s1.bb != d.c_bs)
                        continue;
                    if (e.nstages == 3 && !(e.st[2].bin == BIN_NONE && e.st[2].act == ACT_SQRT)) continue;
                    // the two halves must be the two forms of the SAME bins in the same order (same buffer positions)
                    const std::vector<float> &ot = plan_.consts[f.bias2.id];
                    bool same = (int64_t)ot.size() >= f.bias2.offset + d.nout * 8;
                    for (int64_t c = 0; c < nb && same; c++) {
                        const float *u0 = &ot[(size_t)(f.bias2.offset + c * 8)], *v0 = &ot[(size_t)(f.bias2.offset + (c + nb) * 8)];
                        same = u0[0] == v0[0] && u0[1] == v0[1];
                    }
                    if (!same) continue;
                    std::vector<float> ot2((size_t)nb * 12, 0.0f);
                    for (int64_t c = 0; c < nb; c++) {
                        const float *u0 = &ot[(size_t)(f.bias2.offset + c * 8)], *v0 = &ot[(size_t)(f.bias2.offset + (c + nb) * 8)];
                        float *o = &ot2[(size_t)c * 12];
                        for (int q = 0; q < 6; q++) o[q] = u0[q];
                        for (int q = 0; q < 4; q++) o[8 + q] = v0[2 + q];
                    }
                    FftDesc probe = d;
                    probe.nout = (int32_t)nb; probe.otab_stride = 12; probe.power = e.nstages == 3 ? 2 : 1;
                    if (stft_lds_bytes(probe, 8) > 156 * 1024) continue;
                    d = probe;
                    d.ldc = e.so[0]; d.out_rs = e.so[0]; d.out_cs = 1; d.c_bs = e.bo;
                    f.bias2 = Ref{Space::CONSTS, add_const(ot2), 0};
                    f.out = el.out;
                    f.name += "+" + el.name;
                    f.weight_bytes += 4.0 * (double)ot2.size() - 4.0 * 8.0 * (double)(2 * nb);
                    f.bytes -= 4.0 * (double)d.frames * (double)nb;  // one value per bin is written instead of two; the cos | sin rows are never read back
                    f.macs += 2.0 * (double)d.frames * (double)nb;
                    f.flops_fft += 3.0 * (double)d.frames * (double)nb;
                    plan_.ops.erase(plan_.ops.begin() + u[1]);
                    again = true;
                }
            }
        }
        bool changed = mel_on;
        while (changed) {
            changed = false;
            auto users = users_of();
            for (size_t i = 0; i < plan_.ops.size() && !changed; i++) {
                PlanOp &f = plan_.ops[i];
                if (f.kind != OpKind::FFT || f.fft.nmel || f.out.space != Space::ARENA || plan_.storages[f.out.id].pinned) continue;
                if (f.fft.tpb & (f.fft.tpb - 1)) continue;  // the mel phases index a tile's frames by shifts / 16-frame matrix tiles (M = 5 q banks: tiles of 24)
                const auto &u = users[f.out.id];
                if (u.size() != 2 || u[0] != (int)i) continue;
                PlanOp &g = plan_.ops[u[1]];
                const GemmDesc &gd = g.gemm;
                if (g.kind != OpKind::GEMM || g.a.space != Space::ARENA || g.a.id != f.out.id || g.a.offset != f.out.offset) continue;
                if (gd.fold || gd.has_scale || gd.has_res || gd.rows != f.fft.frames || gd.K != f.fft.nout || gd.lda != gd.K || gd.a_bs != f.fft.c_bs ||
                    f.fft.ldc != f.fft.nout || g.w.space != Space::CONSTS)
                    continue;
                bool stages_ok = stft_act_supported(gd.act);
                for (int q = 0; q < gd.npost; q++) stages_ok = stages_ok && stft_act_supported(gd.post_act[q]);
                if (!stages_ok) continue;
                const std::vector<float> &W = plan_.consts[g.w.id];  // [N][K]
                const int64_t N = gd.N, K = gd.K;
                std::vector<float> mstart, ment;  // row starts; (column, value) pairs
                for (int64_t nn = 0; nn < N; nn++) {
                    mstart.push_back((float)(ment.size() / 2));
                    for (int64_t k = 0; k < K; k++) {
                        const float v = W[(size_t)(g.w.offset + nn * K + k)];
                        if (v != 0.0f) { ment.push_back((float)k); ment.push_back(v); }
                    }
                }
                mstart.push_back((float)(ment.size() / 2));
                const size_t nnz_count = ment.size() / 2;
                if ((double)nnz_count > 0.3 * (double)N * (double)K || nnz_count >= (1u << 23) || N > 1024) continue;  // dense: stays a GEMM
                FftDesc probe = f.fft;
                probe.nmel = (int32_t)N;
                probe.mel_nnz = (int32_t)nnz_count;
                // the tile's spectrum rows join the LDS image.  Where 16 frames per tile no longer fit (v3.0: 513 bins, 128 bands)
                // tiles of 8 would, but the launch then loses to FFT + dense mel GEMM (200 us against 97 + 66 at batch 64,
                // 41.5 k against 44.0 k segments/s): BN_STFT_MEL=force takes the smaller tile anyway (tests)
                if (getenv("BN_STFT_MEL") && std::string(getenv("BN_STFT_MEL")) == "force")
                    while (stft_lds_bytes(probe, 8) > 156 * 1024 && probe.tpb > 8 && (probe.tpb / 2) % probe.F == 0) probe.tpb /= 2;
                if (stft_lds_bytes(probe, 8) > 156 * 1024) continue;
                FftDesc &d = f.fft;
                d.tpb = probe.tpb;
                d.nmel = (int32_t)N;
                d.mel_nnz = (int32_t)nnz_count;
                d.mel_mode = 0; d.mel_groups = 0; d.spec_stride = d.nout;
                // the bank as 16 x 16 tiles for the matrix cores (kernels.h, FftDesc::mel_mode): tile rows of 16 bands, of each
                // only the 16-bin groups that hold a non-zero; a tile is stored in the lane order of the A fragment (lane (i, q)
                // holds band i, bins 16 g + 4 q + 0..3).  BN_STFT_MELMFMA=0 keeps the (column, weight) lists on the vector ALU.
                if (!(getenv("BN_STFT_MELMFMA") && std::string(getenv("BN_STFT_MELMFMA")) == "0") && d.tpb == 16) {
                    const int64_t ntile = (N + 15) / 16, ngrp = (K + 15) / 16;
                    std::vector<float> tab((size_t)ntile + 1, 0.0f), glist, pack;
                    for (int64_t t = 0; t < ntile; t++) {
                        tab[(size_t)t] = (float)glist.size();
                        for (int64_t gq = 0; gq < ngrp; gq++) {
                            bool any = false;
                            for (int64_t i = 0; i < 16 && !any; i++)
                                for (int64_t k = 0; k < 16 && !any; k++) {
                                    const int64_t nn = 16 * t + i, kk = 16 * gq + k;
                                    any = nn < N && kk < K && W[(size_t)(g.w.offset + nn * K + kk)] != 0.0f;
                                }
                            if (!any) continue;
                            glist.push_back((float)gq);
                            for (int lane = 0; lane < 64; lane++)
                                for (int j = 0; j < 4; j++) {
                                    const int64_t nn = 16 * t + (lane & 15), kk = 16 * gq + 4 * (lane >> 4) + j;
                                    pack.push_back(nn < N && kk < K ? W[(size_t)(g.w.offset + nn * K + kk)] : 0.0f);
                                }
                        }
                    }
                    tab[(size_t)ntile] = (float)glist.size();
                    FftDesc probe2 = d;
                    probe2.mel_mode = 1; probe2.mel_groups = (int32_t)glist.size(); probe2.spec_stride = (int32_t)(16 * ngrp + 8);
                    if (!glist.empty() && glist.size() < (1u << 22) && stft_lds_bytes(probe2, 8) <= 156 * 1024) {
                        d = probe2;
                        tab.insert(tab.end(), glist.begin(), glist.end());
                        mstart = tab;
                        ment = pack;
                    }
                }
                if (ment.empty()) { ment.push_back(0.0f); ment.push_back(0.0f); }
                d.mel_has_bias = gd.has_bias;
                d.mel_act = gd.act; d.mel_p0 = gd.p0; d.mel_p1 = gd.p1;
                d.npost = gd.npost;
                for (int q = 0; q < 4; q++) { d.post_act[q] = gd.post_act[q]; d.post_p0[q] = gd.post_p0[q]; d.post_p1[q] = gd.post_p1[q]; }
                d.c_bs = gd.c_bs;
                if (gd.out_strided) { d.out_rs = gd.out_rs; d.out_cs = gd.out_cs; }
                else { d.out_rs = gd.ldc; d.out_cs = 1; }
                f.x[0] = Ref{Space::CONSTS, add_const(mstart), 0};
                f.x[1] = Ref{Space::CONSTS, add_const(ment), 0};
                f.x[3] = g.bias;
                f.out = g.out;
                f.name += "+" + g.name;
                const double nnz = d.mel_mode == 1 ? 256.0 * (double)d.mel_groups : (double)nnz_count;  // multiply-adds performed per frame
                f.flops_fft += 2.0 * nnz * (double)d.frames;
                f.macs += nnz * (double)d.frames;
                f.weight_bytes += 4.0 * (mstart.size() + ment.size());
                f.bytes += g.bytes - 8.0 * (double)d.frames * (double)d.nout;  // the spectrum rows never touch memory
                dft_gemm_macs_ += g.macs;  // the dense mel product belongs to the matrix-product count of the front end
                dft_performed_macs_ += nnz * (double)d.frames;
                dft_fft_equiv_flops_ += 2.0 * nnz * (double)d.frames;
                plan_.ops.erase(plan_.ops.begin() + u[1]);
                changed = true;
            }
        }
        changed = pre_on;
        while (changed) {
            changed = false;
            auto users = users_of();
            for (size_t e = 0; e < plan_.ops.size() && !changed; e++) {
                PlanOp &el = plan_.ops[e];
                const EltDesc &ed = el.elt;
                if (el.kind != OpKind::ELT || el.out.space != Space::ARENA || plan_.storages[el.out.id].pinned) continue;
                if (ed.nd != 1 || ed.so[0] != 1 || ed.sa[0] != 1 || el.out.offset != 0) continue;
                bool scalar_ops = true;
                for (int k = 0; k < ed.nstages; k++)
                    scalar_ops = scalar_ops && stft_act_supported(ed.st[k].act) &&
                                 (ed.st[k].bin == BIN_NONE || (ed.st[k].sb[0] == 0 && !ed.st[k].bsq && stft_bin_supported(ed.st[k].bin)));
                if (!scalar_ops) continue;
                const auto &u = users[el.out.id];
                if (u.size() < 2 || u[0] != (int)e) continue;
                // every reader is a launch that can apply the chain while it loads its span: an STFT, or (round 4) a folded framing GEMM that
                // is certain to run on its LDS-resident kernel (half fold: frame_fold_post_ok; quarter fold: always)
                auto framing_gemm = [&](const PlanOp &f) {
                    const GemmDesc &g = f.gemm;
                    if (f.kind != OpKind::GEMM || f.pre.n != 0 || f.gemm2.N > 0 || f.se_fused || env_int("BN_FRAME_PRE", 1) == 0) return false;
                    if (env_int("BN_FRAMEPAIR", 0) == 1) return false;  // (opt-in rule J runs later and its kernel does not carry the chain)
                    if (!(g.fold == 2 ? frame_fold2_shape_ok(g, nullptr) : (g.fold == 1 || g.fold == -1) && frame_fold_post_ok(g))) return false;
                    return g.a_bs == ed.bo && (g.rows - 1) * g.lda + g.fold_n <= ed.per_sample;
                };
                bool all_fft = true;
                for (size_t q = 1; q < u.size(); q++) {
                    const PlanOp &f = plan_.ops[u[q]];
                    const bool stft_ok = f.kind == OpKind::FFT && f.fft.npre == 0 && f.fft.a_bs == ed.bo && (int64_t)(f.fft.frames - 1) * f.fft.hop + f.fft.L <= ed.per_sample;
                    all_fft = all_fft && (stft_ok || framing_gemm(f)) && f.a.space == Space::ARENA && f.a.id == el.out.id && f.a.offset == 0;
                    // (the launch must read the chain's result only through `a`)
                    std::vector<Ref *> refs;
                    all_refs(const_cast<PlanOp &>(f), refs);
                    int reads = 0;
                    for (Ref *r : refs) reads += (r->space == Space::ARENA && r->id == el.out.id) ? 1 : 0;
                    all_fft = all_fft && reads == 1;
                }
                if (!all_fft) continue;
                for (size_t q = 1; q < u.size(); q++) {
                    PlanOp &f = plan_.ops[u[q]];
                    f.a = el.a;
                    if (f.kind == OpKind::FFT) {
                        f.fft.a_bs = ed.ba;
                        f.fft.npre = ed.nstages;
                    } else {
                        f.gemm.a_bs = ed.ba;
                        f.pre.n = ed.nstages;
                    }
                    for (int k = 0; k < ed.nstages; k++) {
                        if (f.kind == OpKind::FFT) {
                            f.fft.pre_bin[k] = ed.st[k].bin; f.fft.pre_act[k] = ed.st[k].act;
                            f.fft.pre_p0[k] = ed.st[k].p0; f.fft.pre_p1[k] = ed.st[k].p1; f.fft.pre_bb[k] = ed.st[k].bb;
                        } else {
                            f.pre.bin[k] = ed.st[k].bin; f.pre.act[k] = ed.st[k].act;
                            f.pre.p0[k] = ed.st[k].p0; f.pre.p1[k] = ed.st[k].p1; f.pre.bb[k] = ed.st[k].bb;
                        }
                        f.eb[k] = el.eb[k];
                    }
                    f.name = el.name + "+" + f.name;
                }
                plan_.ops.erase(plan_.ops.begin() + (long)e);
                changed = true;
            }
        }
        // (round 5) the zero-padded copy of the signal a padded framing conv reads (pad_copy: one fill + one copy launch) in the STFT's span
        // load: where the only reader of that copy is an FFT launch, the launch reads the signal itself and zero-fills the positions that
        // fall into the padding (FftDesc::pad_l, in_len).  Perch's front end: two launches and a round trip of the signal less.
        changed = env_int("BN_STFT_PAD", 1) != 0;
        while (changed) {
            changed = false;
            auto users = users_of();
            for (size_t e = 0; e + 2 < plan_.ops.size() && !changed; e++) {
                const PlanOp &fill = plan_.ops[e], &copy = plan_.ops[e + 1];
                if (fill.kind != OpKind::ELT || copy.kind != OpKind::ELT || fill.name.rfind("pad.fill:", 0) != 0 || copy.name.rfind("pad.copy:", 0) != 0) continue;
                if (fill.out.space != Space::ARENA || copy.out.space != Space::ARENA || fill.out.id != copy.out.id || fill.out.offset != 0 || plan_.storages[fill.out.id].pinned) continue;
                const EltDesc &fd = fill.elt, &cd = copy.elt;
                if (fd.nd != 1 || cd.nd != 1 || fd.nstages != 1 || cd.nstages != 1 || cd.so[0] != 1 || cd.sa[0] != 1) continue;
                if (fd.st[0].bin != BIN_NONE || fd.st[0].act != ACT_NONE || cd.st[0].bin != BIN_NONE || cd.st[0].act != ACT_NONE) continue;
                if (fill.a.space != Space::CONSTS || plan_.consts[(size_t)fill.a.id].empty() || plan_.consts[(size_t)fill.a.id][(size_t)fill.a.offset] != 0.0f) continue;
                if (copy.a.space != Space::ARENA && copy.a.space != Space::INPUT) continue;
                const auto &u = users[fill.out.id];
                if (u.size() != 3 || u[0] != (int)e || u[1] != (int)e + 1) continue;
                PlanOp &f = plan_.ops[(size_t)u[2]];
                if (f.kind != OpKind::FFT || f.fft.in_len != 0 || f.fft.npre != 0 || f.a.space != Space::ARENA || f.a.id != fill.out.id || f.a.offset != 0) continue;
                if (f.fft.a_bs != fd.bo || (int64_t)(f.fft.frames - 1) * f.fft.hop + f.fft.L > fd.per_sample || copy.out.offset + cd.per_sample > fd.per_sample) continue;
                if (cd.per_sample >= ((int64_t)1 << 30) || copy.out.offset >= ((int64_t)1 << 30)) continue;
                std::vector<Ref *> refs;
                all_refs(f, refs);
                int reads = 0;
                for (Ref *r : refs) reads += (r->space == Space::ARENA && r->id == fill.out.id) ? 1 : 0;
                if (reads != 1) continue;
                f.a = copy.a;
                f.fft.a_bs = cd.ba;
                f.fft.pad_l = (int32_t)copy.out.offset;
                f.fft.in_len = (int32_t)cd.per_sample;
                f.bytes -= 4.0 * (double)(fd.per_sample - cd.per_sample);
                f.name = "pad+" + f.name;
                plan_.ops.erase(plan_.ops.begin() + (long)e, plan_.ops.begin() + (long)e + 2);
                changed = true;
            }
        }
    }

    // (K, round 4) GlobalAveragePool behind a 1x1 conv whose 48 rows per sample sit in one block of the LDS-DMA GEMM (48-row tiles): the
    // epilogue writes the mean over the rows instead of the rows (gemm_dma.hip; fixed order: m-tiles ascending, then a butterfly over the 16
    // rows of a tile), the [48, N] tensor is neither written nor read and the reduction launch is gone.  v2.4's head: Conv_261 (320 -> 1024,
    // ReLU) + GlobalAveragePool_264.  BN_GEMMGAP=0 disables.
    void fuse_gap_into_gemm() {
        if (env_int("BN_GEMMGAP", 1) == 0) return;
        bool changed = true;
        while (changed) {
            changed = false;
            std::vector<std::vector<int>> users(plan_.storages.size());
            std::vector<Ref *> refs;
            for (size_t k = 0; k < plan_.ops.size(); k++) {
                all_refs(plan_.ops[k], refs);
                for (Ref *r : refs)
                    if (r->space == Space::ARENA && (users[r->id].empty() || users[r->id].back() != (int)k)) users[r->id].push_back((int)k);
            }
            for (size_t j = 0; j < plan_.ops.size() && !changed; j++) {
                PlanOp &red = plan_.ops[j];
                const ReduceDesc &r = red.red;
                if (red.kind != OpKind::REDUCE || r.op != RED_MEAN || r.pair || red.a.space != Space::ARENA || red.out.space != Space::ARENA) continue;
                const auto &u = users[red.a.id];
                if (u.size() != 2 || u[1] != (int)j || plan_.storages[red.a.id].pinned) continue;
                PlanOp &g = plan_.ops[u[0]];
                if (g.kind != OpKind::GEMM || g.out.space != Space::ARENA || g.out.id != red.a.id || g.out.offset != red.a.offset || g.se_fused || g.gemm2.N > 0) continue;
                GemmDesc d = g.gemm;
                if (d.gap || !gemm_gap_shape_ok(d) || gemm_dma_shape(d) != 2 || d.ldc != d.N || d.c_bs != d.rows * d.N) continue;
                // the reduction: every channel's mean over the sample's rows, [rows, N] -> [N], dense
                // (the reduced dims -- H, W of the map -- nest into one run of `rows` rows of N floats)
                bool rows_run = r.nr >= 1 && r.nr <= 3 && r.rin[r.nr - 1] == d.N;
                int64_t nrows = 1;
                for (int q = r.nr - 1; q >= 0 && rows_run; q--) {
                    rows_run = r.rin[q] == d.N * nrows;
                    nrows *= r.rsize[q];
                }
                if (r.nk != 1 || !rows_run || nrows != d.rows || r.ksize[0] != d.N || r.kin[0] != 1 || r.kout[0] != 1 || r.bi != d.c_bs) continue;
                if (red.out.offset % 4 != 0 || r.bo % 4 != 0) continue;  // (dwordx4 stores)
                d.gap = 1;
                d.c_bs = r.bo;
                g.gemm = d;
                g.out = red.out;
                g.name += "+" + red.name;
                g.bytes += red.bytes - 8.0 * (double)d.rows * d.N;  // the [rows, N] tensor never touches memory
                plan_.ops.erase(plan_.ops.begin() + (long)j);
                changed = true;
            }
        }
    }

    // (J) The product that consumes the rows of a folded framing GEMM -- the mel filter bank of a spectrogram branch that stayed on the
    // matrix path: K2 = the bank's <= 128 live bins, with the compression chain and the layout copy rule E put into its epilogue --
    // runs behind the K loop of the framing kernel on the block's own spectrum tile (kernels.hip, frame_fold_kernel<true>): the
    // spectrum rows are neither written nor read back, one launch less.  Measured slower (kernels.hip, frame_fold_pair_ok): opt-in, BN_FRAMEPAIR=1.
    void fuse_fold_pairs() {
        bool changed = true;
        while (changed) {
            changed = false;
            std::vector<std::vector<int>> users(plan_.storages.size());
            std::vector<Ref *> refs;
            for (size_t k = 0; k < plan_.ops.size(); k++) {
                all_refs(plan_.ops[k], refs);
                for (Ref *r : refs)
                    if (r->space == Space::ARENA && (users[r->id].empty() || users[r->id].back() != (int)k)) users[r->id].push_back((int)k);
            }
            for (size_t i = 0; i < plan_.ops.size() && !changed; i++) {
                PlanOp &f = plan_.ops[i];
                if (f.kind != OpKind::GEMM || !f.gemm.fold || f.gemm2.N > 0 || f.gemm.npost || f.gemm.out_strided || f.se_fused) continue;
                if (f.out.space != Space::ARENA || plan_.storages[f.out.id].pinned) continue;
                const auto &u = users[f.out.id];
                if (u.size() != 2 || u[0] != (int)i) continue;
                PlanOp &g = plan_.ops[u[1]];
                if (g.kind != OpKind::GEMM || g.se_fused || g.gemm2.N > 0 || g.a.space != Space::ARENA || g.a.id != f.out.id || g.a.offset != f.out.offset) continue;
                if (g.w.space != Space::CONSTS || (g.gemm.has_bias && g.bias.space != Space::CONSTS)) continue;
                if (!frame_fold_pair_ok(f.gemm, g.gemm)) continue;
                f.gemm2 = g.gemm;
                f.w2 = g.w;
                f.bias2 = g.bias;
                f.out = g.out;
                f.name += "+" + g.name;
                f.macs += g.macs;
                f.weight_bytes += g.weight_bytes;
                f.bytes += g.bytes - 8.0 * (double)f.gemm.rows * (double)f.gemm.N;  // the spectrum rows never touch memory
                plan_.ops.erase(plan_.ops.begin() + u[1]);
                changed = true;
            }
        }
    }

    // The untangle table of every FFT launch goes from rows [nout][8 | 12] to PLANES: [nout][4] (two buffer positions, first two
    // coefficients), [nout][2] (the other two), [nout][4] (power mode: the second linear form).  The kernel's lanes read consecutive
    // outputs: 32-byte row strides put the 16-byte reads 2-way and the 8-byte reads 4-way on the LDS banks (53 % of the bin phase's
    // LDS cycles were conflicts); consecutive 16- and 8-byte elements of a plane do not conflict.
    void planar_stft_tables() {
        for (PlanOp &f : plan_.ops) {
            FftDesc &d = f.fft;
            if (f.kind != OpKind::FFT || d.otab_planar || f.bias2.space != Space::CONSTS) continue;
            const int stride = d.otab_stride > 0 ? d.otab_stride : 8;
            const std::vector<float> &ot = plan_.consts[f.bias2.id];
            const int64_t n = d.nout;
            if ((int64_t)ot.size() < f.bias2.offset + n * stride) continue;
            std::vector<float> pl((size_t)(n * (d.power ? 10 : 6)), 0.0f);
            for (int64_t c = 0; c < n; c++) {
                const float *r = &ot[(size_t)(f.bias2.offset + c * stride)];
                for (int q = 0; q < 4; q++) pl[(size_t)(c * 4 + q)] = r[q];
                for (int q = 0; q < 2; q++) pl[(size_t)(4 * n + c * 2 + q)] = r[4 + q];
                if (d.power)
                    for (int q = 0; q < 4; q++) pl[(size_t)(6 * n + c * 4 + q)] = r[8 + q];
            }
            f.bias2 = Ref{Space::CONSTS, add_const(pl), 0};
            d.otab_planar = 1;
        }
    }

    // (F) The chunk stages of a whole-range min and a whole-range max over the SAME input (the min-max normalisation
    // after the max(x - s) rewrite) read it in one pass: the later launch is folded into the earlier one (kernels.h,
    // ReduceDesc::pair).  Exact, order-independent operations: same bits.  BN_REDUCE_PAIR=0 disables.
    void pair_minmax_reductions() {
        if (getenv("BN_REDUCE_PAIR") && std::string(getenv("BN_REDUCE_PAIR")) == "0") return;
        for (size_t i = 0; i < plan_.ops.size(); i++) {
            PlanOp &a = plan_.ops[i];
            if (a.kind != OpKind::REDUCE || a.red.pair || a.red.op != RED_MIN) continue;
            for (size_t j = i + 1; j < plan_.ops.size() && j < i + 4; j++) {
                PlanOp &b = plan_.ops[j];
                if (b.kind != OpKind::REDUCE || b.red.pair || b.red.op != RED_MAX) continue;
                const ReduceDesc &x = a.red, &y = b.red;
                const bool same_in = a.a.space == b.a.space && a.a.id == b.a.id && a.a.offset == b.a.offset;
                const bool chunks = x.nk == 1 && x.nr == 1 && x.rin[0] == 1 && x.kout[0] == 1 && x.red % 4 == 0 && x.kin[0] % 4 == 0 && x.bi % 4 == 0 &&
                                    a.a.offset % 4 == 0 && x.inner_kept == 0;
                const bool same_shape = y.nk == 1 && y.nr == 1 && y.rin[0] == 1 && y.kout[0] == 1 && x.ksize[0] == y.ksize[0] && x.kin[0] == y.kin[0] &&
                                        x.red == y.red && x.bi == y.bi && x.kept == y.kept;
                if (!same_in || !chunks || !same_shape) continue;
                // nothing in between may write the shared input or read the max partials (they are only read later)
                bool clash = false;
                for (size_t k = i + 1; k < j; k++) {
                    const PlanOp &m = plan_.ops[k];
                    clash = clash || (m.out.space == a.a.space && m.out.id == a.a.id) || (m.a.space == b.out.space && m.a.id == b.out.id);
                }
                if (clash) continue;
                a.red.pair = 1;
                a.red.bo2 = y.bo;
                a.b = b.out;
                a.name += "+" + b.name;
                a.bytes += 4.0 * (double)y.kept;
                plan_.ops.erase(plan_.ops.begin() + (long)j);
                break;
            }
        }
    }

    // (E) An elementwise chain of unary stages (power-law compression, affine maps, layout copies) whose primary
    // operand is the dense result of the GEMM launched just for it moves into that GEMM's epilogue: the stages run
    // on the accumulators and the store goes through the chain's output view (kernels.h, GemmDesc::npost /
    // out_strided).  v2.4 front end: mel MatMul -> x^2 -> x^p -> flip / transpose into the 2-channel image, per
    // branch one launch instead of two and no dense [frames, mels] round trip.  BN_GEMMPOST=0 disables.
    void absorb_chains_into_gemms() {
        if (getenv("BN_GEMMPOST") && std::string(getenv("BN_GEMMPOST")) == "0") return;
        bool changed = true;
        while (changed) {
            changed = false;
            std::vector<std::vector<int>> users(plan_.storages.size());
            std::vector<Ref *> refs;
            for (size_t k = 0; k < plan_.ops.size(); k++) {
                all_refs(plan_.ops[k], refs);
                for (Ref *r : refs)
                    if (r->space == Space::ARENA && (users[r->id].empty() || users[r->id].back() != (int)k)) users[r->id].push_back((int)k);
            }
            for (size_t j = 0; j < plan_.ops.size() && !changed; j++) {
                PlanOp &cons = plan_.ops[j];
                if (cons.kind != OpKind::ELT || cons.a.space != Space::ARENA || cons.out.space != Space::ARENA) continue;
                const auto &u = users[cons.a.id];
                if (u.size() != 2 || u[1] != (int)j || plan_.storages[cons.a.id].pinned) continue;
                PlanOp &prod = plan_.ops[u[0]];
                if (prod.kind != OpKind::GEMM || prod.out.space != Space::ARENA || prod.out.id != cons.a.id) continue;
                GemmDesc g = prod.gemm;
                const EltDesc &ce = cons.elt;
                // (round 4: the LDS-resident folded framing GEMM carries a chain of the compact stage functions and the consumer's view too)
                const bool fold_post = (g.fold == 1 || g.fold == -1) && frame_fold_post_ok(g);
                if (g.npost || g.out_strided || !(gemm_accepts_post(g) || fold_post) || g.ldc != g.N || g.c_bs != g.rows * g.N || ce.ba != g.c_bs) continue;
                bool unary = true, compact = true;
                int npost = 0;
                for (int k = 0; k < ce.nstages; k++) {
                    unary = unary && ce.st[k].bin == BIN_NONE;
                    if (ce.st[k].act != ACT_NONE) { npost++; compact = compact && stft_act_supported(ce.st[k].act); }
                }
                if (!unary || npost > 4 || ce.per_sample != g.rows * g.N || (fold_post && !compact)) continue;
                // the chain's index space must be (a signed permutation of) rows x N
                int64_t rs = 0, cs = 0, base = cons.out.offset, want_delta = 0;
                bool ok = true, have_row = g.rows == 1, have_col = g.N == 1;
                for (int k = 0; k < ce.nd && ok; k++) {
                    if (ce.size[k] == 1) continue;
                    const int64_t sa = ce.sa[k], mag = sa < 0 ? -sa : sa;
                    int64_t *dst = nullptr;
                    if (!have_col && !have_row && sa == 1 && ce.size[k] == g.rows * g.N) {  // flat map over the dense result
                        rs = g.N * ce.so[k]; cs = ce.so[k]; have_row = have_col = true;
                        continue;
                    }
                    if (!have_col && mag == 1 && ce.size[k] == g.N) { dst = &cs; have_col = true; }
                    else if (!have_row && mag == g.N && ce.size[k] == g.rows) { dst = &rs; have_row = true; }
                    else { ok = false; break; }
                    if (sa > 0) *dst = ce.so[k];
                    else { *dst = -ce.so[k]; base += (ce.size[k] - 1) * ce.so[k]; want_delta += (ce.size[k] - 1) * mag; }
                }
                if (!ok || !have_row || !have_col || cons.a.offset - prod.out.offset != want_delta) continue;
                g.out_strided = 1;
                g.out_rs = rs; g.out_cs = cs;
                g.c_bs = ce.bo;
                g.npost = 0;
                for (int k = 0; k < ce.nstages; k++)
                    if (ce.st[k].act != ACT_NONE) {
                        g.post_act[g.npost] = ce.st[k].act; g.post_p0[g.npost] = ce.st[k].p0; g.post_p1[g.npost] = ce.st[k].p1;
                        g.npost++;
                    }
                prod.gemm = g;
                prod.out = Ref{cons.out.space, cons.out.id, base};
                prod.name += "+" + cons.name;
                prod.bytes += cons.bytes - 8.0 * (double)ce.per_sample;  // the dense intermediate never touches memory
                plan_.ops.erase(plan_.ops.begin() + j);
                changed = true;
            }
        }
    }

    // --------------------------------------------------------- memory plan
    // (round 5) every GEMM the one-tile-per-block LDS-DMA kernel takes gets its weights as three exact bf16 planes (plan_rules.h, pack_w3):
    // the launch then runs on the bf16 matrix pipe with f32-complete products (gemm_dma3.hip).  Last pass over the descriptors: gate, pooled
    // epilogue, inline squeeze-excite are all decided.  The f32 copy goes away with it when no other launch reads it.
    void pack_bf16x3_weights() {
        auto refs_of = [](PlanOp &op) {
            std::vector<Ref *> r = {&op.a, &op.b, &op.w, &op.bias, &op.res, &op.scale, &op.w2, &op.bias2};
            for (auto &e : op.eb) r.push_back(&e);
            for (auto &e : op.x) r.push_back(&e);
            return r;
        };
        for (auto &op : plan_.ops) {
            if (op.kind != OpKind::GEMM || op.gemm2.N > 0 || op.gemm.fold || op.gemm.w3 || op.w.space != Space::CONSTS) continue;
            if (!gemm_b3_shape_ok(op.gemm) && !gemm_dma3_wanted(op.gemm)) continue;
            const int64_t N = op.gemm.N, K = op.gemm.K;
            if ((int64_t)plan_.consts[(size_t)op.w.id].size() < op.w.offset + N * K) continue;
            // BN_GEMM3: 0 = exact-f32 kernel, 1 = the LDS-DMA form everywhere, 2 (default) = the register-staged form wherever it applies
            const bool b3 = gemm_b3_shape_ok(op.gemm);
            std::vector<float> img = b3 ? pack_w3f(plan_.consts[(size_t)op.w.id].data() + op.w.offset, N, K) : pack_w3(plan_.consts[(size_t)op.w.id].data() + op.w.offset, N, K);
            int users = 0;
            for (auto &o2 : plan_.ops)
                for (Ref *r : refs_of(o2)) users += r->space == Space::CONSTS && r->id == op.w.id;
            if (users == 1 && op.w.offset == 0) plan_.consts[(size_t)op.w.id] = std::move(img);
            else op.w = Ref{Space::CONSTS, add_const(img), 0};
            op.gemm.w3 = b3 ? 2 : 1;
            op.weight_bytes += 2.0 * (double)N * (double)((K + 31) / 32 * 32) + 4.0 * (double)N * (double)((K + 31) / 32 * 32 - K);  // 6 bytes per (padded) weight instead of 4
        }
    }

    void plan_memory() {
        // greedy first-fit over the linear launch order; storages are per-sample extents,
        // rounded to 64 elements so that (offset * max_batch) stays 256-byte aligned.
        struct Block { int64_t off, size; };
        std::vector<Block> free_list;
        int64_t top = 0;
        auto alloc = [&](int64_t size) {
            int best = -1;
            for (size_t k = 0; k < free_list.size(); k++)
                if (free_list[k].size >= size && (best < 0 || free_list[k].size < free_list[best].size)) best = (int)k;
            if (best >= 0) {
                int64_t off = free_list[best].off;
                if (free_list[best].size == size) free_list.erase(free_list.begin() + best);
                else { free_list[best].off += size; free_list[best].size -= size; }
                return off;
            }
            int64_t off = top;
            top += size;
            return off;
        };
        auto release = [&](int64_t off, int64_t size) {
            free_list.push_back({off, size});
            std::sort(free_list.begin(), free_list.end(), [](const Block &a, const Block &b) { return a.off < b.off; });
            for (size_t k = 0; k + 1 < free_list.size();) {
                if (free_list[k].off + free_list[k].size == free_list[k + 1].off) { free_list[k].size += free_list[k + 1].size; free_list.erase(free_list.begin() + k + 1); }
                else k++;
            }
            // give the tail back
            if (!free_list.empty() && free_list.back().off + free_list.back().size == top) { top = free_list.back().off; free_list.pop_back(); }
        };
        auto rounded = [](int64_t e) { return (e + 63) / 64 * 64; };
        int nops = (int)plan_.ops.size();
        std::vector<std::vector<int>> starts(nops), ends(nops);
        for (size_t s = 0; s < plan_.storages.size(); s++) {
            auto &st = plan_.storages[s];
            if (st.first < 0) continue;  // never touched
            starts[st.first].push_back((int)s);
            if (!st.pinned) ends[st.last].push_back((int)s);
        }
        int64_t peak = 0;
        for (size_t s = 0; s < plan_.storages.size(); s++)
            if (plan_.storages[s].persistent && plan_.storages[s].first >= 0) plan_.storages[s].arena_off = alloc(rounded(plan_.storages[s].elems));
        for (int k = 0; k < nops; k++) {
            for (int s : starts[k])
                if (!plan_.storages[s].persistent) plan_.storages[s].arena_off = alloc(rounded(plan_.storages[s].elems));
            peak = std::max(peak, top);
            for (int s : ends[k]) release(plan_.storages[s].arena_off, rounded(plan_.storages[s].elems));
        }
        plan_.arena_elems = std::max<int64_t>(peak, 64);
        // constants arena
        int64_t coff = 0;
        plan_.const_off.resize(plan_.consts.size());
        for (size_t k = 0; k < plan_.consts.size(); k++) { plan_.const_off[k] = coff; coff += rounded((int64_t)plan_.consts[k].size()); }
        plan_.consts_elems = std::max<int64_t>(coff, 64);
    }
};

}  // namespace

bool op_type_mapped(const std::string &t) {
    // the operator types Builder::lower / try_fold dispatch on (keep in step with them: tests/test_host_logic.py plans one node of every
    // type listed here and expects no "outside the native subset" refusal)
    static const std::set<std::string> known = {
        "Constant", "ConstantOfShape", "Range", "Shape", "Identity", "Dropout", "Cast", "Transpose", "Reshape", "Flatten", "Squeeze", "Unsqueeze",
        "Slice", "Concat", "Pad", "Softmax", "LogSoftmax", "Split", "MaxPool", "AveragePool", "Conv", "MatMul", "Gemm", "BatchNormalization",
        "Add", "Sub", "Mul", "Div", "Pow", "Max", "Min", "GlobalAveragePool", "GlobalMaxPool", "ReduceMean", "ReduceSum", "ReduceMax", "ReduceMin",
        "ReduceProd", "ReduceL2", "ReduceSumSquare", "STFT", "DFT", "Expand", "Tile", "PRelu", "InstanceNormalization", "Relu", "Sigmoid", "Tanh",
        "Exp", "Log", "Sqrt", "Abs", "Neg", "Reciprocal", "Floor", "Ceil", "Erf", "Softplus", "HardSwish", "HardSigmoid", "LeakyRelu", "Clip",
        "Greater", "Less", "GreaterOrEqual", "LessOrEqual", "Equal", "Not", "And", "Or", "Xor", "Where", "Gather", "Elu", "Selu", "Celu",
        "ThresholdedRelu", "Softsign", "Mish", "Gelu", "Sign", "Round", "Sum", "Mean", "Size", "ReduceL1", "ReduceLogSum", "ReduceLogSumExp",
        "LayerNormalization"};
    return known.count(t) != 0;
}

IoMeta read_io_meta(const OnnxModel &m) {
    IoMeta io;
    if (!m.inputs.empty()) {
        io.input_name = m.inputs[0].name;
        io.input_shape = m.inputs[0].shape;
        if (!io.input_shape.empty() && io.input_shape[0] <= 0) io.input_shape[0] = -1;
    }
    for (auto &o : m.outputs) {
        io.output_names.push_back(o.name);
        io.output_shapes.push_back(o.shape);
    }
    return io;
}

std::unique_ptr<Plan> build_plan(const OnnxModel &m, const std::vector<int> &wanted_outputs) {
    auto p = std::make_unique<Plan>();
    Builder b(m, *p);
    b.run(wanted_outputs);
    return p;
}

}  // namespace bn
