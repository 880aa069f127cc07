// C ABI of libbirdnet_hip.so (include/birdnet_hip.h).
#include "../../include/birdnet_hip.h"

#include <hip/hip_runtime.h>

#include "hip_gate.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "detect.h"
#include "engine.h"
#include "kernels.h"
#include "plan_rules.h"

using namespace bn;

namespace {

thread_local std::string g_err;

bn_status fail(bn_status st, const std::string &msg) {
    g_err = msg;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(BN_ERR_BACKEND, std::string(#expr) + " failed: " + hipGetErrorString(e_)); \
    } while (0)

struct PlanDev {
    std::unique_ptr<Plan> plan;
    float *d_consts = nullptr;
    ~PlanDev() {
        if (d_consts) (void)gated::Free(d_consts);
    }
};

bool device_is_gfx950(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

// ---- persistent host staging pool ----
// The reference's prepare_input copies every caller slice into one contiguous buffer on the calling thread
// (batch_context.rs:199-211).  At 10 000+ segments per second that copy is 6+ GB/s -- more than one core moves -- so
// the copies of a batch are handed to a small pool of long-lived threads (one task per segment) while the calling
// thread copies too and puts each finished chunk on the wire.  Started on first use, joined at library unload; no
// thread is created per call.  BN_STAGE_THREADS overrides the worker count (0 = the caller copies alone).
struct StageTask {
    const void *src;
    void *dst;
    size_t bytes;
    std::atomic<uint32_t> *done;  // incremented (release) when the copy has landed
};
class StagePool {
   public:
    static StagePool &get() {
        static StagePool pool;
        return pool;
    }
    void push(const StageTask *tasks, size_t n) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t k = 0; k < n; k++) q_.push_back(tasks[k]);
        }
        if (n > 1) cv_.notify_all();
        else cv_.notify_one();
    }
    // the caller helps: run one queued task if there is one
    bool try_run_one() {
        StageTask t;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (q_.empty()) return false;
            t = q_.front();
            q_.pop_front();
        }
        run(t);
        return true;
    }
    size_t workers() const { return th_.size(); }

   private:
    StagePool() {
        unsigned hw = std::thread::hardware_concurrency();
        long n = hw > 2 ? std::min<long>(6, (long)hw / 2) : 0;
        if (const char *e = getenv("BN_STAGE_THREADS")) n = std::max<long>(0, std::min<long>(64, atol(e)));
        for (long k = 0; k < n; k++) th_.emplace_back([this] { loop(); });
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    static void run(const StageTask &t) {
        memcpy(t.dst, t.src, t.bytes);
        t.done->fetch_add(1, std::memory_order_release);
    }
    void loop() {
        for (;;) {
            StageTask t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;  // stop_
                t = q_.front();
                q_.pop_front();
            }
            run(t);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<StageTask> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

}  // namespace

struct bn_model {
    // one reference for the handle bn_model_load returned + one per context created from it: a context may outlive
    // the caller's bn_model_free (the reference's BatchInferenceContext is an owned value with no lifetime tie to
    // the Classifier, src/batch_context.rs:70-85), the model goes when the last holder does
    std::atomic<int> refs{1};
    int device = 0;
    OnnxModel onnx;  // kept for the lazily built all-outputs plan
    bn_model_config cfg{};
    IoMeta io;
    std::unique_ptr<PlanDev> main_plan;  // computes logits (+ embeddings) only
    std::unique_ptr<PlanDev> full_plan;  // every graph output
    std::mutex mu;
};

struct bn_ctx {
    bn_model *model = nullptr;
    PlanDev *pd = nullptr;
    size_t max_batch = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    float *d_arena = nullptr;
    float *d_input = nullptr;
    float *h_out = nullptr;    // pinned staging for logits + embeddings
    size_t h_out_elems = 0;
    size_t device_bytes = 0;
    bool holds_model = false;  // counted in model->refs (set once creation succeeded)
    bool counted = false;      // counted in the device's live-context count (kernels.h device_context_count)
    bool in_flight = false;  // a cancelled/timed-out run may still be executing
    size_t last_batch = 0;
    // top-K scratch
    uint32_t *d_tk_idx = nullptr, *d_tk_cnt = nullptr, *d_tk_flags = nullptr;
    float *d_tk_conf = nullptr;
    size_t tk_cap = 0;  // elements of idx/conf
    // pinned mirrors for bn_step_device
    // bn_step_device: top-K rows of a step live packed [idx: batch*k][conf: batch*k][count: batch] in one device block
    // and one pinned host block, so they cross the bus as ONE copy (each async copy is a ~7 us blit launch)
    uint32_t *d_step = nullptr, *h_step = nullptr;
    size_t step_cap = 0;  // words
    uint32_t *h_tk_idx = nullptr, *h_tk_cnt = nullptr;  // views into h_step for the last step
    float *h_tk_conf = nullptr;
    size_t step_k = 0;
    // ---- asynchronous host-slice path (bn_infer_submit / bn_infer_collect): a ring of two batches per context.
    // Both slots own their device input, pinned input and pinned output buffers (allocated on first use), so a ticket
    // in flight shares nothing with the synchronous entry points (bn_infer_windows / bn_step_*) but the arena and the
    // top-K block, and those are ordered by the context's stream.
    struct HostSlot {
        float *d_input = nullptr, *h_input = nullptr, *h_out = nullptr;
        uint32_t *h_tk = nullptr;  // pinned [idx: batch*k][conf: batch*k][count: batch]
        size_t tk_cap = 0;         // words
        hipEvent_t h2d_done = nullptr, plan_done = nullptr, out_done = nullptr;
        bool owned = false, used = false, busy = false;
        uint64_t ticket = 0;
        size_t batch = 0, k = 0;
    };
    HostSlot slots[2];
    hipStream_t copy_stream = nullptr;
    uint64_t next_ticket = 1;
    struct GraphKey {
        size_t batch;
        const float *in;
        bool operator<(const GraphKey &o) const { return batch != o.batch ? batch < o.batch : in < o.in; }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
    std::map<GraphKey, uint64_t> graph_used;  // replay counter value at each graph's last launch (least recently used goes first)
    uint64_t graph_tick = 0;
    // bn_ctx_get_stats: how the plan reached the stream
    uint64_t n_captures = 0, n_instantiates = 0, n_replays = 0, n_eager_runs = 0, n_capture_fallbacks = 0, n_evictions = 0, n_input_copies = 0;
    std::string last_fallback;  // why the most recent capture did not become a graph
};

namespace {

float *resolve(const bn_ctx *c, const Ref &r, const float *d_in) {
    const Plan &p = *c->pd->plan;
    switch (r.space) {
        case Space::INPUT: return const_cast<float *>(d_in) + r.offset;
        case Space::ARENA: return c->d_arena + p.storages[r.id].arena_off * (int64_t)c->max_batch + r.offset;
        case Space::CONSTS: return c->pd->d_consts + p.const_off[r.id] + r.offset;
        default: return nullptr;
    }
}

SeTail se_tail_of(const bn_ctx *c, const PlanOp &op, const float *d_in) {
    SeTail t{};
    if (!op.se_fused) return t;
    t.on = 1;
    t.se = op.se;
    t.w1 = resolve(c, op.x[0], d_in);
    t.b1 = resolve(c, op.x[1], d_in);
    t.w2t = resolve(c, op.x[2], d_in);
    t.b2 = resolve(c, op.x[3], d_in);
    t.gate = resolve(c, op.res, d_in);
    t.counter = reinterpret_cast<uint32_t *>(resolve(c, op.scale, d_in));
    t.cnt_bs = c->pd->plan->storages[op.scale.id].elems;
    return t;
}

void launch_op(const bn_ctx *c, const PlanOp &op, const float *d_in, int64_t batch) {
    float *out = resolve(c, op.out, d_in);
    const float *a = resolve(c, op.a, d_in);
    switch (op.kind) {
        case OpKind::ELT: {
            const float *eb[ELT_MAX_STAGES];
            for (int k = 0; k < ELT_MAX_STAGES; k++) eb[k] = resolve(c, op.eb[k], d_in);
            launch_eltwise(c->stream, op.elt, out, a, eb, batch);
            break;
        }
        case OpKind::REDUCE: launch_reduce(c->stream, op.red, out, a, batch, op.red.pair ? resolve(c, op.b, d_in) : nullptr); break;
        case OpKind::GEMM:
            if (op.se_fused) {  // squeeze-excite in the GEMM's prologue: only the LDS-DMA kernel carries it (planner rule I)
                SeInline si{};
                si.se = op.se;
                si.partial = resolve(c, op.b, d_in);
                si.w1 = resolve(c, op.x[0], d_in);
                si.b1 = resolve(c, op.x[1], d_in);
                si.w2t = resolve(c, op.x[2], d_in);
                si.b2 = resolve(c, op.x[3], d_in);
                if (!(op.gemm.w3 ? launch_gemm_dma3 : launch_gemm_dma)(c->stream, op.gemm, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.res, d_in),
                                     resolve(c, op.scale, d_in), batch, &si))
                    launch_error("GEMM with inline squeeze-excite: operands are not 16-byte aligned");
                break;
            }
            if (op.gemm2.N > 0) {  // planner rule J: the product over this GEMM's rows runs behind it in the same launch
                launch_gemm_fold_pair(c->stream, op.gemm, op.gemm2, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.w2, d_in),
                                      resolve(c, op.bias2, d_in), batch);
                break;
            }
            {
                FramePre pre = op.pre;  // the signal chain a framing GEMM applies while it loads its span (planner rule G): operands in eb
                for (int k = 0; k < ELT_MAX_STAGES; k++) pre.sc[k] = k < pre.n ? resolve(c, op.eb[k], d_in) : nullptr;
                if (op.gemm.fold == 2) {  // quarter-folded cosine bank: window tables and column map ride in w2 / bias2
                    launch_gemm_fold2(c->stream, op.gemm, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.w2, d_in),
                                      reinterpret_cast<const int32_t *>(resolve(c, op.bias2, d_in)), batch, &pre);
                    break;
                }
                launch_gemm(c->stream, op.gemm, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.res, d_in),
                            resolve(c, op.scale, d_in), batch, &pre);
            }
            break;
        case OpKind::CONV:
            launch_conv(c->stream, op.conv, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.res, d_in), batch);
            break;
        case OpKind::DWCONV: {
            SeTail tail = se_tail_of(c, op, d_in);
            launch_dwconv(c->stream, op.dw, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.b, d_in), batch, tail.on ? &tail : nullptr);
            break;
        }
        case OpKind::GAP: launch_gap_partial(c->stream, op.gap, out, a, batch); break;
        case OpKind::POOL: launch_pool(c->stream, op.pool, out, a, batch); break;
        case OpKind::MBCONV:
        {
            SeTail tail = se_tail_of(c, op, d_in);
            launch_mbconv(c->stream, op.mb, out, a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.w2, d_in),
                          resolve(c, op.bias2, d_in), resolve(c, op.b, d_in), batch, tail.on ? &tail : nullptr);
            break;
        }
        case OpKind::FFT: {
            StftPtrs sp{};
            sp.out = out;
            sp.in = a;
            sp.window = resolve(c, op.w, d_in);
            sp.tw = reinterpret_cast<const float2 *>(resolve(c, op.w2, d_in));
            sp.otab = resolve(c, op.bias2, d_in);
            sp.bias = resolve(c, op.bias, d_in);
            for (int k = 0; k < ELT_MAX_STAGES; k++) sp.pre[k] = resolve(c, op.eb[k], d_in);
            sp.mstart = resolve(c, op.x[0], d_in);
            sp.mcol = resolve(c, op.x[1], d_in);
            sp.mval = resolve(c, op.x[2], d_in);
            sp.mel_bias = resolve(c, op.x[3], d_in);
            launch_stft(c->stream, op.fft, sp, batch);
            break;
        }
        case OpKind::SEFC:
            launch_se_fc(c->stream, op.se, out, resolve(c, op.b, d_in), a, resolve(c, op.w, d_in), resolve(c, op.bias, d_in), resolve(c, op.w2, d_in),
                         resolve(c, op.bias2, d_in), batch);
            break;
    }
}

// Enqueue the whole plan for `batch` segments on the context's stream.
bn_status enqueue_plan(bn_ctx *c, const float *d_in, size_t batch, const volatile int32_t *cancel) {
    const Plan &p = *c->pd->plan;
    bool use_graph = !(c->flags & BN_CTX_NO_GRAPH);
    // The plan always reads the batch from the CONTEXT'S OWN input buffer: a captured graph then depends on the batch
    // size alone (one capture + instantiate per batch size, whatever buffers the caller cycles through), and a caller
    // that wants no copy at all writes its batch into that buffer (bn_ctx_input_device).  Any other device pointer is
    // copied in on the context's stream, ahead of the plan: 18 MB at batch 32, about 10 us.
    if (d_in != c->d_input) {
        static const bool runtime_copy = getenv("BN_INPUT_MEMCPY") && atoi(getenv("BN_INPUT_MEMCPY")) != 0;
        if (runtime_copy) HIP_TRY(hipMemcpyAsync(c->d_input, d_in, batch * (size_t)p.sample_count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        else {
            launch_copy_dev(c->stream, c->d_input, d_in, batch * (size_t)p.sample_count * sizeof(float));
            HIP_TRY(hipGetLastError());
        }
        c->n_input_copies++;
        d_in = c->d_input;
    }
    if (use_graph) {
        // (the grids of some launches depend on whether the device is shared, kernels.h: one graph per form)
        bn_ctx::GraphKey key{batch, bn::device_context_count() > 1 ? reinterpret_cast<const float *>(uintptr_t(1)) : nullptr};
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            hipGraph_t g = nullptr;
            // What sits between Begin and End is kernel launches on this stream and nothing else: the LDS opt-ins and
            // device queries the launchers used to make on first use (hipFuncSetAttribute on the first > 64 KB launch of
            // a plan -- the op the one-in-six invalidated capture of round 2 named) happen in prepare_device() at
            // context creation.  Captures of one process are still serialised (they are rare -- one per context and
            // batch size -- and a capture is not the place to find out what two of them do to each other), in the
            // thread-local mode so that what OTHER threads do to the device meanwhile (uploads, allocations of sibling
            // ranks) is none of this capture's business.
            std::unique_lock<std::shared_mutex> capture_lock(capture_gate());  // hip_gate.h: no allocation / free / synchronous copy of the library overlaps a capture
            (void)hipGetLastError();  // drop stale sticky errors of unrelated earlier calls
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            c->n_captures++;
            hipError_t le = hipSuccess;
            std::string bad;
            (void)take_launch_error();
            std::string refused;
            for (auto &op : p.ops) {
                launch_op(c, op, d_in, (int64_t)batch);
                if (le == hipSuccess && (le = hipGetLastError()) != hipSuccess) bad = op.name;
                if (refused.empty())
                    if (const char *why = take_launch_error()) refused = "launch of '" + op.name + "' refused: " + why;
            }
            hipError_t e = hipStreamEndCapture(c->stream, &g);
            if (!refused.empty()) {
                if (g) (void)hipGraphDestroy(g);
                return fail(BN_ERR_INVALID_ARG, refused);
            }
            hipGraphExec_t ge = nullptr;
            hipError_t ie = hipSuccess;
            if (le == hipSuccess && e == hipSuccess) {
                ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                c->n_instantiates++;
                if (ie != hipSuccess) ge = nullptr;
            }
            if (g) (void)hipGraphDestroy(g);
            if (!ge) {
                // The capture did not become a graph.  Nothing was enqueued, so the batch can still run launch by launch
                // below -- but never silently: the event is counted (bn_ctx_get_stats.capture_fallbacks), its cause kept
                // (.last_fallback) and reported on stderr once per context; BN_STRICT_GRAPH=1 turns it into an error.
                (void)hipGetLastError();
                c->n_capture_fallbacks++;
                c->last_fallback = le != hipSuccess ? "launch of '" + bad + "' failed during capture: " + hipGetErrorString(le)
                                   : e != hipSuccess ? std::string("hipStreamEndCapture: ") + hipGetErrorString(e)
                                                     : std::string("hipGraphInstantiate: ") + hipGetErrorString(ie);
                if (c->n_capture_fallbacks == 1)
                    fprintf(stderr, "libbirdnet_hip: graph capture for batch %zu failed (%s); running the plan launch by launch\n", batch,
                            c->last_fallback.c_str());
                static const bool strict = getenv("BN_STRICT_GRAPH") && atoi(getenv("BN_STRICT_GRAPH")) != 0;
                if (strict) return fail(BN_ERR_BACKEND, "graph capture failed: " + c->last_fallback);
                use_graph = false;
            } else {
                if (c->graphs.size() >= 16) {
                    // bounded cache: the least recently replayed graph goes (a caller cycling through up to 16 input
                    // buffers or batch sizes never re-captures; the whole cache used to be dropped at once); nothing of
                    // this context may still be replaying it
                    auto victim = c->graph_used.begin();
                    for (auto u = c->graph_used.begin(); u != c->graph_used.end(); ++u)
                        if (u->second < victim->second) victim = u;
                    (void)hipStreamSynchronize(c->stream);
                    auto g_old = c->graphs.find(victim->first);
                    if (g_old != c->graphs.end()) {
                        (void)hipGraphExecDestroy(g_old->second);  // (the gate is held exclusively here)
                        c->graphs.erase(g_old);
                    }
                    c->graph_used.erase(victim);
                    c->n_evictions++;
                }
                it = c->graphs.emplace(key, ge).first;
            }
        }
        if (use_graph) {
            c->graph_used[it->first] = ++c->graph_tick;
            HIP_TRY(hipGraphLaunch(it->second, c->stream));
            c->n_replays++;
        }
    }
    if (!use_graph) {
        (void)hipGetLastError();  // drop stale sticky errors of unrelated earlier calls
        (void)take_launch_error();
        c->n_eager_runs++;
        for (auto &op : p.ops) {
            if (cancel && *cancel) return fail(BN_ERR_CANCELLED, "inference was cancelled");
            launch_op(c, op, d_in, (int64_t)batch);
            if (const char *why = take_launch_error()) return fail(BN_ERR_INVALID_ARG, "launch of '" + op.name + "' refused: " + why);
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(BN_ERR_BACKEND, "launch of '" + op.name + "' failed: " + hipGetErrorString(le));
        }
    }
    return BN_OK;
}

bn_status drain_if_needed(bn_ctx *c) {
    if (c->in_flight) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->in_flight = false;
    }
    return BN_OK;
}

// Wait for the stream while honouring cancel / deadline (classifier.rs:527-554 polls every
// 10 ms; here the stream is polled, so a finished batch is never reported as timed out).
bn_status wait_stream(bn_ctx *c, const volatile int32_t *cancel, uint64_t timeout_ns) {
    if (!cancel && timeout_ns == 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return BN_OK;
    }
    const auto start = std::chrono::steady_clock::now();
    int spins = 0;
    while (true) {
        hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return BN_OK;
        if (q != hipErrorNotReady) return fail(BN_ERR_BACKEND, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        if (cancel && *cancel) {
            c->in_flight = true;
            return fail(BN_ERR_CANCELLED, "inference was cancelled");
        }
        if (timeout_ns) {
            const uint64_t el = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - start).count();
            if (el >= timeout_ns) {
                c->in_flight = true;
                return fail(BN_ERR_TIMEOUT, "inference timed out after " + std::to_string(timeout_ns) + " ns");
            }
        }
        if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
    }
}

bn_status make_plan(bn_model *m, const std::vector<int> &wanted, std::unique_ptr<PlanDev> &out) {
    auto pd = std::make_unique<PlanDev>();
    try {
        pd->plan = build_plan(m->onnx, wanted);
    } catch (const UnsupportedModel &e) {
        return fail(BN_ERR_UNSUPPORTED_MODEL, e.what());
    } catch (const std::exception &e) {
        return fail(BN_ERR_MODEL_LOAD, e.what());
    }
    HIP_TRY(bn::use_device(m->device));
    if (!prepare_device(m->device)) return fail(BN_ERR_BACKEND, "device " + std::to_string(m->device) + " refused the kernels' dynamic-LDS opt-in");
    const Plan &p = *pd->plan;
    HIP_TRY(gated::Malloc(&pd->d_consts, (size_t)p.consts_elems * sizeof(float)));
    for (size_t k = 0; k < p.consts.size(); k++)
        if (!p.consts[k].empty())
            HIP_TRY(gated::Memcpy(pd->d_consts + p.const_off[k], p.consts[k].data(), p.consts[k].size() * sizeof(float), hipMemcpyHostToDevice));
    out = std::move(pd);
    return BN_OK;
}

bn_status finish_load(std::unique_ptr<bn_model> m, int32_t device, int32_t model_type_override, bn_model **out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(BN_ERR_NO_DEVICE, "no HIP device is visible; this path has no CPU fallback");
    if (device < 0 || device >= n) return fail(BN_ERR_NO_DEVICE, "device index " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)");
    if (!device_is_gfx950(device) && !getenv("BN_ALLOW_ANY_ARCH")) return fail(BN_ERR_NO_DEVICE, "device " + std::to_string(device) + " is not gfx950 (MI355X)");
    m->device = device;
    m->io = read_io_meta(m->onnx);
    if (m->onnx.inputs.size() != 1) return fail(BN_ERR_MODEL_DETECTION, "model has " + std::to_string(m->onnx.inputs.size()) + " inputs, expected 1");
    // Output shapes missing from the file: infer them by planning every output once.
    bool missing = false;
    for (auto &s : m->io.output_shapes) missing |= s.empty();
    if (missing) {
        std::vector<int> all;
        for (size_t k = 0; k < m->onnx.outputs.size(); k++) all.push_back((int)k);
        try {
            auto p = build_plan(m->onnx, all);
            for (size_t k = 0; k < p->outputs.size(); k++)
                if (m->io.output_shapes[k].empty()) {
                    m->io.output_shapes[k].push_back(-1);
                    for (auto d : p->outputs[k].dims) m->io.output_shapes[k].push_back(d);
                }
        } catch (const UnsupportedModel &e) {
            return fail(BN_ERR_UNSUPPORTED_MODEL, e.what());
        } catch (const std::exception &e) {
            return fail(BN_ERR_MODEL_LOAD, e.what());
        }
    }
    std::string reason;
    if (model_type_override == BN_MODEL_GENERIC) {
        // the range filter's meta model (rangefilter.rs:250-262): any single-input graph; output 0 is the result
        if (m->io.output_shapes.empty() || m->io.output_shapes[0].empty() || m->io.output_shapes[0].back() <= 0)
            return fail(BN_ERR_MODEL_DETECTION, "empty output shape");
        bn_model_config &c = m->cfg;
        memset(&c, 0, sizeof(c));
        c.model_type = BN_MODEL_GENERIC;
        c.sample_count = 1;
        for (size_t k = 1; k < m->io.input_shape.size(); k++) {
            if (m->io.input_shape[k] <= 0) return fail(BN_ERR_MODEL_DETECTION, "generic model needs a static per-row input shape");
            c.sample_count *= (uint64_t)m->io.input_shape[k];
        }
        c.num_species = (uint64_t)m->io.output_shapes[0].back();
        c.logits_output = 0;
        c.embedding_output = -1;
    } else if (!detect_model_type(m->io.input_shape, m->io.output_shapes, model_type_override, m->cfg, reason)) {
        return fail(BN_ERR_MODEL_DETECTION, reason);
    }
    std::vector<int> wanted{m->cfg.logits_output};
    if (m->cfg.embedding_output >= 0) wanted.push_back(m->cfg.embedding_output);
    bn_status st = make_plan(m.get(), wanted, m->main_plan);
    if (st != BN_OK) return st;
    if ((uint64_t)m->main_plan->plan->sample_count != m->cfg.sample_count) return fail(BN_ERR_MODEL_DETECTION, "input element count disagrees with the detected sample count");
    *out = m.release();
    return BN_OK;
}

}  // namespace

extern "C" {

int32_t bn_abi_version(void) { return BN_ABI_VERSION; }

int32_t bn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; d++) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

bn_status bn_model_load(const char *onnx_path, int32_t device, int32_t model_type_override, bn_model **out) {
    if (!onnx_path || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    auto m = std::make_unique<bn_model>();
    try {
        m->onnx = parse_onnx_file(onnx_path);
    } catch (const std::exception &e) {
        return fail(BN_ERR_MODEL_LOAD, e.what());
    }
    return finish_load(std::move(m), device, model_type_override, out);
}

bn_status bn_model_load_buffer(const void *bytes, size_t len, int32_t device, int32_t model_type_override, bn_model **out) {
    if (!bytes || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    auto m = std::make_unique<bn_model>();
    try {
        m->onnx = parse_onnx(static_cast<const uint8_t *>(bytes), len);
    } catch (const std::exception &e) {
        return fail(BN_ERR_MODEL_LOAD, e.what());
    }
    return finish_load(std::move(m), device, model_type_override, out);
}

static void model_unref(bn_model *m) {
    if (m && m->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        (void)bn::use_device(m->device);
        delete m;
    }
}
void bn_model_free(bn_model *m) { model_unref(m); }

int32_t bn_model_device(const bn_model *m) { return m ? m->device : -1; }

bn_status bn_model_io_info(const bn_model *m, bn_io_info *out) {
    if (!m || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    out->input_rank = (int32_t)std::min<size_t>(m->io.input_shape.size(), BN_MAX_RANK);
    for (int k = 0; k < out->input_rank; k++) out->input_shape[k] = m->io.input_shape[k];
    snprintf(out->input_name, BN_NAME_LEN, "%s", m->io.input_name.c_str());
    out->n_outputs = (int32_t)std::min<size_t>(m->io.output_names.size(), BN_MAX_OUTPUTS);
    for (int o = 0; o < out->n_outputs; o++) {
        out->output_rank[o] = (int32_t)std::min<size_t>(m->io.output_shapes[o].size(), BN_MAX_RANK);
        for (int k = 0; k < out->output_rank[o]; k++) out->output_shape[o][k] = m->io.output_shapes[o][k] <= 0 && k == 0 ? -1 : m->io.output_shapes[o][k];
        snprintf(out->output_name[o], BN_NAME_LEN, "%s", m->io.output_names[o].c_str());
    }
    return BN_OK;
}

bn_status bn_model_get_config(const bn_model *m, bn_model_config *out) {
    if (!m || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = m->cfg;
    return BN_OK;
}

bn_status bn_model_get_cost(const bn_model *m, bn_model_cost *out, size_t struct_size) {
    if (!m || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    const Plan &p = *m->main_plan->plan;
    bn_model_cost c{};
    c.macs_mfma = p.macs_mfma;
    c.macs_valu = p.macs_valu;
    c.weight_bytes = p.weight_bytes;
    c.activation_bytes = p.act_bytes;
    c.n_launches = (int32_t)p.ops.size();
    c.dft_gemm_macs = p.dft_gemm_macs;
    c.fft_flops = p.fft_flops;
    c.dft_performed_macs = p.dft_performed_macs;
    c.dft_fft_equiv_flops = p.dft_fft_equiv_flops;
    c.recompute_macs = p.recompute_macs;
    memcpy(out, &c, std::min(struct_size, sizeof(c)));  // never writes past the caller's struct (ABI 2: the struct grew in ABI 1 without a version bump)
    return BN_OK;
}

bn_status bn_detect_model_type(const int64_t *in_shape, size_t in_rank, const int64_t *out_shapes, const size_t *out_ranks, size_t n_out,
                               int32_t model_type_override, bn_model_config *out) {
    if (!in_shape || !out || (n_out && (!out_shapes || !out_ranks))) return fail(BN_ERR_INVALID_ARG, "null argument");
    std::vector<int64_t> in(in_shape, in_shape + in_rank);
    std::vector<std::vector<int64_t>> outs;
    size_t off = 0;
    for (size_t k = 0; k < n_out; k++) {
        outs.emplace_back(out_shapes + off, out_shapes + off + out_ranks[k]);
        off += out_ranks[k];
    }
    std::string reason;
    if (!detect_model_type(in, outs, model_type_override, *out, reason)) return fail(BN_ERR_MODEL_DETECTION, reason);
    return BN_OK;
}

static bn_status ensure_step_block(bn_ctx *c, size_t k);

void bn_set_sharing_mode(int32_t mode) { bn::device_sharing_mode(mode); }

bn_status bn_ctx_create(bn_model *m, size_t max_batch, uint32_t flags, bn_ctx **out) {
    if (!m || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (max_batch == 0 || max_batch > 65535) return fail(BN_ERR_INVALID_ARG, "max_batch must be in 1..65535");
    PlanDev *pd = m->main_plan.get();
    if (flags & BN_CTX_ALL_OUTPUTS) {
        std::lock_guard<std::mutex> lk(m->mu);
        if (!m->full_plan) {
            std::vector<int> all;
            for (size_t k = 0; k < m->onnx.outputs.size(); k++) all.push_back((int)k);
            bn_status st = make_plan(m, all, m->full_plan);
            if (st != BN_OK) return st;
        }
        pd = m->full_plan.get();
    }
    if (getenv("BN_NO_GRAPH")) flags |= BN_CTX_NO_GRAPH;
    auto c = std::make_unique<bn_ctx>();
    c->model = m;
    c->pd = pd;
    c->max_batch = max_batch;
    c->flags = flags;
    const Plan &p = *pd->plan;
    HIP_TRY(bn::use_device(m->device));
    if (!prepare_device(m->device)) return fail(BN_ERR_BACKEND, "device " + std::to_string(m->device) + " refused the kernels' dynamic-LDS opt-in");
    HIP_TRY(gated::StreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t arena_b = (size_t)p.arena_elems * max_batch * sizeof(float);
    const size_t in_b = (size_t)p.sample_count * max_batch * sizeof(float);
    HIP_TRY(gated::Malloc(&c->d_arena, arena_b));
    // the squeeze-excite ticket counters live here and must start at zero; on the context's OWN stream (a legacy-stream
    // hipMemset would serialise against -- and be seen by -- whatever sibling contexts are capturing or running)
    HIP_TRY(hipMemsetAsync(c->d_arena, 0, arena_b, c->stream));
    HIP_TRY(gated::Malloc(&c->d_input, in_b));
    {
        size_t row = (size_t)p.outputs[m->cfg.logits_output].row_elems;
        if (m->cfg.embedding_output >= 0) row += (size_t)p.outputs[m->cfg.embedding_output].row_elems;
        c->h_out_elems = row * max_batch;
    }
    HIP_TRY(gated::HostMalloc(&c->h_out, c->h_out_elems * sizeof(float), hipHostMallocDefault));
    c->device_bytes = arena_b + in_b;
    m->refs.fetch_add(1, std::memory_order_relaxed);
    c->holds_model = true;
    // top-K / step buffers for the usual k (<= 32) are part of the context from the start: the step path allocates
    // nothing (a larger top_k still grows them on first use, behind a stream drain)
    if (m->cfg.model_type != BN_MODEL_GENERIC) {
        const size_t n = (size_t)p.outputs[m->cfg.logits_output].row_elems;
        bn_status st = ensure_step_block(c.get(), std::min<size_t>(32, n));
        if (st != BN_OK) {
            bn_ctx_destroy(c.release());
            return st;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // arena zeroed before anyone captures or launches
    bn::device_context_count_add(m->device, 1);
    c->counted = true;
    *out = c.release();
    return BN_OK;
}

void bn_ctx_destroy(bn_ctx *c) {
    if (!c) return;
    if (c->counted) bn::device_context_count_add(c->model->device, -1);
    (void)bn::use_device(c->model->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->graphs) (void)gated::GraphExecDestroy(kv.second);
    if (c->d_arena) (void)gated::Free(c->d_arena);
    if (c->d_input) (void)gated::Free(c->d_input);
    if (c->h_out) (void)gated::HostFree(c->h_out);
    if (c->d_tk_idx) (void)gated::Free(c->d_tk_idx);
    if (c->d_tk_conf) (void)gated::Free(c->d_tk_conf);
    if (c->d_tk_cnt) (void)gated::Free(c->d_tk_cnt);
    if (c->d_tk_flags) (void)gated::Free(c->d_tk_flags);
    if (c->d_step) (void)gated::Free(c->d_step);
    if (c->h_step) (void)gated::HostFree(c->h_step);
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)gated::StreamDestroy(c->copy_stream);
    }
    for (auto &sl : c->slots) {
        if (sl.owned) {
            if (sl.d_input) (void)gated::Free(sl.d_input);
            if (sl.h_input) (void)gated::HostFree(sl.h_input);
            if (sl.h_out) (void)gated::HostFree(sl.h_out);
        }
        if (sl.h_tk) (void)gated::HostFree(sl.h_tk);
        if (sl.h2d_done) (void)gated::EventDestroy(sl.h2d_done);
        if (sl.plan_done) (void)gated::EventDestroy(sl.plan_done);
        if (sl.out_done) (void)gated::EventDestroy(sl.out_done);
    }
    if (c->stream) (void)gated::StreamDestroy(c->stream);
    bn_model *m = c->holds_model ? c->model : nullptr;
    delete c;
    model_unref(m);
}

size_t bn_ctx_max_batch(const bn_ctx *c) { return c ? c->max_batch : 0; }

bn_status bn_ctx_get_stats(const bn_ctx *c, bn_ctx_stats *out, size_t struct_size) {
    if (!c || !out) return fail(BN_ERR_INVALID_ARG, "null argument");
    bn_ctx_stats st{};
    st.captures = c->n_captures;
    st.instantiates = c->n_instantiates;
    st.replays = c->n_replays;
    st.eager_runs = c->n_eager_runs;
    st.capture_fallbacks = c->n_capture_fallbacks;
    st.evictions = c->n_evictions;
    st.cached_graphs = c->graphs.size();
    st.input_copies = c->n_input_copies;
    snprintf(st.last_fallback, sizeof(st.last_fallback), "%s", c->last_fallback.c_str());
    memcpy(out, &st, std::min(struct_size, sizeof(st)));  // a caller built against an older, shorter struct gets its prefix
    return BN_OK;
}
size_t bn_ctx_device_bytes(const bn_ctx *c) { return c ? c->device_bytes : 0; }
void *bn_ctx_stream(const bn_ctx *c) { return c ? (void *)c->stream : nullptr; }

bn_status bn_ctx_input_device(const bn_ctx *c, float **d_ptr, size_t *capacity_floats) {
    if (!c || !d_ptr) return fail(BN_ERR_INVALID_ARG, "null argument");
    *d_ptr = c->d_input;
    if (capacity_floats) *capacity_floats = (size_t)c->pd->plan->sample_count * c->max_batch;
    return BN_OK;
}

bn_status bn_ctx_synchronize(bn_ctx *c) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    HIP_TRY(bn::use_device(c->model->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->in_flight = false;
    return BN_OK;
}

bn_status bn_infer_device(bn_ctx *c, const float *d_pcm, size_t batch, int32_t sync) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    if (batch == 0) return BN_OK;
    if (!d_pcm) return fail(BN_ERR_INVALID_ARG, "null input");
    if (reinterpret_cast<uintptr_t>(d_pcm) & 15u) return fail(BN_ERR_INVALID_ARG, "device input must be 16-byte aligned");
    if (batch > c->max_batch) return fail(BN_ERR_INVALID_ARG, "batch size " + std::to_string(batch) + " exceeds context max " + std::to_string(c->max_batch));
    HIP_TRY(bn::use_device(c->model->device));
    bn_status st = drain_if_needed(c);
    if (st != BN_OK) return st;
    st = enqueue_plan(c, d_pcm, batch, nullptr);
    if (st != BN_OK) return st;
    c->last_batch = batch;
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return BN_OK;
}

// Device results -> pinned host buffers on the context's stream, by ONE kernel launch storing straight into the
// (device-mapped) pinned memory (topk.hip, copy_out_kernel) instead of one copy-engine transfer per region.
// BN_SDMA_COPY=1 restores the hipMemcpyAsync transfers (A/B measurements).
struct OutRegion {
    void *host;
    const void *dev;
    size_t bytes;
};
static bn_status results_to_host(bn_ctx *c, const OutRegion *regs, int n) {
    static const bool sdma = getenv("BN_SDMA_COPY") && atoi(getenv("BN_SDMA_COPY")) != 0;
    CopyOut co{};
    bool kernel_ok = !sdma && n <= 3;
    for (int r = 0; r < n && kernel_ok; r++) {
        void *dp = nullptr;
        if (regs[r].bytes % 4 || regs[r].bytes / 4 > 0xffffffffull || hipHostGetDevicePointer(&dp, regs[r].host, 0) != hipSuccess || !dp) {
            (void)hipGetLastError();
            kernel_ok = false;
            break;
        }
        co.dst[r] = dp;
        co.src[r] = regs[r].dev;
        co.words[r] = (uint32_t)(regs[r].bytes / 4);
    }
    if (kernel_ok) {
        co.n = n;
        launch_copy_out(c->stream, co);
        HIP_TRY(hipGetLastError());
        return BN_OK;
    }
    for (int r = 0; r < n; r++)
        if (regs[r].bytes) HIP_TRY(hipMemcpyAsync(regs[r].host, regs[r].dev, regs[r].bytes, hipMemcpyDeviceToHost, c->stream));
    return BN_OK;
}

// Tail shared by bn_infer_windows once the plan is enqueued on d_input: output copies to
// pinned staging, wait with cancel / timeout polling, copy out.
static bn_status finish_infer(bn_ctx *c, size_t batch, float *logits_out, float *emb_out, const volatile int32_t *cancel, uint64_t timeout_ns) {
    const Plan &p = *c->pd->plan;
    const bn_model_config &cfg = c->model->cfg;
    bn_status st = BN_OK;
    c->last_batch = batch;
    const OutputInfo &lo = p.outputs[cfg.logits_output];
    const size_t N = (size_t)lo.row_elems;
    float *h_logits = c->h_out;
    float *h_emb = c->h_out + N * c->max_batch;
    OutRegion regs[2] = {{h_logits, resolve(c, lo.ref, c->d_input), batch * N * sizeof(float)}, {nullptr, nullptr, 0}};
    int nreg = 1;
    size_t E = 0;
    if (emb_out && cfg.embedding_output >= 0) {
        const OutputInfo &eo = p.outputs[cfg.embedding_output];
        E = (size_t)eo.row_elems;
        regs[nreg++] = OutRegion{h_emb, resolve(c, eo.ref, c->d_input), batch * E * sizeof(float)};
    }
    st = results_to_host(c, regs, nreg);
    if (st != BN_OK) return st;
    st = wait_stream(c, cancel, timeout_ns);
    if (st != BN_OK) return st;
    memcpy(logits_out, h_logits, batch * N * sizeof(float));
    if (E) memcpy(emb_out, h_emb, batch * E * sizeof(float));
    return BN_OK;
}

// ---- asynchronous host-slice path ------------------------------------------------------------------------------
static bn_status ensure_step_block(bn_ctx *c, size_t k);

static bn_status ensure_slot(bn_ctx *c, bn_ctx::HostSlot &sl, int index) {
    const Plan &p = *c->pd->plan;
    if (!c->copy_stream) HIP_TRY(gated::StreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!sl.h2d_done) {
        HIP_TRY(gated::EventCreateWithFlags(&sl.h2d_done, hipEventDisableTiming));
        HIP_TRY(gated::EventCreateWithFlags(&sl.plan_done, hipEventDisableTiming));
        HIP_TRY(gated::EventCreateWithFlags(&sl.out_done, hipEventDisableTiming));
    }
    if (sl.d_input) return BN_OK;
    (void)index;
    const size_t in_b = (size_t)p.sample_count * c->max_batch * sizeof(float);
    sl.owned = true;
    HIP_TRY(gated::Malloc(&sl.d_input, in_b));
    HIP_TRY(gated::HostMalloc(&sl.h_input, in_b, hipHostMallocDefault));
    HIP_TRY(gated::HostMalloc(&sl.h_out, c->h_out_elems * sizeof(float), hipHostMallocDefault));
    c->device_bytes += in_b;
    return BN_OK;
}

bn_status bn_infer_submit(bn_ctx *c, const float *const *segs, size_t batch, size_t top_k, int32_t has_min, float min_conf, uint64_t *ticket) {
    if (!c || !ticket) return fail(BN_ERR_INVALID_ARG, "null argument");
    *ticket = 0;
    if (batch == 0) return BN_OK;  // nothing to run (classifier.rs:681-683); ticket 0 collects to nothing
    if (!segs) return fail(BN_ERR_INVALID_ARG, "null argument");
    if (batch > c->max_batch) return fail(BN_ERR_INVALID_ARG, "batch size " + std::to_string(batch) + " exceeds context max " + std::to_string(c->max_batch));
    for (size_t b = 0; b < batch; b++)
        if (!segs[b]) return fail(BN_ERR_INVALID_ARG, "segment " + std::to_string(b) + " is null");
    const Plan &p = *c->pd->plan;
    const bn_model_config &cfg = c->model->cfg;
    const OutputInfo &lo = p.outputs[cfg.logits_output];
    const size_t N = (size_t)lo.row_elems;
    const size_t k = std::min(top_k, N);
    if (k && topk_lds_bytes((int64_t)N, (int64_t)k) == 0) return fail(BN_ERR_INVALID_ARG, "top_k too large for the on-chip heap (k <= 9000)");
    HIP_TRY(bn::use_device(c->model->device));
    bn_ctx::HostSlot *slp = nullptr;
    int index = 0;
    for (int q = 0; q < 2 && !slp; q++) {
        const int cand = (int)((c->next_ticket + q) & 1u);
        if (!c->slots[cand].busy) { slp = &c->slots[cand]; index = cand; }
    }
    if (!slp) return fail(BN_ERR_INVALID_ARG, "two batches are already in flight on this context: collect one first");
    bn_ctx::HostSlot &sl = *slp;
    bn_status st = ensure_slot(c, sl, index);
    if (st != BN_OK) return st;
    if (c->in_flight) {  // a timed-out / cancelled batch may still be running: let everything settle first
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(c->copy_stream));
        c->in_flight = false;
    }
    if (k) {
        st = ensure_step_block(c, k);
        if (st != BN_OK) return st;
        const size_t need = c->max_batch * (2 * k + 1);
        if (need > sl.tk_cap) {
            if (sl.h_tk) (void)gated::HostFree(sl.h_tk);
            sl.h_tk = nullptr;
            sl.tk_cap = 0;
            HIP_TRY(gated::HostMalloc(&sl.h_tk, need * sizeof(uint32_t), hipHostMallocDefault));
            sl.tk_cap = need;
        }
    }
    const size_t S = (size_t)p.sample_count;
    if (sl.used) {
        HIP_TRY(hipEventSynchronize(sl.h2d_done));                      // the pinned buffer's last upload has left it
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, sl.plan_done, 0));  // the plan that read the device buffer is through
    }
    // stage + upload in chunks: pool threads (and this one) copy segment by segment; a chunk goes on the wire as soon
    // as its segments have landed in pinned memory, while the later chunks are still being copied
    {
        const size_t bytes_total = batch * S * sizeof(float);
        const size_t nchunk = bytes_total >= (2u << 20) ? std::min<size_t>(8, batch) : 1;
        StagePool &pool = StagePool::get();
        std::vector<StageTask> tasks(batch);
        std::atomic<uint32_t> done[8];
        size_t clo[9];
        for (size_t q = 0; q <= nchunk; q++) clo[q] = batch * q / nchunk;
        for (size_t q = 0; q < nchunk; q++) {
            done[q].store(0, std::memory_order_relaxed);
            for (size_t b = clo[q]; b < clo[q + 1]; b++) tasks[b] = StageTask{segs[b], sl.h_input + b * S, S * sizeof(float), &done[q]};
        }
        if (pool.workers() && batch > 1) pool.push(tasks.data(), batch);
        else
            for (size_t b = 0; b < batch; b++) {
                memcpy(tasks[b].dst, tasks[b].src, tasks[b].bytes);
                tasks[b].done->fetch_add(1, std::memory_order_release);
            }
        hipError_t e = hipSuccess;
        for (size_t q = 0; q < nchunk; q++) {
            const uint32_t want = (uint32_t)(clo[q + 1] - clo[q]);
            while (done[q].load(std::memory_order_acquire) < want)
                if (!pool.try_run_one()) std::this_thread::yield();
            if (e == hipSuccess && want)
                e = hipMemcpyAsync(sl.d_input + clo[q] * S, sl.h_input + clo[q] * S, (size_t)want * S * sizeof(float), hipMemcpyHostToDevice, c->copy_stream);
        }
        // (every task of this call has completed here: `tasks` and `done` may go out of scope)
        if (e != hipSuccess) return fail(BN_ERR_BACKEND, std::string("input upload failed: ") + hipGetErrorString(e));
    }
    sl.used = true;
    HIP_TRY(hipEventRecord(sl.h2d_done, c->copy_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, sl.h2d_done, 0));
    st = enqueue_plan(c, sl.d_input, batch, nullptr);
    if (st != BN_OK) {
        c->in_flight = true;
        return st;
    }
    HIP_TRY(hipEventRecord(sl.plan_done, c->stream));
    c->last_batch = batch;
    const float *d_logits = resolve(c, lo.ref, sl.d_input);
    if (k) {
        (void)hipGetLastError();
        uint32_t *d_idx = c->d_step, *d_cnt = c->d_step + 2 * batch * k;
        float *d_conf = reinterpret_cast<float *>(c->d_step + batch * k);
        launch_topk(c->stream, d_logits, (int64_t)batch, (int64_t)N, (int64_t)k, has_min, min_conf, (int64_t)k, d_idx, d_conf, d_cnt, c->d_tk_flags);
        if (const char *why = take_launch_error()) return fail(BN_ERR_INVALID_ARG, std::string("top-K launch refused: ") + why);
        HIP_TRY(hipGetLastError());
    }
    {
        OutRegion regs[3] = {{sl.h_out, d_logits, batch * N * sizeof(float)}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
        int nreg = 1;
        if (cfg.embedding_output >= 0) {
            const OutputInfo &eo = p.outputs[cfg.embedding_output];
            regs[nreg++] = OutRegion{sl.h_out + N * c->max_batch, resolve(c, eo.ref, sl.d_input), batch * (size_t)eo.row_elems * sizeof(float)};
        }
        if (k) regs[nreg++] = OutRegion{sl.h_tk, c->d_step, batch * (2 * k + 1) * sizeof(uint32_t)};
        st = results_to_host(c, regs, nreg);
        if (st != BN_OK) {
            c->in_flight = true;
            return st;
        }
    }
    HIP_TRY(hipEventRecord(sl.out_done, c->stream));
    sl.busy = true;
    sl.batch = batch;
    sl.k = k;
    sl.ticket = c->next_ticket++;
    *ticket = sl.ticket;
    return BN_OK;
}

bn_status bn_infer_collect(bn_ctx *c, uint64_t ticket, float *logits_out, float *emb_out, size_t k_stride, uint32_t *idx_out, float *conf_out,
                           uint32_t *count_out, const volatile int32_t *cancel, uint64_t timeout_ns) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    if (ticket == 0) return BN_OK;  // the empty batch
    bn_ctx::HostSlot *slp = nullptr;
    for (auto &q : c->slots)
        if (q.busy && q.ticket == ticket) slp = &q;
    if (!slp) return fail(BN_ERR_INVALID_ARG, "unknown or already collected ticket");
    bn_ctx::HostSlot &sl = *slp;
    // batches complete in submission order: collecting the younger one first simply waits for both
    HIP_TRY(bn::use_device(c->model->device));
    const auto start = std::chrono::steady_clock::now();
    int spins = 0;
    while (true) {
        hipError_t q = hipEventQuery(sl.out_done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            sl.busy = false;
            c->in_flight = true;
            return fail(BN_ERR_BACKEND, std::string("hipEventQuery: ") + hipGetErrorString(q));
        }
        if (cancel && *cancel) {
            sl.busy = false;
            c->in_flight = true;
            return fail(BN_ERR_CANCELLED, "inference was cancelled");
        }
        if (timeout_ns) {
            const uint64_t el = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - start).count();
            if (el >= timeout_ns) {
                sl.busy = false;
                c->in_flight = true;
                return fail(BN_ERR_TIMEOUT, "inference timed out after " + std::to_string(timeout_ns) + " ns");
            }
        }
        if (!cancel && !timeout_ns) {  // nothing to poll for: block
            hipError_t e = hipEventSynchronize(sl.out_done);
            if (e != hipSuccess) {
                sl.busy = false;
                c->in_flight = true;
                return fail(BN_ERR_BACKEND, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
            }
            break;
        }
        if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
    }
    const Plan &p = *c->pd->plan;
    const bn_model_config &cfg = c->model->cfg;
    const size_t N = (size_t)p.outputs[cfg.logits_output].row_elems;
    const size_t batch = sl.batch, k = sl.k;
    if (logits_out) memcpy(logits_out, sl.h_out, batch * N * sizeof(float));
    if (emb_out && cfg.embedding_output >= 0) {
        const size_t E = (size_t)p.outputs[cfg.embedding_output].row_elems;
        memcpy(emb_out, sl.h_out + N * c->max_batch, batch * E * sizeof(float));
    }
    if (count_out) {
        if (k == 0) {
            for (size_t r = 0; r < batch; r++) count_out[r] = 0;
        } else {
            if (!idx_out || !conf_out || k_stride < k) {
                sl.busy = false;
                return fail(BN_ERR_INVALID_ARG, "top-K outputs need idx / conf buffers with k_stride >= min(top_k, num_species)");
            }
            const uint32_t *h_idx = sl.h_tk, *h_cnt = sl.h_tk + 2 * batch * k;
            const float *h_conf = reinterpret_cast<const float *>(sl.h_tk + batch * k);
            for (size_t r = 0; r < batch; r++) {
                count_out[r] = h_cnt[r];
                for (size_t j = 0; j < h_cnt[r]; j++) {
                    idx_out[r * k_stride + j] = h_idx[r * k + j];
                    conf_out[r * k_stride + j] = h_conf[r * k + j];
                }
            }
        }
    }
    sl.busy = false;
    return BN_OK;
}

bn_status bn_infer(bn_ctx *c, const float *const *segs, size_t batch, float *logits_out, float *emb_out, const volatile int32_t *cancel,
                   uint64_t timeout_ns) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    if (batch == 0) return BN_OK;
    if (!segs || !logits_out) return fail(BN_ERR_INVALID_ARG, "null argument");
    if (cancel && *cancel) return fail(BN_ERR_CANCELLED, "inference was cancelled");
    uint64_t ticket = 0;
    bn_status st = bn_infer_submit(c, segs, batch, 0, 0, 0.0f, &ticket);
    if (st != BN_OK) return st;
    return bn_infer_collect(c, ticket, logits_out, emb_out, 0, nullptr, nullptr, nullptr, cancel, timeout_ns);
}

bn_status bn_ctx_output_device(const bn_ctx *c, int32_t index, const float **d_ptr, size_t *row_elems) {
    if (!c || !d_ptr || !row_elems) return fail(BN_ERR_INVALID_ARG, "null argument");
    const Plan &p = *c->pd->plan;
    if (index < 0 || index >= (int)p.outputs.size()) return fail(BN_ERR_INVALID_ARG, "output index out of range");
    if (!p.outputs[index].computed) return fail(BN_ERR_INVALID_ARG, "output " + std::to_string(index) + " is not computed by this context (create it with BN_CTX_ALL_OUTPUTS)");
    *d_ptr = resolve(c, p.outputs[index].ref, c->d_input);
    *row_elems = (size_t)p.outputs[index].row_elems;
    return BN_OK;
}

bn_status bn_ctx_read_output(bn_ctx *c, int32_t index, size_t batch, float *host_out) {
    const float *d = nullptr;
    size_t row = 0;
    bn_status st = bn_ctx_output_device(c, index, &d, &row);
    if (st != BN_OK) return st;
    if (!host_out || batch > c->max_batch) return fail(BN_ERR_INVALID_ARG, "bad host buffer / batch");
    HIP_TRY(bn::use_device(c->model->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->in_flight = false;
    HIP_TRY(gated::Memcpy(host_out, d, batch * row * sizeof(float), hipMemcpyDeviceToHost));
    return BN_OK;
}

size_t bn_ctx_time_kernels(bn_ctx *c, size_t batch, char (*names)[BN_NAME_LEN], float *usec, double *macs, double *bytes, size_t cap) {
    if (!c || batch == 0 || batch > c->max_batch) return 0;
    const Plan &p = *c->pd->plan;
    if (bn::use_device(c->model->device) != hipSuccess) return 0;
    (void)hipStreamSynchronize(c->stream);
    std::vector<hipEvent_t> ev(p.ops.size() + 1);
    for (auto &e : ev) (void)gated::EventCreate(&e);
    // warm (instruction caches, clocks) then measure
    for (auto &op : p.ops) launch_op(c, op, c->d_input, (int64_t)batch);
    (void)hipEventRecord(ev[0], c->stream);
    for (size_t k = 0; k < p.ops.size(); k++) {
        launch_op(c, p.ops[k], c->d_input, (int64_t)batch);
        (void)hipEventRecord(ev[k + 1], c->stream);
    }
    // calibration: the same event-to-event interval around an EMPTY kernel.  rocprofv3's kernel trace
    // reports 3.6 us for that empty kernel on MI355X (profiles/r01_v6_kernel_stats_1stream.csv,
    // null_kernel); the rest of its interval is command-processor and event overhead that every
    // interval above also contains, so it is subtracted -- the reported times are then on rocprofv3's
    // scale (family averages agree within a few percent).
    float overhead_us = 0.0f;
    {
        constexpr int NCAL = 16;
        hipEvent_t ce[NCAL + 1];
        for (auto &e : ce) (void)gated::EventCreate(&e);
        launch_null(c->stream);
        (void)hipEventRecord(ce[0], c->stream);
        for (int k = 0; k < NCAL; k++) {
            launch_null(c->stream);
            (void)hipEventRecord(ce[k + 1], c->stream);
        }
        (void)hipStreamSynchronize(c->stream);
        std::vector<float> iv;
        for (int k = 0; k < NCAL; k++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ce[k], ce[k + 1]);
            iv.push_back(ms * 1000.0f);
        }
        std::sort(iv.begin(), iv.end());
        overhead_us = std::max(0.0f, iv[NCAL / 2] - 3.6f);
        for (auto &e : ce) (void)gated::EventDestroy(e);
    }
    (void)hipStreamSynchronize(c->stream);
    for (size_t k = 0; k < p.ops.size() && k < cap; k++) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
        if (usec) usec[k] = std::max(0.5f, ms * 1000.0f - overhead_us);
        if (names) snprintf(names[k], BN_NAME_LEN, "%s", p.ops[k].name.c_str());
        // every multiply-add the launch PERFORMS: a fused MBConv launch carries its expand conv on the matrix cores beside the
        // depthwise taps (macs_mfma_extra, halo / band recompute included), a GEMM with the squeeze-excite products in its
        // prologue carries those on the vector ALU -- so that the launches sum to the plan's macs_mfma + macs_valu
        if (macs) macs[k] = (p.ops[k].macs + p.ops[k].macs_mfma_extra + p.ops[k].macs_valu_extra) * (double)batch;
        if (bytes) bytes[k] = p.ops[k].bytes * (double)batch + p.ops[k].weight_bytes;
    }
    for (auto &e : ev) (void)gated::EventDestroy(e);
    return p.ops.size();
}

size_t bn_ctx_launch_costs(const bn_ctx *c, size_t batch, double *macs_mfma, double *macs_valu, double *macs_recompute, double *bytes, size_t cap) {
    if (!c || batch == 0 || batch > c->max_batch) return 0;
    const Plan &p = *c->pd->plan;
    for (size_t k = 0; k < p.ops.size() && k < cap; k++) {
        const auto &op = p.ops[k];
        if (macs_mfma) macs_mfma[k] = ((op.mfma ? op.macs : 0.0) + op.macs_mfma_extra) * (double)batch;
        if (macs_valu) macs_valu[k] = ((op.mfma ? 0.0 : op.macs) + op.macs_valu_extra) * (double)batch;
        if (macs_recompute) macs_recompute[k] = op.macs_recompute * (double)batch;
        if (bytes) bytes[k] = op.bytes * (double)batch + op.weight_bytes;
    }
    return p.ops.size();
}

static bn_status topk_run(int device, hipStream_t stream, const float *d_logits, size_t rows, size_t n, size_t top_k, int32_t has_min,
                          float min_conf, size_t k_stride, uint32_t *d_idx, float *d_conf, uint32_t *d_cnt, uint32_t *d_flags,
                          uint32_t *idx_out, float *conf_out, uint32_t *count_out) {
    (void)device;
    const size_t k = std::min(top_k, n);
    if (k_stride < k) return fail(BN_ERR_INVALID_ARG, "k_stride smaller than min(top_k, n)");
    if (topk_lds_bytes((int64_t)n, (int64_t)k) == 0) return fail(BN_ERR_INVALID_ARG, "top_k too large for the on-chip heap (k <= 9000)");
    (void)hipGetLastError();
    launch_topk(stream, d_logits, (int64_t)rows, (int64_t)n, (int64_t)k, has_min, min_conf, (int64_t)k, d_idx, d_conf, d_cnt, d_flags);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> h_idx(rows * k), h_cnt(rows);
    std::vector<float> h_conf(rows * k);
    HIP_TRY(hipMemcpyAsync(h_cnt.data(), d_cnt, rows * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_idx.data(), d_idx, rows * k * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_conf.data(), d_conf, rows * k * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (size_t r = 0; r < rows; r++) {
        count_out[r] = h_cnt[r];
        for (size_t j = 0; j < h_cnt[r]; j++) {
            idx_out[r * k_stride + j] = h_idx[r * k + j];
            conf_out[r * k_stride + j] = h_conf[r * k + j];
        }
    }
    return BN_OK;
}

static bn_status ensure_topk_buffers(bn_ctx *c, size_t k);

bn_status bn_topk(bn_ctx *c, size_t batch, size_t top_k, int32_t has_min, float min_conf, size_t k_stride, uint32_t *idx_out, float *conf_out,
                  uint32_t *count_out) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    if (batch == 0) return BN_OK;
    if (!count_out || batch > c->max_batch) return fail(BN_ERR_INVALID_ARG, "bad arguments");
    const Plan &p = *c->pd->plan;
    const OutputInfo &lo = p.outputs[c->model->cfg.logits_output];
    const size_t n = (size_t)lo.row_elems;
    const size_t k = std::min(top_k, n);
    if (k == 0) {
        for (size_t r = 0; r < batch; r++) count_out[r] = 0;
        return BN_OK;
    }
    if (!idx_out || !conf_out) return fail(BN_ERR_INVALID_ARG, "null output");
    HIP_TRY(bn::use_device(c->model->device));
    {
        bn_status est = ensure_topk_buffers(c, k);
        if (est != BN_OK) return est;
    }
    return topk_run(c->model->device, c->stream, resolve(c, lo.ref, c->d_input), batch, n, top_k, has_min, min_conf, k_stride, c->d_tk_idx,
                    c->d_tk_conf, c->d_tk_cnt, c->d_tk_flags, idx_out, conf_out, count_out);
}

static bn_status ensure_topk_buffers(bn_ctx *c, size_t k) {
    const size_t need = c->max_batch * k;
    if (need > c->tk_cap) {
        if (c->d_tk_idx) (void)gated::Free(c->d_tk_idx);
        if (c->d_tk_conf) (void)gated::Free(c->d_tk_conf);
        c->d_tk_idx = nullptr;
        c->d_tk_conf = nullptr;
        HIP_TRY(gated::Malloc(&c->d_tk_idx, need * sizeof(uint32_t)));
        HIP_TRY(gated::Malloc(&c->d_tk_conf, need * sizeof(float)));
        c->tk_cap = need;
    }
    if (!c->d_tk_cnt) HIP_TRY(gated::Malloc(&c->d_tk_cnt, c->max_batch * sizeof(uint32_t)));
    if (!c->d_tk_flags) HIP_TRY(gated::Malloc(&c->d_tk_flags, c->max_batch * sizeof(uint32_t)));
    return BN_OK;
}

// device + pinned blocks of the packed top-K rows of one step (grown on demand; nothing of this context may be in
// flight when they grow: the stream is drained first)
static bn_status ensure_step_block(bn_ctx *c, size_t k) {
    bn_status st = ensure_topk_buffers(c, k);
    if (st != BN_OK) return st;
    const size_t need = c->max_batch * (2 * k + 1);
    if (need > c->step_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_step) (void)gated::Free(c->d_step);
        if (c->h_step) (void)gated::HostFree(c->h_step);
        c->d_step = c->h_step = nullptr;
        c->h_tk_idx = c->h_tk_cnt = nullptr;
        c->h_tk_conf = nullptr;
        c->step_cap = 0;
        HIP_TRY(gated::Malloc(&c->d_step, need * sizeof(uint32_t)));
        HIP_TRY(gated::HostMalloc(&c->h_step, need * sizeof(uint32_t), hipHostMallocDefault));
        c->step_cap = need;
    }
    return BN_OK;
}

bn_status bn_step_device(bn_ctx *c, const float *d_pcm, size_t batch, size_t top_k, int32_t has_min, float min_conf, int32_t sync) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    if (batch == 0) return BN_OK;
    if (!d_pcm || batch > c->max_batch) return fail(BN_ERR_INVALID_ARG, "bad input / batch size exceeds context max");
    if (reinterpret_cast<uintptr_t>(d_pcm) & 15u) return fail(BN_ERR_INVALID_ARG, "device input must be 16-byte aligned");
    const Plan &p = *c->pd->plan;
    const OutputInfo &lo = p.outputs[c->model->cfg.logits_output];
    const size_t n = (size_t)lo.row_elems;
    const size_t k = std::min(top_k, n);
    if (k == 0 || topk_lds_bytes((int64_t)n, (int64_t)k) == 0) return fail(BN_ERR_INVALID_ARG, "top_k must be in 1..9000");
    HIP_TRY(bn::use_device(c->model->device));
    bn_status st = drain_if_needed(c);
    if (st != BN_OK) return st;
    st = ensure_step_block(c, k);
    if (st != BN_OK) return st;
    st = enqueue_plan(c, d_pcm, batch, nullptr);
    if (st != BN_OK) return st;
    c->last_batch = batch;
    const float *d_logits = resolve(c, lo.ref, d_pcm);
    (void)hipGetLastError();
    uint32_t *d_idx = c->d_step, *d_cnt = c->d_step + 2 * batch * k;
    float *d_conf = reinterpret_cast<float *>(c->d_step + batch * k);
    launch_topk(c->stream, d_logits, (int64_t)batch, (int64_t)n, (int64_t)k, has_min, min_conf, (int64_t)k, d_idx, d_conf, d_cnt, c->d_tk_flags);
    HIP_TRY(hipGetLastError());
    {
        const OutRegion regs[2] = {{c->h_out, d_logits, batch * n * sizeof(float)}, {c->h_step, c->d_step, batch * (2 * k + 1) * sizeof(uint32_t)}};
        st = results_to_host(c, regs, 2);
        if (st != BN_OK) return st;
    }
    c->h_tk_idx = c->h_step;
    c->h_tk_conf = reinterpret_cast<float *>(c->h_step + batch * k);
    c->h_tk_cnt = c->h_step + 2 * batch * k;
    c->step_k = k;
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return BN_OK;
}

bn_status bn_step_results(const bn_ctx *c, const float **logits, const uint32_t **idx, const float **conf, const uint32_t **count, size_t *k_stride) {
    if (!c || !c->h_tk_idx) return fail(BN_ERR_INVALID_ARG, "no step has run on this context");
    if (logits) *logits = c->h_out;
    if (idx) *idx = c->h_tk_idx;
    if (conf) *conf = c->h_tk_conf;
    if (count) *count = c->h_tk_cnt;
    if (k_stride) *k_stride = c->step_k;
    return BN_OK;
}

bn_status bn_ctx_step_device_rows(const bn_ctx *c, const uint32_t **d_rows) {
    if (!c || !d_rows) return fail(BN_ERR_INVALID_ARG, "null argument");
    if (!c->d_step) return fail(BN_ERR_INVALID_ARG, "no step has run on this context");
    *d_rows = c->d_step;
    return BN_OK;
}

bn_status bn_topk_device(int32_t device, const float *d_logits, size_t rows, size_t n, size_t top_k, int32_t has_min, float min_conf,
                         size_t k_stride, uint32_t *idx_out, float *conf_out, uint32_t *count_out) {
    if (rows == 0) return BN_OK;
    if (!count_out) return fail(BN_ERR_INVALID_ARG, "null output");
    const size_t k = std::min(top_k, n);
    if (k == 0) {
        for (size_t r = 0; r < rows; r++) count_out[r] = 0;
        return BN_OK;
    }
    if (!d_logits || !idx_out || !conf_out) return fail(BN_ERR_INVALID_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BN_ERR_NO_DEVICE, "no HIP device is visible; this path has no CPU fallback");
    HIP_TRY(bn::use_device(device));
    if (!prepare_device(device)) return fail(BN_ERR_BACKEND, "device refused the kernels' dynamic-LDS opt-in");
    uint32_t *d_idx = nullptr, *d_cnt = nullptr, *d_flags = nullptr;
    float *d_conf = nullptr;
    HIP_TRY(gated::Malloc(&d_idx, rows * k * sizeof(uint32_t)));
    HIP_TRY(gated::Malloc(&d_conf, rows * k * sizeof(float)));
    HIP_TRY(gated::Malloc(&d_cnt, rows * sizeof(uint32_t)));
    HIP_TRY(gated::Malloc(&d_flags, rows * sizeof(uint32_t)));
    bn_status st = topk_run(device, nullptr, d_logits, rows, n, top_k, has_min, min_conf, k_stride, d_idx, d_conf, d_cnt, d_flags, idx_out, conf_out,
                            count_out);
    (void)gated::Free(d_idx);
    (void)gated::Free(d_conf);
    (void)gated::Free(d_cnt);
    (void)gated::Free(d_flags);
    return st;
}

bn_status bn_topk_host(int32_t device, const float *logits, size_t rows, size_t n, size_t top_k, int32_t has_min, float min_conf,
                       size_t k_stride, uint32_t *idx_out, float *conf_out, uint32_t *count_out) {
    if (rows == 0) return BN_OK;
    if (!count_out) return fail(BN_ERR_INVALID_ARG, "null output");
    if (n == 0 || top_k == 0) {
        for (size_t r = 0; r < rows; r++) count_out[r] = 0;
        return BN_OK;
    }
    if (!logits) return fail(BN_ERR_INVALID_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BN_ERR_NO_DEVICE, "no HIP device is visible; this path has no CPU fallback");
    HIP_TRY(bn::use_device(device));
    float *d = nullptr;
    HIP_TRY(gated::Malloc(&d, rows * n * sizeof(float)));
    hipError_t e = gated::Memcpy(d, logits, rows * n * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)gated::Free(d);
        return fail(BN_ERR_BACKEND, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    bn_status st = bn_topk_device(device, d, rows, n, top_k, has_min, min_conf, k_stride, idx_out, conf_out, count_out);
    (void)gated::Free(d);
    return st;
}

// ---- recording-level ingest (birdnet-analyze.rs:683-687, 707-743) ----
struct bn_recording {
    int32_t device = 0;
    int32_t format = BN_PCM_I16;
    void *d_pcm = nullptr;
    size_t n_samples = 0;
    // asynchronous upload (bn_recording_create_async): a thread of the recording's own copies the caller's buffer chunk by chunk
    // (synchronous copies under the capture gate, like every other copy of the library); chunks_done chunks are on the device
    size_t chunk_samples = 0;  // 0: the whole recording was uploaded at creation
    size_t n_chunks = 0;
    std::atomic<size_t> chunks_done{0};
    std::atomic<int> upload_error{0};
    std::thread uploader;
    std::mutex mu;
    std::condition_variable cv;
};

namespace {
// blocks until sample `last` of the recording is on the device (kernels launched afterwards may read it); false if the upload failed
bool recording_wait_samples(const bn_recording *rc, size_t last) {
    bn_recording *r = const_cast<bn_recording *>(rc);
    if (!r->chunk_samples || r->n_chunks == 0) return true;
    const size_t need = std::min(last / r->chunk_samples, r->n_chunks - 1) + 1;
    if (r->chunks_done.load(std::memory_order_acquire) >= need) return r->upload_error.load() == 0;
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv.wait(lk, [&] { return r->chunks_done.load(std::memory_order_acquire) >= need || r->upload_error.load() != 0; });
    return r->upload_error.load() == 0;
}
bool recording_wait_all(const bn_recording *r) { return recording_wait_samples(r, r->n_samples ? r->n_samples - 1 : 0); }
}  // namespace

bn_status bn_recording_create(int32_t device, const void *pcm, size_t n_samples, int32_t format, bn_recording **out) {
    if (!out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (format != BN_PCM_I16 && format != BN_PCM_F32) return fail(BN_ERR_INVALID_ARG, "unknown PCM format");
    if (n_samples && !pcm) return fail(BN_ERR_INVALID_ARG, "null PCM buffer");
    if (bn_device_count() <= 0) return fail(BN_ERR_NO_DEVICE, "no gfx950 device visible");
    HIP_TRY(bn::use_device(device));
    auto r = std::make_unique<bn_recording>();
    r->device = device;
    r->format = format;
    r->n_samples = n_samples;
    const size_t bytes = n_samples * (format == BN_PCM_I16 ? sizeof(int16_t) : sizeof(float));
    HIP_TRY(gated::Malloc(&r->d_pcm, std::max<size_t>(bytes, 16)));
    if (bytes) {
        hipError_t e = gated::Memcpy(r->d_pcm, pcm, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)gated::Free(r->d_pcm);
            return fail(BN_ERR_BACKEND, std::string("recording upload failed: ") + hipGetErrorString(e));
        }
    }
    *out = r.release();
    return BN_OK;
}

bn_status bn_recording_create_async(int32_t device, const void *pcm, size_t n_samples, int32_t format, bn_recording **out) {
    if (!out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (format != BN_PCM_I16 && format != BN_PCM_F32) return fail(BN_ERR_INVALID_ARG, "unknown PCM format");
    if (n_samples && !pcm) return fail(BN_ERR_INVALID_ARG, "null PCM buffer");
    if (bn_device_count() <= 0) return fail(BN_ERR_NO_DEVICE, "no gfx950 device visible");
    HIP_TRY(bn::use_device(device));
    const size_t esz = format == BN_PCM_I16 ? sizeof(int16_t) : sizeof(float);
    // the uploader holds the capture gate (shared) for one chunk's synchronous copy and a context's first-time graph capture takes it
    // exclusively: 4 MiB keeps a capture's wait behind a chunk well under a millisecond (32 MiB chunks stalled exactly the first
    // windows' captures the overlap was meant to help)
    const size_t chunk_mb = getenv("BN_UPLOAD_CHUNK_MB") ? (size_t)std::max(1, atoi(getenv("BN_UPLOAD_CHUNK_MB"))) : 4;
    auto r = std::make_unique<bn_recording>();
    r->device = device;
    r->format = format;
    r->n_samples = n_samples;
    HIP_TRY(gated::Malloc(&r->d_pcm, std::max<size_t>(n_samples * esz, 16)));
    r->chunk_samples = std::max<size_t>(1, (chunk_mb << 20) / esz);
    r->n_chunks = (n_samples + r->chunk_samples - 1) / r->chunk_samples;
    bn_recording *raw = r.get();
    if (r->n_chunks) try {
        r->uploader = std::thread([raw, pcm, esz]() {
            if (bn::use_device(raw->device) != hipSuccess) raw->upload_error.store(1);
            for (size_t k = 0; k < raw->n_chunks && raw->upload_error.load() == 0; k++) {
                const size_t a = k * raw->chunk_samples, n = std::min(raw->chunk_samples, raw->n_samples - a);
                // a synchronous copy of its own kind (null stream; the contexts' streams are non-blocking and do not wait for it)
                if (gated::Memcpy(static_cast<char *>(raw->d_pcm) + a * esz, static_cast<const char *>(pcm) + a * esz, n * esz, hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipGetLastError();
                    raw->upload_error.store(1);
                }
                {
                    std::lock_guard<std::mutex> lk(raw->mu);
                    raw->chunks_done.store(k + 1, std::memory_order_release);
                }
                raw->cv.notify_all();
            }
            if (raw->upload_error.load()) {  // wake every waiter
                std::lock_guard<std::mutex> lk(raw->mu);
                raw->chunks_done.store(raw->n_chunks, std::memory_order_release);
                raw->cv.notify_all();
            }
        });
    } catch (const std::exception &e) {  // std::system_error from the thread constructor must not cross the C boundary
        (void)gated::Free(r->d_pcm);
        r->d_pcm = nullptr;
        return fail(BN_ERR_BACKEND, std::string("could not start the upload thread: ") + e.what());
    }
    *out = r.release();
    return BN_OK;
}

bn_status bn_recording_wait(const bn_recording *r) {
    if (!r) return fail(BN_ERR_INVALID_ARG, "null recording");
    if (!recording_wait_all(r)) return fail(BN_ERR_BACKEND, "recording upload failed");
    return BN_OK;
}

// ---- polyphase resampler (no reference counterpart; design documented in include/birdnet_hip.h) ----
namespace {
double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 500; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}
struct ResampleTable {
    uint32_t L = 1, M = 1, T = 0;
    std::vector<float> coef;  // [L][T]
};
ResampleTable make_resample_table(uint32_t src_rate, uint32_t dst_rate, uint32_t zc) {
    ResampleTable t;
    if (zc == 0) zc = 16;
    uint32_t a = dst_rate, b = src_rate;
    while (b) { const uint32_t r = a % b; a = b; b = r; }
    t.L = dst_rate / a;
    t.M = src_rate / a;
    const double ratio = (double)t.L / (double)t.M;
    const double fc = 0.5 * std::min(1.0, ratio);        // cutoff in cycles per SOURCE sample
    const double half = (double)zc / (2.0 * fc);          // support half-width in source samples
    t.T = 2u * (uint32_t)std::ceil(half);                 // taps per phase (even)
    const double beta = 8.6, i0b = bessel_i0(beta), pi = 3.14159265358979323846;
    t.coef.assign((size_t)t.L * t.T, 0.0f);
    for (uint32_t p = 0; p < t.L; p++) {
        // output sample sits at source position base + p/L; tap j reads source index base + j - (T/2 - 1)
        std::vector<double> h(t.T);
        double sum = 0.0;
        for (uint32_t j = 0; j < t.T; j++) {
            const double tau = (double)p / (double)t.L - ((double)j - (double)(t.T / 2 - 1));  // output time - tap time
            const double u = tau / half;
            double w = 0.0;
            if (std::fabs(u) < 1.0) w = bessel_i0(beta * std::sqrt(1.0 - u * u)) / i0b;
            const double xarg = 2.0 * fc * tau;
            const double sinc = std::fabs(xarg) < 1e-12 ? 1.0 : std::sin(pi * xarg) / (pi * xarg);
            h[j] = 2.0 * fc * sinc * w;
            sum += h[j];
        }
        for (uint32_t j = 0; j < t.T; j++) t.coef[(size_t)p * t.T + j] = (float)(h[j] / sum);
    }
    return t;
}
}  // namespace

size_t bn_resample_table(uint32_t src_rate, uint32_t dst_rate, uint32_t zero_crossings, float *table, size_t cap, uint32_t *L_out, uint32_t *M_out,
                         uint32_t *T_out) {
    if (src_rate == 0 || dst_rate == 0) return 0;
    const ResampleTable t = make_resample_table(src_rate, dst_rate, zero_crossings);
    if (L_out) *L_out = t.L;
    if (M_out) *M_out = t.M;
    if (T_out) *T_out = t.T;
    if (table) memcpy(table, t.coef.data(), std::min(cap, t.coef.size()) * sizeof(float));
    return t.coef.size();
}

bn_status bn_recording_create_resampled(int32_t device, const void *pcm, size_t n_samples, int32_t format, uint32_t src_rate, uint32_t dst_rate,
                                        uint32_t zero_crossings, bn_recording **out) {
    if (!out) return fail(BN_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (src_rate == 0 || dst_rate == 0) return fail(BN_ERR_INVALID_ARG, "sample rates must be positive");
    if (src_rate == dst_rate) return bn_recording_create(device, pcm, n_samples, format, out);
    bn_recording *src = nullptr;
    bn_status st = bn_recording_create(device, pcm, n_samples, format, &src);
    if (st != BN_OK) return st;
    const ResampleTable t = make_resample_table(src_rate, dst_rate, zero_crossings);
    const size_t n_dst = (size_t)(((unsigned __int128)n_samples * t.L + t.M - 1) / t.M);
    auto r = std::make_unique<bn_recording>();
    r->device = device;
    r->format = BN_PCM_F32;
    r->n_samples = n_dst;
    float *d_table = nullptr;
    hipError_t e = gated::Malloc(&r->d_pcm, std::max<size_t>(n_dst * sizeof(float), 16));
    if (e == hipSuccess) e = gated::Malloc(reinterpret_cast<void **>(&d_table), t.coef.size() * sizeof(float));
    if (e == hipSuccess) e = gated::Memcpy(d_table, t.coef.data(), t.coef.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        (void)hipGetLastError();
        launch_resample(nullptr, static_cast<float *>(r->d_pcm), src->d_pcm, format == BN_PCM_I16, d_table, n_samples, n_dst, t.L, t.M, t.T);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = gated::DeviceSynchronize();
    if (d_table) (void)gated::Free(d_table);
    bn_recording_free(src);
    if (e != hipSuccess) {
        if (r->d_pcm) (void)gated::Free(r->d_pcm);
        return fail(BN_ERR_BACKEND, std::string("resampling failed: ") + hipGetErrorString(e));
    }
    *out = r.release();
    return BN_OK;
}

bn_status bn_recording_read_f32(const bn_recording *r, size_t first, size_t count, float *host_out) {
    if (!r || (count && !host_out)) return fail(BN_ERR_INVALID_ARG, "null argument");
    if (r->format != BN_PCM_F32) return fail(BN_ERR_INVALID_ARG, "recording is not f32");
    if (first > r->n_samples || count > r->n_samples - first) return fail(BN_ERR_INVALID_ARG, "sample range exceeds the recording");
    if (count == 0) return BN_OK;
    if (!recording_wait_samples(r, first + count - 1)) return fail(BN_ERR_BACKEND, "recording upload failed");
    HIP_TRY(bn::use_device(r->device));
    HIP_TRY(gated::Memcpy(host_out, static_cast<const float *>(r->d_pcm) + first, count * sizeof(float), hipMemcpyDeviceToHost));
    return BN_OK;
}

void bn_recording_free(bn_recording *r) {
    if (!r) return;
    if (r->uploader.joinable()) r->uploader.join();  // (the caller's buffer is read until here at the latest)
    (void)bn::use_device(r->device);
    if (r->d_pcm) (void)gated::Free(r->d_pcm);
    delete r;
}

size_t bn_recording_samples(const bn_recording *r) { return r ? r->n_samples : 0; }

size_t bn_chunk_count(size_t n_samples, size_t step_samples) {
    if (n_samples == 0 || step_samples == 0) return 0;
    return (n_samples + step_samples - 1) / step_samples;  // one window for every pos = k*step < n_samples
}

static bn_status check_windows(const bn_recording *r, size_t S, size_t step, size_t first, size_t count) {
    if (!r) return fail(BN_ERR_INVALID_ARG, "null recording");
    if (S == 0 || S % 4 != 0 || S > 0xffffffffull) return fail(BN_ERR_INVALID_ARG, "segment_samples must be a positive multiple of 4");
    if (step == 0) return fail(BN_ERR_INVALID_ARG, "step_samples must be positive (overlap shorter than the segment)");
    const size_t total = bn_chunk_count(r->n_samples, step);
    if (first > total || count > total - first) return fail(BN_ERR_INVALID_ARG, "window range [" + std::to_string(first) + ", " + std::to_string(first + count) + ") exceeds the " + std::to_string(total) + " windows of the recording");
    // an asynchronously uploaded recording: the last sample these windows read must have arrived before their kernel is launched
    if (count && r->n_samples && !recording_wait_samples(r, std::min(r->n_samples - 1, (first + count - 1) * step + S - 1)))
        return fail(BN_ERR_BACKEND, "recording upload failed");
    return BN_OK;
}

bn_status bn_recording_windows(const bn_recording *r, size_t segment_samples, size_t step_samples, size_t first_window, size_t count, float *host_out) {
    bn_status st = check_windows(r, segment_samples, step_samples, first_window, count);
    if (st != BN_OK) return st;
    if (count == 0) return BN_OK;
    if (!host_out) return fail(BN_ERR_INVALID_ARG, "null host buffer");
    HIP_TRY(bn::use_device(r->device));
    float *d = nullptr;
    HIP_TRY(gated::Malloc(reinterpret_cast<void **>(&d), count * segment_samples * sizeof(float)));
    (void)hipGetLastError();
    launch_windows(nullptr, d, r->d_pcm, r->format == BN_PCM_I16, r->n_samples, (uint64_t)first_window * step_samples, step_samples, (uint32_t)segment_samples,
                   (uint32_t)count);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = gated::Memcpy(host_out, d, count * segment_samples * sizeof(float), hipMemcpyDeviceToHost);
    (void)gated::Free(d);
    if (e != hipSuccess) return fail(BN_ERR_BACKEND, std::string("window kernel failed: ") + hipGetErrorString(e));
    return BN_OK;
}

bn_status bn_infer_windows(bn_ctx *c, const bn_recording *r, size_t step_samples, size_t first_window, size_t count, float *logits_out, float *emb_out,
                           const volatile int32_t *cancel, uint64_t timeout_ns) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    const size_t S = (size_t)c->pd->plan->sample_count;
    bn_status st = check_windows(r, S, step_samples, first_window, count);
    if (st != BN_OK) return st;
    if (count == 0) return BN_OK;
    if (!logits_out) return fail(BN_ERR_INVALID_ARG, "null argument");
    if (count > c->max_batch) return fail(BN_ERR_INVALID_ARG, "batch size " + std::to_string(count) + " exceeds context max " + std::to_string(c->max_batch));
    if (r->device != c->model->device) return fail(BN_ERR_INVALID_ARG, "recording and context live on different devices");
    HIP_TRY(bn::use_device(c->model->device));
    st = drain_if_needed(c);
    if (st != BN_OK) return st;
    if (cancel && *cancel) return fail(BN_ERR_CANCELLED, "inference was cancelled");
    (void)hipGetLastError();
    launch_windows(c->stream, c->d_input, r->d_pcm, r->format == BN_PCM_I16, r->n_samples, (uint64_t)first_window * step_samples, step_samples, (uint32_t)S,
                   (uint32_t)count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BN_ERR_BACKEND, std::string("window kernel launch failed: ") + hipGetErrorString(e));
    st = enqueue_plan(c, c->d_input, count, cancel);
    if (st != BN_OK) {
        c->in_flight = true;
        return st;
    }
    return finish_infer(c, count, logits_out, emb_out, cancel, timeout_ns);
}

bn_status bn_step_windows(bn_ctx *c, const bn_recording *r, size_t step_samples, size_t first_window, size_t count, size_t top_k, int32_t has_min,
                          float min_conf, int32_t sync) {
    if (!c) return fail(BN_ERR_INVALID_ARG, "null context");
    const size_t S = (size_t)c->pd->plan->sample_count;
    bn_status st = check_windows(r, S, step_samples, first_window, count);
    if (st != BN_OK) return st;
    if (count == 0) return BN_OK;
    if (count > c->max_batch) return fail(BN_ERR_INVALID_ARG, "batch size " + std::to_string(count) + " exceeds context max " + std::to_string(c->max_batch));
    if (r->device != c->model->device) return fail(BN_ERR_INVALID_ARG, "recording and context live on different devices");
    HIP_TRY(bn::use_device(c->model->device));
    (void)hipGetLastError();
    // stream order keeps this behind whatever the context still has in flight
    launch_windows(c->stream, c->d_input, r->d_pcm, r->format == BN_PCM_I16, r->n_samples, (uint64_t)first_window * step_samples, step_samples, (uint32_t)S,
                   (uint32_t)count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BN_ERR_BACKEND, std::string("window kernel launch failed: ") + hipGetErrorString(e));
    return bn_step_device(c, c->d_input, count, top_k, has_min, min_conf, sync);
}

size_t bn_model_survey(const char *onnx_path, char *buf, size_t cap, bn_status *status) {
    bn_status dummy;
    if (!status) status = &dummy;
    *status = BN_OK;
    std::string text;
    char line[1024];
    try {
        if (!onnx_path) throw std::runtime_error("null path");
        OnnxModel om = parse_onnx_file(onnx_path);
        IoMeta io = read_io_meta(om);
        auto shape_str = [](const std::vector<int64_t> &s) {
            std::string r = "[";
            for (size_t k = 0; k < s.size(); k++) r += (k ? ", " : "") + (s[k] < 0 ? std::string("?") : std::to_string(s[k]));
            return r + "]";
        };
        snprintf(line, sizeof(line), "opset %lld, %zu nodes, %zu initializers\n", (long long)om.opset, om.nodes.size(), om.initializers.size());
        text += line;
        text += "input  " + io.input_name + " " + shape_str(io.input_shape) + "\n";
        for (size_t k = 0; k < io.output_names.size(); k++) text += "output " + std::to_string(k) + " " + io.output_names[k] + " " + shape_str(io.output_shapes[k]) + "\n";
        bn_model_config cfg{};
        std::string reason;
        const bool detected = detect_model_type(io.input_shape, io.output_shapes, -1, cfg, reason);
        if (detected) {
            snprintf(line, sizeof(line), "detected model_type=%d sample_rate=%u sample_count=%llu num_species=%llu embedding_dim=%llu logits_output=%d embedding_output=%d\n",
                     cfg.model_type, cfg.sample_rate, (unsigned long long)cfg.sample_count, (unsigned long long)cfg.num_species,
                     (unsigned long long)cfg.embedding_dim, cfg.logits_output, cfg.embedding_output);
            text += line;
        } else text += "detection FAILED: " + reason + "\n";
        // operator types: count, and whether the lowering has a rule for the type at all
        std::map<std::string, std::pair<int, std::string>> hist;
        for (const auto &nd : om.nodes) {
            auto &h = hist[nd.op_type];
            if (h.first++ == 0) h.second = nd.name.empty() ? (nd.outputs.empty() ? std::string("?") : nd.outputs[0]) : nd.name;
        }
        int unmapped_types = 0, unmapped_nodes = 0;
        for (const auto &kv : hist) {
            const bool ok = op_type_mapped(kv.first);
            if (!ok) { unmapped_types++; unmapped_nodes += kv.second.first; }
            snprintf(line, sizeof(line), "op %-24s x%-5d %s%s\n", kv.first.c_str(), kv.second.first, ok ? "mapped" : "NOT MAPPED   first: ", ok ? "" : kv.second.second.c_str());
            text += line;
        }
        snprintf(line, sizeof(line), "unmapped: %d operator types, %d nodes\n", unmapped_types, unmapped_nodes);
        text += line;
        // the plan itself (every graph output, so that no refusal hides in a branch the audio path does not read; then the audio path)
        for (int all = 1; all >= 0; all--) {
            std::vector<int> wanted;
            if (all) for (size_t k = 0; k < om.outputs.size(); k++) wanted.push_back((int)k);
            else if (detected) { wanted.push_back(cfg.logits_output); if (cfg.embedding_output >= 0) wanted.push_back(cfg.embedding_output); }
            else continue;
            try {
                auto p = build_plan(om, wanted);
                snprintf(line, sizeof(line), "plan (%s): OK, %zu launches, %.4g MMAC/segment on the matrix cores, %.4g on the vector ALU, arena %.3g MB/segment, weights %.3g MB\n",
                         all ? "every graph output" : "logits + embeddings", p->ops.size(), p->macs_mfma / 1e6, p->macs_valu / 1e6, p->arena_elems * 4.0 / 1e6, p->weight_bytes / 1e6);
                text += line;
            } catch (const std::exception &e) {
                if (*status == BN_OK || !all) *status = BN_ERR_UNSUPPORTED_MODEL;
                text += std::string("plan (") + (all ? "every graph output" : "logits + embeddings") + "): REFUSED: " + e.what() + "\n";
                if (!all) (void)fail(BN_ERR_UNSUPPORTED_MODEL, e.what());
            }
        }
    } catch (const std::exception &e) {
        *status = fail(BN_ERR_MODEL_LOAD, e.what());
        return 0;
    }
    if (buf && cap) {
        const size_t n = std::min(cap - 1, text.size());
        memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return text.size();
}

size_t bn_plan_describe(const char *onnx_path, int32_t model_type_override, int32_t all_outputs, char *buf, size_t cap, bn_status *status) {
    bn_status dummy;
    if (!status) status = &dummy;
    *status = BN_OK;
    std::string text;
    try {
        if (!onnx_path) throw std::runtime_error("null path");
        OnnxModel om = parse_onnx_file(onnx_path);
        IoMeta io = read_io_meta(om);
        std::vector<int> wanted;
        bn_model_config cfg{};
        std::string reason;
        if (all_outputs) {
            for (size_t k = 0; k < om.outputs.size(); k++) wanted.push_back((int)k);
        } else {
            if (!detect_model_type(io.input_shape, io.output_shapes, model_type_override, cfg, reason)) {
                *status = fail(BN_ERR_MODEL_DETECTION, reason);
                return 0;
            }
            wanted.push_back(cfg.logits_output);
            if (cfg.embedding_output >= 0) wanted.push_back(cfg.embedding_output);
        }
        auto p = build_plan(om, wanted);
        static const char *kinds[] = {"ELT", "REDUCE", "GEMM", "CONV", "DWCONV", "GAP", "SEFC", "MBCONV", "POOL", "FFT"};
        char line[512];
        for (size_t k = 0; k < p->ops.size(); k++) {
            const PlanOp &op = p->ops[k];
            std::string extra;
            if (op.kind == OpKind::GEMM) {
                snprintf(line, sizeof(line), " rows=%lld K=%d N=%d lda=%lld act=%d bias=%d res=%d gate=%d", (long long)op.gemm.rows, op.gemm.K, op.gemm.N, (long long)op.gemm.lda, op.gemm.act, op.gemm.has_bias, op.gemm.has_res, op.gemm.has_scale);
                extra = line;
                if (op.gemm.fold) { snprintf(line, sizeof(line), " fold=%d/%d", op.gemm.fold, op.gemm.fold_n); extra += line; }
                if (op.pre.n) { snprintf(line, sizeof(line), " pre=%d", op.pre.n); extra += line; }
                if (op.gemm.gap) extra += " gap=1";
                if (op.gemm.se_inline) { snprintf(line, sizeof(line), " se_inline=%d/%d", op.se.C, op.se.Cr); extra += line; }
                if (op.gemm2.N > 0) { snprintf(line, sizeof(line), " pair=%dx%d post=%d out_rs=%lld out_cs=%lld", op.gemm2.K, op.gemm2.N, op.gemm2.npost, (long long)op.gemm2.out_rs, (long long)op.gemm2.out_cs); extra += line; }
                // which of the three matrix kernels the launcher picks (the LDS-resident framing kernel may still fall back to
                // the generic one at launch: BN_FRAMELDS=0 or a span that does not fit)
                extra += op.gemm.fold == 2 ? (op.gemm.fold_wpk == 2 ? " kernel=frame_fold2q" : op.gemm.fold_wpk ? " kernel=frame_fold2p" : " kernel=frame_fold2") : op.gemm.fold ? " kernel=frame_fold" : gemm_dma_shape(op.gemm) == 3 ? " kernel=dma-stream" : op.gemm.w3 == 2 ? " kernel=b3" : op.gemm.w3 ? " kernel=dma3" : gemm_dma_shape(op.gemm) ? " kernel=dma" : (!(op.gemm.npost || op.gemm.out_strided) && gemm_use_splitk(op.gemm) ? " kernel=splitk" : " kernel=tiled");
                if (op.gemm.npost || op.gemm.out_strided) {
                    snprintf(line, sizeof(line), " post=%d out_rs=%lld out_cs=%lld", op.gemm.npost, (long long)op.gemm.out_rs, (long long)op.gemm.out_cs);
                    extra += line;
                }
            } else if (op.kind == OpKind::DWCONV) {
                snprintf(line, sizeof(line), " %dx%dx%d->%dx%d k=%dx%d s=%d act=%d tiled=%d squeeze=%d nblk=%d se=%d", op.dw.H, op.dw.W, op.dw.C, op.dw.OH, op.dw.OW, op.dw.kh, op.dw.kw, op.dw.sh, op.dw.act, op.dw.tiled, op.dw.has_gap, op.dw.nblk, op.se_fused);
                extra = line;
            } else if (op.kind == OpKind::CONV) {
                snprintf(line, sizeof(line), " %dx%dx%d->%dx%dx%d k=%dx%d s=%d g=%d act=%d", op.conv.H, op.conv.W, op.conv.Cin, op.conv.OH, op.conv.OW, op.conv.Cout, op.conv.kh, op.conv.kw, op.conv.sh, op.conv.groups, op.conv.act);
                extra = line;
            } else if (op.kind == OpKind::ELT) {
                int w = snprintf(line, sizeof(line), " n=%lld nd=%d stages=%d bin0=%d act0=%d size/so/sa=", (long long)op.elt.per_sample, op.elt.nd, op.elt.nstages, op.elt.st[0].bin, op.elt.st[0].act);
                for (int q = 0; q < op.elt.nd && w < (int)sizeof(line) - 40; q++)
                    w += snprintf(line + w, sizeof(line) - w, "%s%lld/%lld/%lld", q ? "," : "", (long long)op.elt.size[q], (long long)op.elt.so[q], (long long)op.elt.sa[q]);
                extra = line;
            } else if (op.kind == OpKind::MBCONV) {
                snprintf(line, sizeof(line), " %dx%dx%d->(%d)->%dx%dx%d k=%d s=%d tiles=%dx%d squeeze=%d se=%d rows=%d%s", op.mb.H, op.mb.W, op.mb.Cin, op.mb.C, op.mb.OH, op.mb.OW, op.mb.C, op.mb.k, op.mb.s, op.mb.tiles_y, op.mb.tiles_x, op.mb.has_gap, op.se_fused,
                         op.mb.row_mode ? op.mb.toh : 0, op.mb.row_mode && op.mb.row_tr ? "(columns)" : "");
                extra = line;
                if (op.mb.whole_map == 2) {  // mbmap.hip: configuration, bands, transposition, padded k
                    snprintf(line, sizeof(line), " map=cfg%d%s%s%s kpad=%d", mbmap_config(op.mb), op.mb.map_bands > 1 ? ",bands" : "", op.mb.map_tr ? ",transposed" : "", op.mb.map_ws ? ",ws" : op.mb.map_b3 ? ",b3" : "", op.mb.cin_pad);
                    extra += line;
                }
            } else if (op.kind == OpKind::POOL) {
                snprintf(line, sizeof(line), " %dx%dx%d->%dx%d k=%dx%d s=%dx%d max=%d", op.pool.H, op.pool.W, op.pool.C, op.pool.OH, op.pool.OW, op.pool.kh, op.pool.kw,
                         op.pool.sh, op.pool.sw, op.pool.is_max);
                extra = line;
            } else if (op.kind == OpKind::GAP) {
                snprintf(line, sizeof(line), " HW=%lld C=%d splits=%d", (long long)op.gap.HW, op.gap.C, op.gap.splits);
                extra = line;
            } else if (op.kind == OpKind::SEFC) {
                snprintf(line, sizeof(line), " C=%d Cr=%d act1=%d act2=%d", op.se.C, op.se.Cr, op.se.act1, op.se.act2);
                extra = line;
            } else if (op.kind == OpKind::FFT) {
                snprintf(line, sizeof(line), " frames=%d L=%d hop=%d bins=%d power=%d tpb=%d mel=%d%s pre=%d post=%d out_rs=%lld out_cs=%lld fft_flops=%.3g", op.fft.frames, op.fft.L,
                         op.fft.hop, op.fft.nout, op.fft.power, op.fft.tpb, op.fft.nmel, op.fft.nmel ? (op.fft.mel_mode == 1 ? "(mfma)" : "(csr)") : "", op.fft.npre, op.fft.npost, (long long)op.fft.out_rs, (long long)op.fft.out_cs, op.flops_fft);
                extra = line;
            } else {
                snprintf(line, sizeof(line), " kept=%lld red=%lld op=%d inner_kept=%d", (long long)op.red.kept, (long long)op.red.red, op.red.op, op.red.inner_kept);
                extra = line;
            }
            snprintf(line, sizeof(line), "%3zu %-6s %-40s macs=%.3g bytes=%.3g", k, kinds[(int)op.kind], op.name.c_str(), op.macs, op.bytes);
            text += line + extra + "\n";
        }
        snprintf(line, sizeof(line), "TOTAL launches=%zu macs_mfma=%.6g macs_valu=%.6g act_bytes=%.6g weight_bytes=%.6g arena_bytes_per_sample=%lld consts_bytes=%lld dft_gemm_macs=%.6g fft_flops=%.6g\n",
                 p->ops.size(), p->macs_mfma, p->macs_valu, p->act_bytes, p->weight_bytes, (long long)p->arena_elems * 4, (long long)p->consts_elems * 4, p->dft_gemm_macs, p->fft_flops);
        text += line;
        for (size_t k = 0; k < p->outputs.size(); k++) {
            snprintf(line, sizeof(line), "OUTPUT %zu %s computed=%d row_elems=%lld\n", k, p->outputs[k].name.c_str(), (int)p->outputs[k].computed, (long long)p->outputs[k].row_elems);
            text += line;
        }
    } catch (const UnsupportedModel &e) {
        *status = fail(BN_ERR_UNSUPPORTED_MODEL, e.what());
        return 0;
    } catch (const std::exception &e) {
        *status = fail(BN_ERR_MODEL_LOAD, e.what());
        return 0;
    }
    if (buf && cap) snprintf(buf, cap, "%s", text.c_str());
    return text.size();
}

size_t bn_last_error(char *buf, size_t cap) {
    if (buf && cap) snprintf(buf, cap, "%s", g_err.c_str());
    return g_err.size();
}

}  // extern "C"
