// Multi-GPU behind the C ABI (include/birdnet_hip.h, bn_group_*): BASELINE.json configs[4] -- a long recording,
// sharded by window over the GPUs of one node, ONE all-gather of the logits (or of the top-K rows) at the end.
//
// The reference has no multi-device code at all (its only knob is device_id, src/cuda_config.rs:179-182); the unit
// of work it defines is the window of chunk_audio (src/bin/birdnet-analyze.rs:707-743), and windows are independent,
// so the partition is by contiguous window range: rank r of R owns [r * ceil(G/R), min(G, (r+1) * ceil(G/R))).
//
// One process, one host thread per device for the duration of a call (each with its own contexts = HIP streams on its
// device); results stay on the device until the collective: every rank's rows are copied device-to-device into its
// slab of a [R * ceil(G/R), row] buffer and all-gathered in place with RCCL (ncclAllGather over xGMI, one collective
// for the logits and one for the packed top-K rows).  RCCL is loaded with dlopen on first use, so a single-GPU host
// never needs it; when two ranks share a device (tests on a one-GPU box) or RCCL is absent, the gather degenerates
// to device copies between the ranks' buffers -- same bytes, same layout.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include "hip_gate.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/birdnet_hip.h"

using namespace bn;

namespace {

thread_local std::string g_group_err;

// ---- RCCL through dlopen (types restated from rccl.h: opaque communicator, result code 0 = success, ncclFloat32 = 7,
// ncclUint32 = 3)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // BN_RCCL_LIB names the library to load instead (a site's own build -- or the test stub that lets the RCCL branch
        // below execute on a one-GPU box, tests/stubs/rccl_stub.cpp); when it is set, nothing else is tried
        if (const char *forced = getenv("BN_RCCL_LIB")) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) fprintf(stderr, "libbirdnet_hip: BN_RCCL_LIB=%s could not be loaded (%s); the group gathers with device copies\n", forced, dlerror());
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        }
        if (!r.lib) return;
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    });
    return r;
}
constexpr int kNcclFloat32 = 7, kNcclUint32 = 3;

}  // namespace

struct bn_group {
    struct Rank {
        bn_model *model = nullptr;
        int device = 0;
        std::vector<bn_ctx *> ctxs;
        hipStream_t stream = nullptr;  // staging copies + the collective
        float *d_logits = nullptr;     // [world * per_rank, N]: this rank's slab is filled locally, the rest by the gather
        uint32_t *d_topk = nullptr;    // [world * per_rank, 2k + 1] packed rows: idx[k] | conf[k] | count
        size_t cap_rows = 0, cap_k = 0, cap_rows_logits = 0;
        void *comm = nullptr;
        std::string error;
        bn_status status = BN_OK;
    };
    std::vector<Rank> ranks;
    size_t max_batch = 0;
    size_t N = 0;
    bn_model_config cfg{};
    bool use_rccl = false;
};

namespace {

bn_status gfail(bn_status st, const std::string &msg) {
    g_group_err = msg;
    return st;
}

void free_rank_buffers(bn_group::Rank &r) {
    (void)bn::use_device(r.device);
    if (r.d_logits) (void)gated::Free(r.d_logits);
    if (r.d_topk) (void)gated::Free(r.d_topk);
    r.d_logits = nullptr;
    r.d_topk = nullptr;
    r.cap_rows = r.cap_k = r.cap_rows_logits = 0;
}

}  // namespace

extern "C" {

size_t bn_group_last_error(char *buf, size_t cap) {
    if (buf && cap) snprintf(buf, cap, "%s", g_group_err.c_str());
    return g_group_err.size();
}

void bn_shard_range(size_t n_windows, int32_t rank, int32_t world, size_t *lo, size_t *hi) {
    const size_t w = world > 0 ? (size_t)world : 1, r = rank > 0 ? (size_t)rank : 0;
    const size_t per = (n_windows + w - 1) / w;
    const size_t a = std::min(n_windows, r * per), b = std::min(n_windows, (r + 1) * per);
    if (lo) *lo = a;
    if (hi) *hi = b;
}

bn_status bn_group_create(bn_model *const *models, const int32_t *devices, int32_t n, size_t max_batch, int32_t contexts_per_device, bn_group **out) {
    if (!models || !devices || !out || n <= 0) return gfail(BN_ERR_INVALID_ARG, "null argument / empty group");
    *out = nullptr;
    if (max_batch == 0 || contexts_per_device <= 0 || contexts_per_device > 8) return gfail(BN_ERR_INVALID_ARG, "max_batch must be positive, contexts_per_device in 1..8");
    auto g = std::make_unique<bn_group>();
    g->max_batch = max_batch;
    g->ranks.resize((size_t)n);
    std::set<int> distinct;
    for (int r = 0; r < n; r++) {
        if (!models[r]) return gfail(BN_ERR_INVALID_ARG, "model " + std::to_string(r) + " is null");
        bn_model_config c{};
        bn_status st = bn_model_get_config(models[r], &c);
        if (st != BN_OK) return gfail(st, "bn_model_get_config failed");
        if (r == 0) g->cfg = c;
        else if (c.sample_count != g->cfg.sample_count || c.num_species != g->cfg.num_species || c.model_type != g->cfg.model_type)
            return gfail(BN_ERR_INVALID_ARG, "the models of a group must be replicas of one file");
        if (bn_model_device(models[r]) != devices[r])
            return gfail(BN_ERR_INVALID_ARG, "model " + std::to_string(r) + " was loaded on device " + std::to_string(bn_model_device(models[r])) + ", not on " + std::to_string(devices[r]));
        g->ranks[(size_t)r].model = models[r];
        g->ranks[(size_t)r].device = devices[r];
        distinct.insert(devices[r]);
    }
    auto cleanup = [&](bn_status st, const std::string &msg) {
        for (auto &rk : g->ranks) {
            for (bn_ctx *c : rk.ctxs) bn_ctx_destroy(c);
            if (rk.stream) { (void)bn::use_device(rk.device); (void)gated::StreamDestroy(rk.stream); }
            if (rk.comm && rccl().ok()) (void)rccl().CommDestroy(rk.comm);
        }
        return gfail(st, msg);
    };
    for (auto &rk : g->ranks) {
        if (bn::use_device(rk.device) != hipSuccess) return cleanup(BN_ERR_NO_DEVICE, "device " + std::to_string(rk.device) + " is not usable");
        if (gated::StreamCreateWithFlags(&rk.stream, hipStreamNonBlocking) != hipSuccess) return cleanup(BN_ERR_BACKEND, "hipStreamCreate failed");
        for (int k = 0; k < contexts_per_device; k++) {
            bn_ctx *c = nullptr;
            bn_status st = bn_ctx_create(rk.model, max_batch, BN_CTX_DEFAULT, &c);
            if (st != BN_OK) {
                char msg[512];
                bn_last_error(msg, sizeof(msg));
                return cleanup(st, std::string("bn_ctx_create: ") + msg);
            }
            rk.ctxs.push_back(c);
        }
    }
    {
        const float *dp = nullptr;
        size_t row = 0;
        if (bn_ctx_output_device(g->ranks[0].ctxs[0], g->cfg.logits_output, &dp, &row) != BN_OK) return cleanup(BN_ERR_BACKEND, "logits output not planned");
        g->N = row;
    }
    // RCCL only for n > 1 ranks on pairwise distinct devices (one communicator per device in this process)
    // (BN_GROUP_FORCE_RCCL=1 takes the branch for ranks that SHARE a device too: the real library refuses such a communicator --
    // its error is reported below --, the test stub accepts it, which is how the branch is exercised on a one-GPU box)
    const bool force_rccl = getenv("BN_GROUP_FORCE_RCCL") && atoi(getenv("BN_GROUP_FORCE_RCCL")) != 0;
    if (n > 1 && ((int)distinct.size() == n || force_rccl) && !getenv("BN_GROUP_NO_RCCL")) {
        Rccl &rc = rccl();
        if (rc.ok()) {
            std::vector<void *> comms((size_t)n, nullptr);
            std::vector<int> devs(devices, devices + n);
            const int res = rc.CommInitAll(comms.data(), n, devs.data());
            if (res != 0) return cleanup(BN_ERR_BACKEND, std::string("ncclCommInitAll: ") + (rc.GetErrorString ? rc.GetErrorString(res) : "error"));
            for (int r = 0; r < n; r++) g->ranks[(size_t)r].comm = comms[(size_t)r];
            g->use_rccl = true;
        }
    }
    *out = g.release();
    return BN_OK;
}

void bn_group_destroy(bn_group *g) {
    if (!g) return;
    for (auto &rk : g->ranks) {
        for (bn_ctx *c : rk.ctxs) bn_ctx_destroy(c);
        free_rank_buffers(rk);
        if (rk.comm && rccl().ok()) (void)rccl().CommDestroy(rk.comm);
        if (rk.stream) { (void)bn::use_device(rk.device); (void)gated::StreamDestroy(rk.stream); }
    }
    delete g;
}

int32_t bn_group_size(const bn_group *g) { return g ? (int32_t)g->ranks.size() : 0; }
int32_t bn_group_uses_rccl(const bn_group *g) { return g && g->use_rccl ? 1 : 0; }

bn_status bn_group_get_stats(const bn_group *g, bn_ctx_stats *out, size_t struct_size) {
    if (!g || !out) return gfail(BN_ERR_INVALID_ARG, "null argument");
    bn_ctx_stats sum{};
    for (const auto &rk : g->ranks)
        for (const bn_ctx *c : rk.ctxs) {
            bn_ctx_stats s{};
            if (bn_ctx_get_stats(c, &s, sizeof(s)) != BN_OK) continue;
            sum.captures += s.captures;
            sum.instantiates += s.instantiates;
            sum.replays += s.replays;
            sum.eager_runs += s.eager_runs;
            sum.capture_fallbacks += s.capture_fallbacks;
            sum.evictions += s.evictions;
            sum.cached_graphs += s.cached_graphs;
            sum.input_copies += s.input_copies;
            if (s.last_fallback[0]) memcpy(sum.last_fallback, s.last_fallback, sizeof(sum.last_fallback));
        }
    memcpy(out, &sum, std::min(struct_size, sizeof(sum)));
    return BN_OK;
}

bn_status bn_group_analyze_recording(bn_group *g, const void *pcm, size_t n_samples, int32_t format, size_t step_samples, size_t top_k, int32_t has_min,
                                     float min_conf, float *logits_out, size_t k_stride, uint32_t *idx_out, float *conf_out, uint32_t *count_out,
                                     size_t *n_windows_out) {
    if (!g) return gfail(BN_ERR_INVALID_ARG, "null group");
    if (format != BN_PCM_I16 && format != BN_PCM_F32) return gfail(BN_ERR_INVALID_ARG, "unknown PCM format");
    if (n_samples && !pcm) return gfail(BN_ERR_INVALID_ARG, "null PCM buffer");
    const size_t S = (size_t)g->cfg.sample_count, N = g->N;
    const size_t G = bn_chunk_count(n_samples, step_samples);  // 0 when the step saturated to 0 (chunk_audio returns nothing)
    if (n_windows_out) *n_windows_out = G;
    if (G == 0) return BN_OK;
    const size_t k = std::min(top_k, N);
    if (count_out && k && (!idx_out || !conf_out || k_stride < k)) return gfail(BN_ERR_INVALID_ARG, "top-K outputs need idx / conf buffers with k_stride >= min(top_k, num_species)");
    const size_t R = g->ranks.size();
    const size_t per = (G + R - 1) / R;  // rows per rank in the gathered buffers (the last ranks' tails are padding)
    const size_t tkw = 2 * k + 1;        // words of one packed top-K row
    const size_t esz = format == BN_PCM_I16 ? sizeof(int16_t) : sizeof(float);

    // ---- per-rank work: upload the slice, step the windows, collect rows device-to-device into the rank's slab
    auto work = [&](size_t r) {
        bn_group::Rank &rk = g->ranks[r];
        auto fail = [&](bn_status st, const std::string &m) { rk.status = st; rk.error = "rank " + std::to_string(r) + ": " + m; };
        rk.status = BN_OK;
        if (bn::use_device(rk.device) != hipSuccess) return fail(BN_ERR_NO_DEVICE, "hipSetDevice failed");
        // the [R * per, N] logits slab (751 MB for the 24 h recording of BASELINE configs[4]) exists only on request;
        // the default gather moves the packed top-K rows alone (2k + 1 words per window)
        if (per > rk.cap_rows || k > rk.cap_k || (logits_out && per > rk.cap_rows_logits)) {
            const bool had_logits = rk.cap_rows_logits > 0;
            free_rank_buffers(rk);
            if ((logits_out || had_logits) && gated::Malloc(&rk.d_logits, R * per * N * sizeof(float)) != hipSuccess)
                return fail(BN_ERR_BACKEND, "out of device memory for the gathered logits");
            if (rk.d_logits) rk.cap_rows_logits = per;
            if (k && gated::Malloc(&rk.d_topk, R * per * tkw * sizeof(uint32_t)) != hipSuccess) return fail(BN_ERR_BACKEND, "out of device memory for the gathered top-K rows");
            rk.cap_rows = per;
            rk.cap_k = k;
        }
        size_t lo, hi;
        bn_shard_range(G, (int32_t)r, (int32_t)R, &lo, &hi);
        const size_t n_local = hi - lo;
        // padding rows of this rank's slab are defined (zero): they travel through the collective
        if (n_local < per) {
            if (logits_out) (void)hipMemsetAsync(rk.d_logits + (r * per + n_local) * N, 0, (per - n_local) * N * sizeof(float), rk.stream);
            if (k) (void)hipMemsetAsync(rk.d_topk + (r * per + n_local) * tkw, 0, (per - n_local) * tkw * sizeof(uint32_t), rk.stream);
        }
        if (n_local == 0) {
            (void)hipStreamSynchronize(rk.stream);
            return;
        }
        // the samples windows [lo, hi) touch: [lo * step, min(n, (hi - 1) * step + S))
        const size_t a = lo * step_samples, bnd = std::min(n_samples, (hi - 1) * step_samples + S);
        bn_recording *rec = nullptr;
        // (asynchronous: the rank's first windows are analysed while the rest of its slice is still crossing the bus; the caller's
        // buffer is only read inside this call -- bn_recording_free below joins the upload)
        bn_status st = bn_recording_create_async(rk.device, static_cast<const char *>(pcm) + a * esz, bnd - a, format, &rec);
        if (st != BN_OK) {
            char msg[512];
            bn_last_error(msg, sizeof(msg));
            return fail(st, std::string("recording upload: ") + msg);
        }
        const size_t C = rk.ctxs.size(), B = g->max_batch;
        struct Job { size_t first, count; };
        std::vector<Job> jobs;
        for (size_t f = 0; f < n_local; f += B) jobs.push_back({f, std::min(B, n_local - f)});
        auto collect = [&](size_t j) -> bool {
            bn_ctx *c = rk.ctxs[j % C];
            // the context's stream is ordered: plan -> top-K -> D2H into pinned; the device-side rows are copied from the
            // context's buffers on the SAME stream, so nothing else needs to wait for the host
            const float *d_logits = nullptr;
            size_t row = 0;
            if (bn_ctx_output_device(c, g->cfg.logits_output, &d_logits, &row) != BN_OK) return false;
            hipStream_t cs = static_cast<hipStream_t>(bn_ctx_stream(c));
            const Job &jb = jobs[j];
            if (logits_out &&
                hipMemcpyAsync(rk.d_logits + (r * per + jb.first) * N, d_logits, jb.count * N * sizeof(float), hipMemcpyDeviceToDevice, cs) != hipSuccess)
                return false;
            if (k) {
                const uint32_t *d_step = nullptr;
                if (bn_ctx_step_device_rows(c, &d_step) != BN_OK) return false;
                // the step block is [idx: m*k][conf: m*k][count: m] for the m rows of the step: re-pack row by row on the way
                uint32_t *dst = rk.d_topk + (r * per + jb.first) * tkw;
                if (hipMemcpy2DAsync(dst, tkw * 4, d_step, k * 4, k * 4, jb.count, hipMemcpyDeviceToDevice, cs) != hipSuccess) return false;
                if (hipMemcpy2DAsync(dst + k, tkw * 4, d_step + jb.count * k, k * 4, k * 4, jb.count, hipMemcpyDeviceToDevice, cs) != hipSuccess) return false;
                if (hipMemcpy2DAsync(dst + 2 * k, tkw * 4, d_step + 2 * jb.count * k, 4, 4, jb.count, hipMemcpyDeviceToDevice, cs) != hipSuccess) return false;
            }
            return true;
        };
        bool ok = true;
        for (size_t j = 0; j < jobs.size() && ok; j++) {
            bn_ctx *c = rk.ctxs[j % C];
            // (a context's next step follows its previous step's copies in stream order)
            st = bn_step_windows(c, rec, step_samples, jobs[j].first, jobs[j].count, std::max<size_t>(top_k, 1), has_min, min_conf, 0);
            if (st != BN_OK) {
                char msg[512];
                bn_last_error(msg, sizeof(msg));
                fail(st, std::string("bn_step_windows: ") + msg);
                ok = false;
                break;
            }
            ok = collect(j);
            if (!ok) fail(BN_ERR_BACKEND, "device-side collection of a step's rows failed");
        }
        for (bn_ctx *c : rk.ctxs) (void)bn_ctx_synchronize(c);
        (void)hipStreamSynchronize(rk.stream);
        bn_recording_free(rec);
    };
    {
        std::vector<std::thread> th;
        for (size_t r = 1; r < R; r++) th.emplace_back(work, r);
        work(0);
        for (auto &t : th) t.join();
    }
    for (auto &rk : g->ranks)
        if (rk.status != BN_OK) return gfail(rk.status, rk.error);

    // ---- the collective: every rank's slab to every rank, in place
    if (R > 1) {
        if (g->use_rccl) {
            Rccl &rc = rccl();
            auto gather = [&](bool topk) -> int {
                int res = rc.GroupStart();
                for (size_t r = 0; r < R && res == 0; r++) {
                    bn_group::Rank &rk = g->ranks[r];
                    (void)bn::use_device(rk.device);
                    if (!topk) res = rc.AllGather(rk.d_logits + r * per * N, rk.d_logits, per * N, kNcclFloat32, rk.comm, rk.stream);
                    else res = rc.AllGather(rk.d_topk + r * per * tkw, rk.d_topk, per * tkw, kNcclUint32, rk.comm, rk.stream);
                }
                const int e = rc.GroupEnd();
                return res ? res : e;
            };
            int res = logits_out ? gather(false) : 0;
            if (res == 0 && k && count_out) res = gather(true);
            if (res != 0) return gfail(BN_ERR_BACKEND, std::string("ncclAllGather: ") + (rc.GetErrorString ? rc.GetErrorString(res) : "error"));
        } else {
            // ranks sharing a device (or no RCCL): the same data movement as plain copies, slab r of rank r into slab r of
            // every other rank
            for (size_t dst = 0; dst < R; dst++)
                for (size_t src = 0; src < R; src++) {
                    if (src == dst) continue;
                    bn_group::Rank &d = g->ranks[dst], &s = g->ranks[src];
                    (void)bn::use_device(d.device);
                    hipError_t e = hipSuccess;
                    if (logits_out) e = hipMemcpyPeerAsync(d.d_logits + src * per * N, d.device, s.d_logits + src * per * N, s.device, per * N * sizeof(float), d.stream);
                    if (e == hipSuccess && k && count_out)
                        e = hipMemcpyPeerAsync(d.d_topk + src * per * tkw, d.device, s.d_topk + src * per * tkw, s.device, per * tkw * sizeof(uint32_t), d.stream);
                    if (e != hipSuccess) return gfail(BN_ERR_BACKEND, std::string("peer copy: ") + hipGetErrorString(e));
                }
        }
        for (auto &rk : g->ranks) {
            (void)bn::use_device(rk.device);
            if (hipStreamSynchronize(rk.stream) != hipSuccess) return gfail(BN_ERR_BACKEND, "the collective failed");
        }
    }

    // ---- results: every rank now holds all G rows; the host reads them from the LAST rank (its copy exists only through
    // the collective for every slab but its own, so a broken gather cannot go unnoticed)
    {
        bn_group::Rank &rk = g->ranks[R - 1];
        (void)bn::use_device(rk.device);
        for (size_t r = 0; r < R; r++) {
            size_t lo, hi;
            bn_shard_range(G, (int32_t)r, (int32_t)R, &lo, &hi);
            if (hi == lo) continue;
            if (logits_out && gated::Memcpy(logits_out + lo * N, rk.d_logits + r * per * N, (hi - lo) * N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
                return gfail(BN_ERR_BACKEND, "download of the gathered logits failed");
            if (k && count_out) {
                std::vector<uint32_t> rows((hi - lo) * tkw);
                if (gated::Memcpy(rows.data(), rk.d_topk + r * per * tkw, rows.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
                    return gfail(BN_ERR_BACKEND, "download of the gathered top-K rows failed");
                for (size_t i = 0; i < hi - lo; i++) {
                    const uint32_t *row = &rows[i * tkw];
                    const uint32_t cnt = std::min<uint32_t>(row[2 * k], (uint32_t)k);
                    count_out[lo + i] = cnt;
                    for (uint32_t j = 0; j < cnt; j++) {
                        idx_out[(lo + i) * k_stride + j] = row[j];
                        memcpy(&conf_out[(lo + i) * k_stride + j], &row[k + j], sizeof(float));
                    }
                }
            }
        }
        if (count_out && k == 0)
            for (size_t i = 0; i < G; i++) count_out[i] = 0;
    }
    return BN_OK;
}

}  // extern "C"
