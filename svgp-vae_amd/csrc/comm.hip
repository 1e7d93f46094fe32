// Data-parallel exchange of the SVGPVAE step over RCCL, enqueued on the SAME stream as the kernels.
//
// The step has three sum-exchanges (SVGPVAE_model.py:328-334,339-340 are the statistics that
// couple batch rows): ws[statA] after phase 0, ws[statB] after phase 1, ws[gradC] after phase 2.  Issuing
// ncclAllReduce from here, on the compute stream, keeps the whole step one in-order queue: no host
// round trip and no cross-stream event wait per collective (each costs ~10 us on this part, the same
// order as the collectives themselves at 135 KB).
//
// RCCL is resolved at run time: the process normally has librccl.so.1 mapped already (PyTorch's), and a
// box without RCCL can still load this library for everything that is not multi-GPU.
#include "common.hpp"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

int rccl_get(Rccl** out) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.ok) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy the process already uses
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
        if (!h) h = dlopen("librccl.so", RTLD_NOW);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
        SVGP_REQUIRE(h != nullptr, SVGP_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
        g_rccl.handle = h;
#define SYM(field, name)                                                                           \
        *(void**)(&g_rccl.field) = dlsym(h, name);                                                 \
        SVGP_REQUIRE(g_rccl.field != nullptr, SVGP_ERR_UNSUPPORTED, "RCCL symbol %s missing", name)
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(AllReduce, "ncclAllReduce");
        SYM(ReduceScatter, "ncclReduceScatter");
        SYM(AllGather, "ncclAllGather");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        g_rccl.ok = true;
    }
    *out = &g_rccl;
    return SVGP_OK;
}

#define SVGP_CHECK_RCCL(r, expr)                                                                   \
    do {                                                                                           \
        ncclResult_t _e = (expr);                                                                  \
        if (_e != ncclSuccess) {                                                                   \
            svgp_set_error("%s failed: %s (%s:%d)", #expr, (r)->GetErrorString(_e), __FILE__, __LINE__); \
            return SVGP_ERR_COMM;                                                                  \
        }                                                                                          \
    } while (0)

#define SVGP_COMM_MAX_POINTS 8
struct Comm {
    ncclComm_t comm;
    int rank, nranks;
    // optional per-exchange-point timing of svgp_mnist_train_step_dp (svgp_comm_timing)
    bool timing = false;
    int npoints = 0;
    hipEvent_t ev[2 * SVGP_COMM_MAX_POINTS] = {};
};

// ---- tile-packed symmetric matrices (include/svgpvae_hip.h svgp_sym_pack)
#define SP_T 32
__device__ __forceinline__ void sym_tile_of(int t, int& ti, int& tj) {
    ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
}
__global__ __launch_bounds__(256) void k_sym_pack(int m, int avg, long long pe, const double* __restrict__ src,
                                                  double* __restrict__ dst) {
    __shared__ double tr[SP_T][SP_T + 1];
    int ti, tj;
    sym_tile_of((int)blockIdx.x, ti, tj);
    const double* X = src + (size_t)blockIdx.y * m * m;
    double* P = dst + (size_t)blockIdx.y * pe + (size_t)blockIdx.x * SP_T * SP_T;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    if (avg && ti != tj) {                       // the mirrored tile (tj, ti), transposed through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 8 * k, gi = tj * SP_T + r, gj = ti * SP_T + c;
            tr[r][c] = (gi < m && gj < m) ? X[(size_t)gi * m + gj] : 0.0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k, gi = ti * SP_T + r, gj = tj * SP_T + c;
        double v = (gi < m && gj < m) ? X[(size_t)gi * m + gj] : 0.0;
        if (avg) {
            const double w = ti != tj ? tr[c][r] : ((gi < m && gj < m) ? X[(size_t)gj * m + gi] : 0.0);
            v = 0.5 * (v + w);
        }
        P[r * SP_T + c] = v;
    }
}
__global__ __launch_bounds__(256) void k_sym_unpack(int m, long long pe, const double* __restrict__ src,
                                                    double* __restrict__ dst) {
    __shared__ double tr[SP_T][SP_T + 1];
    int ti, tj;
    sym_tile_of((int)blockIdx.x, ti, tj);
    const double* P = src + (size_t)blockIdx.y * pe + (size_t)blockIdx.x * SP_T * SP_T;
    double* X = dst + (size_t)blockIdx.y * m * m;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k, gi = ti * SP_T + r, gj = tj * SP_T + c;
        const double v = P[r * SP_T + c];
        tr[r][c] = v;
        // diagonal tiles: the lower part is written here, the upper part below from the transposed read
        if (gi < m && gj < m && (ti != tj || c <= r)) X[(size_t)gi * m + gj] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k, gi = tj * SP_T + r, gj = ti * SP_T + c;      // element (r, c) of the mirrored tile
        if (gi < m && gj < m && (ti != tj || c > r)) X[(size_t)gi * m + gj] = tr[c][r];
    }
}

}  // namespace

extern "C" int svgp_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int svgp_comm_unique_id(void* out, int nbytes) {
    SVGP_REQUIRE(out && nbytes == (int)sizeof(ncclUniqueId), SVGP_ERR_INVALID, "unique id buffer must be %d bytes",
                 (int)sizeof(ncclUniqueId));
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    SVGP_CHECK_RCCL(r, r->GetUniqueId((ncclUniqueId*)out));
    return SVGP_OK;
}

extern "C" int svgp_comm_init(const void* unique_id, int nbytes, int rank, int nranks, void** comm_out) {
    SVGP_REQUIRE(unique_id && comm_out && nbytes == (int)sizeof(ncclUniqueId), SVGP_ERR_INVALID,
                 "unique id buffer must be %d bytes", (int)sizeof(ncclUniqueId));
    SVGP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, SVGP_ERR_INVALID, "rank %d of %d", rank, nranks);
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    Comm* c = new Comm{nullptr, rank, nranks};
    ncclResult_t e = r->CommInitRank(&c->comm, nranks, id, rank);   // binds to the calling thread's current device
    if (e != ncclSuccess) {
        delete c;
        svgp_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, r->GetErrorString(e));
        return SVGP_ERR_COMM;
    }
    *comm_out = c;
    return SVGP_OK;
}

extern "C" int svgp_comm_destroy(void* comm) {
    if (!comm) return SVGP_OK;
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    Comm* c = (Comm*)comm;
    ncclResult_t e = r->CommDestroy(c->comm);
    for (int k = 0; k < 2 * SVGP_COMM_MAX_POINTS; ++k)          // the timing events of svgp_comm_timing
        if (c->ev[k]) (void)hipEventDestroy(c->ev[k]);
    delete c;
    SVGP_REQUIRE(e == ncclSuccess, SVGP_ERR_COMM, "ncclCommDestroy failed: %s", r->GetErrorString(e));
    return SVGP_OK;
}

extern "C" int svgp_allreduce_sum_f64(void* comm, double* buf, int64_t count, void* stream) {
    SVGP_REQUIRE(comm && buf && count >= 0, SVGP_ERR_INVALID, "NULL communicator / buffer");
    if (count == 0) return SVGP_OK;
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    Comm* c = (Comm*)comm;
    SVGP_CHECK_RCCL(r, r->AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream));
    return SVGP_OK;
}

extern "C" int svgp_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream) {
    SVGP_REQUIRE(comm && buf && count >= 0, SVGP_ERR_INVALID, "NULL communicator / buffer");
    if (count == 0) return SVGP_OK;
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    Comm* c = (Comm*)comm;
    SVGP_CHECK_RCCL(r, r->AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
    return SVGP_OK;
}

// In-place reduce-scatter / all-gather of a buffer of nranks equal chunks: rank r's chunk is buf[r * count, (r + 1) * count).
// Channel-sharded exchange of the large statistics blocks: reduce-scatter S (L,m,m) over the channels,
// every rank factors its L / nranks channels, all-gather of what the row stage needs.
extern "C" int svgp_reduce_scatter_sum_f64(void* comm, double* buf, int64_t count_per_rank, void* stream) {
    SVGP_REQUIRE(comm && buf && count_per_rank >= 0, SVGP_ERR_INVALID, "NULL communicator / buffer");
    if (count_per_rank == 0) return SVGP_OK;
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    Comm* c = (Comm*)comm;
    SVGP_CHECK_RCCL(r, r->ReduceScatter(buf, buf + (size_t)c->rank * count_per_rank, (size_t)count_per_rank, ncclFloat64, ncclSum,
                                        c->comm, (hipStream_t)stream));
    return SVGP_OK;
}
extern "C" int svgp_allgather_f64(void* comm, double* buf, int64_t count_per_rank, void* stream) {
    SVGP_REQUIRE(comm && buf && count_per_rank >= 0, SVGP_ERR_INVALID, "NULL communicator / buffer");
    if (count_per_rank == 0) return SVGP_OK;
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    Comm* c = (Comm*)comm;
    SVGP_CHECK_RCCL(r, r->AllGather(buf + (size_t)c->rank * count_per_rank, buf, (size_t)count_per_rank, ncclFloat64, c->comm,
                                    (hipStream_t)stream));
    return SVGP_OK;
}

extern "C" int64_t svgp_sym_packed_elems(int m) {
    if (m < 1) return 0;
    const int64_t nt = (m + SP_T - 1) / SP_T;
    return nt * (nt + 1) / 2 * SP_T * SP_T;
}
extern "C" int svgp_sym_pack(int m, int L, int avg, const double* src, double* dst, void* stream) {
    SVGP_REQUIRE(m >= 1 && L >= 0 && src && dst, SVGP_ERR_INVALID, "bad argument");
    if (L == 0) return SVGP_OK;
    const int nt = (m + SP_T - 1) / SP_T;
    hipLaunchKernelGGL(k_sym_pack, dim3(nt * (nt + 1) / 2, L), dim3(256), 0, (hipStream_t)stream, m, avg,
                       (long long)svgp_sym_packed_elems(m), src, dst);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_sym_unpack(int m, int L, const double* src, double* dst, void* stream) {
    SVGP_REQUIRE(m >= 1 && L >= 0 && src && dst, SVGP_ERR_INVALID, "bad argument");
    if (L == 0) return SVGP_OK;
    const int nt = (m + SP_T - 1) / SP_T;
    hipLaunchKernelGGL(k_sym_unpack, dim3(nt * (nt + 1) / 2, L), dim3(256), 0, (hipStream_t)stream, m,
                       (long long)svgp_sym_packed_elems(m), src, dst);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_comm_group_begin(void* comm) {
    SVGP_REQUIRE(comm, SVGP_ERR_INVALID, "NULL communicator");
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    SVGP_CHECK_RCCL(r, r->GroupStart());
    return SVGP_OK;
}
extern "C" int svgp_comm_group_end(void* comm) {
    SVGP_REQUIRE(comm, SVGP_ERR_INVALID, "NULL communicator");
    Rccl* r;
    int rc = rccl_get(&r);
    if (rc) return rc;
    SVGP_CHECK_RCCL(r, r->GroupEnd());
    return SVGP_OK;
}

extern "C" int svgp_comm_timing(void* comm, int enable) {
    SVGP_REQUIRE(comm, SVGP_ERR_INVALID, "NULL communicator");
    Comm* c = (Comm*)comm;
    if (enable && !c->ev[0])
        for (int k = 0; k < 2 * SVGP_COMM_MAX_POINTS; ++k) SVGP_CHECK_HIP(hipEventCreate(&c->ev[k]));
    c->timing = enable != 0;
    c->npoints = 0;
    return SVGP_OK;
}
extern "C" int svgp_comm_timing_read(void* comm, float* us, int cap, int* n) {
    SVGP_REQUIRE(comm && us && n && cap >= 0, SVGP_ERR_INVALID, "bad argument");
    Comm* c = (Comm*)comm;
    *n = 0;
    for (int k = 0; k < c->npoints && k < cap; ++k) {
        float ms = 0;
        SVGP_CHECK_HIP(hipEventSynchronize(c->ev[2 * k + 1]));
        SVGP_CHECK_HIP(hipEventElapsedTime(&ms, c->ev[2 * k], c->ev[2 * k + 1]));
        us[k] = ms * 1000.0f;
        *n = k + 1;
    }
    return SVGP_OK;
}

// One data-parallel step, everything enqueued on `stream`:
//   phase 0 | all-reduce statA | phase 1 | all-reduce statB | phase 2 | all-reduce gradC | phase 3
// c->b is this rank's row count, c->b_global the global batch, c->rep_weight 1 on exactly one rank.
//
// Large-m path (m > 64) with L divisible by the rank count (Hensman branch): the channel-sharded schedule instead
//  -- the (L,m,m) statistics are reduce-SCATTERED over the channels, every rank factors its L / G channels
// (svgp_gp_factor_*_channels) and what the row stages need is all-gathered:
//   encoder + kernel matrices + statistics | reduce-scatter S, v | window factor stage | all-gather Sigma^-1, t, u |
//   row stage, decoder fwd + bwd, backward statistics | reduce-scatter A2, ud, td | window reverse factor stage |
//   all-gather Ssym, vbar, KL (round 4: M2 = Ki A Ki is neither formed nor exchanged, gp_large.hip "W form") |
//   row gradients, kernel-matrix VJP (every rank's Kbar share counts), encoder reverse pass,
//   gradient reduction | all-reduce gradC | phase 3
// At config 3 on 8 ranks: 2 channels of 256 x 256 per rank instead of 16, 8.4 MB blocks moved as 7/8 of their size.
namespace {
int rs(void* comm, double* p, int64_t total, int nranks, void* stream) {
    return svgp_reduce_scatter_sum_f64(comm, p, total / nranks, stream);
}
int ag(void* comm, double* p, int64_t total, int nranks, void* stream) { return svgp_allgather_f64(comm, p, total / nranks, stream); }
// exchange-point bracket: optional events + one RCCL group
struct Point {
    Comm* cm; hipStream_t st; int idx;
    int begin() {
        idx = -1;
        if (cm->timing && cm->npoints < SVGP_COMM_MAX_POINTS) {
            idx = cm->npoints++;
            SVGP_CHECK_HIP(hipEventRecord(cm->ev[2 * idx], st));
        }
        return SVGP_OK;
    }
    int end() {
        if (idx >= 0) SVGP_CHECK_HIP(hipEventRecord(cm->ev[2 * idx + 1], st));
        return SVGP_OK;
    }
};
// (measured with a 1-rank communicator at m = 256, L = 16: the pack / unpack launches cost ~100 us per step, the 44 % of 8.4 MB
// they take off each of the four points is worth ~15 us apiece on xGMI; at m = 800, L = 64 a point is 328 MB)
bool dp_pack_default(int m) { return m >= 512; }
// Whatever way svgp_mnist_train_step_dp returns, an RCCL group it opened is closed and a side branch it forked is joined
// (an error inside a group left ncclGroup depth above zero -- every later RCCL call of the thread deferred --, an
// error behind the fork an unjoined branch, which under stream capture is an unjoined capture).
struct DpGuard {
    void* comm; void* stream;
    bool group = false, forked = false;
    int begin_group() { int rc = svgp_comm_group_begin(comm); group = rc == SVGP_OK; return rc; }
    int end_group() { group = false; return svgp_comm_group_end(comm); }
    int fork(void** side) { int rc = svgp_side_branch_fork(stream, side); forked = rc == SVGP_OK; return rc; }
    int join() { forked = false; return svgp_side_branch_join(stream); }
    ~DpGuard() {
        if (group) (void)svgp_comm_group_end(comm);
        if (forked) (void)svgp_side_branch_join(stream);
    }
};
}  // namespace

extern "C" int svgp_mnist_train_step_dp(const svgp_mnist_cfg* c, void* comm, double* theta, const double* images,
                                        const double* aux, const double* eps, double* ws, double* state,
                                        double* adam_m, double* adam_v, void* stream) {
    SVGP_REQUIRE(c && comm, SVGP_ERR_INVALID, "NULL cfg / communicator");
    svgp_mnist_ws_layout wl;
    int rc = svgp_mnist_ws_layout_get(c, &wl);
    if (rc) return rc;
    Comm* cm = (Comm*)comm;
    const int G = cm->nranks, L = c->L, m = c->m;
    const bool sharded = m > SVGP_M_MAX && L % G == 0 && !c->titsias && !c->kl_form;
    Point pt{cm, (hipStream_t)stream, -1};
    DpGuard guard{comm, stream};
    if (cm->timing) cm->npoints = 0;
#define RUN(call) do { rc = (call); if (rc) return rc; } while (0)
    if (!sharded && c->split_grad_exchange) {
        // The closing all-reduce in two parts (round 6; prepared for small-message all-reduce latencies above ~20 us on 8 ranks,
        // where three of them per 165 us step would cap weak scaling below 6x): gradC[n_enc:] -- decoder + GP parameters + scalar
        // sums -- is complete once the kernel-matrix reverse pass and reduction part 1 are done and travels on the side branch WHILE
        // the encoder's reverse pass runs on the caller's stream; gradC[:n_enc] follows it.  Same sums, same order on every rank.
        svgp_mnist_param_layout pl;
        RUN(svgp_mnist_param_layout_get(c, &pl));
        const int64_t off2[2] = {wl.statA, wl.statB}, len2[2] = {wl.statA_len, wl.statB_len};
        for (int ph = 0; ph < 2; ++ph) {
            RUN(svgp_mnist_step_phase_deferred(c, ph, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
            RUN(pt.begin());
            RUN(svgp_allreduce_sum_f64(comm, ws + off2[ph], len2[ph], stream));
            RUN(pt.end());
        }
        RUN(svgp_mnist_step_phase_deferred(c, 4, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
        void* side = stream;
        RUN(guard.fork(&side));
        Point pts{cm, (hipStream_t)side, -1};
        RUN(pts.begin());
        RUN(svgp_allreduce_sum_f64(comm, ws + wl.gradC + pl.n_enc, wl.gradC_len - pl.n_enc, side));
        RUN(pts.end());
        RUN(svgp_mnist_step_phase_deferred(c, 5, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
        RUN(pt.begin());
        RUN(svgp_allreduce_sum_f64(comm, ws + wl.gradC, pl.n_enc, stream));
        RUN(pt.end());
        RUN(guard.join());
        RUN(svgp_mnist_step_phase_deferred(c, 3, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
        return SVGP_OK;
    }
    if (!sharded) {
        const int64_t off[3] = {wl.statA, wl.statB, wl.gradC}, len[3] = {wl.statA_len, wl.statB_len, wl.gradC_len};
        for (int ph = 0; ph < 4; ++ph) {
            RUN(svgp_mnist_step_phase_deferred(c, ph, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
            if (ph < 3) {
                RUN(pt.begin());
                RUN(svgp_allreduce_sum_f64(comm, ws + off[ph], len[ph], stream));
                RUN(pt.end());
            }
        }
        return SVGP_OK;
    }
    SVGP_REQUIRE(theta && images && aux && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    svgp_mnist_cfg cc = *c;
    cc.rep_weight = 1.0;                         // Kbar = this rank's channel-window share: every share counts
    const int nl = L / G, l0 = cm->rank * nl;
    const int64_t mm = (int64_t)m * m, Lmm = (int64_t)L * mm, Lm = (int64_t)L * m;
    const char* ev_pack = getenv("SVGP_DP_PACK");
    const bool pack = ev_pack ? ev_pack[0] != '0' : dp_pack_default(m);
    const char* ev_side = getenv("SVGP_SIDE_STREAMS");
    const bool fork = !(ev_side && ev_side[0] == '0');
    const int64_t pe = svgp_sym_packed_elems(m);
    SVGP_REQUIRE(!pack || wl.xpack_len >= (int64_t)L * pe, SVGP_ERR_INVALID,
                 "the packed exchange needs the workspace's wire buffer: lay the workspace out with cfg.single_stat_block = 1");
    double* xp0 = ws + wl.xpack;
    // a symmetric (L,m,m) block on the wire: the tile-packed buffer (all channels / the rank's window) or the block itself
    auto rs_sym = [&](double* blk, double* xp) -> int {        // (inside a group)
        return pack ? rs(comm, xp, L * pe, G, stream) : rs(comm, blk, Lmm, G, stream);
    };
    auto ag_sym = [&](double* blk, double* xp) -> int {
        return pack ? ag(comm, xp, L * pe, G, stream) : ag(comm, blk, Lmm, G, stream);
    };
    RUN(svgp_mnist_encoder_kernel_matrix_fwd(&cc, theta, images, aux, ws, stream));
    // the channel-independent block of the forward factor stage ((K + jI)^-1, Kn Ki, q, W, P^T -- every rank computes it, and with
    // L / G channels per rank it is most of the stage) on the side branch from here on, beside the statistics, exchange point 1
    // and the window's channel inverses; joined where u = Ki mu needs it (as svgp_mnist_train_step does on one GPU, api.hip)
    const char* ev_k = getenv("SVGP_KONLY_BRANCH");
    const bool ksplit = fork && m < SVGP_CHOL_INVERSE_MIN_M && !(ev_k && ev_k[0] == '0');
    void* side0 = stream;
    if (ksplit) RUN(guard.fork(&side0));
    RUN(svgp_gp_stats_fwd(&cc, ws, stream));         // (issued first: the branch's 15 launches would hold the caller's stream back)
    if (ksplit) RUN(svgp_big_factor_fwd(&cc, wl, ws, side0, l0, nl, 5));
    // ---- point 1: reduce-scatter [S | v] over the channels
    RUN(pt.begin());
    if (pack) RUN(svgp_sym_pack(m, L, 0, ws + wl.S, xp0, stream));
    RUN(guard.begin_group());
    RUN(rs_sym(ws + wl.S, xp0));
    RUN(rs(comm, ws + wl.v, Lm, G, stream));
    RUN(guard.end_group());
    if (pack) RUN(svgp_sym_unpack(m, nl, xp0 + (size_t)l0 * pe, ws + wl.S + (size_t)l0 * mm, stream));
    RUN(pt.end());
    // window factor stage without its tail
    if (ksplit) {
        RUN(svgp_big_factor_fwd(&cc, wl, ws, stream, l0, nl, 6));
        RUN(guard.join());
        RUN(svgp_big_factor_fwd(&cc, wl, ws, stream, l0, nl, 7));
    } else {
        RUN(svgp_big_factor_fwd(&cc, wl, ws, stream, l0, nl, 1));
    }
    // ---- point 2: all-gather [Sigma^-1 | t | u]
    RUN(pt.begin());
    // the window goes to the wire format first, so that the side branch below never reads a block that is being rewritten
    // (Sigma^-1 is exactly symmetric in memory: its lower tiles ARE the matrix and the owner keeps its own window as it is)
    if (pack) RUN(svgp_sym_pack(m, nl, 0, ws + wl.Si + (size_t)l0 * mm, xp0 + (size_t)l0 * pe, stream));
    // the tail ((A_hat + jI)^-1, log det, KL) and the early half of the reverse factor stage: on the side branch, beside the
    // all-gather, the row stage, the networks and the reverse statistics.  The branch is forked here (it depends on the window
    // stage only) but its launches are ISSUED behind the collective: enqueued first, the branch's GEMMs fill every CU and the
    // collective's kernel waits for a slot -- with a 1-rank communicator the point measured 245 us at config 3 for a no-op
    // gather (round 3: 260 us), and the row stage on the caller's stream waits behind it.
    void* side = stream;
    if (fork) RUN(guard.fork(&side));
    RUN(guard.begin_group());
    RUN(ag_sym(ws + wl.Si, xp0));
    RUN(ag(comm, ws + wl.t, Lm, G, stream));
    RUN(ag(comm, ws + wl.u, Lm, G, stream));
    RUN(guard.end_group());
    if (pack) {                                  // the other ranks' windows (the branch reads the rank's own window only)
        const int hi0 = l0 + nl, nhi = L - hi0;
        RUN(svgp_sym_unpack(m, l0, xp0, ws + wl.Si, stream));
        RUN(svgp_sym_unpack(m, nhi, xp0 + (size_t)hi0 * pe, ws + wl.Si + (size_t)hi0 * mm, stream));
    }
    RUN(pt.end());
    // (the row stage is issued first: the branch's ~25 launches take the host ~100 us to enqueue, during which the caller's stream
    // would have nothing to run; the branch has that much slack)
    RUN(svgp_gp_posterior_fwd(&cc, eps, ws, state, stream));
    RUN(svgp_big_factor_fwd(&cc, wl, ws, side, l0, nl, 2));
    if (fork) RUN(svgp_big_factor_bwd(&cc, wl, ws, state, side, l0, nl, 1));
    RUN(svgp_mnist_decoder_fwd(&cc, theta, images, ws, stream));
    RUN(svgp_mnist_decoder_bwd(&cc, theta, images, ws, state, stream));
    RUN(svgp_gp_stats_bwd(&cc, ws, state, stream));
    // ---- point 3: reduce-scatter [A2 | ud | td]
    RUN(pt.begin());
    if (pack) RUN(svgp_sym_pack(m, L, 0, ws + wl.A2, xp0, stream));
    RUN(guard.begin_group());
    RUN(rs_sym(ws + wl.A2, xp0));
    RUN(rs(comm, ws + wl.ud, Lm, G, stream));
    RUN(rs(comm, ws + wl.td, Lm, G, stream));
    RUN(guard.end_group());
    if (pack) RUN(svgp_sym_unpack(m, nl, xp0 + (size_t)l0 * pe, ws + wl.A2 + (size_t)l0 * mm, stream));
    RUN(pt.end());
    if (fork) {
        RUN(guard.join());
        // round 6 (as svgp_mnist_train_step does on one GPU, api.hip): the single-matrix chain of the gradient of Ki -- five small launches
        // that every rank runs in full, while the channel block covers its L / G channels only -- on the branch that has just been
        // joined, beside the channel block.  SVGP_KBAR_BRANCH=0: one launch after the other.
        const char* ev_kb = getenv("SVGP_KBAR_BRANCH");
        if (!(ev_kb && ev_kb[0] == '0')) {
            RUN(svgp_big_factor_bwd(&cc, wl, ws, state, stream, l0, nl, 6));
            void* side2 = stream;
            RUN(guard.fork(&side2));
            RUN(svgp_big_factor_bwd(&cc, wl, ws, state, side2, l0, nl, 9));
            RUN(svgp_big_factor_bwd(&cc, wl, ws, state, stream, l0, nl, 8));
            RUN(guard.join());
            RUN(svgp_big_factor_bwd(&cc, wl, ws, state, stream, l0, nl, 10));
        } else
        RUN(svgp_big_factor_bwd(&cc, wl, ws, state, stream, l0, nl, 2));
    } else {
        RUN(svgp_big_factor_bwd(&cc, wl, ws, state, stream, l0, nl, 0));
    }
    // ---- point 4: all-gather [Ssym | vbar | KL]
    RUN(pt.begin());
    if (pack) RUN(svgp_sym_pack(m, nl, 0, ws + wl.Ssym + (size_t)l0 * mm, xp0 + (size_t)l0 * pe, stream));
    RUN(guard.begin_group());
    RUN(ag_sym(ws + wl.Ssym, xp0));
    RUN(ag(comm, ws + wl.vbar, Lm, G, stream));
    RUN(ag(comm, ws + wl.KL, L, G, stream));
    RUN(guard.end_group());
    if (pack) RUN(svgp_sym_unpack(m, L, xp0, ws + wl.Ssym, stream));
    RUN(pt.end());
    RUN(svgp_gp_posterior_bwd(&cc, ws, state, stream));
    RUN(svgp_kernel_matrix_bwd_partials(&cc, theta, aux, ws, stream));
    RUN(svgp_mnist_encoder_bwd(&cc, theta, images, ws, stream));
    RUN(svgp_mnist_grad_reduce_all(&cc, aux, ws, stream));
    // ---- point 5: gradients + scalar sums
    RUN(pt.begin());
    RUN(svgp_allreduce_sum_f64(comm, ws + wl.gradC, wl.gradC_len, stream));
    RUN(pt.end());
    RUN(svgp_mnist_step_phase_deferred(&cc, 3, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
#undef RUN
    return SVGP_OK;
}
