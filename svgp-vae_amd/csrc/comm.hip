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

struct Comm {
    ncclComm_t comm;
    int rank, nranks;
};

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

// One data-parallel step, everything enqueued on `stream`:
//   phase 0 | all-reduce statA | phase 1 | all-reduce statB | phase 2 | all-reduce gradC | phase 3
// c->b is this rank's row count, c->b_global the global batch, c->rep_weight 1 on exactly one rank.
//
// Large-m path (m > 64) with L divisible by the rank count (Hensman branch): the channel-sharded schedule instead
//  -- the (L,m,m) statistics are reduce-SCATTERED over the channels, every rank factors its L / G channels
// (svgp_gp_factor_*_channels) and what the row stages need is all-gathered:
//   encoder + kernel matrices + statistics | reduce-scatter S, v | window factor stage | all-gather Sigma^-1, M2, t, u, KL |
//   row stage, decoder fwd + bwd, backward statistics | reduce-scatter A2, ud, td | window reverse factor stage |
//   all-gather Ssym, vbar (Kn Q is formed from Kn Ssym and the forward pass's Kn M2: Q itself is not exchanged) |
//   row gradients, kernel-matrix VJP (every rank's Kbar share counts), encoder reverse pass,
//   gradient reduction | all-reduce gradC | phase 3
// At config 3 on 8 ranks: 2 channels of 256 x 256 per rank instead of 16, 8.4 MB blocks moved as 7/8 of their size.
namespace {
int rs(void* comm, double* p, int64_t total, int nranks, void* stream) {
    return svgp_reduce_scatter_sum_f64(comm, p, total / nranks, stream);
}
int ag(void* comm, double* p, int64_t total, int nranks, void* stream) { return svgp_allgather_f64(comm, p, total / nranks, stream); }
}  // namespace

extern "C" int svgp_mnist_train_step_dp(const svgp_mnist_cfg* c, void* comm, double* theta, const double* images,
                                        const double* aux, const double* eps, double* ws, double* state,
                                        double* adam_m, double* adam_v, void* stream) {
    SVGP_REQUIRE(c && comm, SVGP_ERR_INVALID, "NULL cfg / communicator");
    svgp_mnist_ws_layout wl;
    int rc = svgp_mnist_ws_layout_get(c, &wl);
    if (rc) return rc;
    const Comm* cm = (const Comm*)comm;
    const int G = cm->nranks, L = c->L, m = c->m;
    const bool sharded = m > SVGP_M_MAX && L % G == 0 && !c->titsias && !c->kl_form;
    if (!sharded) {
        const int64_t off[3] = {wl.statA, wl.statB, wl.gradC}, len[3] = {wl.statA_len, wl.statB_len, wl.gradC_len};
        for (int ph = 0; ph < 4; ++ph) {
            rc = svgp_mnist_step_phase_deferred(c, ph, theta, images, aux, eps, ws, state, adam_m, adam_v, stream);
            if (rc) return rc;
            if (ph < 3) {
                rc = svgp_allreduce_sum_f64(comm, ws + off[ph], len[ph], stream);
                if (rc) return rc;
            }
        }
        return SVGP_OK;
    }
    SVGP_REQUIRE(theta && images && aux && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    svgp_mnist_cfg cc = *c;
    cc.rep_weight = 1.0;                         // Kbar = this rank's channel-window share: every share counts
    const int nl = L / G, l0 = cm->rank * nl;
    const int64_t mm = (int64_t)m * m, Lmm = (int64_t)L * mm, Lm = (int64_t)L * m;
#define RUN(call) do { rc = (call); if (rc) return rc; } while (0)
    RUN(svgp_mnist_encoder_kernel_matrix_fwd(&cc, theta, images, aux, ws, stream));
    RUN(svgp_gp_stats_fwd(&cc, ws, stream));
    RUN(rs(comm, ws + wl.S, Lmm, G, stream));
    RUN(rs(comm, ws + wl.v, Lm, G, stream));
    RUN(svgp_gp_factor_fwd_channels(&cc, l0, nl, ws, stream));
    RUN(ag(comm, ws + wl.Si, Lmm, G, stream));
    RUN(ag(comm, ws + wl.M2, Lmm, G, stream));
    RUN(ag(comm, ws + wl.t, Lm, G, stream));
    RUN(ag(comm, ws + wl.u, Lm, G, stream));
    RUN(ag(comm, ws + wl.KL, L, G, stream));
    RUN(svgp_gp_posterior_fwd(&cc, eps, ws, state, stream));
    RUN(svgp_mnist_decoder_fwd(&cc, theta, images, ws, stream));
    RUN(svgp_mnist_decoder_bwd(&cc, theta, images, ws, state, stream));
    RUN(svgp_gp_stats_bwd(&cc, ws, state, stream));
    RUN(rs(comm, ws + wl.A2, Lmm, G, stream));
    RUN(rs(comm, ws + wl.ud, Lm, G, stream));
    RUN(rs(comm, ws + wl.td, Lm, G, stream));
    RUN(svgp_gp_factor_bwd_channels(&cc, l0, nl, ws, state, stream));
    RUN(ag(comm, ws + wl.Ssym, Lmm, G, stream));
    RUN(ag(comm, ws + wl.vbar, Lm, G, stream));
    RUN(svgp_gp_posterior_bwd(&cc, ws, state, stream));
    RUN(svgp_kernel_matrix_bwd_partials(&cc, theta, aux, ws, stream));
    RUN(svgp_mnist_encoder_bwd(&cc, theta, images, ws, stream));
    RUN(svgp_mnist_grad_reduce_all(&cc, aux, ws, stream));
    RUN(svgp_allreduce_sum_f64(comm, ws + wl.gradC, wl.gradC_len, stream));
    RUN(svgp_mnist_step_phase_deferred(&cc, 3, theta, images, aux, eps, ws, state, adam_m, adam_v, stream));
#undef RUN
    return SVGP_OK;
}
