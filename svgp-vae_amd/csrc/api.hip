// C-ABI glue: error strings, layouts, phase orchestration, HIP graph / event helpers.
#include <cstdarg>

#include "common.hpp"
#include <map>
#include <mutex>
#include <set>
#include <utility>
#include <cstdlib>

static thread_local char g_err[512] = "";

void svgp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* svgp_last_error(void) { return g_err; }
extern "C" int svgp_version(void) { return SVGP_VERSION_MAJOR * 100 + SVGP_VERSION_MINOR; }
extern "C" int svgp_struct_sizeof(int which) {
    switch (which) {
    case 0: return (int)sizeof(svgp_mnist_cfg);
    case 1: return (int)sizeof(svgp_mnist_param_layout);
    case 2: return (int)sizeof(svgp_mnist_ws_layout);
    case 3: return (int)sizeof(svgp_stream_kdesc);
    case 4: return (int)sizeof(svgp_conv_desc);
    case 5: return (int)sizeof(svgp_sprites_kcfg);
    case 6: return (int)sizeof(svgp_pearce_bufs);
    case 7: return (int)sizeof(svgp_sum_job);
    default: return -1;
    }
}

int svgp_check_cfg(const svgp_mnist_cfg* c) {
    SVGP_REQUIRE(c != nullptr, SVGP_ERR_INVALID, "cfg is NULL");
    SVGP_REQUIRE(c->b >= 1 && c->b_global >= c->b, SVGP_ERR_INVALID, "need 1 <= b <= b_global (b=%d b_global=%d)",
                 c->b, c->b_global);
    SVGP_REQUIRE(c->m >= 1 && c->L >= 1 && c->M >= 1 && c->n_obj >= 0, SVGP_ERR_INVALID,
                 "bad shape m=%d L=%d M=%d n_obj=%d", c->m, c->L, c->M, c->n_obj);
    SVGP_REQUIRE(c->m <= SVGP_M_LIMIT, SVGP_ERR_UNSUPPORTED,
                 "m=%d inducing points: this build supports m <= %d", c->m, SVGP_M_LIMIT);
    SVGP_REQUIRE(c->M <= 128, SVGP_ERR_UNSUPPORTED, "M=%d: object-vector dimension > 128 not supported", c->M);
    SVGP_REQUIRE(c->L <= 64, SVGP_ERR_UNSUPPORTED, "L=%d: more than 64 latent channels not supported", c->L);
    SVGP_REQUIRE(c->N_train > 0 && c->jitter >= 0, SVGP_ERR_INVALID, "bad N_train / jitter");
    SVGP_REQUIRE(c->kl_form == 0 || c->kl_form == 1, SVGP_ERR_INVALID, "kl_form=%d (0 or 1)", c->kl_form);
    SVGP_REQUIRE(c->clip_pv >= 0 && c->clip_pv <= 2, SVGP_ERR_INVALID, "clip_pv=%d (0, 1 or 2)", c->clip_pv);
    SVGP_REQUIRE(c->gemm_f32 >= 0 && c->gemm_f32 <= 2, SVGP_ERR_INVALID, "gemm_f32=%d (0, 1 or 2)", c->gemm_f32);
    SVGP_REQUIRE(!(c->kl_form && c->m > SVGP_M_MAX), SVGP_ERR_UNSUPPORTED,
                 "kl_form=1 (moving-ball SVGP) is implemented for m <= %d inducing points (m=%d)", SVGP_M_MAX, c->m);
    SVGP_REQUIRE(!(c->kl_form && c->b != c->b_global), SVGP_ERR_UNSUPPORTED,
                 "kl_form=1 couples all channels of the batch; it is not sharded over ranks");
    return SVGP_OK;
}

extern "C" int svgp_mnist_param_layout_get(const svgp_mnist_cfg* c, svgp_mnist_param_layout* o) {
    int rc = svgp_check_cfg(c);
    if (rc) return rc;
    SVGP_REQUIRE(o != nullptr, SVGP_ERR_INVALID, "out is NULL");
    int64_t p = 0;
    auto take = [&](int64_t n) { int64_t r = p; p += n; return r; };
    o->enc_c1_w = take(3 * 3 * 1 * 8);  o->enc_c1_b = take(8);
    o->enc_c2_w = take(3 * 3 * 8 * 8);  o->enc_c2_b = take(8);
    o->enc_c3_w = take(3 * 3 * 8 * 8);  o->enc_c3_b = take(8);
    o->enc_d_w = take(32 * 2 * c->L);   o->enc_d_b = take(2 * c->L);
    o->n_enc = p;
    o->dec_d_w = take((int64_t)c->L * 128); o->dec_d_b = take(128);
    o->dec_c1_w = take(3 * 3 * 8 * 8);  o->dec_c1_b = take(8);
    o->dec_c2_w = take(3 * 3 * 8 * 8);  o->dec_c2_b = take(8);
    o->dec_c3_w = take(3 * 3 * 8 * 1);  o->dec_c3_b = take(1);
    o->n_vae = p;
    o->ip = take((int64_t)c->m * (2 + c->M));
    o->l_GP = take(1);
    o->amplitude = take(1);
    o->ov = take((int64_t)c->n_obj * c->M);
    o->n_total = p;
    return SVGP_OK;
}

extern "C" int svgp_mnist_ws_layout_get(const svgp_mnist_cfg* c, svgp_mnist_ws_layout* o) {
    svgp_mnist_param_layout pl;
    int rc = svgp_mnist_param_layout_get(c, &pl);
    if (rc) return rc;
    SVGP_REQUIRE(o != nullptr, SVGP_ERR_INVALID, "out is NULL");
    SVGP_REQUIRE(c->b_cap == 0 || c->b_cap >= c->b, SVGP_ERR_INVALID, "b_cap=%d < b=%d", c->b_cap, c->b);
    svgp_mnist_cfg cc = *c;              // layout is a function of the capacity, not of the current rows
    cc.b = c->b_cap > 0 ? c->b_cap : c->b;
    const int64_t b = cc.b, m = c->m, L = c->L, M = c->M;
    int64_t p = 0;
    // every field starts on a 16-element (128-byte) boundary
    auto take = [&](int64_t n) { int64_t r = p; p += (n + 15) / 16 * 16; return r; };
    o->enc_a1 = take(b * 13 * 13 * 8); o->enc_a2 = take(b * 6 * 6 * 8); o->enc_a3 = take(b * 32);
    o->qnet_mu = take(b * L); o->qnet_var_raw = take(b * L); o->qnet_var = take(b * L);
    // m > 64: the rows W = Kn Ki K (gp_large.hip, "W form") sit right behind the CURRENT b rows of K_nm, so that [Kn; W] is
    // one (2 b, m) operand of the row-stage product and of the reverse statistic
    o->K = take(m * m); o->Kn = take(m > SVGP_M_MAX ? 2 * b * m : b * m); o->knn = take(b);
    const int64_t P = svgp_stat_parts(c);      // capacity from b_cap
    o->stat_parts = P;
    o->statA = p; o->S = p; p += P * L * m * m; o->v = p; p += P * L * m;
    o->tit_S2 = p; if (c->titsias) p += L * m * m;
    o->tit_v2 = p; if (c->titsias) p += L * m;
    o->statA_len = p - o->statA; take(0);
    // Si (L,m,m) and Ki (m,m) form ONE contiguous (L+1, m, m) block whenever that keeps Ki on a 16-element boundary
    // (m a multiple of 4): the large-m path then inverts them as one batch
    o->ldK = take(1);
    o->Si = take(L * m * m + ((L * m * m) % 16 == 0 ? m * m : 0));
    o->Ki = (L * m * m) % 16 == 0 ? o->Si + L * m * m : take(m * m);
    o->t = take(L * m); o->G = take(L * m * m); o->A = take(L * m * m);
    // (M2 = Ki A Ki and Qm exist on the LDS-resident path only: the large-m path evaluates k^T M2 k as w^T Si w)
    o->Aji = take(L * m * m); o->mu_hat = take(L * m); o->u = take(L * m); o->M2 = take(m > SVGP_M_MAX ? 0 : L * m * m);
    o->KL = take(2 * L); o->q = take(b);   // [KL | kl_form 1: tr(Ki A_l A_l)]
    o->p_m = take(b * L); o->p_v = take(b * L); o->e = take(b * L); o->d = take(b * L);
    o->eps = take(b * L); o->z = take(b * L);
    o->dec_h0 = take(b * 128); o->dec_a1 = take(b * 512); o->dec_a2 = take(b * 1568); o->recon = take(b * 784);
    o->dec_d2 = take(b * 1568); o->dec_d1 = take(b * 512); o->dec_dh0 = take(b * 128); o->dec_weff = take(2176);
    o->flags = take(64);
    o->zbar = take(b * L); o->g_pv = take(b * L); o->g_pm = take(b * L); o->mvbar = take(b * L);
    o->statB = p; o->A2 = p; p += P * L * m * m; o->ud = p; p += P * L * m; o->td = p; p += P * L * m;
    o->statB_len = p - o->statB; take(0);
    o->Kbar = take(m * m); o->fb_part = take(2 * L * m * m); o->Qm = take(m > SVGP_M_MAX ? 0 : L * m * m); o->vbar = take(L * m);
    o->Ssym = take(L * m * m); o->Knbar_part = take(L * b * m);
    // scratch of the large-m path (gp_large.hip)
    // m > 64: (L,b,m) scratch + the forward products [Kn; W] Si_l (L,2b,m), Kn Ki (b,m) kept for the reverse pass + X (b,2m)
    o->scr_bm = take(m > SVGP_M_MAX ? 3 * L * b * m + 3 * b * m : L * b * m); o->scr_mm = take(4 * L * m * m);
    o->scr_vec = take(3 * L * m + 3 * L + 16 + b);
    o->scr_inv = take((int64_t)svgp_spd_inverse_workspace_elems((int)m, (int)L + 1)); o->scr_bl = take(2 * b * L);
    // m > 64: ten channel-independent m x m matrices + the split-K scratch of the (2 m, m, b) row contraction of the W form (K Ki, Kn^T Wbar, Kn^T diag(qbar) Kn, channel sums, temporaries)
    o->scr_sm = take(m > SVGP_M_MAX ? 10 * m * m + svgp_dgemm_splitk_scratch_elems((int)(2 * m), (int)m, (int)b) : 0);
    o->Knbar = take(b * m); o->knnbar = take(b); o->ybar = take(b * L); o->s2bar = take(b * L);
    o->d_on = take(b * M);
    o->n_part = svgp_n_part(&cc);
    o->part_dec = take(o->n_part * (pl.n_vae - pl.n_enc));
    o->part_enc = take(o->n_part * pl.n_enc);
    o->part_gp = take((m + svgp_n_postblk(&cc)) * 2);
    o->n_post = (int64_t)L * svgp_n_postblk(&cc);
    o->part_sums = take(o->n_part * 4 + o->n_post * 2);
    o->gradC = p; o->grad = p; p += pl.n_total; o->sums = p; p += 8; o->gradC_len = p - o->gradC; take(0);
    o->tit_Si = take(c->titsias ? L * m * m : 0); o->tit_t = take(c->titsias ? L * m : 0);
    o->tit_scal = take(c->titsias ? 2 * L + 1 : 0);
    // the wire buffer of the channel-sharded exchange: one tile-packed (L, m, m) block; only when the batch is sharded over
    // ranks (cfg.single_stat_block, set by the engines for world sizes > 1): a single-GPU workspace does not carry it
    o->xpack_len = (m > SVGP_M_MAX && c->single_stat_block) ? L * svgp_sym_packed_elems((int)m) : 0;
    o->xpack = take(o->xpack_len);
    o->total = p;
    return SVGP_OK;
}

// ---------------------------------------------------------------------------------------------
// phases
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Side streams.  Most stages of the step are short dependent launches that fill only a part of the chip
// (the GP factor kernels run L workgroups on 256 CUs), so work that is OFF the critical path -- the
// kernel-matrix reverse pass and the VAE weight gradients -- is forked onto a library-owned side stream
// and joined where its result is consumed.  hipEventRecord / hipStreamWaitEvent are capturable, so under
// stream capture the fork becomes a parallel branch of the hipGraph.
// ---------------------------------------------------------------------------------------------
namespace {
// One Side per (device, caller stream): two engines that enqueue on different streams (or from different threads) never
// share a fork event, a join decision or a side stream.  `open[k]` (branch k forked and not yet joined) and `early_ws`
// (workspaces whose early reverse factor half was issued by phase 1 and not yet consumed by phase 2) are only touched under
// the Side's own mutex, which fork / join hold for the whole record + wait pair.
struct Side {
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
    bool open[2] = {false, false};
    std::set<const void*> early_ws;
    std::set<const void*> konly_ws;      // workspaces whose channel-independent factor block was issued by phase 0 on branch 1
    std::mutex mu;
};
std::mutex g_side_mu;
std::map<std::pair<int, hipStream_t>, Side*> g_side;

// ---- do two streams run concurrently?  HIP maps streams onto a small pool of hardware queues (GPU_MAX_HW_QUEUES, 4 by default)
// in creation order; two streams that share a queue execute one after the other whatever events say.  Which streams collide
// depends on every stream the process has created so far -- torch's, RCCL's, ours: measured round 4, config 3 with a 1-rank RCCL
// communicator: the library's side branch shared the caller's queue and hid NOTHING (1.454 ms against 1.313 ms with 8 queues), while
// the SPRITES step lost 5 ms with 8 queues for the same reason in the other direction.  So the side streams are chosen by a
// probe instead of by luck: a ~40 us single-workgroup spin on stream a, an empty kernel on stream b issued right behind it; b
// runs beside a iff its kernel finishes well before the spin does.
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_noop() {}
int streams_overlap(hipStream_t a, hipStream_t b, bool* out) {
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    SVGP_CHECK_HIP(hipEventCreate(&e0)); SVGP_CHECK_HIP(hipEventCreate(&ea)); SVGP_CHECK_HIP(hipEventCreate(&eb));
    int votes = 0;
    for (int rep = 0; rep < 3; ++rep) {
        SVGP_CHECK_HIP(hipStreamSynchronize(a)); SVGP_CHECK_HIP(hipStreamSynchronize(b));
        SVGP_CHECK_HIP(hipEventRecord(e0, a));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 4000LL);        // 100 MHz wall clock: 40 us
        SVGP_CHECK_HIP(hipEventRecord(ea, a));
        hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, b);
        SVGP_CHECK_HIP(hipEventRecord(eb, b));
        SVGP_CHECK_HIP(hipEventSynchronize(ea)); SVGP_CHECK_HIP(hipEventSynchronize(eb));
        float ta = 0, tb = 0;
        SVGP_CHECK_HIP(hipEventElapsedTime(&ta, e0, ea));
        SVGP_CHECK_HIP(hipEventElapsedTime(&tb, e0, eb));
        if (tb < 0.6f * ta) ++votes;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    *out = votes >= 2;
    return SVGP_OK;
}

int side_get(hipStream_t main, Side** out) {
    int dev = 0;
    SVGP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto key = std::make_pair(dev, main);
    auto it = g_side.find(key);
    if (it == g_side.end()) {
        Side* sd = new Side();
        // Two side streams that run concurrently with the caller's stream AND with each other, picked from up to 12 candidates by
        // the probe above.  Not under stream capture (the probe synchronises) and not with SVGP_STREAM_PROBE=0: then the first two.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(main, &cap);
        const char* ev = getenv("SVGP_STREAM_PROBE");
        const bool probe = cap == hipStreamCaptureStatusNone && !(ev && ev[0] == '0');
        hipStream_t cand[12];
        int n_cand = 0, picked = 0;
        while (picked < 2 && n_cand < (probe ? 12 : 2)) {
            hipStream_t c = nullptr;
            SVGP_CHECK_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
            cand[n_cand++] = c;
            bool ok = true;
            if (probe) {
                int rc = streams_overlap(main, c, &ok);
                if (rc) return rc;
                if (ok && picked == 1) { rc = streams_overlap(sd->s[0], c, &ok); if (rc) return rc; }
            }
            if (ok) sd->s[picked++] = c;
        }
        for (int i = 0; i < n_cand; ++i) {           // candidates that collide with the caller's stream (or with each other) go back
            if (cand[i] == sd->s[0] || cand[i] == sd->s[1]) continue;
            if (picked < 2) sd->s[picked++] = cand[i];                      // fewer than two concurrent queues exist: take what there is
            else (void)hipStreamDestroy(cand[i]);
        }
        for (int k = 0; k < 2; ++k) SVGP_CHECK_HIP(hipEventCreateWithFlags(&sd->done[k], hipEventDisableTiming));
        SVGP_CHECK_HIP(hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming));
        it = g_side.emplace(key, sd).first;
    }
    *out = it->second;
    return SVGP_OK;
}
// side stream k continues after everything issued on `main` so far
int side_fork(Side* sd, int k, hipStream_t main) {
    std::lock_guard<std::mutex> lk(sd->mu);
    SVGP_CHECK_HIP(hipEventRecord(sd->fork, main));
    SVGP_CHECK_HIP(hipStreamWaitEvent(sd->s[k], sd->fork, 0));
    sd->open[k] = true;
    return SVGP_OK;
}
// `main` continues after everything issued on side stream k (no-op when this Side's branch k is not open)
int side_join(Side* sd, int k, hipStream_t main) {
    std::lock_guard<std::mutex> lk(sd->mu);
    if (!sd->open[k]) return SVGP_OK;
    SVGP_CHECK_HIP(hipEventRecord(sd->done[k], sd->s[k]));
    SVGP_CHECK_HIP(hipStreamWaitEvent(main, sd->done[k], 0));
    sd->open[k] = false;
    return SVGP_OK;
}
void side_mark_early(Side* sd, const void* ws) { std::lock_guard<std::mutex> lk(sd->mu); sd->early_ws.insert(ws); }
bool side_take_early(Side* sd, const void* ws) { std::lock_guard<std::mutex> lk(sd->mu); return sd->early_ws.erase(ws) > 0; }
void side_mark_konly(Side* sd, const void* ws) { std::lock_guard<std::mutex> lk(sd->mu); sd->konly_ws.insert(ws); }
bool side_take_konly(Side* sd, const void* ws) { std::lock_guard<std::mutex> lk(sd->mu); return sd->konly_ws.erase(ws) > 0; }

}  // namespace

// comm.hip (channel-sharded step, branch 1) / cholesky.hip (look-ahead of the blocked factorisation, branch 0): side branch k
// of the caller's stream.  _fork: the branch continues after everything issued on the caller's stream so far (re-forking an
// open branch just adds that dependency); _join: the caller's stream continues after everything issued on the branch.
int svgp_side_branch_fork(void* main_stream, void** side_stream_out, int k) {
    Side* sd = nullptr;
    int rc = side_get((hipStream_t)main_stream, &sd);
    if (rc) return rc;
    rc = side_fork(sd, k, (hipStream_t)main_stream);
    if (rc) return rc;
    *side_stream_out = (void*)sd->s[k];
    return SVGP_OK;
}
int svgp_side_branch_join(void* main_stream, int k) {
    Side* sd = nullptr;
    int rc = side_get((hipStream_t)main_stream, &sd);
    if (rc) return rc;
    return side_join(sd, k, (hipStream_t)main_stream);
}

namespace {
// m <= 64 (round 6): the decoder's reverse pass as the data half (the chain to zbar) in phase 1 and the weight half as riders of
// the reverse factor launch in phase 2 (vae_dev.hpp).  SVGP_DEC_SPLIT=0: the one-kernel form of rounds 1-5.
// Read per call (tests compare the two forms in one process); a caller must not change it between phase 1 and phase 2 of a step.
bool dec_split_on() {
    const char* e = getenv("SVGP_DEC_SPLIT");
    return !(e && e[0] == '0');
}
// m <= 64 (round 6): kernel-matrix VJP + encoder reverse pass in one launch.  SVGP_ENC_KM_MERGE=0: two launches.
bool enc_km_merge_on() {
    const char* e = getenv("SVGP_ENC_KM_MERGE");
    return !(e && e[0] == '0');
}
bool sum_merge_on() {
    const char* e = getenv("SVGP_SUM_MERGE");
    return !(e && e[0] == '0');
}
// m <= 64, phases issued back to back with nothing exchanged in between (round 6): the reverse statistics ride in the reverse factor
// launch (svgp_gp_stats_factor_bwd_wgrad).  SVGP_STAT_MERGE=0: their own launch at the end of phase 1.
bool stat_merge_on() {
    const char* e = getenv("SVGP_STAT_MERGE");
    return !(e && e[0] == '0');
}
// m <= 32 (round 6): the deferred (A_hat + jI)^-1 rides in the decoder's data-reverse launch instead of the forward row-stage launch
// (svgp_mnist_decoder_bwd_data_pre_aji).  SVGP_AJI_DEC=0: svgp_gp_posterior_fwd_with_aji.
bool aji_dec_on() {
    const char* e = getenv("SVGP_AJI_DEC");
    return !(e && e[0] == '0');
}
bool kbar_branch_on() {
    const char* e = getenv("SVGP_KBAR_BRANCH");
    return !(e && e[0] == '0');
}
bool konly_on() {      // (read per call: tests compare the two orders in one process)
    const char* e = getenv("SVGP_KONLY_BRANCH");
    return !(e && e[0] == '0');
}
// `defer`: the caller issues all four phases back to back on one stream (svgp_mnist_train_step), so a
// branch forked in one phase may be joined in a later one; otherwise every phase joins before returning
// (each phase may be captured into its own graph, with a collective in between).  defer == 2: ... and NOTHING is exchanged
// between the phases (the single-GPU step), so a stage may move across a phase boundary.
int step_phase_impl(const svgp_mnist_cfg* c, int phase, double* theta, const double* images, const double* aux,
                    const double* eps, double* ws, double* state, double* adam_m, double* adam_v, void* stream,
                    int defer) {
    int rc = svgp_check_cfg(c);
    if (rc) return rc;
    SVGP_REQUIRE(theta && images && aux && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(phase >= 0 && phase <= 5, SVGP_ERR_INVALID, "phase %d out of range 0..5", phase);
    hipStream_t ms = (hipStream_t)stream;
    // Measured on MI355X (tools/fork_probe.py): a fork + join costs ~10 us of cross-stream signalling.  The
    // kernel-matrix reverse pass || encoder reverse pass branch hides ~20 us, which pays only in the
    // per-phase-graph replay form (phase 2: 110 -> 100 us) and loses in the eager in-order form
    // (261 -> 283 us per step), so it is opt-in: SVGP_SIDE_STREAMS=1.
    const char* fk = getenv("SVGP_SIDE_STREAMS");
    const bool fork2 = fk && fk[0] == '1';
    // Large-m path: the tail of the forward factor stage ((A_hat + jI)^-1, its log det, KL: a whole batched inverse that only
    // the reverse factor stage and the final ELBO need) runs on side stream 1, beside the row stage, the decoder and the
    // reverse statistics.  That hides ~140 us at config 3 for ~10 us of signalling, so it is on unless SVGP_SIDE_STREAMS=0.
    // Not with cfg.titsias: svgp_gp_titsias_fwd inverts its own batch through the SAME inverse scratch (ws.scr_inv) on the
    // caller's stream, and the early reverse half would only multiply zero seeds.
    const bool large = c->m > SVGP_M_MAX;
    const bool fork1 = large && !c->titsias && !(fk && fk[0] == '0');
    Side* sd = nullptr;
    if (fork2 || large) {
        rc = side_get(ms, &sd);
        if (rc) return rc;
    }
    hipStream_t s2 = fork2 ? sd->s[0] : ms;
#define RUN(call) do { rc = (call); if (rc) return rc; } while (0)
    const bool ksplit = phase == 0 && fork1 && defer && c->m < SVGP_CHOL_INVERSE_MIN_M && konly_on();
    const bool sum_rides = phase == 2 && !large && !fork2 && !c->titsias && enc_km_merge_on() && sum_merge_on();
    const bool aji_in_dec = phase == 1 && c->m <= 32 && !c->titsias && dec_split_on() && aji_dec_on();
    const bool stat_rides = defer == 2 && !large && !c->titsias && c->L <= 56 && dec_split_on() && stat_merge_on();
    switch (phase) {
    case 0:
        RUN(svgp_mnist_encoder_kernel_matrix_fwd(c, theta, images, aux, ws, stream));   // one launch for the two
        // 64 < m < 512, phases issued back to back (round 5): everything of the forward factor stage that is a function of the
        // KERNEL MATRICES alone -- (K + jI)^-1 and its log det, Kn Ki, q, W = (Kn Ki) K, P^T = K Ki: one single-matrix blocked inverse
        // (a chain of 8 block steps, as long as the channel batch's) and three products -- goes to side branch 1 now, beside the
        // forward statistics and the channel inverses, instead of behind them on the caller's stream.  Joined in phase 1 where
        // u = Ki mu needs it.  SVGP_KONLY_BRANCH=0: the in-line order.
        // (The branch is FORKED here but its launches are ISSUED behind the statistics': the host -- and a replayed graph, which
        // submits its nodes in capture order -- takes ~2.5 us per launch, and the branch's 15 launches in front of the statistics'
        // first kernel left the caller's stream idle for 36 us in the kernel trace.)
        if (ksplit) RUN(side_fork(sd, 1, ms));
        RUN(svgp_gp_stats_fwd(c, ws, stream));
        if (ksplit) {
            RUN(svgp_gp_factor_fwd_part(c, ws, (void*)sd->s[1], 5));
            side_mark_konly(sd, ws);
        }
        if (c->titsias) RUN(svgp_gp_titsias_stats(c, ws, stream));
        break;
    case 1:
        if (large && side_take_konly(sd, ws)) {                // the channel block; then what needs the branch's (K + jI)^-1 too
            RUN(svgp_gp_factor_fwd_part(c, ws, stream, 6));
            RUN(side_join(sd, 1, ms));
            RUN(svgp_gp_factor_fwd_part(c, ws, stream, 7));
        } else
        RUN(svgp_gp_factor_fwd_defer_aji(c, ws, stream));      // m <= 64: (A_hat + jI)^-1 finishes inside the row-stage launch
        // m > 64: the tail of the stage and the early half of the REVERSE factor stage (no reverse statistic needed; phase 2 then
        // runs the late half only) go to the side stream.  The branch is FORKED here but ISSUED behind the row stage: its ~25
        // launches take the host ~100 us to enqueue, during which the caller's stream had nothing to run (config 3, kernel trace of
        // round 4: a 101 us hole in front of the row stage's product) -- the branch has that much slack, the caller's stream none.
        if (c->m > SVGP_M_MAX && fork1) RUN(side_fork(sd, 1, ms));
        if (aji_in_dec) RUN(svgp_gp_posterior_fwd(c, eps, ws, state, stream));
        else RUN(svgp_gp_posterior_fwd_with_aji(c, eps, ws, state, stream));
        if (c->m > SVGP_M_MAX) {
            RUN(svgp_gp_factor_fwd_aji_tail(c, ws, fork1 ? (void*)sd->s[1] : stream));
            if (fork1) {
                RUN(svgp_gp_factor_bwd_early(c, ws, state, (void*)sd->s[1]));
                side_mark_early(sd, ws);                       // phase 2 of THIS workspace may run the late half only
            }
        }
        if (c->titsias) RUN(svgp_gp_titsias_fwd(c, ws, state, stream));
        // m <= 64 with the split on: the `_pre` forms read the effective up-convolution weights phase 0 of this step left in ws.dec_weff
        if (!large && dec_split_on()) {
            RUN(svgp_mnist_decoder_fwd_pre(c, theta, images, ws, stream));
            if (aji_in_dec) RUN(svgp_mnist_decoder_bwd_data_pre_aji(c, theta, images, ws, state, stream));
            else RUN(svgp_mnist_decoder_bwd_data_pre(c, theta, images, ws, state, stream));
        } else {
            RUN(svgp_mnist_decoder_fwd(c, theta, images, ws, stream));
            RUN(svgp_mnist_decoder_bwd(c, theta, images, ws, state, stream));
        }
        if (!stat_rides) RUN(svgp_gp_stats_bwd(c, ws, state, stream));      // (else: at the head of phase 2's first launch)
        if (fork1 && !defer) RUN(side_join(sd, 1, ms));        // phase-at-a-time callers: joined before the phase returns
        break;
    case 2:
    case 4:     // phase 2 up to and including the kernel-matrix reverse pass + gradient reduction part 1 (cfg.split_grad_exchange)
    case 5:     // ... the encoder's reverse pass + gradient reduction part 2
        if (phase == 5) {
            RUN(svgp_mnist_encoder_bwd(c, theta, images, ws, stream));
            RUN(svgp_mnist_grad_reduce_part(c, aux, ws, 2, stream));
            break;
        }
        // The late half alone is valid only if phase 1 of this library issued the early half on this workspace (recorded per
        // workspace, not inferred from the environment): a caller that ran the phase-1 stages through the individual entry
        // points, or changed SVGP_SIDE_STREAMS in between, gets the full reverse factor stage.
        if (large && side_take_early(sd, ws)) {
            RUN(svgp_gp_factor_bwd_late_a(c, ws, state, stream));       // what does not read the branch's results: before the join
            RUN(side_join(sd, 1, ms));                                  // (a no-op unless phase 1 left the branch open)
            // round 6: the single-matrix chain of the gradient of Ki (five ~9 us launches) on the branch that has just been joined,
            // beside the channel block on the caller's stream; SVGP_KBAR_BRANCH=0: one after the other
            if (kbar_branch_on()) {
                RUN(side_fork(sd, 1, ms));
                RUN(svgp_gp_factor_bwd_late_b_kbar(c, ws, state, (void*)sd->s[1]));
                RUN(svgp_gp_factor_bwd_late_b_channels(c, ws, state, stream));
                RUN(side_join(sd, 1, ms));
                RUN(svgp_gp_factor_bwd_late_b_final(c, ws, state, stream));
            } else
            RUN(svgp_gp_factor_bwd_late_b(c, ws, state, stream));
        } else {
            // channel sum Kbar: inside the next launch; m <= 64: + the decoder's weight gradients as riders (phase 1 ran the data half)
            if (stat_rides) RUN(svgp_gp_stats_factor_bwd_wgrad(c, images, ws, state, stream));
            else if (!large && dec_split_on()) RUN(svgp_gp_factor_bwd_nofinal_wgrad(c, images, ws, state, stream));
            else RUN(svgp_gp_factor_bwd_nofinal(c, ws, state, stream));
        }
        // m <= 64 (round 6): pass 2 of the reverse row stage (the sums over channels, consumed by the kernel-matrix VJP only) rides in the
        // encoder's reverse launch in front of the VJP workgroups.  Not with cfg.titsias (its reverse stage adds to Kbar / Knbar in
        // between), the split gradient exchange (phase 4) or SVGP_ENC_KM_MERGE=0.
        if (sum_rides) RUN(svgp_gp_posterior_bwd_rows(c, ws, state, stream));
        else RUN(svgp_gp_posterior_bwd_with_final(c, ws, state, stream));
        if (c->titsias) RUN(svgp_gp_titsias_bwd(c, ws, state, stream));
        // (m > 64, measured round 5: the kernel-matrix reverse pass on side branch 0 beside the encoder's does NOT overlap -- 68 KB +
        // 104 KB of LDS per workgroup do not fit one CU; the kernel-matrix launch stretched from 49 to 103 us and the step was unchanged)
        if (phase == 4) {
            RUN(svgp_kernel_matrix_bwd_partials(c, theta, aux, ws, stream));
            RUN(svgp_mnist_grad_reduce_part(c, aux, ws, 1, stream));
            break;
        }
        if (sum_rides) {
            RUN(svgp_mnist_encoder_bwd_km_sum(c, theta, images, aux, ws, state, stream));
        } else if (!large && !fork2 && enc_km_merge_on()) {
            RUN(svgp_mnist_encoder_bwd_km(c, theta, images, aux, ws, stream));
        } else {
            if (fork2) RUN(side_fork(sd, 0, ms));
            RUN(svgp_kernel_matrix_bwd_partials(c, theta, aux, ws, s2));
            RUN(svgp_mnist_encoder_bwd(c, theta, images, ws, stream));
            if (fork2) RUN(side_join(sd, 0, ms));
        }
        RUN(svgp_mnist_grad_reduce_all(c, aux, ws, stream));
        break;
    case 3: {
        svgp_mnist_param_layout pl;
        svgp_mnist_ws_layout wl;
        RUN(svgp_mnist_param_layout_get(c, &pl));
        RUN(svgp_mnist_ws_layout_get(c, &wl));
        if (adam_m != nullptr) {
            SVGP_REQUIRE(adam_v != nullptr, SVGP_ERR_INVALID, "adam_v is NULL");
            RUN(svgp_adam_tf1_finalize(c, pl.n_total, theta, ws + wl.grad, adam_m, adam_v, ws, state, 0.9, 0.999,
                                       1e-8, stream));
        } else {
            RUN(svgp_elbo_finalize_noadam(c, ws, state, stream));
        }
        break;
    }
    default:
        SVGP_REQUIRE(false, SVGP_ERR_INVALID, "phase %d out of range 0..3", phase);
    }
#undef RUN
    return SVGP_OK;
}
}  // namespace

extern "C" int svgp_mnist_step_phase(const svgp_mnist_cfg* c, int phase, double* theta, const double* images,
                                     const double* aux, const double* eps, double* ws, double* state,
                                     double* adam_m, double* adam_v, void* stream) {
    return step_phase_impl(c, phase, theta, images, aux, eps, ws, state, adam_m, adam_v, stream, 0);
}

// internal (comm.hip): one phase of a step whose phases are all issued back to back on one stream
int svgp_mnist_step_phase_deferred(const svgp_mnist_cfg* c, int phase, double* theta, const double* images,
                                   const double* aux, const double* eps, double* ws, double* state, double* adam_m,
                                   double* adam_v, void* stream) {
    return step_phase_impl(c, phase, theta, images, aux, eps, ws, state, adam_m, adam_v, stream, 1);
}

extern "C" int svgp_mnist_train_step(const svgp_mnist_cfg* c, double* theta, const double* images,
                                     const double* aux, const double* eps, double* ws, double* state,
                                     double* adam_m, double* adam_v, void* stream) {
    SVGP_REQUIRE(c && c->b == c->b_global, SVGP_ERR_INVALID,
                 "svgp_mnist_train_step is the single-GPU form (b == b_global); use svgp_mnist_step_phase "
                 "with all-reduces between phases for data parallelism");
    for (int ph = 0; ph < 4; ++ph) {
        int rc = step_phase_impl(c, ph, theta, images, aux, eps, ws, state, adam_m, adam_v, stream, 2);
        if (rc) return rc;
    }
    return SVGP_OK;
}

// ---------------------------------------------------------------------------------------------
// runtime helpers
// ---------------------------------------------------------------------------------------------
// Creates the library's side branches of `stream` now (outside any stream capture), with the concurrency probe; and the probe
// itself for callers that own their side streams (sprites.py).
extern "C" int svgp_side_streams_prepare(void* stream) {
    Side* sd = nullptr;
    return side_get((hipStream_t)stream, &sd);
}
extern "C" int svgp_streams_overlap(void* a, void* b, int* out) {
    SVGP_REQUIRE(out, SVGP_ERR_INVALID, "out is NULL");
    bool ok = false;
    int rc = streams_overlap((hipStream_t)a, (hipStream_t)b, &ok);
    *out = ok ? 1 : 0;
    return rc;
}
extern "C" int svgp_stream_create(void** out) {
    SVGP_REQUIRE(out, SVGP_ERR_INVALID, "out is NULL");
    hipStream_t s;
    SVGP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void*)s;
    return SVGP_OK;
}
extern "C" int svgp_stream_destroy(void* s) { SVGP_CHECK_HIP(hipStreamDestroy((hipStream_t)s)); return SVGP_OK; }
extern "C" int svgp_stream_sync(void* s) { SVGP_CHECK_HIP(hipStreamSynchronize((hipStream_t)s)); return SVGP_OK; }

extern "C" int svgp_graph_begin(void* s) {
    SVGP_CHECK_HIP(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal));
    return SVGP_OK;
}
extern "C" int svgp_graph_end(void* s, void** exec_out) {
    SVGP_REQUIRE(exec_out, SVGP_ERR_INVALID, "exec_out is NULL");
    hipGraph_t g = nullptr;
    SVGP_CHECK_HIP(hipStreamEndCapture((hipStream_t)s, &g));
    hipGraphExec_t e = nullptr;
    hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    SVGP_CHECK_HIP(err);
    *exec_out = (void*)e;
    return SVGP_OK;
}
extern "C" int svgp_graph_launch(void* e, void* s) {
    SVGP_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)e, (hipStream_t)s));
    return SVGP_OK;
}
extern "C" int svgp_graph_destroy(void* e) { SVGP_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)e)); return SVGP_OK; }

extern "C" int svgp_event_create(void** out) {
    SVGP_REQUIRE(out, SVGP_ERR_INVALID, "out is NULL");
    hipEvent_t ev;
    SVGP_CHECK_HIP(hipEventCreate(&ev));
    *out = (void*)ev;
    return SVGP_OK;
}
extern "C" int svgp_event_record(void* ev, void* s) {
    SVGP_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)s));
    return SVGP_OK;
}
extern "C" int svgp_event_elapsed_ms(void* a, void* b, float* ms) {
    SVGP_REQUIRE(ms, SVGP_ERR_INVALID, "ms is NULL");
    SVGP_CHECK_HIP(hipEventSynchronize((hipEvent_t)b));
    SVGP_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return SVGP_OK;
}
extern "C" int svgp_event_destroy(void* ev) { SVGP_CHECK_HIP(hipEventDestroy((hipEvent_t)ev)); return SVGP_OK; }
