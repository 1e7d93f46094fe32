// TF1-formula Adam and the scalar epilogue of the step (ELBO assembly, GECO state, counters).
#include "common.hpp"

namespace {

// tf.train.AdamOptimizer (TF 1.15): lr_t = lr sqrt(1-b2^t)/(1-b1^t); m,v moments;
// theta -= lr_t m / (sqrt(v) + eps)   (epsilon outside the bias correction).
// t = state[ADAM_T] + 1; the counter itself is advanced by k_elbo_finalize.
__global__ __launch_bounds__(SVGP_BLOCK) void k_adam_tf1(long long n, real* __restrict__ theta,
                                                         const real* __restrict__ grad, real* __restrict__ mo,
                                                         real* __restrict__ vo, const real* __restrict__ state,
                                                         real beta1, real beta2, real eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const real t = state[SVGP_ST_ADAM_T] + real(1);
    const real lr_t = state[SVGP_ST_LR] * sqrt(real(1) - pow(beta2, t)) / (real(1) - pow(beta1, t));
    const real g = grad[i];
    const real mi = beta1 * mo[i] + (real(1) - beta1) * g;
    const real vi = beta2 * vo[i] + (real(1) - beta2) * g * g;
    mo[i] = mi;
    vo[i] = vi;
    theta[i] -= lr_t * mi / (sqrt(vi) + eps);
}

struct FinArgs {
    int b_global, L, geco, did_adam, n_pix;
    real N_train, kappa_squared, alpha_next;
    const real* sums;   // [L3 data term, CE, recon sq, rows, Titsias row sum] summed over ranks
    const real* KL;     // (L)
    const real* tit;    // Titsias: [log det Sigma2 (L) | v2.t2 (L)], else NULL
    const real* ldK;
    real* state;
};

// SVGPVAE_model.py:880-925 (scalar assembly), MNIST_experiment.py:330-340 (GECO state carry)
__device__ __forceinline__ void elbo_finalize(const FinArgs& a) {
    real* st = a.state;
    const real bg = (real)a.b_global, Lr = (real)a.L;
    real sumKL = 0;
    for (int l = 0; l < a.L; ++l) sumKL += a.KL[l];
    real inside_recon = a.sums[0] - real(0.5) * Lr * bg * real(SVGP_LOG_2PI);
    real inside = inside_recon - (bg / a.N_train) * sumKL;
    if (a.tit) {   // SVGPVAE_model.py:246-259, 882-883: inside-ELBO = sum_l L_2, no KL part
        real mat = 0;
        for (int l = 0; l < a.L; ++l) mat += a.tit[l] - a.ldK[0] - a.tit[a.L + l];
        inside_recon = real(-0.5) * (Lr * bg * real(SVGP_LOG_2PI) + a.sums[4] + mat);
        inside = inside_recon;
        sumKL = 0;
    }
    const real ce = a.sums[1];
    const real KL_term = -ce + inside;
    const real sq = a.sums[2];
    real elbo, recon_loss;
    if (a.geco) {
        recon_loss = sq / (real)a.n_pix - bg * a.kappa_squared;
        const real alpha = st[SVGP_ST_ALPHA], lam = st[SVGP_ST_LAGRANGE];
        const real C_ma = alpha * st[SVGP_ST_C_MA] + (real(1) - alpha) * recon_loss / bg;
        elbo = -KL_term + lam * (recon_loss / bg + (C_ma - recon_loss / bg));
        st[SVGP_ST_C_MA] = C_ma;
        st[SVGP_ST_LAGRANGE] = lam * exp(C_ma);
    } else {
        recon_loss = sq / (real)a.n_pix;
        elbo = -recon_loss + (st[SVGP_ST_BETA] / Lr) * KL_term;
    }
    st[SVGP_ST_ELBO] = elbo;
    st[SVGP_ST_RECON_LOSS] = recon_loss;
    st[SVGP_ST_KL_TERM] = KL_term;
    st[SVGP_ST_INSIDE_ELBO] = inside;
    st[SVGP_ST_CE_TERM] = ce;
    st[SVGP_ST_INSIDE_RECON] = inside_recon;
    st[SVGP_ST_INSIDE_KL] = sumKL;
    st[SVGP_ST_ALPHA] = a.alpha_next;
    if (a.did_adam) st[SVGP_ST_ADAM_T] += real(1);
    st[SVGP_ST_RNG_CTR] += real(1);
}

__global__ void k_elbo_finalize(FinArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    elbo_finalize(a);
}

// Adam update and the scalar epilogue in ONE launch.  The epilogue advances state[ADAM_T], which every
// workgroup of the update reads, so it runs in whichever workgroup takes the last ticket -- by then all
// the others have read the old value.  The ticket lives in state[SVGP_ST_TICKET] (u64 bits, left at 0).
__global__ __launch_bounds__(SVGP_BLOCK) void k_adam_tf1_finalize(long long n, real* __restrict__ theta,
                                                                  const real* __restrict__ grad,
                                                                  real* __restrict__ mo, real* __restrict__ vo,
                                                                  real beta1, real beta2, real eps, FinArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const real* state = a.state;
    const real t = state[SVGP_ST_ADAM_T] + real(1);
    const real lr_t = state[SVGP_ST_LR] * sqrt(real(1) - pow(beta2, t)) / (real(1) - pow(beta1, t));
    if (i < n) {
        const real g = grad[i];
        const real mi = beta1 * mo[i] + (real(1) - beta1) * g;
        const real vi = beta2 * vo[i] + (real(1) - beta2) * g * g;
        mo[i] = mi;
        vo[i] = vi;
        theta[i] -= lr_t * mi / (sqrt(vi) + eps);
    }
    __shared__ int last;
    __syncthreads();   // every lane of this workgroup has read the state
    if (threadIdx.x == 0) {
        unsigned long long* ticket = reinterpret_cast<unsigned long long*>(a.state + SVGP_ST_TICKET);
        last = atomicAdd(ticket, 1ULL) == (unsigned long long)gridDim.x - 1ULL;
        if (last) {
            *ticket = 0ULL;
            elbo_finalize(a);
        }
    }
}

}  // namespace

extern "C" int svgp_adam_tf1_step(int64_t n, double* theta, const double* grad, double* adam_m, double* adam_v,
                                  const double* state, double beta1, double beta2, double epsilon, void* stream) {
    SVGP_REQUIRE(n >= 0 && theta && grad && adam_m && adam_v && state, SVGP_ERR_INVALID, "NULL device pointer");
    if (n == 0) return SVGP_OK;
    hipLaunchKernelGGL(k_adam_tf1, dim3((unsigned)((n + SVGP_BLOCK - 1) / SVGP_BLOCK)), dim3(SVGP_BLOCK), 0,
                       (hipStream_t)stream, (long long)n, theta, grad, adam_m, adam_v, state, beta1, beta2, epsilon);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

static int finalize_impl(const svgp_mnist_cfg* c, double* ws, double* state, int did_adam, void* stream) {
    svgp_mnist_ws_layout wl;
    int rc = svgp_mnist_ws_layout_get(c, &wl);
    if (rc) return rc;
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    FinArgs a;
    a.b_global = c->b_global; a.L = c->L; a.geco = c->geco; a.did_adam = did_adam;
    a.n_pix = c->n_pix > 0 ? c->n_pix : 784;
    a.N_train = c->N_train; a.kappa_squared = c->kappa_squared; a.alpha_next = c->alpha;
    a.sums = ws + wl.sums; a.KL = ws + wl.KL; a.state = state;
    a.tit = c->titsias ? ws + wl.tit_scal : nullptr; a.ldK = ws + wl.ldK;
    hipLaunchKernelGGL(k_elbo_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_adam_tf1_finalize(const svgp_mnist_cfg* c, int64_t n, double* theta, const double* grad,
                                      double* adam_m, double* adam_v, double* ws, double* state, double beta1,
                                      double beta2, double epsilon, void* stream) {
    svgp_mnist_ws_layout wl;
    int rc = svgp_mnist_ws_layout_get(c, &wl);
    if (rc) return rc;
    SVGP_REQUIRE(n > 0 && theta && grad && adam_m && adam_v && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    FinArgs a;
    a.b_global = c->b_global; a.L = c->L; a.geco = c->geco; a.did_adam = 1;
    a.n_pix = c->n_pix > 0 ? c->n_pix : 784;
    a.N_train = c->N_train; a.kappa_squared = c->kappa_squared; a.alpha_next = c->alpha;
    a.sums = ws + wl.sums; a.KL = ws + wl.KL; a.state = state;
    a.tit = c->titsias ? ws + wl.tit_scal : nullptr; a.ldK = ws + wl.ldK;
    hipLaunchKernelGGL(k_adam_tf1_finalize, dim3((unsigned)((n + SVGP_BLOCK - 1) / SVGP_BLOCK)), dim3(SVGP_BLOCK), 0,
                       (hipStream_t)stream, (long long)n, theta, grad, adam_m, adam_v, beta1, beta2, epsilon, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_elbo_finalize(const svgp_mnist_cfg* c, double* ws, double* state, void* stream) {
    return finalize_impl(c, ws, state, 1, stream);
}

// used by svgp_mnist_step_phase(3) when no optimiser step is requested (adam_m == NULL)
extern "C" int svgp_elbo_finalize_noadam(const svgp_mnist_cfg* c, double* ws, double* state, void* stream) {
    return finalize_impl(c, ws, state, 0, stream);
}
