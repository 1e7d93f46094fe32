// mnistVAE encoder / decoder, forward and reverse, for gfx950.
//
// Reference semantics: VAE_utils.py:99-162 (Keras NHWC, 3x3 kernels (kh,kw,cin,cout), ELU after
// every conv incl. the last decoder conv, 'valid' encoder convs with stride 2, decoder
// UpSampling2D(2) nearest + conv 'same'/'valid'/'same').
//
// Design: one workgroup (256 threads = 4 waves) owns one image at a time; all activations of an
// image (13.5 k values) and all weights stay in LDS, so HBM traffic is exactly: image in,
// activations out once (kept for the reverse pass), weight-gradient partials out once per
// workgroup.  UpSampling2D is never materialised (index >> 1 on the LDS read).  Weight gradients
// are accumulated in LDS over the images a workgroup walks and written as per-workgroup partials;
// svgp_mnist_grad_reduce sums them in a fixed order (bitwise reproducible, no float atomics).
#include <cstdlib>

#include "common.hpp"
#include "vae_dev.hpp"

namespace {

using namespace svgp_vae;

// ------------------------------------------------------------------------------------------
// encoder forward: images -> a1,a2,a3 (saved), qnet_mu, qnet_var_raw = exp(.), qnet_var = clip
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VAE_NT) void k_encoder_fwd(int b, int L, int clip, const real* __restrict__ th_enc,
                                                            const real* __restrict__ images, real* __restrict__ a1g,
                                                            real* __restrict__ a2g, real* __restrict__ a3g,
                                                            real* __restrict__ mu, real* __restrict__ var_raw,
                                                            real* __restrict__ var, int n_img_blocks, SvgpKernArgs ka,
                                                            real* __restrict__ Kmm, real* __restrict__ Knm,
                                                            real* __restrict__ knn, int n_km_blocks,
                                                            const real* __restrict__ th_dec, real* __restrict__ weff) {
    extern __shared__ __align__(16) real smem[];
    if ((int)blockIdx.x >= n_img_blocks + n_km_blocks) {
        // training phases (round 6): ONE more rider builds the effective (parity-class) weights of the decoder's three up-convolutions
        // for the step -- theta does not change before Adam -- so that the decoder's forward and data-reverse launches load them
        // instead of rebuilding them in each of their 256 workgroups (~1 us on the chain, twice)
        const DecOff od = dec_off(L);
        UpC1::build_weff(th_dec + od.c1w, weff);
        UpC2::build_weff(th_dec + od.c2w, weff + UpC1::NWE);
        UpC3::build_weff(th_dec + od.c3w, weff + UpC1::NWE + UpC2::NWE);
        return;
    }
    if ((int)blockIdx.x >= n_img_blocks) {
        // training phases: the kernel-matrix build (independent of the encoder) rides in extra workgroups of this launch
        svgp_km_fwd_element(ka, (long long)(blockIdx.x - n_img_blocks) * blockDim.x + threadIdx.x, Kmm, Knm, knn);
        return;
    }
    const EncOff eo = enc_off(L);
    real* w = smem;                 // eo.n
    real* img = w + eo.n;           // 784
    real* a1 = img + 784;           // 1352
    real* a2 = a1 + 1352;           // 288
    real* a3 = a2 + 288;            // 32
    lds_copy_in(w, th_enc, eo.n);
    for (int n = blockIdx.x; n < b; n += n_img_blocks) {
        __syncthreads();
        lds_copy_in(img, images + (size_t)n * 784, 784);
        __syncthreads();
        EncC1::fwd(img, w + eo.c1w, w + eo.c1b, a1);
        __syncthreads();
        EncC2::fwd(a1, w + eo.c2w, w + eo.c2b, a2);
        __syncthreads();
        EncC3::fwd(a2, w + eo.c3w, w + eo.c3b, a3);
        __syncthreads();
        lds_copy_out(a1g + (size_t)n * 1352, a1, 1352);
        lds_copy_out(a2g + (size_t)n * 288, a2, 288);
        lds_copy_out(a3g + (size_t)n * 32, a3, 32);
        const int twoL = 2 * L;
        for (int j = threadIdx.x; j < twoL; j += blockDim.x) {
            real acc = w[eo.db + j];
            for (int i = 0; i < 32; ++i) acc += a3[i] * w[eo.dw + i * twoL + j];
            if (j < L) {
                mu[(size_t)n * L + j] = acc;
            } else {
                const real vr = exp(acc);
                var_raw[(size_t)n * L + j - L] = vr;
                var[(size_t)n * L + j - L] = clip ? fmin(fmax(vr, 1e-3), 10.0) : vr;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// encoder reverse: (ybar, s2bar) -> encoder weight-gradient partials (encoder_bwd_images, vae_dev.hpp; the training step
// runs the same device function in a launch that also carries the kernel-matrix VJP: svgp_mnist_encoder_bwd_km)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VAE_NT) void k_encoder_bwd(EncBwdArgs a) {
    extern __shared__ __align__(16) real smem[];
    encoder_bwd_images<VAE_NT>(a, blockIdx.x, gridDim.x, smem);
}

// ------------------------------------------------------------------------------------------
// decoder forward: z -> h0, a1, a2 (saved), recon, per-workgroup sum of squared errors
// ------------------------------------------------------------------------------------------
// PRE: the effective weights come from ws.dec_weff (built once per step by a rider of the encoder launch)
template <bool PRE>
__global__ __launch_bounds__(VAE_NT) void k_decoder_fwd(int b, int L, const real* __restrict__ th_dec,
                                                            const real* __restrict__ images,
                                                            const real* __restrict__ zg, real* __restrict__ h0g,
                                                            real* __restrict__ a1g, real* __restrict__ a2g,
                                                            real* __restrict__ recon, real* __restrict__ part_sums,
                                                            const real* __restrict__ weff) {
    extern __shared__ __align__(16) real smem[];
    const DecOff od = dec_off(L);
    real* w = smem;                  // od.n
    real* z = w + od.n;              // 64
    real* h0 = z + 64;               // 128
    real* a1 = h0 + 128;             // 512
    real* a2 = a1 + 512;             // 1568
    real* out = a2 + 1568;           // 784
    real* red = out + 784;           // 16
    real* We1 = red + 16;            // effective weights of the three up-convolutions
    real* We2 = We1 + UpC1::NWE;
    real* We3 = We2 + UpC2::NWE;
    lds_copy_in(w, th_dec, od.n);
    if (PRE) {
        lds_copy_in(We1, weff, DEC_NWE);
    } else {
        __syncthreads();
        UpC1::build_weff(w + od.c1w, We1);       // from the LDS copy (one coalesced read of the raw weights)
        UpC2::build_weff(w + od.c2w, We2);
        UpC3::build_weff(w + od.c3w, We3);
    }
    real sq = 0;
    for (int n = blockIdx.x; n < b; n += gridDim.x) {
        __syncthreads();
        lds_copy_in(z, zg + (size_t)n * L, L);
        __syncthreads();
        if (threadIdx.x < 128) {
            real acc = w[od.db + threadIdx.x];
            for (int i = 0; i < L; ++i) acc += z[i] * w[od.dw + i * 128 + threadIdx.x];
            h0[threadIdx.x] = acc;
        }
        __syncthreads();
        UpC1::fwd_valu(h0, We1, w + od.c1b, a1);
        __syncthreads();
        UpC2::fwd_valu(a1, We2, w + od.c2b, a2);
        __syncthreads();
        UpC3::fwd_valu(a2, We3, w + od.c3b, out);
        __syncthreads();
        lds_copy_out(h0g + (size_t)n * 128, h0, 128);
        lds_copy_out(a1g + (size_t)n * 512, a1, 512);
        lds_copy_out(a2g + (size_t)n * 1568, a2, 1568);
        for (int i = threadIdx.x; i < 784; i += blockDim.x) {
            const real o = out[i];
            recon[(size_t)n * 784 + i] = o;
            const real df = images[(size_t)n * 784 + i] - o;
            sq += df * df;
        }
    }
    const real tot = block_sum(sq, red);
    if (threadIdx.x == 0) part_sums[blockIdx.x * 4 + 2] = tot;
}

// ------------------------------------------------------------------------------------------
// decoder reverse: d loss / d recon -> zbar, decoder weight-gradient partials
// gscale = d loss / d (sum of squared errors): beta-ELBO 1/784; GECO lagrange_mult/(b_global*784)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VAE_NT) void k_decoder_bwd(int b, int L, int geco, real inv_bglobal,
                                                            const real* __restrict__ state,
                                                            const real* __restrict__ th_dec,
                                                            const real* __restrict__ images,
                                                            const real* __restrict__ zg, const real* __restrict__ h0g,
                                                            const real* __restrict__ a1g, const real* __restrict__ a2g,
                                                            const real* __restrict__ recon, real* __restrict__ zbar,
                                                            real* __restrict__ part) {
    extern __shared__ __align__(16) real smem[];
    const DecOff od = dec_off(L);
    real* w = smem;                  // dense weights only: L*128
    real* g = w + L * 128;           // od.n   gradient in the raw-parameter layout
    real* We1 = g + od.n;            // effective weights / their gradients
    real* We2 = We1 + UpC1::NWE;
    real* We3 = We2 + UpC2::NWE;
    real* gWe1 = We3 + UpC3::NWE;
    real* gWe2 = gWe1 + UpC1::NWE;
    real* gWe3 = gWe2 + UpC2::NWE;
    real* z = gWe3 + UpC3::NWE;      // 64
    real* h0 = z + 64;               // 128
    real* a1 = h0 + 128;             // 512
    real* a2 = a1 + 512;             // 1568
    real* d3 = a2 + 1568;            // 784
    real* d2 = d3 + 784;             // 1568
    real* d1 = d2 + 1568;            // 512
    real* dh0 = d1 + 512;            // 128
    real* scratch = dh0 + 128;       // VAE_SCRATCH
    lds_copy_in(w, th_dec + od.dw, L * 128);
    lds_zero(g, od.n);
    lds_zero(gWe1, DEC_NWE);
    lds_copy_in(scratch, th_dec + od.c1w, od.n - od.c1w);      // raw conv weights (+biases), staged once
    __syncthreads();
    UpC1::build_weff(scratch, We1);
    UpC2::build_weff(scratch + (od.c2w - od.c1w), We2);
    UpC3::build_weff(scratch + (od.c3w - od.c1w), We3);
    const real gscale = (geco ? state[SVGP_ST_LAGRANGE] * inv_bglobal : real(1)) / real(784);
    for (int n = blockIdx.x; n < b; n += gridDim.x) {
        __syncthreads();
        lds_copy_in(z, zg + (size_t)n * L, L);
        lds_copy_in(h0, h0g + (size_t)n * 128, 128);
        lds_copy_in(a1, a1g + (size_t)n * 512, 512);
        lds_copy_in(a2, a2g + (size_t)n * 1568, 1568);
        for (int i = threadIdx.x; i < 784; i += blockDim.x) {
            const real o = recon[(size_t)n * 784 + i];
            d3[i] = real(2) * gscale * (o - images[(size_t)n * 784 + i]) * elu_grad_from_out(o);
        }
        __syncthreads();
        UpC3::bwd_weight_mfma(a2, d3, gWe3, g + od.c3b, scratch);     // COUT = 1: chunked VALU form inside
        UpC3::bwd_data_valu(d3, We3, d2);
        __syncthreads();
        for (int i = threadIdx.x; i < 1568; i += blockDim.x) d2[i] *= elu_grad_from_out(a2[i]);
        __syncthreads();
        UpC2::bwd_weight_valu(a1, d2, gWe2, g + od.c2b, scratch);
        UpC2::bwd_data_mfma(d2, We2, d1);
        __syncthreads();
        for (int i = threadIdx.x; i < 512; i += blockDim.x) d1[i] *= elu_grad_from_out(a1[i]);
        __syncthreads();
        UpC1::bwd_weight_mfma(h0, d1, gWe1, g + od.c1b, scratch);
        UpC1::bwd_data_mfma(d1, We1, dh0);
        __syncthreads();
        // dense (no activation): weight / bias gradients and zbar
        for (int o = threadIdx.x; o < L * 128; o += blockDim.x) g[od.dw + o] += z[o / 128] * dh0[o % 128];
        if (threadIdx.x < 128) g[od.db + threadIdx.x] += dh0[threadIdx.x];
        // zbar[i] = sum_j dh0[j] w[i][j]: 8 lanes per latent channel, xor-shuffle combine
        {
            const int i = threadIdx.x >> 3, part8 = threadIdx.x & 7;
            real acc = 0;
            if (i < L)
                for (int j = part8; j < 128; j += 8) acc += dh0[j] * w[i * 128 + j];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (i < L && part8 == 0) zbar[(size_t)n * L + i] = acc;
        }
    }
    __syncthreads();
    UpC1::fold_grad(gWe1, g + od.c1w);
    UpC2::fold_grad(gWe2, g + od.c2w);
    UpC3::fold_grad(gWe3, g + od.c3w);
    __syncthreads();
    lds_copy_out(part + (size_t)blockIdx.x * od.n, g, od.n);
}

// ------------------------------------------------------------------------------------------
// decoder reverse, DATA half: d loss / d recon -> d2, d1, dh0 (stored for the weight half) -> zbar.  The chain the GP
// reverse stages wait for; the weight gradients (decoder_wgrad_rider, vae_dev.hpp) ride in a later launch.
// ------------------------------------------------------------------------------------------
template <bool PRE>
__global__ __launch_bounds__(VAE_NT) void k_decoder_bwd_data(DecBwdDataArgs a) {
    extern __shared__ __align__(16) real smem[];
    decoder_bwd_data_images<PRE>(a, blockIdx.x, gridDim.x, smem);
}

// decoder reverse, WEIGHT half as a launch of its own (stand-alone entry point, probes; the training step runs the same device
// function as riders of the reverse factor launch)
template <int NT>
__global__ __launch_bounds__(NT) void k_decoder_bwd_weights(DecWgradArgs a) {
    extern __shared__ __align__(16) real smem[];
    decoder_wgrad_rider<NT>(a, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// fixed-order reduction of the per-workgroup partials into [grad | sums] in ONE launch.
// blocks [0, nb_enc): encoder weights, [nb_enc, nb_enc+nb_dec): decoder weights, next block: scalars, then (training
// phases only, n_scatter > 0) the object-table scatter / GP hyper-parameter sums of the kernel-matrix VJP.
// thread (i_local = tid % 16, chunk = tid / 16): 16 chunks over the partial rows, LDS combine.
// ------------------------------------------------------------------------------------------
struct KmScatter {
    int n_blocks, M, n_obj, n_gp_part, train_gp, train_ov;
    const real* aux; const real* d_on; const real* part_gp;
    real* d_ov; real* d_ls; real* d_amp;
};

__global__ __launch_bounds__(SVGP_BLOCK) void k_grad_reduce(int n_part, int n_enc, int n_dec, int nb_enc, int nb_dec,
                                                            int n_post, int b, const real* __restrict__ part_enc,
                                                            const real* __restrict__ part_dec,
                                                            const real* __restrict__ part_sums,
                                                            const real* __restrict__ tit_rowsum,
                                                            real* __restrict__ grad, real* __restrict__ sums,
                                                            KmScatter ks, int blk0) {
    __shared__ real s[16][17];
    __shared__ real red[16];
    const int bid = (int)blockIdx.x + blk0;          // blk0: the launch covers the logical blocks [blk0, blk0 + gridDim.x)
    if (bid > nb_enc + nb_dec) {
        svgp_km_scatter_block(bid - (nb_enc + nb_dec + 1), ks.n_blocks, b, ks.M, ks.n_obj, ks.aux, ks.n_gp_part,
                              ks.train_gp, ks.train_ov, ks.d_on, ks.part_gp, ks.d_ov, ks.d_ls, ks.d_amp);
        return;
    }
    if (bid == nb_enc + nb_dec) {
        real l3 = 0, ce = 0, sq = 0;
        for (int i = threadIdx.x; i < n_part; i += blockDim.x) sq += part_sums[i * 4 + 2];
        const real* pp = part_sums + (size_t)n_part * 4;
        for (int i = threadIdx.x; i < n_post; i += blockDim.x) { l3 += pp[i * 2]; ce += pp[i * 2 + 1]; }
        l3 = block_sum(l3, red);
        ce = block_sum(ce, red);
        sq = block_sum(sq, red);
        if (threadIdx.x == 0) {
            sums[0] = l3; sums[1] = ce; sums[2] = sq; sums[3] = (real)b;
            sums[4] = tit_rowsum ? *tit_rowsum : real(0);   // Titsias: this rank's sum_n,l [log d + y^2/d + (k_nn - q)/var]
            sums[5] = 0; sums[6] = 0; sums[7] = 0;
        }
        return;
    }
    const bool enc = bid < nb_enc;
    const int n = enc ? n_enc : n_dec, blk = enc ? bid : bid - nb_enc;
    const real* part = enc ? part_enc : part_dec;
    real* out = enc ? grad : grad + n_enc;
    const int il = threadIdx.x & 15, ch = threadIdx.x >> 4;
    const int i = blk * 16 + il;
    real acc = 0;
    if (i < n) {
#pragma unroll 4
        for (int wq = ch; wq < n_part; wq += 16) acc += part[(size_t)wq * n + i];
    }
    s[ch][il] = acc;
    __syncthreads();
    if (ch == 0 && i < n) {
        real t = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) t += s[c][il];
        out[i] = t;
    }
}

template <typename F>
int set_dyn_lds(F kernel, size_t bytes) {
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SVGP_OK;
}

}  // namespace

#define GET_LAYOUTS()                                          \
    svgp_mnist_param_layout pl;                                \
    svgp_mnist_ws_layout wl;                                   \
    {                                                          \
        int rc_ = svgp_mnist_param_layout_get(c, &pl);         \
        if (rc_) return rc_;                                   \
        rc_ = svgp_mnist_ws_layout_get(c, &wl);                \
        if (rc_) return rc_;                                   \
    }

static int encoder_fwd_impl(const svgp_mnist_cfg* c, const double* theta, const double* images, const double* aux,
                            double* ws, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws, SVGP_ERR_INVALID, "NULL device pointer");
    const size_t lds = (size_t)(pl.n_enc + 784 + 1352 + 288 + 32) * sizeof(real);
    int rc = set_dyn_lds(k_encoder_fwd, lds);
    if (rc) return rc;
    const int n_img = svgp_n_part(c);
    SvgpKernArgs ka;
    memset(&ka, 0, sizeof(ka));
    int n_km = 0;
    if (aux) {
        ka = svgp_make_kern_args(c, pl, theta, aux);
        const long long nel = (long long)c->b * c->m + (long long)c->m * c->m + c->b;
        n_km = (int)((nel + VAE_NT - 1) / VAE_NT);
    }
    // (the weff rider only in the training-phase form: svgp_mnist_decoder_fwd_pre / _bwd_data_pre are issued by the same step)
    hipLaunchKernelGGL(k_encoder_fwd, dim3(n_img + n_km + (aux ? 1 : 0)), dim3(VAE_NT), lds, (hipStream_t)stream, c->b, c->L,
                       c->clip_qs, theta, images, ws + wl.enc_a1, ws + wl.enc_a2, ws + wl.enc_a3, ws + wl.qnet_mu,
                       ws + wl.qnet_var_raw, ws + wl.qnet_var, n_img, ka, ws + wl.K, ws + wl.Kn, ws + wl.knn, n_km,
                       theta + pl.n_enc, ws + wl.dec_weff);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_mnist_encoder_fwd(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                      void* stream) {
    return encoder_fwd_impl(c, theta, images, nullptr, ws, stream);
}

// phase form: svgp_mnist_encoder_fwd + svgp_kernel_matrix_fwd in one launch (the two are independent)
extern "C" int svgp_mnist_encoder_kernel_matrix_fwd(const svgp_mnist_cfg* c, const double* theta, const double* images,
                                                    const double* aux, double* ws, void* stream) {
    SVGP_REQUIRE(aux, SVGP_ERR_INVALID, "NULL device pointer");
    return encoder_fwd_impl(c, theta, images, aux, ws, stream);
}

extern "C" int svgp_mnist_encoder_bwd(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                      void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws, SVGP_ERR_INVALID, "NULL device pointer");
    const size_t lds = (size_t)enc_bwd_lds((int)pl.n_enc) * sizeof(real);
    int rc = set_dyn_lds(k_encoder_bwd, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_encoder_bwd, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream,
                       svgp_make_enc_bwd_args(c, wl, theta, images, ws));
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

static int decoder_fwd_impl(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws, bool pre, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws, SVGP_ERR_INVALID, "NULL device pointer");
    const int64_t n_dec = pl.n_vae - pl.n_enc;
    const size_t lds = (size_t)(n_dec + 64 + 128 + 512 + 1568 + 784 + 16 + DEC_NWE) * sizeof(real);
    int rc = pre ? set_dyn_lds(k_decoder_fwd<true>, lds) : set_dyn_lds(k_decoder_fwd<false>, lds);
    if (rc) return rc;
#define DEC_FWD_ARGS c->b, c->L, theta + pl.n_enc, images, ws + wl.z, ws + wl.dec_h0, ws + wl.dec_a1, ws + wl.dec_a2, ws + wl.recon, \
                     ws + wl.part_sums, ws + wl.dec_weff
    if (pre) hipLaunchKernelGGL(k_decoder_fwd<true>, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, DEC_FWD_ARGS);
    else hipLaunchKernelGGL(k_decoder_fwd<false>, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, DEC_FWD_ARGS);
#undef DEC_FWD_ARGS
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_mnist_decoder_fwd(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                      void* stream) {
    return decoder_fwd_impl(c, theta, images, ws, false, stream);
}
// training-step forms: the effective weights of the up-convolutions are read from ws.dec_weff, which
// svgp_mnist_encoder_kernel_matrix_fwd of the SAME step (same theta) has written
extern "C" int svgp_mnist_decoder_fwd_pre(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                          void* stream) {
    return decoder_fwd_impl(c, theta, images, ws, true, stream);
}

extern "C" int svgp_mnist_decoder_bwd(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                      const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    const int64_t n_dec = pl.n_vae - pl.n_enc;
    const size_t lds = (size_t)(c->L * 128 + n_dec + 2 * DEC_NWE + 64 + 128 + 512 + 1568 + 784 + 1568 + 512 + 128 + VAE_SCRATCH) * sizeof(real);
    int rc = set_dyn_lds(k_decoder_bwd, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_decoder_bwd, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, c->b, c->L,
                       c->geco, 1.0 / (double)c->b_global, state, theta + pl.n_enc, images, ws + wl.z, ws + wl.dec_h0,
                       ws + wl.dec_a1, ws + wl.dec_a2, ws + wl.recon, ws + wl.zbar, ws + wl.part_dec);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// The two halves of svgp_mnist_decoder_bwd (see k_decoder_bwd_data): _data writes zbar and ws.dec_d2 / dec_d1 / dec_dh0, _weights
// the decoder weight-gradient partials from them.  _data + _weights == svgp_mnist_decoder_bwd up to summation order.
// gp_kernels.hip (svgp_mnist_decoder_bwd_data_pre_aji) and the launches below
svgp_vae::DecBwdDataArgs svgp_make_dec_bwd_data_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* theta,
                                                     const double* images, double* ws, const double* state) {
    svgp_mnist_param_layout pl;
    svgp_mnist_param_layout_get(c, &pl);
    DecBwdDataArgs a;
    a.b = c->b; a.L = c->L; a.geco = c->geco; a.inv_bglobal = 1.0 / (double)c->b_global;
    a.state = state; a.th_dec = theta + pl.n_enc; a.images = images; a.a1g = ws + wl.dec_a1; a.a2g = ws + wl.dec_a2;
    a.recon = ws + wl.recon; a.d2g = ws + wl.dec_d2; a.d1g = ws + wl.dec_d1; a.dh0g = ws + wl.dec_dh0; a.zbar = ws + wl.zbar;
    a.weff = ws + wl.dec_weff;
    return a;
}
static int decoder_bwd_data_impl(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                 const double* state, bool pre, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    const int64_t n_dec = pl.n_vae - pl.n_enc;
    const size_t lds = (size_t)dec_bwd_data_lds(c->L, (int)n_dec, pre) * sizeof(real);
    int rc = pre ? set_dyn_lds(k_decoder_bwd_data<true>, lds) : set_dyn_lds(k_decoder_bwd_data<false>, lds);
    if (rc) return rc;
    const DecBwdDataArgs a = svgp_make_dec_bwd_data_args(c, wl, theta, images, ws, state);
    if (pre) hipLaunchKernelGGL(k_decoder_bwd_data<true>, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_decoder_bwd_data<false>, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_mnist_decoder_bwd_data(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                           const double* state, void* stream) {
    return decoder_bwd_data_impl(c, theta, images, ws, state, false, stream);
}
extern "C" int svgp_mnist_decoder_bwd_data_pre(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                               const double* state, void* stream) {
    return decoder_bwd_data_impl(c, theta, images, ws, state, true, stream);
}

// gp_kernels.hip (svgp_mnist_encoder_bwd_km) and svgp_mnist_encoder_bwd above
svgp_vae::EncBwdArgs svgp_make_enc_bwd_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* theta,
                                            const double* images, double* ws) {
    EncBwdArgs a;
    a.b = c->b; a.L = c->L; a.clip = c->clip_qs; a.th_enc = theta; a.images = images;
    a.a1g = ws + wl.enc_a1; a.a2g = ws + wl.enc_a2; a.a3g = ws + wl.enc_a3; a.var_raw = ws + wl.qnet_var_raw;
    a.ybar = ws + wl.ybar; a.s2bar = ws + wl.s2bar; a.part = ws + wl.part_enc;
    return a;
}

// gp_kernels.hip (the riders of the reverse factor launch) and the stand-alone launch below
svgp_vae::DecWgradArgs svgp_make_dec_wgrad_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* images,
                                                double* ws, const double* state, int n_types) {
    DecWgradArgs a;
    a.b = c->b; a.L = c->L; a.geco = c->geco; a.n_slots = svgp_n_part(c); a.n_types = n_types; a.inv_bglobal = 1.0 / (double)c->b_global;
    a.state = state; a.images = images; a.z = ws + wl.z; a.h0 = ws + wl.dec_h0; a.a1 = ws + wl.dec_a1; a.a2 = ws + wl.dec_a2;
    a.recon = ws + wl.recon; a.d2 = ws + wl.dec_d2; a.d1 = ws + wl.dec_d1; a.dh0 = ws + wl.dec_dh0; a.part = ws + wl.part_dec;
    return a;
}

extern "C" int svgp_mnist_decoder_bwd_weights(const svgp_mnist_cfg* c, const double* images, double* ws, const double* state,
                                              int threads, int n_types, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(images && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(threads == 256 || threads == 512, SVGP_ERR_INVALID, "threads=%d (256 or 512)", threads);
    SVGP_REQUIRE(n_types >= 1 && n_types <= 3, SVGP_ERR_INVALID, "n_types=%d (1, 2 or 3 workgroups per image)", n_types);
    const DecWgradArgs a = svgp_make_dec_wgrad_args(c, wl, images, ws, state, n_types);
    const size_t lds = (size_t)dec_wgrad_lds(threads, c->L, n_types) * sizeof(real);
    int rc = threads == 256 ? set_dyn_lds(k_decoder_bwd_weights<256>, lds) : set_dyn_lds(k_decoder_bwd_weights<512>, lds);
    if (rc) return rc;
    const dim3 grid(a.n_slots * a.n_types);
    if (threads == 256) hipLaunchKernelGGL(k_decoder_bwd_weights<256>, grid, dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_decoder_bwd_weights<512>, grid, dim3(512), lds, (hipStream_t)stream, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

static int grad_reduce_impl(const svgp_mnist_cfg* c, const double* aux, double* ws, bool with_scatter, void* stream, int part = 0) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    const int n_part = svgp_n_part(c);
    const int n_enc = (int)pl.n_enc, n_dec = (int)(pl.n_vae - pl.n_enc);
    const int nb_enc = (n_enc + 15) / 16, nb_dec = (n_dec + 15) / 16;
    KmScatter ks;
    memset(&ks, 0, sizeof(ks));
    size_t lds = 0;
    if (with_scatter) {
        SVGP_REQUIRE(aux, SVGP_ERR_INVALID, "NULL device pointer");
        const int RBk = svgp_rows_per_block(c), nrb = (c->b + RBk - 1) / RBk;
        ks.n_blocks = (c->n_obj * c->M + SVGP_BLOCK - 1) / SVGP_BLOCK + 1;
        ks.M = c->M; ks.n_obj = c->n_obj; ks.n_gp_part = c->m + nrb; ks.train_gp = c->train_gp; ks.train_ov = c->train_ov;
        ks.aux = aux; ks.d_on = ws + wl.d_on; ks.part_gp = ws + wl.part_gp;
        ks.d_ov = ws + wl.grad + pl.ov; ks.d_ls = ws + wl.grad + pl.l_GP; ks.d_amp = ws + wl.grad + pl.amplitude;
        lds = svgp_km_scatter_lds(c->b, c->M);
        int rc = set_dyn_lds(k_grad_reduce, lds);
        if (rc) return rc;
    }
    // part 0: all logical blocks; 2: the encoder's [0, nb_enc); 1: the rest
    const int total = nb_enc + nb_dec + 1 + ks.n_blocks, blk0 = part == 1 ? nb_enc : 0, nblk = part == 2 ? nb_enc : total - blk0;
    hipLaunchKernelGGL(k_grad_reduce, dim3(nblk), dim3(SVGP_BLOCK), lds, (hipStream_t)stream,
                       n_part, n_enc, n_dec, nb_enc, nb_dec, svgp_n_post_actual(c), c->b, ws + wl.part_enc, ws + wl.part_dec,
                       ws + wl.part_sums, c->titsias ? ws + wl.tit_scal + 2 * c->L : (const real*)nullptr, ws + wl.grad,
                       ws + wl.sums, ks, blk0);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_mnist_grad_reduce(const svgp_mnist_cfg* c, double* ws, void* stream) {
    return grad_reduce_impl(c, nullptr, ws, false, stream);
}

// + the object-table scatter and hyper-parameter sums left open by svgp_kernel_matrix_bwd_partials
extern "C" int svgp_mnist_grad_reduce_all(const svgp_mnist_cfg* c, const double* aux, double* ws, void* stream) {
    return grad_reduce_impl(c, aux, ws, true, stream);
}
extern "C" int svgp_mnist_grad_reduce_part(const svgp_mnist_cfg* c, const double* aux, double* ws, int part, void* stream) {
    SVGP_REQUIRE(part == 1 || part == 2, SVGP_ERR_INVALID, "part %d (1 or 2)", part);
    return grad_reduce_impl(c, aux, ws, true, stream, part);
}
