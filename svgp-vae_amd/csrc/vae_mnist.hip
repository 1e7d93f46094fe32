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

namespace {

#define VAE_NT 512          // threads per workgroup of the VAE kernels (8 waves: 2 per SIMD)
#define VAE_SCRATCH 4096    // reals of LDS scratch for the weight-gradient chunk reduction

typedef double d4v_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// In-LDS gather-GEMM on the f64 MFMA: the conv im2col-GEMMs of the per-image kernels.
//   gg_fwd  : out[(y*osy+ooy)*Wo + x*osx+oox][co] = epi(bias[co] + sum_t sum_ci in[(y*sy+oy_t)*Wi + x*sx+ox_t][ci] W_t[ci][co])
//             A[i = pixel][k = ci] gathered per lane (zero outside the tile), B[k = ci][j = co] from LDS weights,
//             one v_mfma_f64_16x16x4 per (tap, 4 channels); a wave owns 16 consecutive pixels of the iteration space.
//             TW: weights stored [co][ci] are read transposed (data gradient).
//   gg_wgrad: gW_t[ci][co] += sum_pixels in_t[pixel][ci] * dout[pixel][co]; A[i = ci][k = pixel], B[k = pixel][j = co];
//             a wave owns whole taps, so the LDS accumulators need no atomics.
// LDS bank conflicts of the 16-pixel gathers (stride Ci doubles) cost a few cycles per fetch and hide under the
// 64-cycle issue of the f64 MFMA.
// ---------------------------------------------------------------------------------------------
template <int NTAP, bool TW, bool ELU_BIAS, int CI, int CO>
__device__ __forceinline__ void gg_fwd(const real* in, int Hi, int Wi, int Hs, int Ws, int sy, int sx,
                                       const int (&oy)[NTAP], const int (&ox)[NTAP], const int (&woff)[NTAP],
                                       const real* W, int ldw, const real* bias, real* out, int Wo, int osy, int osx,
                                       int ooy, int oox) {
    constexpr int KQ = (CI + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = VAE_NT >> 6, r = lane & 15, q = lane >> 4;
    const int NP = Hs * Ws, ngrp = (NP + 15) >> 4;
    // B operands (weights) do not depend on the pixel group: fetched once, unconditionally (clamped index + select)
    real breg[NTAP * KQ];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int c = kq * 4 + q, cc = c < CI ? c : CI - 1, rc = r < CO ? r : CO - 1;
            const real v = W[woff[t] + (TW ? rc * ldw + cc : cc * ldw + rc)];
            breg[t * KQ + kq] = (c < CI && r < CO) ? v : real(0);
        }
    for (int grp = wave; grp < ngrp; grp += nwave) {
        const int pa = grp * 16 + r;
        const bool pv = pa < NP;
        const int ya = pv ? pa / Ws : 0, xa = pv ? pa % Ws : 0;
        // A operands of the whole group first (independent LDS reads in flight together), then the MFMA chain
        real areg[NTAP * KQ];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int iy = ya * sy + oy[t], ix = xa * sx + ox[t];
            const bool valid = pv && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
            const real* ap = in + (valid ? (iy * Wi + ix) * CI : 0);
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const int c = kq * 4 + q, cc = c < CI ? c : CI - 1;
                const real v = ap[cc];
                areg[t * KQ + kq] = (valid && c < CI) ? v : real(0);
            }
        }
        d4v_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NTAP * KQ; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[i], breg[i], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int po = grp * 16 + q + 4 * e;
            if (po < NP && r < CO) {
                const int y = po / Ws, x = po % Ws;
                real v = acc[e];
                if (ELU_BIAS) v = elu_f(v + bias[r]);
                out[((y * osy + ooy) * Wo + x * osx + oox) * CO + r] = v;
            }
        }
    }
}

struct TapP { int oy, ox, woff, ooy, oox; };
template <int NTAP, int CI, int CO, typename TapFn>
__device__ __forceinline__ void gg_wgrad(const real* in, int Hi, int Wi, int Hs, int Ws, int sy, int sx, TapFn tapfn,
                                         const real* dout, int Wo, int osy, int osx, real* gW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = VAE_NT >> 6, r = lane & 15, q = lane >> 4;
    const int NP = Hs * Ws;
    const int rci = r < CI ? r : CI - 1, rco = r < CO ? r : CO - 1;
#pragma unroll 1
    for (int t = wave; t < NTAP; t += nwave) {
        const TapP tp = tapfn(t);
        d4v_t acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < NP; k0 += 16) {       // four k-steps per trip, operands fetched together
            real av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = k0 + 4 * u + q;
                const bool pv = p < NP;
                const int y = pv ? p / Ws : 0, x = pv ? p % Ws : 0;
                const int iy = y * sy + tp.oy, ix = x * sx + tp.ox;
                const bool valid = pv && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
                const real a0 = in[(valid ? (iy * Wi + ix) * CI : 0) + rci];
                const real b0 = dout[((y * osy + tp.ooy) * Wo + x * osx + tp.oox) * CO + rco];
                av[u] = (valid && r < CI) ? a0 : real(0);
                bv[u] = (pv && r < CO) ? b0 : real(0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = q + 4 * e;
            if (ci < CI && r < CO) gW[tp.woff + ci * CO + r] += acc[e];
        }
    }
}

// bias gradient: gb[co] += sum_pixels dpre[p][co]; 32 pixel chunks per channel combined through scratch
template <int NPIX, int COUT>
__device__ __forceinline__ void bias_grad(const real* dpre, real* gb, real* scratch) {
    constexpr int BCH = 32;
    if (threadIdx.x < BCH * COUT) {
        const int co = threadIdx.x % COUT, chunk2 = threadIdx.x / COUT;
        real s = 0;
        for (int p = chunk2; p < NPIX; p += BCH) s += dpre[p * COUT + co];
        scratch[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < COUT) {
        real t = 0;
#pragma unroll
        for (int c = 0; c < BCH; ++c) t += scratch[c * COUT + threadIdx.x];
        gb[threadIdx.x] += t;
    }
    __syncthreads();
}

// 3x3 convolution, stride STRIDE, no padding, no upsampling (the mnistVAE encoder layers, VAE_utils.py:117-122)
// on an LDS-resident NHWC tile, via the MFMA gather-GEMMs above.
template <int HS, int UPS, int PAD, int STRIDE, int CIN, int COUT, int HOUT>
struct Conv3 {
    static_assert(UPS == 1 && PAD == 0 && STRIDE == 2, "encoder layers: stride-2 valid convolutions");
    static constexpr int NW = 9 * CIN * COUT;
    static constexpr int NPIX = HOUT * HOUT;

    static __device__ void fwd(const real* in, const real* w, const real* bias, real* out) {
        if constexpr (CIN == 1) {
            // one input channel (first encoder layer): 9 MACs per output; the MFMA gather-GEMM would use 1 of 4 k-lanes
            for (int it = threadIdx.x; it < NPIX * COUT; it += VAE_NT) {
                const int co = it % COUT, p = it / COUT, y = p / HOUT, x = p % HOUT;
                real acc = bias[co];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc += in[(2 * y + ky) * HS + 2 * x + kx] * w[(ky * 3 + kx) * COUT + co];
                out[it] = elu_f(acc);
            }
            return;
        }
        const int oy[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2}, ox[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2};
        int wo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wo[t] = t * CIN * COUT;
        gg_fwd<9, false, true, CIN, COUT>(in, HS, HS, HOUT, HOUT, 2, 2, oy, ox, wo, w, COUT, bias, out, HOUT, 1, 1, 0, 0);
    }

    // din[iy][ix][ci] = sum_{ky,kx: (iy-ky), (ix-kx) even} dpre[(iy-ky)/2][(ix-kx)/2][co] w[ky][kx][ci][co]:
    // four input-parity classes, each a gather-GEMM with the taps of matching parity
    static __device__ void bwd_data(const real* dpre, const real* w, real* din) {
        // parity 0: ky in {0, 2} (offsets 0, -1);  parity 1: ky = 1 (offset 0)
        {   // (py, px) = (0, 0): 4 taps
            const int oy[4] = {0, 0, -1, -1}, ox[4] = {0, -1, 0, -1};
            const int wo[4] = {(0 * 3 + 0) * CIN * COUT, (0 * 3 + 2) * CIN * COUT, (2 * 3 + 0) * CIN * COUT, (2 * 3 + 2) * CIN * COUT};
            gg_fwd<4, true, false, COUT, CIN>(dpre, HOUT, HOUT, (HS + 1) / 2, (HS + 1) / 2, 1, 1, oy, ox, wo, w, COUT, nullptr,
                                              din, HS, 2, 2, 0, 0);
        }
        {   // (0, 1): ky in {0,2}, kx = 1
            const int oy[2] = {0, -1}, ox[2] = {0, 0};
            const int wo[2] = {(0 * 3 + 1) * CIN * COUT, (2 * 3 + 1) * CIN * COUT};
            gg_fwd<2, true, false, COUT, CIN>(dpre, HOUT, HOUT, (HS + 1) / 2, HS / 2, 1, 1, oy, ox, wo, w, COUT, nullptr, din,
                                              HS, 2, 2, 0, 1);
        }
        {   // (1, 0): ky = 1, kx in {0,2}
            const int oy[2] = {0, 0}, ox[2] = {0, -1};
            const int wo[2] = {(1 * 3 + 0) * CIN * COUT, (1 * 3 + 2) * CIN * COUT};
            gg_fwd<2, true, false, COUT, CIN>(dpre, HOUT, HOUT, HS / 2, (HS + 1) / 2, 1, 1, oy, ox, wo, w, COUT, nullptr, din,
                                              HS, 2, 2, 1, 0);
        }
        {   // (1, 1): ky = kx = 1
            const int oy[1] = {0}, ox[1] = {0};
            const int wo[1] = {(1 * 3 + 1) * CIN * COUT};
            gg_fwd<1, true, false, COUT, CIN>(dpre, HOUT, HOUT, HS / 2, HS / 2, 1, 1, oy, ox, wo, w, COUT, nullptr, din, HS, 2,
                                              2, 1, 1);
        }
    }

    // gw += sum_pixels in * dpre ; gb += sum_pixels dpre.  Ends with a barrier.
    static __device__ void bwd_weight(const real* in, const real* dpre, real* gw, real* gb, real* scratch) {
        if constexpr (CIN == 1) {
            // one input channel (first encoder layer): 9 * COUT outputs of NPIX MACs each.  On the MFMA this uses 1 of 16
            // rows (9.7 us of the 23.4 us launch, in-kernel timestamps); here thread = (output o = tap * COUT + co, pixel
            // chunk), the chunks are combined through LDS in fixed order (2.7 us incl. the bias gradient).
            constexpr int NO = 9 * COUT, NCH = VAE_NT / NO;
            static_assert(NCH >= 1 && NO * NCH <= VAE_SCRATCH, "chunk layout");
            const int o = threadIdx.x % NO, ch = threadIdx.x / NO, t = o / COUT, co = o % COUT, ky = t / 3, kx = t % 3;
            if (ch < NCH) {
                real acc = 0;
                for (int p = ch; p < NPIX; p += NCH) {
                    const int y = p / HOUT, x = p % HOUT;
                    acc += in[(2 * y + ky) * HS + 2 * x + kx] * dpre[p * COUT + co];
                }
                scratch[ch * NO + o] = acc;
            }
            __syncthreads();
            if (threadIdx.x < NO) {
                real tsum = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) tsum += scratch[c * NO + threadIdx.x];
                gw[threadIdx.x] += tsum;           // raw layout (ky, kx, 0, co) = o
            }
            __syncthreads();
        } else {
            auto tapfn = [](int t) { return TapP{t / 3, t % 3, t * CIN * COUT, 0, 0}; };
            gg_wgrad<9, CIN, COUT>(in, HS, HS, HOUT, HOUT, 2, 2, tapfn, dpre, HOUT, 1, 1, gw);
        }
        bias_grad<NPIX, COUT>(dpre, gb, scratch);
    }
};

// ---------------------------------------------------------------------------------------------
// UpSampling2D(2) + 3x3 convolution (stride 1, PAD 0|1) as FOUR parity-specific 2x2 convolutions on
// the low-resolution stored input (HS x HS x CIN): for output row y, base = y - PAD, parity
// pi = base & 1, Y = base >> 1, the three taps ky read source rows Y + T(pi,ky) with
// T(pi,k) = (k + pi >= 2), so taps sharing a source row are pre-summed into effective weights
//   We[pi_y][pi_x][ty][tx][ci][co] = sum_{ky: T(pi_y,ky)=ty} sum_{kx: T(pi_x,kx)=tx} w[ky][kx][ci][co].
// 4 taps instead of 9 in the forward, 16 instead of 36 in the data gradient, and the weight gradient
// is accumulated on We (folded back to w once per workgroup).  Mathematically identical to the
// reference's UpSampling2D + Conv2D (VAE_utils.py:132-140); summation order differs (1e-16 level).
// ---------------------------------------------------------------------------------------------
template <int HS, int PAD, int CIN, int COUT>
struct UpConv3 {
    static constexpr int HOUT = 2 * HS - 2 + 2 * PAD;
    static constexpr int NPIX = HOUT * HOUT;
    static constexpr int NWE = 16 * CIN * COUT;          // effective weights
    static constexpr int NW = 9 * CIN * COUT;            // raw weights
    static constexpr int COG = (COUT % 2 == 0) ? 2 : 1;
    static constexpr int NCG = COUT / COG;
    static constexpr int CIG = (CIN % 2 == 0) ? 2 : 1;
    static constexpr int NIG = CIN / CIG;
    static __device__ __forceinline__ int T(int pi, int k) { return (k + pi >= 2) ? 1 : 0; }

    // We (LDS) from raw w (global or LDS)
    static __device__ void build_weff(const real* w, real* We) {
        for (int e = threadIdx.x; e < NWE; e += VAE_NT) {
            const int co = e % COUT, ci = (e / COUT) % CIN, tap = (e / (COUT * CIN)) % 4, cls = e / (COUT * CIN * 4);
            const int ty = tap >> 1, tx = tap & 1, py = cls >> 1, px = cls & 1;
            real s = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    if (T(py, ky) == ty && T(px, kx) == tx) s += w[((ky * 3 + kx) * CIN + ci) * COUT + co];
            We[e] = s;
        }
    }
    // raw-weight gradient from the effective-weight gradient: gw[ky][kx] = sum_classes gWe[cls][T,T]
    static __device__ void fold_grad(const real* gWe, real* gw) {
        for (int e = threadIdx.x; e < NW; e += VAE_NT) {
            const int co = e % COUT, ci = (e / COUT) % CIN, kx = (e / (COUT * CIN)) % 3, ky = e / (COUT * CIN * 3);
            real s = 0;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
                    s += gWe[(((py * 2 + px) * 4 + T(py, ky) * 2 + T(px, kx)) * CIN + ci) * COUT + co];
            gw[e] = s;
        }
    }

    // ---- VALU variants (register-tiled LDS loops): faster than the MFMA forms where the tile would be mostly
    //      padding (single output channel) or the gather arithmetic dominates; chosen per layer from ablation timings
    // out = elu(conv(up(in)) + bias); item = (output pixel, group of COG channels)
    static __device__ void fwd_valu(const real* in, const real* We, const real* bias, real* out) {
        for (int it = threadIdx.x; it < NPIX * NCG; it += VAE_NT) {
            const int cg = it % NCG, p = it / NCG, x = p % HOUT, y = p / HOUT;
            const int by = y - PAD, bx = x - PAD, py = by & 1, px = bx & 1, Y = by >> 1, X = bx >> 1;
            real acc[COG];
#pragma unroll
            for (int g = 0; g < COG; ++g) acc[g] = bias[cg * COG + g];
            const real* wc = We + ((py * 2 + px) * 4) * CIN * COUT + cg * COG;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int sy = Y + ty;
                const bool vy = (unsigned)sy < (unsigned)HS;
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    const int sx = X + tx;
                    const bool valid = vy && ((unsigned)sx < (unsigned)HS);
                    const real* src = in + (valid ? (sy * HS + sx) * CIN : 0);
                    const real* wk = wc + (ty * 2 + tx) * CIN * COUT;
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        const real a = valid ? src[ci] : real(0);
#pragma unroll
                        for (int g = 0; g < COG; ++g) acc[g] += a * wk[ci * COUT + g];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < COG; ++g) out[p * COUT + cg * COG + g] = elu_f(acc[g]);
        }
    }

    // din (HS x HS x CIN) from dpre (HOUT x HOUT x COUT); item = (stored pixel, group of CIG channels)
    static __device__ void bwd_data_valu(const real* dpre, const real* We, real* din) {
        for (int it = threadIdx.x; it < HS * HS * NIG; it += VAE_NT) {
            const int ig = it % NIG, ps = it / NIG, Xs = ps % HS, Ys = ps / HS;
            real acc[CIG];
#pragma unroll
            for (int g = 0; g < CIG; ++g) acc[g] = 0;
#pragma unroll 1
            for (int cy = 0; cy < 4; ++cy) {            // (pi_y, ty)
                const int py = cy >> 1, ty = cy & 1;
                const int y = 2 * (Ys - ty) + py + PAD;
                const bool vy = (unsigned)y < (unsigned)HOUT;
#pragma unroll
                for (int cx = 0; cx < 4; ++cx) {        // (pi_x, tx)
                    const int px = cx >> 1, tx = cx & 1;
                    const int x = 2 * (Xs - tx) + px + PAD;
                    const bool valid = vy && ((unsigned)x < (unsigned)HOUT);
                    const real* dp = dpre + (valid ? (y * HOUT + x) * COUT : 0);
                    const real* wk = We + (((py * 2 + px) * 4 + ty * 2 + tx) * CIN + ig * CIG) * COUT;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const real d = valid ? dp[co] : real(0);
#pragma unroll
                        for (int g = 0; g < CIG; ++g) acc[g] += d * wk[g * COUT + co];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < CIG; ++g) din[ps * CIN + ig * CIG + g] = acc[g];
        }
    }

    // gWe += sum_pixels in * dpre ; gb += sum_pixels dpre.  item = (class, tap, ci, pixel chunk) -> COUT
    // outputs; chunks combined through `scratch` (>= VAE_SCRATCH reals) in fixed order.  Ends with a barrier.
    static __device__ void bwd_weight_valu(const real* in, const real* dpre, real* gWe, real* gb, real* scratch) {
        constexpr int NTC = 16 * CIN;
        constexpr int NCH0 = VAE_NT / NTC, NCH1 = VAE_SCRATCH / NWE;
        constexpr int NCH = NCH0 < NCH1 ? (NCH0 < 1 ? 1 : NCH0) : NCH1;
        constexpr int NG = HS + 1;                      // candidate Y (and X) values: -1 .. HS-1
        const int tc = threadIdx.x % NTC, chunk = threadIdx.x / NTC;
        if (chunk < NCH) {
            const int ci = tc % CIN, tap = (tc / CIN) % 4, cls = tc / (CIN * 4);
            const int ty = tap >> 1, tx = tap & 1, py = cls >> 1, px = cls & 1;
            real acc[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = 0;
#pragma unroll 2
            for (int idx = chunk; idx < NG * NG; idx += NCH) {
                const int Y = idx / NG - 1, X = idx % NG - 1;
                const int y = 2 * Y + py + PAD, x = 2 * X + px + PAD, sy = Y + ty, sx = X + tx;
                const bool valid = ((unsigned)y < (unsigned)HOUT) && ((unsigned)x < (unsigned)HOUT) &&
                                   ((unsigned)sy < (unsigned)HS) && ((unsigned)sx < (unsigned)HS);
                const real a = valid ? in[(sy * HS + sx) * CIN + ci] : real(0);
                const real* dp = dpre + (valid ? (y * HOUT + x) * COUT : 0);
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] += a * dp[co];
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co) scratch[chunk * NWE + tc * COUT + co] = acc[co];
        }
        __syncthreads();
        for (int widx = threadIdx.x; widx < NWE; widx += VAE_NT) {
            real s = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) s += scratch[c * NWE + widx];
            gWe[widx] += s;
        }
        __syncthreads();
        constexpr int BCH = 32;
        if (threadIdx.x < BCH * COUT) {
            const int co = threadIdx.x % COUT, chunk2 = threadIdx.x / COUT;
            real s = 0;
            for (int p = chunk2; p < NPIX; p += BCH) s += dpre[p * COUT + co];
            scratch[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x < COUT) {
            real t = 0;
#pragma unroll
            for (int c = 0; c < BCH; ++c) t += scratch[c * COUT + threadIdx.x];
            gb[threadIdx.x] += t;
        }
        __syncthreads();
    }

    // out = elu(conv(up(in)) + bias): four output-parity classes, each a 4-tap MFMA gather-GEMM on the low-res input
    static __device__ void fwd_mfma(const real* in, const real* We, const real* bias, real* out) {
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int opy = cls >> 1, opx = cls & 1;
            const int by = opy - PAD, bx = opx - PAD, py = by & 1, px = bx & 1, dY = (by - py) / 2, dX = (bx - px) / 2;
            const int oy[4] = {dY, dY, dY + 1, dY + 1}, ox[4] = {dX, dX + 1, dX, dX + 1};
            int wo[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wo[t] = ((py * 2 + px) * 4 + t) * CIN * COUT;
            gg_fwd<4, false, true, CIN, COUT>(in, HS, HS, HOUT / 2, HOUT / 2, 1, 1, oy, ox, wo, We, COUT, bias, out, HOUT, 2, 2,
                                              opy, opx);
        }
    }

    // din (HS x HS x CIN) from dpre (HOUT x HOUT x COUT): one 16-tap gather-GEMM with input stride 2 over dpre
    static __device__ void bwd_data_mfma(const real* dpre, const real* We, real* din) {
        int oy[16], ox[16], wo[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int py = (t >> 3) & 1, ty = (t >> 2) & 1, px = (t >> 1) & 1, tx = t & 1;
            oy[t] = py + PAD - 2 * ty;
            ox[t] = px + PAD - 2 * tx;
            wo[t] = ((py * 2 + px) * 4 + ty * 2 + tx) * CIN * COUT;
        }
        gg_fwd<16, true, false, COUT, CIN>(dpre, HOUT, HOUT, HS, HS, 2, 2, oy, ox, wo, We, COUT, nullptr, din, HS, 1, 1, 0, 0);
    }

    // gWe += sum_pixels in * dpre ; gb += sum_pixels dpre.  Ends with a barrier.
    static __device__ void bwd_weight_mfma(const real* in, const real* dpre, real* gWe, real* gb, real* scratch) {
        if (COUT >= 2) {
            // 16 (class, tap) pairs = 16 "taps" of one gather-GEMM over the HOUT/2 x HOUT/2 class grid
            auto tapfn = [](int t) {
                const int opy = (t >> 3) & 1, opx = (t >> 2) & 1, ty = (t >> 1) & 1, tx = t & 1;
                const int by = opy - PAD, bx = opx - PAD, py = by & 1, px = bx & 1;
                return TapP{(by - py) / 2 + ty, (bx - px) / 2 + tx, ((py * 2 + px) * 4 + ty * 2 + tx) * CIN * COUT, opy, opx};
            };
            gg_wgrad<16, CIN, COUT>(in, HS, HS, HOUT / 2, HOUT / 2, 1, 1, tapfn, dpre, HOUT, 2, 2, gWe);
        } else {
            // single output channel: item = (class, tap, pixel chunk), 32 chunks; all CIN inputs of a pixel per item
            constexpr int NCH = VAE_NT / 16;
            constexpr int NG = HS + 1;                  // candidate Y (and X): -1 .. HS-1
            const int ct = threadIdx.x & 15, chunk = threadIdx.x >> 4;
            const int cls = ct >> 2, tap = ct & 3, ty = tap >> 1, tx = tap & 1, py = cls >> 1, px = cls & 1;
            real acc[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0;
            for (int idx = chunk; idx < NG * NG; idx += NCH) {
                const int Y = idx / NG - 1, X = idx % NG - 1;
                const int y = 2 * Y + py + PAD, x = 2 * X + px + PAD, sy = Y + ty, sx = X + tx;
                const bool valid = ((unsigned)y < (unsigned)HOUT) && ((unsigned)x < (unsigned)HOUT) &&
                                   ((unsigned)sy < (unsigned)HS) && ((unsigned)sx < (unsigned)HS);
                const real d = valid ? dpre[y * HOUT + x] : real(0);
                const real* ip = in + (valid ? (sy * HS + sx) * CIN : 0);
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[ci] += ip[ci] * d;
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) scratch[chunk * (16 * CIN) + ct * CIN + ci] = acc[ci];
            __syncthreads();
            for (int widx = threadIdx.x; widx < 16 * CIN; widx += VAE_NT) {
                real s = 0;
#pragma unroll 8
                for (int c = 0; c < NCH; ++c) s += scratch[c * (16 * CIN) + widx];
                gWe[widx] += s;
            }
        }
        __syncthreads();
        bias_grad<NPIX, COUT>(dpre, gb, scratch);
    }
};

using UpC1 = UpConv3<4, 1, 8, 8>;     // (4,4,8)  -> up 8x8   -> same  -> (8,8,8)
using UpC2 = UpConv3<8, 0, 8, 8>;     // (8,8,8)  -> up 16x16 -> valid -> (14,14,8)
using UpC3 = UpConv3<14, 1, 8, 1>;    // (14,14,8)-> up 28x28 -> same  -> (28,28,1)
#define DEC_NWE (UpC1::NWE + UpC2::NWE + UpC3::NWE)   // 1024 + 1024 + 128

using EncC1 = Conv3<28, 1, 0, 2, 1, 8, 13>;
using EncC2 = Conv3<13, 1, 0, 2, 8, 8, 6>;
using EncC3 = Conv3<6, 1, 0, 2, 8, 8, 2>;

__device__ __forceinline__ void lds_copy_in(real* dst, const real* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ void lds_copy_out(real* __restrict__ dst, const real* src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ void lds_zero(real* dst, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = 0;
}

struct EncOff { int c1w, c1b, c2w, c2b, c3w, c3b, dw, db, n; };
struct DecOff { int dw, db, c1w, c1b, c2w, c2b, c3w, c3b, n; };

__device__ __host__ inline EncOff enc_off(int L) {
    EncOff o; int p = 0;
    o.c1w = p; p += 72; o.c1b = p; p += 8; o.c2w = p; p += 576; o.c2b = p; p += 8;
    o.c3w = p; p += 576; o.c3b = p; p += 8; o.dw = p; p += 32 * 2 * L; o.db = p; p += 2 * L; o.n = p;
    return o;
}
__device__ __host__ inline DecOff dec_off(int L) {
    DecOff o; int p = 0;
    o.dw = p; p += L * 128; o.db = p; p += 128; o.c1w = p; p += 576; o.c1b = p; p += 8;
    o.c2w = p; p += 576; o.c2b = p; p += 8; o.c3w = p; p += 72; o.c3b = p; p += 1; o.n = p;
    return o;
}

// ------------------------------------------------------------------------------------------
// encoder forward: images -> a1,a2,a3 (saved), qnet_mu, qnet_var_raw = exp(.), qnet_var = clip
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VAE_NT) void k_encoder_fwd(int b, int L, int clip, const real* __restrict__ th_enc,
                                                            const real* __restrict__ images, real* __restrict__ a1g,
                                                            real* __restrict__ a2g, real* __restrict__ a3g,
                                                            real* __restrict__ mu, real* __restrict__ var_raw,
                                                            real* __restrict__ var, int n_img_blocks, SvgpKernArgs ka,
                                                            real* __restrict__ Kmm, real* __restrict__ Knm,
                                                            real* __restrict__ knn) {
    extern __shared__ __align__(16) real smem[];
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
// encoder reverse: (ybar, s2bar) -> encoder weight-gradient partials
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VAE_NT) void k_encoder_bwd(int b, int L, int clip, const real* __restrict__ th_enc,
                                                            const real* __restrict__ images,
                                                            const real* __restrict__ a1g, const real* __restrict__ a2g,
                                                            const real* __restrict__ a3g,
                                                            const real* __restrict__ var_raw,
                                                            const real* __restrict__ ybar,
                                                            const real* __restrict__ s2bar, real* __restrict__ part) {
    extern __shared__ __align__(16) real smem[];
    const EncOff eo = enc_off(L);
    real* w = smem;                  // eo.n
    real* g = w + eo.n;              // eo.n   gradient accumulators
    real* img = g + eo.n;            // 784
    real* a1 = img + 784;            // 1352
    real* a2 = a1 + 1352;            // 288
    real* a3 = a2 + 288;             // 32
    real* d1 = a3 + 32;              // 1352
    real* d2 = d1 + 1352;            // 288
    real* d3 = d2 + 288;             // 32
    real* dout = d3 + 32;            // 2L (<=128)
    real* scratch = dout + 128;      // VAE_SCRATCH
    lds_copy_in(w, th_enc, eo.n);
    lds_zero(g, eo.n);
    const int twoL = 2 * L;
    for (int n = blockIdx.x; n < b; n += gridDim.x) {
        __syncthreads();
        lds_copy_in(img, images + (size_t)n * 784, 784);
        lds_copy_in(a1, a1g + (size_t)n * 1352, 1352);
        lds_copy_in(a2, a2g + (size_t)n * 288, 288);
        lds_copy_in(a3, a3g + (size_t)n * 32, 32);
        for (int j = threadIdx.x; j < twoL; j += blockDim.x) {
            real dj;
            if (j < L) {
                dj = ybar[(size_t)n * L + j];
            } else {
                const real vr = var_raw[(size_t)n * L + j - L];
                const bool pass = !clip || (vr >= 1e-3 && vr <= 10.0);
                dj = pass ? s2bar[(size_t)n * L + j - L] * vr : real(0);
            }
            dout[j] = dj;
        }
        __syncthreads();
        // dense: weight / bias gradients and da3
        for (int o = threadIdx.x; o < 32 * twoL; o += blockDim.x) g[eo.dw + o] += a3[o / twoL] * dout[o % twoL];
        for (int j = threadIdx.x; j < twoL; j += blockDim.x) g[eo.db + j] += dout[j];
        if (threadIdx.x < 32) {
            real acc = 0;
            for (int j = 0; j < twoL; ++j) acc += dout[j] * w[eo.dw + threadIdx.x * twoL + j];
            d3[threadIdx.x] = acc * elu_grad_from_out(a3[threadIdx.x]);
        }
        __syncthreads();
        EncC3::bwd_weight(a2, d3, g + eo.c3w, g + eo.c3b, scratch);
        EncC3::bwd_data(d3, w + eo.c3w, d2);
        __syncthreads();
        for (int i = threadIdx.x; i < 288; i += blockDim.x) d2[i] *= elu_grad_from_out(a2[i]);
        __syncthreads();
        EncC2::bwd_weight(a1, d2, g + eo.c2w, g + eo.c2b, scratch);
        EncC2::bwd_data(d2, w + eo.c2w, d1);
        __syncthreads();
        for (int i = threadIdx.x; i < 1352; i += blockDim.x) d1[i] *= elu_grad_from_out(a1[i]);
        __syncthreads();
        EncC1::bwd_weight(img, d1, g + eo.c1w, g + eo.c1b, scratch);
    }
    __syncthreads();
    lds_copy_out(part + (size_t)blockIdx.x * eo.n, g, eo.n);
}

// ------------------------------------------------------------------------------------------
// decoder forward: z -> h0, a1, a2 (saved), recon, per-workgroup sum of squared errors
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VAE_NT) void k_decoder_fwd(int b, int L, const real* __restrict__ th_dec,
                                                            const real* __restrict__ images,
                                                            const real* __restrict__ zg, real* __restrict__ h0g,
                                                            real* __restrict__ a1g, real* __restrict__ a2g,
                                                            real* __restrict__ recon, real* __restrict__ part_sums) {
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
    __syncthreads();
    UpC1::build_weff(w + od.c1w, We1);       // from the LDS copy (one coalesced read of the raw weights)
    UpC2::build_weff(w + od.c2w, We2);
    UpC3::build_weff(w + od.c3w, We3);
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
                                                            KmScatter ks) {
    __shared__ real s[16][17];
    __shared__ real red[16];
    if ((int)blockIdx.x > nb_enc + nb_dec) {
        svgp_km_scatter_block(blockIdx.x - (nb_enc + nb_dec + 1), ks.n_blocks, b, ks.M, ks.n_obj, ks.aux, ks.n_gp_part,
                              ks.train_gp, ks.train_ov, ks.d_on, ks.part_gp, ks.d_ov, ks.d_ls, ks.d_amp);
        return;
    }
    if ((int)blockIdx.x == nb_enc + nb_dec) {
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
    const bool enc = (int)blockIdx.x < nb_enc;
    const int n = enc ? n_enc : n_dec, blk = enc ? blockIdx.x : blockIdx.x - nb_enc;
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
    hipLaunchKernelGGL(k_encoder_fwd, dim3(n_img + n_km), dim3(VAE_NT), lds, (hipStream_t)stream, c->b, c->L,
                       c->clip_qs, theta, images, ws + wl.enc_a1, ws + wl.enc_a2, ws + wl.enc_a3, ws + wl.qnet_mu,
                       ws + wl.qnet_var_raw, ws + wl.qnet_var, n_img, ka, ws + wl.K, ws + wl.Kn, ws + wl.knn);
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
    const size_t lds = (size_t)(2 * pl.n_enc + 784 + 1352 + 288 + 32 + 1352 + 288 + 32 + 128 + VAE_SCRATCH) * sizeof(real);
    int rc = set_dyn_lds(k_encoder_bwd, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_encoder_bwd, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, c->b, c->L,
                       c->clip_qs, theta, images, ws + wl.enc_a1, ws + wl.enc_a2, ws + wl.enc_a3,
                       ws + wl.qnet_var_raw, ws + wl.ybar, ws + wl.s2bar, ws + wl.part_enc);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_mnist_decoder_fwd(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                      void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws, SVGP_ERR_INVALID, "NULL device pointer");
    const int64_t n_dec = pl.n_vae - pl.n_enc;
    const size_t lds = (size_t)(n_dec + 64 + 128 + 512 + 1568 + 784 + 16 + DEC_NWE) * sizeof(real);
    int rc = set_dyn_lds(k_decoder_fwd, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_decoder_fwd, dim3(svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, c->b, c->L,
                       theta + pl.n_enc, images, ws + wl.z, ws + wl.dec_h0, ws + wl.dec_a1, ws + wl.dec_a2,
                       ws + wl.recon, ws + wl.part_sums);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
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

static int grad_reduce_impl(const svgp_mnist_cfg* c, const double* aux, double* ws, bool with_scatter, void* stream) {
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
    hipLaunchKernelGGL(k_grad_reduce, dim3(nb_enc + nb_dec + 1 + ks.n_blocks), dim3(SVGP_BLOCK), lds, (hipStream_t)stream,
                       n_part, n_enc, n_dec, nb_enc, nb_dec, svgp_n_post_actual(c), c->b, ws + wl.part_enc, ws + wl.part_dec,
                       ws + wl.part_sums, c->titsias ? ws + wl.tit_scal + 2 * c->L : (const real*)nullptr, ws + wl.grad,
                       ws + wl.sums, ks);
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
