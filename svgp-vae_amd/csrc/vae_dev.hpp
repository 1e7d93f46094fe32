// Device code of the mnistVAE kernels (vae_mnist.hip) that other translation units instantiate too: the decoder's
// weight-gradient work rides as extra workgroups of the GP reverse factor launch (gp_kernels.hip), which runs
// SVGP_BLOCK = 256 threads per workgroup, so every helper takes its thread count NT as a template argument.
#pragma once
#include "common.hpp"

namespace svgp_vae {

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
template <int NTAP, bool TW, bool ELU_BIAS, int CI, int CO, int NT = VAE_NT>
__device__ __forceinline__ void gg_fwd(const real* in, int Hi, int Wi, int Hs, int Ws, int sy, int sx,
                                       const int (&oy)[NTAP], const int (&ox)[NTAP], const int (&woff)[NTAP],
                                       const real* W, int ldw, const real* bias, real* out, int Wo, int osy, int osx,
                                       int ooy, int oox) {
    constexpr int KQ = (CI + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = NT >> 6, r = lane & 15, q = lane >> 4;
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
template <int NTAP, int CI, int CO, int NT, typename TapFn>
__device__ __forceinline__ void gg_wgrad(const real* in, int Hi, int Wi, int Hs, int Ws, int sy, int sx, TapFn tapfn,
                                         const real* dout, int Wo, int osy, int osx, real* gW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = NT >> 6, r = lane & 15, q = lane >> 4;
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
template <int NPIX, int COUT, int NT = VAE_NT>
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
template <int HS, int UPS, int PAD, int STRIDE, int CIN, int COUT, int HOUT, int NT = VAE_NT>
struct Conv3 {
    static_assert(UPS == 1 && PAD == 0 && STRIDE == 2, "encoder layers: stride-2 valid convolutions");
    static constexpr int NW = 9 * CIN * COUT;
    static constexpr int NPIX = HOUT * HOUT;

    static __device__ void fwd(const real* in, const real* w, const real* bias, real* out) {
        if constexpr (CIN == 1) {
            // one input channel (first encoder layer): 9 MACs per output; the MFMA gather-GEMM would use 1 of 4 k-lanes
            for (int it = threadIdx.x; it < NPIX * COUT; it += NT) {
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
        gg_fwd<9, false, true, CIN, COUT, NT>(in, HS, HS, HOUT, HOUT, 2, 2, oy, ox, wo, w, COUT, bias, out, HOUT, 1, 1, 0, 0);
    }

    // din[iy][ix][ci] = sum_{ky,kx: (iy-ky), (ix-kx) even} dpre[(iy-ky)/2][(ix-kx)/2][co] w[ky][kx][ci][co]:
    // four input-parity classes, each a gather-GEMM with the taps of matching parity
    static __device__ void bwd_data(const real* dpre, const real* w, real* din) {
        // parity 0: ky in {0, 2} (offsets 0, -1);  parity 1: ky = 1 (offset 0)
        {   // (py, px) = (0, 0): 4 taps
            const int oy[4] = {0, 0, -1, -1}, ox[4] = {0, -1, 0, -1};
            const int wo[4] = {(0 * 3 + 0) * CIN * COUT, (0 * 3 + 2) * CIN * COUT, (2 * 3 + 0) * CIN * COUT, (2 * 3 + 2) * CIN * COUT};
            gg_fwd<4, true, false, COUT, CIN, NT>(dpre, HOUT, HOUT, (HS + 1) / 2, (HS + 1) / 2, 1, 1, oy, ox, wo, w, COUT, nullptr,
                                              din, HS, 2, 2, 0, 0);
        }
        {   // (0, 1): ky in {0,2}, kx = 1
            const int oy[2] = {0, -1}, ox[2] = {0, 0};
            const int wo[2] = {(0 * 3 + 1) * CIN * COUT, (2 * 3 + 1) * CIN * COUT};
            gg_fwd<2, true, false, COUT, CIN, NT>(dpre, HOUT, HOUT, (HS + 1) / 2, HS / 2, 1, 1, oy, ox, wo, w, COUT, nullptr, din,
                                              HS, 2, 2, 0, 1);
        }
        {   // (1, 0): ky = 1, kx in {0,2}
            const int oy[2] = {0, 0}, ox[2] = {0, -1};
            const int wo[2] = {(1 * 3 + 0) * CIN * COUT, (1 * 3 + 2) * CIN * COUT};
            gg_fwd<2, true, false, COUT, CIN, NT>(dpre, HOUT, HOUT, HS / 2, (HS + 1) / 2, 1, 1, oy, ox, wo, w, COUT, nullptr, din,
                                              HS, 2, 2, 1, 0);
        }
        {   // (1, 1): ky = kx = 1
            const int oy[1] = {0}, ox[1] = {0};
            const int wo[1] = {(1 * 3 + 1) * CIN * COUT};
            gg_fwd<1, true, false, COUT, CIN, NT>(dpre, HOUT, HOUT, HS / 2, HS / 2, 1, 1, oy, ox, wo, w, COUT, nullptr, din, HS, 2,
                                              2, 1, 1);
        }
    }

    // gw += sum_pixels in * dpre ; gb += sum_pixels dpre.  Ends with a barrier.
    static __device__ void bwd_weight(const real* in, const real* dpre, real* gw, real* gb, real* scratch) {
        if constexpr (CIN == 1) {
            // one input channel (first encoder layer): 9 * COUT outputs of NPIX MACs each.  On the MFMA this uses 1 of 16
            // rows (9.7 us of the 23.4 us launch, in-kernel timestamps); here thread = (output o = tap * COUT + co, pixel
            // chunk), the chunks are combined through LDS in fixed order (2.7 us incl. the bias gradient).
            constexpr int NO = 9 * COUT, NCH = NT / NO;
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
            gg_wgrad<9, CIN, COUT, NT>(in, HS, HS, HOUT, HOUT, 2, 2, tapfn, dpre, HOUT, 1, 1, gw);
        }
        bias_grad<NPIX, COUT, NT>(dpre, gb, scratch);
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
template <int HS, int PAD, int CIN, int COUT, int NT = VAE_NT>
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
        for (int e = threadIdx.x; e < NWE; e += NT) {
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
        for (int e = threadIdx.x; e < NW; e += NT) {
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
        for (int it = threadIdx.x; it < NPIX * NCG; it += NT) {
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
        for (int it = threadIdx.x; it < HS * HS * NIG; it += NT) {
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
        constexpr int NCH0 = NT / NTC, NCH1 = VAE_SCRATCH / NWE;
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
        for (int widx = threadIdx.x; widx < NWE; widx += NT) {
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
            gg_fwd<4, false, true, CIN, COUT, NT>(in, HS, HS, HOUT / 2, HOUT / 2, 1, 1, oy, ox, wo, We, COUT, bias, out, HOUT, 2, 2,
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
        gg_fwd<16, true, false, COUT, CIN, NT>(dpre, HOUT, HOUT, HS, HS, 2, 2, oy, ox, wo, We, COUT, nullptr, din, HS, 1, 1, 0, 0);
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
            gg_wgrad<16, CIN, COUT, NT>(in, HS, HS, HOUT / 2, HOUT / 2, 1, 1, tapfn, dpre, HOUT, 2, 2, gWe);
        } else {
            // single output channel: item = (class, tap, pixel chunk), 32 chunks; all CIN inputs of a pixel per item
            constexpr int NCH = NT / 16;
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
            for (int widx = threadIdx.x; widx < 16 * CIN; widx += NT) {
                real s = 0;
#pragma unroll 8
                for (int c = 0; c < NCH; ++c) s += scratch[c * (16 * CIN) + widx];
                gWe[widx] += s;
            }
        }
        __syncthreads();
        bias_grad<NPIX, COUT, NT>(dpre, gb, scratch);
    }
};

template <int NT> using UpC1T = UpConv3<4, 1, 8, 8, NT>;
using UpC1 = UpC1T<VAE_NT>;     // (4,4,8)  -> up 8x8   -> same  -> (8,8,8)
template <int NT> using UpC2T = UpConv3<8, 0, 8, 8, NT>;
using UpC2 = UpC2T<VAE_NT>;     // (8,8,8)  -> up 16x16 -> valid -> (14,14,8)
template <int NT> using UpC3T = UpConv3<14, 1, 8, 1, NT>;
using UpC3 = UpC3T<VAE_NT>;    // (14,14,8)-> up 28x28 -> same  -> (28,28,1)
#define DEC_NWE 2176   // 1024 + 1024 + 128
static_assert(UpC1::NWE + UpC2::NWE + UpC3::NWE == DEC_NWE, "effective weights of the three up-convolutions");

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
// Decoder reverse pass in two halves (round 6).  The DATA half (k_decoder_bwd_data, vae_mnist.hip) is the chain
// d recon -> d2 -> d1 -> dh0 -> zbar that the GP reverse stages wait for; it stores the pre-activation gradients
// d2 (14,14,8), d1 (8,8,8), dh0 (128) of every image.  The WEIGHT half below needs only those and the stored forward
// activations and feeds nothing but the closing gradient reduction, so it runs as extra workgroups ("riders") of a launch
// that leaves most of the chip idle anyway: the reverse factor stage (L workgroups on 256 CUs, gp_kernels.hip).
// One rider = NT threads walking images slot, slot + n_slots, ... for ONE group of layers; weight gradients accumulate in LDS (the
// convolutions' on the effective parity-class weights); one partial (raw-parameter layout, od.n reals) per slot, its ranges
// written by the slot's riders, summed over slots in fixed order by svgp_mnist_grad_reduce.  LDS: dec_wgrad_lds(NT, L, n_types).
// Reference semantics: tf.gradients of VAE_utils.py:128-141,154-162 w.r.t. the decoder variables (MNIST_experiment.py:202-205).
// ------------------------------------------------------------------------------------------
struct DecWgradArgs {
    int b, L, geco, n_slots, n_types;    // riders = n_slots * n_types; rider r: partial slot r / n_types, layer group r % n_types
    real inv_bglobal;
    const real* state; const real* images; const real* z; const real* h0; const real* a1; const real* a2; const real* recon;
    const real* d2; const real* d1; const real* dh0;
    real* part;                          // (n_slots, od.n)
};
// Layer groups of a rider (bit 0: UpC3, bit 1: UpC2, bit 2: UpC1 + dense).  n_types 1: one rider per image does everything;
// 2: {UpC3} | {UpC2, UpC1, dense}; 3: one group each.  The groups of a slot write disjoint ranges of the same partial.
__host__ __device__ constexpr int dec_wgrad_mask(int n_types, int ty) { return n_types == 1 ? 7 : n_types == 2 ? (ty == 0 ? 1 : 6) : (1 << ty); }
// LDS reals of one rider: inputs + accumulators of its groups + bias accumulators + the chunk-reduction scratch
// (NT / 16 chunks x 128 for UpC3, NT / 128 chunks x 1024 for UpC2, 256 for the bias sums alone)
__host__ __device__ constexpr int dec_wgrad_lds_mask(int nt, int L, int mask) {
    return ((mask & 1) ? 1568 + 784 + 128 : 0) + ((mask & 2) ? 512 + 1568 + 1024 : 0) +
           ((mask & 4) ? 128 + 512 + 128 + 64 + 1024 + L * 128 + 128 : 0) + 32 + ((mask & 3) ? (nt <= 256 ? 2048 : 4096) : 256);
}
__host__ __device__ constexpr int dec_wgrad_lds(int nt, int L, int n_types) {
    int mx = 0;
    for (int ty = 0; ty < n_types; ++ty) {
        const int v = dec_wgrad_lds_mask(nt, L, dec_wgrad_mask(n_types, ty));
        mx = v > mx ? v : mx;
    }
    return mx;
}

template <int NT>
__device__ __forceinline__ void decoder_wgrad_rider(const DecWgradArgs& a, int rider, real* smem) {
    static_assert(NT >= 256 && NT <= 512, "bias / chunk layouts are sized for 256..512 threads");
    const DecOff od = dec_off(a.L);
    // Riders are numbered TYPE-major, heaviest layer group first (UpC2: 113 K MACs per image, UpC3: 56 K, UpC1 + dense: 39 K): a launch that
    // cannot hold all of them at once (the reverse factor launch: 848 workgroups for 768 slots) then starts its LAST riders, the ones
    // that wait for a slot, on the lightest group.  (Slot-major, types interleaved: the late riders included UpC2 ones and the launch
    // ended with them, 20.5 us against 17.2 for the channel chain alone.)
    const int tyo = rider / a.n_slots, slot = rider - tyo * a.n_slots;
    const int ty = a.n_types == 3 ? (tyo == 0 ? 1 : tyo == 1 ? 0 : 2) : a.n_types == 2 ? 1 - tyo : 0;
    const int mask = dec_wgrad_mask(a.n_types, ty);
    const int ndense = a.L * 128;
    real* p = smem;
    auto take = [&](int n) { real* r = p; p += n; return r; };
    real *a2 = nullptr, *d3 = nullptr, *gWe3 = nullptr, *a1 = nullptr, *d2 = nullptr, *gWe2 = nullptr, *h0 = nullptr, *d1 = nullptr,
         *dh0 = nullptr, *z = nullptr, *gWe1 = nullptr, *dacc = nullptr;
    if (mask & 1) { a2 = take(1568); d3 = take(784); gWe3 = take(128); }
    if (mask & 2) { a1 = take(512); d2 = take(1568); gWe2 = take(1024); }
    if (mask & 4) { h0 = take(128); d1 = take(512); dh0 = take(128); z = take(64); gWe1 = take(1024); dacc = take(ndense + 128); }
    real* gb = take(32);             // c1b (8) | c2b (8) | c3b (1)
    real* scratch = p;
    if (mask & 1) for (int i = threadIdx.x; i < 128; i += NT) gWe3[i] = 0;
    if (mask & 2) for (int i = threadIdx.x; i < 1024; i += NT) gWe2[i] = 0;
    if (mask & 4) for (int i = threadIdx.x; i < 1024 + ndense + 128; i += NT) gWe1[i] = 0;     // gWe1 | dacc (contiguous)
    if (threadIdx.x < 32) gb[threadIdx.x] = 0;
    const real gscale = (a.geco ? a.state[SVGP_ST_LAGRANGE] * a.inv_bglobal : real(1)) / real(784);
    for (int n = slot; n < a.b; n += a.n_slots) {
        __syncthreads();
        if (mask & 1) {
            for (int i = threadIdx.x; i < 784; i += NT) {
                const real o = a.recon[(size_t)n * 784 + i];
                d3[i] = real(2) * gscale * (o - a.images[(size_t)n * 784 + i]) * elu_grad_from_out(o);
            }
            for (int i = threadIdx.x; i < 1568; i += NT) a2[i] = a.a2[(size_t)n * 1568 + i];
        }
        if (mask & 2) {
            for (int i = threadIdx.x; i < 1568; i += NT) d2[i] = a.d2[(size_t)n * 1568 + i];
            for (int i = threadIdx.x; i < 512; i += NT) a1[i] = a.a1[(size_t)n * 512 + i];
        }
        if (mask & 4) {
            for (int i = threadIdx.x; i < 512; i += NT) d1[i] = a.d1[(size_t)n * 512 + i];
            if (threadIdx.x < 128) { h0[threadIdx.x] = a.h0[(size_t)n * 128 + threadIdx.x]; dh0[threadIdx.x] = a.dh0[(size_t)n * 128 + threadIdx.x]; }
            if ((int)threadIdx.x < a.L) z[threadIdx.x] = a.z[(size_t)n * a.L + threadIdx.x];
        }
        __syncthreads();
        if (mask & 1) UpC3T<NT>::bwd_weight_mfma(a2, d3, gWe3, gb + 16, scratch);     // COUT = 1: chunked VALU form inside
        if (mask & 2) UpC2T<NT>::bwd_weight_valu(a1, d2, gWe2, gb + 8, scratch);
        if (mask & 4) {
            UpC1T<NT>::bwd_weight_mfma(h0, d1, gWe1, gb, scratch);
            for (int o = threadIdx.x; o < ndense; o += NT) dacc[o] += z[o >> 7] * dh0[o & 127];
            if (threadIdx.x < 128) dacc[ndense + threadIdx.x] += dh0[threadIdx.x];
        }
    }
    __syncthreads();
    real* part = a.part + (size_t)slot * od.n;
    if (mask & 1) {
        UpC3T<NT>::fold_grad(gWe3, part + od.c3w);
        if (threadIdx.x == 0) part[od.c3b] = gb[16];
    }
    if (mask & 2) {
        UpC2T<NT>::fold_grad(gWe2, part + od.c2w);
        if (threadIdx.x < 8) part[od.c2b + threadIdx.x] = gb[8 + threadIdx.x];
    }
    if (mask & 4) {
        UpC1T<NT>::fold_grad(gWe1, part + od.c1w);
        if (threadIdx.x < 8) part[od.c1b + threadIdx.x] = gb[threadIdx.x];
        for (int o = threadIdx.x; o < ndense; o += NT) part[od.dw + o] = dacc[o];
        if (threadIdx.x < 128) part[od.db + threadIdx.x] = dacc[ndense + threadIdx.x];
    }
}

// ------------------------------------------------------------------------------------------
// decoder reverse, DATA half: d loss / d recon -> d2, d1, dh0 (stored for the weight half) -> zbar.  The chain the GP
// reverse stages wait for; the weight gradients (decoder_wgrad_rider above) ride in a later launch.  One workgroup of VAE_NT
// threads walks images first, first + stride, ...  PRE: the effective up-convolution weights come from ws.dec_weff (74 KB of
// LDS: two workgroups per CU -- the launch that also carries the deferred (A_hat + jI)^-1, svgp_mnist_decoder_bwd_data_pre_aji,
// relies on that); otherwise they are rebuilt from the raw weights (84 KB).
// Reference semantics: tf.gradients of VAE_utils.py:128-141,154-162 (MNIST_experiment.py:202-205).
// ------------------------------------------------------------------------------------------
struct DecBwdDataArgs {
    int b, L, geco;
    real inv_bglobal;
    const real* state; const real* th_dec; const real* images; const real* a1g; const real* a2g; const real* recon;
    real* d2g; real* d1g; real* dh0g; real* zbar;
    const real* weff;
};
__host__ __device__ constexpr int dec_bwd_data_lds(int L, int n_dec, bool pre) {
    return L * 128 + DEC_NWE + 512 + 1568 + 784 + 1568 + 512 + 128 + (pre ? 0 : n_dec - L * 128 - 128);
}
template <bool PRE>
__device__ __forceinline__ void decoder_bwd_data_images(const DecBwdDataArgs& a, int first, int stride, real* smem) {
    const DecOff od = dec_off(a.L);
    real* w = smem;                  // dense weights only: L*128
    real* We1 = w + a.L * 128;         // effective weights
    real* We2 = We1 + UpC1::NWE;
    real* We3 = We2 + UpC2::NWE;
    real* a1 = We3 + UpC3::NWE;      // 512
    real* a2 = a1 + 512;             // 1568
    real* d3 = a2 + 1568;            // 784
    real* d2 = d3 + 784;             // 1568
    real* d1 = d2 + 1568;            // 512
    real* dh0 = d1 + 512;            // 128
    real* raw = dh0 + 128;           // raw conv weights (+ biases), staged once: od.n - od.c1w
    lds_copy_in(w, a.th_dec + od.dw, a.L * 128);
    if (PRE) {
        lds_copy_in(We1, a.weff, DEC_NWE);
    } else {
        lds_copy_in(raw, a.th_dec + od.c1w, od.n - od.c1w);
        __syncthreads();
        UpC1::build_weff(raw, We1);
        UpC2::build_weff(raw + (od.c2w - od.c1w), We2);
        UpC3::build_weff(raw + (od.c3w - od.c1w), We3);
    }
    const real gscale = (a.geco ? a.state[SVGP_ST_LAGRANGE] * a.inv_bglobal : real(1)) / real(784);
    for (int n = first; n < a.b; n += stride) {
        __syncthreads();
        lds_copy_in(a1, a.a1g + (size_t)n * 512, 512);
        lds_copy_in(a2, a.a2g + (size_t)n * 1568, 1568);
        for (int i = threadIdx.x; i < 784; i += blockDim.x) {
            const real o = a.recon[(size_t)n * 784 + i];
            d3[i] = real(2) * gscale * (o - a.images[(size_t)n * 784 + i]) * elu_grad_from_out(o);
        }
        __syncthreads();
        UpC3::bwd_data_valu(d3, We3, d2);
        __syncthreads();
        for (int i = threadIdx.x; i < 1568; i += blockDim.x) {
            const real v = d2[i] * elu_grad_from_out(a2[i]);
            d2[i] = v;
            a.d2g[(size_t)n * 1568 + i] = v;
        }
        svgp_lds_barrier();             // (the stores to d2g / d1g / dh0g are for a later launch: no wave waits for them here)
        UpC2::bwd_data_mfma(d2, We2, d1);
        svgp_lds_barrier();
        for (int i = threadIdx.x; i < 512; i += blockDim.x) {
            const real v = d1[i] * elu_grad_from_out(a1[i]);
            d1[i] = v;
            a.d1g[(size_t)n * 512 + i] = v;
        }
        svgp_lds_barrier();
        UpC1::bwd_data_mfma(d1, We1, dh0);
        svgp_lds_barrier();
        if (threadIdx.x < 128) a.dh0g[(size_t)n * 128 + threadIdx.x] = dh0[threadIdx.x];
        // zbar[i] = sum_j dh0[j] w[i][j]: 8 lanes per latent channel, xor-shuffle combine
        {
            const int i = threadIdx.x >> 3, part8 = threadIdx.x & 7;
            real acc = 0;
            if (i < a.L)
                for (int j = part8; j < 128; j += 8) acc += dh0[j] * w[i * 128 + j];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (i < a.L && part8 == 0) a.zbar[(size_t)n * a.L + i] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// encoder reverse: (ybar, s2bar) -> encoder weight-gradient partials; one workgroup of NT threads walks images first,
// first + stride, ...; partial slot = first.  LDS: enc_bwd_lds(n_enc) reals (75 KB at L = 16: the launch that carries the
// kernel-matrix VJP as well, svgp_mnist_encoder_bwd_km, holds a VJP workgroup and an image workgroup on one CU).
// Reference semantics: tf.gradients of VAE_utils.py:112-126,143-152 w.r.t. the encoder variables (MNIST_experiment.py:202-205).
// ------------------------------------------------------------------------------------------
struct EncBwdArgs {
    int b, L, clip;
    const real* th_enc; const real* images; const real* a1g; const real* a2g; const real* a3g; const real* var_raw;
    const real* ybar; const real* s2bar;
    real* part;                          // (n_part, eo.n)
};
#define ENC_BWD_SCRATCH 512              // 72 outputs x 7 pixel chunks of the first layer's weight gradient; 256 for the bias sums
__host__ __device__ constexpr int enc_bwd_lds(int n_enc) {
    return 2 * n_enc + 784 + 1352 + 288 + 32 + 1352 + 288 + 32 + 128 + ENC_BWD_SCRATCH;
}

template <int NT>
__device__ __forceinline__ void encoder_bwd_images(const EncBwdArgs& a, int first, int stride, real* smem) {
    const int L = a.L;
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
    real* scratch = dout + 128;      // ENC_BWD_SCRATCH
    using C1 = Conv3<28, 1, 0, 2, 1, 8, 13, NT>;
    using C2 = Conv3<13, 1, 0, 2, 8, 8, 6, NT>;
    using C3 = Conv3<6, 1, 0, 2, 8, 8, 2, NT>;
    static_assert(72 * (NT / 72) <= ENC_BWD_SCRATCH, "first-layer chunk scratch");
    lds_copy_in(w, a.th_enc, eo.n);
    lds_zero(g, eo.n);
    const int twoL = 2 * L;
    for (int n = first; n < a.b; n += stride) {
        __syncthreads();
        lds_copy_in(img, a.images + (size_t)n * 784, 784);
        lds_copy_in(a1, a.a1g + (size_t)n * 1352, 1352);
        lds_copy_in(a2, a.a2g + (size_t)n * 288, 288);
        lds_copy_in(a3, a.a3g + (size_t)n * 32, 32);
        for (int j = threadIdx.x; j < twoL; j += NT) {
            real dj;
            if (j < L) {
                dj = a.ybar[(size_t)n * L + j];
            } else {
                const real vr = a.var_raw[(size_t)n * L + j - L];
                const real sb = a.s2bar[(size_t)n * L + j - L];      // (unconditional: behind the test it was a second, dependent round trip)
                const bool pass = !a.clip || (vr >= 1e-3 && vr <= 10.0);
                dj = pass ? sb * vr : real(0);
            }
            dout[j] = dj;
        }
        __syncthreads();
        // dense: weight / bias gradients and da3
        for (int o = threadIdx.x; o < 32 * twoL; o += NT) g[eo.dw + o] += a3[o / twoL] * dout[o % twoL];
        for (int j = threadIdx.x; j < twoL; j += NT) g[eo.db + j] += dout[j];
        if (threadIdx.x < 32) {
            real acc = 0;
            for (int j = 0; j < twoL; ++j) acc += dout[j] * w[eo.dw + threadIdx.x * twoL + j];
            d3[threadIdx.x] = acc * elu_grad_from_out(a3[threadIdx.x]);
        }
        __syncthreads();
        C3::bwd_weight(a2, d3, g + eo.c3w, g + eo.c3b, scratch);
        C3::bwd_data(d3, w + eo.c3w, d2);
        __syncthreads();
        for (int i = threadIdx.x; i < 288; i += NT) d2[i] *= elu_grad_from_out(a2[i]);
        __syncthreads();
        C2::bwd_weight(a1, d2, g + eo.c2w, g + eo.c2b, scratch);
        C2::bwd_data(d2, w + eo.c2w, d1);
        __syncthreads();
        for (int i = threadIdx.x; i < 1352; i += NT) d1[i] *= elu_grad_from_out(a1[i]);
        __syncthreads();
        C1::bwd_weight(img, d1, g + eo.c1w, g + eo.c1b, scratch);
    }
    __syncthreads();
    lds_copy_out(a.part + (size_t)first * eo.n, g, eo.n);
}

}  // namespace svgp_vae

// vae_mnist.hip: the weight half's arguments from a configuration + workspace (riders of the reverse factor launch, gp_kernels.hip)
svgp_vae::DecWgradArgs svgp_make_dec_wgrad_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* images,
                                                double* ws, const double* state, int n_types);
svgp_vae::DecBwdDataArgs svgp_make_dec_bwd_data_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* theta,
                                                     const double* images, double* ws, const double* state);
svgp_vae::EncBwdArgs svgp_make_enc_bwd_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* theta,
                                            const double* images, double* ws);
