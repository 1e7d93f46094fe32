// Generic NHWC float64 convolution as a "tap-table" gather-GEMM on the f64 MFMA (gfx950):
//
//   out[n][y*osy+ooy][x*osx+oox][co] = act( bias[co] + sum_t sum_ci in[n][y*sy+oy_t][x*sx+ox_t][ci] * W_t[ci][co] )
//
// for (y,x) in an Hs x Ws iteration space; reads outside the input are zero.  One descriptor = one
// "class"; up to 4 classes per launch (blockIdx.z = image * ncls + class).  With suitable tap tables this
// one kernel is: Conv2D 3x3 / 2x2, stride 1 / 2, 'same' / 'valid' (Keras padding) forward; UpSampling2D(2)+
// Conv2D as four parity classes with pre-summed effective weights (2.25x fewer MACs); and every data
// gradient (transposed weights; parity classes for stride 2).  `svgp_conv_taps_wgrad` is the matching weight
// gradient: dW_t[ci][co] = sum_{n,y,x} in[...][ci] * dout[...][co].
// Reference layers: spritesVAE / sprites_representation_network (VAE_utils.py:275-391).
//
// Mapping: a wave owns one 16-pixel row segment; A[i=pixel][k=ci] from an LDS halo tile (pixel stride
// Ci4+2 doubles -> conflict-free 16-pixel fetch), B[k=ci][j=co] from LDS weights, one
// v_mfma_f64_16x16x4 per (tap, 4 input channels).  Workgroup = 4 waves = 16 x 8 output pixels.
#include "common.hpp"


#define CT_TW 16      // tile width  (pixels per MFMA row segment)
#define CT_TH 8       // tile height (2 rows per wave)
#define CT_MAXT 16

struct ConvLaunch {
    int ncls;
    svgp_conv_desc d[4];
};

namespace {

__device__ __forceinline__ void tap_range(const svgp_conv_desc& d, int& omin_y, int& omax_y, int& omin_x, int& omax_x) {
    omin_y = omax_y = d.oy[0]; omin_x = omax_x = d.ox[0];
    for (int t = 1; t < d.nt; ++t) {
        omin_y = min(omin_y, d.oy[t]); omax_y = max(omax_y, d.oy[t]);
        omin_x = min(omin_x, d.ox[t]); omax_x = max(omax_x, d.ox[t]);
    }
}


// Staging helpers.  The loads of a chunk are all issued before the first LDS store (U independent loads in flight per
// thread): a workgroup has only 4 waves, so a load-then-store loop would expose one full memory latency per element.
// Thread = (channel pair, pixel lane); channels of a pixel are contiguous in NHWC, pixels of a tile row as well.
#define CT_U 8
// tile[(py * hw + px) * ps + c] = in[hy0 + py][hx0 + px][c]   (zero outside the image / for padded channels)
template <typename T>
__device__ __forceinline__ void stage_halo(T* __restrict__ tile, const T* __restrict__ inn, int Hi, int Wi, int Ci,
                                           int hy0, int hx0, int hh, int hw, int Ci4, int ps) {
    typedef typename SvgpMfma<T>::pair_t pair_t;
    const int cp = Ci4 >> 1, npl = (int)blockDim.x / cp;
    const int c2 = ((int)threadIdx.x % cp) * 2, pl = (int)threadIdx.x / cp, npix = hh * hw;
    if (pl >= npl) return;
    const bool pair = (Ci & 1) == 0;              // even channel count: 16-byte aligned pairs
    for (int p0 = pl; p0 < npix; p0 += npl * CT_U) {
        T v0[CT_U], v1[CT_U];
#pragma unroll
        for (int u = 0; u < CT_U; ++u) {
            const int p = p0 + u * npl, py = p / hw, px = p - py * hw, gy = hy0 + py, gx = hx0 + px;
            v0[u] = 0; v1[u] = 0;
            if (p < npix && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi) {
                const T* src = inn + ((size_t)gy * Wi + gx) * Ci + c2;
                if (pair && c2 + 1 < Ci) {
                    const pair_t t = *reinterpret_cast<const pair_t*>(src);
                    v0[u] = t.x; v1[u] = t.y;
                } else {
                    if (c2 < Ci) v0[u] = src[0];
                    if (c2 + 1 < Ci) v1[u] = src[1];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CT_U; ++u) {
            const int p = p0 + u * npl;
            if (p < npix) { tile[p * ps + c2] = v0[u]; tile[p * ps + c2 + 1] = v1[u]; }
        }
    }
}

// union of the tap ranges of all classes of a launch (they share the staged input tile)
__device__ __forceinline__ void tap_range_all(const ConvLaunch& L, int& oy0, int& oy1, int& ox0, int& ox1) {
    tap_range(L.d[0], oy0, oy1, ox0, ox1);
    for (int c = 1; c < L.ncls; ++c) {
        int a0, a1, b0, b1;
        tap_range(L.d[c], a0, a1, b0, b1);
        oy0 = min(oy0, a0); oy1 = max(oy1, a1); ox0 = min(ox0, b0); ox1 = max(ox1, b1);
    }
}

// CI4 / NT: compile-time padded channel count and tap count (0: run time).  With both known the 9 x 4 (or 4 x 4) MFMA
// groups of a 16-pixel x 16-channel output tile are straight-line code: the run-time loops paid a taken branch (~32 cycles)
// per 4-channel step and a scalar load of the tap offsets per tap, next to MFMAs of 32 (f32) / 64 (f64) cycles:
// 64 x 64, 16 -> 16, 500 frames: 198 -> 168 us in float32 (56 TFLOP/s), 310 -> 280 us in float64.  Where the rest goes
// (temporary switches, float32): without the halo staging 134 us, without the stores 144, with neither 105 (90 TFLOP/s =
// 57 % of the f32 MFMA peak) -- the three phases of a workgroup add up although 7 workgroups per CU are resident.
template <typename T, int CI4, int NT>
__global__ __launch_bounds__(256) void k_conv_taps_fwd(ConvLaunch L, int nchunk, const T* __restrict__ in,
                                                       const T* __restrict__ w, const T* __restrict__ bias,
                                                       T* __restrict__ out) {
    typedef SvgpMfma<T> MF;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    // The workgroup keeps ONE tile position and walks the images blockIdx.z, blockIdx.z + nchunk, ...: the tap weights of
    // every class are staged once, and the classes of a launch (the four output parities of an upsample-fused or
    // stride-2-transposed layer) share one staged input tile per image.
    const svgp_conv_desc& d = L.d[0];                      // geometry shared by the classes: n, Hs, Ws, sy, sx, input
    const int tiles_x = (d.Ws + CT_TW - 1) / CT_TW;
    const int x0 = (blockIdx.x % tiles_x) * CT_TW, y0 = (blockIdx.x / tiles_x) * CT_TH;
    if (y0 >= d.Hs) return;
    const int Ci4 = CI4 ? CI4 : (d.Ci + 3) & ~3, ps = Ci4 + 2;         // channels padded to 4, pixel stride
    int oy0, oy1, ox0, ox1;
    tap_range_all(L, oy0, oy1, ox0, ox1);
    const int hy0 = y0 * d.sy + oy0, hx0 = x0 * d.sx + ox0;
    const int hh = (CT_TH - 1) * d.sy + (oy1 - oy0) + 1, hw = (CT_TW - 1) * d.sx + (ox1 - ox0) + 1;
    T* tile = smem;                                        // hh x hw x ps
    T* wl = tile + hh * hw * ps;                           // per class: nt x Ci4 x 16 (co padded to 16), packed
    int wbase[4] = {0, 0, 0, 0};
    for (int cls = 0, o = 0; cls < L.ncls; ++cls) {
        const svgp_conv_desc& dc = L.d[cls];
        wbase[cls] = o;
        for (int t = threadIdx.x; t < dc.nt * Ci4 * 16; t += blockDim.x) {
            const int co = t & 15, c = (t >> 4) % Ci4, tp = t / (16 * Ci4);
            wl[o + t] = (c < dc.Ci && co < dc.Co) ? w[dc.woff[tp] + c * dc.Co + co] : T(0);
        }
        o += dc.nt * Ci4 * 16;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const T bias_r = (d.act && r < d.Co) ? bias[r] : T(0);
    for (int n = blockIdx.z; n < d.n; n += nchunk) {
        __syncthreads();                                   // previous image's MFMA reads of `tile` are done
        stage_halo(tile, in + (size_t)n * d.Hi * d.Wi * d.Ci, d.Hi, d.Wi, d.Ci, hy0, hx0, hh, hw, Ci4, ps);
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < 2 * L.ncls; ++it) {
            const int rr = it & 1, cls = it >> 1;
            const svgp_conv_desc& dc = L.d[cls];
            const int ly = wave * 2 + rr, y = y0 + ly;
            if (y >= dc.Hs) continue;
            const int wb = cls == 0 ? wbase[0] : cls == 1 ? wbase[1] : cls == 2 ? wbase[2] : wbase[3];
            typename MF::acc_t acc = {0, 0, 0, 0};
            if (NT > 0 && CI4 > 0) {
                // two accumulators: consecutive MFMAs do not wait on each other
                typename MF::acc_t acc2 = {0, 0, 0, 0};
#pragma unroll
                for (int tp = 0; tp < NT; ++tp) {
                    const T* ap = tile + ((ly * d.sy + dc.oy[tp] - oy0) * hw + (r * d.sx + dc.ox[tp] - ox0)) * ps + q;
                    const T* bp = wl + wb + (tp * CI4 + q) * 16 + r;
#pragma unroll
                    for (int c0 = 0; c0 < CI4; c0 += 4) {
                        if (((tp * (CI4 / 4) + c0 / 4) & 1) == 0) acc = MF::mma(ap[c0], bp[c0 * 16], acc);
                        else acc2 = MF::mma(ap[c0], bp[c0 * 16], acc2);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] += acc2[g];
            } else {
                for (int tp = 0; tp < dc.nt; ++tp) {
                    const T* ap = tile + ((ly * d.sy + dc.oy[tp] - oy0) * hw + (r * d.sx + dc.ox[tp] - ox0)) * ps + q;
                    const T* bp = wl + wb + (tp * Ci4 + q) * 16 + r;
                    for (int c0 = 0; c0 < Ci4; c0 += 4) acc = MF::mma(ap[c0], bp[c0 * 16], acc);
                }
            }
            // D: column (co) = lane & 15, row (pixel) = MF::row(q, g)
            const int gy = y * dc.osy + dc.ooy;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int x = x0 + MF::row(q, g);
                if (x < dc.Ws && r < dc.Co) {
                    T v = acc[g];
                    if (dc.act) v += bias_r;
                    if (dc.act == 1) v = v > 0 ? v : (T)(exp(v) - T(1));
                    out[(((size_t)n * dc.Ho + gy) * dc.Wo + (x * dc.osx + dc.oox)) * dc.Co + r] = v;
                }
            }
        }
    }
}

// Weight gradient.  grid (nwg, ncls): workgroup g of a class walks tiles g, g+nwg, ... of ALL images and
// keeps dW_t (Ci4 x 16 per tap) in MFMA accumulators: A[i=ci][k=pixel] = in, B[k=pixel][j=co] = dout,
// k-steps of 4 consecutive pixels of a row segment.  Partials: part[g][woff_t + ci * Co + co]; the classes' tap ranges are
// disjoint, so they share row g.
template <typename T>
__global__ __launch_bounds__(256) void k_conv_taps_wgrad(ConvLaunch L, int nwg, const T* __restrict__ in,
                                                         const T* __restrict__ dout, T* __restrict__ part,
                                                         int part_stride) {
    typedef SvgpMfma<T> MF;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int cls = blockIdx.y;
    const svgp_conv_desc& d = L.d[cls];
    const int tiles_x = (d.Ws + CT_TW - 1) / CT_TW, tiles_y = (d.Hs + CT_TH - 1) / CT_TH;
    const int ntile = tiles_x * tiles_y * d.n;
    const int Ci4 = (d.Ci + 3) & ~3, ps = Ci4 + 2;
    int oy0, oy1, ox0, ox1;
    tap_range(d, oy0, oy1, ox0, ox1);
    const int hh = (CT_TH - 1) * d.sy + (oy1 - oy0) + 1, hw = (CT_TW - 1) * d.sx + (ox1 - ox0) + 1;
    T* tile = smem;                             // hh x hw x ps      (input halo)
    T* dt = tile + hh * hw * ps;                // CT_TH x CT_TW x 18 (dout tile, co padded to 16)
    T* red = dt + CT_TH * CT_TW * 18;           // 4 waves x 64 lanes x 4  (cross-wave combine)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    // accumulators: per tap one 16x16 tile (rows ci, cols co); Ci4 <= 16.  Every wave takes ALL taps of 2 of the 8 tile rows
    // (balanced for any tap count; the dout fragment of a row is loaded once and reused by every tap); the four waves'
    // partial tiles are added through LDS once per workgroup, after the last tile.
    typename MF::acc_t acc[CT_MAXT];
#pragma unroll
    for (int a = 0; a < CT_MAXT; ++a) acc[a] = typename MF::acc_t{0, 0, 0, 0};
    for (int tl = blockIdx.x; tl < ntile; tl += nwg) {
        const int n = tl / (tiles_x * tiles_y), tt = tl % (tiles_x * tiles_y);
        const int x0 = (tt % tiles_x) * CT_TW, y0 = (tt / tiles_x) * CT_TH;
        const int hy0 = y0 * d.sy + oy0, hx0 = x0 * d.sx + ox0;
        const T* inn = in + (size_t)n * d.Hi * d.Wi * d.Ci;
        __syncthreads();
        stage_halo(tile, inn, d.Hi, d.Wi, d.Ci, hy0, hx0, hh, hw, Ci4, ps);
        {   // dout tile: thread = (co, pixel lane of 16), 8 pixels in flight per thread
            const int co = threadIdx.x & 15, pl = threadIdx.x >> 4;
            T v[CT_TH];
#pragma unroll
            for (int u = 0; u < CT_TH; ++u) {
                const int p = pl + 16 * u, px = p % CT_TW, py = p / CT_TW, y = y0 + py, x = x0 + px;
                v[u] = 0;
                if (co < d.Co && y < d.Hs && x < d.Ws)
                    v[u] = dout[(((size_t)n * d.Ho + (y * d.osy + d.ooy)) * d.Wo + (x * d.osx + d.oox)) * d.Co + co];
            }
#pragma unroll
            for (int u = 0; u < CT_TH; ++u) dt[(pl + 16 * u) * 18 + co] = v[u];
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int py = wave * 2 + rr;
            T bv[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) bv[k4] = dt[(py * CT_TW + 4 * k4 + q) * 18 + r];          // co = r
#pragma unroll
            for (int tp = 0; tp < CT_MAXT; ++tp) {
                if (tp < d.nt) {
                    const T* ap = tile + ((py * d.sy + d.oy[tp] - oy0) * hw + (d.ox[tp] - ox0)) * ps + r;  // ci = r
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const T av = (r < Ci4) ? ap[((4 * k4 + q) * d.sx) * ps] : T(0);
                        acc[tp] = MF::mma(av, bv[k4], acc[tp]);
                    }
                }
            }
        }
    }
    // cross-wave combine (fixed order) and store.  D: col (co) = r, row (ci) = MF::row(q, g)
    // the classes of a launch write disjoint weight ranges (their own taps), so they share partial row blockIdx.x
    T* po = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
    for (int tp = 0; tp < CT_MAXT; ++tp) {
        if (tp < d.nt) {
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g) red[(wave * 64 + lane) * 4 + g] = acc[tp][g];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const T t = red[lane * 4 + g] + red[(64 + lane) * 4 + g] + red[(128 + lane) * 4 + g] +
                                red[(192 + lane) * 4 + g];
                    const int ci = MF::row(q, g);
                    if (ci < d.Ci && r < d.Co) po[d.woff[tp] + ci * d.Co + r] = t;
                }
            }
        }
    }
}

// UpSampling2D(2) + 3x3 conv as four parity classes: effective weights we[py][px][ty][tx] = sum of the raw taps (ky,kx) that
// land on low-resolution offset (ty,tx) for output parity (py,px): tap group T(p,k) = (k + p >= 2).  fold = the transpose
// of that sum (gradient of the raw weights from the gradient of the effective ones).
template <typename T>
__global__ void k_upconv_weff(int cc, const T* __restrict__ w, T* __restrict__ we) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (py,px,ty,tx, ci*co)
    if (i >= 16 * cc) return;
    const int e = i % cc, g = i / cc, tx = g & 1, ty = (g >> 1) & 1, px = (g >> 2) & 1, py = g >> 3;
    T s = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            if (((ky + py >= 2) ? 1 : 0) == ty && ((kx + px >= 2) ? 1 : 0) == tx) s += w[(ky * 3 + kx) * cc + e];
    we[i] = s;
}
template <typename T>
__global__ void k_upconv_fold(int cc, const T* __restrict__ ge, T* __restrict__ g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (ky,kx, ci*co)
    if (i >= 9 * cc) return;
    const int e = i % cc, k = i / cc, kx = k % 3, ky = k / 3;
    T s = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int ty = (ky + py >= 2) ? 1 : 0, tx = (kx + px >= 2) ? 1 : 0;
            s += ge[((((py * 2 + px) * 2 + ty) * 2 + tx)) * cc + e];
        }
    g[i] = s;
}

// out[i] (+)= sum_g part[g][i]   (fixed order)
// one workgroup per 16 outputs: thread = (output il = tid & 15, partial lane ch = tid >> 4); lane ch adds partials ch, ch + 16,
// ... (8 loads in flight), the 16 lanes are combined through LDS in fixed order
template <typename T>
__global__ __launch_bounds__(256) void k_sum_partials(int ng, int len, int stride, const T* __restrict__ part,
                                                      T* __restrict__ out, int accumulate) {
    __shared__ T sh[16][17];
    const int il = threadIdx.x & 15, ch = threadIdx.x >> 4, i = blockIdx.x * 16 + il;
    T s = 0;
    if (i < len) {
        for (int g0 = ch; g0 < ng; g0 += 16 * 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int g = g0 + 16 * u; v[u] = g < ng ? part[(size_t)g * stride + i] : T(0); }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    sh[ch][il] = s;
    __syncthreads();
    if (ch == 0 && i < len) {
        T t = accumulate ? out[i] : T(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) t += sh[c][il];
        out[i] = t;
    }
}

// dpre = dout * elu'(out) in place on dout; also per-block column sums for the bias gradient.  HBM-bound (read out, read
// dout, write dout): 8 independent element pairs in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void k_elu_bwd_colsum(long long npix, int C, const T* __restrict__ outv,
                                                        T* __restrict__ dout, T* __restrict__ part) {
    __shared__ T sh[256];
    // thread t handles channel t % C of pixels t / C + k * (256 / C)   (C <= 16 divides into 256 evenly enough)
    const int c = threadIdx.x % C, lp = threadIdx.x / C, ppb = blockDim.x / C;
    T s = 0;
    if (lp < ppb) {
        const long long stride = (long long)gridDim.x * ppb;
        for (long long p0 = (long long)blockIdx.x * ppb + lp; p0 < npix; p0 += stride * 8) {
            T dv[8], ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long p = p0 + u * stride;
                dv[u] = 0; ov[u] = 1;
                if (p < npix) {
                    const size_t o = (size_t)p * C + c;
                    dv[u] = dout[o];
                    if (outv) ov[u] = outv[o];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long p = p0 + u * stride;
                if (p < npix) {
                    if (outv) { dv[u] *= (ov[u] > 0 ? T(1) : ov[u] + T(1)); dout[(size_t)p * C + c] = dv[u]; }
                    s += dv[u];
                }
            }
        }
    }
    sh[threadIdx.x] = (lp < ppb) ? s : T(0);
    __syncthreads();
    if (threadIdx.x < C) {
        T t = 0;
        for (int k = 0; k < ppb; ++k) t += sh[k * C + threadIdx.x];
        part[blockIdx.x * C + threadIdx.x] = t;
    }
}

size_t fwd_lds(const svgp_conv_desc& d) {
    int oy0 = d.oy[0], oy1 = d.oy[0], ox0 = d.ox[0], ox1 = d.ox[0];
    for (int t = 1; t < d.nt; ++t) {
        oy0 = oy0 < d.oy[t] ? oy0 : d.oy[t]; oy1 = oy1 > d.oy[t] ? oy1 : d.oy[t];
        ox0 = ox0 < d.ox[t] ? ox0 : d.ox[t]; ox1 = ox1 > d.ox[t] ? ox1 : d.ox[t];
    }
    const int Ci4 = (d.Ci + 3) & ~3, ps = Ci4 + 2;
    const int hh = (CT_TH - 1) * d.sy + (oy1 - oy0) + 1, hw = (CT_TW - 1) * d.sx + (ox1 - ox0) + 1;
    return (size_t)hh * hw * ps;
}

size_t fwd_lds_all(const svgp_conv_desc* d, int ncls) {      // halo tile over the union of the classes' tap ranges
    int oy0 = d[0].oy[0], oy1 = oy0, ox0 = d[0].ox[0], ox1 = ox0;
    for (int c = 0; c < ncls; ++c)
        for (int t = 0; t < d[c].nt; ++t) {
            oy0 = oy0 < d[c].oy[t] ? oy0 : d[c].oy[t]; oy1 = oy1 > d[c].oy[t] ? oy1 : d[c].oy[t];
            ox0 = ox0 < d[c].ox[t] ? ox0 : d[c].ox[t]; ox1 = ox1 > d[c].ox[t] ? ox1 : d[c].ox[t];
        }
    const int Ci4 = (d[0].Ci + 3) & ~3, ps = Ci4 + 2;
    const int hh = (CT_TH - 1) * d[0].sy + (oy1 - oy0) + 1, hw = (CT_TW - 1) * d[0].sx + (ox1 - ox0) + 1;
    return (size_t)hh * hw * ps;
}

int check_desc(const svgp_conv_desc* d, int ncls) {
    SVGP_REQUIRE(d && ncls >= 1 && ncls <= 4, SVGP_ERR_INVALID, "need 1..4 conv classes");
    for (int c = 0; c < ncls; ++c) {
        SVGP_REQUIRE(d[c].nt >= 1 && d[c].nt <= CT_MAXT, SVGP_ERR_INVALID, "taps must be 1..16");
        SVGP_REQUIRE(d[c].Ci >= 1 && d[c].Ci <= 16 && d[c].Co >= 1 && d[c].Co <= 16, SVGP_ERR_UNSUPPORTED,
                     "conv_taps supports 1..16 input and output channels (got %d, %d)", d[c].Ci, d[c].Co);
        SVGP_REQUIRE(d[c].n >= 1 && d[c].Hs >= 1 && d[c].Ws >= 1 && d[c].sy >= 1 && d[c].sx >= 1, SVGP_ERR_INVALID,
                     "bad conv descriptor");
        SVGP_REQUIRE(d[c].n == d[0].n && d[c].Hs == d[0].Hs && d[c].Ws == d[0].Ws, SVGP_ERR_INVALID,
                     "classes of one launch must share n and the iteration space");
    }
    return SVGP_OK;
}

}  // namespace

template <typename T>
static int conv_taps_fwd_impl(const svgp_conv_desc* d, int ncls, const T* in, const T* w, const T* bias, T* out,
                              void* stream) {
    int rc = check_desc(d, ncls);
    if (rc) return rc;
    SVGP_REQUIRE(in && w && out, SVGP_ERR_INVALID, "NULL device pointer");
    ConvLaunch L;
    L.ncls = ncls;
    for (int c = 0; c < ncls; ++c) {
        L.d[c] = d[c];
        SVGP_REQUIRE(!d[c].act || bias, SVGP_ERR_INVALID, "bias is NULL but act != 0");
        SVGP_REQUIRE(d[c].n == d[0].n && d[c].Hs == d[0].Hs && d[c].Ws == d[0].Ws && d[c].sy == d[0].sy &&
                         d[c].sx == d[0].sx && d[c].Hi == d[0].Hi && d[c].Wi == d[0].Wi && d[c].Ci == d[0].Ci &&
                         d[c].Co == d[0].Co && d[c].act == d[0].act,
                     SVGP_ERR_INVALID, "the classes of one launch share the input geometry (n, Hs, Ws, strides, Ci, Co, act)");
    }
    // union halo tile of all classes + the packed tap weights of every class
    size_t lds = fwd_lds_all(d, ncls);
    for (int c = 0; c < ncls; ++c) lds += (size_t)d[c].nt * ((d[0].Ci + 3) & ~3) * 16;
    lds *= sizeof(T);
    SVGP_REQUIRE(lds <= 160 * 1024, SVGP_ERR_UNSUPPORTED, "conv tile needs %zu bytes of LDS", lds);
    const int tiles = ((d[0].Ws + CT_TW - 1) / CT_TW) * ((d[0].Hs + CT_TH - 1) / CT_TH);
    // enough workgroups for ~8 per CU, each walking n / nchunk images of its tile position
    int nchunk = (2048 + tiles - 1) / tiles;
    if (nchunk > d[0].n) nchunk = d[0].n;
    if (nchunk < 1) nchunk = 1;
    // straight-line instances for the layer shapes of the SPRITES networks: 16 (or 4 = padded 3) input channels, 9 taps
    // (3 x 3) or 4 taps (the parity classes of the upsample-fused / transposed stride-2 layers), every class alike
    const int Ci4 = (d[0].Ci + 3) & ~3;
    int nt_all = d[0].nt;
    for (int c = 1; c < ncls; ++c) if (d[c].nt != nt_all) nt_all = 0;
#define CONV_FWD_LAUNCH(CI4_, NT_)                                                                                          \
    do {                                                                                                                    \
        SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_taps_fwd<T, CI4_, NT_>),                     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        hipLaunchKernelGGL((k_conv_taps_fwd<T, CI4_, NT_>), dim3(tiles, 1, nchunk), dim3(256), lds, (hipStream_t)stream, L,   \
                           nchunk, in, w, bias, out);                                                                       \
    } while (0)
    if (Ci4 == 16 && nt_all == 9) CONV_FWD_LAUNCH(16, 9);
    else if (Ci4 == 16 && nt_all == 4) CONV_FWD_LAUNCH(16, 4);
    else if (Ci4 == 4 && nt_all == 9) CONV_FWD_LAUNCH(4, 9);
    else CONV_FWD_LAUNCH(0, 0);
#undef CONV_FWD_LAUNCH
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// part: (ncls * nwg, part_stride) scratch; dw (part_stride values, the layer's weight layout via woff) receives
// the fixed-order sum over workgroups and classes (accumulate != 0 adds to dw).
template <typename T>
static int conv_taps_wgrad_impl(const svgp_conv_desc* d, int ncls, const T* in, const T* dout, T* part, int nwg,
                                int part_stride, T* dw, int accumulate, void* stream) {
    int rc = check_desc(d, ncls);
    if (rc) return rc;
    SVGP_REQUIRE(in && dout && part && dw && nwg >= 1 && part_stride >= 1, SVGP_ERR_INVALID, "bad argument");
    ConvLaunch L;
    L.ncls = ncls;
    size_t lds = 0;
    for (int c = 0; c < ncls; ++c) {
        L.d[c] = d[c];
        const size_t e = fwd_lds(d[c]) + (size_t)CT_TH * CT_TW * 18 + 1024;
        lds = e > lds ? e : lds;
    }
    lds *= sizeof(T);
    SVGP_REQUIRE(lds <= 160 * 1024, SVGP_ERR_UNSUPPORTED, "conv tile needs %zu bytes of LDS", lds);
    SVGP_CHECK_HIP(hipMemsetAsync(part, 0, (size_t)nwg * part_stride * sizeof(T), (hipStream_t)stream));
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_taps_wgrad<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_conv_taps_wgrad<T>, dim3(nwg, ncls), dim3(256), lds, (hipStream_t)stream, L, nwg, in, dout, part,
                       part_stride);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials<T>, dim3((part_stride + 15) / 16), dim3(256), 0, (hipStream_t)stream, nwg,
                       part_stride, part_stride, (const T*)part, dw, accumulate);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// dpre = dout * elu'(out) (in place on dout; out == NULL skips the activation) and db[c] = sum dpre[.., c].
// part: (1024, C) scratch.
template <typename T>
static int elu_bwd_bias_impl(long long npix, int C, const T* out, T* dout, T* part, T* db, void* stream) {
    SVGP_REQUIRE(npix >= 1 && C >= 1 && C <= 16 && dout && part && db, SVGP_ERR_INVALID, "bad argument");
    const int nblk = 1024;     // 4 workgroups per CU keep the HBM queues full
    hipLaunchKernelGGL(k_elu_bwd_colsum<T>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, npix, C, out, dout, part);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials<T>, dim3(1), dim3(256), 0, (hipStream_t)stream, nblk, C, C, (const T*)part, db, 0);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// effective weights of an upsample-fused 3x3 convolution (conv.py ConvLayer(up=True)): w (3,3,Ci,Co) -> we (2,2,2,2,Ci,Co);
// and the gradient of w from the gradient of we
template <typename T>
static int upconv_weights_impl(int Ci, int Co, const T* w, T* we, void* stream) {
    SVGP_REQUIRE(Ci >= 1 && Co >= 1 && w && we, SVGP_ERR_INVALID, "bad argument");
    const int cc = Ci * Co;
    hipLaunchKernelGGL(k_upconv_weff<T>, dim3((16 * cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, cc, w, we);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
template <typename T>
static int upconv_fold_impl(int Ci, int Co, const T* ge, T* g, void* stream) {
    SVGP_REQUIRE(Ci >= 1 && Co >= 1 && ge && g, SVGP_ERR_INVALID, "bad argument");
    const int cc = Ci * Co;
    hipLaunchKernelGGL(k_upconv_fold<T>, dim3((9 * cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, cc, ge, g);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_conv_taps_fwd(const svgp_conv_desc* d, int ncls, const double* in, const double* w,
                                  const double* bias, double* out, void* stream) {
    return conv_taps_fwd_impl<double>(d, ncls, in, w, bias, out, stream);
}
extern "C" int svgp_conv_taps_wgrad(const svgp_conv_desc* d, int ncls, const double* in, const double* dout,
                                    double* part, int nwg, int part_stride, double* dw, int accumulate, void* stream) {
    return conv_taps_wgrad_impl<double>(d, ncls, in, dout, part, nwg, part_stride, dw, accumulate, stream);
}
extern "C" int svgp_elu_bwd_bias(long long npix, int C, const double* out, double* dout, double* part, double* db,
                                 void* stream) {
    return elu_bwd_bias_impl<double>(npix, C, out, dout, part, db, stream);
}
extern "C" int svgp_upconv_weights(int Ci, int Co, const double* w, double* we, void* stream) {
    return upconv_weights_impl<double>(Ci, Co, w, we, stream);
}
extern "C" int svgp_upconv_fold_wgrad(int Ci, int Co, const double* ge, double* g, void* stream) {
    return upconv_fold_impl<double>(Ci, Co, ge, g, stream);
}
// ---- float32 instantiations: the reference's dtype for the SPRITES networks (VAE_utils.py:277); same tap tables, the
// gather-GEMM runs on v_mfma_f32_16x16x4_f32 (twice the matrix rate of the f64 form, half the LDS and HBM bytes)
extern "C" int svgp_conv_taps_fwd_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* w, const float* bias,
                                      float* out, void* stream) {
    return conv_taps_fwd_impl<float>(d, ncls, in, w, bias, out, stream);
}
extern "C" int svgp_conv_taps_wgrad_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* dout, float* part,
                                        int nwg, int part_stride, float* dw, int accumulate, void* stream) {
    return conv_taps_wgrad_impl<float>(d, ncls, in, dout, part, nwg, part_stride, dw, accumulate, stream);
}
extern "C" int svgp_elu_bwd_bias_f32(long long npix, int C, const float* out, float* dout, float* part, float* db,
                                     void* stream) {
    return elu_bwd_bias_impl<float>(npix, C, out, dout, part, db, stream);
}
extern "C" int svgp_upconv_weights_f32(int Ci, int Co, const float* w, float* we, void* stream) {
    return upconv_weights_impl<float>(Ci, Co, w, we, stream);
}
extern "C" int svgp_upconv_fold_wgrad_f32(int Ci, int Co, const float* ge, float* g, void* stream) {
    return upconv_fold_impl<float>(Ci, Co, ge, g, stream);
}
