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

// slots nobody reads (4 x 64): where the lanes of an unpredicated store that have nothing to store write (external linkage: stores
// to an internal variable that is never read are deleted)
__device__ __attribute__((aligned(128))) double svgp_conv_trash[256];

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

// ---- deferred partial sums.  Every layer of a network's reverse pass ends in two of these small reductions (weights, bias): 32
// launches of ~10 us per SPRITES step, each in the middle of a chain of dependent launches.  Their results are only read by the
// optimiser, so the *_jobs entry points run the convolution kernels as usual but hand the reductions back to the caller as
// svgp_sum_job descriptors, and svgp_sum_partials_multi runs any number of them as ONE launch.  The caller keeps every partial
// buffer intact until then (one scratch region per layer) and orders the stream of the multi-launch behind every stream that
// produced partials.  No state outlives a call: the capture target below is set for the duration of one *_jobs call only.
#define SUM_MAX_JOBS 64
template <typename T>
struct SumJobs {
    const T* part[SUM_MAX_JOBS]; T* out[SUM_MAX_JOBS];
    int ng[SUM_MAX_JOBS], len[SUM_MAX_JOBS], stride[SUM_MAX_JOBS], acc[SUM_MAX_JOBS], blk0[SUM_MAX_JOBS + 1];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void k_sum_partials_multi(SumJobs<T> J) {
    __shared__ T sh[16][17];
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.blk0[j + 1]) ++j;
    const int ng = J.ng[j], len = J.len[j], stride = J.stride[j];
    const T* __restrict__ part = J.part[j];
    const int il = threadIdx.x & 15, ch = threadIdx.x >> 4, i = ((int)blockIdx.x - J.blk0[j]) * 16 + il;
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
        T t = J.acc[j] ? J.out[j][i] : T(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) t += sh[c][il];
        J.out[j][i] = t;
    }
}
struct SumCapture { svgp_sum_job* jobs = nullptr; int cap = 0, n = 0; };
static SumCapture& sum_capture() { static thread_local SumCapture c; return c; }
// the one way k_sum_partials is launched: same order of additions whether it runs now or in svgp_sum_partials_multi
template <typename T>
static int sum_partials(int ng, int len, int stride, const T* part, T* out, int accumulate, void* stream) {
    SumCapture& c = sum_capture();
    if (c.jobs) {
        SVGP_REQUIRE(c.n < c.cap, SVGP_ERR_INVALID, "more partial-sum jobs than the caller's array holds (%d)", c.cap);
        svgp_sum_job& j = c.jobs[c.n++];
        j.part = part; j.out = out; j.ng = ng; j.len = len; j.stride = stride; j.accumulate = accumulate;
        return SVGP_OK;
    }
    hipLaunchKernelGGL(k_sum_partials<T>, dim3((len + 15) / 16), dim3(256), 0, (hipStream_t)stream, ng, len, stride, part, out,
                       accumulate);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
template <typename T>
static int sum_partials_multi_impl(const svgp_sum_job* jobs, int n, void* stream) {
    SVGP_REQUIRE(n >= 0 && (n == 0 || jobs), SVGP_ERR_INVALID, "bad job list");
    for (int lo = 0; lo < n; lo += SUM_MAX_JOBS) {
        SumJobs<T> J;
        J.n = n - lo < SUM_MAX_JOBS ? n - lo : SUM_MAX_JOBS;
        J.blk0[0] = 0;
        for (int k = 0; k < J.n; ++k) {
            const svgp_sum_job& j = jobs[lo + k];
            SVGP_REQUIRE(j.part && j.out && j.ng >= 1 && j.len >= 1 && j.stride >= j.len, SVGP_ERR_INVALID, "bad partial-sum job %d", lo + k);
            J.part[k] = static_cast<const T*>(j.part); J.out[k] = static_cast<T*>(j.out);
            J.ng[k] = j.ng; J.len[k] = j.len; J.stride[k] = j.stride; J.acc[k] = j.accumulate;
            J.blk0[k + 1] = J.blk0[k] + (j.len + 15) / 16;
        }
        hipLaunchKernelGGL(k_sum_partials_multi<T>, dim3(J.blk0[J.n]), dim3(256), 0, (hipStream_t)stream, J);
        SVGP_LAUNCH_CHECK();
    }
    return SVGP_OK;
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


// =====================================================================================================================
// Direct kernels for 16 input channels (every spritesVAE layer but the first; all of their data gradients but the last
// layer's): no workgroup-wide staging and no barrier in the pixel loop.
//
// Forward / data gradient, transposed GEMM  out^T[co][pixel] = sum_t W_t^T[co][ci] in^T[ci][pixel + off_t]:
//   * B operand (input): lane (r = pixel, q) loads ONE 4-vector in[pixel + off_t][4q .. 4q+3] -- a wave reads 16 pixels x 16
//     channels = 16 x 64 contiguous bytes (float), fully coalesced, straight from global memory (the 3 x 3 overlap of the taps is
//     served by L1 / L2); its four components are the B operands of four MFMAs whose k index runs over channels {j, 4+j, 8+j,
//     12+j} -- the order of the contraction index is free as long as A uses the same one;
//   * A operand (weights): W_t[ci = 4q + j][co = lane & 15], NT x 4 registers per lane, loaded once per workgroup;
//   * D: lane (r = pixel, q) holds out[pixel][4q .. 4q+3] (float64: the weight rows are permuted so that the q + 4 reg row
//     map of the f64 MFMA lands on the same channels) -- one 4-vector store per lane, 16 x 64 contiguous bytes per wave.
// A wave walks the rows of a 16-pixel-wide strip; with several waves per SIMD one wave's loads run under the others' MFMAs.
// =====================================================================================================================
// 128 bytes of zeros: an out-of-image tap reads from here (the ADDRESS is selected, not the loaded value: a select on the value
// would sit right behind the load and put its s_waitcnt in the middle of the MFMA sequence the load is meant to run under)
__device__ __attribute__((aligned(128))) double g_conv_zero[16];
// exp of the ELU epilogue of the rolling kernel.  float32: v_exp_f32 on x * log2(e) (2 instructions; the library expf is ~25 per
// element, and the epilogue's VALU work does not hide under the other waves' MFMAs: tools/micro/mfma_f32_issue.hip -- 36 MFMAs
// per row alone 144 TFLOP/s, with the register rotation 131, with expf + store 108); arguments are <= 0 here, absolute error
// <= 2e-7.  float64: the library exp.
__device__ __forceinline__ float conv_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double conv_exp(double x) { return exp(x); }
template <typename T> struct DirT;
template <> struct DirT<float> { static __device__ __forceinline__ int corow(int i) { return i; } };
template <> struct DirT<double> { static __device__ __forceinline__ int corow(int i) { return ((i & 3) << 2) | (i >> 2); } };

template <typename T, int NT>
__global__ __launch_bounds__(256) void k_conv16_fwd(svgp_conv_desc d, int strips, int R, int nseg, const T* __restrict__ in,
                                                    const T* __restrict__ w, const T* __restrict__ bias,
                                                    T* __restrict__ out) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    // consecutive work items (strips of one image) on one XCD: workgroup ids go round-robin over the 8 XCDs
    int b = blockIdx.x;
    { const int per = (int)gridDim.x >> 3; if ((per << 3) == (int)gridDim.x) b = (b & 7) * per + (b >> 3); }
    const int xs = b % nseg, st = (b / nseg) % strips, n = b / (nseg * strips);
    const int segw = d.Ws < 16 ? d.Ws : 16, rpw = 16 / segw;          // narrow images: a wave step covers rpw rows
    const int ry = r / segw, px = xs * 16 + (r - ry * segw);
    const int co_a = DirT<T>::corow(r);
    T wr[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[t][j] = co_a < d.Co ? w[d.woff[t] + (4 * q + j) * d.Co + co_a] : T(0);
    v4 bv = {0, 0, 0, 0};
    if (d.act)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = (4 * q + g < d.Co) ? bias[4 * q + g] : T(0);
    const T* inn = in + (size_t)n * d.Hi * d.Wi * 16;
    const int y_end = min(d.Hs, (st + 1) * R);
    constexpr int TCH = NT <= 9 ? NT : 8;                           // taps in flight
    for (int y0 = st * R + wave * rpw; y0 < y_end; y0 += 4 * rpw) {
        const int y = y0 + ry;
        const bool vo = ry < rpw && y < y_end && px < d.Ws;
        typename MF::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TCH) {
            v4 f[TCH];
#pragma unroll
            for (int u = 0; u < TCH; ++u) {
                const int t = t0 + u;
                f[u] = v4{0, 0, 0, 0};
                if (t < NT) {
                    const int yi = y * d.sy + d.oy[t], xi = px * d.sx + d.ox[t];
                    if (vo && (unsigned)yi < (unsigned)d.Hi && (unsigned)xi < (unsigned)d.Wi)
                        f[u] = *reinterpret_cast<const v4*>(inn + ((size_t)yi * d.Wi + xi) * 16 + 4 * q);
                }
            }
#pragma unroll
            for (int u = 0; u < TCH; ++u) {
                const int t = t0 + u;
                if (t < NT) {
                    acc0 = MF::mma(wr[t][0], f[u][0], acc0);
                    acc1 = MF::mma(wr[t][1], f[u][1], acc1);
                    acc0 = MF::mma(wr[t][2], f[u][2], acc0);
                    acc1 = MF::mma(wr[t][3], f[u][3], acc1);
                }
            }
        }
        if (vo) {
            v4 v;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                T e = acc0[g] + acc1[g];
                if (d.act) e += bv[g];
                if (d.act == 1) e = e > 0 ? e : (T)(exp(e) - T(1));
                v[g] = e;
            }
            T* o = out + (((size_t)n * d.Ho + (y * d.osy + d.ooy)) * d.Wo + (px * d.osx + d.oox)) * d.Co;
            if (d.Co == 16) {
                *reinterpret_cast<v4*>(o + 4 * q) = v;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) if (4 * q + g < d.Co) o[4 * q + g] = v[g];
            }
        }
    }
}

// The same GEMM for tap tables that form a full NR x NC grid (rows oy0 + k, any column offsets; conv 3 x 3 / 2 x 2, the parity
// classes of the fused-upsample and transposed stride-2 layers, the 4 x 4 table of the fused-upsample data gradient): a wave owns
// RW CONSECUTIVE output rows of its 16-pixel strip and keeps the NR x NC input vectors in registers -- going one output row down
// shifts them by SH = sy rows, so only min(SH, NR) x NC vectors are new; they are requested before the MFMAs of the current row
// (PF) and arrive under them.  3 x 3, stride 1: 3 loads per 36 MFMAs instead of 9, none on the critical path.
template <typename T, int NR, int NC, int SH, bool PF, bool FULL>      // FULL: Ws a multiple of 16 and 16 output channels
__global__ __launch_bounds__(256) void k_conv16_fwd_roll(svgp_conv_desc d, int strips, int RW, int nseg, int ntask,
                                                         const T* __restrict__ in, const T* __restrict__ w,
                                                         const T* __restrict__ bias, T* __restrict__ out) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    constexpr int NEW = SH < NR ? SH : NR, KEEP = NR - NEW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    // persistent workgroups: workgroup `b` takes a contiguous range of tasks (strip x segment x image); ids that share an XCD
    // (b mod 8) take neighbouring ranges.  The tap weights are loaded once per workgroup.
    int b = blockIdx.x;
    const int G = (int)gridDim.x;
    { const int per8 = G >> 3; if ((per8 << 3) == G) b = (b & 7) * per8 + (b >> 3); }
    const int per = (ntask + G - 1) / G, t_beg = b * per, t_end = min(ntask, t_beg + per);
    const int co_a = DirT<T>::corow(r), co_c = min(co_a, d.Co - 1);
    T wr[NR][NC][4];
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const T v = w[d.woff[k * NC + c] + (4 * q + j) * d.Co + co_c];      // unconditional load, then select
                wr[k][c][j] = co_a < d.Co ? v : T(0);
            }
    v4 bv = {0, 0, 0, 0};
    if (d.act)
#pragma unroll
        for (int g = 0; g < 4; ++g) { const T v = bias[min(4 * q + g, d.Co - 1)]; bv[g] = (4 * q + g < d.Co) ? v : T(0); }
    const int oy0 = d.oy[0];
    for (int task = t_beg; task < t_end; ++task) {
        const int xs = task % nseg, st = (task / nseg) % strips, n = task / (nseg * strips);
        const int px = xs * 16 + r;
        const T* inn = in + (size_t)n * d.Hi * d.Wi * 16 + 4 * q;
        const int ya = (st * 4 + wave) * RW, yb = min(d.Hs, ya + RW);
        if (ya >= yb) continue;
        const bool vx = FULL || px < d.Ws;
        int xc[NC];
        bool xok[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int xi = px * d.sx + d.ox[c];
            xok[c] = vx && (unsigned)xi < (unsigned)d.Wi;
            xc[c] = min(max(xi, 0), d.Wi - 1) * 16;
        }
        // branch-free: out-of-image taps read the zero page
        const T* zp = reinterpret_cast<const T*>(g_conv_zero) + 4 * q;
        auto ld = [&](int yi, int c) -> v4 {
            const T* src = (xok[c] && (unsigned)yi < (unsigned)d.Hi) ? inn + (size_t)yi * d.Wi * 16 + xc[c] : zp;
            return *reinterpret_cast<const v4*>(src);
        };
        v4 buf[NR][NC], nxt[NEW][NC];
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int c = 0; c < NC; ++c) buf[k][c] = ld(ya * d.sy + oy0 + k, c);
        for (int y = ya; y < yb; ++y) {
            if (PF) {                                           // (last row: a clamped, unused request)
#pragma unroll
                for (int k = 0; k < NEW; ++k)
#pragma unroll
                    for (int c = 0; c < NC; ++c) nxt[k][c] = ld((y + 1) * d.sy + oy0 + KEEP + k, c);
            }
            typename MF::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    acc0 = MF::mma(wr[k][c][0], buf[k][c][0], acc0);
                    acc1 = MF::mma(wr[k][c][1], buf[k][c][1], acc1);
                    acc0 = MF::mma(wr[k][c][2], buf[k][c][2], acc0);
                    acc1 = MF::mma(wr[k][c][3], buf[k][c][3], acc1);
                }
            if (vx) {
                v4 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    T e = acc0[g] + acc1[g];
                    if (d.act) e += bv[g];
                    if (d.act == 1) e = e > 0 ? e : (T)(conv_exp(e) - T(1));
                    v[g] = e;
                }
                T* o = out + (((size_t)n * d.Ho + (y * d.osy + d.ooy)) * d.Wo + (px * d.osx + d.oox)) * d.Co;
                if (FULL || d.Co == 16) {
                    *reinterpret_cast<v4*>(o + 4 * q) = v;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (4 * q + g < d.Co) o[4 * q + g] = v[g];
                }
            }
#pragma unroll
            for (int k = 0; k < KEEP; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) buf[k][c] = buf[k + NEW][c];
#pragma unroll
            for (int k = 0; k < NEW; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) buf[KEEP + k][c] = PF ? nxt[k][c] : ld((y + 1) * d.sy + oy0 + KEEP + k, c);
        }
    }
}

// Forward for 16 -> CT (3) output channels (the last decoder layer), stride 1, full NR x NC tap grid.  The rolling kernel above
// spends 4 NT MFMAs per 16 pixels on an output tile of which 13 of 16 channel rows are padding (123 us at 64 x 64 x 500 frames for
// 156 MB of traffic).  Here the contraction runs over the input channels only and the TAPS sit in the MFMA row index:
//   P[j = (t, co)][pixel'] = sum_ci W_t[ci][co] in[pixel'][ci]        (JT = NT CT = 27 rows in NB = 2 blocks: 8 MFMAs per 16 INPUT pixels)
//   out[y][x][co] = bias + sum_(kr, kc) P_(input row y + oy0 + kr)[(kr, kc, co)][x + ox0 + kc]
// A wave owns 16 - (NC - 1) = 14 output columns (the 16 input columns they read) and RW rows: every input row is loaded once (one
// 4-vector per lane, three rows ahead of its use in three alternating register sets), multiplied, and its P written to a
// wave-private LDS ring of NR rows ([pixel][j], pixel stride JS = 36: conflict-free for the 128-bit writes and for the gather);
// the 42 values of an output row segment are then 9 LDS reads each by 42 lanes, stored as one contiguous run.
template <typename T, int NR, int NC, int CT>
__global__ __launch_bounds__(256) void k_conv16_thin_fwd(svgp_conv_desc d, int ntask, int nseg, int strips, int RW,
                                                         const T* __restrict__ in, const T* __restrict__ w,
                                                         const T* __restrict__ bias, T* __restrict__ out) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    static_assert(NR == 3, "the row loop is unrolled by the ring period");
    constexpr int NT = NR * NC, JT = NT * CT, NB = (JT + 15) / 16, NPO = 16 - (NC - 1), JS = 16 * NB + 4, ROWP = 16 * JS;
    __shared__ __align__(16) T smem[4 * NR * ROWP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    T* ring = smem + wave * (NR * ROWP);
    const int oy0 = d.oy[0], ox0 = d.ox[0], Hi = d.Hi, Wi = d.Wi, Ho = d.Ho, Wo = d.Wo, Hs = d.Hs, Ws = d.Ws, act = d.act;
    // A operand: row i of block nb is the logical row j = 16 nb + corow(i) (float64: so that the D rows of a lane are consecutive j)
    T wr[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = 16 * nb + DirT<T>::corow(r), jc = min(j, JT - 1), t = jc / CT, co = jc - t * CT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const T v = w[d.woff[t] + (4 * q + ks) * CT + co];
            wr[nb][ks] = j < JT ? v : T(0);
        }
    }
    // gather: lane e < NPO CT owns output element (column xl, channel gco) of the segment
    const int e = min(lane, NPO * CT - 1), xl = e / CT, gco = e - xl * CT;
    const T bvs = act ? bias[gco] : T(0);
    const T* gptr = ring + xl * JS + gco;
    T* wptr = ring + r * JS + 4 * q;
    // a task = RW output rows x one segment of one image, per WAVE (RW a multiple of 3: the row loop runs whole ring periods)
    int b = blockIdx.x;
    const int G = (int)gridDim.x;
    { const int per8 = G >> 3; if ((per8 << 3) == G) b = (b & 7) * per8 + (b >> 3); }
    for (int task = b * 4 + wave; task < ntask; task += G * 4) {
        const int xs = task % nseg, rb = (task / nseg) % strips, n = task / (nseg * strips);
        const int ya = rb * RW, yb = min(Hs, ya + RW);
        const int xo = xs * NPO, px = xo + ox0 + r, yi0 = ya + oy0, nin = yb - ya + NR - 1;
        const bool colok = (unsigned)px < (unsigned)Wi;
        const T* inn = in + (size_t)n * Hi * Wi * 16 + (unsigned)(min(max(px, 0), Wi - 1) * 16 + 4 * q);
        T* on = out + ((size_t)n * Ho * Wo + xo) * CT + e;
        const bool st_ok = lane < NPO * CT && xo + xl < Ws;
        T* trash = reinterpret_cast<T*>(svgp_conv_trash) + lane;
        auto gload = [&](int ii) -> v4 {                                   // input row ii of the task (clamped: rows past the last
            const int gy = yi0 + min(ii, nin - 1);                          // one re-read it, rows outside the image a valid one)
            return *reinterpret_cast<const v4*>(inn + (size_t)min(max(gy, 0), Hi - 1) * Wi * 16);
        };
        // one input row: P -> ring slot SLOT; the refill of its register set; with OUT the output row that this row completes.
        // Branch-free: rows past the task's last one (a short last block) are computed from a repeated input row and stored to the
        // trash slots, like the lanes that own no output element -- under a predicate or an early exit the wait-count pass has to
        // assume the shortest path and waits for the requests issued one row ago instead of three.
        auto step = [&](int ii, v4& R, auto SLOTc, auto OUTc) {
            constexpr int SLOT = decltype(SLOTc)::value;
            const bool ok = colok && (unsigned)(yi0 + ii) < (unsigned)Hi;
            v4 v;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) v[ks] = ok ? R[ks] : T(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            R = gload(ii + 3);
            __builtin_amdgcn_sched_barrier(0);
            typename MF::acc_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = typename MF::acc_t{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = MF::mma(wr[nb][ks], v[ks], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<v4*>(wptr + SLOT * ROWP + 16 * nb) = v4{acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]};
            if constexpr (decltype(OUTc)::value) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                T sum = bvs;
#pragma unroll
                for (int kr = 0; kr < NR; ++kr)
#pragma unroll
                    for (int kc = 0; kc < NC; ++kc)
                        sum += gptr[((SLOT + 1 + kr) % NR) * ROWP + kc * JS + (kr * NC + kc) * CT];
                if (act == 1) sum = sum > 0 ? sum : (T)(conv_exp(sum) - T(1));
                const int y = ya + ii - (NR - 1);
                *((st_ok && y < yb) ? on + (size_t)y * Wo * CT : trash) = sum;
            } else {
                trash[64 + 64 * SLOT] = T(0);      // the wait counts of the loop are the minimum over the paths into it: keep the two opening rows alike
            }
        };
        const auto S0 = std::integral_constant<int, 0>{};
        const auto S1 = std::integral_constant<int, 1>{};
        const auto S2 = std::integral_constant<int, 2>{};
        const auto NO = std::integral_constant<bool, false>{};
        const auto YES = std::integral_constant<bool, true>{};
        v4 R0 = gload(0), R1 = gload(1), R2 = gload(2);
        trash[192] = T(0);
        step(0, R0, S0, NO);
        step(1, R1, S1, NO);
        for (int ii = 2; ii < nin; ii += 3) {
            step(ii, R2, S2, YES);
            step(ii + 1, R0, S0, YES);
            step(ii + 2, R1, S1, YES);
        }
    }
}

// Weight gradient, rolling form of the kernel below for images at least 16 pixels wide: a wave owns RW consecutive output rows of a
// 16-pixel strip; its LDS region is a ring of HH = (tap row range) input rows (slot = input row mod HH), so an output row stages only
// the SY new input rows -- requested (together with the next row's dout / out values) BEFORE the MFMAs of the current row and written
// to LDS after them.
template <typename T, int NT, int SY>
__global__ __launch_bounds__(256) void k_conv16_wgrad_roll(ConvLaunch L, int nwg, int RW, const T* __restrict__ in,
                                                           const T* __restrict__ outv, T* __restrict__ dout,
                                                           T* __restrict__ part, int part_stride, T* __restrict__ part_b,
                                                           int lds_per_wave) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    constexpr int NI = 3;                                              // 16-pixel staging passes per input row (HW <= 48)
    extern __shared__ __align__(32) unsigned char smem_raw[];
    const int cls = blockIdx.y;
    const svgp_conv_desc& d = L.d[cls];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    T* lds = reinterpret_cast<T*>(smem_raw) + (size_t)wave * lds_per_wave;
    int oy0, oy1, ox0, ox1;
    tap_range(d, oy0, oy1, ox0, ox1);
    const int nseg = (d.Ws + 15) / 16, nrb = (d.Hs + RW - 1) / RW;
    const int HH = oy1 - oy0 + 1, HW = 15 * d.sx + (ox1 - ox0) + 1, PS = d.sx == 1 ? 16 : 24;
    typename MF::acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = typename MF::acc_t{0, 0, 0, 0};
    T bsum = 0;
    int aidx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aidx[s] = ((4 * s + q) * d.sx) * PS + r;
    const int sp = lane >> 2, sc = 4 * (lane & 3);                     // staging: pixel sp + 16 i, channels sc ..
    const int ntask = d.n * nrb * nseg;
    for (int task = blockIdx.x * 4 + wave; task < ntask; task += nwg * 4) {
        const int xs = task % nseg, rb = (task / nseg) % nrb, n = task / (nseg * nrb);
        const T* inn = in + (size_t)n * d.Hi * d.Wi * 16 + sc;
        const int ya = rb * RW, yb = min(d.Hs, ya + RW), xf = xs * 16;
        const int X0 = xf * d.sx + ox0;
        const int ni = (HW + 15) >> 4;
        const T* zp = reinterpret_cast<const T*>(g_conv_zero) + sc;
        auto gload = [&](int gy, int i) -> v4 {                          // branch-free: out-of-image pixels read the zero page
            const int p = sp + 16 * i, gx = X0 + p;
            v4 v = {0, 0, 0, 0};
            if (i < ni) {                                               // (uniform)
                const T* src = (p < HW && (unsigned)gy < (unsigned)d.Hi && (unsigned)gx < (unsigned)d.Wi)
                                   ? inn + ((size_t)gy * d.Wi + gx) * 16 : zp;
                v = *reinterpret_cast<const v4*>(src);
            }
            return v;
        };
        // ring of HH input rows: slot of input row ya * sy + oy0 + rel = rel mod HH, tracked incrementally (`base` = slot of the
        // first row the current output row needs)
        auto wrap = [&](int x) -> int { while (x >= HH) x -= HH; while (x < 0) x += HH; return x; };
        auto lwrite = [&](int sl, int i, v4 v) {
            const int p = sp + 16 * i;
            if (p < HW) *reinterpret_cast<v4*>(lds + (sl * HW + p) * PS + sc) = v;
        };
        // first row: the whole ring
        for (int k = 0; k < HH; ++k) {
            const int gy = ya * d.sy + oy0 + k;
#pragma unroll
            for (int i = 0; i < NI; ++i) lwrite(k, i, gload(gy, i));
        }
        int base = 0;
        T cd[4], co_[4];
        size_t co_off[4];
        bool cv[4];
        auto bload = [&](int y, T* dv, T* ov, size_t* off, bool* ok) {
            const int yc = min(y, d.Hs - 1), rc = min(r, d.Co - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int x = xf + 4 * s + q, xc = min(x, d.Ws - 1);
                ok[s] = x < d.Ws && r < d.Co && y < d.Hs;
                off[s] = (((size_t)n * d.Ho + (yc * d.osy + d.ooy)) * d.Wo + (xc * d.osx + d.oox)) * d.Co + rc;
                dv[s] = *(ok[s] ? dout + off[s] : reinterpret_cast<const T*>(g_conv_zero));
                ov[s] = outv ? outv[off[s]] : T(1);
            }
        };
        bload(ya, cd, co_, co_off, cv);
        for (int y = ya; y < yb; ++y) {
            const bool more = y + 1 < yb;
            v4 pre[SY][NI];
            T nd[4], no[4];
            size_t noff[4];
            bool nv[4];
            {                                                           // (last row: clamped, unused requests)
#pragma unroll
                for (int k = 0; k < SY; ++k)
#pragma unroll
                    for (int i = 0; i < NI; ++i) pre[k][i] = gload((y + 1) * d.sy + oy1 - (SY - 1) + k, i);
                bload(y + 1, nd, no, noff, nv);
            }
            T bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                T dv = cd[s];
                if (outv) { dv *= (co_[s] > 0 ? T(1) : co_[s] + T(1)); if (cv[s]) dout[co_off[s]] = dv; }
                bv[s] = dv;
                bsum += dv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int toff = (wrap(base + d.oy[t] - oy0) * HW + (d.ox[t] - ox0)) * PS;
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[t] = MF::mma(lds[aidx[s] + toff], bv[s], acc[t]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (more) {
                base = wrap(base + SY);
#pragma unroll
                for (int k = 0; k < SY; ++k)
#pragma unroll
                    for (int i = 0; i < NI; ++i) lwrite(wrap(base + HH - SY + k), i, pre[k][i]);
#pragma unroll
                for (int s = 0; s < 4; ++s) { cd[s] = nd[s]; co_[s] = no[s]; co_off[s] = noff[s]; cv[s] = nv[s]; }
            }
        }
    }
    // ---- cross-wave combine (fixed order) and store.  D: col (co) = r, row (ci) = MF::row(q, g)
    T* red = reinterpret_cast<T*>(smem_raw);                            // 4 waves x 64 lanes x 4
    T* po = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(wave * 64 + lane) * 4 + g] = acc[t][g];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const T v = red[lane * 4 + g] + red[(64 + lane) * 4 + g] + red[(128 + lane) * 4 + g] + red[(192 + lane) * 4 + g];
                const int ci = MF::row(q, g);
                if (r < d.Co) po[d.woff[t] + ci * d.Co + r] = v;
            }
        }
    }
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < 16) {
        T v = 0;
        for (int k = 0; k < 16; ++k) v += red[k * 16 + threadIdx.x];
        part_b[((size_t)blockIdx.y * nwg + blockIdx.x) * 16 + threadIdx.x] = v;
    }
}

// The rolling weight gradient for the layers a SPRITES step is made of: 16 -> 16 channels, image width a multiple of 16, taps = a full
// NR x NC grid of consecutive row AND column offsets (3 x 3 stride 1 / 2, the 2 x 2 parity classes of the fused up-sampling layers,
// 2 x 2 stride 2).  Same mapping as k_conv16_wgrad_roll, but with every extent a template constant the row loop is ONE basic block:
//   * the 4 NT operand reads of a row are ds_read_b32 / b64 off NR per-lane row pointers with immediate offsets (no per-tap
//     address arithmetic, no loops for the ring wrap-around, no bounds branches), so they are issued ahead of the MFMAs
//     instead of four at a time in front of each tap with the LDS latency exposed nine times per row;
//   * the next row's input vectors and dout / out values are requested at the top of the row under ONE uniform branch (none on the
//     last row of a task: the roll kernel's clamped requests re-read 1 / RW of every tensor), and only the input vectors are
//     waited for at the bottom, where they go into the ring slot the row has just vacated;
//   * consecutive workgroup ids of one XCD walk neighbouring tasks (the halo rows two row blocks share stay in that XCD's L2).
template <typename T, int NR, int NC, int SY, int SX, bool ACT>
__global__ __launch_bounds__(256) void k_conv16_wgrad_grid(ConvLaunch L, int nwg, int RW, const T* __restrict__ in,
                                                           const T* __restrict__ outv, T* __restrict__ dout,
                                                           T* __restrict__ part, int part_stride, T* __restrict__ part_b) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    static_assert(SY <= NR, "a row step may not skip input rows");
    constexpr int NT = NR * NC, NEW = SY, KEEP = NR - NEW;
    constexpr int HW = 15 * SX + NC, PS = SX == 1 ? 16 : 24, NI = (HW + 15) / 16, ROWE = HW * PS;
    extern __shared__ __align__(32) unsigned char smem_raw[];
    const int cls = blockIdx.y;
    const svgp_conv_desc& d = L.d[cls];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    T* ring = reinterpret_cast<T*>(smem_raw) + (size_t)wave * (NR * ROWE);
    const int sp = lane >> 2, sc = 4 * (lane & 3);                     // staging: pixel sp + 16 i, channels sc .. sc + 3
    const T* const rbase = ring + q * SX * PS + r;                      // operand of k-step s, column c: + ((4 s SX + c) PS)
    // the descriptor is indexed by the class: its fields are copied to scalars once (left in the argument block they are re-read,
    // under branches, inside the row loop)
    const int oy0 = d.oy[0], ox0 = d.ox[0], Hi = d.Hi, Wi = d.Wi, Ho = d.Ho, Wo = d.Wo, Hs = d.Hs, osy = d.osy, osx = d.osx,
              ooy = d.ooy, oox = d.oox;
    const int nseg = d.Ws >> 4, nrb = (Hs + RW - 1) / RW, ntask = d.n * nrb * nseg;
    typename MF::acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = typename MF::acc_t{0, 0, 0, 0};
    T bsum = 0;
    int b = blockIdx.x;
    { const int per8 = nwg >> 3; if ((per8 << 3) == nwg) b = (b & 7) * per8 + (b >> 3); }
    for (int task = b * 4 + wave; task < ntask; task += nwg * 4) {
        const int xs = task % nseg, rb = (task / nseg) % nrb, n = task / (nseg * nrb);
        const T* inn = in + (size_t)n * Hi * Wi * 16;
        const int ya = rb * RW, yb = min(Hs, ya + RW), xf = xs * 16, X0 = xf * SX + ox0;
        // Every request goes to a valid address (row and column clamped into the image: uniform row pointer + 32-bit lane offset, no
        // branch, no address select); what lies outside the image is zeroed when the vector is written to the ring.
        // Lanes beyond the HW pixels of a ring row (last staging pass) repeat its last pixel: same request, same value, same ring
        // address -- the ring write needs no predicate (under one the compiler sinks the REQUEST into the predicated block, behind the
        // MFMAs, and waits for it there with vmcnt(0)).
        unsigned goff[NI], woff[NI];
        bool gok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = min(sp + 16 * i, HW - 1), gx = X0 + p;
            gok[i] = (unsigned)gx < (unsigned)Wi;
            goff[i] = (unsigned)(min(max(gx, 0), Wi - 1) * 16 + sc);
            woff[i] = (unsigned)(p * PS + sc);
        }
        auto gload = [&](int gy, int i) -> v4 {
            const T* row = inn + (size_t)min(max(gy, 0), Hi - 1) * Wi * 16;
            return *reinterpret_cast<const v4*>(row + goff[i]);
        };
        auto lwrite = [&](int slot, int i, v4 v, bool rowok) {
            if (!(rowok && gok[i])) v = v4{0, 0, 0, 0};
            *reinterpret_cast<v4*>(ring + slot * ROWE + woff[i]) = v;
        };
        unsigned doff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) doff[s] = (unsigned)(((xf + 4 * s + q) * osx + oox) * 16 + r);
        auto rowbase = [&](int y) -> size_t { return ((size_t)n * Ho + (y * osy + ooy)) * Wo * 16; };
        // the ring of NR input rows: slot of input row ya SY + oy0 + rel = rel mod NR; `base` = slot of the current row's tap row 0
        {
            v4 v[NR][NI];
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) v[k][i] = gload(ya * SY + oy0 + k, i);
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) lwrite(k, i, v[k][i], (unsigned)(ya * SY + oy0 + k) < (unsigned)Hi);
        }
        int base = 0;
        T cdA[4], coA[4], cdB[4], coB[4];
        {
            const size_t rb0 = rowbase(ya);
#pragma unroll
            for (int s = 0; s < 4; ++s) { cdA[s] = (dout + rb0)[doff[s]]; coA[s] = ACT ? (outv + rb0)[doff[s]] : T(1); }
        }
        // One output row; branch-free (with the next row's requests under `if (more)` the wait-count pass could not tell them from the
        // ones it has to wait for, and every row began with s_waitcnt vmcnt(0)): on the last row of a task the requests re-read rows
        // this wave has just read (cache hits) and their values are dropped.  cd / co_: this row's dout / out values (requested one row
        // ago), nd / no: the next row's -- the caller alternates two register sets (a copy at the end of the row would wait for the
        // requests it has just issued).
        auto row = [&](int y, T (&cd)[4], T (&co_)[4], T (&nd)[4], T (&no)[4]) {
            const bool more = y + 1 < yb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (1) requests for the next row, pinned in front of this row's work
            const int yn = more ? y + 1 : y, gy0 = yn * SY + oy0 + KEEP;
            v4 pre[NEW][NI];
#pragma unroll
            for (int k = 0; k < NEW; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) pre[k][i] = gload(gy0 + k, i);
            {
                const size_t rb1 = rowbase(yn);
#pragma unroll
                for (int s = 0; s < 4; ++s) { nd[s] = (dout + rb1)[doff[s]]; no[s] = ACT ? (outv + rb1)[doff[s]] : T(1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (2) this row: dpre = dout elu'(out), its store, the 4 NT products
            const T* rk[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) { int sl = base + k; if (sl >= NR) sl -= NR; rk[k] = rbase + sl * ROWE; }
            T bv[4];
            {
                T* dr = dout + rowbase(y);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    T dv = cd[s];
                    if (ACT) { dv *= (co_[s] > 0 ? T(1) : co_[s] + T(1)); dr[doff[s]] = dv; }
                    bv[s] = dv;
                    bsum += dv;
                }
            }
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[k * NC + c] = MF::mma(rk[k][(4 * s * SX + c) * PS], bv[s], acc[k * NC + c]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (3) the next row's new input rows go where this row's first SY tap rows were
#pragma unroll
            for (int k = 0; k < NEW; ++k) {
                int sl = base + k; if (sl >= NR) sl -= NR;
#pragma unroll
                for (int i = 0; i < NI; ++i) lwrite(sl, i, pre[k][i], (unsigned)(gy0 + k) < (unsigned)Hi);
            }
            base += SY; if (base >= NR) base -= NR;
        };
        for (int y = ya; y < yb; y += 2) {
            row(y, cdA, coA, cdB, coB);
            if (y + 1 >= yb) break;
            row(y + 1, cdB, coB, cdA, coA);
        }
    }
    // ---- cross-wave combine (fixed order) and store.  D: col (co) = r, row (ci) = MF::row(q, g)
    T* red = reinterpret_cast<T*>(smem_raw);                            // 4 waves x 64 lanes x 4
    T* po = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(wave * 64 + lane) * 4 + g] = acc[t][g];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const T v = red[lane * 4 + g] + red[(64 + lane) * 4 + g] + red[(128 + lane) * 4 + g] + red[(192 + lane) * 4 + g];
                po[d.woff[t] + MF::row(q, g) * 16 + r] = v;
            }
        }
    }
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < 16) {
        T v = 0;
        for (int k = 0; k < 16; ++k) v += red[k * 16 + threadIdx.x];
        part_b[((size_t)blockIdx.y * nwg + blockIdx.x) * 16 + threadIdx.x] = v;
    }
}

// The same row pipeline for the THIN layers (CT = 3 channels on one side: spritesVAE's first encoder layer and the representation
// network's first layer, 3 -> 16; the last decoder layer, 16 -> 3), whose weight gradient k_convS_wgrad formed from per-lane dword
// gathers issued right in front of their MFMA (256 us at 64 x 64 x 500 frames for 418 MB of traffic).
//   MODE 0 (3 -> 16): dW[(t, ci)][co] = sum_pix dpre[pix][co] in[pix (+) t][ci]
//     wide operand A = dpre [co = r][pixel k] exactly as in k_conv16_wgrad_grid (ELU', in-place store, bias sum, alternating
//     register sets); thin operand B = in [pixel k][j = (t, ci)];
//   MODE 1 (16 -> 3, stride 1): dW[ci][(t, co)] = sum_pix' in[pix'][ci] dpre[pix' (-) t][co] -- the sum re-indexed over INPUT pixels so
//     that the 16-channel operand is tap-independent: wide operand A = in [ci = r][pixel' k], thin operand B = dpre (ELU' applied
//     by a separate pass over the 3-channel tensor), tap grid mirrored.
// The thin operand is read from a wave-private LDS ring of NR rows -- a ring row is the contiguous run of HW x CT values (54 floats
// for 3 x 3, stride 1): ONE dword request per lane per new row; ring rows are ROWE elements apart so that the 18 / 24 values one
// k-step touches in each row fall into disjoint banks.  j in blocks of 16: 2 NB MFMAs per 4 pixels; per output row 8 MFMAs, 8 LDS
// reads, 5 - 9 requests, 4 stores: the kernel runs at the HBM rate.
template <typename T, int NR, int NC, int S, int CT, bool ACT, int MODE>
__global__ __launch_bounds__(256) void k_convS_wgrad_ring(svgp_conv_desc d, int nwg, int RW, const T* __restrict__ in,
                                                          const T* __restrict__ outv, T* __restrict__ dout,
                                                          T* __restrict__ part, int part_stride, T* __restrict__ part_b) {
    typedef SvgpMfma<T> MF;
    constexpr int NT = NR * NC, JT = NT * CT, NB = (JT + 15) / 16, KEEP = NR - S;
    constexpr int HW = 15 * S + NC, RUN = HW * CT, NI = (RUN + 63) / 64, ROWE = S == 1 ? 84 : 96;
    static_assert(S <= NR && RUN <= ROWE && (MODE == 0 || (S == 1 && !ACT)), "ring row / mode");
    __shared__ T smem[4 * NR * ROWE > 1024 ? 4 * NR * ROWE : 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    T* ring = smem + wave * (NR * ROWE);
    const int oy0 = d.oy[0], ox0 = d.ox[0], Hi = d.Hi, Wi = d.Wi, Ho = d.Ho, Wo = d.Wo, osy = d.osy, osx = d.osx, ooy = d.ooy,
              oox = d.oox;
    // iteration space: output pixels (MODE 0) / input pixels (MODE 1); thin tensor: the input / the output gradient
    const int Hit = MODE ? Hi : d.Hs, Wit = MODE ? Wi : d.Ws, TH = MODE ? Ho : Hi, TW = MODE ? Wo : Wi;
    const T* thin = MODE ? dout : in;
    const int nseg = Wit >> 4, nrb = (Hit + RW - 1) / RW, ntask = d.n * nrb * nseg;
    // this lane's column j = (t, c) of each block: tap row (ring slot offset) and element offset inside a ring row
    int jkr[NB], jofs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = min(16 * nb + r, JT - 1), t = j / CT, c = j - t * CT, kr = t / NC, kc = t - kr * NC;
        jkr[nb] = MODE ? NR - 1 - kr : kr;
        jofs[nb] = (q * S + (MODE ? NC - 1 - kc : kc)) * CT + c;
    }
    typename MF::acc_t acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = typename MF::acc_t{0, 0, 0, 0};
    T bsum = 0;
    int b = blockIdx.x;
    { const int per8 = nwg >> 3; if ((per8 << 3) == nwg) b = (b & 7) * per8 + (b >> 3); }
    for (int task = b * 4 + wave; task < ntask; task += nwg * 4) {
        const int xs = task % nseg, rb = (task / nseg) % nrb, n = task / (nseg * nrb);
        const T* thn = thin + (size_t)n * TH * TW * CT;
        const int ya = rb * RW, yb = min(Hit, ya + RW), xf = xs * 16, X0 = MODE ? xf - ox0 - (NC - 1) : xf * S + ox0;
        auto trow0 = [&](int y) -> int { return MODE ? y - oy0 - (NR - 1) : y * S + oy0; };      // first thin row of wide row y
        // staging: element e = lane + 64 i of the run (lanes beyond it repeat its last element: no predicate on the ring write)
        unsigned goff[NI], woff[NI];
        bool gok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = min(lane + 64 * i, RUN - 1), p = e / CT, gx = X0 + p;
            gok[i] = (unsigned)gx < (unsigned)TW;
            goff[i] = (unsigned)(min(max(gx, 0), TW - 1) * CT + (e - p * CT));
            woff[i] = (unsigned)e;
        }
        auto gload = [&](int gy, int i) -> T { return (thn + (size_t)min(max(gy, 0), TH - 1) * TW * CT)[goff[i]]; };
        auto lwrite = [&](int slot, int i, T v, bool rowok) { ring[slot * ROWE + woff[i]] = (rowok && gok[i]) ? v : T(0); };
        unsigned doff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            doff[s] = MODE ? (unsigned)((xf + 4 * s + q) * 16 + r) : (unsigned)(((xf + 4 * s + q) * osx + oox) * 16 + r);
        auto rowbase = [&](int y) -> size_t {
            return MODE ? ((size_t)n * Hi + y) * Wi * 16 : ((size_t)n * Ho + (y * osy + ooy)) * Wo * 16;
        };
        const T* wide = MODE ? in : dout;
        {
            T v[NR][NI];
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) v[k][i] = gload(trow0(ya) + k, i);
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) lwrite(k, i, v[k][i], (unsigned)(trow0(ya) + k) < (unsigned)TH);
        }
        int base = 0;
        T cdA[4], coA[4], cdB[4], coB[4];
        {
            const size_t rb0 = rowbase(ya);
#pragma unroll
            for (int s = 0; s < 4; ++s) { cdA[s] = (wide + rb0)[doff[s]]; coA[s] = ACT ? (outv + rb0)[doff[s]] : T(1); }
        }
        auto row = [&](int y, T (&cd)[4], T (&co_)[4], T (&nd)[4], T (&no)[4]) {      // (see k_conv16_wgrad_grid)
            const bool more = y + 1 < yb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int yn = more ? y + 1 : y, gy0 = trow0(yn) + KEEP;
            T pre[S][NI];
#pragma unroll
            for (int k = 0; k < S; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) pre[k][i] = gload(gy0 + k, i);
            {
                const size_t rb1 = rowbase(yn);
#pragma unroll
                for (int s = 0; s < 4; ++s) { nd[s] = (wide + rb1)[doff[s]]; no[s] = ACT ? (outv + rb1)[doff[s]] : T(1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            const T* rj[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { int sl = base + jkr[nb]; if (sl >= NR) sl -= NR; rj[nb] = ring + sl * ROWE + jofs[nb]; }
            T bv[4];
            {
                T* dr = dout + rowbase(y);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    T dv = cd[s];
                    if (ACT) { dv *= (co_[s] > 0 ? T(1) : co_[s] + T(1)); dr[doff[s]] = dv; }
                    bv[s] = dv;
                    if (MODE == 0) bsum += dv;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = MF::mma(bv[s], rj[nb][4 * s * S * CT], acc[nb]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < S; ++k) {
                int sl = base + k; if (sl >= NR) sl -= NR;
#pragma unroll
                for (int i = 0; i < NI; ++i) lwrite(sl, i, pre[k][i], (unsigned)(gy0 + k) < (unsigned)TH);
            }
            base += S; if (base >= NR) base -= NR;
        };
        for (int y = ya; y < yb; y += 2) {
            row(y, cdA, coA, cdB, coB);
            if (y + 1 >= yb) break;
            row(y + 1, cdB, coB, cdA, coA);
        }
    }
    // cross-wave combine (fixed order); D: column j = 16 nb + r, row i (wide channel) = MF::row(q, g)
    T* red = smem;
    T* po = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(wave * 64 + lane) * 4 + g] = acc[nb][g];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const T v = red[lane * 4 + g] + red[(64 + lane) * 4 + g] + red[(128 + lane) * 4 + g] + red[(192 + lane) * 4 + g];
                const int j = 16 * nb + r, i = MF::row(q, g);
                if (j < JT) {
                    const int t = j / CT, c = j - t * CT;
                    po[d.woff[t] + (MODE ? i * CT + c : c * 16 + i)] = v;
                }
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < 16) {
            T v = 0;
            for (int k = 0; k < 16; ++k) v += red[k * 16 + threadIdx.x];
            part_b[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
        }
    }
}

// Weight gradient for 16 input channels, fused with the ELU reverse and the bias gradient:
//   dpre = dout * elu'(out) (written back in place: the data gradient reads it), db[co] = sum dpre, dW_t[ci][co] = sum in * dpre.
// GEMM per tap: A[i = ci][k = pixel] = in, B[k = pixel][j = co] = dpre, k-steps of 4 pixels of a 16-pixel segment.
//   * B: lane (r = co, q) loads dout / out [pixel 4s + q][r] directly (64 contiguous bytes per 16 lanes), applies elu', stores;
//   * A: each WAVE stages the halo of its own segment (all 16 channels of a pixel by four lanes, one 4-vector each) into a
//     wave-private LDS region and reads it back as [ci = r][pixel]; LDS operations of one wave complete in order, so the pixel
//     loop has no workgroup barrier; pixel stride 16 (input stride 1) or 24 (stride 2): the four q groups hit distinct banks;
//   * dW_t stays in NT accumulator tiles per wave; one cross-wave combine per workgroup, fixed-order partial sums after.
template <typename T, int NT>
__global__ __launch_bounds__(256) void k_conv16_wgrad(ConvLaunch L, int nwg, int R, const T* __restrict__ in,
                                                      const T* __restrict__ outv, T* __restrict__ dout, T* __restrict__ part,
                                                      int part_stride, T* __restrict__ part_b, int lds_per_wave) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(32) unsigned char smem_raw[];
    const int cls = blockIdx.y;
    const svgp_conv_desc& d = L.d[cls];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    T* lds = reinterpret_cast<T*>(smem_raw) + (size_t)wave * lds_per_wave;
    int oy0, oy1, ox0, ox1;
    tap_range(d, oy0, oy1, ox0, ox1);
    const int segw = d.Ws < 16 ? d.Ws : 16, rpw = 16 / segw;
    const int nseg = (d.Ws + 15) / 16, strips = (d.Hs + R - 1) / R;
    const int HH = (rpw - 1) * d.sy + (oy1 - oy0) + 1, HW = (segw - 1) * d.sx + (ox1 - ox0) + 1, PS = d.sx == 1 ? 16 : 24;
    const unsigned magic = 0xFFFFFFFFu / (unsigned)HW + 1u;           // p / HW = umulhi(p, magic) for the small p used here
    // k-step s covers segment pixels 4s + q -> (row ry_s, column pxl_s) of the segment
    int ry_s[4], px_s[4], aidx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int kp = 4 * s + q;
        ry_s[s] = kp / segw; px_s[s] = kp - ry_s[s] * segw;
        aidx[s] = ry_s[s] < rpw ? ((ry_s[s] * d.sy) * HW + px_s[s] * d.sx) * PS + r : r;   // (unused pixel: any staged value)
    }
    typename MF::acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = typename MF::acc_t{0, 0, 0, 0};
    T bsum = 0;
    const int ntask = d.n * strips;
    for (int task = blockIdx.x; task < ntask; task += nwg) {
        const int n = task / strips, st = task - n * strips;
        const T* inn = in + (size_t)n * d.Hi * d.Wi * 16;
        const int y_beg = st * R, y_end = min(d.Hs, y_beg + R);
        const int nrow_steps = (y_end - y_beg + rpw - 1) / rpw;
        for (int it = wave; it < nrow_steps * nseg; it += 4) {
            const int ys = it / nseg, xs = it - ys * nseg;
            const int yf = y_beg + ys * rpw, xf = xs * 16;               // first row / column of the segment
            // ---- stage the halo: pixel p = lane / 4 + 16 i, channels 4 (lane % 4) ..
            const int Y0 = yf * d.sy + oy0, X0 = xf * d.sx + ox0;
            for (int p = lane >> 2; p < HH * HW; p += 16) {
                const int yy = (int)__umulhi((unsigned)p, magic), xx = p - yy * HW;
                const int gy = Y0 + yy, gx = X0 + xx;
                v4 v = {0, 0, 0, 0};
                if ((unsigned)gy < (unsigned)d.Hi && (unsigned)gx < (unsigned)d.Wi)
                    v = *reinterpret_cast<const v4*>(inn + ((size_t)gy * d.Wi + gx) * 16 + 4 * (lane & 3));
                *reinterpret_cast<v4*>(lds + p * PS + 4 * (lane & 3)) = v;
            }
            // ---- B operands: dpre of the segment's 16 pixels, 4 per k-step
            T bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int y = yf + ry_s[s], x = xf + px_s[s];
                bv[s] = 0;
                if (ry_s[s] < rpw && y < y_end && x < d.Ws && r < d.Co) {
                    const size_t o = (((size_t)n * d.Ho + (y * d.osy + d.ooy)) * d.Wo + (x * d.osx + d.oox)) * d.Co + r;
                    T dv = dout[o];
                    if (outv) { const T ov = outv[o]; dv *= (ov > 0 ? T(1) : ov + T(1)); dout[o] = dv; }
                    bv[s] = dv;
                    bsum += dv;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int toff = ((d.oy[t] - oy0) * HW + (d.ox[t] - ox0)) * PS;
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[t] = MF::mma(lds[aidx[s] + toff], bv[s], acc[t]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- cross-wave combine (fixed order) and store.  D: col (co) = r, row (ci) = MF::row(q, g)
    T* red = reinterpret_cast<T*>(smem_raw);                            // 4 waves x 64 lanes x 4
    T* po = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(wave * 64 + lane) * 4 + g] = acc[t][g];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const T v = red[lane * 4 + g] + red[(64 + lane) * 4 + g] + red[(128 + lane) * 4 + g] + red[(192 + lane) * 4 + g];
                const int ci = MF::row(q, g);
                if (r < d.Co) po[d.woff[t] + ci * d.Co + r] = v;
            }
        }
    }
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < 16) {
        T v = 0;
        for (int k = 0; k < 16; ++k) v += red[k * 16 + threadIdx.x];
        part_b[((size_t)blockIdx.y * nwg + blockIdx.x) * 16 + threadIdx.x] = v;
    }
}

// =====================================================================================================================
// Thin layers (3 input OR 3 output channels: the first encoder / representation layers, the last decoder layer): with one
// 16 x 16 x 4 MFMA per (tap, 4 channels) thirteen of sixteen rows or columns of every MFMA are padding.  Here the TAPS go into a
// GEMM index instead, gathered per lane:
//   forward / data gradient, nt * Ci <= 32:   out^T[co][pixel] = sum_k W[k = (t, ci)][co] * in[pixel (+) t][ci]
//       B operand lane (r = pixel, q): k = 4 s + q -> its own (tap, channel): one dword gather per k-step of 4; 7 MFMAs per 16
//       pixels for 3 x 3 x 3 instead of 9 + an LDS halo tile; output as one 4-vector store per lane.
//   weight gradient, nt * Ci <= 32 (MODE 0):  dW[(t, ci)][co] = sum_pix dpre[pix][co] * in[pix (+) t][ci]
//       A = dpre [co][pixel] (16 channels contiguous; ELU', in-place store and bias sum fused as in k_conv16_wgrad), B = the
//       gathered input, N = (t, ci) in one or two 16-column blocks: 2 MFMAs per 4 pixels instead of 9.
//   weight gradient, Ci = 16, nt * Co <= 32, stride 1 (MODE 1):  dW[ci][(t, co)] = sum_pix' in[pix'][ci] * dpre[pix' (-) t][co]
//       (the sum re-indexed over INPUT pixels, so the 16-channel operand is tap-independent): A = in, B = the gathered dpre
//       (ELU' applied by a separate pass over the 3-channel tensor).
// =====================================================================================================================
template <typename T, int KS>
__global__ __launch_bounds__(256) void k_convS_fwd(svgp_conv_desc d, int ntask, int nseg, int RW, const T* __restrict__ in,
                                                   const T* __restrict__ w, const T* __restrict__ bias, T* __restrict__ out) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    int b = blockIdx.x;
    const int G = (int)gridDim.x;
    { const int per8 = G >> 3; if ((per8 << 3) == G) b = (b & 7) * per8 + (b >> 3); }
    const int per = (ntask + G - 1) / G, t_beg = b * per, t_end = min(ntask, t_beg + per);
    const int co_a = DirT<T>::corow(r), co_c = min(co_a, d.Co - 1), KT = d.nt * d.Ci;
    // per k-step: this lane's (tap, channel) and the weight it multiplies
    T wr[KS];
    int koy[KS], kox[KS], kci[KS];
    bool kok[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + q, kc = min(k, KT - 1), t = kc / d.Ci, ci = kc - t * d.Ci;
        kok[s] = k < KT;
        koy[s] = d.oy[t]; kox[s] = d.ox[t]; kci[s] = ci;
        const T v = w[d.woff[t] + ci * d.Co + co_c];
        wr[s] = (kok[s] && co_a < d.Co) ? v : T(0);
    }
    v4 bv = {0, 0, 0, 0};
    if (d.act)
#pragma unroll
        for (int g = 0; g < 4; ++g) { const T v = bias[min(4 * q + g, d.Co - 1)]; bv[g] = (4 * q + g < d.Co) ? v : T(0); }
    const T* zp = reinterpret_cast<const T*>(g_conv_zero);
    const int strips = (d.Hs + 4 * RW - 1) / (4 * RW);
    for (int task = t_beg; task < t_end; ++task) {
        const int xs = task % nseg, st = (task / nseg) % strips, n = task / (nseg * strips);
        const int px = xs * 16 + r;
        const T* inn = in + (size_t)n * d.Hi * d.Wi * d.Ci;
        const int ya = (st * 4 + wave) * RW, yb = min(d.Hs, ya + RW);
        const bool vx = px < d.Ws;
        for (int y = ya; y < yb; ++y) {
            T f[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int yi = y * d.sy + koy[s], xi = px * d.sx + kox[s];
                const bool ok = kok[s] && vx && (unsigned)yi < (unsigned)d.Hi && (unsigned)xi < (unsigned)d.Wi;
                f[s] = *(ok ? inn + ((size_t)yi * d.Wi + xi) * d.Ci + kci[s] : zp);
            }
            typename MF::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s & 1) acc1 = MF::mma(wr[s], f[s], acc1);
                else acc0 = MF::mma(wr[s], f[s], acc0);
            }
            if (vx) {
                v4 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    T e = acc0[g] + acc1[g];
                    if (d.act) e += bv[g];
                    if (d.act == 1) e = e > 0 ? e : (T)(conv_exp(e) - T(1));
                    v[g] = e;
                }
                T* o = out + (((size_t)n * d.Ho + (y * d.osy + d.ooy)) * d.Wo + (px * d.osx + d.oox)) * d.Co;
                if (d.Co == 16) {
                    *reinterpret_cast<v4*>(o + 4 * q) = v;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (4 * q + g < d.Co) o[4 * q + g] = v[g];
                }
            }
        }
    }
}

// Forward / data gradient of the thin layers with CI = 3 INPUT channels (first encoder / representation layers forward, the last
// decoder layer's data gradient) on the ring of k_convS_wgrad_ring instead of k_convS_fwd's per-lane global gathers:
//   out^T[co][pixel] = sum_(k = (t, ci)) W[k][co] in[pixel (+) t][ci],  K = NT CI (27 -> 7 k-steps of 4)
// B operand lane (r = pixel, q): k = 4 s + q -> its own (tap, channel), read from the wave-private ring of NR input rows (a row =
// the contiguous run of HW x CI values, one dword request per lane per new row, requested THREE output rows ahead in three
// alternating register sets -- a row is only 7 MFMAs long); D: one 4-vector store per lane.  Rows in groups of three, branch-free
// (rows past the end of a short last block are computed from repeated input rows and stored to the trash slots).
template <typename T, int NR, int NC, int S, int CI, bool FULL>       // FULL: 16 output channels
__global__ __launch_bounds__(256) void k_convS_fwd_ring(svgp_conv_desc d, int ntask, int nseg, int nrb, int RW,
                                                        const T* __restrict__ in, const T* __restrict__ w,
                                                        const T* __restrict__ bias, T* __restrict__ out) {
    typedef SvgpMfma<T> MF;
    typedef T v4 __attribute__((ext_vector_type(4)));
    constexpr int NT = NR * NC, KT = NT * CI, KS = (KT + 3) / 4, KEEP = NR - S;
    constexpr int HW = 15 * S + NC, RUN = HW * CI, NI = (RUN + 63) / 64, ROWE = S == 1 ? 84 : 96;
    static_assert(S <= NR && RUN <= ROWE, "ring row");
    __shared__ T smem[4 * NR * ROWE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    T* ring = smem + wave * (NR * ROWE);
    const int oy0 = d.oy[0], ox0 = d.ox[0], Hi = d.Hi, Wi = d.Wi, Ho = d.Ho, Wo = d.Wo, Hs = d.Hs, Ws = d.Ws, Co = d.Co, act = d.act,
              osy = d.osy, osx = d.osx, ooy = d.ooy, oox = d.oox;
    // per k-step: this lane's (tap row, element offset in a ring row) and the weight it multiplies
    const int co_a = DirT<T>::corow(r), co_c = min(co_a, Co - 1);
    T wr[KS];
    int kkr[KS], kofs[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + q, kc_ = min(k, KT - 1), t = kc_ / CI, ci = kc_ - t * CI, kr = t / NC, kc = t - kr * NC;
        kkr[s] = kr;
        kofs[s] = (r * S + kc) * CI + ci;
        const T v = w[d.woff[t] + ci * Co + co_c];
        wr[s] = (k < KT && co_a < Co) ? v : T(0);
    }
    v4 bv = {0, 0, 0, 0};
    if (act)
#pragma unroll
        for (int g = 0; g < 4; ++g) { const T v = bias[min(4 * q + g, Co - 1)]; bv[g] = (4 * q + g < Co) ? v : T(0); }
    int b = blockIdx.x;
    const int G = (int)gridDim.x;
    { const int per8 = G >> 3; if ((per8 << 3) == G) b = (b & 7) * per8 + (b >> 3); }
    for (int task = b * 4 + wave; task < ntask; task += G * 4) {
        const int xs = task % nseg, rb = (task / nseg) % nrb, n = task / (nseg * nrb);
        const T* inn = in + (size_t)n * Hi * Wi * CI;
        const int ya = rb * RW, yb = min(Hs, ya + RW), xf = xs * 16, X0 = xf * S + ox0;
        unsigned goff[NI], woff[NI];
        bool gok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = min(lane + 64 * i, RUN - 1), p = e / CI, gx = X0 + p;
            gok[i] = (unsigned)gx < (unsigned)Wi;
            goff[i] = (unsigned)(min(max(gx, 0), Wi - 1) * CI + (e - p * CI));
            woff[i] = (unsigned)e;
        }
        // thin rows are addressed by the output row they complete: row set `yy` = the S new input rows of output row yy
        auto rowc = [&](int yy) -> int { return min(yy, yb - 1) * S + oy0 + KEEP; };      // (clamped: past the block, re-read)
        auto gload = [&](int gy, int i) -> T { return (inn + (size_t)min(max(gy, 0), Hi - 1) * Wi * CI)[goff[i]]; };
        auto lwrite = [&](int slot, int i, T v, bool rowok) { ring[slot * ROWE + woff[i]] = (rowok && gok[i]) ? v : T(0); };
        {
            T v[NR][NI];
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) v[k][i] = gload(ya * S + oy0 + k, i);
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) lwrite(k, i, v[k][i], (unsigned)(ya * S + oy0 + k) < (unsigned)Hi);
        }
        int base = 0;
        const bool vx = xf + r < Ws;
        T* trash = reinterpret_cast<T*>(svgp_conv_trash) + 4 * lane;
        T* on = out + (((size_t)n * Ho + ooy) * Wo + ((xf + r) * osx + oox)) * Co + 4 * q;
        T Q0[S][NI], Q1[S][NI], Q2[S][NI];
#pragma unroll
        for (int k = 0; k < S; ++k)
#pragma unroll
            for (int i = 0; i < NI; ++i) { Q0[k][i] = gload(rowc(ya + 1) + k, i); Q1[k][i] = gload(rowc(ya + 2) + k, i); Q2[k][i] = gload(rowc(ya + 3) + k, i); }
        // one output row y from the ring; then Q (the new input rows of output row y + 1) into the ring and its refill for row y + 4
        auto step = [&](int y, T (&Q)[S][NI]) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            T f[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) { int sl = base + kkr[s]; if (sl >= NR) sl -= NR; f[s] = ring[sl * ROWE + kofs[s]]; }
            typename MF::acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s & 1) acc1 = MF::mma(wr[s], f[s], acc1);
                else acc0 = MF::mma(wr[s], f[s], acc0);
            }
            v4 v;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                T e = acc0[g] + acc1[g];
                if (act) e += bv[g];
                if (act == 1) e = e > 0 ? e : (T)(conv_exp(e) - T(1));
                v[g] = e;
            }
            const bool ok = vx && y < yb;
            if (FULL) {
                *reinterpret_cast<v4*>(ok ? on + (size_t)y * osy * Wo * Co : trash) = v;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) *((ok && 4 * q + g < Co) ? on + (size_t)y * osy * Wo * Co + g : trash + g) = v[g];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int g0 = rowc(y + 1);
#pragma unroll
            for (int k = 0; k < S; ++k) {
                int sl = base + k; if (sl >= NR) sl -= NR;
#pragma unroll
                for (int i = 0; i < NI; ++i) lwrite(sl, i, Q[k][i], (unsigned)(g0 + k) < (unsigned)Hi);
            }
            base += S; if (base >= NR) base -= NR;
            const int g4 = rowc(y + 4);
#pragma unroll
            for (int k = 0; k < S; ++k)
#pragma unroll
                for (int i = 0; i < NI; ++i) Q[k][i] = gload(g4 + k, i);
        };
        for (int y = ya; y < yb; y += 3) {
            step(y, Q0);
            step(y + 1, Q1);
            step(y + 2, Q2);
        }
    }
}

template <typename T, int NB, int MODE>
__global__ __launch_bounds__(256) void k_convS_wgrad(svgp_conv_desc d, int nwg, int RW, const T* __restrict__ in,
                                                     const T* __restrict__ outv, T* __restrict__ dout, T* __restrict__ part,
                                                     int part_stride, T* __restrict__ part_b) {
    typedef SvgpMfma<T> MF;
    __shared__ T red[1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    // MODE 0: rows i = co (A = dpre), columns j = (t, ci) (B = gathered input), iteration over OUTPUT pixels
    // MODE 1: rows i = ci (A = input), columns j = (t, co) (B = gathered dpre), iteration over INPUT pixels (stride 1)
    const int CB = MODE == 0 ? d.Ci : d.Co, JT = d.nt * CB;
    const int Hit = MODE == 0 ? d.Hs : d.Hi, Wit = MODE == 0 ? d.Ws : d.Wi;
    int joy[NB], jox[NB], jc[NB];
    bool jok[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = 16 * nb + r, jcl = min(j, JT - 1), t = jcl / CB;
        jok[nb] = j < JT;
        joy[nb] = d.oy[t]; jox[nb] = d.ox[t]; jc[nb] = jcl - t * CB;
    }
    typename MF::acc_t acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = typename MF::acc_t{0, 0, 0, 0};
    T bsum = 0;
    const T* zp = reinterpret_cast<const T*>(g_conv_zero);
    const int nseg = (Wit + 15) / 16, nrb = (Hit + RW - 1) / RW, ntask = d.n * nrb * nseg;
    for (int task = blockIdx.x * 4 + wave; task < ntask; task += nwg * 4) {
        const int xs = task % nseg, rb = (task / nseg) % nrb, n = task / (nseg * nrb);
        const T* inn = in + (size_t)n * d.Hi * d.Wi * d.Ci;
        T* dn = dout + (size_t)n * d.Ho * d.Wo * d.Co;
        const T* on = outv ? outv + (size_t)n * d.Ho * d.Wo * d.Co : nullptr;
        const int ya = rb * RW, yb = min(Hit, ya + RW), xf = xs * 16;
        for (int y = ya; y < yb; ++y) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int x = xf + 4 * s + q;
                const bool vx = x < Wit;
                T av;
                if (MODE == 0) {
                    // A = dpre[pixel (y, x)][co = r]: ELU' fused, written back in place, bias sum
                    const bool ok = vx && r < d.Co;
                    const size_t o = ((size_t)(y * d.osy + d.ooy) * d.Wo + (min(x, Wit - 1) * d.osx + d.oox)) * d.Co + min(r, d.Co - 1);
                    T dv = *(ok ? dn + o : zp);
                    if (on) { const T ov = on[o]; dv *= (ov > 0 ? T(1) : ov + T(1)); if (ok) dn[o] = dv; }
                    bsum += dv;
                    av = dv;
                } else {
                    av = *(vx ? inn + ((size_t)y * d.Wi + x) * d.Ci + r : zp);           // A = in[pixel'][ci = r], Ci == 16
                }
                T bvv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (MODE == 0) {
                        const int yi = y * d.sy + joy[nb], xi = x * d.sx + jox[nb];
                        const bool ok = jok[nb] && vx && (unsigned)yi < (unsigned)d.Hi && (unsigned)xi < (unsigned)d.Wi;
                        bvv[nb] = *(ok ? inn + ((size_t)yi * d.Wi + xi) * d.Ci + jc[nb] : zp);
                    } else {
                        const int yo = y - joy[nb], xo = x - jox[nb];                    // output pixel whose tap t reads (y, x)
                        const bool ok = jok[nb] && vx && (unsigned)yo < (unsigned)d.Hs && (unsigned)xo < (unsigned)d.Ws;
                        bvv[nb] = *(ok ? dn + ((size_t)(yo * d.osy + d.ooy) * d.Wo + (xo * d.osx + d.oox)) * d.Co + jc[nb] : zp);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = MF::mma(av, bvv[nb], acc[nb]);
            }
        }
    }
    // cross-wave combine (fixed order); D: column j = r (+ 16 nb), row i = MF::row(q, g)
    T* po = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) red[(wave * 64 + lane) * 4 + g] = acc[nb][g];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const T v = red[lane * 4 + g] + red[(64 + lane) * 4 + g] + red[(128 + lane) * 4 + g] + red[(192 + lane) * 4 + g];
                const int i = MF::row(q, g), j = 16 * nb + r;
                if (j < JT) {
                    const int t = j / CB, c = j - t * CB;
                    const int ci = MODE == 0 ? c : i, co = MODE == 0 ? i : c;
                    if (ci < d.Ci && co < d.Co) po[d.woff[t] + ci * d.Co + co] = v;
                }
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < 16) {
            T v = 0;
            for (int k = 0; k < 16; ++k) v += red[k * 16 + threadIdx.x];
            part_b[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
        }
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

// ---- host side of the direct kernels
static bool conv16_enabled() {
    static const int on = [] { const char* e = getenv("SVGP_CONV_DIRECT"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}
static bool conv16_nt_ok(int nt) { return nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 6 || nt == 9 || nt == 16; }
static bool conv16_direct_ok(const svgp_conv_desc* d, int ncls, bool same_nt) {
    if (!conv16_enabled()) return false;
    for (int c = 0; c < ncls; ++c) {
        if (d[c].Ci != 16 || !conv16_nt_ok(d[c].nt)) return false;
        if (same_nt && (d[c].nt != d[0].nt || (d[c].nt != 4 && d[c].nt != 9))) return false;
    }
    return true;
}
static int conv16_rows(const svgp_conv_desc& d) {            // output rows per wave (rolling kernels) / per strip / 4 (others)
    static const int forced = [] { const char* e = getenv("SVGP_CONV_ROWS"); return e ? atoi(e) : 0; }();
    int rw = forced > 0 ? forced : 8;
    const int q4 = (d.Hs + 3) / 4;
    return rw < q4 ? rw : (q4 < 1 ? 1 : q4);
}
// Tap table = full NR x NC grid with consecutive row offsets?  `g` receives the descriptor with its taps in grid order
// (row-major; g.oy[k * NC] = row offset k, g.ox[c] = column offset c).
static bool conv16_grid(const svgp_conv_desc& d, svgp_conv_desc* g, int* NR, int* NC) {
    int ys[16], xs[16], ny = 0, nx = 0;
    for (int t = 0; t < d.nt; ++t) {
        int k = 0;
        while (k < ny && ys[k] != d.oy[t]) ++k;
        if (k == ny) ys[ny++] = d.oy[t];
        k = 0;
        while (k < nx && xs[k] != d.ox[t]) ++k;
        if (k == nx) xs[nx++] = d.ox[t];
    }
    if (ny * nx != d.nt) return false;
    for (int a = 0; a < ny; ++a) for (int b2 = a + 1; b2 < ny; ++b2) if (ys[b2] < ys[a]) { int v = ys[a]; ys[a] = ys[b2]; ys[b2] = v; }
    for (int a = 0; a < nx; ++a) for (int b2 = a + 1; b2 < nx; ++b2) if (xs[b2] < xs[a]) { int v = xs[a]; xs[a] = xs[b2]; xs[b2] = v; }
    for (int k = 1; k < ny; ++k) if (ys[k] != ys[0] + k) return false;
    *g = d;
    for (int k = 0; k < ny; ++k)
        for (int c = 0; c < nx; ++c) {
            int t = 0;
            while (t < d.nt && !(d.oy[t] == ys[k] && d.ox[t] == xs[c])) ++t;
            if (t == d.nt) return false;
            g->oy[k * nx + c] = ys[k]; g->ox[k * nx + c] = xs[c]; g->woff[k * nx + c] = d.woff[t];
        }
    *NR = ny; *NC = nx;
    return true;
}

template <typename T>
static int conv16_fwd_launch(const svgp_conv_desc* d, int ncls, const T* in, const T* w, const T* bias, T* out,
                             void* stream) {
    static const int roll_on = [] { const char* e = getenv("SVGP_CONV_ROLL"); return (e && e[0] == '0') ? 0 : 1; }();
    for (int c = 0; c < ncls; ++c) {
        const svgp_conv_desc& dc = d[c];
        SVGP_REQUIRE(!dc.act || bias, SVGP_ERR_INVALID, "bias is NULL but act != 0");
        const int RW = conv16_rows(dc), R = 4 * RW, strips = (dc.Hs + R - 1) / R, nseg = (dc.Ws + 15) / 16;
        const int ntask = dc.n * strips * nseg;
        const dim3 grid((unsigned)ntask);
        static const int pgrid = [] { const char* e = getenv("SVGP_CONV_GRID"); return e ? atoi(e) : 1024; }();
        const dim3 grid_p((unsigned)(ntask < pgrid ? ntask : pgrid));        // persistent form: <= 4 workgroups per CU
        svgp_conv_desc g;
        int NR = 0, NC = 0;
        bool done = false;
        if (roll_on && dc.Ws >= 16 && conv16_grid(dc, &g, &NR, &NC)) {
#define C16R(NR_, NC_, SH_, PF_)                                                                                              \
            if (!done && NR == NR_ && NC == NC_ && dc.sy == SH_) {                                                          \
                if (dc.Ws % 16 == 0 && dc.Co == 16)                                                                         \
                    hipLaunchKernelGGL((k_conv16_fwd_roll<T, NR_, NC_, SH_, PF_, true>), grid_p, dim3(256), 0,                \
                                       (hipStream_t)stream, g, strips, RW, nseg, ntask, in, w, bias, out);                  \
                else                                                                                                        \
                    hipLaunchKernelGGL((k_conv16_fwd_roll<T, NR_, NC_, SH_, PF_, false>), grid_p, dim3(256), 0,               \
                                       (hipStream_t)stream, g, strips, RW, nseg, ntask, in, w, bias, out);                  \
                done = true;                                                                                                \
            }
            C16R(3, 3, 1, true) C16R(3, 3, 2, true) C16R(2, 2, 1, true) C16R(2, 2, 2, true) C16R(4, 4, 2, false)
            C16R(1, 1, 1, true) C16R(1, 2, 1, true) C16R(2, 1, 1, true)
#undef C16R
        }
        if (!done) {
#define C16F(NT_) hipLaunchKernelGGL((k_conv16_fwd<T, NT_>), grid, dim3(256), 0, (hipStream_t)stream, dc, strips, R, nseg, in, w, \
                                     bias, out)
            switch (dc.nt) {
            case 1: C16F(1); break;
            case 2: C16F(2); break;
            case 3: C16F(3); break;
            case 4: C16F(4); break;
            case 6: C16F(6); break;
            case 9: C16F(9); break;
            default: C16F(16); break;
            }
#undef C16F
        }
        SVGP_LAUNCH_CHECK();
    }
    return SVGP_OK;
}

// ---- thin layers (k_convS_*)
static bool convS_fwd_ok(const svgp_conv_desc* d, int ncls) {
    if (!conv16_enabled()) return false;
    for (int c = 0; c < ncls; ++c)
        if (d[c].Ci >= 16 || d[c].nt * d[c].Ci > 32) return false;
    return true;
}
template <typename T>
static int convS_fwd_launch(const svgp_conv_desc* d, int ncls, const T* in, const T* w, const T* bias, T* out, void* stream) {
    static const int fring_on = [] { const char* e = getenv("SVGP_CONV_FWD_RING"); return (e && e[0] == '0') ? 0 : 1; }();
    for (int c = 0; c < ncls; ++c) {
        const svgp_conv_desc& dc = d[c];
        SVGP_REQUIRE(!dc.act || bias, SVGP_ERR_INVALID, "bias is NULL but act != 0");
        // 3 input channels, width a multiple of 16, full grid of consecutive offsets: k_convS_fwd_ring
        if (fring_on && dc.Ci == 3 && dc.Ws % 16 == 0 && dc.sy == dc.sx) {
            svgp_conv_desc g;
            int NR = 0, NC = 0;
            bool ok = conv16_grid(dc, &g, &NR, &NC);
            for (int x = 1; ok && x < NC; ++x) ok = g.ox[x] == g.ox[0] + x;
            const int S = dc.sy;
            ok = ok && ((NR == 3 && NC == 3 && S == 1) || (NR == 2 && NC == 2 && S == 2));
            if (ok) {
                int RW = 12;
                if (RW > (g.Hs + 2) / 3 * 3) RW = (g.Hs + 2) / 3 * 3;
                const int nrb = (g.Hs + RW - 1) / RW, nseg = g.Ws / 16, ntask = g.n * nrb * nseg, nwg = (ntask + 3) / 4;
                const dim3 grid((unsigned)(nwg < 1024 ? nwg : 1024));
#define CSFR(NR_, NC_, S_, FULL_) hipLaunchKernelGGL((k_convS_fwd_ring<T, NR_, NC_, S_, 3, FULL_>), grid, dim3(256), 0,         \
                                                     (hipStream_t)stream, g, ntask, nseg, nrb, RW, in, w, bias, out)
                if (S == 1) { if (g.Co == 16) CSFR(3, 3, 1, true); else CSFR(3, 3, 1, false); }
                else { if (g.Co == 16) CSFR(2, 2, 2, true); else CSFR(2, 2, 2, false); }
#undef CSFR
                SVGP_LAUNCH_CHECK();
                continue;
            }
        }
        const int RW = conv16_rows(dc), strips = (dc.Hs + 4 * RW - 1) / (4 * RW), nseg = (dc.Ws + 15) / 16;
        const int ntask = dc.n * strips * nseg, KS = (dc.nt * dc.Ci + 3) / 4;
        const dim3 grid((unsigned)(ntask < 2048 ? ntask : 2048));
#define CSF(KS_) hipLaunchKernelGGL((k_convS_fwd<T, KS_>), grid, dim3(256), 0, (hipStream_t)stream, dc, ntask, nseg, RW, in, w, bias, out)
        switch (KS) {
        case 1: CSF(1); break;
        case 2: CSF(2); break;
        case 3: CSF(3); break;
        case 4: CSF(4); break;
        case 5: CSF(5); break;
        case 6: CSF(6); break;
        case 7: CSF(7); break;
        default: CSF(8); break;
        }
#undef CSF
        SVGP_LAUNCH_CHECK();
    }
    return SVGP_OK;
}

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
    // 16 -> 3 channels, 3 x 3 grid of consecutive offsets, stride 1, plain output placement: the taps in the MFMA row index
    {
        static const int thin_on = [] { const char* e = getenv("SVGP_CONV_THIN_FWD"); return (e && e[0] == '0') ? 0 : 1; }();
        svgp_conv_desc g;
        int NR = 0, NC = 0;
        bool ok = thin_on && conv16_enabled() && ncls == 1 && d[0].Ci == 16 && d[0].Co == 3 && d[0].sy == 1 && d[0].sx == 1 &&
                  d[0].osy == 1 && d[0].osx == 1 && d[0].ooy == 0 && d[0].oox == 0 && conv16_grid(d[0], &g, &NR, &NC) && NR == 3 &&
                  NC == 3;
        for (int x = 1; ok && x < NC; ++x) ok = g.ox[x] == g.ox[0] + x;
        if (ok) {
            static const int rows_env = [] { const char* e = getenv("SVGP_CONV_THIN_ROWS"); return e ? atoi(e) : 0; }();
            int RW = rows_env > 0 ? (rows_env + 2) / 3 * 3 : 12;              // whole ring periods
            if (RW > (g.Hs + 2) / 3 * 3) RW = (g.Hs + 2) / 3 * 3;
            const int nrb = (g.Hs + RW - 1) / RW, nseg = (g.Ws + 13) / 14, ntask = g.n * nrb * nseg;   // tasks of one wave each
            const int nwg = (ntask + 3) / 4;
            hipLaunchKernelGGL((k_conv16_thin_fwd<T, 3, 3, 3>), dim3((unsigned)(nwg < 1024 ? nwg : 1024)), dim3(256), 0,
                               (hipStream_t)stream, g, ntask, nseg, nrb, RW, in, w, bias, out);
            SVGP_LAUNCH_CHECK();
            return SVGP_OK;
        }
    }
    if (conv16_direct_ok(d, ncls, false)) return conv16_fwd_launch<T>(d, ncls, in, w, bias, out, stream);
    if (convS_fwd_ok(d, ncls)) return convS_fwd_launch<T>(d, ncls, in, w, bias, out, stream);
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
    return sum_partials<T>(nwg, part_stride, part_stride, (const T*)part, dw, accumulate, stream);
}

// One pass for the reverse of a layer's bias / activation / weights: dpre = dout * elu'(out) in place (out == NULL: dpre = dout),
// db = column sums of dpre, dW_t = sum in * dpre.  16 input channels with 4 or 9 taps per class: the fused direct kernel;
// otherwise the separate kernels in sequence.  part_b: (1024, 16) scratch, part: (nwg, part_stride) scratch.
template <typename T> static int elu_bwd_bias_impl(long long, int, const T*, T*, T*, T*, void*);
template <typename T>
static int conv_wgrad_fused_impl(const svgp_conv_desc* d, int ncls, const T* in, const T* outv, T* dout, T* part, T* part_b,
                                 int nwg, int part_stride, T* dw, T* db, void* stream) {
    int rc = check_desc(d, ncls);
    if (rc) return rc;
    SVGP_REQUIRE(in && dout && part && part_b && dw && db && nwg >= 1 && part_stride >= 1, SVGP_ERR_INVALID, "bad argument");
    // thin layers: the taps in a GEMM index (k_convS_wgrad)
    {
        bool m0 = conv16_enabled(), m1 = conv16_enabled() && ncls == 1;
        for (int c = 0; c < ncls; ++c) {
            if (d[c].Ci >= 16 || d[c].nt * d[c].Ci > 32) m0 = false;
            if (d[c].Ci != 16 || d[c].nt * d[c].Co > 32 || d[c].sy != 1 || d[c].sx != 1 || d[c].osy != 1 || d[c].osx != 1 ||
                d[c].ooy || d[c].oox || d[c].Hs != d[c].Ho || d[c].Ws != d[c].Wo) m1 = false;
        }
        // 3 -> 16 channels, one class, width a multiple of 16, full grid of consecutive offsets: k_convS_wgrad_ring
        static const int ring_on = [] { const char* e = getenv("SVGP_CONV_WGRAD_RING"); return (e && e[0] == '0') ? 0 : 1; }();
        if (m0 && ring_on && ncls == 1 && d[0].Ci == 3 && d[0].Co == 16 && d[0].Ws % 16 == 0 && d[0].sy == d[0].sx) {
            svgp_conv_desc g;
            int NR = 0, NC = 0;
            bool ok = conv16_grid(d[0], &g, &NR, &NC);
            for (int x = 1; ok && x < NC; ++x) ok = g.ox[x] == g.ox[0] + x;
            const int S = d[0].sy;
            ok = ok && ((NR == 3 && NC == 3 && S == 1) || (NR == 2 && NC == 2 && S == 2));
            if (ok) {
                const int nw = nwg > 1024 ? 1024 : nwg, RW = conv16_rows(d[0]);
#define CSR(NR_, NC_, S_)                                                                                                   \
                if (NR == NR_ && NC == NC_ && S == S_) {                                                                    \
                    if (outv) hipLaunchKernelGGL((k_convS_wgrad_ring<T, NR_, NC_, S_, 3, true, 0>), dim3(nw), dim3(256), 0,    \
                                                 (hipStream_t)stream, g, nw, RW, in, outv, dout, part, part_stride, part_b);\
                    else hipLaunchKernelGGL((k_convS_wgrad_ring<T, NR_, NC_, S_, 3, false, 0>), dim3(nw), dim3(256), 0,        \
                                            (hipStream_t)stream, g, nw, RW, in, outv, dout, part, part_stride, part_b);    \
                }
                CSR(3, 3, 1) CSR(2, 2, 2)
#undef CSR
                SVGP_LAUNCH_CHECK();
                rc = sum_partials<T>(nw, part_stride, part_stride, (const T*)part, dw, 0, stream);
                if (rc) return rc;
                return sum_partials<T>(nw, 16, 16, (const T*)part_b, db, 0, stream);
            }
        }
        if (m0 || m1) {
            int nw = nwg > 1024 ? 1024 : nwg;
            if (m1) {       // ELU' + bias sums by their own pass over the thin dout; the kernel then gathers dpre
                rc = elu_bwd_bias_impl<T>((long long)d[0].n * d[0].Ho * d[0].Wo, d[0].Co, outv, dout, part_b, db, stream);
                if (rc) return rc;
                // 16 -> 3 channels, 3 x 3 grid of consecutive offsets, input width a multiple of 16: k_convS_wgrad_ring, MODE 1
                svgp_conv_desc g;
                int NR = 0, NC = 0;
                bool ok = ring_on && d[0].Co == 3 && d[0].Wi % 16 == 0 && conv16_grid(d[0], &g, &NR, &NC) && NR == 3 && NC == 3;
                for (int x = 1; ok && x < NC; ++x) ok = g.ox[x] == g.ox[0] + x;
                if (ok) {
                    svgp_conv_desc gi = g;
                    gi.Hs = g.Hi;                      // rows per wave from the iteration space of this mode (input pixels)
                    hipLaunchKernelGGL((k_convS_wgrad_ring<T, 3, 3, 1, 3, false, 1>), dim3(nw), dim3(256), 0, (hipStream_t)stream, g,
                                       nw, conv16_rows(gi), in, (const T*)nullptr, dout, part, part_stride, part_b);
                    SVGP_LAUNCH_CHECK();
                    return sum_partials<T>(nw, part_stride, part_stride, (const T*)part, dw, 0, stream);
                }
            } else if (nw * ncls > 1024) {
                nw = 1024 / ncls;
            }
            SVGP_CHECK_HIP(hipMemsetAsync(part, 0, (size_t)nw * part_stride * sizeof(T), (hipStream_t)stream));
            for (int c = 0; c < ncls; ++c) {
                const svgp_conv_desc& dc = d[c];
                const int RW = conv16_rows(dc), NB = ((m0 ? dc.nt * dc.Ci : dc.nt * dc.Co) + 15) / 16;
                T* pb = part_b + (size_t)c * nw * 16;
#define CSW(NB_, MODE_) hipLaunchKernelGGL((k_convS_wgrad<T, NB_, MODE_>), dim3(nw), dim3(256), 0, (hipStream_t)stream, dc, nw, RW, in,  \
                                           m0 ? outv : (const T*)nullptr, dout, part, part_stride, pb)
                if (m0) { if (NB == 1) CSW(1, 0); else CSW(2, 0); }
                else { if (NB == 1) CSW(1, 1); else CSW(2, 1); }
#undef CSW
                SVGP_LAUNCH_CHECK();
            }
            rc = sum_partials<T>(nw, part_stride, part_stride, (const T*)part, dw, 0, stream);
            if (rc) return rc;
            if (m0) return sum_partials<T>(nw * ncls, d[0].Co, 16, (const T*)part_b, db, 0, stream);
            return SVGP_OK;
        }
    }
    if (!conv16_direct_ok(d, ncls, true)) {
        rc = elu_bwd_bias_impl<T>((long long)d[0].n * d[0].Ho * d[0].Wo, d[0].Co, outv, dout, part_b, db, stream);
        if (rc) return rc;
        return conv_taps_wgrad_impl<T>(d, ncls, in, dout, part, nwg, part_stride, dw, 0, stream);
    }
    static const int roll_on = [] { const char* e = getenv("SVGP_CONV_ROLL"); return (e && e[0] == '0') ? 0 : 1; }();
    static const int grid_on = [] { const char* e = getenv("SVGP_CONV_WGRAD_GRID"); return (e && e[0] == '0') ? 0 : 1; }();
    ConvLaunch L;
    L.ncls = ncls;
    int nwg_c = nwg / ncls;
    if (nwg_c < 1) nwg_c = 1;
    if (nwg_c * ncls > 1024) nwg_c = 1024 / ncls;
    // 16 -> 16 channels, width a multiple of 16, every class the same full grid of consecutive offsets: k_conv16_wgrad_grid
    if (roll_on && grid_on) {
        ConvLaunch G;
        G.ncls = ncls;
        int NR = 0, NC = 0;
        bool ok = true;
        for (int c = 0; c < ncls && ok; ++c) {
            int nr = 0, nc = 0;
            ok = d[c].Co == 16 && d[c].Ws % 16 == 0 && d[c].sy == d[0].sy && d[c].sx == d[0].sx && d[c].sy == d[c].sx &&
                 d[c].Hs == d[0].Hs && d[c].Ws == d[0].Ws && conv16_grid(d[c], &G.d[c], &nr, &nc);
            if (ok && c == 0) { NR = nr; NC = nc; }
            ok = ok && nr == NR && nc == NC;
            for (int x = 1; ok && x < nc; ++x) ok = G.d[c].ox[x] == G.d[c].ox[0] + x;
        }
        const int SY = d[0].sy;
        ok = ok && ((NR == 3 && NC == 3 && (SY == 1 || SY == 2)) || (NR == 2 && NC == 2 && (SY == 1 || SY == 2)));
        if (ok) {
            const int RW = conv16_rows(d[0]), HW = 15 * SY + NC, PS = SY == 1 ? 16 : 24;
            size_t lds = (size_t)4 * NR * HW * PS;
            if (lds < 1024) lds = 1024;
            lds *= sizeof(T);
#define C16G(NR_, NC_, S_)                                                                                                  \
            if (NR == NR_ && NC == NC_ && SY == S_) {                                                                       \
                if (outv) {                                                                                                 \
                    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv16_wgrad_grid<T, NR_, NC_, S_, S_, true>),  \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
                    hipLaunchKernelGGL((k_conv16_wgrad_grid<T, NR_, NC_, S_, S_, true>), dim3(nwg_c, ncls), dim3(256), lds,  \
                                       (hipStream_t)stream, G, nwg_c, RW, in, outv, dout, part, part_stride, part_b);      \
                } else {                                                                                                    \
                    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv16_wgrad_grid<T, NR_, NC_, S_, S_, false>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
                    hipLaunchKernelGGL((k_conv16_wgrad_grid<T, NR_, NC_, S_, S_, false>), dim3(nwg_c, ncls), dim3(256), lds, \
                                       (hipStream_t)stream, G, nwg_c, RW, in, outv, dout, part, part_stride, part_b);      \
                }                                                                                                           \
            }
            C16G(3, 3, 1) C16G(3, 3, 2) C16G(2, 2, 1) C16G(2, 2, 2)
#undef C16G
            SVGP_LAUNCH_CHECK();
            rc = sum_partials<T>(nwg_c, part_stride, part_stride, (const T*)part, dw, 0, stream);
            if (rc) return rc;
            return sum_partials<T>(nwg_c * ncls, 16, 16, (const T*)part_b, db, 0, stream);
        }
    }
    size_t lpw = 0, lpw_roll = 0;
    const int RW = conv16_rows(d[0]);
    bool roll = roll_on && d[0].Ws >= 16 && (d[0].sy == 1 || d[0].sy == 2);
    for (int c = 0; c < ncls; ++c) {
        L.d[c] = d[c];
        SVGP_REQUIRE(d[c].Co == d[0].Co && d[c].Hs == d[0].Hs && d[c].Ws == d[0].Ws && d[c].sy == d[0].sy, SVGP_ERR_INVALID,
                     "classes of one launch share Co, the row stride and the iteration space");
        int oy0 = d[c].oy[0], oy1 = oy0, ox0 = d[c].ox[0], ox1 = ox0;
        for (int t = 1; t < d[c].nt; ++t) {
            oy0 = oy0 < d[c].oy[t] ? oy0 : d[c].oy[t]; oy1 = oy1 > d[c].oy[t] ? oy1 : d[c].oy[t];
            ox0 = ox0 < d[c].ox[t] ? ox0 : d[c].ox[t]; ox1 = ox1 > d[c].ox[t] ? ox1 : d[c].ox[t];
        }
        const int segw = d[c].Ws < 16 ? d[c].Ws : 16, rpw = 16 / segw, PS = d[c].sx == 1 ? 16 : 24;
        const size_t e = (size_t)((rpw - 1) * d[c].sy + (oy1 - oy0) + 1) * ((segw - 1) * d[c].sx + (ox1 - ox0) + 1) * PS;
        lpw = e > lpw ? e : lpw;
        const int HWr = 15 * d[c].sx + (ox1 - ox0) + 1;
        if (HWr > 48) roll = false;
        const size_t er = (size_t)(oy1 - oy0 + 1) * HWr * PS;
        lpw_roll = er > lpw_roll ? er : lpw_roll;
    }
    if (roll) lpw = lpw_roll;
    lpw = (lpw + 7) & ~(size_t)7;
    size_t lds = 4 * lpw > 1024 ? 4 * lpw : 1024;
    lds *= sizeof(T);
    SVGP_REQUIRE(lds <= 160 * 1024, SVGP_ERR_UNSUPPORTED, "conv halo needs %zu bytes of LDS", lds);
    // (no zero fill of `part`: every workgroup (x, class) stores all tap blocks of its class, and the classes' tap ranges tile a
    // partial row -- tests/test_gpu_conv.py runs on NaN-filled scratch)
#define C16W(KERNEL_, ROWS_)                                                                                                \
    do {                                                                                                                    \
        SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL_),                                          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        hipLaunchKernelGGL(KERNEL_, dim3(nwg_c, ncls), dim3(256), lds, (hipStream_t)stream, L, nwg_c, ROWS_, in,            \
                           outv, dout, part, part_stride, part_b, (int)lpw);                                                \
    } while (0)
    if (roll) {
        if (d[0].nt == 9 && d[0].sy == 1) C16W((k_conv16_wgrad_roll<T, 9, 1>), RW);
        else if (d[0].nt == 9) C16W((k_conv16_wgrad_roll<T, 9, 2>), RW);
        else if (d[0].sy == 1) C16W((k_conv16_wgrad_roll<T, 4, 1>), RW);
        else C16W((k_conv16_wgrad_roll<T, 4, 2>), RW);
    } else {
        if (d[0].nt == 9) C16W((k_conv16_wgrad<T, 9>), 4 * RW); else C16W((k_conv16_wgrad<T, 4>), 4 * RW);
    }
#undef C16W
    SVGP_LAUNCH_CHECK();
    rc = sum_partials<T>(nwg_c, part_stride, part_stride, (const T*)part, dw, 0, stream);
    if (rc) return rc;
    return sum_partials<T>(nwg_c * ncls, d[0].Co, 16, (const T*)part_b, db, 0, stream);
}

// dpre = dout * elu'(out) (in place on dout; out == NULL skips the activation) and db[c] = sum dpre[.., c].
// part: (1024, C) scratch.
template <typename T>
static int elu_bwd_bias_impl(long long npix, int C, const T* out, T* dout, T* part, T* db, void* stream) {
    SVGP_REQUIRE(npix >= 1 && C >= 1 && C <= 16 && dout && part && db, SVGP_ERR_INVALID, "bad argument");
    const int nblk = 1024;     // 4 workgroups per CU keep the HBM queues full
    hipLaunchKernelGGL(k_elu_bwd_colsum<T>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, npix, C, out, dout, part);
    SVGP_LAUNCH_CHECK();
    return sum_partials<T>(nblk, C, C, (const T*)part, db, 0, stream);
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
extern "C" int svgp_conv_taps_wgrad_fused(const svgp_conv_desc* d, int ncls, const double* in, const double* out, double* dout,
                                          double* part, double* part_b, int nwg, int part_stride, double* dw, double* db,
                                          void* stream) {
    return conv_wgrad_fused_impl<double>(d, ncls, in, out, dout, part, part_b, nwg, part_stride, dw, db, stream);
}
extern "C" int svgp_conv_taps_wgrad_fused_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* out, float* dout,
                                              float* part, float* part_b, int nwg, int part_stride, float* dw, float* db,
                                              void* stream) {
    return conv_wgrad_fused_impl<float>(d, ncls, in, out, dout, part, part_b, nwg, part_stride, dw, db, stream);
}
// as svgp_conv_taps_wgrad_fused, but the closing reductions come back as job descriptors instead of being launched
template <typename T>
static int wgrad_fused_jobs(const svgp_conv_desc* d, int ncls, const T* in, const T* out, T* dout, T* part, T* part_b, int nwg,
                            int part_stride, T* dw, T* db, svgp_sum_job* jobs, int cap, int* n_jobs, void* stream) {
    SVGP_REQUIRE(jobs && n_jobs && cap >= 1, SVGP_ERR_INVALID, "bad job array");
    SumCapture& c = sum_capture();
    c.jobs = jobs; c.cap = cap; c.n = 0;
    const int rc = conv_wgrad_fused_impl<T>(d, ncls, in, out, dout, part, part_b, nwg, part_stride, dw, db, stream);
    *n_jobs = c.n;
    c.jobs = nullptr; c.cap = 0; c.n = 0;
    return rc;
}
extern "C" int svgp_conv_taps_wgrad_fused_jobs(const svgp_conv_desc* d, int ncls, const double* in, const double* out,
                                               double* dout, double* part, double* part_b, int nwg, int part_stride, double* dw,
                                               double* db, svgp_sum_job* jobs, int cap, int* n_jobs, void* stream) {
    return wgrad_fused_jobs<double>(d, ncls, in, out, dout, part, part_b, nwg, part_stride, dw, db, jobs, cap, n_jobs, stream);
}
extern "C" int svgp_conv_taps_wgrad_fused_jobs_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* out,
                                                   float* dout, float* part, float* part_b, int nwg, int part_stride, float* dw,
                                                   float* db, svgp_sum_job* jobs, int cap, int* n_jobs, void* stream) {
    return wgrad_fused_jobs<float>(d, ncls, in, out, dout, part, part_b, nwg, part_stride, dw, db, jobs, cap, n_jobs, stream);
}
extern "C" int svgp_sum_partials_multi(const svgp_sum_job* jobs, int n, void* stream) {
    return sum_partials_multi_impl<double>(jobs, n, stream);
}
extern "C" int svgp_sum_partials_multi_f32(const svgp_sum_job* jobs, int n, void* stream) {
    return sum_partials_multi_impl<float>(jobs, n, stream);
}
// (nt, A, B) -> (nt, B, A): the transposed tap weights of a data gradient (conv.py weights_bwd), and element-type casts
template <typename T>
__global__ void k_transpose_taps(int A, int B, long long tot, const T* __restrict__ w, T* __restrict__ wt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const int ab = A * B, t = (int)(i / ab), o = (int)(i % ab), bb = o / A, aa = o % A;      // wt[t][bb][aa] = w[t][aa][bb]
    wt[i] = w[(size_t)t * ab + (size_t)aa * B + bb];
}
template <typename TI, typename TO>
__global__ void k_cast(long long n, const TI* __restrict__ x, TO* __restrict__ y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (TO)x[i];
}
#define SIMPLE_LAUNCH(KERNEL_, N_, ...)                                                                              \
    do {                                                                                                             \
        SVGP_REQUIRE((N_) >= 1, SVGP_ERR_INVALID, "bad argument");                                                   \
        hipLaunchKernelGGL(KERNEL_, dim3((unsigned)(((N_) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        SVGP_LAUNCH_CHECK();                                                                                         \
        return SVGP_OK;                                                                                              \
    } while (0)
extern "C" int svgp_transpose_taps(int nt, int A, int B, const double* w, double* wt, void* stream) {
    SVGP_REQUIRE(w && wt && nt >= 1 && A >= 1 && B >= 1, SVGP_ERR_INVALID, "bad argument");
    SIMPLE_LAUNCH(k_transpose_taps<double>, (long long)nt * A * B, A, B, (long long)nt * A * B, w, wt);
}
extern "C" int svgp_transpose_taps_f32(int nt, int A, int B, const float* w, float* wt, void* stream) {
    SVGP_REQUIRE(w && wt && nt >= 1 && A >= 1 && B >= 1, SVGP_ERR_INVALID, "bad argument");
    SIMPLE_LAUNCH(k_transpose_taps<float>, (long long)nt * A * B, A, B, (long long)nt * A * B, w, wt);
}
extern "C" int svgp_cast_f64_f32(long long n, const double* x, float* y, void* stream) {
    SVGP_REQUIRE(x && y, SVGP_ERR_INVALID, "NULL pointer");
    SIMPLE_LAUNCH((k_cast<double, float>), n, n, x, y);
}
extern "C" int svgp_cast_f32_f64(long long n, const float* x, double* y, void* stream) {
    SVGP_REQUIRE(x && y, SVGP_ERR_INVALID, "NULL pointer");
    SIMPLE_LAUNCH((k_cast<float, double>), n, n, x, y);
}
#undef SIMPLE_LAUNCH
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
