// Full-data GP statistics in float32, streamed over the N dimension (config 5 of SURVEY 8d; the
// conditional-generation pass of SVGPVAE_model.py:989-1023 `precompute_GP_params_SVGPVAE`):
//
//   features  : per-row feature vectors so that every kernel of the reference is a function of two dot
//               products (periodic x linear, linear x linear, SE x SE; SVGPVAE_model.py:416-417,427-476,
//               530-600)
//   K_nm build: K (n, m) materialised                               -- HBM-write-bound
//   statistics: S_l = K_nm^T diag(1/var_l) K_nm  (L, m, m),          -- fp32 MFMA-bound
//               v_l = K_nm^T (mean_l / var_l)     (L, m)             -- HBM-read-bound
//
// Nothing here factorises; adding K_mm and inverting is svgp_spd_inverse_batched's job.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { KM_EXP_DOT = 0, KM_DOT_DOT = 1, KM_EXP = 2 };

__device__ __forceinline__ float recip_no_nan_f(float x) { return x == 0.0f ? 0.0f : 1.0f / x; }

// ------------------------------------------------------------------------------------------------
// features.  Row layout (n, D) with D = d1 + d2, followed by the per-row scalar r (n).
//   periodic x linear : [cos a, sin a | o]            K = amp^2 exp((<.,.>_1 - 1) / l^2) <.,.>_2
//   linear x linear   : [act | chr]                   K = <.,.>_1 <.,.>_2
//   SE x SE           : [act | chr], r = |act|^2/(2 l1^2) + |chr|^2/(2 l2^2)
//                                                     K = s1^2 s2^2 exp(<.,.>_1/l1^2 + <.,.>_2/l2^2 - r_a - r_b)
// `normalize` divides a linear segment by its norm (cosine kernel, :465-474, :576-598).
// ------------------------------------------------------------------------------------------------
struct FeatArgs {
    long long n;
    int kind, d1, d2, normalize, ldx, n_table, inducing;
    float p[4];
    const float* x;
    const float* table;
    float* feat;
};

__global__ __launch_bounds__(256) void k_features_f32(FeatArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int D = a.d1 + a.d2;
    const float* xr = a.x + i * a.ldx;
    float* f = a.feat + i * D;
    float r = 0.0f;
    if (a.kind == 0) {
        // mnistSVGP: x = [id, angle, o_1..o_M]; batch rows take o from the table when there is one
        const float ang = xr[1];
        f[0] = cosf(ang);
        f[1] = sinf(ang);
        const float* o = xr + 2;
        if (!a.inducing && a.n_table > 0) {
            int id = (int)xr[0];
            id = id < 0 ? 0 : (id >= a.n_table ? a.n_table - 1 : id);
            o = a.table + (size_t)id * a.d2;
        }
        float nn = 0.0f;
        for (int k = 0; k < a.d2; ++k) nn += o[k] * o[k];
        const float sc = a.normalize ? rsqrtf(nn) : 1.0f;
        for (int k = 0; k < a.d2; ++k) f[2 + k] = o[k] * sc;
    } else {
        // spritesSVGP: batch rows x = [action_id, chr_1..chr_d2], action vector gathered from the GPLVM
        // table (72, d1); inducing rows x = [act_1..act_d1, chr_1..chr_d2]
        const float* act;
        const float* chr;
        if (a.inducing) {
            act = xr;
            chr = xr + a.d1;
        } else {
            int id = (int)xr[0];
            id = id < 0 ? 0 : (id >= a.n_table ? a.n_table - 1 : id);
            act = a.table + (size_t)id * a.d1;
            chr = xr + 1;
        }
        float n1 = 0.0f, n2 = 0.0f;
        for (int k = 0; k < a.d1; ++k) n1 += act[k] * act[k];
        for (int k = 0; k < a.d2; ++k) n2 += chr[k] * chr[k];
        if (a.kind == 1) {
            const float s1 = a.normalize ? rsqrtf(n1) : 1.0f, s2 = a.normalize ? rsqrtf(n2) : 1.0f;
            for (int k = 0; k < a.d1; ++k) f[k] = act[k] * s1;
            for (int k = 0; k < a.d2; ++k) f[a.d1 + k] = chr[k] * s2;
        } else {
            for (int k = 0; k < a.d1; ++k) f[k] = act[k];
            for (int k = 0; k < a.d2; ++k) f[a.d1 + k] = chr[k];
            r = n1 / (2.0f * a.p[0] * a.p[0]) + n2 / (2.0f * a.p[2] * a.p[2]);
        }
    }
    a.feat[a.n * D + i] = r;
}

// ------------------------------------------------------------------------------------------------
// K_nm build.  Each lane owns 4 adjacent columns whose inducing features live in registers; the row
// features are uniform per iteration (scalar loads).  One 16-byte non-temporal store per row per lane:
// a wave writes 1 KB contiguous.  Arithmetic is ~(2 D + 8) flop per 4 output bytes, far under the
// ridge, so the kernel is bounded by the HBM write stream.
// ------------------------------------------------------------------------------------------------
struct KnmArgs {
    long long n;
    int m, rb, npanel;
    float c0, s1, s2;
    const float* fa;   // (n, D) | r (n)
    const float* fb;   // (m, D) | r (m)
    float* K;          // (n, m)
};

constexpr int KNM_RB = 32;

// G column groups of 4 per lane, 1024 columns apart: with G = 2 and m = 2048 a workgroup writes whole rows,
// i.e. one contiguous RB x 8 KB region.
template <int D1, int D2, int MODE, int G = 1, bool NT = true>
__global__ __launch_bounds__(256) void k_knm_f32(KnmArgs a) {
    constexpr int D = D1 + D2;
    // the column panel is the fastest-varying part of the workgroup index, so concurrently resident
    // workgroups write whole rows back to back
    const unsigned panel = blockIdx.x % a.npanel;
    const long long rblk = blockIdx.x / a.npanel;
    int col0[G];
    float fb[G][4][D], rb[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        col0[g] = ((panel * G + g) * 256 + threadIdx.x) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = col0[g] + c < a.m ? col0[g] + c : a.m - 1;
#pragma unroll
            for (int k = 0; k < D; ++k) fb[g][c][k] = a.fb[(size_t)col * D + k];
            rb[g][c] = a.fb[(size_t)a.m * D + col];
        }
    }
    if (col0[0] >= a.m) return;
    const bool vec = (a.m & 3) == 0;   // then col0 + 3 < m and every row start is 16-byte aligned
    const long long r0 = rblk * a.rb;
    const float* __restrict__ ra = a.fa + a.n * D;
#pragma unroll 2
    for (int r = 0; r < a.rb; ++r) {
        const long long row = r0 + r;
        if (row >= a.n) break;
        const float* __restrict__ fa = a.fa + row * D;
        float d1[G][4], d2[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) d1[g][c] = d2[g][c] = 0.0f;
#pragma unroll
        for (int k = 0; k < D1; ++k) {
            const float x = fa[k];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) d1[g][c] = fmaf(x, fb[g][c][k], d1[g][c]);
        }
#pragma unroll
        for (int k = 0; k < D2; ++k) {
            const float x = fa[D1 + k];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) d2[g][c] = fmaf(x, fb[g][c][D1 + k], d2[g][c]);
        }
        const float rr = ra[row];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (col0[g] >= a.m) continue;
            float out[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == KM_EXP_DOT) out[c] = a.c0 * __expf(a.s1 * (d1[g][c] - 1.0f)) * d2[g][c];
                else if (MODE == KM_DOT_DOT) out[c] = d1[g][c] * d2[g][c];
                else out[c] = a.c0 * __expf(fmaf(a.s1, d1[g][c], fmaf(a.s2, d2[g][c], -rr - rb[g][c])));
            }
            float* dst = a.K + row * a.m + col0[g];
            if (vec) {
                f32x4 o = {out[0], out[1], out[2], out[3]};
                if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst));
                else *reinterpret_cast<f32x4*>(dst) = o;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (col0[g] + c < a.m) dst[c] = out[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weights, transposed for the row-streaming kernels: pT[l][n] = 1/var (reciprocal_no_nan, :1013),
// pyT[l][n] = mean/var (:1015-1016)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats_weights_f32(long long n, int L, const float* __restrict__ means,
                                                           const float* __restrict__ vars, float* __restrict__ pT,
                                                           float* __restrict__ pyT) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int l = 0; l < L; ++l) {
        const float p = recip_no_nan_f(vars[i * L + l]);
        pT[(size_t)l * n + i] = p;
        pyT[(size_t)l * n + i] = p * means[i * L + l];
    }
}

// ------------------------------------------------------------------------------------------------
// S_l = K^T diag(p_l) K on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One workgroup (8 waves) owns a 256 x 256 tile of one S_l and one slice of the rows; only tile pairs
// ti <= tj are computed (S is symmetric), the mirror is written by the reduction.  Per row chunk the
// two KC x 256 slabs K[:, i-tile] and p_l * K[:, j-tile] go global -> registers -> LDS (double buffered,
// one barrier per chunk) and every wave runs 64 MFMAs on its 128 x 64 sub-tile: 64 flop per slab byte,
// so the slab traffic (L2 hits: all (pair, l) workgroups of one row slice walk the same rows) stays
// ~2.5 TB/s at full MFMA rate.  Chunks are 32 rows (template parameter).
// MFMA operand layout: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31],
// D reg r -> row 8 (r / 4) + 4 (lane >> 5) + (r & 3), column lane & 31.
// ------------------------------------------------------------------------------------------------
constexpr int ST_T = 256, ST_LD = ST_T + 32, ST_NT = 512;

struct StatsArgs {
    long long n, rows_per_split;
    int m, L, ntile;
    const float* K;
    const float* pT;
    float* part;   // (nsplit, L, m, m), upper tiles only
};

__device__ __forceinline__ f32x4 load_row4(const float* __restrict__ K, long long row, long long row_end, int m, int col,
                                           bool vec) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < row_end) {
        const float* src = K + row * m + col;
        if (vec && col + 3 < m) {
            v = *reinterpret_cast<const f32x4*>(src);
        } else {
            if (col < m) v.x = src[0];
            if (col + 1 < m) v.y = src[1];
            if (col + 2 < m) v.z = src[2];
            if (col + 3 < m) v.w = src[3];
        }
    }
    return v;
}

template <int ST_KC, bool DIAG>
__global__ __launch_bounds__(ST_NT) void k_stats_mfma_f32(StatsArgs a) {
    extern __shared__ __align__(16) float lds[];
    float* As = lds;                          // [2][ST_KC][ST_LD]
    float* Bs = lds + 2 * ST_KC * ST_LD;      // [2][ST_KC][ST_LD]
    // pair index -> (ti <= tj)
    // DIAG: the ntile diagonal tiles (blockIdx.x = ti = tj); else the pairs ti < tj
    int ti = 0, rem = blockIdx.x;
    if (!DIAG) {
        while (rem >= a.ntile - 1 - ti) { rem -= a.ntile - 1 - ti; ++ti; }
    }
    const int tj = DIAG ? (ti = blockIdx.x) : ti + 1 + rem;
    const int l = blockIdx.y;
    const long long n_begin = (long long)blockIdx.z * a.rows_per_split;
    long long n_end = n_begin + a.rows_per_split;
    if (n_end > a.n) n_end = a.n;
    const int i0 = ti * ST_T, j0 = tj * ST_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;   // 2 x 4 waves: 128 x 64 per wave
    const bool vec = (a.m & 3) == 0;
    const int lr = __builtin_amdgcn_readfirstlane(tid >> 6), lc = (tid & 63) * 4;   // staging: rows lr, lr + 8 (wave-uniform: the p_l loads are scalar); columns lc..lc+3
    const float* __restrict__ pl = a.pT + (size_t)l * a.n;
    // diagonal tiles (ti == tj): S is symmetric, so only the blocks (bi <= bj) of the 8 x 8 grid of 32 x 32 blocks are
    // computed; the reduction mirrors them.  Block list of wave w (row bi shared with row 7 - bi between two waves):
    //   even w = 2 r: (r, r) .. (r, r + 4);   odd w = 2 r + 1: the 3 - r remaining blocks of row r, then row 7 - r (r + 1 blocks)
    constexpr bool diag = DIAG;
    int dbi[5], dbj[5], dnv = 5;
    {
        const int wv = __builtin_amdgcn_readfirstlane(wave), r = wv >> 1;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int bi, bj;
            if (!(wv & 1)) { bi = r; bj = r + t; }
            else if (t < 3 - r) { bi = r; bj = r + 5 + t; }
            else { bi = 7 - r; bj = 7 - r + (t - (3 - r)); }
            if (bj > 7) { bi = 7 - r; bj = 7; }       // the fifth (repeated) block of an odd wave: computed, not stored
            dbi[t] = 32 * bi; dbj[t] = 32 * bj;
        }
        if (wv & 1) dnv = 4;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;

    constexpr int NH = ST_KC / 8;
    f32x4 ga[NH], gb[NH];
    float gp[NH];      // p_l of the fetched rows: applied when the chunk is staged (a product right behind the loads parks both
                       // waves of every SIMD -- they sit at the same point of the same workgroup -- for a memory latency per chunk:
                       // 80.1 -> 77.1 ms per pass at the config-5 shard).  Tried on top and removed: the three-chunks-in-flight
                       // loop of the float64 GEMM (stores behind the first MFMA group, barrier before the last): 77.1 again; with
                       // the stores / loads placed between the MFMAs by sched_group_barrier the kernel spills (265 ms).
    // interior tiles / full chunks (all of them at the stress shape) take a branch-free path: plain 16-byte loads
    const bool tile_full = vec && i0 + ST_T <= a.m && j0 + ST_T <= a.m;
    auto fetch = [&](long long nb) {
        if (tile_full && nb + ST_KC <= n_end) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const long long row = nb + lr + 8 * h;
                const float* src = a.K + row * a.m;
                ga[h] = *reinterpret_cast<const f32x4*>(src + i0 + lc);
                gb[h] = *reinterpret_cast<const f32x4*>(src + j0 + lc);
                gp[h] = pl[row];
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const long long row = nb + lr + 8 * h;
            ga[h] = load_row4(a.K, row, n_end, a.m, i0 + lc, vec);
            gb[h] = load_row4(a.K, row, n_end, a.m, j0 + lc, vec);
            gp[h] = row < n_end ? pl[row] : 0.0f;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            *reinterpret_cast<f32x4*>(As + (buf * ST_KC + lr + 8 * h) * ST_LD + lc) = ga[h];
            *reinterpret_cast<f32x4*>(Bs + (buf * ST_KC + lr + 8 * h) * ST_LD + lc) = gb[h] * gp[h];
        }
    };

    if (n_begin < n_end) {
        fetch(n_begin);
        stage(0);
    }
    __syncthreads();
    int cur = 0;
    for (long long nb = n_begin; nb < n_end; nb += ST_KC) {
        const bool more = nb + ST_KC < n_end;
        if (more) fetch(nb + ST_KC);
        const float* Ab = As + cur * ST_KC * ST_LD + wi * 128 + (lane & 31);
        const float* Bb = Bs + cur * ST_KC * ST_LD + wj * 64 + (lane & 31);
        if constexpr (diag) {
            // diagonal tile: the 36 upper 32 x 32 blocks only, 5 (or 4 + one repeated) per wave -- 5 / 8 of the full tile's MFMAs
            const float* A0 = As + cur * ST_KC * ST_LD + (lane & 31);
            const float* B0 = Bs + cur * ST_KC * ST_LD + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < ST_KC / 2; ++kk) {
                const int k = 2 * kk + (lane >> 5);
                float av[5], bv[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) { av[t] = A0[k * ST_LD + dbi[t]]; bv[t] = B0[k * ST_LD + dbj[t]]; }
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    acc[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[t >> 1][t & 1], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < ST_KC / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            float av[4], bv[2];
#pragma unroll
            for (int x = 0; x < 4; ++x) av[x] = Ab[k * ST_LD + 32 * x];
#pragma unroll
            for (int y = 0; y < 2; ++y) bv[y] = Bb[k * ST_LD + 32 * y];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    float* out = a.part + ((size_t)blockIdx.z * a.L + l) * a.m * a.m;
    if constexpr (diag) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (t >= dnv) break;
            const int j = j0 + dbj[t] + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + dbi[t] + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (i < a.m && j < a.m) out[(size_t)i * a.m + j] = acc[t >> 1][t & 1][r];
            }
        }
        return;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int j = j0 + wj * 64 + 32 * y + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * 128 + 32 * x + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (i < a.m && j < a.m) out[(size_t)i * a.m + j] = acc[x][y][r];
            }
        }
}

// S[l][i][j] = sum over row slices of the partial tile; lower tiles read the mirrored element
__global__ __launch_bounds__(256) void k_stats_reduce_f32(int m, int L, int nsplit, const float* __restrict__ part,
                                                          float* __restrict__ S) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, l = blockIdx.z;
    if (j >= m) return;
    const bool upper = (i >> 5) <= (j >> 5);       // computed: tile pairs ti < tj in full, of a diagonal tile the blocks bi <= bj
    const size_t src = upper ? (size_t)i * m + j : (size_t)j * m + i;
    float acc = 0.0f;
    for (int s = 0; s < nsplit; ++s) acc += part[((size_t)s * L + l) * m * m + src];
    S[((size_t)l * m + i) * m + j] = acc;
}

// v_l = K^T (p_l y_l): rows split over blockIdx.y, 16 channels per pass.  VEC: 4 adjacent columns per lane (16-byte
// loads, 4 rows in flight); the scalar form handles m % 4 != 0.  HBM-read-bound (reads K_nm once).
constexpr int SV_LC = 16;
template <bool VEC>
__global__ __launch_bounds__(256) void k_stats_v_f32(long long n, long long rows_per_split, int m, int L,
                                                     const float* __restrict__ K, const float* __restrict__ pyT,
                                                     float* __restrict__ partv) {
    constexpr int CW = VEC ? 4 : 1;
    const int i = (blockIdx.x * 256 + threadIdx.x) * CW;
    const long long nb = (long long)blockIdx.y * rows_per_split;
    long long ne = nb + rows_per_split;
    if (ne > n) ne = n;
    if (i >= m) return;
    for (int l0 = 0; l0 < L; l0 += SV_LC) {
        float acc[SV_LC][CW];
#pragma unroll
        for (int q = 0; q < SV_LC; ++q)
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[q][c] = 0.0f;
        long long r = nb;
        if (VEC) {
            for (; r + 4 <= ne; r += 4) {
                f32x4 kv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kv[u] = *reinterpret_cast<const f32x4*>(K + (r + u) * m + i);
#pragma unroll
                for (int q = 0; q < SV_LC; ++q) {
                    if (l0 + q < L) {
                        const float* py = pyT + (size_t)(l0 + q) * n + r;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float w = py[u];
#pragma unroll
                            for (int c = 0; c < CW; ++c) acc[q][c] = fmaf(kv[u][c], w, acc[q][c]);
                        }
                    }
                }
            }
        }
        for (; r < ne; ++r) {
            float kv[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) kv[c] = K[r * m + i + c];
#pragma unroll
            for (int q = 0; q < SV_LC; ++q)
                if (l0 + q < L) {
                    const float w = pyT[(size_t)(l0 + q) * n + r];
#pragma unroll
                    for (int c = 0; c < CW; ++c) acc[q][c] = fmaf(kv[c], w, acc[q][c]);
                }
        }
#pragma unroll
        for (int q = 0; q < SV_LC; ++q)
            if (l0 + q < L)
#pragma unroll
                for (int c = 0; c < CW; ++c) partv[((size_t)blockIdx.y * L + l0 + q) * m + i + c] = acc[q][c];
    }
}

__global__ __launch_bounds__(256) void k_stats_v_reduce_f32(int m, int L, int nsplit, const float* __restrict__ partv,
                                                            float* __restrict__ v) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L * m) return;
    float acc = 0.0f;
    for (int s = 0; s < nsplit; ++s) acc += partv[(size_t)s * L * m + idx];
    v[idx] = acc;
}

struct StatsPlan {
    int ntile, npair, nsplit, nsplit_v;
    long long rows_per_split, rows_per_split_v;
    size_t off_pT, off_pyT, off_part, off_partv, total;
};

StatsPlan stats_plan(long long n, int m, int L) {
    StatsPlan p;
    p.ntile = (m + ST_T - 1) / ST_T;
    p.npair = p.ntile * (p.ntile + 1) / 2;
    // enough workgroups for ~8 rounds over 256 CUs, but at least 2048 rows per slice
    long long want = (2048 + (long long)p.npair * L - 1) / ((long long)p.npair * L);
    long long cap = n / 2048 > 1 ? n / 2048 : 1;
    long long ns = want < cap ? want : cap;
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    p.nsplit = (int)ns;
    long long rps = (n + ns - 1) / ns;
    p.rows_per_split = (rps + 31) / 32 * 32;
    const int mt = (m & 3) == 0 ? (m / 4 + 255) / 256 : (m + 255) / 256;
    long long nv = (2048 + mt - 1) / mt;
    long long capv = n / 256 > 1 ? n / 256 : 1;
    if (nv > capv) nv = capv;
    p.nsplit_v = (int)nv;
    p.rows_per_split_v = (n + nv - 1) / nv;
    size_t o = 0;
    p.off_pT = o; o += (size_t)L * n;
    p.off_pyT = o; o += (size_t)L * n;
    o = (o + 3) / 4 * 4;
    p.off_part = o; o += (size_t)p.nsplit * L * m * m;
    p.off_partv = o; o += (size_t)p.nsplit_v * L * m;
    p.total = o;
    return p;
}

template <typename F>
int set_lds(F kernel, size_t bytes) {
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SVGP_OK;
}

int check_kdesc(const svgp_stream_kdesc* kd) {
    SVGP_REQUIRE(kd != nullptr, SVGP_ERR_INVALID, "kernel descriptor is NULL");
    SVGP_REQUIRE(kd->kind >= 0 && kd->kind <= 2, SVGP_ERR_INVALID, "kernel kind %d", kd->kind);
    SVGP_REQUIRE(kd->d1 >= 1 && kd->d2 >= 1, SVGP_ERR_INVALID, "feature split %d + %d", kd->d1, kd->d2);
    SVGP_REQUIRE(kd->kind != 0 || kd->d1 == 2, SVGP_ERR_INVALID, "periodic x linear has d1 = 2 (cos, sin)");
    return SVGP_OK;
}

}  // namespace

extern "C" int64_t svgp_stream_feature_elems(const svgp_stream_kdesc* kd, int64_t n) {
    if (!kd || n < 0) return 0;
    return n * (int64_t)(kd->d1 + kd->d2) + n;
}

extern "C" int svgp_stream_features_f32(const svgp_stream_kdesc* kd, int64_t n, const float* x, int ldx, int inducing,
                                        const float* table, float* feat, void* stream) {
    int rc = check_kdesc(kd);
    if (rc) return rc;
    SVGP_REQUIRE(n >= 0 && x && feat, SVGP_ERR_INVALID, "NULL device pointer");
    const int need = kd->kind == 0 ? 2 + kd->d2 : (inducing ? kd->d1 + kd->d2 : 1 + kd->d2);
    SVGP_REQUIRE(ldx >= need, SVGP_ERR_INVALID, "row stride %d < %d columns", ldx, need);
    const bool gathers = !inducing && (kd->kind != 0 || kd->n_table > 0);
    SVGP_REQUIRE(!gathers || (table && kd->n_table > 0), SVGP_ERR_INVALID, "gather table missing");
    if (n == 0) return SVGP_OK;
    FeatArgs a;
    a.n = n; a.kind = kd->kind; a.d1 = kd->d1; a.d2 = kd->d2; a.normalize = kd->normalize; a.ldx = ldx;
    a.n_table = kd->n_table; a.inducing = inducing;
    for (int k = 0; k < 4; ++k) a.p[k] = kd->p[k];
    a.x = x; a.table = table; a.feat = feat;
    hipLaunchKernelGGL(k_features_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_stream_knm_f32(const svgp_stream_kdesc* kd, int64_t n, int m, const float* feat_rows,
                                   const float* feat_inducing, float* K_nm, void* stream) {
    int rc = check_kdesc(kd);
    if (rc) return rc;
    SVGP_REQUIRE(n >= 0 && m >= 1 && feat_rows && feat_inducing && K_nm, SVGP_ERR_INVALID, "NULL device pointer");
    if (n == 0) return SVGP_OK;
    KnmArgs a;
    a.n = n; a.m = m; a.fa = feat_rows; a.fb = feat_inducing; a.K = K_nm;
    a.c0 = 1.0f; a.s1 = 0.0f; a.s2 = 0.0f;
    if (kd->kind == 0) {          // p = [l_GP, amplitude]
        a.c0 = kd->p[1] * kd->p[1];
        a.s1 = 1.0f / (kd->p[0] * kd->p[0]);
    } else if (kd->kind == 2) {   // p = [l1, sigma1, l2, sigma2]
        a.c0 = kd->p[1] * kd->p[1] * kd->p[3] * kd->p[3];
        a.s1 = 1.0f / (kd->p[0] * kd->p[0]);
        a.s2 = 1.0f / (kd->p[2] * kd->p[2]);
    }
    // the per-lane inducing features (4 D strided loads) are the prologue; wide features amortise it over more rows
    a.rb = kd->d1 + kd->d2 <= 12 ? KNM_RB : 4 * KNM_RB;
    hipStream_t s = (hipStream_t)stream;
#define KNM_LAUNCH(G_, ...)                                                                   \
    do {                                                                                      \
        a.npanel = (m + 1024 * G_ - 1) / (1024 * G_);                                         \
        const dim3 grid((unsigned)(((n + a.rb - 1) / a.rb) * a.npanel));                      \
        hipLaunchKernelGGL((k_knm_f32<__VA_ARGS__>), grid, dim3(256), 0, s, a);               \
        SVGP_LAUNCH_CHECK();                                                                  \
        return SVGP_OK;                                                                       \
    } while (0)
    // measured on MI355X at n = 131072, m = 2048 (round-1 probe): 32-64 rows per workgroup and
    // non-temporal 16-byte stores give 5.0-5.4 TB/s; plain stores 3.5-4.5; 8 adjacent columns per lane 2.4;
    // two column groups per lane (whole rows per workgroup) no better than one.  torch's fill_ reaches 6.9
    // on the same buffer, its broadcast add (generated data) 3.9.
#define KNM_CASE(K_, D1_, D2_, MODE_) \
    if (kd->kind == K_ && kd->d1 == D1_ && kd->d2 == D2_) KNM_LAUNCH(1, D1_, D2_, MODE_, 1, true);
    KNM_CASE(0, 2, 4, KM_EXP_DOT)
    KNM_CASE(0, 2, 8, KM_EXP_DOT)
    KNM_CASE(0, 2, 16, KM_EXP_DOT)
    KNM_CASE(0, 2, 32, KM_EXP_DOT)
    KNM_CASE(1, 8, 16, KM_DOT_DOT)
    KNM_CASE(2, 8, 16, KM_EXP)
    KNM_CASE(1, 4, 6, KM_DOT_DOT)
    KNM_CASE(2, 4, 6, KM_EXP)
#undef KNM_CASE
#undef KNM_LAUNCH
    SVGP_REQUIRE(false, SVGP_ERR_UNSUPPORTED,
                 "float32 K_nm build: feature split (%d, %d) of kind %d has no instantiation (periodic x linear: "
                 "M in {4, 8, 16, 32}; SPRITES: (8, 16), (4, 6))", kd->d1, kd->d2, kd->kind);
    return SVGP_ERR_UNSUPPORTED;
}

extern "C" int64_t svgp_stream_stats_workspace_elems(int64_t n, int m, int L) {
    if (n < 0 || m < 1 || L < 1) return 0;
    return (int64_t)stats_plan(n, m, L).total;
}

extern "C" int svgp_stream_stats_f32(int64_t n, int m, int L, const float* K_nm, const float* means,
                                     const float* vars, float* S, float* v, float* ws, int64_t ws_elems, void* stream) {
    SVGP_REQUIRE(n >= 1 && m >= 1 && L >= 1, SVGP_ERR_INVALID, "n=%lld m=%d L=%d", (long long)n, m, L);
    SVGP_REQUIRE(K_nm && means && vars && S && v && ws, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(L <= 65535, SVGP_ERR_UNSUPPORTED, "L=%d", L);
    const StatsPlan p = stats_plan(n, m, L);
    SVGP_REQUIRE(ws_elems >= (int64_t)p.total, SVGP_ERR_INVALID, "workspace has %lld float32 elements, need %lld",
                 (long long)ws_elems, (long long)p.total);
    hipStream_t s = (hipStream_t)stream;
    float* pT = ws + p.off_pT;
    float* pyT = ws + p.off_pyT;
    float* part = ws + p.off_part;
    float* partv = ws + p.off_partv;
    hipLaunchKernelGGL(k_stats_weights_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (long long)n, L, means,
                       vars, pT, pyT);
    SVGP_LAUNCH_CHECK();
    StatsArgs a;
    a.n = n; a.rows_per_split = p.rows_per_split; a.m = m; a.L = L; a.ntile = p.ntile;
    a.K = K_nm; a.pT = pT; a.part = part;
    // 32-row chunks: 128 MFMAs per wave between barriers (16-row chunks measured 6 % slower)
    constexpr int KC = 32;
    const size_t lds = (size_t)4 * KC * ST_LD * sizeof(float);
    int rc = set_lds(k_stats_mfma_f32<KC, false>, lds);
    if (rc) return rc;
    rc = set_lds(k_stats_mfma_f32<KC, true>, lds);
    if (rc) return rc;
    // the full tile pairs first, then the (5 / 8 as expensive) diagonal tiles fill the tail
    if (p.npair > p.ntile)
        hipLaunchKernelGGL((k_stats_mfma_f32<KC, false>), dim3(p.npair - p.ntile, L, p.nsplit), dim3(ST_NT), lds, s, a);
    hipLaunchKernelGGL((k_stats_mfma_f32<KC, true>), dim3(p.ntile, L, p.nsplit), dim3(ST_NT), lds, s, a);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_stats_reduce_f32, dim3((m + 255) / 256, m, L), dim3(256), 0, s, m, L, p.nsplit, part, S);
    SVGP_LAUNCH_CHECK();
    if ((m & 3) == 0)
        hipLaunchKernelGGL(k_stats_v_f32<true>, dim3((m / 4 + 255) / 256, p.nsplit_v), dim3(256), 0, s, (long long)n,
                           p.rows_per_split_v, m, L, K_nm, pyT, partv);
    else
        hipLaunchKernelGGL(k_stats_v_f32<false>, dim3((m + 255) / 256, p.nsplit_v), dim3(256), 0, s, (long long)n,
                           p.rows_per_split_v, m, L, K_nm, pyT, partv);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_stats_v_reduce_f32, dim3((L * m + 255) / 256), dim3(256), 0, s, m, L, p.nsplit_v, partv, v);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
