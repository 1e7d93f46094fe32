// SPRITES pieces of the SVGPVAE path (SURVEY 8a rows a2, a8 glue):
//   spritesSVGP.kernel_matrix (SVGPVAE_model.py:550-600): K = k_action * k_character, each Linear
//   (optionally cosine-normalised, :576-598) or ExponentiatedQuadratic (K_SE, :530-544); batch rows gather
//   their action vector from the GPLVM table by id (:565) and carry the character vector in aux[:,1:].
//   Forward + hand-derived VJP (inducing points, GPLVM table, SE hyper-parameters, batch character vectors).
//   aux_data_SVGPVAE_sprites (:1086-1115): segment_mean over each character's frames, repeat, prepend id.
//   Small element-wise helpers of the SPRITES step (encoder head, pooling, squared error, clipping).
#include "common.hpp"

namespace {

#define SP_MAXD 32

struct SpK {
    int b, m, La, Lc, n_act, kind;   // kind: 0 linear, 1 normalised linear, 2 SE
    real rep_weight;
    const real* aux;    // (b, 1+Lc)
    const real* ip;     // (m, La+Lc)
    const real* table;  // (n_act, La)
    const real* se;     // l_action, sigma_action, l_character, sigma_character
};

__device__ __forceinline__ real dotd(const real* x, const real* y, int D) {
    real s = 0;
    for (int k = 0; k < D; ++k) s += x[k] * y[k];
    return s;
}
// group kernel value and the coefficients of d k / d y = ca * x + cb * y   (symmetric in x <-> y)
__device__ __forceinline__ real grp_k(int kind, const real* x, const real* y, int D, real ell, real sig, real& ca,
                                      real& cb, real& d2) {
    d2 = 0;
    if (kind == 2) {
        for (int k = 0; k < D; ++k) { const real t = x[k] - y[k]; d2 += t * t; }
        const real kv = sig * sig * exp(-d2 / (real(2) * ell * ell));
        ca = kv / (ell * ell); cb = -ca;
        return kv;
    }
    const real xy = dotd(x, y, D);
    if (kind == 0) { ca = 1; cb = 0; return xy; }
    const real nx = sqrt(dotd(x, x, D)), ny = sqrt(dotd(y, y, D));
    const real kv = xy / (nx * ny);
    ca = real(1) / (nx * ny); cb = -kv / (ny * ny);
    return kv;
}
__device__ __forceinline__ const real* act_row(const SpK& a, int n) {
    return a.table + (size_t)((long long)a.aux[(size_t)n * (1 + a.Lc)]) * a.La;
}

__global__ __launch_bounds__(256) void k_sprites_kernel_fwd(SpK a, real* __restrict__ K, real* __restrict__ Kn,
                                                            real* __restrict__ knn) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nbm = (long long)a.b * a.m, nmm = (long long)a.m * a.m;
    const int D = a.La + a.Lc;
    const real la = a.se[0], sa = a.se[1], lc = a.se[2], sc = a.se[3];
    real ca, cb, d2;
    if (idx < nbm) {
        const int n = (int)(idx / a.m), j = (int)(idx % a.m);
        const real* z = a.ip + (size_t)j * D;
        Kn[idx] = grp_k(a.kind, act_row(a, n), z, a.La, la, sa, ca, cb, d2) *
                  grp_k(a.kind, a.aux + (size_t)n * (1 + a.Lc) + 1, z + a.La, a.Lc, lc, sc, ca, cb, d2);
    } else if (idx < nbm + nmm) {
        const long long o = idx - nbm;
        const real* zi = a.ip + (size_t)(o / a.m) * D;
        const real* zj = a.ip + (size_t)(o % a.m) * D;
        K[o] = grp_k(a.kind, zi, zj, a.La, la, sa, ca, cb, d2) * grp_k(a.kind, zi + a.La, zj + a.La, a.Lc, lc, sc, ca, cb, d2);
    } else if (idx < nbm + nmm + a.b) {
        const int n = (int)(idx - nbm - nmm);
        const real* xa = act_row(a, n);
        const real* xc = a.aux + (size_t)n * (1 + a.Lc) + 1;
        knn[n] = grp_k(a.kind, xa, xa, a.La, la, sa, ca, cb, d2) * grp_k(a.kind, xc, xc, a.Lc, lc, sc, ca, cb, d2);
    }
}

// The same values by 16 x 16 output tiles: the 16 + 16 feature rows of a tile are staged in LDS with coalesced loads (the element-wise
// kernel above reads 2 D per-lane doubles with a row stride between the lanes for every entry: 165 us for the 1.04 M entries of the
// SPRITES shape), and the norms of the cosine-normalised kernel are formed once per row instead of once per entry -- by the same
// expressions, so the entries are bit-identical to k_sprites_kernel_fwd's (which stays for k_nn and as the reference form).
// blocks [0, nbt nmt): tiles of K_nm; [.., + nmt nmt): tiles of K_mm; the rest: k_nn, 256 entries each.
__global__ __launch_bounds__(256) void k_sprites_kernel_fwd_tiles(SpK a, int nbt, int nmt, real* __restrict__ K, real* __restrict__ Kn,
                                                                  real* __restrict__ knn) {
    __shared__ real xs[16][2 * SP_MAXD + 1], zs[16][2 * SP_MAXD + 1], nrm[2][16][2];
    const int La = a.La, Lc = a.Lc, D = La + Lc;
    const real la = a.se[0], sa = a.se[1], lc = a.se[2], sc = a.se[3];
    int blk = blockIdx.x;
    if (blk >= nbt * nmt + nmt * nmt) {                     // k_nn
        const int n = (blk - nbt * nmt - nmt * nmt) * 256 + threadIdx.x;
        if (n < a.b) {
            real ca, cb, d2;
            const real* xa = act_row(a, n);
            const real* xc = a.aux + (size_t)n * (1 + Lc) + 1;
            knn[n] = grp_k(a.kind, xa, xa, La, la, sa, ca, cb, d2) * grp_k(a.kind, xc, xc, Lc, lc, sc, ca, cb, d2);
        }
        return;
    }
    const bool kn = blk < nbt * nmt;
    if (!kn) blk -= nbt * nmt;
    const int ti = blk / nmt, tj = blk - ti * nmt, i0 = ti * 16, j0 = tj * 16, ni = kn ? a.b : a.m;
    // stage: row rr of the tile's row side / column side (clamped; entries beyond the matrix are not stored)
    for (int e = threadIdx.x; e < 16 * D; e += 256) {
        const int rr = e / D, k = e - rr * D, i = min(i0 + rr, ni - 1), j = min(j0 + rr, a.m - 1);
        zs[rr][k] = a.ip[(size_t)j * D + k];
        if (kn) xs[rr][k] = k < La ? act_row(a, i)[k] : a.aux[(size_t)i * (1 + Lc) + 1 + (k - La)];
        else xs[rr][k] = a.ip[(size_t)i * D + k];
    }
    __syncthreads();
    if (a.kind == 1 && threadIdx.x < 64) {                  // norms: side (x / z), row, group (action / character)
        const int side = threadIdx.x >> 5, rr = (threadIdx.x >> 1) & 15, grp = threadIdx.x & 1;
        const real* v = (side ? zs[rr] : xs[rr]) + (grp ? La : 0);
        nrm[side][rr][grp] = sqrt(dotd(v, v, grp ? Lc : La));
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15, gi = i0 + r, gj = j0 + c;
    if (gi >= ni || gj >= a.m) return;
    const real* x = xs[r];
    const real* z = zs[c];
    real kA, kC;
    if (a.kind == 2) {
        real dA = 0, dC = 0;
        for (int k = 0; k < La; ++k) { const real t = x[k] - z[k]; dA += t * t; }
        for (int k = 0; k < Lc; ++k) { const real t = x[La + k] - z[La + k]; dC += t * t; }
        kA = sa * sa * exp(-dA / (real(2) * la * la));
        kC = sc * sc * exp(-dC / (real(2) * lc * lc));
    } else {
        kA = dotd(x, z, La);
        kC = dotd(x + La, z + La, Lc);
        if (a.kind == 1) { kA = kA / (nrm[0][r][0] * nrm[1][c][0]); kC = kC / (nrm[0][r][1] * nrm[1][c][1]); }
    }
    (kn ? Kn : K)[(size_t)gi * a.m + gj] = kA * kC;
}

// VJP, inducing side: workgroup j.  d_ip[j] (La+Lc) and SE-parameter partials part_se[j][4].
// The source rows (batch rows, then inducing rows) go through LDS in chunks of 256, staged with coalesced loads (row stride DP = D | 1
// doubles); the target z_j sits in LDS too.  (Read straight from global memory every (source, target) pair cost 2 D per-lane 8-byte
// loads with a row stride between the lanes -- the kernel ran at the rate of the L1 tag look-ups: 200-240 us inside the SPRITES step,
// where it now heads the chain that ends the step.)
template <int LA, int LC>       // compile-time bounds of the two feature groups (La <= LA, Lc <= LC): accumulators and loops of exact size
__global__ __launch_bounds__(256) void k_sprites_kernel_bwd_cols(SpK a, const real* __restrict__ Kbar,
                                                                 const real* __restrict__ Knbar,
                                                                 real* __restrict__ d_ip, real* __restrict__ part_se) {
    __shared__ real wred[4][LA + LC + 4];
    __shared__ real zt[LA + LC];
    extern __shared__ real src[];                 // 256 x DP
    const int j = blockIdx.x, La = a.La, Lc = a.Lc, D = La + Lc, DP = D | 1;
    const real la = a.se[0], sa = a.se[1], lc = a.se[2], sc = a.se[3];
    if (threadIdx.x < D) zt[threadIdx.x] = a.ip[(size_t)j * D + threadIdx.x];
    const real* zj = zt;
    real acc[LA + LC + 4];
#pragma unroll
    for (int k = 0; k < LA + LC + 4; ++k) acc[k] = 0;
    auto add = [&](const real* xa, const real* xc, real c_vec, real c_par) {
        real caA, cbA, d2A, caC, cbC, d2C;
        const real kA = grp_k(a.kind, xa, zj, La, la, sa, caA, cbA, d2A);
        const real kC = grp_k(a.kind, xc, zj + La, Lc, lc, sc, caC, cbC, d2C);
        const real cA = c_vec * kC, cC = c_vec * kA;
#pragma unroll
        for (int k = 0; k < LA; ++k)
            if (k < La) acc[k] += cA * (caA * xa[k] + cbA * zj[k]);
#pragma unroll
        for (int k = 0; k < LC; ++k)
            if (k < Lc) acc[LA + k] += cC * (caC * xc[k] + cbC * zj[La + k]);
        if (a.kind == 2) {
            const real kk = c_par * kA * kC;
            acc[LA + LC + 0] += kk * d2A / (la * la * la);
            acc[LA + LC + 1] += kk * real(2) / sa;
            acc[LA + LC + 2] += kk * d2C / (lc * lc * lc);
            acc[LA + LC + 3] += kk * real(2) / sc;
        }
    };
    for (int n0 = 0; n0 < a.b; n0 += 256) {
        const int cnt = min(256, a.b - n0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * La; e += 256) { const int rr = e / La, k = e - rr * La; src[rr * DP + k] = act_row(a, n0 + rr)[k]; }
        for (int e = threadIdx.x; e < cnt * (1 + Lc); e += 256) {
            const int rr = e / (1 + Lc), k = e - rr * (1 + Lc);
            if (k > 0) src[rr * DP + La + k - 1] = a.aux[(size_t)n0 * (1 + Lc) + e];
        }
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const real c = Knbar[(size_t)(n0 + threadIdx.x) * a.m + j];
            add(src + threadIdx.x * DP, src + threadIdx.x * DP + La, c, c);
        }
    }
    for (int i0 = 0; i0 < a.m; i0 += 256) {
        const int cnt = min(256, a.m - i0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * D; e += 256) { const int rr = e / D, k = e - rr * D; src[rr * DP + k] = a.ip[(size_t)i0 * D + e]; }
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const int i = i0 + threadIdx.x;
            const real g_ji = a.rep_weight * Kbar[(size_t)j * a.m + i], g_ij = a.rep_weight * Kbar[(size_t)i * a.m + j];
            add(src + threadIdx.x * DP, src + threadIdx.x * DP + La, g_ji + g_ij, g_ji);       // hyper-parameters: each entry (j,i) once
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < LA + LC + 4; ++k) {
        const bool used = (k < La) || (k >= LA && k < LA + Lc) || (k >= LA + LC);
        if (used) { const real t = wave_sum(acc[k]); if (lane == 0) wred[wv][k] = t; }
    }
    __syncthreads();
    if (threadIdx.x < LA + LC + 4) {
        const int k = threadIdx.x;
        const real s = wred[0][k] + wred[1][k] + wred[2][k] + wred[3][k];
        if (k < La) d_ip[(size_t)j * D + k] = s;
        else if (k >= LA && k < LA + Lc) d_ip[(size_t)j * D + La + (k - LA)] = s;
        else if (k >= LA + LC) part_se[j * 4 + (k - LA - LC)] = s;
    }
}

// VJP, batch-row side: one wave per row.  d_xa (b,La) [for the table scatter], d_char (b,Lc), SE partials per row.
template <int LA, int LC>
__global__ __launch_bounds__(64) void k_sprites_kernel_bwd_rows(SpK a, const real* __restrict__ Knbar,
                                                                const real* __restrict__ knnbar,
                                                                real* __restrict__ d_xa, real* __restrict__ d_char,
                                                                real* __restrict__ part_se) {
    __shared__ real xt[LA + LC];
    extern __shared__ real src[];                 // 64 x DP: a chunk of inducing rows, staged with coalesced loads (see the column kernel)
    const int n = blockIdx.x, lane = threadIdx.x, La = a.La, Lc = a.Lc, D = La + Lc, DP = D | 1;
    const real la = a.se[0], sa = a.se[1], lc = a.se[2], sc = a.se[3];
    if (lane < La) xt[lane] = act_row(a, n)[lane];
    if (lane < Lc) xt[LA + lane] = a.aux[(size_t)n * (1 + Lc) + 1 + lane];
    const real* xa = xt;
    const real* xc = xt + LA;
    real acc[LA + LC];
#pragma unroll
    for (int k = 0; k < LA + LC; ++k) acc[k] = 0;
    for (int j0 = 0; j0 < a.m; j0 += 64) {
        const int cnt = min(64, a.m - j0);
        __syncthreads();
        for (int e = lane; e < cnt * D; e += 64) { const int rr = e / D, k = e - rr * D; src[rr * DP + k] = a.ip[(size_t)j0 * D + e]; }
        __syncthreads();
        if (lane < cnt) {
            const real* zj = src + lane * DP;
            real caA, cbA, d2A, caC, cbC, d2C;
            const real kA = grp_k(a.kind, zj, xa, La, la, sa, caA, cbA, d2A);          // d/d(second arg) = ca*first + cb*second
            const real kC = grp_k(a.kind, zj + La, xc, Lc, lc, sc, caC, cbC, d2C);
            const real c = Knbar[(size_t)n * a.m + j0 + lane], cA = c * kC, cC = c * kA;
#pragma unroll
            for (int k = 0; k < LA; ++k)
                if (k < La) acc[k] += cA * (caA * zj[k] + cbA * xa[k]);
#pragma unroll
            for (int k = 0; k < LC; ++k)
                if (k < Lc) acc[LA + k] += cC * (caC * zj[La + k] + cbC * xc[k]);
        }
    }
    // k_nn = kA(xa,xa) kC(xc,xc)
    const real g = knnbar[n];
    real separt[4] = {0, 0, 0, 0};
    if (lane == 0) {
        if (a.kind == 0) {
            const real na2 = dotd(xa, xa, La), nc2 = dotd(xc, xc, Lc);
            // (compile-time indices: a run-time index would put the whole accumulator array into scratch memory -- it did: 560 bytes
            // of scratch per lane and every accumulation of the row loop a scratch load / store)
#pragma unroll
            for (int k = 0; k < LA; ++k)
                if (k < La) acc[k] += g * real(2) * nc2 * xa[k];
#pragma unroll
            for (int k = 0; k < LC; ++k)
                if (k < Lc) acc[LA + k] += g * real(2) * na2 * xc[k];
        } else if (a.kind == 2) {
            const real knn = sa * sa * sc * sc;
            separt[1] = g * real(2) * knn / sa;
            separt[3] = g * real(2) * knn / sc;
        }
    }
#pragma unroll
    for (int k = 0; k < LA; ++k)
        if (k < La) { const real t = wave_sum(acc[k]); if (lane == 0) d_xa[(size_t)n * La + k] = t; }
#pragma unroll
    for (int k = 0; k < LC; ++k)
        if (k < Lc) { const real t = wave_sum(acc[LA + k]); if (lane == 0) d_char[(size_t)n * Lc + k] = t; }
    if (lane == 0)
        for (int k = 0; k < 4; ++k) part_se[(size_t)(a.m + n) * 4 + k] = separt[k];
}

// ---- the reverse pass by tiles (what the step runs; the two kernels above stay as the reference form for feature groups beyond
// the template bounds).  A workgroup = 16 TARGETS (mode 0: inducing points z_j, mode 1: batch rows x_n) x one split of the SOURCE rows
// (mode 0: the batch rows, coefficient Knbar[n][j], then the inducing rows, coefficients rep_weight (Kbar[j][i] + Kbar[i][j]) for the
// vector part and rep_weight Kbar[j][i] for the hyper-parameters; mode 1: the inducing rows, coefficient Knbar[n][j]).  Per chunk
// of 64 source rows the rows AND the 64 x 16 coefficient tile are staged in LDS with coalesced loads (the per-target kernels read a
// matrix COLUMN per workgroup and re-staged every source row for each single target); thread (target tt, source lane sl) takes the
// sources sl, sl + 16, ...; the 16 source lanes of a target are neighbouring lanes (one DPP row): their sums meet by four xor
// shuffles.  Partial sums per (split, target) go to scratch and are added in split order by k_sprites_kernel_bwd_finish.
#define KB_TT 16
#define KB_SC 64
template <int LA, int LC>
__global__ __launch_bounds__(256) void k_sprites_kernel_bwd_tiles(SpK a, int nsplit0, int nsplit1, const real* __restrict__ Kbar,
                                                                  const real* __restrict__ Knbar, real* __restrict__ part0,
                                                                  real* __restrict__ part1) {
    constexpr int NV = LA + LC + 4, DPM = (LA + LC) | 1;
    __shared__ real tgt[KB_TT][DPM], src[KB_SC][DPM], cv[KB_SC][KB_TT + 1], cpar[KB_SC][KB_TT + 1];
    const int mode = blockIdx.z, La = a.La, Lc = a.Lc, D = La + Lc;
    const int nt = mode ? a.b : a.m, nsplit = mode ? nsplit1 : nsplit0, split = blockIdx.y, t0 = blockIdx.x * KB_TT;
    if (t0 >= nt || split >= nsplit) return;
    const real la = a.se[0], sa = a.se[1], lc = a.se[2], sc = a.se[3];
    const int tid = threadIdx.x, tt = tid >> 4, sl = tid & 15;
    auto feat = [&](bool batch_row, int row, int k) -> real {           // feature k of a batch row (gathered action vector | character) / inducing row
        return batch_row ? (k < La ? act_row(a, row)[k] : a.aux[(size_t)row * (1 + Lc) + 1 + (k - La)]) : a.ip[(size_t)row * D + k];
    };
    for (int e = tid; e < KB_TT * D; e += 256) {
        const int rr = e / D, k = e - rr * D;
        tgt[rr][k < La ? k : LA + (k - La)] = feat(mode == 1, min(t0 + rr, nt - 1), k);
    }
    real acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0;
    const int cx = mode ? 0 : (a.b + KB_SC - 1) / KB_SC, cz = (a.m + KB_SC - 1) / KB_SC;
    for (int c = split; c < cx + cz; c += nsplit) {
        const bool is_x = c < cx;
        const int r0 = KB_SC * (is_x ? c : c - cx), ns = is_x ? a.b : a.m;
        __syncthreads();
        for (int e = tid; e < KB_SC * D; e += 256) {
            const int rr = e / D, k = e - rr * D;
            src[rr][k < La ? k : LA + (k - La)] = feat(is_x, min(r0 + rr, ns - 1), k);
        }
        if (mode == 0 && is_x) {
            for (int e = tid; e < KB_SC * KB_TT; e += 256) {
                const int s_ = e >> 4, t_ = e & 15, n = r0 + s_, j = t0 + t_;
                const real v = (n < a.b && j < a.m) ? Knbar[(size_t)n * a.m + j] : real(0);
                cv[s_][t_] = 0; cpar[s_][t_] = v;                       // (vector coefficient = cv + cpar)
            }
        } else if (mode == 0) {
            for (int e = tid; e < KB_SC * KB_TT; e += 256) {
                const int t_ = e >> 6, s_ = e & 63, i = r0 + s_, j = t0 + t_;
                cpar[s_][t_] = (i < a.m && j < a.m) ? a.rep_weight * Kbar[(size_t)j * a.m + i] : real(0);
            }
            for (int e = tid; e < KB_SC * KB_TT; e += 256) {
                const int s_ = e >> 4, t_ = e & 15, i = r0 + s_, j = t0 + t_;
                cv[s_][t_] = (i < a.m && j < a.m) ? a.rep_weight * Kbar[(size_t)i * a.m + j] : real(0);
            }
        } else {
            for (int e = tid; e < KB_SC * KB_TT; e += 256) {
                const int t_ = e >> 6, s_ = e & 63, j = r0 + s_, n = t0 + t_;
                cv[s_][t_] = 0; cpar[s_][t_] = (j < a.m && n < a.b) ? Knbar[(size_t)n * a.m + j] : real(0);
            }
        }
        __syncthreads();
        const real* y = tgt[tt];
#pragma unroll
        for (int q = 0; q < KB_SC / 16; ++q) {
            const int s_ = sl + 16 * q;
            const real* x = src[s_];
            const real c_par = cpar[s_][tt], c_vec = cv[s_][tt] + c_par;
            real caA, cbA, d2A, caC, cbC, d2C;
            const real kA = grp_k(a.kind, x, y, La, la, sa, caA, cbA, d2A);             // d k / d (second argument) = ca first + cb second
            const real kC = grp_k(a.kind, x + LA, y + LA, Lc, lc, sc, caC, cbC, d2C);
            const real cA = c_vec * kC, cC = c_vec * kA;
#pragma unroll
            for (int k = 0; k < LA; ++k)
                if (k < La) acc[k] += cA * (caA * x[k] + cbA * y[k]);
#pragma unroll
            for (int k = 0; k < LC; ++k)
                if (k < Lc) acc[LA + k] += cC * (caC * x[LA + k] + cbC * y[LA + k]);
            if (mode == 0 && a.kind == 2) {
                const real kk = c_par * kA * kC;
                acc[LA + LC + 0] += kk * d2A / (la * la * la);
                acc[LA + LC + 1] += kk * real(2) / sa;
                acc[LA + LC + 2] += kk * d2C / (lc * lc * lc);
                acc[LA + LC + 3] += kk * real(2) / sc;
            }
        }
    }
    real* po = (mode ? part1 : part0) + ((size_t)split * nt + min(t0 + tt, nt - 1)) * NV;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const bool used = (k < La) || (k >= LA && k < LA + Lc) || (k >= LA + LC && mode == 0);
        if (used) {
            real v = acc[k];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
            if (sl == 0 && t0 + tt < nt) po[k] = v;
        }
    }
}
// partial sums over the splits, in split order; the k_nn terms of the batch rows (k_nn = k_A(x_a, x_a) k_C(x_c, x_c))
template <int LA, int LC>
__global__ void k_sprites_kernel_bwd_finish(SpK a, int nsplit0, int nsplit1, const real* __restrict__ part0,
                                            const real* __restrict__ part1, const real* __restrict__ knnbar, real* __restrict__ d_ip,
                                            real* __restrict__ d_xa, real* __restrict__ d_char, real* __restrict__ part_se) {
    constexpr int NV = LA + LC + 4;
    const int La = a.La, Lc = a.Lc, D = La + Lc;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, n0 = (long long)a.m * NV, n1 = (long long)a.b * NV;
    if (i < n0) {
        const int j = (int)(i / NV), k = (int)(i % NV);
        const bool used = (k < La) || (k >= LA && k < LA + Lc) || k >= LA + LC;
        if (!used) return;
        real s = 0;
        for (int sp = 0; sp < nsplit0; ++sp) s += part0[((size_t)sp * a.m + j) * NV + k];
        if (k < La) d_ip[(size_t)j * D + k] = s;
        else if (k < LA + LC) d_ip[(size_t)j * D + La + (k - LA)] = s;
        else part_se[(size_t)j * 4 + (k - LA - LC)] = s;
    } else if (i < n0 + n1) {
        const int n = (int)((i - n0) / NV), k = (int)((i - n0) % NV);
        const real g = knnbar[n];
        if (k >= LA + LC) {                                  // hyper-parameter partials of k_nn
            const int q = k - LA - LC;
            real v = 0;
            if (a.kind == 2 && (q == 1 || q == 3)) { const real sa = a.se[1], sc = a.se[3]; v = g * real(2) * sa * sa * sc * sc / (q == 1 ? sa : sc); }
            part_se[(size_t)(a.m + n) * 4 + q] = v;
            return;
        }
        const bool used = (k < La) || (k >= LA && k < LA + Lc);
        if (!used) return;
        real s = 0;
        for (int sp = 0; sp < nsplit1; ++sp) s += part1[((size_t)sp * a.b + n) * NV + k];
        if (a.kind == 0) {
            const real* xa = act_row(a, n);
            const real* xc = a.aux + (size_t)n * (1 + Lc) + 1;
            if (k < La) s += g * real(2) * dotd(xc, xc, Lc) * xa[k];
            else s += g * real(2) * dotd(xa, xa, La) * xc[k - LA];
        }
        if (k < La) d_xa[(size_t)n * La + k] = s;
        else d_char[(size_t)n * Lc + (k - LA)] = s;
    }
}

// table scatter + final SE-parameter sums (last block).  Workgroup r < n_act: d_table[r][k] = sum of d_xa[n][k] over the batch rows with
// action id r -- thread (row lane nl, k) walks rows nl, nl + NL, ..., the NL lane sums are added in lane order (fixed order: deterministic).
// (One thread per table entry walking all b rows was a 40 us serial chain behind the row kernel.)
__global__ __launch_bounds__(256) void k_sprites_kernel_bwd_scatter(SpK a, const real* __restrict__ d_xa,
                                                                    const real* __restrict__ part_se,
                                                                    real* __restrict__ d_table, real* __restrict__ d_se) {
    __shared__ real red[256];
    if (blockIdx.x == gridDim.x - 1) {
        for (int k = 0; k < 4; ++k) {
            real s = 0;
            for (int i = threadIdx.x; i < a.m + a.b; i += blockDim.x) s += part_se[(size_t)i * 4 + k];
            s = block_sum(s, red);
            if (threadIdx.x == 0) d_se[k] = s;
            __syncthreads();
        }
        return;
    }
    const int r = blockIdx.x, La = a.La;
    int KL = 1;
    while (KL < La) KL <<= 1;                       // k lanes (a power of two <= 32), NL row lanes
    const int NL = 256 / KL, k = threadIdx.x % KL, nl = threadIdx.x / KL;
    real acc = 0;
    if (k < La)
        for (int n = nl; n < a.b; n += NL)
            if ((int)a.aux[(size_t)n * (1 + a.Lc)] == r) acc += d_xa[(size_t)n * La + k];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (nl == 0 && k < La) {
        real t = 0;
        for (int q = 0; q < NL; ++q) t += red[q * KL + k];
        d_table[(size_t)r * La + k] = t;
    }
}

// ---- aux data: segment mean over seg_len consecutive rows, repeat, prepend the action id --------------
__global__ void k_sprites_aux_fwd(int b, int seg_len, int Lc, const real* __restrict__ repr,
                                  const real* __restrict__ action_ids, real* __restrict__ aux) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * (1 + Lc)) return;
    const int n = i / (1 + Lc), c = i % (1 + Lc);
    if (c == 0) { aux[i] = action_ids[n]; return; }
    const int g0 = (n / seg_len) * seg_len;
    real s = 0;
    for (int t = 0; t < seg_len; ++t) s += repr[(size_t)(g0 + t) * Lc + c - 1];
    aux[i] = s / (real)seg_len;
}
__global__ void k_sprites_aux_bwd(int b, int seg_len, int Lc, const real* __restrict__ d_char, real* __restrict__ d_repr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * Lc) return;
    const int n = i / Lc, c = i % Lc, g0 = (n / seg_len) * seg_len;
    real s = 0;
    for (int t = 0; t < seg_len; ++t) s += d_char[(size_t)(g0 + t) * Lc + c];
    d_repr[i] = s / (real)seg_len;
}
// ---- average pooling over all HW positions of an (n, HW, C) map and its reverse ---------------------------
template <typename T>
__global__ void k_avgpool_fwd(int n, int HW, int Cc, const T* __restrict__ x, T* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * Cc) return;
    const int im = i / Cc, c = i % Cc;
    T s = 0;
    for (int p = 0; p < HW; ++p) s += x[((size_t)im * HW + p) * Cc + c];
    y[i] = s / (T)HW;
}
template <typename T>
__global__ void k_avgpool_bwd(long long tot, int HW, int Cc, const T* __restrict__ dy, T* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const int c = (int)(i % Cc);
    const long long im = i / ((long long)HW * Cc);
    dx[i] = dy[im * Cc + c] / (T)HW;
}
// ---- encoder head: enc (b,2L) (+bias) -> mu, var_raw = exp, var = clip; and its reverse -----------------
__global__ void k_enc_head_fwd(int b, int L, int clip, const real* __restrict__ bias, real* __restrict__ enc,
                               real* __restrict__ mu, real* __restrict__ var_raw, real* __restrict__ var) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * 2 * L) return;
    const int n = i / (2 * L), j = i % (2 * L);
    const real v = enc[i] + bias[j];
    enc[i] = v;
    if (j < L) mu[(size_t)n * L + j] = v;
    else {
        const real vr = exp(v);
        var_raw[(size_t)n * L + j - L] = vr;
        var[(size_t)n * L + j - L] = clip ? fmin(fmax(vr, 1e-3), 10.0) : vr;
    }
}
__global__ void k_enc_head_bwd(int b, int L, int clip, const real* __restrict__ var_raw, const real* __restrict__ ybar,
                               const real* __restrict__ s2bar, real* __restrict__ d_enc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * 2 * L) return;
    const int n = i / (2 * L), j = i % (2 * L);
    if (j < L) d_enc[i] = ybar[(size_t)n * L + j];
    else {
        const real vr = var_raw[(size_t)n * L + j - L];
        const bool pass = !clip || (vr >= 1e-3 && vr <= 10.0);
        d_enc[i] = pass ? s2bar[(size_t)n * L + j - L] * vr : real(0);
    }
}
template <typename T>
__global__ void k_bias_add(long long tot, int Cc, const T* __restrict__ bias, T* __restrict__ x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) x[i] += bias[i % Cc];
}
// ---- squared reconstruction error: per-block partial sums (n_part blocks) and its gradient -----------------
// (float32 inputs: the differences and their squares are formed and summed in float64 -- the partial sums feed the
// float64 scalar epilogue)
template <typename T>
__global__ __launch_bounds__(256) void k_sqerr_fwd(long long tot, const T* __restrict__ x, const T* __restrict__ xh,
                                                   real* __restrict__ part_sums) {
    __shared__ real red[16];
    real s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const real d = (real)x[i] - (real)xh[i];
        s += d * d;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part_sums[blockIdx.x * 4 + 2] = s;
}
template <typename T>
__global__ void k_sqerr_bwd(long long tot, int geco, real inv_bglobal, real inv_npix, const real* __restrict__ state,
                            const T* __restrict__ x, const T* __restrict__ xh, T* __restrict__ dxh) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const real gscale = (geco ? state[SVGP_ST_LAGRANGE] * inv_bglobal : real(1)) * inv_npix;
    dxh[i] = (T)(real(2) * gscale * ((real)xh[i] - (real)x[i]));
}
__global__ void k_clip(long long tot, real thr, real* __restrict__ g) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) g[i] = fmin(fmax(g[i], -thr), thr);
}

inline unsigned nb256(long long n) { return (unsigned)((n + 255) / 256); }

int make_spk(const svgp_sprites_kcfg* c, const double* aux, const double* ip, const double* table, const double* se,
             SpK& a) {
    SVGP_REQUIRE(c && aux && ip && table && se, SVGP_ERR_INVALID, "NULL pointer");
    SVGP_REQUIRE(c->b >= 1 && c->m >= 1 && c->La >= 1 && c->Lc >= 1 && c->La <= SP_MAXD && c->Lc <= SP_MAXD &&
                 c->n_act >= 1, SVGP_ERR_UNSUPPORTED, "sprites kernel: need 1 <= L_action, L_character <= %d", SP_MAXD);
    a.b = c->b; a.m = c->m; a.La = c->La; a.Lc = c->Lc; a.n_act = c->n_act;
    a.kind = c->k_se ? 2 : (c->normalize ? 1 : 0);
    a.rep_weight = c->rep_weight; a.aux = aux; a.ip = ip; a.table = table; a.se = se;
    return SVGP_OK;
}

}  // namespace

extern "C" int svgp_sprites_kernel_matrix_fwd(const svgp_sprites_kcfg* c, const double* aux, const double* ip,
                                              const double* table, const double* se, double* K, double* Kn, double* knn,
                                              void* stream) {
    SpK a;
    int rc = make_spk(c, aux, ip, table, se, a);
    if (rc) return rc;
    SVGP_REQUIRE(K && Kn && knn, SVGP_ERR_INVALID, "NULL pointer");
    static const int tiles_on = [] { const char* e = getenv("SVGP_SPRITES_KFWD_TILES"); return (e && e[0] == '0') ? 0 : 1; }();
    if (tiles_on) {
        const int nbt = (a.b + 15) / 16, nmt = (a.m + 15) / 16;
        hipLaunchKernelGGL(k_sprites_kernel_fwd_tiles, dim3((unsigned)(nbt * nmt + nmt * nmt) + nb256(a.b)), dim3(256), 0,
                           (hipStream_t)stream, a, nbt, nmt, K, Kn, knn);
    } else {
        const long long tot = (long long)a.b * a.m + (long long)a.m * a.m + a.b;
        hipLaunchKernelGGL(k_sprites_kernel_fwd, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, a, K, Kn, knn);
    }
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// split counts of the tiled reverse pass: about 512 workgroups per mode, at least one chunk of sources per split
static void sprites_bwd_splits(const SpK& a, int* s0, int* s1) {
    const int t0 = (a.m + KB_TT - 1) / KB_TT, t1 = (a.b + KB_TT - 1) / KB_TT;
    const int c0 = (a.b + KB_SC - 1) / KB_SC + (a.m + KB_SC - 1) / KB_SC, c1 = (a.m + KB_SC - 1) / KB_SC;
    int v0 = (512 + t0 - 1) / t0, v1 = (512 + t1 - 1) / t1;
    *s0 = v0 < 1 ? 1 : (v0 > c0 ? c0 : v0);
    *s1 = v1 < 1 ? 1 : (v1 > c1 ? c1 : v1);
}
// template bucket of the tiled reverse pass for feature groups (La, Lc): 1 = (8, 16) (the SPRITES defaults), 2 = (16, 32), 0 = none
static int sprites_bwd_bucket(int La, int Lc) { return (La <= 8 && Lc <= 16) ? 1 : ((La <= 16 && Lc <= 32) ? 2 : 0); }
// doubles of `scratch` one call of svgp_sprites_kernel_matrix_bwd with exactly b rows needs
static long long sprites_bwd_need(int b, int m, int La, int Lc) {
    long long n = (long long)b * La + (long long)(m + b) * 4 + 16;
    const int bk = sprites_bwd_bucket(La, Lc);
    if (bk) {
        SpK a; a.b = b; a.m = m; a.La = La; a.Lc = Lc;
        int s0, s1;
        sprites_bwd_splits(a, &s0, &s1);
        n += ((long long)s0 * m + (long long)s1 * b) * (bk == 1 ? 8 + 16 + 4 : 16 + 32 + 4);
    }
    return n;
}
// doubles of `scratch` for svgp_sprites_kernel_matrix_bwd: cfg.b is the row CAPACITY -- the value covers every call with
// 1 <= b <= cfg.b.  (The need is not monotone in b: the split count of the batch-row targets grows as b shrinks, so s1 * b at
// b < capacity can exceed its value at the capacity -- ADVICE r4: m = 800, capacity 936, b = 720 needed 3 888 doubles more.)
extern "C" long long svgp_sprites_kernel_bwd_scratch_elems(const svgp_sprites_kcfg* c) {
    if (!c || c->b < 1 || c->m < 1 || c->La < 1 || c->Lc < 1) return -1;
    long long n = 0;
    for (int b = 1; b <= c->b; ++b) {
        const long long nb = sprites_bwd_need(b, c->m, c->La, c->Lc);
        n = nb > n ? nb : n;
    }
    return n;
}
// scratch: scratch_elems doubles (svgp_sprites_kernel_bwd_scratch_elems of the row capacity); checked against this call's need.
// Outputs: d_ip (m,La+Lc), d_table (n_act,La), d_char (b,Lc), d_se (4).
extern "C" int svgp_sprites_kernel_matrix_bwd(const svgp_sprites_kcfg* c, const double* aux, const double* ip,
                                              const double* table, const double* se, const double* Kbar,
                                              const double* Knbar, const double* knnbar, double* d_ip, double* d_table,
                                              double* d_char, double* d_se, double* scratch, long long scratch_elems,
                                              void* stream) {
    SpK a;
    int rc = make_spk(c, aux, ip, table, se, a);
    if (rc) return rc;
    SVGP_REQUIRE(Kbar && Knbar && knnbar && d_ip && d_table && d_char && d_se && scratch, SVGP_ERR_INVALID, "NULL pointer");
    SVGP_REQUIRE(scratch_elems >= sprites_bwd_need(a.b, a.m, a.La, a.Lc), SVGP_ERR_INVALID,
                 "scratch too small: size it with svgp_sprites_kernel_bwd_scratch_elems at the row capacity");
    real* d_xa = scratch;
    real* part_se = scratch + (size_t)a.b * a.La;
    hipStream_t st = (hipStream_t)stream;
    static const int tiles_on = [] { const char* e = getenv("SVGP_SPRITES_KBWD_TILES"); return (e && e[0] == '0') ? 0 : 1; }();
    const int bk = tiles_on ? sprites_bwd_bucket(a.La, a.Lc) : 0;
    if (bk) {                                             // the tiled form
        int s0, s1;
        sprites_bwd_splits(a, &s0, &s1);
        const int NV = bk == 1 ? 8 + 16 + 4 : 16 + 32 + 4;
        real* part0 = part_se + (size_t)(a.m + a.b) * 4 + 16;
        real* part1 = part0 + (size_t)s0 * a.m * NV;
        const int t0 = (a.m + KB_TT - 1) / KB_TT, t1 = (a.b + KB_TT - 1) / KB_TT;
        const dim3 grid(t0 > t1 ? t0 : t1, s0 > s1 ? s0 : s1, 2), gridf(nb256((long long)(a.m + a.b) * NV));
        if (bk == 1) {
            hipLaunchKernelGGL((k_sprites_kernel_bwd_tiles<8, 16>), grid, dim3(256), 0, st, a, s0, s1, Kbar, Knbar, part0, part1);
            SVGP_LAUNCH_CHECK();
            hipLaunchKernelGGL((k_sprites_kernel_bwd_finish<8, 16>), gridf, dim3(256), 0, st, a, s0, s1, part0, part1, knnbar, d_ip, d_xa,
                               d_char, part_se);
        } else {
            hipLaunchKernelGGL((k_sprites_kernel_bwd_tiles<16, 32>), grid, dim3(256), 0, st, a, s0, s1, Kbar, Knbar, part0, part1);
            SVGP_LAUNCH_CHECK();
            hipLaunchKernelGGL((k_sprites_kernel_bwd_finish<16, 32>), gridf, dim3(256), 0, st, a, s0, s1, part0, part1, knnbar, d_ip, d_xa,
                               d_char, part_se);
        }
        SVGP_LAUNCH_CHECK();
    } else {
        const size_t dp_ = (size_t)((a.La + a.Lc) | 1);
        if (256 * dp_ * sizeof(real) > 48 * 1024)
            SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sprites_kernel_bwd_cols<SP_MAXD, SP_MAXD>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(256 * dp_ * sizeof(real))));
        hipLaunchKernelGGL((k_sprites_kernel_bwd_cols<SP_MAXD, SP_MAXD>), dim3(a.m), dim3(256), 256 * dp_ * sizeof(real), st, a, Kbar, Knbar, d_ip, part_se);
        SVGP_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_sprites_kernel_bwd_rows<SP_MAXD, SP_MAXD>), dim3(a.b), dim3(64), 64 * dp_ * sizeof(real), st, a, Knbar, knnbar, d_xa, d_char, part_se);
        SVGP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_sprites_kernel_bwd_scatter, dim3((unsigned)a.n_act + 1), dim3(256), 0, st, a, d_xa,
                       part_se, d_table, d_se);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_sprites_aux_fwd(int b, int seg_len, int Lc, const double* repr, const double* action_ids, double* aux,
                                    void* stream) {
    SVGP_REQUIRE(b >= 1 && seg_len >= 1 && b % seg_len == 0 && repr && action_ids && aux, SVGP_ERR_INVALID,
                 "batch must be a multiple of the frames-per-character count (SPRITES_experiment.py:40-41)");
    hipLaunchKernelGGL(k_sprites_aux_fwd, dim3(nb256((long long)b * (1 + Lc))), dim3(256), 0, (hipStream_t)stream, b, seg_len,
                       Lc, repr, action_ids, aux);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_sprites_aux_bwd(int b, int seg_len, int Lc, const double* d_char, double* d_repr, void* stream) {
    SVGP_REQUIRE(b >= 1 && seg_len >= 1 && b % seg_len == 0 && d_char && d_repr, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_sprites_aux_bwd, dim3(nb256((long long)b * Lc)), dim3(256), 0, (hipStream_t)stream, b, seg_len, Lc,
                       d_char, d_repr);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_avgpool_fwd(int n, int HW, int Cc, const double* x, double* y, void* stream) {
    SVGP_REQUIRE(n >= 1 && HW >= 1 && Cc >= 1 && x && y, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_avgpool_fwd<double>, dim3(nb256((long long)n * Cc)), dim3(256), 0, (hipStream_t)stream, n, HW, Cc, x, y);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_avgpool_bwd(int n, int HW, int Cc, const double* dy, double* dx, void* stream) {
    SVGP_REQUIRE(n >= 1 && HW >= 1 && Cc >= 1 && dy && dx, SVGP_ERR_INVALID, "bad argument");
    const long long tot = (long long)n * HW * Cc;
    hipLaunchKernelGGL(k_avgpool_bwd<double>, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, tot, HW, Cc, dy, dx);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_enc_head_fwd(int b, int L, int clip, const double* bias, double* enc, double* mu, double* var_raw,
                                 double* var, void* stream) {
    SVGP_REQUIRE(b >= 1 && L >= 1 && bias && enc && mu && var_raw && var, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_enc_head_fwd, dim3(nb256((long long)b * 2 * L)), dim3(256), 0, (hipStream_t)stream, b, L, clip, bias,
                       enc, mu, var_raw, var);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_enc_head_bwd(int b, int L, int clip, const double* var_raw, const double* ybar, const double* s2bar,
                                 double* d_enc, void* stream) {
    SVGP_REQUIRE(b >= 1 && L >= 1 && var_raw && ybar && s2bar && d_enc, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_enc_head_bwd, dim3(nb256((long long)b * 2 * L)), dim3(256), 0, (hipStream_t)stream, b, L, clip,
                       var_raw, ybar, s2bar, d_enc);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
// element-wise E_{N(mu1,var1)}[log N(.|mu2,var2)] (utils.py:483-504), the stand-alone form of the term the per-sample
// kernels evaluate in place
__global__ void k_gauss_cross_entropy(long long n, const double* __restrict__ mu1, const double* __restrict__ var1,
                                      const double* __restrict__ mu2, const double* __restrict__ var2,
                                      double* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = mu1[i], b = mu2[i], v2 = var2[i];
    out[i] = -0.5 * (1.8378770664093453 + log(v2) + (var1[i] + a * a - 2.0 * a * b + b * b) / v2);
}
extern "C" int svgp_gauss_cross_entropy(long long n, const double* mu1, const double* var1, const double* mu2,
                                        const double* var2, double* out, void* stream) {
    SVGP_REQUIRE(n >= 1 && mu1 && var1 && mu2 && var2 && out, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_gauss_cross_entropy, dim3(nb256(n)), dim3(256), 0, (hipStream_t)stream, n, mu1, var1, mu2, var2, out);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_bias_add(long long rows, int Cc, const double* bias, double* x, void* stream) {
    SVGP_REQUIRE(rows >= 1 && Cc >= 1 && bias && x, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_bias_add<double>, dim3(nb256(rows * Cc)), dim3(256), 0, (hipStream_t)stream, rows * Cc, Cc, bias, x);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
// part_sums: the workspace's partial-sum area; n_part blocks write [blk*4+2]
extern "C" int svgp_sqerr_fwd(long long tot, int n_part, const double* x, const double* xhat, double* part_sums,
                              void* stream) {
    SVGP_REQUIRE(tot >= 1 && n_part >= 1 && x && xhat && part_sums, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_sqerr_fwd<double>, dim3(n_part), dim3(256), 0, (hipStream_t)stream, tot, x, xhat, part_sums);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_sqerr_bwd(long long tot, int geco, int b_global, int n_pix, const double* state, const double* x,
                              const double* xhat, double* dxhat, void* stream) {
    SVGP_REQUIRE(tot >= 1 && b_global >= 1 && n_pix >= 1 && state && x && xhat && dxhat, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_sqerr_bwd<double>, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, tot, geco, 1.0 / b_global,
                       1.0 / n_pix, state, x, xhat, dxhat);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
// ---- float32 instantiations of the per-frame network glue (reference dtype of the SPRITES networks, VAE_utils.py:277);
// the partial sums of the reconstruction error and the device state stay float64
extern "C" int svgp_avgpool_fwd_f32(int n, int HW, int Cc, const float* x, float* y, void* stream) {
    SVGP_REQUIRE(n >= 1 && HW >= 1 && Cc >= 1 && x && y, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_avgpool_fwd<float>, dim3(nb256((long long)n * Cc)), dim3(256), 0, (hipStream_t)stream, n, HW, Cc, x, y);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_avgpool_bwd_f32(int n, int HW, int Cc, const float* dy, float* dx, void* stream) {
    SVGP_REQUIRE(n >= 1 && HW >= 1 && Cc >= 1 && dy && dx, SVGP_ERR_INVALID, "bad argument");
    const long long tot = (long long)n * HW * Cc;
    hipLaunchKernelGGL(k_avgpool_bwd<float>, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, tot, HW, Cc, dy, dx);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_bias_add_f32(long long rows, int Cc, const float* bias, float* x, void* stream) {
    SVGP_REQUIRE(rows >= 1 && Cc >= 1 && bias && x, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_bias_add<float>, dim3(nb256(rows * Cc)), dim3(256), 0, (hipStream_t)stream, rows * Cc, Cc, bias, x);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_sqerr_fwd_f32(long long tot, int n_part, const float* x, const float* xhat, double* part_sums,
                                  void* stream) {
    SVGP_REQUIRE(tot >= 1 && n_part >= 1 && x && xhat && part_sums, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_sqerr_fwd<float>, dim3(n_part), dim3(256), 0, (hipStream_t)stream, tot, x, xhat, part_sums);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_sqerr_bwd_f32(long long tot, int geco, int b_global, int n_pix, const double* state, const float* x,
                                  const float* xhat, float* dxhat, void* stream) {
    SVGP_REQUIRE(tot >= 1 && b_global >= 1 && n_pix >= 1 && state && x && xhat && dxhat, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_sqerr_bwd<float>, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, tot, geco, 1.0 / b_global,
                       1.0 / n_pix, state, x, xhat, dxhat);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
namespace {
// one workgroup per row: numerically stable log-sum-exp, loss_n = lse - logit[label], dlogits = (softmax - onehot) / n
__global__ __launch_bounds__(256) void k_softmax_xent(int n, int C, const real* __restrict__ logits,
                                                      const real* __restrict__ labels, real* __restrict__ row_loss,
                                                      real* __restrict__ dlogits) {
    __shared__ real red[16];
    __shared__ real bc[2];
    const int r = blockIdx.x;
    const real* z = logits + (size_t)r * C;
    real mx = -1e300;
    for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmax(mx, z[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { real t = red[0]; for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, red[w]); bc[0] = t; }
    __syncthreads();
    mx = bc[0];
    real se = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) se += exp(z[c] - mx);
    se = block_sum(se, red);
    if (threadIdx.x == 0) bc[1] = se;
    __syncthreads();
    se = bc[1];
    const int lab = (int)labels[r];
    const real inv_n = real(1) / (real)n;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        dlogits[(size_t)r * C + c] = (exp(z[c] - mx) / se - (c == lab ? real(1) : real(0))) * inv_n;
    if (threadIdx.x == 0) row_loss[r] = log(se) + mx - z[lab];
}
__global__ __launch_bounds__(256) void k_mean_rows(int n, const real* __restrict__ x, real* __restrict__ out) {
    __shared__ real red[16];
    real acc = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += x[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) *out = acc / (real)n;
}
}  // namespace

// tf.reduce_mean(tf.nn.sparse_softmax_cross_entropy_with_logits(labels, logits)) (SPRITES_utils.py:358) and its
// gradient w.r.t. the logits; labels are class ids stored as float64
extern "C" int svgp_softmax_xent(int n, int C, const double* logits, const double* labels, double* row_loss,
                                 double* loss, double* dlogits, void* stream) {
    SVGP_REQUIRE(n >= 1 && C >= 1 && logits && labels && row_loss && loss && dlogits, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_softmax_xent, dim3(n), dim3(256), 0, (hipStream_t)stream, n, C, logits, labels, row_loss, dlogits);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mean_rows, dim3(1), dim3(256), 0, (hipStream_t)stream, n, row_loss, loss);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_clip_by_value(long long tot, double thr, double* g, void* stream) {
    SVGP_REQUIRE(tot >= 0 && thr > 0 && (g || tot == 0), SVGP_ERR_INVALID, "bad argument");
    if (tot == 0) return SVGP_OK;
    hipLaunchKernelGGL(k_clip, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, tot, thr, g);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
