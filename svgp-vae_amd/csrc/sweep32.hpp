// Single-wave in-register Gauss-Jordan sweep of a 32 x 32 SPD block (no pivoting), shared by the LDS-resident GP kernels
// (gp_kernels.hip: K^-1, Sigma_l^-1, (A_l + jI)^-1 at m <= 32) and the pivot blocks of the blocked inverse (linalg.hip).
//
// Layout (round 5): the wave is 4 DPP rows x 16 lanes; lane (R = lane >> 4, c = lane & 15) holds rows 8R .. 8R+7 of the
// two columns 2c, 2c+1 -- a[8][2].  Per pivot k every lane needs
//   * the pivot-column values of its 8 rows, A[8R + r][k]: they sit in lane k/2 of the SAME DPP row -> `row_newbcast`
//     (a VALU move, no LDS crossbar, no wait) -- 8 values;
//   * the pivot-row values of its 2 columns, A[k][2c + j]: they sit in DPP row k/8, same c -> ds_bpermute -- 2 values;
//   * the pivot itself: v_readlane.
// The previous layout (8 x 8 lanes of 4 x 4 blocks) moved 4 + 4 values per pivot through ds_bpermute (16 dword permutes and
// their waits at the head of every step's dependency chain).  The arithmetic per matrix element is unchanged
// (a_ij <- a_ij - a_ik (a_kj / a_kk), the pivot row scaled by the Newton-refined reciprocal): results are bit-identical to
// the 8 x 8 form, tools/micro/sweep32_probe.hip checks that and times both: 6.03 -> 4.39 us per sweep alone on the chip,
// 7.85 -> 5.1 us with 1 024 workgroups sweeping at once.  A single wave issues one instruction per ~5.8 cycles here whatever
// the mix, so the time IS the instruction count: 78 -> 54 per pivot (20 FMA incl. the reciprocal's Newton steps, 10 mul,
// 8 v_mov_b64_dpp, 6 v_cndmask, 4 ds_bpermute, 3 v_readlane, 2 waits, 1 rcp).  Measured and not kept: the row broadcast folded
// into v_fmac_f64_dpp by inline asm (the pivot column's own half still needs the 8 moves, + s_nop / sign flip: 57 per pivot).
#pragma once
#include <hip/hip_runtime.h>

namespace sweep32 {

typedef double real;

__device__ __forceinline__ real rcp_refined(real x) {
    real r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, real(1)), r, r);
    r = fma(fma(-x, r, real(1)), r, r);
    return r;
}
__device__ __forceinline__ real readlane_f64(real v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
//  the 64-bit DPP move with row_newbcast exists on gfx90a / gfx94x / gfx950 only; this library is written for gfx950 and
// every m <= 32 inverse of the step goes through it: any other --offload-arch must fail HERE, loudly, not miscompile
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx90a__) || defined(__gfx940__) || defined(__gfx941__) || defined(__gfx942__) || \
                                        defined(__gfx950__))
#error "sweep32.hpp: v_mov_b64_dpp row_newbcast needs gfx90a / gfx94x / gfx950 (build with --offload-arch=gfx950)"
#endif
// value of lane SRC (0..15) of the caller's own 16-lane DPP row: ONE v_mov_b64_dpp (gfx90a+: 64-bit DPP with row_newbcast)
template <int SRC>
__device__ __forceinline__ real row_bcast_f64(real v) {
    return __builtin_amdgcn_update_dpp(real(0), v, 0x150 + SRC, 0xf, 0xf, true);
}

template <int K>
__device__ __forceinline__ void step(real (&a)[8][2], real& mypiv, int lane) {
    constexpr int kq = K / 8, kr = K % 8, kc = K / 2, kp = K % 2;
    const int R = lane >> 4, c16 = lane & 15;
    const real piv = readlane_f64(a[kr][kp], kq * 16 + kc);
    real rowk[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) rowk[c] = __shfl(a[kr][c], kq * 16 + c16, 64);
    const real ipiv = rcp_refined(piv);
    if (lane == K) mypiv = piv;
    const bool jl = (c16 == kc), il = (R == kq);
    real rkj[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) rkj[c] = (c == kp && jl) ? ipiv : rowk[c] * ipiv;
    // the pivot column's own lanes start from 0: a * keep with keep = 0 / 1 is ONE instruction where a 64-bit select is two
    // (a * 1 = a exactly; a * 0 = +-0, and (+-0) - colk rkj = -colk rkj exactly as 0 - colk rkj)
    const real keep = jl ? real(0) : real(1);
    constexpr int co = 1 - kp;
    real colk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) colk[r] = row_bcast_f64<kc>(a[r][kp]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const real v0 = fma(-colk[r], rkj[co], a[r][co]);
        a[r][co] = (r == kr && il) ? rkj[co] : v0;
        const real kept = a[r][kp] * keep;                    // (a separate rounding-free product: the FMA below must be
        const real v1 = fma(-colk[r], rkj[kp], kept);         //  -colk rkj + a, not a keep - round(colk rkj))
        a[r][kp] = (r == kr && il) ? rkj[kp] : v1;
    }
}

template <int K0, int K1>
struct Steps {
    static __device__ __forceinline__ void run(real (&a)[8][2], real& mypiv, int lane, int m) {
        if (K0 < m) step<K0>(a, mypiv, lane);          // (m: wave-uniform; rows / columns >= m are an identity pad)
        Steps<K0 + 1, K1>::run(a, mypiv, lane, m);
    }
};
template <int K1>
struct Steps<K1, K1> {
    static __device__ __forceinline__ void run(real (&)[8][2], real&, int, int) {}
};

// One wave (lane = 0..63 of the calling wave; every lane of it must call).  get(i, j) -> element (i, j) of the block (the
// caller supplies the identity pad beyond m), put(i, j, v) receives the inverse; returns this lane's pivot (lane k < 32: pivot k,
// others 1) -- log det = sum over the wave of log(pivot).
template <typename Get, typename Put>
__device__ __forceinline__ real gauss_jordan_32(int lane, int m, Get get, Put put) {
    const int R = lane >> 4, c16 = lane & 15;
    real a[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) a[r][c] = get(8 * R + r, 2 * c16 + c);
    real mypiv = 1;
    Steps<0, 32>::run(a, mypiv, lane, m);
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) put(8 * R + r, 2 * c16 + c, a[r][c]);
    return mypiv;
}

}  // namespace sweep32
