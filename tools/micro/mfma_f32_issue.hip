// Issue-rate experiments for v_mfma_f32_16x16x4_f32 on gfx950 in the shape of the direct convolution loop (tuning aid, not part of
// the library): 36 MFMAs per "row" on NACC accumulator chains, A operands = 36 weight registers, B operands = 36 input registers.
//   MODE 0  MFMAs only              MODE 1  + the 9 x 4-register rotation (v_mov) per row        MODE 2  + exp epilogue + 16-byte store
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f32_issue mfma_f32_issue.hip ; run: ./mfma_f32_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(int rows, float* out, const float* in) {
    const int tid = threadIdx.x;
    float w[36];
    f4 b[9];
    for (int i = 0; i < 36; ++i) w[i] = in[tid + i];
    for (int i = 0; i < 9; ++i) b[i] = *reinterpret_cast<const f4*>(in + 4 * tid + 1024 * i);
    f4 tot = {0, 0, 0, 0};
    for (int y = 0; y < rows; ++y) {
        f4 acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = f4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[(4 * t + j) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * t + j], b[t][j], acc[(4 * t + j) % NACC], 0, 0, 0);
        f4 e = acc[0];
#pragma unroll
        for (int a = 1; a < NACC; ++a) e += acc[a];
        if (MODE >= 1) {
            const f4 t0 = b[0], t1 = b[1], t2 = b[2];
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = b[i + 3];
            b[6] = t0 + e; b[7] = t1; b[8] = t2;
        }
        if (MODE >= 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) e[g] = e[g] > 0 ? e[g] : expf(e[g]) - 1.0f;
            *reinterpret_cast<f4*>(out + ((size_t)(blockIdx.x * rows + y) * 256 + tid) * 4) = e;
        } else {
            tot += e;
        }
    }
    if (MODE < 2) *reinterpret_cast<f4*>(out + (size_t)(blockIdx.x * 256 + tid) * 4) = tot;
}

template <int NACC, int MODE> void run(int wgs, int rows, float* out, const float* in) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<NACC, MODE>), dim3(wgs), dim3(256), 0, 0, rows, out, in);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 10.0 * wgs * 4.0 * rows * 36 * 2048.0;
    printf("NACC=%d mode=%d wgs=%5d (%.0f waves/SIMD) rows=%d: %8.1f us/launch  %6.1f TFLOP/s\n", NACC, MODE, wgs, wgs / 256.0, rows, ms * 100.0,
           flops / (ms * 1e-3) * 1e-12);
}

int main() {
    float *out, *in;
    hipMalloc(&out, (size_t)1024 << 20); hipMalloc(&in, 1 << 20);
    std::vector<float> h(262144, 1e-3f);
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int wgs : {256, 512, 1024}) {
        const int rows = 64 * 1024 / wgs;
        run<1, 0>(wgs, rows, out, in); run<2, 0>(wgs, rows, out, in); run<4, 0>(wgs, rows, out, in);
        run<2, 1>(wgs, rows, out, in); run<4, 1>(wgs, rows, out, in);
        run<2, 2>(wgs, rows, out, in); run<4, 2>(wgs, rows, out, in);
    }
    return 0;
}
