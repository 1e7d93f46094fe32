// In-kernel timeline of the one-launch blocked inverse (k_bgj_flow, linalg.hip) for matrix 0 of a 256 x 17 batch: the wave that owns
// pivot tile k stamps s_memrealtime (100 MHz) at: 0 top of step k-1, 1 operands of other tiles requested, 2 P^-1 flag seen,
// 4 own update done (sweep staging starts), 5 staged, 6 sweep done + P^-1 stored, 7 stores drained + flag raised.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSVGP_FLOW_TRACE -I svgp-vae_amd/csrc tools/micro/flow_trace.hip -o tools/micro/flow_trace
// (stand-alone: includes linalg.hip; svgp_set_error and the Cholesky entry points it references are stubbed here)
#include <cstdarg>
#include <vector>
void svgp_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
#include "linalg.hip"
extern "C" size_t svgp_potrf_workspace_elems(int, int) { return 0; }
extern "C" size_t svgp_potri_workspace_elems(int, int) { return 0; }
int svgp_potri_batched_wide(int, int, double*, const double*, double*, void*) { return 1; }
int svgp_potrf_batched_band(int, int, double*, int, long long, double*, double*, void*) { return 1; }
int main() {
    const int m = 256, batch = 17;
    std::vector<double> h((size_t)batch * m * m);
    unsigned s = 1;
    for (int l = 0; l < batch; ++l)
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                s = s * 1664525u + 1013904223u;
                h[((size_t)l * m + i) * m + j] = (i == j ? 40.0 : 0.0) + 1.0 / (1 + abs(i - j)) + 1e-3 * ((s >> 8) & 0xff) * (i == j);
            }
    double *A, *ld, *w;
    const size_t we = svgp_spd_inverse_workspace_elems(m, batch);
    hipMalloc(&A, h.size() * 8); hipMalloc(&ld, batch * 8); hipMalloc(&w, we * 8);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        if (svgp_spd_inverse_batched(m, batch, A, ld, w, nullptr)) return 1;
        hipDeviceSynchronize();
    }
    unsigned long long t[256];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_flow_trace), sizeof t);
    const unsigned long long t00 = t[0 * 16 + 4];
    for (int k = 0; k < 8; ++k) {
        printf("pivot %d:", k);
        for (int j : {0, 1, 2, 4, 5, 6, 7}) printf("  m%d %7.2f", j, t[k * 16 + j] ? (double)(t[k * 16 + j] - t00) / 100.0 : -1.0);
        printf("   us\n");
    }
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_flow_trace_tile), sizeof t);
    printf("tile (3, 5): 0 top of step, 1 row / column operands requested, 2 P^-1 flag seen, 3 P^-1 requested, 4 update issued, 5 published\n");
    for (int k = 0; k < 8; ++k) {
        printf("step %d:", k);
        for (int j : {0, 1, 2, 3, 4, 5}) printf("  m%d %7.2f", j, t[k * 16 + j] ? (double)(t[k * 16 + j] - t00) / 100.0 : -1.0);
        printf("   us\n");
    }
    return 0;
}
