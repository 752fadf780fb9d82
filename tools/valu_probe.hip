// valu_probe.hip - FP64 VALU issue rates on gfx950: v_fma_f64 vs v_mul_f64 / v_add_f64 (the
// bit-exact deterministic sweep may not fuse), 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WHICH>
__global__ __launch_bounds__(256) void k(int iters, double* sink) {
    double v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3 + k;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (WHICH == 0) v[k] = __builtin_fma(v[k], b, a);
                if (WHICH == 1) { v[k] = v[k] * b; }
                if (WHICH == 2) { v[k] = v[k] + a; }
                if (WHICH == 3) { v[k] = v[k] * b; v[k] = v[k] + a; }
            }
    }
    double s = 0; for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 12345.678) sink[0] = s;
}
template <int WHICH> void run(const char* name, int per_cu) {
    double* sink; hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * per_cu;
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 32 * (WHICH == 3 ? 2 : 1);
    const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * per_cu);   // per SIMD: per_cu waves per SIMD
    printf("%-12s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, per_cu, ms, cyc);
}
int main() {
    for (int pc : {1, 2, 4}) { run<0>("v_fma_f64", pc); run<1>("v_mul_f64", pc); run<2>("v_add_f64", pc); run<3>("mul+add", pc); }
    return 0;
}
