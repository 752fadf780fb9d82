#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, double* out, int shift) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = -1.0;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(size_t)(smem + 128) ;   // LDS byte address (address space cast below)
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)(smem + 128 + shift));
    const unsigned voff = lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:0"
                 :: "s"(lds_base), "v"(voff), "s"(src) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = smem[i];
    (void)lds_addr;
}
int main() {
    std::vector<double> h(128), o(1024);
    for (int i = 0; i < 128; ++i) h[i] = i;
    double *d, *od;
    hipMalloc(&d, 128 * 8); hipMalloc(&od, 1024 * 8);
    hipMemcpy(d, h.data(), 128 * 8, hipMemcpyHostToDevice);
    for (int shift : {0, 6}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, od, shift);
        hipMemcpy(o.data(), od, 1024 * 8, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < 1024; ++i) {
            double want = (i >= 128 + shift && i < 256 + shift) ? (double)(i - 128 - shift) : -1.0;
            if (o[i] != want) { ++bad; if (first < 0) first = i; }
        }
        printf("shift %d: %d mismatches (first %d: got %g)\n", shift, bad, first, first >= 0 ? o[first] : 0.0);
    }
    return 0;
}
