// How expensive is COLD straight-line code on gfx950?  One wave per workgroup runs an unrolled chain of N dependent v_fma_f64 (8-byte
// instructions: N * 8 bytes of code) three times and stamps each pass with the 100 MHz s_memtime: pass 0 fetches the instructions
// for the first time, passes 1 and 2 run from the instruction cache.  Build: hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N>
__device__ __forceinline__ double chain(double x, double a, double b) {
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
    return x;
}
template <int N>
__global__ void probe(double* out, unsigned long long* t, double a, double b, int passes) {
    double x = threadIdx.x;
    for (int p = 0; p < passes; ++p) {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x) : : "memory");   // the stamps are pinned to x: in-order issue
        x = chain<N>(x, a, b);
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x) : : "memory");
        if (threadIdx.x == 0) t[blockIdx.x * 4 + p] = t1 - t0;
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
template <int N>
void run(int blocks) {
    double* out;
    unsigned long long* t;
    hipMalloc(&out, blocks * 64 * 8);
    hipMalloc(&t, blocks * 4 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(64), 0, 0, out, t, 1.0000001, 1e-9, 3);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), t, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double s[3] = {0, 0, 0};
        for (int b = 0; b < blocks; ++b)
            for (int p = 0; p < 3; ++p) s[p] += (double)h[b * 4 + p];
        printf("N %5d (%3d KB code) blocks %4d launch %d: pass0 %.2f us  pass1 %.2f us  pass2 %.2f us  (%.1f / %.1f ns per instruction)\n", N, N * 8 / 1024, blocks, rep,
               s[0] / blocks * 0.01, s[1] / blocks * 0.01, s[2] / blocks * 0.01, s[0] / blocks * 10.0 / N, s[1] / blocks * 10.0 / N);
    }
    hipFree(out);
    hipFree(t);
}
int main() {
    run<512>(256);
    run<2048>(256);
    run<4096>(256);
    run<4096>(1024);
    run<4096>(8);
    run<8192>(256);
    return 0;
}
