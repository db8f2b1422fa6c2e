// readlane_gather_probe.hip - the gather at the end of wave_sum_n<21> (lk_map_kernels.h) in its two forms, v_readlane (42 scalar results)
// and ds_bpermute (__shfl), under the conditions the round-3 failure was seen in: scalar-register pressure (SGPR spills), a loop the
// compiler must treat as divergent although every lane runs the same trips (t = threadIdx.x >> 6), and - as a control - a truly partial
// EXEC mask.  Prints, per mode, how many of the 21 x 64 x waves results differ between the two forms and from a host sum.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o readlane_gather_probe readlane_gather_probe.hip && ./readlane_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N, bool RL>
__device__ __forceinline__ void wsn(double* v) {
    const int lane = threadIdx.x & 63;
    constexpr int H5 = (N + 1) / 2, H4 = (H5 + 1) / 2, H3 = (H4 + 1) / 2, H2 = (H3 + 1) / 2, H1 = (H2 + 1) / 2, H0 = (H1 + 1) / 2;
    auto step = [&](const int n, const int half, const int mask) {
        const bool hi = (lane & mask) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const double a = v[j], b = (half + j < n) ? v[half + j] : 0.0;
            v[j] = (hi ? b : a) + __shfl_xor(hi ? a : b, mask, 64);
        }
    };
    step(N, H5, 32), step(H5, H4, 16), step(H4, H3, 8), step(H3, H2, 4), step(H2, H1, 2), step(H1, H0, 1);
    double r[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        int rem = c, src = 0;
        if (rem >= H5) rem -= H5, src |= 32;
        if (rem >= H4) rem -= H4, src |= 16;
        if (rem >= H3) rem -= H3, src |= 8;
        if (rem >= H2) rem -= H2, src |= 4;
        if (rem >= H1) rem -= H1, src |= 2;
        if (rem >= H0) rem -= H0, src |= 1;
        if (RL) r[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v[0]), src), __builtin_amdgcn_readlane(__double2loint(v[0]), src));
        else r[c] = __shfl(v[0], src, 64);
    }
#pragma unroll
    for (int c = 0; c < N; ++c) v[c] = r[c];
}
template <bool RL>
__global__ void __launch_bounds__(256) probe(const double* in, const double* uni, double* out, int mode, int reps) {
    const int lane = threadIdx.x & 63, nw = (gridDim.x * 256) >> 6;
    double s[64];   // mode >= 1: 64 wave-uniform doubles kept live across the gather (SGPR pressure -> spills)
    if (mode >= 1)
#pragma unroll
        for (int k = 0; k < 64; ++k) s[k] = uni[k];
    for (int t = (blockIdx.x * 256 + threadIdx.x) >> 6; t < reps; t += nw) {   // "divergent" to the compiler, uniform at run time
        if (mode == 3 && lane >= 40) continue;
        double v[21];
#pragma unroll
        for (int c = 0; c < 21; ++c) v[c] = in[((size_t)t * 64 + lane) * 21 + c];
        wsn<21, RL>(v);
        double extra = 0.0;
        if (mode >= 1)
#pragma unroll
            for (int k = 0; k < 64; ++k) extra += s[k] * v[k % 21];
#pragma unroll
        for (int c = 0; c < 21; ++c) out[((size_t)t * 64 + lane) * 21 + c] = v[c] + (mode >= 1 ? 0.0 * extra : 0.0);
    }
}
int main() {
    const int reps = 4096;
    const size_t n = (size_t)reps * 64 * 21;
    std::vector<double> h(n), u(64), a(n), b(n);
    for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000003u) / 1000003.0 - 0.5;
    for (int k = 0; k < 64; ++k) u[k] = 1.0 + k;
    double *d_in, *d_u, *d_a, *d_b;
    hipMalloc(&d_in, n * 8), hipMalloc(&d_u, 64 * 8), hipMalloc(&d_a, n * 8), hipMalloc(&d_b, n * 8);
    hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice), hipMemcpy(d_u, u.data(), 64 * 8, hipMemcpyHostToDevice);
    const char* names[4] = {"uniform EXEC, no pressure", "64 live scalars across the gather", "same, 512 workgroups", "lanes >= 40 inactive (control)"};
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(d_a, 0, n * 8), hipMemset(d_b, 0, n * 8);
        const int grid = mode == 2 ? 512 : 64;
        hipLaunchKernelGGL(probe<true>, dim3(grid), dim3(256), 0, 0, d_in, d_u, d_a, mode, reps);
        hipLaunchKernelGGL(probe<false>, dim3(grid), dim3(256), 0, 0, d_in, d_u, d_b, mode, reps);
        hipDeviceSynchronize();
        hipMemcpy(a.data(), d_a, n * 8, hipMemcpyDeviceToHost), hipMemcpy(b.data(), d_b, n * 8, hipMemcpyDeviceToHost);
        size_t diff = 0, bad_rl = 0, bad_sh = 0;
        for (int t = 0; t < reps; ++t)
            for (int c = 0; c < 21; ++c) {
                double ref = 0.0;
                for (int l = 0; l < (mode == 3 ? 40 : 64); ++l) ref += h[((size_t)t * 64 + l) * 21 + c];
                for (int l = 0; l < (mode == 3 ? 40 : 64); ++l) {
                    const size_t i = ((size_t)t * 64 + l) * 21 + c;
                    diff += a[i] != b[i];
                    bad_rl += !(fabs(a[i] - ref) < 1e-9);
                    bad_sh += !(fabs(b[i] - ref) < 1e-9);
                }
            }
        printf("mode %d (%s): readlane != shfl in %zu results; wrong vs host sum: readlane %zu, shfl %zu (of %zu)\n", mode, names[mode], diff, bad_rl, bad_sh,
               (size_t)reps * 21 * (mode == 3 ? 40 : 64));
    }
    return 0;
}
