#include <hip/hip_runtime.h>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const u2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const u2 h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    v += dpp_mov<0x128>(v);                    // row_ror:8  == lane ^ 8
    v += dpp_mov<0x1B>(dpp_mov<0x141>(v));     // row_half_mirror then quad_perm [3,2,1,0] == lane ^ 4
    v += dpp_mov<0x4E>(v);                     // quad_perm [2,3,0,1] == lane ^ 2
    v += dpp_mov<0xB1>(v);                     // quad_perm [1,0,3,2] == lane ^ 1
    return v;
}
__device__ __forceinline__ double wave_sum_ref(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void probe(const double* in, double* a, double* b) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    a[i] = wave_sum_dpp(in[i]);
    b[i] = wave_sum_ref(in[i]);
}
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() {
    const int N = 64 * 4096;
    double *h = (double*)malloc(N * 8), *ha = (double*)malloc(N * 8), *hb = (double*)malloc(N * 8);
    srand(7);
    for (int i = 0; i < N; ++i) h[i] = ((double)rand() / RAND_MAX - 0.5) * pow(10.0, (rand() % 12) - 6);
    double *d, *da, *db;
    hipMalloc(&d, N * 8); hipMalloc(&da, N * 8); hipMalloc(&db, N * 8);
    hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
    probe<<<N / 64, 64>>>(d, da, db);
    hipMemcpy(ha, da, N * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, db, N * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) bad += memcmp(&ha[i], &hb[i], 8) != 0;
    printf("wave_sum dpp vs shfl_xor butterfly: %d of %d lanes differ (first: %.17g vs %.17g)\n", bad, N, ha[0], hb[0]);
    return bad != 0;
}
