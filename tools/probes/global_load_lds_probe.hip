// Where does global_load_lds_dwordx4 (gfx950) put a lane's 16 bytes?  Expected: LDS[base + 16 * lane].  Each lane loads src[64 + lane]
// straight into LDS (no VGPR for the data), the wave waits (vmcnt) and reads buf[lane] back.
// hipcc --offload-arch=gfx950 -O3 -o global_load_lds_probe global_load_lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* __restrict__ src, float4* dst) {
    __shared__ float4 pad[8];     // a non-zero base address
    __shared__ float4 buf[64];
    const int lane = threadIdx.x;
    pad[lane & 7] = make_float4(0, 0, 0, 0);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 64 + lane), (void __attribute__((address_space(3)))*)buf, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[lane] = buf[lane];
    if (lane < 8) dst[64 + lane] = pad[lane];
}
int main() {
    float4 h[128], o[72];
    for (int i = 0; i < 128; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *d, *e;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += !(o[i].x == h[64 + i].x && o[i].y == h[64 + i].y && o[i].z == h[64 + i].z && o[i].w == h[64 + i].w);
    printf("global_load_lds_dwordx4: %d of 64 lanes landed elsewhere; lane 5 got (%g %g %g %g)\n", bad, o[5].x, o[5].y, o[5].z, o[5].w);
    return bad != 0;
}
