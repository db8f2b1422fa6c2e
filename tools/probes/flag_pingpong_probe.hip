// How long does a flag hand-over between two WORKGROUPS take on gfx950 (possibly on different XCDs)?  Two single-wave workgroups
// play ping-pong through two global words: store + release fence on one side, acquire spin on the other, 2000 round trips.
// With payload: the sender also writes 256 B before the release and the receiver reads them after the acquire.
// hipcc --offload-arch=gfx950 -O3 -o flag_pingpong_probe flag_pingpong_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void pingpong(unsigned int* flags, double* payload, unsigned long long* out, int rounds, int with_payload) {
    const int me = blockIdx.x, lane = threadIdx.x;
    unsigned int* mine = flags + 64 * me;        // separate cache lines
    unsigned int* other = flags + 64 * (1 - me);
    double acc = 0.0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            if (with_payload && lane < 32) payload[lane] = r + lane;
            __threadfence();
            if (lane == 0) __hip_atomic_store(mine, (unsigned int)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(other, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)r) __builtin_amdgcn_s_sleep(1);
            __threadfence();
            if (with_payload && lane < 32) acc += payload[32 + lane];
        } else {
            while (__hip_atomic_load(other, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)r) __builtin_amdgcn_s_sleep(1);
            __threadfence();
            if (with_payload && lane < 32) acc += payload[lane], payload[32 + lane] = acc;
            __threadfence();
            if (lane == 0) __hip_atomic_store(mine, (unsigned int)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) out[me] = t1 - t0;
    if (acc == 12345.678) out[2] = 1;
}
int main() {
    unsigned int* flags;
    double* payload;
    unsigned long long *out, h[3];
    hipMalloc(&flags, 128 * 4);
    hipMalloc(&payload, 64 * 8);
    hipMalloc(&out, 3 * 8);
    for (int wp = 0; wp < 2; ++wp)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(flags, 0, 128 * 4);
            hipLaunchKernelGGL(pingpong, dim3(2), dim3(64), 0, 0, flags, payload, out, 2000, wp);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            printf("payload %d: round trip %.2f us (one hand-over %.2f us)\n", wp, h[0] * 0.01 / 2000, h[0] * 0.01 / 4000);
        }
    return 0;
}
