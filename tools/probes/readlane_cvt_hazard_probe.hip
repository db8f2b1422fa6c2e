// readlane_cvt_hazard_probe.hip - the instruction sequence of the round-3 wrong-result build (lk_insert_root_kernel<true> at commit
// bdcb3a3 with the v_readlane gather, tools/probes/README_readlane.md), replayed verbatim with fixed registers:
//     v_add_f64 v[88:89], v[50:51], v[88:89] ; s_nop 0 ; v_readlane_b32 s43, v89, 0 ; v_readlane_b32 s42, v88, 0 ;
//     v_cvt_f64_i32 v[50:51], v17 ; v_readlane_b32 s59, v89, 4 ; v_div_scale_f64 v[90:91], s[2:3], v[50:51], v[50:51], s[42:43] ; ...
// against the order every passing build has (the convert ahead of the readlanes).  Every lane must end with v[50:51] == (double)cnt and
// the same v_div_scale result in both orders.
//   hipcc --offload-arch=gfx950 -O3 -o readlane_cvt_hazard_probe readlane_cvt_hazard_probe.hip && ./readlane_cvt_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define PRE                                                                                                              \
    "v_mov_b32 v88, %[lo]\n v_mov_b32 v89, %[hi]\n v_mov_b32 v50, %[wlo]\n v_mov_b32 v51, %[whi]\n v_mov_b32 v17, %[cnt]\n" \
    "s_nop 4\n v_add_f64 v[88:89], v[50:51], v[88:89]\n s_nop 0\n"
#define POST                                                                                              \
    "v_rcp_f64_e32 v[92:93], v[90:91]\n v_readlane_b32 s58, v88, 4\n v_readlane_b32 s61, v89, 8\n s_nop 4\n" \
    "v_mov_b32 %[o0], v50\n v_mov_b32 %[o1], v51\n v_mov_b32 %[o2], v90\n v_mov_b32 %[o3], v91\n v_mov_b32 %[o4], s42\n v_mov_b32 %[o5], s43\n"
#define IO                                                                                                                                       \
    : [o0] "=v"(o[0]), [o1] "=v"(o[1]), [o2] "=v"(o[2]), [o3] "=v"(o[3]), [o4] "=v"(o[4]), [o5] "=v"(o[5])                                        \
    : [lo] "v"(lo), [hi] "v"(hi), [wlo] "v"(wlo), [whi] "v"(whi), [cnt] "v"(cnt)                                                                  \
    : "v17", "v50", "v51", "v88", "v89", "v90", "v91", "v92", "v93", "s2", "s3", "s42", "s43", "s58", "s59", "s61", "vcc", "memory"
__global__ void probe(const double* a, const double* w, const int* n, unsigned int* out, int variant) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const unsigned int lo = (unsigned int)__double2loint(a[i]), hi = (unsigned int)__double2hiint(a[i]);
    const unsigned int wlo = (unsigned int)__double2loint(w[i]), whi = (unsigned int)__double2hiint(w[i]);
    const int cnt = n[blockIdx.x];
    unsigned int o[6];
    if (variant == 0)   // the failing build's order
        asm volatile(PRE "v_readlane_b32 s43, v89, 0\n v_readlane_b32 s42, v88, 0\n v_cvt_f64_i32_e32 v[50:51], v17\n v_readlane_b32 s59, v89, 4\n"
                         "v_div_scale_f64 v[90:91], s[2:3], v[50:51], v[50:51], s[42:43]\n" POST IO);
    else                // the passing builds' order
        asm volatile(PRE "v_cvt_f64_i32_e32 v[50:51], v17\n v_readlane_b32 s43, v89, 0\n v_readlane_b32 s42, v88, 0\n v_readlane_b32 s59, v89, 4\n v_readlane_b32 s58, v88, 4\n"
                         "v_div_scale_f64 v[90:91], s[2:3], v[50:51], v[50:51], s[42:43]\n" POST IO);
    for (int k = 0; k < 6; ++k) out[(size_t)i * 6 + k] = o[k];
}
int main() {
    const int blocks = 4096, n = blocks * 64;
    std::vector<double> a(n), w(n);
    std::vector<int> c(blocks);
    for (int i = 0; i < n; ++i) a[i] = 1.0 + (i % 977) * 1e-3, w[i] = -3.4e-6 * (1 + i % 13);
    for (int b = 0; b < blocks; ++b) c[b] = 6 + b % 45;
    double *da, *dw;
    int* dc;
    unsigned int* dout;
    hipMalloc(&da, n * 8), hipMalloc(&dw, n * 8), hipMalloc(&dc, blocks * 4), hipMalloc(&dout, (size_t)n * 24);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice), hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice), hipMemcpy(dc, c.data(), blocks * 4, hipMemcpyHostToDevice);
    std::vector<unsigned int> r[2];
    for (int v = 0; v < 2; ++v) {
        r[v].resize((size_t)n * 6);
        hipMemset(dout, 0xff, (size_t)n * 24);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, da, dw, dc, dout, v);
        hipDeviceSynchronize();
        hipMemcpy(r[v].data(), dout, (size_t)n * 24, hipMemcpyDeviceToHost);
    }
    size_t bad_cvt[2] = {0, 0}, diff_scale = 0, bad_sgpr[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const double want = (double)c[i / 64];
        const double lane0 = a[(i / 64) * 64] + w[(i / 64) * 64];
        unsigned long long l0;
        memcpy(&l0, &lane0, 8);
        for (int v = 0; v < 2; ++v) {
            unsigned long long bits = ((unsigned long long)r[v][(size_t)i * 6 + 1] << 32) | r[v][(size_t)i * 6];
            double got;
            memcpy(&got, &bits, 8);
            bad_cvt[v] += got != want;
            bad_sgpr[v] += (((unsigned long long)r[v][(size_t)i * 6 + 5] << 32) | r[v][(size_t)i * 6 + 4]) != l0;
        }
        diff_scale += r[0][(size_t)i * 6 + 2] != r[1][(size_t)i * 6 + 2] || r[0][(size_t)i * 6 + 3] != r[1][(size_t)i * 6 + 3];
    }
    printf("failing build's order : v_cvt result wrong in %zu of %d lanes, readlane pair wrong in %zu\n", bad_cvt[0], n, bad_sgpr[0]);
    printf("passing builds' order : v_cvt result wrong in %zu of %d lanes, readlane pair wrong in %zu\n", bad_cvt[1], n, bad_sgpr[1]);
    printf("v_div_scale results differ between the two orders in %zu of %d lanes\n", diff_scale, n);
    return 0;
}
