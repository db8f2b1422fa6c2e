// Probe of v_mfma_f64_4x4x4_4b_f64's operand layout on gfx950: which (block, i, k) / (block, k, j) / (block, i, j) each lane holds.
// A and B are filled with distinct tags; the product is compared against all candidate index maps on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double* a, const double* b, double* d) {
    const int lane = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], acc, 0, 0, 0);
    d[lane] = acc;
}
int main() {
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    hipMalloc(&da, 512), hipMalloc(&db, 512), hipMalloc(&dd, 512);
    // one-hot probing: set A[la] = 1, B[lb] = 1, see which output lanes become 1
    int amap_i[64], amap_k[64], amap_b[64];
    printf("pairs (la, lb) -> output lanes with value 1 (only la,lb < 64 sampled on a grid)\n");
    for (int la = 0; la < 64; la += 1) {
        for (int lb = 0; lb < 64; lb += 1) {
            for (int i = 0; i < 64; ++i) ha[i] = hb[i] = 0.0;
            ha[la] = 1.0, hb[lb] = 1.0;
            hipMemcpy(da, ha, 512, hipMemcpyHostToDevice), hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
            hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
            for (int o = 0; o < 64; ++o)
                if (hd[o] != 0.0) printf("A@%d x B@%d -> D@%d\n", la, lb, o);
        }
    }
    return 0;
}
