// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, checked against the layout the covariance update assumes:
//   A (16 x 4):  lane l holds A[l % 16][l / 16]        B (4 x 16):  lane l holds B[l / 16][l % 16]
//   C / D (16 x 16), four doubles per lane:  d[v] of lane l = D[4 * (l / 16) + v][l % 16]  or  D[l / 16 + 4 v][l % 16] (both tried)
// Random A, B, C; D = C + A B on the host; prints the number of mismatching entries (0 expected).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_16x16x4_layout mfma_f64_16x16x4_layout.hip && ./mfma_f64_16x16x4_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ int drow(int l, int v, int mode) { return mode == 0 ? 4 * (l / 16) + v : (l / 16) + 4 * v; }
__global__ void probe(const double* a, const double* b, const double* c, double* d, int mode) {
    const int l = threadIdx.x;
    double4_t acc;
    for (int v = 0; v < 4; ++v) acc[v] = c[drow(l, v, mode) * 16 + (l % 16)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(l % 16) * 4 + l / 16], b[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[drow(l, v, mode) * 16 + (l % 16)] = acc[v];
}
int main() {
    double ha[64], hb[64], hc[256], hd[256], *da, *db, *dc, *dd;
    for (int i = 0; i < 64; ++i) ha[i] = sin(1.0 + i), hb[i] = cos(2.0 + 3 * i);
    for (int i = 0; i < 256; ++i) hc[i] = 0.01 * i;
    (void)hipMalloc(&da, 512), (void)hipMalloc(&db, 512), (void)hipMalloc(&dc, 2048), (void)hipMalloc(&dd, 2048);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice), (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice), (void)hipMemcpy(dc, hc, 2048, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd, mode);
    (void)hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    double worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double r = hc[i * 16 + j];
            for (int k = 0; k < 4; ++k) r += ha[i * 4 + k] * hb[k * 16 + j];
            const double e = fabs(r - hd[i * 16 + j]);
            worst = e > worst ? e : worst;
            bad += e > 1e-13;
        }
    printf("v_mfma_f64_16x16x4_f64, d[v] of lane l = D[%s][l %% 16]: %d of 256 entries differ from C + A B (worst %.3g)\n", mode == 0 ? "4 (l / 16) + v" : "l / 16 + 4 v", bad, worst);
  }
    return 0;
}
