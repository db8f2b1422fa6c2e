// lk_pre_kernels.h — the two steps that feed the path, kept on the device so that a raw scan
// never returns to the host between decode and the ESKF update:
//   pcl::VoxelGrid centroid filter as used at KILO.cc:356-360  -> cell index, stable radix sort by cell, one thread
//       per cell sums its points sequentially in input order in float32 (all four fields incl. curvature)
//   std::sort by curvature at KILO.cc:369-370                   -> stable radix sort on the order-preserving bit image
// Definition (PCL leaves the order inside a cell and the output order undefined): oracle/preprocess_oracle.py.
// HBM-bound streaming kernels: one lk_point (16 B, float4) per lane, coalesced.
#pragma once
#include "lk_device.h"

__device__ __forceinline__ int lk_f2ord(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : (i ^ 0x7fffffff);
}
__device__ __forceinline__ float lk_ord2f(int i) { return __int_as_float(i >= 0 ? i : (i ^ 0x7fffffff)); }

// mm[0..2] = min x,y,z ; mm[3..5] = max x,y,z   (order-preserving int image; init: INT_MAX / INT_MIN)
__global__ void __launch_bounds__(256) lk_pre_minmax_kernel(const lk_point* __restrict__ pts, int n, int* mm) {
    __shared__ int smin[3][4], smax[3][4];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = reinterpret_cast<const float4*>(pts)[i];
        const int v[3] = {lk_f2ord(p.x), lk_f2ord(p.y), lk_f2ord(p.z)};
#pragma unroll
        for (int c = 0; c < 3; ++c) lo[c] = min(lo[c], v[c]), hi[c] = max(hi[c], v[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[c] = min(lo[c], __shfl_xor(lo[c], o, LK_WAVE));
            hi[c] = max(hi[c], __shfl_xor(hi[c], o, LK_WAVE));
        }
        if ((threadIdx.x & 63) == 0) smin[c][threadIdx.x >> 6] = lo[c], smax[c][threadIdx.x >> 6] = hi[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        atomicMin(&mm[c], min(min(smin[c][0], smin[c][1]), min(smin[c][2], smin[c][3])));
        atomicMax(&mm[3 + c], max(max(smax[c][0], smax[c][1]), max(smax[c][2], smax[c][3])));
    }
}

// cell index idx = ijk0 + ijk1*div0 + ijk2*div0*div1 with ijk = floor(p * inv) - min_b (float arithmetic)
__global__ void __launch_bounds__(256)
    lk_pre_cellidx_kernel(const lk_point* __restrict__ pts, int n, float inv, const int* __restrict__ mm,
                          unsigned int* __restrict__ keys, int* __restrict__ vals, unsigned int* err) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int mn[3], dv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = (int)floorf(lk_ord2f(mm[c]) * inv);
        dv[c] = (int)floorf(lk_ord2f(mm[3 + c]) * inv) - mn[c] + 1;
    }
    if (i == 0 && (double)dv[0] * (double)dv[1] * (double)dv[2] > 2147483647.0) atomicOr(err, 1u);  // PCL refuses this too
    const float4 p = reinterpret_cast<const float4*>(pts)[i];
    const int i0 = (int)floorf(p.x * inv) - mn[0], i1 = (int)floorf(p.y * inv) - mn[1], i2 = (int)floorf(p.z * inv) - mn[2];
    keys[i] = (unsigned int)(i0 + i1 * dv[0] + i2 * dv[0] * dv[1]);
    vals[i] = i;
}

__global__ void __launch_bounds__(256) lk_pre_heads_kernel(const unsigned int* __restrict__ keys, int n, unsigned int* __restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
    lk_pre_starts_kernel(const unsigned int* __restrict__ flags, const unsigned int* __restrict__ pos, int n, int* __restrict__ starts,
                         unsigned int* __restrict__ ncells) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) starts[pos[i]] = i;
    if (i == n - 1) *ncells = pos[i] + flags[i];
}

// one thread per cell: sequential float32 sums in input order (vals are stably sorted by cell), centroid, time key
__global__ void __launch_bounds__(256)
    lk_pre_centroid_kernel(const lk_point* __restrict__ pts, const int* __restrict__ vals, const int* __restrict__ starts,
                           const unsigned int* __restrict__ ncells_p, int n, lk_point* __restrict__ cells,
                           unsigned int* __restrict__ tkeys, int* __restrict__ tvals) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ncells = (int)*ncells_p;
    if (c >= ncells) return;
    const int b = starts[c], e = (c + 1 < ncells) ? starts[c + 1] : n;
    float sx = 0.f, sy = 0.f, sz = 0.f, sc = 0.f;
    for (int k = b; k < e; ++k) {
        const float4 p = reinterpret_cast<const float4*>(pts)[vals[k]];
        sx = sx + p.x, sy = sy + p.y, sz = sz + p.z, sc = sc + p.w;
    }
    const float cnt = (float)(e - b);
    const float4 o = make_float4(sx / cnt, sy / cnt, sz / cnt, sc / cnt);
    reinterpret_cast<float4*>(cells)[c] = o;
    unsigned int u = __float_as_uint(o.w);
    tkeys[c] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving image of the float time stamp
    tvals[c] = c;
}

__global__ void __launch_bounds__(256)
    lk_pre_gather_kernel(const lk_point* __restrict__ cells, const int* __restrict__ order, int n, lk_point* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(cells)[order[i]];
}

// ------------------------------------------------------------------ sensor decode (lidar_processing.cc:25-108)
struct LkDecodeArgs {
    lk_cloud_layout lay;
    double time_scale;
    int filter_num;
    float blind;
};
__device__ __forceinline__ float lk_ld_f32(const unsigned char* p) {  // PointCloud2 fields are not 4-B aligned in general
    unsigned int u = (unsigned int)p[0] | ((unsigned int)p[1] << 8) | ((unsigned int)p[2] << 16) | ((unsigned int)p[3] << 24);
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned int lk_ld_u32(const unsigned char* p) {
    return (unsigned int)p[0] | ((unsigned int)p[1] << 8) | ((unsigned int)p[2] << 16) | ((unsigned int)p[3] << 24);
}
__device__ __forceinline__ double lk_ld_f64(const unsigned char* p) {
    unsigned long long u = (unsigned long long)lk_ld_u32(p) | ((unsigned long long)lk_ld_u32(p + 4) << 32);
    return __longlong_as_double((long long)u);
}
// per point: keep flag (every filter_num-th point outside the blind radius)
__global__ void __launch_bounds__(256)
    lk_decode_flags_kernel(const unsigned char* __restrict__ data, int n, LkDecodeArgs a, unsigned int* __restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned char* p = data + (size_t)i * a.lay.point_step;
    const float x = lk_ld_f32(p + a.lay.off_x), y = lk_ld_f32(p + a.lay.off_y), z = lk_ld_f32(p + a.lay.off_z);
    const bool blind = a.blind * a.blind > x * x + y * y + z * z;  // blindCheck, lidar_processing.h:96-98
    flags[i] = ((i % a.filter_num) || blind) ? 0u : 1u;
}
// scatter the kept points in input order; time arithmetic per handler
__global__ void __launch_bounds__(256)
    lk_decode_scatter_kernel(const unsigned char* __restrict__ data, int n, LkDecodeArgs a, const unsigned int* __restrict__ flags,
                             const unsigned int* __restrict__ pos, lk_point* __restrict__ out, unsigned int* __restrict__ n_out,
                             double* __restrict__ first_last) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned char* p0 = data + a.lay.off_time;
    const unsigned char* pl = data + (size_t)(n - 1) * a.lay.point_step + a.lay.off_time;
    const unsigned char* p = data + (size_t)i * a.lay.point_step;
    float curv;
    double first_d, last_d;
    if (a.lay.lidar_type == 3) {  // hesaiHandler: doubles
        first_d = a.time_scale * lk_ld_f64(p0);
        last_d = a.time_scale * lk_ld_f64(pl);
        const double cur = a.time_scale * lk_ld_f64(p + a.lay.off_time);
        curv = (float)(round((cur - first_d) * (double)500.0f) / (double)500.0f);
    } else {
        float first_f, last_f, cur_f;
        if (a.lay.lidar_type == 2) {  // ousterHander: uint32 t
            first_f = (float)(a.time_scale * (double)lk_ld_u32(p0));
            last_f = (float)(a.time_scale * (double)lk_ld_u32(pl));
            cur_f = (float)(a.time_scale * (double)lk_ld_u32(p + a.lay.off_time));
        } else {  // velodyneHandler: float time
            first_f = (float)(a.time_scale * (double)lk_ld_f32(p0));
            last_f = (float)(a.time_scale * (double)lk_ld_f32(pl));
            cur_f = (float)(a.time_scale * (double)lk_ld_f32(p + a.lay.off_time));
        }
        first_d = (double)first_f, last_d = (double)last_f;
        curv = roundf((cur_f - first_f) * 500.0f) / 500.0f;
    }
    if (i == 0) first_last[0] = first_d, first_last[1] = last_d;
    if (i == n - 1) *n_out = pos[i] + flags[i];
    if (flags[i]) {
        lk_point o;
        o.x = lk_ld_f32(p + a.lay.off_x), o.y = lk_ld_f32(p + a.lay.off_y), o.z = lk_ld_f32(p + a.lay.off_z);
        o.curvature = curv;
        out[pos[i]] = o;
    }
}


// ---------------------------------------------------------------- time buckets of a batch of scans, found on the device
// KILO.cc:375-378: a bucket is a run of EXACTLY equal curvature inside a time-sorted scan.  For a batch laid out back to back
// (scan s = points [scan_off[s], scan_off[s+1])) the bucket tables of lk_batch_replay_ragged_dev are built here instead of on the
// host: flag the run starts, exclusive-scan the flags, scatter the runs' first indices and times (CSR over all scans).
__device__ __forceinline__ int lk_scan_of_point(const unsigned long long* __restrict__ scan_off, int S, unsigned long long i) {
    int lo = 0, hi = S;   // scan_off[lo] <= i < scan_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (scan_off[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}
__global__ void __launch_bounds__(256)
    lk_rag_flag_kernel(const lk_point* __restrict__ pts, unsigned long long n, const unsigned long long* __restrict__ scan_off, int S,
                       unsigned int* __restrict__ flag, unsigned int* __restrict__ stats) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = lk_scan_of_point(scan_off, S, i);
    const float c = pts[i].curvature;
    const bool head = i == scan_off[s];
    const float cp = head ? c : pts[i - 1].curvature;
    flag[i] = (head || c != cp) ? 1u : 0u;
    if (!(c >= cp) || !isfinite(c)) stats[3] = 1u;   // out of time order, NaN or +-inf (an infinite stamp would make the predict's dt infinite):
                                                     // the caller skipped the sort of KILO.cc:367 or handed over a corrupt cloud
}
// stats: [0] total buckets B, [1] largest bucket (points), [2] most buckets in a scan, [3] 1: some scan is not sorted by time
__global__ void __launch_bounds__(256)
    lk_rag_scatter_kernel(const lk_point* __restrict__ pts, unsigned long long n, const unsigned long long* __restrict__ scan_off, int S,
                          const unsigned int* __restrict__ flag, const unsigned int* __restrict__ rank, const double* __restrict__ t_begin,
                          unsigned long long* __restrict__ pt_start, double* __restrict__ tb, unsigned int* __restrict__ bstart,
                          unsigned int* __restrict__ stats) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) {
        const int s = lk_scan_of_point(scan_off, S, i);
        const unsigned int g = rank[i];
        pt_start[g] = i;
        tb[g] = t_begin[s] + (double)pts[i].curvature;   // KILO.cc:376
        if (i == scan_off[s]) bstart[s] = g;
    }
    if (i == n - 1) {
        const unsigned int B = rank[i] + flag[i];
        pt_start[B] = n;
        bstart[S] = B;
        stats[0] = B;
    }
}
__global__ void __launch_bounds__(256)
    lk_rag_stats_kernel(const unsigned long long* __restrict__ pt_start, const unsigned int* __restrict__ bstart, int S,
                        unsigned int* __restrict__ stats) {
    const unsigned int B = stats[0];
    const unsigned int g = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256u >= B && blockIdx.x * 256u >= (unsigned int)S) return;
    unsigned int big = g < B ? (unsigned int)(pt_start[g + 1] - pt_start[g]) : 0u;
    unsigned int most = g < (unsigned int)S ? bstart[g + 1] - bstart[g] : 0u;
    for (int m = 32; m; m >>= 1) {   // one atomic per wave, not per bucket
        big = max(big, (unsigned int)__shfl_xor((int)big, m));
        most = max(most, (unsigned int)__shfl_xor((int)most, m));
    }
    if ((threadIdx.x & 63) == 0) {
        if (big) atomicMax(&stats[1], big);
        if (most) atomicMax(&stats[2], most);
    }
}
