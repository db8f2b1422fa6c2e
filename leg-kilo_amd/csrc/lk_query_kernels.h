// Per-point queries of the voxel map - the class-surface form of VoxelMapManager::build_single_residual (voxel_map.cc:363-427):
// a caller-supplied WORLD point with a caller-supplied 3x3 covariance (pointWithVar::point_w / ::var), started on the root voxel of
// a caller-supplied key with is_success = false, prob = 0 the way KILO.cc:149-155 starts it.  The bucket kernels
// (lk_point_kernels.h) fold the point's covariance out of the state instead and never see a pointWithVar; this is the entry for a
// caller who holds one.  One thread per point, the full 6x6 plane_var of lk_plane_rec (not the folded match record), gate and
// probability in the reference's own form (sqrt, not squared; every passing candidate's probability evaluated).
#pragma once
#include "lk_device.h"

struct LkMatchOut {
    unsigned char* found;     // a root voxel exists at the key (KILO.cc:149)
    unsigned char* success;   // is_success
    double* prob;
    double* normal;           // n x 3   single_ptpl.normal_
    double* center;           // n x 3   single_ptpl.center_
    double* d;                //         single_ptpl.d_
    float* dis_to_plane;      //         single_ptpl.dis_to_plane_ (signed, stored as float: voxel_map.h:92)
    int* layer;               //         single_ptpl.layer_, -1 when no plane was taken
};

// one plane node: voxel_map.cc:371-413.  upper(r, c) of the row-major upper triangle: row r starts at r*6 - r*(r-1)/2
__device__ inline void lk_query_plane(const lk_plane_rec* pl, const double* p, const double* var, double sigma_num, int layer, bool& success,
                                      double& prob, int& best_node, int& best_layer, int node) {
    const double nx = pl->normal[0], ny = pl->normal[1], nz = pl->normal[2];
    const double cx = pl->center[0], cy = pl->center[1], cz = pl->center[2];
    const double sd = nx * p[0] + ny * p[1] + nz * p[2] + (double)pl->d;
    const float dis_to_plane = (float)fabs(sd);
    const float dis_to_center = (float)((cx - p[0]) * (cx - p[0]) + (cy - p[1]) * (cy - p[1]) + (cz - p[2]) * (cz - p[2]));
    const float range_dis = sqrtf(dis_to_center - dis_to_plane * dis_to_plane);
    if (!((double)range_dis <= 3.0 * (double)pl->radius)) return;   // radius_k = 3
    const double J[6] = {p[0] - cx, p[1] - cy, p[2] - cz, -nx, -ny, -nz};
    double sigma_l = 0.0;
    for (int c = 0; c < 6; ++c) {   // (J plane_var) J^T, the row vector first as Eigen evaluates the product chain
        double t = 0.0;
        for (int r = 0; r < 6; ++r) {
            const int a = r < c ? r : c, b = r < c ? c : r;
            t += J[r] * pl->plane_var[a * 6 - a * (a - 1) / 2 + (b - a)];
        }
        sigma_l += t * J[c];
    }
    double nvn = 0.0;
    const double n3[3] = {nx, ny, nz};
    for (int c = 0; c < 3; ++c) {
        double t = 0.0;
        for (int r = 0; r < 3; ++r) t += n3[r] * var[3 * r + c];
        nvn += t * n3[c];
    }
    sigma_l += nvn;
    if (!((double)dis_to_plane < sigma_num * sqrt(sigma_l))) return;
    success = true;
    const double this_prob = 1.0 / (sqrt(sigma_l)) * exp(-0.5 * (double)dis_to_plane * (double)dis_to_plane / sigma_l);
    if (this_prob > prob) {
        prob = this_prob;
        best_node = node;
        best_layer = layer;
    }
}

__global__ void lk_match_points_kernel(LkMap m, LkParams pr, const int* __restrict__ keys, const double* __restrict__ pw,
                                       const double* __restrict__ var9, int n, LkMatchOut out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int root = hash_find(m, keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
    bool success = false;
    double prob = 0.0;
    int best_node = -1, best_layer = -1;
    out.found[i] = root >= 0;
    if (root >= 0) {
        double p[3] = {pw[3 * i], pw[3 * i + 1], pw[3 * i + 2]};
        double var[9];
        for (int k = 0; k < 9; ++k) var[k] = var9[9 * (size_t)i + k];
        // pre-order walk, children in index order (voxel_map.cc:415-421); at most LK_MAX_LAYER + 1 levels
        int node_at[LK_MAX_LAYER + 1], next_child[LK_MAX_LAYER + 1];
        int level = 0;
        node_at[0] = root, next_child[0] = -1;   // -1: the node itself has not been looked at yet
        while (level >= 0) {
            const int node = node_at[level];
            if (next_child[level] < 0) {
                const lk_plane_rec* pl = &m.planes[node];
                if (pl->flags & LK_PLANE_IS_PLANE) {
                    lk_query_plane(pl, p, var, pr.sigma_num, level, success, prob, best_node, best_layer, node);
                    --level;
                    continue;
                }
                if (level >= pr.max_layer || level >= LK_MAX_LAYER) {
                    --level;
                    continue;
                }
                next_child[level] = 0;
            }
            int child = -1;
            while (next_child[level] < 8 && child < 0) child = m.nodes[node].child[next_child[level]++];
            if (child >= 0) {
                ++level;
                node_at[level] = child, next_child[level] = -1;
            } else {
                --level;
            }
        }
    }
    out.success[i] = success;
    out.prob[i] = prob;
    out.layer[i] = best_layer;
    if (best_node >= 0) {
        const lk_plane_rec* pl = &m.planes[best_node];
        for (int k = 0; k < 3; ++k) out.normal[3 * i + k] = pl->normal[k], out.center[3 * i + k] = pl->center[k];
        out.d[i] = (double)pl->d;
        out.dis_to_plane[i] = (float)(pl->normal[0] * pw[3 * i] + pl->normal[1] * pw[3 * i + 1] + pl->normal[2] * pw[3 * i + 2] + (double)pl->d);
    } else {
        for (int k = 0; k < 3; ++k) out.normal[3 * i + k] = 0.0, out.center[3 * i + k] = 0.0;
        out.d[i] = 0.0;
        out.dis_to_plane[i] = 0.0f;
    }
}
