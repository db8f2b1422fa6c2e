// oracle/shim — TEST INFRASTRUCTURE ONLY: restatement of pcl::VoxelGrid<PointT>::filter (PCL 1.8, third-party, absent
// here) as the reference uses it (KILO.cc:80-81, :356-360): cell index from floor(xyz / leaf) relative to the cloud's
// minimum, one output point per occupied cell = the centroid of ALL fields of its points (float accumulation, as PCL's
// CentroidPoint does), output in ascending cell index.  PCL sorts (cell, point) pairs with the unstable std::sort, so the
// summation order inside a cell is unspecified there; here it is the input order (the choice oracle/preprocess_oracle.py
// and the device kernels make too).  The pinning tests feed clouds that already have one point per cell, on which the
// filter is the identity.
#ifndef LK_SHIM_PCL_VOXEL_GRID
#define LK_SHIM_PCL_VOXEL_GRID
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>
#include "../point_cloud.h"
#include "../point_types.h"
namespace pcl {
template <class PointT>
class VoxelGrid {
    float lx_ = 0, ly_ = 0, lz_ = 0;
    typename PointCloud<PointT>::ConstPtr in_;

   public:
    void setLeafSize(float lx, float ly, float lz) { lx_ = lx, ly_ = ly, lz_ = lz; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { in_ = c; }
    void filter(PointCloud<PointT>& out) {
        out.clear();
        if (!in_ || in_->empty()) return;
        float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
        float mx[3] = {-mn[0], -mn[1], -mn[2]};
        for (const PointT& p : in_->points) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            mn[0] = std::min(mn[0], p.x), mn[1] = std::min(mn[1], p.y), mn[2] = std::min(mn[2], p.z);
            mx[0] = std::max(mx[0], p.x), mx[1] = std::max(mx[1], p.y), mx[2] = std::max(mx[2], p.z);
        }
        const float inv[3] = {1.0f / lx_, 1.0f / ly_, 1.0f / lz_};
        int minb[3], maxb[3], divb[3];
        for (int a = 0; a < 3; ++a) {
            minb[a] = (int)std::floor(mn[a] * inv[a]), maxb[a] = (int)std::floor(mx[a] * inv[a]);
            divb[a] = maxb[a] - minb[a] + 1;
        }
        const int mul[3] = {1, divb[0], divb[0] * divb[1]};
        std::vector<std::pair<unsigned int, unsigned int>> idx;
        idx.reserve(in_->size());
        for (unsigned int i = 0; i < in_->size(); ++i) {
            const PointT& p = in_->points[i];
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            const int i0 = (int)std::floor(p.x * inv[0]) - minb[0], i1 = (int)std::floor(p.y * inv[1]) - minb[1],
                      i2 = (int)std::floor(p.z * inv[2]) - minb[2];
            idx.push_back({(unsigned int)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), i});
        }
        std::stable_sort(idx.begin(), idx.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (size_t s = 0; s < idx.size();) {
            size_t e = s;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            while (e < idx.size() && idx[e].first == idx[s].first) {
                const PointT& p = in_->points[idx[e].second];
                acc[0] += p.x, acc[1] += p.y, acc[2] += p.z, acc[3] += p.normal_x, acc[4] += p.normal_y, acc[5] += p.normal_z;
                acc[6] += p.intensity, acc[7] += p.curvature;
                ++e;
            }
            const float n = (float)(e - s);
            PointT q;
            q.x = acc[0] / n, q.y = acc[1] / n, q.z = acc[2] / n, q.normal_x = acc[3] / n, q.normal_y = acc[4] / n, q.normal_z = acc[5] / n;
            q.intensity = acc[6] / n, q.curvature = acc[7] / n;
            out.push_back(q);
            s = e;
        }
    }
};
}  // namespace pcl
#endif
