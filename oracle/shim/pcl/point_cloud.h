// oracle/shim — TEST INFRASTRUCTURE ONLY: a container with the members of pcl::PointCloud the reference touches.
#ifndef LK_SHIM_PCL_POINT_CLOUD
#define LK_SHIM_PCL_POINT_CLOUD
#include <cstddef>
#include <memory>
#include <vector>
namespace pcl {
template <class PointT>
class PointCloud {
   public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    unsigned int width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); }
    void resize(size_t n) { points.resize(n); }
    void reserve(size_t n) { points.reserve(n); }
    void push_back(const PointT& p) { points.push_back(p); }
    PointT& operator[](size_t i) { return points[i]; }
    const PointT& operator[](size_t i) const { return points[i]; }
    typename std::vector<PointT>::iterator begin() { return points.begin(); }
    typename std::vector<PointT>::iterator end() { return points.end(); }
    typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
    typename std::vector<PointT>::const_iterator end() const { return points.end(); }
};
}  // namespace pcl
#endif
