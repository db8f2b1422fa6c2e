// oracle/export_blob.hpp — TEST INFRASTRUCTURE ONLY.
// Serialises a voxel map (std::unordered_map<key, VoxelOctoTree*>) into the blob format of include/legkilo_hip.h in
// canonical order: roots sorted by key, nodes in DFS pre-order, children by octant index.  A template over the tree /
// plane / point types so that the SAME code walks the oracle's restated classes (lko::, oracle_voxel_map.hpp) and the
// reference's own classes (legkilo::, voxel_map.h:96-176, compiled into oracle/_ref): both expose the member names of
// the reference.
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <utility>
#include <vector>

#include "../include/legkilo_hip.h"

namespace lkx {

template <class Tree, class Plane, class PV>
struct Exporter {
    std::vector<lk_root_rec> roots;
    std::vector<lk_node_rec> nodes;
    std::vector<lk_plane_rec> planes;
    std::vector<lk_block_rec> blocks;
    int max_layer = 0;

    int add(const Tree* t, const int* key) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        planes.emplace_back();
        lk_node_rec n;
        std::memset(&n, 0, sizeof(n));
        lk_plane_rec p;
        std::memset(&p, 0, sizeof(p));
        for (int c = 0; c < 3; ++c) n.voxel_center[c] = t->voxel_center_[c];
        n.quater_length = t->quater_length_;
        n.layer = t->layer_;
        n.npts = (int)t->temp_points_.size();
        n.new_points = t->new_points_;
        n.state = (t->init_octo_ ? LK_NODE_INIT_OCTO : 0u) | (t->update_enable_ ? LK_NODE_UPDATE_ENABLE : 0u) |
                  (t->octo_state_ ? LK_NODE_OCTO_STATE : 0u);
        n.block = -1;
        n.list_head = -1;
        if (key)
            for (int c = 0; c < 3; ++c) n.key[c] = key[c];
        const Plane& pl = *t->plane_ptr_;
        bool dead = t->init_octo_ && !pl.is_plane_ && t->layer_ < max_layer;  // points never read again
        if (!dead && n.npts > 0) {
            if (n.npts <= LK_BLOCK_PTS) {
                n.block = (int)blocks.size();
                blocks.emplace_back();
                lk_block_rec& b = blocks.back();
                std::memset(&b, 0, sizeof(b));
                for (int i = 0; i < n.npts; ++i) {
                    const PV& pv = t->temp_points_[i];
                    for (int c = 0; c < 3; ++c) b.pts[i].pw[c] = pv.point_w[c];
                    b.pts[i].var[0] = pv.var(0, 0), b.pts[i].var[1] = pv.var(0, 1), b.pts[i].var[2] = pv.var(0, 2);
                    b.pts[i].var[3] = pv.var(1, 1), b.pts[i].var[4] = pv.var(1, 2), b.pts[i].var[5] = pv.var(2, 2);
                }
            } else {
                n.state |= LK_NODE_PTS_DROPPED;
            }
        }
        for (int c = 0; c < 3; ++c) p.center[c] = pl.center_[c], p.normal[c] = pl.normal_[c];
        p.d = pl.d_;
        p.radius = pl.radius_;
        p.flags = (pl.is_plane_ ? LK_PLANE_IS_PLANE : 0u) | (pl.is_init_ ? LK_PLANE_IS_INIT : 0u);
        p.points_size = pl.points_size_;
        int k = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) p.plane_var[k++] = pl.plane_var_(r, c);
        p.min_eigen_value = pl.min_eigen_value_;
        p.mid_eigen_value = pl.mid_eigen_value_;
        p.max_eigen_value = pl.max_eigen_value_;
        for (int l = 0; l < 8; ++l) n.child[l] = -1;
        nodes[id] = n;
        planes[id] = p;
        for (int l = 0; l < 8; ++l)
            if (t->leaves_[l]) {
                int cid = add(t->leaves_[l], nullptr);
                nodes[id].child[l] = cid;
            }
        return id;
    }
};

template <class Tree, class Plane, class PV, class Map>
int export_map(const Map& voxel_map, double voxel_size, int max_layer, int max_points_num, void* blob, size_t* bytes) {
    std::vector<std::pair<std::array<int, 3>, const Tree*>> sorted;
    sorted.reserve(voxel_map.size());
    for (const auto& kv : voxel_map) sorted.push_back({{kv.first[0], kv.first[1], kv.first[2]}, kv.second});
    std::sort(sorted.begin(), sorted.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    Exporter<Tree, Plane, PV> ex;
    ex.max_layer = max_layer;
    for (auto& kv : sorted) {
        int id = ex.add(kv.second, kv.first.data());
        ex.roots.push_back(lk_root_rec{{kv.first[0], kv.first[1], kv.first[2]}, id});
    }
    lk_blob_header hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = LK_BLOB_MAGIC;
    hd.version = LK_ABI_VERSION;
    hd.n_roots = (uint32_t)ex.roots.size();
    hd.n_nodes = (uint32_t)ex.nodes.size();
    hd.n_blocks = (uint32_t)ex.blocks.size();
    hd.block_pts = LK_BLOCK_PTS;
    hd.voxel_size = voxel_size;
    hd.max_layer = max_layer;
    hd.max_points_num = max_points_num;
    size_t total = sizeof(hd) + ex.roots.size() * sizeof(lk_root_rec) + ex.nodes.size() * sizeof(lk_node_rec) +
                   ex.planes.size() * sizeof(lk_plane_rec) + ex.blocks.size() * sizeof(lk_block_rec);
    hd.bytes = total;
    if (!blob) {
        *bytes = total;
        return 0;
    }
    if (*bytes < total) return -1;
    char* p = (char*)blob;
    std::memcpy(p, &hd, sizeof(hd));
    p += sizeof(hd);
    std::memcpy(p, ex.roots.data(), ex.roots.size() * sizeof(lk_root_rec));
    p += ex.roots.size() * sizeof(lk_root_rec);
    std::memcpy(p, ex.nodes.data(), ex.nodes.size() * sizeof(lk_node_rec));
    p += ex.nodes.size() * sizeof(lk_node_rec);
    std::memcpy(p, ex.planes.data(), ex.planes.size() * sizeof(lk_plane_rec));
    p += ex.planes.size() * sizeof(lk_plane_rec);
    std::memcpy(p, ex.blocks.data(), ex.blocks.size() * sizeof(lk_block_rec));
    *bytes = total;
    return 0;
}

}  // namespace lkx
