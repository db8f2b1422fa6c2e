// lk_overlay_kernels.h — batch replay WITH the map insert: every scan of the batch owns a copy-on-write OVERLAY of the shared map.
//
// KILO::process inserts every bucket's points into the map before the next bucket is matched (KILO.cc:216-233, voxel_map.cc:336-361),
// so buckets 2..n of a scan see planes their own scan has refitted, cut or created.  The frozen-map batch entries leave that out; this
// path keeps it: SURVEY 8(d) config 5, "scan-local insert overlay".  The shared (base) map stays read-only.  Scan s (= filter slot s)
// owns a complete private LkMap - hash table, node / plane / match / point-block pools, per-bucket work lists - that starts EMPTY and
// only ever holds the root voxels this scan's inserts touch:
//   * the re-projection pass looks a point's root up in the slot's private table first, then in the base map; a point that the base
//     (or private) tree would ignore - it lands in a frozen leaf - is dropped right there, as on the stream path.  Otherwise the root
//     is made private: a NEW private root is claimed (lock-free, as root_find_or_create) and, when the base map has a voxel at that
//     key, remembers it (lk_node_rec::pad_[LK_PAD_COWSRC]); the point is queued on the private root;
//   * lk_ov_cow_kernel, one wave per touched root, copies the base voxel's whole octree - node, plane, match records and the live
//     leaves' points - into the slot's pools (child / block ids renumbered).  Whole subtrees are copied on first touch, so a private
//     tree never points into the base pools;
//   * the root / apply / fallback passes of the stream path (dev_insert_root, dev_insert_apply, dev_insert_fallback) then run
//     UNCHANGED on the private LkMap, with the slot as a second grid dimension: thousands of roots per launch instead of ~1000.
//   * the residual pass of the next bucket finds a key's root through one bit per base grid cell ("this slot has a private root
//     here", 12.8 KB per slot for the bench map: L2-resident): clear -> the frozen-map grid cell as before (match_flat); set -> the
//     slot's private table and the pre-order walk of the private tree (match_root, the stream path's matcher).
// Results per slot are what KILO::process gives for that scan alone on a private copy of the map (oracle: import the base blob,
// process_scan with insert ON): tests/test_gpu_parity.py::test_batch_replay_overlay.
#pragma once
#include "lk_device.h"
#include "lk_map_kernels.h"
#include "lk_point_kernels.h"

#define LK_PAD_QCOUNT 0   // lk_node_rec::pad_[0]: points queued on a root in the current bucket (queue_point_on_root)
#define LK_PAD_COWSRC 3   // lk_node_rec::pad_[3] of a PRIVATE root: 1 + id of the base root still to be copied into it (0: nothing pending)
#define LK_CTR_NODES0 13  // LkMap::counters[13] of a private map: its node count when the current bucket's insert began

// The overlay pools of all slots, passed by value.  Slot s owns element range [s * cap, (s + 1) * cap) of every array.
struct LkOverlay {
    int4* hash;
    lk_plane_rec* planes;
    lk_match_rec* match;
    lk_node_rec* nodes;
    lk_block_rec* blocks;
    unsigned int* counters;      // [S][LK_CTR_COUNT]
    int* touched;                // [S][scan_cap]
    int* next;                   // [S][scan_cap]
    int* scratch;                // [S][scan_cap]
    int* gidx;                   // [S][scan_cap]
    int* groups;                 // [S][2 * scan_cap * 16]  (LkGroup = 16 ints)
    int* slots;                  // [S][nodes_cap][LK_SLOTS]
    int* free_list;              // [S][blocks_cap]
    int* freed_next;             // [S][blocks_cap]
    unsigned int* dirty;         // [S][nodes_cap]
    unsigned int* newroot;       // one shared dummy table (epoch 0: only ever written with 0)
    unsigned int* spec;          // one shared dummy
    unsigned int* bits;          // [S][bit_words]: bit c = the slot has a private root at base grid cell c
    unsigned int hash_cap, nodes_cap, blocks_cap, scan_cap, bit_words;
};

__host__ __device__ inline LkMap ov_slot_map(const LkOverlay& ov, unsigned int slot) {
    LkMap m = {};
    const size_t s = slot;
    m.hash = ov.hash + s * ov.hash_cap;
    m.planes = ov.planes + s * ov.nodes_cap;
    m.match = ov.match + s * ov.nodes_cap;
    m.nodes = ov.nodes + s * ov.nodes_cap;
    m.blocks = ov.blocks + s * ov.blocks_cap;
    m.counters = ov.counters + s * LK_CTR_COUNT;
    m.touched = ov.touched + s * ov.scan_cap;
    m.heavy = nullptr;
    m.next = ov.next + s * ov.scan_cap;
    m.slots = ov.slots + s * ov.nodes_cap * LK_SLOTS;
    m.scratch = ov.scratch + s * ov.scan_cap;
    m.groups = ov.groups + s * ov.scan_cap * 32;
    m.gidx = ov.gidx + s * ov.scan_cap;
    m.free_list = ov.free_list + s * ov.blocks_cap;
    m.freed_next = ov.freed_next + s * ov.blocks_cap;
    m.hash_mask = ov.hash_cap - 1;
    m.max_nodes = ov.nodes_cap;
    m.max_blocks = ov.blocks_cap;
    m.max_scan = ov.scan_cap;
    m.dirty = ov.dirty + s * ov.nodes_cap;
    m.newroot = ov.newroot;
    m.spec = ov.spec;
    m.epoch = 0;
    m.grid_on = 0;
    return m;
}

// ---------------------------------------------------------------- start of a replay: every slot's private map is empty
__global__ void __launch_bounds__(256) lk_ov_reset_kernel(LkOverlay ov) {
    const unsigned int slot = blockIdx.y;
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ov.hash_cap) ov.hash[(size_t)slot * ov.hash_cap + i] = make_int4((int)0x80000000, (int)0x80000000, (int)0x80000000, LK_EMPTY);
    if (i < ov.bit_words) ov.bits[(size_t)slot * ov.bit_words + i] = 0u;
    if (i < LK_CTR_COUNT) ov.counters[(size_t)slot * LK_CTR_COUNT + i] = 0u;
}

// start of a bucket's insert in every slot: dev_bucket_begin_wave + the node count the bucket starts from
__global__ void __launch_bounds__(LK_WAVE) lk_ov_begin_kernel(LkOverlay ov) {
    const LkMap pm = ov_slot_map(ov, blockIdx.x);
    dev_bucket_begin_wave(pm);
    if (threadIdx.x == 0) pm.counters[LK_CTR_NODES0] = min(pm.counters[LK_CTR_NODES], pm.max_nodes);
}

// ---------------------------------------------------------------- re-projection + hashing half of the insert
// Would UpdateOctoTree ignore this point (voxel_map.cc:185-241)?  The walk of dev_reproject_point: down through initialised
// non-planar nodes below max_layer (they never change again) to the node the point would be pushed into; ignored iff that node
// is frozen.
__device__ __forceinline__ bool ov_walk_ignored(const LkMap& m, int root, const V3& pw, int max_layer) {
    int node = root;
    for (int depth = 0; depth <= LK_MAX_LAYER; ++depth) {
        const lk_node_rec* nr = &m.nodes[node];
        const unsigned int st = nr->state;
        const unsigned int pf = m.planes[node].flags;
        if (!(st & LK_NODE_INIT_OCTO)) return false;
        const bool is_plane = (pf & LK_PLANE_IS_PLANE) != 0;
        if (is_plane || nr->layer >= max_layer) return !(st & LK_NODE_UPDATE_ENABLE);
        const int oct = ((pw.x > nr->voxel_center[0]) ? 4 : 0) + ((pw.y > nr->voxel_center[1]) ? 2 : 0) + ((pw.z > nr->voxel_center[2]) ? 1 : 0);
        const int child = nr->child[oct];
        if (child < 0) return false;
        node = child;
    }
    return false;
}

// bit of a key in the slot's private-root bitmap (keys outside the base grid's box have none: they go to the private table directly)
__device__ __forceinline__ bool ov_cell_of(const LkMap& base, const int* key, unsigned int* cell) {
    const unsigned int ux = (unsigned int)(key[0] - base.gmin[0]), uy = (unsigned int)(key[1] - base.gmin[1]), uz = (unsigned int)(key[2] - base.gmin[2]);
    if (ux >= (unsigned int)base.gdim[0] || uy >= (unsigned int)base.gdim[1] || uz >= (unsigned int)base.gdim[2]) return false;
    *cell = (uz * (unsigned int)base.gdim[1] + uy) * (unsigned int)base.gdim[0] + ux;
    return true;
}

// root_find_or_create (lk_point_kernels.h) on the slot's private table, without its plain-load fast path (the caller has tried
// it).  A root created here is a complete empty root voxel (voxel_map.cc:345-357) that remembers the base voxel it stands for.
__device__ __forceinline__ int ov_root_find_or_create(const LkMap& pm, const LkMap& base, unsigned int* __restrict__ bits, const LkParams& pr,
                                                      const int* key, int base_root) {
    unsigned int s = lk_hash3(key[0], key[1], key[2]) & pm.hash_mask;
    int* slotw = reinterpret_cast<int*>(pm.hash);
    for (unsigned int trips = 0; trips < 64u * (pm.hash_mask + 1u); ++trips) {
        const int w = __hip_atomic_load(&slotw[4 * s + 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (w == LK_EMPTY) {
            int expected = LK_EMPTY;
            if (__hip_atomic_compare_exchange_strong(&slotw[4 * s + 3], &expected, LK_LOCKED, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
                const unsigned int id = atomicAdd(&pm.counters[LK_CTR_NODES], 1u);
                if (id >= pm.max_nodes) {
                    atomicOr(&pm.counters[LK_CTR_ERR], LK_E_NODES_FULL);
                    __hip_atomic_store(&slotw[4 * s + 3], LK_EMPTY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    return -1;
                }
                atomicAdd(&pm.counters[LK_CTR_ROOTS], 1u);
                lk_node_rec* nd = &pm.nodes[id];
                const double vs = (double)pr.voxel_size_f;
#pragma unroll
                for (int c = 0; c < 8; ++c) nd->child[c] = -1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    nd->voxel_center[c] = (0.5 + key[c]) * vs;  // voxel_map.cc:355-357
                    nd->key[c] = key[c];
                }
                nd->quater_length = pr.voxel_size_f / 4;          // voxel_map.cc:354
                nd->layer = 0;
                nd->npts = 0;
                nd->new_points = 0;
                nd->state = LK_NODE_UPDATE_ENABLE;
                nd->block = -1;
                nd->list_head = -1;
                nd->pad_[LK_PAD_QCOUNT] = 0;
                nd->pad_[LK_PAD_COWSRC] = (unsigned int)(base_root + 1);
                pm.planes[id].flags = 0;
                pm.match[id].flags = 0;
                unsigned int cell;
                if (ov_cell_of(base, key, &cell)) atomicOr(&bits[cell >> 5], 1u << (cell & 31u));
                slotw[4 * s + 0] = key[0];
                slotw[4 * s + 1] = key[1];
                slotw[4 * s + 2] = key[2];
                __hip_atomic_store(&slotw[4 * s + 3], (int)id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return (int)id;
            }
            continue;  // lost the race: re-read this slot
        }
        if (w == LK_LOCKED) {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const int kx = __hip_atomic_load(&slotw[4 * s + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ky = __hip_atomic_load(&slotw[4 * s + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int kz = __hip_atomic_load(&slotw[4 * s + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kx == key[0] && ky == key[1] && kz == key[2]) return w;
        s = (s + 1) & pm.hash_mask;
    }
    atomicOr(&pm.counters[LK_CTR_ERR], LK_E_HASH_FULL);
    return -1;
}

// KILO.cc:216-230 + the hashing half of UpdateVoxelMap (voxel_map.cc:343-358) for bucket point i of slot blockIdx.y, on the
// slot's overlay.  A private root created in an EARLIER bucket (id below the node count this bucket began with) is complete and
// is walked itself; one created in THIS launch is still an empty placeholder - the base voxel it stands for is walked instead
// (the insert's first phase is read-only on every tree, so the base voxel is still the truth).
__global__ void __launch_bounds__(LK_WAVE)
    lk_ov_reproject_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                           size_t pts_slot_stride, int n) {
    const int i = blockIdx.x * LK_WAVE + threadIdx.x;
    if (i >= n) return;
    const unsigned int slot = blockIdx.y;
    const LkMap pm = ov_slot_map(ov, slot);
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    const float4 p = reinterpret_cast<const float4*>(pts + (size_t)slot * pts_slot_stride)[i];
    const V3 pw = point_world(p.x, p.y, p.z, bc, pr);
    int key[3];
    key_floor(pw, pr.voxel_size_f, key);
    const int nodes0 = (int)pm.counters[LK_CTR_NODES0];
    const int proot = hash_find(pm, key[0], key[1], key[2]);
    int broot = -1;
    if (proot >= 0 && proot < nodes0) {
        if (ov_walk_ignored(pm, proot, pw, pr.max_layer)) return;
    } else {
        broot = hash_find(base, key[0], key[1], key[2]);
        if (broot >= 0 && ov_walk_ignored(base, broot, pw, pr.max_layer)) return;
    }
    const int root = proot >= 0 ? proot : ov_root_find_or_create(pm, base, ov.bits + (size_t)slot * ov.bit_words, pr, key, broot);
    if (root < 0) return;
    queue_point_on_root(pm, root, i);
}

// ---------------------------------------------------------------- copy-on-write of a base voxel's octree
// One wave copies node `src` of the base map (and, recursively, its children) into node `dst` of the private map: plane and match
// records as 16-B chunks, a live leaf's points into a private block, child and block ids renumbered.  The root keeps the queue
// fields the re-projection pass has filled in (pad_[LK_PAD_QCOUNT], list_head).
template <int L>
__device__ __forceinline__ void ov_copy_node(const LkMap& pm, const LkMap& base, const int src, const int dst) {
    const int lane = threadIdx.x & 63;
    const lk_node_rec* sn = &base.nodes[src];
    lk_node_rec* dn = &pm.nodes[dst];
    const int s_block = bcast0(sn->block), s_npts = bcast0(sn->npts);
    if (lane < 16) reinterpret_cast<uint4*>(&pm.planes[dst])[lane] = reinterpret_cast<const uint4*>(&base.planes[src])[lane];
    else if (lane < 25) reinterpret_cast<uint4*>(&pm.match[dst])[lane - 16] = reinterpret_cast<const uint4*>(&base.match[src])[lane - 16];
    int nblock = -1;
    if (s_block >= 0) {
        nblock = alloc_block(pm);
        const int nd8 = min(max(s_npts, 0), LK_BLOCK_PTS) * 9;   // lk_pt_rec = 9 doubles; only the first npts records hold points
        const double* sp = reinterpret_cast<const double*>(&base.blocks[s_block]);
        double* dp = reinterpret_cast<double*>(&pm.blocks[nblock]);
        for (int k = lane; k < nd8; k += LK_WAVE) dp[k] = sp[k];
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dn->voxel_center[c] = sn->voxel_center[c], dn->key[c] = sn->key[c];
        dn->quater_length = sn->quater_length;
        dn->layer = sn->layer;
        dn->npts = s_npts;
        dn->new_points = sn->new_points;
        dn->state = sn->state;
        dn->block = nblock;
        if (L > 0) dn->list_head = -1, dn->pad_[LK_PAD_QCOUNT] = 0;
        dn->pad_[LK_PAD_COWSRC] = 0;
    }
    // children one at a time (nothing is kept across the recursion but the loop index)
    for (int c = 0; c < 8; ++c) {
        const int ch = bcast0(sn->child[c]);
        int id = -1;
        if (ch >= 0) {
            if (lane == 0) {
                unsigned int nn = atomicAdd(&pm.counters[LK_CTR_NODES], 1u);
                if (nn >= pm.max_nodes) {
                    atomicOr(&pm.counters[LK_CTR_ERR], LK_E_NODES_FULL);
                    nn = pm.max_nodes - 1;   // memory-safe; the error flag fails the call
                }
                id = (int)nn;
            }
            id = bcast0(id);
            if constexpr (L < LK_MAX_LAYER) ov_copy_node<L + 1>(pm, base, ch, id);
        }
        if (lane == 0) dn->child[c] = id;
    }
}
__global__ void __launch_bounds__(LK_MB) lk_ov_cow_kernel(LkMap base, LkOverlay ov) {
    const LkMap pm = ov_slot_map(ov, blockIdx.y);
    const int wave = (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * LK_MB) >> 6);
    const int n_touched = (int)pm.counters[LK_CTR_TOUCHED];
    for (int t = wave; t < n_touched; t += nwaves) {
        const int root = bcast0(pm.touched[t]);
        const int src1 = bcast0((int)pm.nodes[root].pad_[LK_PAD_COWSRC]);
        if (src1 > 0) ov_copy_node<0>(pm, base, src1 - 1, root);
    }
}

// ---------------------------------------------------------------- the ordered insert, slot = blockIdx.y
__global__ void __launch_bounds__(LK_MB)
    lk_ov_insert_root_kernel(LkOverlay ov, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts, size_t pts_slot_stride, int n) {
    const LkMap pm = ov_slot_map(ov, blockIdx.y);
    dev_insert_root<false>(pm, pr, filters + blockIdx.y, pts + (size_t)blockIdx.y * pts_slot_stride, (const lk_pt_rec*)nullptr, n,
                           (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
}
__global__ void __launch_bounds__(LK_MB)
    lk_ov_insert_apply_kernel(LkOverlay ov, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts, size_t pts_slot_stride, int n) {
    const LkMap pm = ov_slot_map(ov, blockIdx.y);
    dev_insert_apply<false>(pm, pr, filters + blockIdx.y, pts + (size_t)blockIdx.y * pts_slot_stride, (const lk_pt_rec*)nullptr, n,
                            (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
}
__global__ void __launch_bounds__(LK_MB)
    lk_ov_insert_fallback_kernel(LkOverlay ov, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts, size_t pts_slot_stride, int n) {
    const LkMap pm = ov_slot_map(ov, blockIdx.y);
    dev_insert_fallback<false>(pm, pr, filters + blockIdx.y, pts + (size_t)blockIdx.y * pts_slot_stride, (const lk_pt_rec*)nullptr, n,
                               (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
}

// ---------------------------------------------------------------- residual pass against base grid + overlay
// lk_residual_kernel's body with the overlay root lookup (residual_tile<..., GRID = 3>).
template <bool XID>
__global__ void LK_RES_BOUNDS
    lk_ov_residual_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                          size_t pts_slot_stride, int n, double* __restrict__ partials, size_t part_slot_stride) {
    __shared__ double stage[64 * LK_ROW2];
    const unsigned int slot = blockIdx.y;
    const int lane = threadIdx.x;
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    LkOvView ovv;
    ovv.hash = ov.hash + (size_t)slot * ov.hash_cap;
    ovv.hash_mask = ov.hash_cap - 1;
    ovv.match = ov.match + (size_t)slot * ov.nodes_cap;
    ovv.nodes = ov.nodes + (size_t)slot * ov.nodes_cap;
    ovv.bits = ov.bits + (size_t)slot * ov.bit_words;
    ResidualOut out;
    out.h6 = nullptr, out.z = nullptr, out.R = nullptr, out.valid = nullptr, out.world = nullptr, out.ids = nullptr;
    const double acc = residual_tile<false, 3, XID, false, false>(base, pr, bc, reinterpret_cast<const float4*>(pts + (size_t)slot * pts_slot_stride),
                                                                  blockIdx.x * LK_RB + lane, n, stage, lane, out, (size_t)0, &ovv);
    if (lane < LK_NPART) partials[(size_t)slot * part_slot_stride + (size_t)blockIdx.x * LK_NPART + lane] = (lane < 29) ? acc : 0.0;
}

// ---------------------------------------------------------------- status of all slots after a replay
// out[0] = OR of the slots' error words, out[1..3] = largest node / block / root count of any slot, out[4] = first slot with an error
__global__ void __launch_bounds__(256) lk_ov_status_kernel(LkOverlay ov, unsigned int n_slots, unsigned int* __restrict__ out) {
    for (unsigned int s = blockIdx.x * 256 + threadIdx.x; s < n_slots; s += gridDim.x * 256) {
        const unsigned int* c = ov.counters + (size_t)s * LK_CTR_COUNT;
        if (c[LK_CTR_ERR]) {
            atomicOr(&out[0], c[LK_CTR_ERR]);
            atomicMin(&out[4], s);
        }
        atomicMax(&out[1], c[LK_CTR_NODES]);
        atomicMax(&out[2], c[LK_CTR_BLOCKS]);
        atomicMax(&out[3], c[LK_CTR_ROOTS]);
    }
}
