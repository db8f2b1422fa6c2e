// lk_overlay_kernels.h — batch replay WITH the map insert: every scan of the batch owns a copy-on-write OVERLAY of the shared map.
//
// KILO::process inserts every bucket's points into the map before the next bucket is matched (KILO.cc:216-233, voxel_map.cc:336-361),
// so buckets 2..n of a scan see planes their own scan has refitted, cut or created.  The frozen-map batch entries leave that out; this
// path keeps it: SURVEY 8(d) config 5, "scan-local insert overlay".  The shared (base) map stays read-only.  Scan s (= filter slot s)
// owns a complete private LkMap - hash table, node / plane / match / point-block pools, per-bucket work lists - that starts EMPTY and
// only ever holds the root voxels this scan's inserts touch:
//   * the re-projection pass drops a point whose base voxel is a frozen leaf after one bit test (lk_ov_frozen_bits_kernel: such a
//     voxel ignores points for good, so no scan ever owns a copy of it); otherwise it looks the point's root up in the slot's private
//     table first, then in the base map, drops the point if the tree it finds would ignore it, else claims the key in the slot's
//     table (one relaxed 64-bit CAS; the table index is the root's node id) and queues the point on that root record;
//   * lk_ov_materialise_kernel makes the touched roots that do not exist yet private: 64 roots per wave, one lane each - an empty
//     root where the base map has none; for a leaf voxel the node record by its lane, plane + match records of the chunk as one
//     flat list of 16-B pieces, the POINTS left to the root pass ("thin" root); only a cut voxel's octree is copied node by node
//     (child / block ids renumbered).  A private tree never points into the base pools once its bucket is over;
//   * the root / apply / fallback passes of the stream path (dev_insert_root, dev_insert_apply, dev_insert_fallback) then run on
//     the private LkMap with the slot as a second grid dimension - 10^6 roots per launch instead of ~1000 - in their batch form
//     (template flag OV): a thin root's old points are read from the base block and written, with the new ones, to the private
//     block; the next root's record is requested a root ahead; the plane fit that ends a leaf's bucket is only DECIDED and left as
//     a 96-B job to lk_ov_fit_lane_kernel, which fits one plane per LANE;
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
// LK_PAD_LIVE (pad_[3]) / LK_PAD_COWBLK (pad_[5]) of a PRIVATE root record: lk_map_kernels.h (dev_insert_root<.., OV> reads them).  LIVE: 0 the key may
// have been claimed in this bucket but the root does not exist in the slot's map yet; 1 it exists; 2 "thin": it exists, only its old points are still
// the base map's (block COWBLK - 1) - a state that lasts from the copy-on-write pass to the root pass of the same bucket.
#define LK_PAD_BASE 4     // lk_node_rec::pad_[4] of a private root record that does not exist yet: 1 + id of the base map's voxel of that key (0: none)
#define LK_PLANE_LAZY (-2)   // lk_plane_rec::points_size of a private root whose plane record (but for d, radius, flags) has not been made private: the base map's holds
#define LK_OV_EMPTY 0x8000000000000000ull   // empty entry of a slot's key table (a packed key never has bit 63 set)

// Moment sums of the FIRST n points of a leaf's block (sum p, sum p p^T: what every refit event's plane test is made from,
// voxel_map.cc:46-53).  A leaf's points are append-only until it freezes or is cut, so a prefix stays valid whoever appends; the fast
// root pass keeps it up to date at every refit event (whose sums it has computed anyway) and only reads the points behind the prefix.
struct LkLeafSum {   // 80 B = 5 x 16 B
    double s9[9];
    int n, pad_;
};
static_assert(sizeof(LkLeafSum) == 80, "leaf sums record must be 80 B");

// Where the points of the CURRENT bucket of slot s are.  Uniform batch (lk_batch_replay_overlay_dev: every scan the same shape): pts + s * stride,
// n points.  Ragged batch (lk_batch_replay_overlay_ragged_dev: every scan its own size and bucket table, LkRagged's padded form): bucket b of
// slot s = pts[pt_off[s][b] .. pt_off[s][b + 1]), none when the scan has fewer than b + 1 buckets.
struct LkPtSrc {
    const lk_point* pts;
    size_t stride;
    int n;
    const unsigned long long* pt_off;   // null: uniform
    const unsigned int* nb;
    int ldb, b;
    const int* b_slot;                  // optional [S]: every slot its OWN bucket index (the scan-resident replay: a launch behind it finds the slots at different buckets); null: b
};
__device__ __forceinline__ int ov_pt_src(const LkPtSrc& s, unsigned int slot, const lk_point** p) {
    if (!s.pt_off) {
        *p = s.pts + (size_t)slot * s.stride;
        return s.n;
    }
    *p = s.pts;
    const int b = s.b_slot ? s.b_slot[slot] : s.b;
    if (b < 0 || b >= (int)s.nb[slot]) return 0;
    const unsigned long long* po = s.pt_off + (size_t)slot * (size_t)(s.ldb + 1);
    *p = s.pts + po[b];
    return (int)(po[b + 1] - po[b]);
}

// The overlay pools of all slots, passed by value.  Slot s owns element range [s * cap, (s + 1) * cap) of every array.
// Private root table of a slot: open addressing over PACKED 64-bit keys (3 x 21 bits), and the root's node id IS its table index -
// node records [0, hash_cap) of the slot are its roots, children are allocated from hash_cap upwards.  A key is claimed with ONE
// relaxed 64-bit compare-and-swap and nothing has to be published to the other lanes: no lock word, no fences (an agent-scope
// acquire / release is an L2 invalidate / write-back on a part whose eight XCDs each have their own L2 - the first version of this
// pass spun on one per point and took 14 ms per launch where this one takes a fraction of a millisecond).
struct LkOverlay {
    unsigned long long* keys;    // [S][hash_cap]
    lk_plane_rec* planes;
    lk_match_rec* match;
    lk_node_rec* nodes;          // [S][nodes_cap], nodes_cap = hash_cap + children
    lk_block_rec* blocks;
    unsigned int* counters;      // [S][LK_CTR_COUNT]
    int* touched;                // [S][scan_cap]
    int* next;                   // [S][scan_cap]
    int* scratch;                // [S][scan_cap]
    int* gidx;                   // [S][scan_cap]
    int* groups;                 // [S][2 * scan_cap * 16]  (LkGroup = 16 ints)
    int* slots;                  // [S][hash_cap][LK_SLOTS] x float4 {x, y, z, index}: the points queued on a root in the current bucket (only roots queue points)
    int* free_list;              // [S][blocks_cap]
    int* freed_next;             // [S][blocks_cap]
    unsigned int* dirty;         // [S][hash_cap] (roots only; epoch 0)
    unsigned int* newroot;       // one shared dummy table (epoch 0: only ever written with 0)
    unsigned int* spec;          // one shared dummy
    unsigned int* bits;          // [S][bit_words]: bit c = the slot has a private root at base grid cell c
    unsigned int* frozen;        // [2 * bit_words], shared by all slots: TWO bits per grid cell c of the BASE map (lk_ov_frozen_bits_kernel): bit 0 = its voxel is a
                                 // frozen leaf (UpdateOctoTree ignores the point), bit 1 = its voxel takes the point AT THE ROOT (a live leaf, or not initialised
                                 // yet): not ignored, and nothing below the root to look at - the re-projection needs no read of the base tree for it
    struct LkFitJob* jobs;       // [S][hash_cap][LK_INLINE_GROUPS]: the plane fits the root pass leaves to lk_ov_fit_lane_kernel (current bucket)
    int4* jobhdr;                // [S][LK_INLINE_GROUPS][hash_cap]: their headers {leaf, block, cnt, decided}, dense (what the fit passes scan)
    struct LkLeafSum* sums;      // [S][hash_cap]: moment sums of a leading part of a private ROOT leaf's points (lk_ov_root_lane_kernel)
    struct LkLeafSum* base_sums; // [base max_nodes], shared: the same for the BASE map's root leaves, once per replay (lk_ov_base_sums_kernel)
    int* cplx;                   // [S][2 * scan_cap]: {root, index in the touched list} of the roots the fast root pass leaves to the generic one
    int* ptroot;                 // [S][scan_cap]: per bucket point, the private root it was queued on in a slot line (-1: dropped, or queued in the overflow list)
    unsigned int hash_cap, nodes_cap, blocks_cap, scan_cap, bit_words;
};

__host__ __device__ inline LkMap ov_slot_map(const LkOverlay& ov, unsigned int slot) {
    LkMap m = {};
    const size_t s = slot;
    m.hash = nullptr;            // the private table is ov.keys (ov_key_find / ov_key_claim)
    m.planes = ov.planes + s * ov.nodes_cap;
    m.match = ov.match + s * ov.nodes_cap;
    m.nodes = ov.nodes + s * ov.nodes_cap;
    m.blocks = ov.blocks + s * ov.blocks_cap;
    m.counters = ov.counters + s * LK_CTR_COUNT;
    m.touched = ov.touched + s * ov.scan_cap;
    m.heavy = ov.cplx + s * ov.scan_cap * 2;   // the fast root pass's hand-over list (counter LK_CTR_HEAVY)
    m.next = ov.next + s * ov.scan_cap;
    m.slots = ov.slots + s * ov.hash_cap * LK_SLOTS * 4;   // 16-B entries (dev_insert_root<.., OV> reads them as float4)
    m.scratch = ov.scratch + s * ov.scan_cap;
    m.groups = ov.groups + s * ov.scan_cap * 32;
    m.gidx = ov.gidx + s * ov.scan_cap;
    m.free_list = ov.free_list + s * ov.blocks_cap;
    m.freed_next = ov.freed_next + s * ov.blocks_cap;
    m.hash_mask = ov.hash_cap - 1;
    m.max_nodes = ov.nodes_cap;
    m.max_blocks = ov.blocks_cap;
    m.max_scan = ov.scan_cap;
    m.dirty = ov.dirty + s * ov.hash_cap;
    m.newroot = ov.newroot;
    m.spec = ov.spec;
    m.epoch = 0;
    m.grid_on = 0;
    return m;
}

// 3 x 21-bit two's complement fields; false when a component does not fit (|key| >= 2^20 voxels)
__host__ __device__ inline bool ov_pack_key(int kx, int ky, int kz, unsigned long long* out) {
    const int lim = 1 << 20;
    if (kx < -lim || kx >= lim || ky < -lim || ky >= lim || kz < -lim || kz >= lim) return false;
    *out = ((unsigned long long)((unsigned int)kx & 0x1fffffu)) | ((unsigned long long)((unsigned int)ky & 0x1fffffu) << 21) |
           ((unsigned long long)((unsigned int)kz & 0x1fffffu) << 42);
    return true;
}
__host__ __device__ inline void ov_unpack_key(unsigned long long k, int* key) {
    for (int c = 0; c < 3; ++c) {
        const int f = (int)((k >> (21 * c)) & 0x1fffffu);
        key[c] = (f & 0x100000) ? f - 0x200000 : f;
    }
}
__device__ __forceinline__ unsigned int ov_slot_hash(int kx, int ky, int kz) { return lk_hash3(kx, ky, kz); }
// plain loads, two consecutive entries per round trip; the entry index (= the root's node id) or -1.  Entries only ever go from
// empty to a key, so a stale view can only miss a key that was claimed in the running launch.
__device__ __forceinline__ int ov_key_find(const unsigned long long* __restrict__ keys, unsigned int mask, unsigned long long pk, unsigned int h) {
    unsigned int s = h & mask;
    for (unsigned int probe = 0; probe <= mask; probe += 2) {
        const unsigned long long e0 = keys[s], e1 = keys[(s + 1) & mask];
        if (e0 == pk) return (int)s;
        if (e0 == LK_OV_EMPTY) return -1;
        if (e1 == pk) return (int)((s + 1) & mask);
        if (e1 == LK_OV_EMPTY) return -1;
        s = (s + 2) & mask;
    }
    return -1;
}
// find-or-claim: one relaxed 64-bit CAS per probed entry that looks empty; -1 = table full.  *claimed: THIS lane's CAS put the key in
// (exactly one lane per new key ever sees that)
__device__ __forceinline__ int ov_key_claim(unsigned long long* keys, unsigned int mask, unsigned long long pk, unsigned int h, bool* claimed) {
    unsigned int s = h & mask;
    *claimed = false;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        unsigned long long e = keys[s];
        if (e == LK_OV_EMPTY) {
            e = atomicCAS(&keys[s], LK_OV_EMPTY, pk);   // returns the old value: empty -> we claimed it
            if (e == LK_OV_EMPTY) {
                *claimed = true;
                return (int)s;
            }
        }
        if (e == pk) return (int)s;
        s = (s + 1) & mask;
    }
    return -1;
}

// ---------------------------------------------------------------- start of a replay: every slot's private map is empty
__device__ __forceinline__ void ov_root_record_reset(lk_node_rec* nd) {   // a root record's bucket-local queue and "exists" flag
    nd->list_head = -1;
    nd->pad_[LK_PAD_QCOUNT] = 0;
    nd->pad_[LK_PAD_LIVE] = 0;
    nd->pad_[LK_PAD_BASE] = 0;
    // "no block": what lk_ov_point_geom_kernel (thread per point, no error test of its own) sees in a root the re-projection has claimed but
    // lk_ov_materialise_kernel never filled in because the slot's pools had overflowed - whatever the memory held before would be a block id
#ifndef LK_X_NO_BLOCK_RESET   // (build switch of the regression check only: tools/gpu_r05_poison.sh shows the fault without this line)
    nd->block = -1;
#endif
}
// once per allocation: every table entry empty, every root record's queue fields clean
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(256) lk_ov_init_kernel(LkOverlay ov) {
    const unsigned int slot = blockIdx.y;
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ov.hash_cap) {
        ov.keys[(size_t)slot * ov.hash_cap + i] = LK_OV_EMPTY;
        ov_root_record_reset(&ov.nodes[(size_t)slot * ov.nodes_cap + i]);
    }
}
#endif
// per replay: only the entries the PREVIOUS replay claimed are touched (a 100 000-point scan claims ~5 000 of 32 768: resetting every
// root record cost 3 ms per 1024-scan batch, this pass reads the key tables - 8 B per entry - and rewrites the claimed records)
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(256) lk_ov_reset_kernel(LkOverlay ov) {
    const unsigned int slot = blockIdx.y;
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ov.hash_cap) {
        unsigned long long* k = &ov.keys[(size_t)slot * ov.hash_cap + i];
        if (*k != LK_OV_EMPTY) {
            *k = LK_OV_EMPTY;
            ov_root_record_reset(&ov.nodes[(size_t)slot * ov.nodes_cap + i]);
        }
    }
    if (i < ov.bit_words) ov.bits[(size_t)slot * ov.bit_words + i] = 0u;
    if (i < LK_CTR_COUNT) ov.counters[(size_t)slot * LK_CTR_COUNT + i] = (i == LK_CTR_NODES) ? ov.hash_cap : 0u;
    if (i == 0) {
        // the record an exhausted node pool clamps its ids to (create_child, ov_copy_node) must be a sane empty node: the call fails
        // with LK_ERR_CAPACITY, but nothing may follow a garbage child or block id on the way there
        const size_t last = (size_t)slot * ov.nodes_cap + ov.nodes_cap - 1;
        lk_node_rec* nd = &ov.nodes[last];
        for (int c = 0; c < 8; ++c) nd->child[c] = -1;
        nd->voxel_center[0] = nd->voxel_center[1] = nd->voxel_center[2] = 0.0;
        nd->quater_length = 0.f;
        nd->layer = LK_MAX_LAYER, nd->npts = 0, nd->new_points = 0, nd->state = 0, nd->block = -1;
        nd->list_head = -1;
        for (int c = 0; c < 8; ++c) nd->pad_[c] = 0;
        ov.planes[last].flags = 0;
        ov.match[last].flags = 0;
    }
}
#endif

// start of a bucket's insert in every slot
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_WAVE) lk_ov_begin_kernel(LkOverlay ov) {
    const LkMap pm = ov_slot_map(ov, blockIdx.x);
    dev_bucket_begin_wave(pm);
}
#endif

// ---------------------------------------------------------------- re-projection + hashing half of the insert
// Would UpdateOctoTree ignore this point (voxel_map.cc:185-241)?  The walk of dev_reproject_point: down through initialised
// non-planar nodes below max_layer (they never change again) to the node the point would be pushed into; ignored iff that node
// is frozen.
// (Round 5, measured and not kept: the node's fields requested together, one round trip per node instead of one per field - 3.4 -> 3.9 ms per
// batch.  At eight waves per SIMD this pass is bound by the NUMBER of memory requests, not by their latency: the compiler's habit of fetching
// a field only behind the test that needs it is the cheaper one here.)
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

// Which voxels of the base map are frozen leaves (a plane, or a non-planar leaf at max_layer, with update_enable_ == false,
// voxel_map.cc:199,232): UpdateOctoTree ignores every point that lands in one, for good - so no scan ever gets a private copy of such
// a voxel, and the re-projection pass can drop those points (half of a scan on the bench's mature map) after ONE bit test instead of
// the private-table probe, the base table probe and the walk.  One thread per entry of the base map's root table, once per replay.
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(256) lk_ov_frozen_bits_kernel(LkMap base, unsigned int n_hash, int max_layer, unsigned int* __restrict__ frozen, int root_bits) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_hash) return;
    const int4 e = base.hash[i];
    if (e.w < 0) return;
    const unsigned int st = base.nodes[e.w].state, pf = base.planes[e.w].flags;
    // ov_walk_ignored at the root: not initialised -> the point is taken here; a leaf (plane, or max_layer) -> ignored iff update_enable_ is off;
    // anything else descends (no bit: the re-projection walks the tree)
    unsigned int cls;
    if (!(st & LK_NODE_INIT_OCTO)) cls = 2u;
    else if ((pf & LK_PLANE_IS_PLANE) != 0 || base.nodes[e.w].layer >= max_layer) cls = (st & LK_NODE_UPDATE_ENABLE) ? 2u : 1u;
    else return;
    if (cls == 2u && !root_bits) return;   // LEGKILO_OV_ROOT_BITS=0 (A/B): such a voxel through the walk, as before round 6
    const int key[3] = {e.x, e.y, e.z};
    unsigned int cell;
    if (ov_cell_of(base, key, &cell)) atomicOr(&frozen[cell >> 4], cls << (2u * (cell & 15u)));
}
#endif

// KILO.cc:216-230 + the hashing half of UpdateVoxelMap (voxel_map.cc:343-358) for bucket point i of slot blockIdx.y, on the
// slot's overlay.  A private root that exists (created by an earlier bucket: LK_PAD_LIVE) is walked itself; otherwise the base
// voxel of the key, if any, is still the truth - the insert's first phase is read-only on every tree.  A point that is not ignored
// claims its key in the slot's table (if nobody has) and is queued on that entry = root record; lk_ov_materialise_kernel then
// creates the roots that do not exist yet.
// returns the private root the point was queued on in a slot line, or -1 (ignored, or queued in a long root's overflow list)
__device__ __forceinline__ int ov_reproject_point(const LkMap& base, const LkOverlay& ov, const LkParams& pr, const LkFilter* __restrict__ filters,
                                                  const lk_point* __restrict__ pts, const int i, const unsigned int slot) {
    const LkMap pm = ov_slot_map(ov, slot);
    unsigned long long* keys = ov.keys + (size_t)slot * ov.hash_cap;
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    const float4 p = reinterpret_cast<const float4*>(pts)[i];
    const V3 pw = point_world(p.x, p.y, p.z, bc, pr);
    int key[3];
    key_floor(pw, pr.voxel_size_f, key);
    unsigned long long pk;
    if (!ov_pack_key(key[0], key[1], key[2], &pk)) {
        atomicOr(&pm.counters[LK_CTR_ERR], LK_E_KEY_RANGE);
        return -1;
    }
    unsigned int cell = 0;
    const bool in_grid = ov_cell_of(base, key, &cell);
    // the base voxel of this key is a frozen leaf: the point is ignored, and no private voxel of that key can exist
    const unsigned int cls = in_grid ? (ov.frozen[cell >> 4] >> (2u * (cell & 15u))) & 3u : 0u;
    if (cls & 1u) return -1;
    const bool base_takes_at_root = (cls & 2u) != 0;   // a base voxel exists and takes the point at its root: no walk, its node id only for the lane that claims the key
    const unsigned int hk = ov_slot_hash(key[0], key[1], key[2]);
    int root = ov_key_find(keys, pm.hash_mask, pk, hk);
    if (root >= 0 && pm.nodes[root].pad_[LK_PAD_LIVE] != 0) {
        if (ov_walk_ignored(pm, root, pw, pr.max_layer)) return -1;
    } else {
        // the base map's voxel of this key: its node id sits in the frozen-map grid cell (arithmetic index, 4 bytes; a key outside the grid's
        // box has no base voxel) - no probe of the base table
        int broot = -1;
        if (in_grid && !base_takes_at_root) {
            const unsigned int bnode = base.match[(size_t)base.grid_base + cell].pad_;
            if (bnode != LK_GRID_EMPTY) broot = (int)bnode;
            if (broot >= 0 && ov_walk_ignored(base, broot, pw, pr.max_layer)) return -1;
        }
        if (root < 0) {
            bool claimed;
            root = ov_key_claim(keys, pm.hash_mask, pk, hk, &claimed);
            if (root < 0) {
                atomicOr(&pm.counters[LK_CTR_ERR], LK_E_HASH_FULL);
                return -1;
            }
            // what the materialise pass copies into this root (no lookup there): stored by the ONE lane that claimed the key - a store
            // per point cost 2.8 ms per 1024-scan batch
            if (claimed) {
                if (base_takes_at_root) broot = (int)base.match[(size_t)base.grid_base + cell].pad_;
                pm.nodes[root].pad_[LK_PAD_BASE] = (unsigned int)(broot + 1);
            }
        }
    }
    // queue_point_on_root with the POINT in the slot line ({x, y, z, index}): the root pass gets a root's points with one coalesced read
    // of its line instead of a gather from the scan (only the overflow of a root with more than LK_SLOTS points is kept by index)
    const unsigned int k = atomicAdd(&pm.nodes[root].pad_[0], 1u);
    if (k < (unsigned int)LK_SLOTS) {
        reinterpret_cast<float4*>(pm.slots)[(size_t)root * LK_SLOTS + k] = make_float4(p.x, p.y, p.z, __int_as_float(i));
    } else {
        pm.next[i] = atomicExch(&pm.nodes[root].list_head, i);
    }
    if (k == 0) {
        const unsigned int t = atomicAdd(&pm.counters[LK_CTR_TOUCHED], 1u);
        pm.touched[t] = root;
    }
    return k < (unsigned int)LK_SLOTS ? root : -1;
}
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_WAVE)
    lk_ov_reproject_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* __restrict__ filters, LkPtSrc src) {
    const int i = blockIdx.x * LK_WAVE + threadIdx.x;
    const lk_point* pts;
    const int n = ov_pt_src(src, blockIdx.y, &pts);
    if (i >= n) return;
    const int r = ov_reproject_point(base, ov, pr, filters, pts, i, blockIdx.y);
    ov.ptroot[(size_t)blockIdx.y * ov.scan_cap + i] = r;   // for lk_ov_point_geom_kernel
}
#endif

// ---------------------------------------------------------------- copy-on-write of a base voxel's octree
// One wave copies node `src` of the base map (and, recursively, its children) into node `dst` of the private map, in TWO memory round
// trips per node: (1) the source's node record (one 16-B chunk per lane 0..7), plane record (lanes 8..23) and match record (lanes
// 24..32) are requested together; (2) the live leaf's points (<= 8 doubles per lane), the private block id and the children's node
// ids (ONE bump of the node counter for all of them) are requested together; then everything is stored.  Child and block ids are
// renumbered.  The root (L == 0) keeps the queue fields the re-projection pass has filled in (list_head, pad_[]).
template <int L>
__device__ __forceinline__ void ov_copy_node(const LkMap& pm, const LkMap& base, const int src, const int dst, const int pre_block = -2) {
    const int lane = threadIdx.x & 63;
    // ---- trip 1
    int4 rec = make_int4(0, 0, 0, 0);
    uint4 pv = make_uint4(0, 0, 0, 0);
    if (lane < 8) rec = reinterpret_cast<const int4*>(&base.nodes[src])[lane];
    else if (lane < 24) pv = reinterpret_cast<const uint4*>(&base.planes[src])[lane - 8];
    else if (lane < 33) pv = reinterpret_cast<const uint4*>(&base.match[src])[lane - 24];
    int child[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int comp = (c & 3) == 0 ? rec.x : (c & 3) == 1 ? rec.y : (c & 3) == 2 ? rec.z : rec.w;
        child[c] = __builtin_amdgcn_readlane(comp, c >> 2);
    }
    const int s_npts = __builtin_amdgcn_readlane(rec.x, 4), s_block = __builtin_amdgcn_readlane(rec.w, 4);   // bytes 64..79: npts, new_points, state, block
    int n_child = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) n_child += child[c] >= 0 ? 1 : 0;
    // ---- trip 2
    const int nd8 = s_block >= 0 ? min(max(s_npts, 0), LK_BLOCK_PTS) * 9 : 0;   // lk_pt_rec = 9 doubles; only the first npts records hold points
    double pt[8];
    {
        const double* sp = reinterpret_cast<const double*>(&base.blocks[s_block >= 0 ? s_block : 0]);
#pragma unroll
        for (int k = 0; k < 8; ++k) pt[k] = (lane + 64 * k < nd8) ? sp[lane + 64 * k] : 0.0;
    }
    int nblock = -1, cbase = -1;
    if (lane == 0) {
        if (s_block >= 0) nblock = pre_block != -2 ? pre_block : pop_or_bump_block(pm);   // pre_block: allocated by the caller for a whole chunk of roots
        if (n_child > 0) {
            unsigned int nn = atomicAdd(&pm.counters[LK_CTR_NODES], (unsigned int)n_child);
            if (nn + (unsigned int)n_child > pm.max_nodes) {
                atomicOr(&pm.counters[LK_CTR_ERR], LK_E_NODES_FULL);
                nn = pm.max_nodes - (unsigned int)n_child;   // memory-safe; the error flag fails the call (lk_ov_reset_kernel keeps the last record sane)
            }
            cbase = (int)nn;
        }
    }
    nblock = bcast0(nblock), cbase = bcast0(cbase);
    // ---- stores
    if (lane >= 8 && lane < 24) reinterpret_cast<uint4*>(&pm.planes[dst])[lane - 8] = pv;
    else if (lane >= 24 && lane < 33) reinterpret_cast<uint4*>(&pm.match[dst])[lane - 24] = pv;
    if (nd8 > 0) {
        double* dp = reinterpret_cast<double*>(&pm.blocks[nblock]);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (lane + 64 * k < nd8) dp[lane + 64 * k] = pt[k];
    }
    int nchild[8];
    {
        int run = cbase;
#pragma unroll
        for (int c = 0; c < 8; ++c) nchild[c] = child[c] >= 0 ? run++ : -1;
    }
    int4* drec = reinterpret_cast<int4*>(&pm.nodes[dst]);
    if (lane == 0) drec[0] = make_int4(nchild[0], nchild[1], nchild[2], nchild[3]);
    else if (lane == 1) drec[1] = make_int4(nchild[4], nchild[5], nchild[6], nchild[7]);
    else if (lane == 2 || lane == 3) drec[lane] = rec;                        // voxel_center, quater_length, layer
    else if (lane == 4) drec[4] = make_int4(rec.x, rec.y, rec.z, nblock);     // npts, new_points, state, block
    else if (lane == 5) {
        if (L > 0) drec[5] = make_int4(rec.x, rec.y, rec.z, -1);              // key, list_head
        else pm.nodes[dst].key[0] = rec.x, pm.nodes[dst].key[1] = rec.y, pm.nodes[dst].key[2] = rec.z;
    } else if ((lane == 6 || lane == 7) && L > 0) drec[lane] = make_int4(0, 0, 0, 0);   // pad_: a child queues nothing
    if constexpr (L < LK_MAX_LAYER) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (child[c] >= 0) ov_copy_node<L + 1>(pm, base, child[c], nchild[c]);
    }
}
// Copy-on-write of the touched roots that do not exist in the slot's map yet.  A wave takes 64 of them, ONE LANE EACH:
//   * no voxel of that key in the base map: an empty root voxel is created (voxel_map.cc:345-357);
//   * a CHILDLESS base voxel (a leaf: nearly all of them): the lane copies the node record, all lanes together copy the plane and match
//     records of the chunk's voxels as one flat list of 16-B pieces (25 per voxel, independent loads), and the voxel's points are NOT
//     copied here: the root becomes "thin" (LK_PAD_LIVE = 2) and the root pass of this bucket, which has to read those points anyway,
//     reads them from the base block and writes old + new points to the private block (dev_insert_root<.., OV>).  Copying every leaf's
//     points in this pass (one voxel after the other, two dependent round trips each) took 7.1 ms of a 33 ms batch;
//   * a base voxel WITH children (a cut voxel): its octree is copied node by node (ov_copy_node), all lanes on one voxel.
// Point blocks are allocated with ONE bump of the block counter per chunk.  Then the key's bit is set for the residual pass.
// LEAN (the fast root path, lk_ov_root_lane_kernel): a childless base voxel's private copy is its NODE record and one 16-B piece of its plane
// record (d, radius, flags; points_size = LK_PLANE_LAZY) - nothing else.  Its match record, its moment sums and its "has a private root" bit
// for the residual pass are not made here: until the leaf's plane is fitted again the base map's plane IS the scan's plane, so the residual
// pass keeps matching the base grid cell (the bit is set by the root pass when it refits the leaf or hands the root to the generic pass, which
// also gets the match record made first); the sums are read from the base map's (LK_PAD_SUMSRC).  35 -> 15 memory requests per new root.
template <bool LEAN>
__device__ __forceinline__ void ov_materialise_body(const LkMap& base, const LkOverlay& ov, const LkParams& pr, const unsigned int slot_, const int bx_, const int gx_, const int tid_,
                                                    const int mb_ = LK_MB /* threads per workgroup */) {
    const unsigned int slot = slot_;
    const LkMap pm = ov_slot_map(ov, slot);
    if (pm.counters[LK_CTR_ERR]) return;
    const unsigned long long* keys = ov.keys + (size_t)slot * ov.hash_cap;
    unsigned int* bits = ov.bits + (size_t)slot * ov.bit_words;
    const int lane = tid_ & 63;
    const int wave = (int)((bx_ * mb_ + tid_) >> 6), nwaves = (int)((gx_ * mb_) >> 6);
    const int n_touched = (int)pm.counters[LK_CTR_TOUCHED];
    for (int t0 = wave * LK_WAVE; t0 < n_touched; t0 += nwaves * LK_WAVE) {
        const int tt = t0 + lane;
        int my_root = -1, my_base = 0;
        bool need = false;
        unsigned long long my_key = 0ull;
        if (tt < n_touched) {
            my_root = pm.touched[tt];
            const lk_node_rec* nd = &pm.nodes[my_root];
            need = nd->pad_[LK_PAD_LIVE] == 0;
            my_base = (int)nd->pad_[LK_PAD_BASE];
            my_key = keys[my_root];
        }
        // the base voxel's node record, whole (six 16-B pieces per lane, one round trip for the chunk)
        const bool has_base = need && my_base > 0;
        int4 rec[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) rec[c] = make_int4(-1, -1, -1, -1);
        if (has_base) {
            const int4* bp = reinterpret_cast<const int4*>(&base.nodes[my_base - 1]);
#pragma unroll
            for (int c = 0; c < 6; ++c) rec[c] = bp[c];
        }
        const int s_block = has_base ? rec[4].w : -1, s_npts = has_base ? rec[4].x : 0;   // bytes 64..79: npts, new_points, state, block
        const bool childless = rec[0].x < 0 && rec[0].y < 0 && rec[0].z < 0 && rec[0].w < 0 && rec[1].x < 0 && rec[1].y < 0 && rec[1].z < 0 && rec[1].w < 0;
        const bool thin = has_base && childless;
        // one bump of the block counter serves the whole chunk (a returning device-scope atomic is a ~2 us round trip)
        // (a root the base map has no voxel for gets its block here too: it was created because a point is waiting for it, and a returning
        // atomic in the root pass is a 2-us round trip on that root's chain)
        const bool want_block = need && (s_block >= 0 || my_base == 0);
        const unsigned long long blk_mask = __ballot(want_block);
        int blk_base = 0;
        if (blk_mask) {
            const int total = __popcll(blk_mask);
            if (lane == 0) {
                unsigned int bb = atomicAdd(&pm.counters[LK_CTR_BLOCKS], (unsigned int)total);
                if (bb + (unsigned int)total > pm.max_blocks) {
                    atomicOr(&pm.counters[LK_CTR_ERR], LK_E_BLOCKS_FULL);
                    bb = pm.max_blocks - (unsigned int)total;   // memory-safe; the error flag fails the call
                }
                blk_base = (int)bb;
            }
            blk_base = bcast0(blk_base);
        }
        const int my_block = want_block ? blk_base + __popcll(blk_mask & ((1ull << lane) - 1ull)) : -1;
        int key[3] = {0, 0, 0};
        if (need) ov_unpack_key(my_key, key);
        // roots the base map has no voxel for: an empty root voxel each, written by the root's own lane
        if (need && my_base == 0) {
            lk_node_rec* nd = &pm.nodes[my_root];
            const double vs = (double)pr.voxel_size_f;
#pragma unroll
            for (int c = 0; c < 8; ++c) nd->child[c] = -1;
#pragma unroll
            for (int c = 0; c < 3; ++c) nd->voxel_center[c] = (0.5 + key[c]) * vs, nd->key[c] = key[c];  // voxel_map.cc:355-357
            nd->quater_length = pr.voxel_size_f / 4;                                                        // voxel_map.cc:354
            nd->layer = 0, nd->npts = 0, nd->new_points = 0, nd->state = LK_NODE_UPDATE_ENABLE, nd->block = my_block;
            pm.planes[my_root].flags = 0;
            pm.match[my_root].flags = 0;
        }
        if (need) {
            pm.nodes[my_root].pad_[LK_PAD_SPLIT] = 0;   // (a word of the previous replay may still be there)
            // where the root's moment sums are (lk_ov_root_lane_kernel): the base leaf's, made once per replay, for a childless base voxel; none otherwise
            pm.nodes[my_root].pad_[LK_PAD_SUMSRC] = (has_base && childless) ? 1u : 2u;
        }
        // childless base voxels: the node record by the root's own lane (the queue fields list_head / pad_[] are the re-projection pass's) ...
        const bool pending = thin && s_block >= 0 && s_npts > 0;
        if (thin) {
            int4* drec = reinterpret_cast<int4*>(&pm.nodes[my_root]);
            drec[0] = rec[0], drec[1] = rec[1], drec[2] = rec[2], drec[3] = rec[3];
            drec[4] = make_int4(rec[4].x, rec[4].y, rec[4].z, my_block);
            pm.nodes[my_root].key[0] = rec[5].x, pm.nodes[my_root].key[1] = rec[5].y, pm.nodes[my_root].key[2] = rec[5].z;
            pm.nodes[my_root].pad_[LK_PAD_COWBLK] = pending ? (unsigned int)(s_block + 1) : 0u;
        }
        // ... and (not LEAN) their match records (9 pieces of 16 B) as one flat list over the chunk's thin voxels, five loads in flight per lane.
        // The 256-B PLANE record is never copied (round 5): every reader on the replay's path wants its flags word only - plane piece 3
        // {d, radius, flags, points_size = LK_PLANE_LAZY} is written, a fit rewrites the whole record, lk_ov_merge_split_kernel fetches the base
        // map's record for an export of a voxel that was never refitted
        if (LEAN) {
            if (thin) {
                uint4 v = reinterpret_cast<const uint4*>(&base.planes[my_base - 1])[3];
                v.w = (unsigned int)LK_PLANE_LAZY;
                reinterpret_cast<uint4*>(&pm.planes[my_root])[3] = v;
            }
        } else {
            const unsigned long long thin_mask = __ballot(thin);
            const int n_thin = __popcll(thin_mask);
            const int pos = thin ? __popcll(thin_mask & ((1ull << lane) - 1ull)) : 63;   // n_thin == 64: every lane is a member
            const int c_src = __builtin_amdgcn_ds_permute(pos << 2, my_base - 1), c_dst = __builtin_amdgcn_ds_permute(pos << 2, my_root);
            constexpr int PIECES = 9, U = 5;
            const int total = n_thin * PIECES;
            for (int j0 = 0; j0 < total; j0 += 64 * U) {
                uint4 v[U];
                int dsts[U], cs[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = j0 + 64 * u + lane;
                    const int r = idx / PIECES;
                    cs[u] = idx - r * PIECES;
                    const int src = __shfl(c_src, r & 63, LK_WAVE);
                    dsts[u] = __shfl(c_dst, r & 63, LK_WAVE);
                    v[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (idx < total) v[u] = reinterpret_cast<const uint4*>(&base.match[src])[cs[u]];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = j0 + 64 * u + lane;
                    if (idx < total) {
                        reinterpret_cast<uint4*>(&pm.match[dsts[u]])[cs[u]] = v[u];
                        if (cs[u] == 3) reinterpret_cast<uint4*>(&pm.planes[dsts[u]])[3] = make_uint4(v[u].x, v[u].y, v[u].z, (unsigned int)LK_PLANE_LAZY);
                    }
                }
            }
        }
        // cut voxels: the whole octree, one voxel after the other, all lanes on one voxel
        unsigned long long todo = __ballot(has_base && !thin);
        while (todo) {
            const int src_lane = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            ov_copy_node<0>(pm, base, __builtin_amdgcn_readlane(my_base, src_lane) - 1, __builtin_amdgcn_readlane(my_root, src_lane),
                            __builtin_amdgcn_readlane(my_block, src_lane));
        }
        if (need) {
            pm.nodes[my_root].pad_[LK_PAD_LIVE] = pending ? 2u : 1u;
            unsigned int cell;
            // LEAN: only a voxel the base map does not have at all gets its bit here - its mere EXISTENCE changes what the residual pass does
            // (a point whose home voxel exists but gives no match tries one neighbour voxel, KILO.cc:152-178; without a home voxel it
            // does not), plane or no plane.  A copied voxel exists in the base map, too: its bit waits for its first change
            if ((!LEAN || my_base == 0) && ov_cell_of(base, key, &cell)) atomicOr(&bits[cell >> 5], 1u << (cell & 31u));
        }
        const int n_new = __popcll(__ballot(need));
        if (lane == 0 && n_new) atomicAdd(&pm.counters[LK_CTR_ROOTS], (unsigned int)n_new);
    }
}
template <bool LEAN>
__global__ void __launch_bounds__(LK_MB) lk_ov_materialise_kernel(LkMap base, LkOverlay ov, LkParams pr) {
    ov_materialise_body<LEAN>(base, ov, pr, blockIdx.y, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}

// ---------------------------------------------------------------- the plane fits of the root pass, ONE LANE PER FIT
// The wave-per-root pass (dev_insert_root<.., OV>, lk_map_kernels.h) spreads a leaf's points over the lanes - right for the per-point work
// and for coalesced access - but the fit that ends a leaf's bucket is serial work every lane repeats for ONE leaf: the eigen-decomposition
// of its scatter matrix (23 % of the pass's time by switching pieces off), the 21 wave-reduced sums of plane_var (12 %), the commit.  In a
// batch replay a launch holds ~10^6 such fits, so the root pass only DECIDES (apply_leaf<DEFER>: the refit events' is_plane tests, which is
// all its control flow needs) and leaves a 96-B job per fit - the leaf, its block, the event's point count, decision and moment sums.  Here
// a LANE takes a job: init_plane (voxel_map.cc:42-117) from the sums - plane_test_regs' / plane_var_regs' / plane_commit's expressions,
// the points read back from the leaf's block (a block retired by a freeze is not handed out again before the next bucket).
// (Measured and not kept: the whole root pass one lane per root - queue, sums, state machine - is bound by memory TRANSACTIONS, every
// lane's 8-B access its own: 9.4 + 2.2 ms against 10.8 ms for the wave-per-root pass, 512 scans; profiles/EXPERIMENTS.md.)
#ifndef LK_FIT_WAVES
#define LK_FIT_WAVES 4
#endif

// (Sorting a workgroup's jobs by point count, so that a wave's lanes run similar loop lengths, changes nothing: 4.04 vs 3.99 ms per 1024
// scans - the pass moves 1.7 KB per fit in 16-B pieces, 2 TB/s.)
// First half of a fit, one lane per job: init_plane's centroid, scatter matrix and eigen-decomposition (voxel_map.cc:46-66) from the event's moment
// sums.  A "not a plane" event ends here (flag cleared); for a plane the centre, the three eigenvectors and eigenvalues are left in the leaf's
// private plane record (centre, normal = v_min; v_mid, v_max and the eigenvalues in the first nine plane_var words) for lk_ov_fit_lane_kernel,
// which overwrites them with the finished plane.  Two kernels because the closed-form eigen-solver (acos, two cos) and the loop over the leaf's
// points each fit 128 registers and together do not: the single kernel ran at two waves per SIMD.
// (thread i_first of i_stride over the slot's jobs: the kernel below, and one phase of lk_ov_tail_kernel)
__device__ __forceinline__ void ov_fit_eig_body(const LkMap& base, const LkOverlay& ov, const LkParams& pr, const unsigned int slot, const int i_first, const int i_stride) {
    const LkMap pm = ov_slot_map(ov, slot);
    if (pm.counters[LK_CTR_ERR]) return;
    const int n_touched = (int)pm.counters[LK_CTR_TOUCHED];
    const LkFitJob* jobs = ov.jobs + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;
    const int4* jhdr = ov.jobhdr + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;
    for (int i = i_first; i < n_touched * LK_INLINE_GROUPS; i += i_stride) {
        const int g = i / n_touched, t = i - g * n_touched;
        const LkFitJob* job = &jobs[(size_t)g * ov.hash_cap + t];
        const int4 hd = jhdr[(size_t)g * ov.hash_cap + t];
        const int root = hd.x, cnt = hd.z;
        if (cnt <= 0) continue;
        lk_plane_rec* pl = &pm.planes[root];
        if (hd.w == 0) {   // plane_commit's "not a plane" branch
            // init_plane's else-branch keeps the previous fit's centre, normal and plane_var (voxel_map.cc:112-115).  A lazily copied root
            // (lk_ov_materialise_kernel<LEAN>) has them in the BASE map's record only: make them private before the marker that says so goes
            // (lk_overlay_export would otherwise emit whatever the pool held)
            if (pl->points_size == LK_PLANE_LAZY) {
                const int bnode = (int)pm.nodes[root].pad_[LK_PAD_BASE] - 1;
                if (bnode >= 0) {
                    const uint4* bp4 = reinterpret_cast<const uint4*>(&base.planes[bnode]);
                    uint4* pp4 = reinterpret_cast<uint4*>(pl);
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (q != 3) pp4[q] = bp4[q];   // piece 3 = d, radius, flags, points_size: d / radius / flags are private already
                }
            }
            pl->points_size = cnt;
            const unsigned int fl = pl->flags & ~LK_PLANE_IS_PLANE;
            pl->flags = fl;
            pm.match[root].flags = fl;
            continue;
        }
        PlaneFit ev;
#pragma unroll
        for (int q = 0; q < 9; ++q) ev.s9[q] = job->s9[q];
        const PlaneFit fit = plane_test_regs<false, true>(nullptr, false, cnt, pr.planer_threshold, &ev);
#pragma unroll
        for (int k = 0; k < 3; ++k) pl->center[k] = fit.c[k], pl->normal[k] = fit.vmin[k], pl->plane_var[k] = fit.vmid[k], pl->plane_var[3 + k] = fit.vmax[k];
        pl->plane_var[6] = fit.emin, pl->plane_var[7] = fit.emid, pl->plane_var[8] = fit.emax;
    }
}
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_WAVE, 5) lk_ov_fit_eig_kernel(LkMap base, LkOverlay ov, LkParams pr) {
    ov_fit_eig_body(base, ov, pr, blockIdx.y, (int)(blockIdx.x * LK_WAVE + threadIdx.x), (int)(gridDim.x * LK_WAVE));
}
#endif
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_WAVE, LK_FIT_WAVES) lk_ov_fit_lane_kernel(LkMap base, LkOverlay ov, LkParams pr) {
    const unsigned int slot = blockIdx.y;
    const LkMap pm = ov_slot_map(ov, slot);
    if (pm.counters[LK_CTR_ERR]) return;
    const int n_touched = (int)pm.counters[LK_CTR_TOUCHED];
    const LkFitJob* jobs = ov.jobs + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;   // entry [g][t]: inline leaf group g of touched root t
    for (int i = blockIdx.x * LK_WAVE + threadIdx.x; i < n_touched * LK_INLINE_GROUPS; i += gridDim.x * LK_WAVE) {
        const int g = i / n_touched, t = i - g * n_touched;
        const LkFitJob* job = &jobs[(size_t)g * ov.hash_cap + t];
        const int4 hd = (ov.jobhdr + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS)[(size_t)g * ov.hash_cap + t];
        const int root = hd.x /* the leaf's node id */, block = hd.y, cnt = hd.z;
        if (cnt <= 0) continue;
        if (hd.w == 0) continue;   // the event said "not a plane": lk_ov_fit_eig_kernel has cleared the flag, there is no plane_var to make
        // centre, eigenvectors and eigenvalues of the fit: left in the leaf's private plane record by lk_ov_fit_eig_kernel
        PlaneFit fit;
        {
            const lk_plane_rec* pl = &pm.planes[root];
#pragma unroll
            for (int k = 0; k < 3; ++k) fit.c[k] = pl->center[k], fit.vmin[k] = pl->normal[k], fit.vmid[k] = pl->plane_var[k], fit.vmax[k] = pl->plane_var[3 + k];
            fit.emin = pl->plane_var[6], fit.emid = pl->plane_var[7], fit.emax = pl->plane_var[8];
        }
        fit.is_plane = true;
        double acc21[21];
#pragma unroll
        for (int q = 0; q < 21; ++q) acc21[q] = 0.0;
        if (fit.is_plane) {
            // plane_var = sum_i J_i var_i J_i^T (voxel_map.cc:76-95) with J_i = [A_i ; I / n], A_i = v_mid FA_i^T + v_max FB_i^T (rank 2; the row of
            // v_min is zero), FA_i = ((q.v_mid) v_min + (q.v_min) v_mid) / den_A, FB_i likewise with v_max, q = p_i - centre.  With a = var FA, b = var FB:
            //   A var A^T = v_mid v_mid^T (FA.a) + (v_mid v_max^T + v_max v_mid^T) (FA.b) + v_max v_max^T (FB.b),   A var = v_mid a^T + v_max b^T
            // so the 21 sums of 3 x 3 products per point become 15 running sums of a few dot products - the same value (a different, equally
            // valid rounding), a quarter of the arithmetic and half the registers: four waves per SIMD instead of two
            const double invA = 1.0 / (cnt * (fit.emin - fit.emid)), invB = 1.0 / (cnt * (fit.emin - fit.emax));
            const double invn = 1.0 / cnt;
            const lk_pt_rec* __restrict__ bp = pm.blocks[block].pts;
            // a split leaf (lk_ov_root_lane_kernel): its first n_base points are still the base map's
            const int n_base = job->n_base;
            const lk_pt_rec* __restrict__ bb = n_base > 0 ? base.blocks[job->base_block].pts : bp;
            double sa[3] = {0.0, 0.0, 0.0}, sb[3] = {0.0, 0.0, 0.0}, saa = 0.0, sab = 0.0, sbb = 0.0, sV[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int j = 0; j < cnt; ++j) {
                double pw[3], var[6];
                const lk_pt_rec* __restrict__ pj = (j < n_base ? bb : bp) + j;
#pragma unroll
                for (int c = 0; c < 3; ++c) pw[c] = pj->pw[c];
#pragma unroll
                for (int c = 0; c < 6; ++c) var[c] = pj->var[c];
                const double q0 = pw[0] - fit.c[0], q1 = pw[1] - fit.c[1], q2 = pw[2] - fit.c[2];
                const double dmin = q0 * fit.vmin[0] + q1 * fit.vmin[1] + q2 * fit.vmin[2];
                const double dmid = (q0 * fit.vmid[0] + q1 * fit.vmid[1] + q2 * fit.vmid[2]) * invA;
                const double dmax = (q0 * fit.vmax[0] + q1 * fit.vmax[1] + q2 * fit.vmax[2]) * invB;
                const double dmA = dmin * invA, dmB = dmin * invB;
                double FA[3], FB[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) FA[c] = dmid * fit.vmin[c] + dmA * fit.vmid[c], FB[c] = dmax * fit.vmin[c] + dmB * fit.vmax[c];
                const double a0 = var[0] * FA[0] + var[1] * FA[1] + var[2] * FA[2], a1 = var[1] * FA[0] + var[3] * FA[1] + var[4] * FA[2],
                             a2 = var[2] * FA[0] + var[4] * FA[1] + var[5] * FA[2];
                const double b0 = var[0] * FB[0] + var[1] * FB[1] + var[2] * FB[2], b1 = var[1] * FB[0] + var[3] * FB[1] + var[4] * FB[2],
                             b2 = var[2] * FB[0] + var[4] * FB[1] + var[5] * FB[2];
                saa += FA[0] * a0 + FA[1] * a1 + FA[2] * a2;
                sab += FA[0] * b0 + FA[1] * b1 + FA[2] * b2;
                sbb += FB[0] * b0 + FB[1] * b1 + FB[2] * b2;
                sa[0] += a0, sa[1] += a1, sa[2] += a2, sb[0] += b0, sb[1] += b1, sb[2] += b2;
#pragma unroll
                for (int c = 0; c < 6; ++c) sV[c] += var[c];
            }
            __builtin_amdgcn_sched_barrier(0);
            // the 21 unique entries, upper triangle row by row: rows 0..2 = [A var A^T | A var / n], rows 3..5 = [. | sum var / n^2]
            int kk = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int cc = r; cc < 3; ++cc)
                    acc21[kk++] = fit.vmid[r] * fit.vmid[cc] * saa + (fit.vmid[r] * fit.vmax[cc] + fit.vmax[r] * fit.vmid[cc]) * sab + fit.vmax[r] * fit.vmax[cc] * sbb;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) acc21[kk++] = (fit.vmid[r] * sa[cc] + fit.vmax[r] * sb[cc]) * invn;
            }
            const double invn2 = invn * invn;
            acc21[15] = sV[0] * invn2, acc21[16] = sV[1] * invn2, acc21[17] = sV[2] * invn2, acc21[18] = sV[3] * invn2, acc21[19] = sV[4] * invn2, acc21[20] = sV[5] * invn2;
        }
        plane_commit<true>(&pm.planes[root], &pm.match[root], fit, acc21, cnt);
    }
}
#endif

// Round 6: the same fit by a GROUP of eight lanes.  One lane per fit walks the leaf's <= 50 points one after the other - 34 of the 129 us of GPU
// time a bucket index of the recorded-run batch costs (1 024 slots x one or two fits: a latency chain), and in the uniform batch the lanes of a wave
// each read their own 72-B records (2.6 of 17.9 ms).  Here a wave first finds the real jobs among its 64 job slots (most are empty: a slot per touched
// root and inline group), then takes them eight at a time: lane `sub` of a group handles points sub, sub + 8, ... - eight consecutive 72-B records per
// step, 576 contiguous bytes -, the fifteen running sums are combined over the group with three exchange steps, and the group's first lane commits.
// Another order of the same sums than the one-lane loop (as that one already was against the 21-sum form): the same value to rounding.
#ifndef LK_FIT_GROUP
#define LK_FIT_GROUP 8
#endif
// (wave w_first of w_stride over the slot's jobs, `owner` = LK_WAVE * 5 ints of LDS that belong to this wave: the kernel below, and one phase of lk_ov_tail_kernel)
__device__ __forceinline__ void ov_fit_group_body(const LkMap& base, const LkOverlay& ov, const LkParams& pr, const unsigned int slot, int* owner, const int w_first,
                                                  const int w_stride, const int lane) {
    const LkMap pm = ov_slot_map(ov, slot);
    if (pm.counters[LK_CTR_ERR]) return;
    const int n_touched = (int)pm.counters[LK_CTR_TOUCHED];
    const LkFitJob* jobs = ov.jobs + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;   // entry [g][t]: inline leaf group g of touched root t
    const int4* jhdr = ov.jobhdr + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;
    const int sub = lane & (LK_FIT_GROUP - 1), grp = lane / LK_FIT_GROUP;
    const int total = n_touched * LK_INLINE_GROUPS;
    for (int i0 = w_first * LK_WAVE; i0 < total; i0 += w_stride * LK_WAVE) {   // wave-uniform
        const int i = i0 + lane;
        int4 hd = make_int4(0, 0, 0, 0);
        int2 bs = make_int2(0, 0);
        if (i < total) {
            const int g = i / n_touched, t = i - g * n_touched;
            const LkFitJob* job = &jobs[(size_t)g * ov.hash_cap + t];
            hd = jhdr[(size_t)g * ov.hash_cap + t];
            if (hd.z > 0 && hd.w != 0) bs = make_int2(job->base_block, job->n_base);
        }
        const bool has = hd.z > 0 && hd.w != 0;   // ("not a plane" events ended in lk_ov_fit_eig_kernel)
        const unsigned long long m = __ballot(has);
        const int njobs = __popcll(m);
        if (has) {   // the real jobs of this round, in slot order: {leaf, block, points, base block, points in it}
            const int rk = __popcll(m & ((1ull << lane) - 1ull));
            owner[rk * 5 + 0] = hd.x, owner[rk * 5 + 1] = hd.y, owner[rk * 5 + 2] = hd.z, owner[rk * 5 + 3] = bs.x, owner[rk * 5 + 4] = bs.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int r0 = 0; r0 < njobs; r0 += LK_WAVE / LK_FIT_GROUP) {   // wave-uniform
            const int k = r0 + grp;
            const bool act = k < njobs;
            const int* jr = &owner[(act ? k : 0) * 5];
            const int root = jr[0], block = jr[1], cnt = act ? jr[2] : 0, base_block = jr[3], n_base = jr[4];
            PlaneFit fit;
            {
                const lk_plane_rec* pl = &pm.planes[root];   // (an idle group reads its slot's first job's record: valid memory, never used)
#pragma unroll
                for (int c = 0; c < 3; ++c) fit.c[c] = pl->center[c], fit.vmin[c] = pl->normal[c], fit.vmid[c] = pl->plane_var[c], fit.vmax[c] = pl->plane_var[3 + c];
                fit.emin = pl->plane_var[6], fit.emid = pl->plane_var[7], fit.emax = pl->plane_var[8];
            }
            fit.is_plane = true;
            const int cn = cnt > 0 ? cnt : 1;
            const double invA = 1.0 / (cn * (fit.emin - fit.emid)), invB = 1.0 / (cn * (fit.emin - fit.emax));
            const double invn = 1.0 / cn;
            const lk_pt_rec* __restrict__ bp = pm.blocks[block].pts;
            const lk_pt_rec* __restrict__ bb = n_base > 0 ? base.blocks[base_block].pts : bp;
            double sm[15];   // sa (3), sb (3), saa, sab, sbb, sV (6)
#pragma unroll
            for (int q = 0; q < 15; ++q) sm[q] = 0.0;
#pragma unroll 1
            for (int j = sub; j < cnt; j += LK_FIT_GROUP) {
                double pw[3], var[6];
                const lk_pt_rec* __restrict__ pj = (j < n_base ? bb : bp) + j;
#pragma unroll
                for (int c = 0; c < 3; ++c) pw[c] = pj->pw[c];
#pragma unroll
                for (int c = 0; c < 6; ++c) var[c] = pj->var[c];
                const double q0 = pw[0] - fit.c[0], q1 = pw[1] - fit.c[1], q2 = pw[2] - fit.c[2];
                const double dmin = q0 * fit.vmin[0] + q1 * fit.vmin[1] + q2 * fit.vmin[2];
                const double dmid = (q0 * fit.vmid[0] + q1 * fit.vmid[1] + q2 * fit.vmid[2]) * invA;
                const double dmax = (q0 * fit.vmax[0] + q1 * fit.vmax[1] + q2 * fit.vmax[2]) * invB;
                const double dmA = dmin * invA, dmB = dmin * invB;
                double FA[3], FB[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) FA[c] = dmid * fit.vmin[c] + dmA * fit.vmid[c], FB[c] = dmax * fit.vmin[c] + dmB * fit.vmax[c];
                const double a0 = var[0] * FA[0] + var[1] * FA[1] + var[2] * FA[2], a1 = var[1] * FA[0] + var[3] * FA[1] + var[4] * FA[2],
                             a2 = var[2] * FA[0] + var[4] * FA[1] + var[5] * FA[2];
                const double b0 = var[0] * FB[0] + var[1] * FB[1] + var[2] * FB[2], b1 = var[1] * FB[0] + var[3] * FB[1] + var[4] * FB[2],
                             b2 = var[2] * FB[0] + var[4] * FB[1] + var[5] * FB[2];
                sm[0] += a0, sm[1] += a1, sm[2] += a2, sm[3] += b0, sm[4] += b1, sm[5] += b2;
                sm[6] += FA[0] * a0 + FA[1] * a1 + FA[2] * a2;
                sm[7] += FA[0] * b0 + FA[1] * b1 + FA[2] * b2;
                sm[8] += FB[0] * b0 + FB[1] * b1 + FB[2] * b2;
#pragma unroll
                for (int c = 0; c < 6; ++c) sm[9 + c] += var[c];
            }
            // the group's eight partial sums, in a fixed order (xor 1, 2, 4): every lane of the group ends with the total
#pragma unroll
            for (int q = 0; q < 15; ++q) {
#pragma unroll
                for (int w = 1; w < LK_FIT_GROUP; w <<= 1) sm[q] += __shfl_xor(sm[q], w, LK_WAVE);
            }
            if (act && sub == 0) {
                double acc21[21];
                int kk = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int cc = r; cc < 3; ++cc)
                        acc21[kk++] = fit.vmid[r] * fit.vmid[cc] * sm[6] + (fit.vmid[r] * fit.vmax[cc] + fit.vmax[r] * fit.vmid[cc]) * sm[7] + fit.vmax[r] * fit.vmax[cc] * sm[8];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) acc21[kk++] = (fit.vmid[r] * sm[cc] + fit.vmax[r] * sm[3 + cc]) * invn;
                }
                const double invn2 = invn * invn;
#pragma unroll
                for (int c = 0; c < 6; ++c) acc21[15 + c] = sm[9 + c] * invn2;
                plane_commit<true>(&pm.planes[root], &pm.match[root], fit, acc21, cnt);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // owner[] is rewritten by the next round
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_WAVE, LK_FIT_WAVES) lk_ov_fit_group_kernel(LkMap base, LkOverlay ov, LkParams pr) {
    __shared__ int owner[LK_WAVE * 5];
    ov_fit_group_body(base, ov, pr, blockIdx.y, owner, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}
#endif

// ---------------------------------------------------------------- the root pass of the batch replay, FAST PATH (round 5)
// Nearly every touched root of a batch bucket is a leaf that takes its few new points (in input order), maybe passes a refit event or
// two, maybe freezes.  The generic pass (dev_insert_root<.., OV>) spends its time - 168 VGPRs, three waves per SIMD, 5.5 GB per launch -
// on what that case does not need: every old point of the leaf re-read (9 doubles at a 72-B stride per lane: 36 cache lines per load
// instruction for 512 B) and, for a voxel seen for the first time, written back the same way; the per-lane walk, the grouping, the
// hand-over lists.  The fast path (lk_ov_point_geom_kernel + lk_ov_root_lane_kernel below) does the common case only and leaves
// everything else UNTOUCHED on a list for the generic pass (map.heavy / LK_CTR_HEAVY -> lk_ov_insert_root_kernel<.., CPLX>):
//   * the leaf's moment sums are kept per root (LkLeafSum: sums of its first n points, updated at every refit event; for a base voxel
//     computed once per replay, lk_ov_base_sums_kernel).  A refit event's plane test needs prefix + the points behind it - a handful,
//     usually only the new ones.  The old points are not read at all;
//   * a voxel seen for the first time ("thin") keeps its old points in the base map's block: the leaf becomes SPLIT (LK_PAD_SPLIT);
//   * the simulation of voxel_map.cc:186-204 is side-effect free, so a root that turns out to need the generic code (a cut:
//     init_octo_tree says "not a plane"; a plane that stops being one; more than a slot line of points; a tree below the root) is
//     handed over as it was found.
// The fit that ends a leaf's bucket is left to lk_ov_fit_eig_kernel / lk_ov_fit_lane_kernel (96-B job).
// (First version of the round, measured and replaced: the same fast path one WAVE per root - 128 VGPRs, four waves per SIMD, real
// one-root-ahead requests - 5.6 ms per batch against the generic pass's 10.8: bound by VALU issue, ~600 wave instructions per root for
// the ~8 lanes a root's points occupy.  profiles/EXPERIMENTS.md.)
__device__ __forceinline__ bool ov_plane_decide(const double* s, int count, float planer_threshold) {   // plane_test_regs<decide_only>: lambda_min < t
    const double n = (double)count;
    const double c0 = s[0] / n, c1 = s[1] / n, c2 = s[2] / n;
    const double t = (double)planer_threshold;
    const double b11 = (s[3] / n - c0 * c0) - t, bxy = s[4] / n - c0 * c1, bxz = s[5] / n - c0 * c2;
    const double b22 = (s[6] / n - c1 * c1) - t, byz = s[7] / n - c1 * c2, b33 = (s[8] / n - c2 * c2) - t;
    const double m2 = b11 * b22 - bxy * bxy;
    const double m3 = b11 * (b22 * b33 - byz * byz) - bxy * (bxy * b33 - byz * bxz) + bxz * (bxy * byz - b22 * bxz);
    return !(b11 > 0.0 && m2 > 0.0 && m3 > 0.0);
}
// Before anything reads a slot's private blocks as whole blocks (lk_overlay_export): the leaves the fast root pass left split get their base part.
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_MB) lk_ov_merge_split_kernel(LkMap base, LkOverlay ov, unsigned int slot) {
    const LkMap pm = ov_slot_map(ov, slot);
    const unsigned long long* keys = ov.keys + (size_t)slot * ov.hash_cap;
    const int lane = threadIdx.x & 63;
    const unsigned int wave = (blockIdx.x * LK_MB + threadIdx.x) >> 6, nwaves = (gridDim.x * LK_MB) >> 6;
    for (unsigned int r = wave; r < ov.hash_cap; r += nwaves) {
        if (keys[r] == LK_OV_EMPTY) continue;
        lk_node_rec* nd = &pm.nodes[r];
        const int nb = (int)nd->pad_[LK_PAD_SPLIT], cb = (int)nd->pad_[LK_PAD_COWBLK] - 1, blk = nd->block;
        if (nd->pad_[LK_PAD_LIVE] == 0) continue;
        const int bnode = (int)nd->pad_[LK_PAD_BASE] - 1;
        if (bnode >= 0 && pm.planes[r].points_size == LK_PLANE_LAZY) {   // never refitted in this replay: everything but d / radius / flags is the base record's
            if (lane < 16) {
                uint4 v = reinterpret_cast<const uint4*>(&base.planes[bnode])[lane];
                if (lane == 3) {
                    const uint4 mine = reinterpret_cast<const uint4*>(&pm.planes[r])[3];
                    v.x = mine.x, v.y = mine.y, v.z = mine.z;   // w = the base record's points_size
                }
                reinterpret_cast<uint4*>(&pm.planes[r])[lane] = v;
            }
        }
        if (nb <= 0) continue;
        if (blk >= 0 && cb >= 0) {
            const double* sp = reinterpret_cast<const double*>(base.blocks[cb].pts);
            double* dp = reinterpret_cast<double*>(pm.blocks[blk].pts);
            for (int j = lane; j < min(nb, nd->npts) * 9; j += LK_WAVE) dp[j] = sp[j];
        }
        if (lane == 0) nd->pad_[LK_PAD_SPLIT] = 0;
    }
}
#endif

// moment sums of the BASE map's root leaves that can still take points, once per replay: one thread per entry of the root table
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(256) lk_ov_base_sums_kernel(LkMap base, unsigned int n_hash, LkLeafSum* __restrict__ out) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_hash) return;
    const int4 e = base.hash[i];
    if (e.w < 0) return;
    const lk_node_rec* nd = &base.nodes[e.w];
    LkLeafSum r;
#pragma unroll
    for (int q = 0; q < 9; ++q) r.s9[q] = 0.0;
    r.n = 0, r.pad_ = 0;
    const int npts = nd->npts, block = nd->block;
    if (block >= 0 && npts > 0 && npts <= LK_BLOCK_PTS && !(nd->state & LK_NODE_PTS_DROPPED) && (nd->state & LK_NODE_UPDATE_ENABLE)) {
        const lk_pt_rec* p = base.blocks[block].pts;
        for (int j = 0; j < npts; ++j) {
            const double x = p[j].pw[0], y = p[j].pw[1], z = p[j].pw[2];
            r.s9[0] += x, r.s9[1] += y, r.s9[2] += z;
            r.s9[3] += x * x, r.s9[4] += x * y, r.s9[5] += x * z, r.s9[6] += y * y, r.s9[7] += y * z, r.s9[8] += z * z;
        }
        r.n = npts;
    }
    out[e.w] = r;
}
#endif


// ---------------------------------------------------------------- root pass of the batch replay, one LANE per point / per root (round 5)
// The wave-per-root fast pass above is bound by VALU ISSUE: ~600 wave instructions per root - a third of them the fp64 covariance of the
// new points - executed for the ~8 points a root queues, one eighth of the lanes.  This form splits the root pass by what it parallelises over:
//   lk_ov_point_geom_kernel  one THREAD per bucket point (every lane busy): the point's world position and covariance (KILO.cc:216-230,
//                            point_geom: the very expressions of every other pass), stored straight to its final place in the root's private
//                            block - n0 + its rank among the root's queued points (input order = ascending index; the slot line is read for
//                            the rank, its neighbours in the wave mostly read the same line).  Stores past the end of what the leaf finally
//                            takes (a freeze) or into a leaf the generic pass re-does are the same bytes written again, or beyond npts.
//   lk_ov_root_lane_kernel   one LANE per touched root: voxel_map.cc:186-204 literally, point after point, on the positions the first
//                            kernel has just stored (read back, 24 B per point; the old points are covered by the leaf's moment sums) -
//                            counters, refit events decided from prefix sums, freeze, the fit job, the sums.  Whatever is not a root leaf
//                            that appends / refits / freezes goes to the generic wave-per-root pass as it was found (map.heavy).
// Per root the second kernel issues ~40 small memory requests and a few hundred lane-instructions: 64 roots per wave instead of one.
__device__ __forceinline__ void ov_point_geom_body(const LkOverlay& ov, const LkParams& pr, const LkFilter* __restrict__ filters, const LkPtSrc& src, const unsigned int slot_, const int bx_, const int tid_) {
    const int i = bx_ * 256 + tid_;
    const unsigned int slot = slot_;
    const lk_point* pts;
    const int n = ov_pt_src(src, slot, &pts);
    if (i >= n) return;
    const int root = ov.ptroot[(size_t)slot * ov.scan_cap + i];
    if (root < 0) return;
    const LkMap pm = ov_slot_map(ov, slot);
    const lk_node_rec* nd = &pm.nodes[root];
    const int4 cnt = reinterpret_cast<const int4*>(nd)[4];   // npts, new_points, state, block
    const int m = (int)nd->pad_[0];
    if (cnt.w < 0 || m > LK_SLOTS) return;                    // no block: not a leaf that takes points; a long queue: the generic pass
    const float4* line = reinterpret_cast<const float4*>(pm.slots) + (size_t)root * LK_SLOTS;
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += (__float_as_int(line[j].w) < i) ? 1 : 0;
    const int pos = cnt.x + rank;
    if (pos >= LK_BLOCK_PTS) return;
    BucketConst bc;
    load_bucket_const(&filters[slot], pr, bc);
    const float4 p = reinterpret_cast<const float4*>(pts)[i];
    const PointGeom gm = point_geom(p.x, p.y, p.z, bc, pr);
    lk_pt_rec* d = &pm.blocks[cnt.w].pts[pos];
    d->pw[0] = gm.p_w.x, d->pw[1] = gm.p_w.y, d->pw[2] = gm.p_w.z;
    d->var[0] = gm.var.xx, d->var[1] = gm.var.xy, d->var[2] = gm.var.xz;
    d->var[3] = gm.var.yy, d->var[4] = gm.var.yz, d->var[5] = gm.var.zz;
}
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(256) lk_ov_point_geom_kernel(LkOverlay ov, LkParams pr, const LkFilter* __restrict__ filters, LkPtSrc src) {
    ov_point_geom_body(ov, pr, filters, src, blockIdx.y, (int)blockIdx.x, (int)threadIdx.x);
}
#endif
__device__ __forceinline__ void ov_root_lane_body(const LkMap& base, const LkOverlay& ov, const LkParams& pr, const unsigned int slot_, const int bx_, const int gx_, const int tid_) {
    const unsigned int slot = slot_;
    const LkMap map = ov_slot_map(ov, slot);
    if (map.counters[LK_CTR_ERR]) return;
    const int n_touched = (int)map.counters[LK_CTR_TOUCHED];
    LkFitJob* jobs = ov.jobs + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;
    int4* jhdr = ov.jobhdr + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS;
    const size_t job_stride = ov.hash_cap;
    LkLeafSum* sums = ov.sums + (size_t)slot * ov.hash_cap;
    const int thr = pr.layer_init_num[0];
    for (int t = bx_ * LK_WAVE + tid_; t < n_touched; t += gx_ * LK_WAVE) {
        const int root = map.touched[t];
        lk_node_rec* nd = &map.nodes[root];
        const int4 cnt = reinterpret_cast<const int4*>(nd)[4];   // npts, new_points, state, block
        const int4 qw = reinterpret_cast<const int4*>(nd)[6];    // pad_[0..3]: queued, ., ., LIVE
        const int4 cw = reinterpret_cast<const int4*>(nd)[7];    // pad_[4..7]: BASE, COWBLK, SPLIT, .
        const uint4 pl3 = reinterpret_cast<const uint4*>(&map.planes[root])[3];   // d, radius, flags, points_size
        const unsigned int rpf = pl3.z;
        const bool lazy_plane = (int)pl3.w == LK_PLANE_LAZY;   // plane AND match record are still the base map's (lk_ov_materialise_kernel<LEAN>)
        const int rlayer = nd->layer;
        const int n0 = cnt.x, rnewp = cnt.y, rblock = cnt.w, m = qw.x, ov_live = qw.w, cow_blk = cw.y - 1, split0 = cw.z;
        const unsigned int rst = (unsigned int)cnt.z;
        const bool thin = ov_live == 2 && cow_blk >= 0;
        const bool uninit = !(rst & LK_NODE_INIT_OCTO), lplane = (rpf & LK_PLANE_IS_PLANE) != 0, live = (rst & LK_NODE_UPDATE_ENABLE) != 0;
        bool complex_root = m > LK_SLOTS || m <= 0 || !(uninit || (lplane && live)) || (rst & LK_NODE_PTS_DROPPED) != 0 || n0 + 1 >= LK_BLOCK_PTS || rlayer != 0 ||
                            (ov_live == 2 && cow_blk < 0 && n0 > 0) || rblock < 0;
        const bool may_refit = uninit ? (n0 + m > thr) : (rnewp + m > 5);
        int cur = n0, newp = rnewp, fit_count = 0;
        bool frozen = false, fitted = false;
        double sq[9], sev[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) sq[q] = 0.0, sev[q] = 0.0;
        if (!complex_root) {
            const lk_pt_rec* bpts = map.blocks[rblock].pts;
            const int split_n = thin ? n0 : split0;   // points [0, split_n) of the leaf are the base block's
            const lk_pt_rec* cpts = (split_n > 0 && cow_blk >= 0) ? base.blocks[cow_blk].pts : bpts;
            auto add_point = [&](int j) {
                const lk_pt_rec* q = (j < split_n ? cpts : bpts) + j;
                const double x = q->pw[0], y = q->pw[1], z = q->pw[2];
                sq[0] += x, sq[1] += y, sq[2] += z;
                sq[3] += x * x, sq[4] += x * y, sq[5] += x * z, sq[6] += y * y, sq[7] += y * z, sq[8] += z * z;
            };
            if (may_refit) {   // the sums of the leaf's first sum_n points, then the old points behind them (normally none)
                const LkLeafSum* sr = cw.w == 1 ? &ov.base_sums[cw.x - 1] : &sums[root];   // LK_PAD_SUMSRC: 1 the base leaf's, 0 the root's own, 2 none
                int sum_n = cw.w == 2 ? 0 : sr->n;
                if (sum_n < 0 || sum_n > n0) sum_n = 0;
                if (sum_n > 0) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) sq[q] = sr->s9[q];
                }
                for (int j = sum_n; j < n0; ++j) add_point(j);
            }
            int mode = uninit ? 0 : 1;
            for (int j = 0; j < m && !frozen; ++j) {   // voxel_map.cc:186-204, one pushed point per step
                if (may_refit) add_point(n0 + j);
                const int m0 = mode;
                cur += 1, newp += 1;
                if (m0 == 0 ? cur > thr : newp > 5) {
                    if (!ov_plane_decide(sq, cur, pr.planer_threshold)) {
                        complex_root = true;   // a cut, or a plane that stops being one: the generic pass
                        break;
                    }
#pragma unroll
                    for (int q = 0; q < 9; ++q) sev[q] = sq[q];
                    fit_count = cur, fitted = true, newp = 0;
                    if (m0 == 0) {
                        mode = 1;
                        if (cur > pr.max_points_num) frozen = true;
                    }
                }
                if (m0 == 1 && cur >= pr.max_points_num) frozen = true;
            }
        }
        // the residual pass of the next bucket has to look at this scan's own voxel from now on: the generic pass may change its planes, a fit will
        if (complex_root || fitted) {
            int key[3];
            ov_unpack_key(ov.keys[(size_t)slot * ov.hash_cap + root], key);   // (the table entry IS the key; a copied node record's key words are whatever the base record held)
            unsigned int cell;
            if (ov_cell_of(base, key, &cell)) atomicOr(&ov.bits[(size_t)slot * ov.bit_words + (cell >> 5)], 1u << (cell & 31u));
        }
        if (complex_root) {   // record, queue and sums untouched: the generic pass finds the root as the earlier passes left it
            if (lazy_plane && cw.x > 0) {   // ... with the match record its residual-side twin reads made private first (lk_ov_materialise_kernel<LEAN> left it out)
                const uint4* sp = reinterpret_cast<const uint4*>(&base.match[cw.x - 1]);
                uint4* dp = reinterpret_cast<uint4*>(&map.match[root]);
#pragma unroll 1
                for (int c = 0; c < 9; ++c) dp[c] = sp[c];   // (a rare path: piece after piece, no registers held for it)
            }
            const unsigned int c = atomicAdd(&map.counters[LK_CTR_HEAVY], 1u);
            if (c < map.max_scan) map.heavy[2 * c] = root, map.heavy[2 * c + 1] = t;
            else atomicOr(&map.counters[LK_CTR_ERR], LK_E_SCRATCH_FULL);
            continue;
        }
        unsigned int nst = rst;
        int nnpts = cur, nblock = rblock;
        if (fitted) nst = (nst | LK_NODE_INIT_OCTO) & ~LK_NODE_OCTO_STATE;
        if (frozen) {   // node_freeze (voxel_map.cc:199-203); the block is not handed out again before the next bucket: the fit still reads it
            nst &= ~LK_NODE_UPDATE_ENABLE;
            nnpts = 0, newp = 0, nblock = -1;
            retire_block(map, rblock);
        }
        reinterpret_cast<int4*>(nd)[4] = make_int4(nnpts, newp, (int)nst, nblock);
        nd->list_head = -1;
        reinterpret_cast<int4*>(nd)[6] = make_int4(0, qw.y, qw.z, 1);          // queue consumed; complete
        if (thin) nd->pad_[LK_PAD_SPLIT] = (unsigned int)n0;                    // the n0 old points stay in the base block
#pragma unroll
        for (int g = 1; g < LK_INLINE_GROUPS; ++g) jhdr[(size_t)g * job_stride + t].z = 0;
        LkFitJob* jb = &jobs[t];
        if (fitted) {
            const int nb = thin ? n0 : split0;
            jhdr[t] = make_int4(root, rblock, fit_count, 1);
#pragma unroll
            for (int q = 0; q < 9; ++q) jb->s9[q] = sev[q];
            jb->base_block = nb > 0 ? cow_blk : -1, jb->n_base = nb > 0 ? nb : 0;
            LkLeafSum* sr = &sums[root];
#pragma unroll
            for (int q = 0; q < 9; ++q) sr->s9[q] = sev[q];
            sr->n = fit_count;
            if (cw.w != 0) nd->pad_[LK_PAD_SUMSRC] = 0;   // the root has its own sums record now
        } else {
            jhdr[t].z = 0;
        }
    }
}
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_WAVE, 4) lk_ov_root_lane_kernel(LkMap base, LkOverlay ov, LkParams pr) {
    ov_root_lane_body(base, ov, pr, blockIdx.y, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}
#endif
// Round 6, small buckets (the recorded-run batch: lk_batch_replay_overlay_ragged_dev with every bucket <= LK_SCAN_WAVE_MAX points): copy-on-write, point
// geometry and the lane-per-root pass of ONE slot as one workgroup of one launch - three launches of ~5 us each whatever they do, for a dozen points
// per slot.  What a phase writes the next one reads through the same CU's L1 (workgroup-scope fences; no L2 write-back).
template <bool LEAN>
__global__ void __launch_bounds__(LK_MB) lk_ov_mid_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* __restrict__ filters, LkPtSrc src) {
    const unsigned int slot = blockIdx.x;
    const int tid = (int)threadIdx.x, T = (int)blockDim.x;   // launched with 64 .. LK_MB threads (a multiple of 64 that divides 256)
    ov_materialise_body<LEAN>(base, ov, pr, slot, 0, 1, tid, T);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {
        const lk_point* pts;
        const int n = ov_pt_src(src, slot, &pts);
        for (int i0 = 0; i0 < n; i0 += T) ov_point_geom_body(ov, pr, filters, src, slot, (i0 + tid) >> 8, (i0 + tid) & 255);   // (the body's point index = 256 bx + tid)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (tid < LK_WAVE) ov_root_lane_body(base, ov, pr, slot, 0, 1, tid);
}

// ---------------------------------------------------------------- the ordered insert, slot = blockIdx.y
// W = waves per SIMD the register allocation aims at (2: the stream path's 216 VGPRs; 3: <= 168): this launch is a throughput pass over
// ~10^6 roots, each a chain of dependent round trips - concurrency, not the single wave's speed, sets its duration
template <int W, bool CPLX>
__global__ void __launch_bounds__(LK_MB, W)
    lk_ov_insert_root_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* filters, LkPtSrc src) {
    const LkMap pm = ov_slot_map(ov, blockIdx.y);
    if (pm.counters[LK_CTR_ERR]) return;   // this slot's pools overflowed: the call fails, nothing more is built on clamped ids
    const lk_point* pts;
    const int n = ov_pt_src(src, blockIdx.y, &pts);
    if (n == 0) return;
    dev_insert_root<false, true, CPLX>(pm, pr, filters + blockIdx.y, pts, (const lk_pt_rec*)nullptr, n,
                                 (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6), &base,
                                 ov.jobs + (size_t)blockIdx.y * ov.hash_cap * LK_INLINE_GROUPS, ov.hash_cap, nullptr,
                                 ov.jobhdr + (size_t)blockIdx.y * ov.hash_cap * LK_INLINE_GROUPS);
}
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_MB)
    lk_ov_insert_apply_kernel(LkOverlay ov, LkParams pr, const LkFilter* filters, LkPtSrc src) {
    const LkMap pm = ov_slot_map(ov, blockIdx.y);
    if (pm.counters[LK_CTR_ERR]) return;   // this slot's pools overflowed: the call fails, nothing more is built on clamped ids
    const lk_point* pts;
    const int n = ov_pt_src(src, blockIdx.y, &pts);
    if (n == 0) return;
    dev_insert_apply<false>(pm, pr, filters + blockIdx.y, pts, (const lk_pt_rec*)nullptr, n,
                            (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
}
#endif
// Round 6, small buckets (the recorded-run batch, every bucket <= LK_SCAN_WAVE_MAX points): what follows the fast root pass - the generic root pass over the
// roots it left (lk_ov_insert_root_kernel<.., CPLX>), both halves of the plane fits, the apply pass - of ONE slot as one workgroup of one launch.  As four
// launches each costs its slowest slot's chain plus a launch (7.5 + 9.0 + 9.2 + 5.4 us per bucket index, profiles/r06_ragged_overlay_pmc.json) whatever the
// slot at hand has to do - mostly nothing: a slot without work leaves here after its counter reads, and one with work runs its phases back to back.  The
// registers are the apply pass's (two waves per SIMD); what a phase writes the next one reads through the same CU's L1 (workgroup-scope fences).
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_MB, 2) lk_ov_tail_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* filters, LkPtSrc src) {
    __shared__ int owner[(LK_MB / LK_WAVE) * LK_WAVE * 5];
    const unsigned int slot = blockIdx.x;
    const int tid = (int)threadIdx.x, wave = tid >> 6, nw = (int)blockDim.x >> 6, lane = tid & 63;   // launched with 1 .. LK_MB / 64 waves
    const LkMap pm = ov_slot_map(ov, slot);
    if (pm.counters[LK_CTR_ERR]) return;   // this slot's pools overflowed: the call fails, nothing more is built on clamped ids
    const lk_point* pts;
    const int n = ov_pt_src(src, slot, &pts);
    if (n != 0)
        dev_insert_root<false, true, true>(pm, pr, filters + slot, pts, (const lk_pt_rec*)nullptr, n, wave, nw, &base,
                                           ov.jobs + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS, ov.hash_cap, nullptr,
                                           ov.jobhdr + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    ov_fit_eig_body(base, ov, pr, slot, tid, (int)blockDim.x);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    ov_fit_group_body(base, ov, pr, slot, owner + wave * (LK_WAVE * 5), wave, nw, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (n != 0 && !pm.counters[LK_CTR_ERR]) dev_insert_apply<false>(pm, pr, filters + slot, pts, (const lk_pt_rec*)nullptr, n, wave, nw);
}
#endif
// A FEW workgroups for the whole batch.  The code needs 254 + 126 registers and 6.3 KB of scratch per lane, and a launch with a workgroup per
// slot costs ~70 us whether or not a single item exists - on the bench's map none does in most buckets, and the recorded-run replay pays
// that once per bucket LEVEL (360 of them: half of its time).  Every workgroup reads ALL slots' counters (one contiguous array: S / 256
// coalesced loads per thread) and builds the list of the slots that have items - in slot order, so that it is the SAME list in every
// workgroup -; the workgroups then split that list by position, a slot's items are shared by the four waves of the workgroup that takes it.
#define LK_OV_FB_LIST 4096
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(LK_MB)
    lk_ov_insert_fallback_kernel(LkOverlay ov, LkParams pr, const LkFilter* filters, LkPtSrc src, const int n_slots) {
    __shared__ int active[LK_OV_FB_LIST];
    __shared__ unsigned int bits[LK_OV_FB_LIST / 32];
    __shared__ int n_active;
    const int lane = (int)(threadIdx.x & 63);
    for (int s0 = 0; s0 < n_slots; s0 += LK_OV_FB_LIST) {   // (one round unless the batch has more than 4 096 slots)
        __syncthreads();
        if (threadIdx.x < LK_OV_FB_LIST / 32) bits[threadIdx.x] = 0u;
        __syncthreads();
        const int s1 = min(n_slots, s0 + LK_OV_FB_LIST);
        for (int s = s0 + (int)threadIdx.x; s < s1; s += LK_MB) {
            const unsigned int* c = ov.counters + (size_t)s * LK_CTR_COUNT;
            // (a slot whose pools overflowed: the call fails, nothing more is built on clamped ids)
            if (c[LK_CTR_FALLBACK] != 0u && c[LK_CTR_ERR] == 0u) atomicOr(&bits[(s - s0) >> 5], 1u << ((s - s0) & 31));
        }
        __syncthreads();
        // the list in SLOT ORDER, the same in every workgroup (they split it among themselves by position): wave 0, two bitmap words per lane
        if (threadIdx.x < LK_WAVE) {
            unsigned int w0 = bits[2 * lane], w1 = bits[2 * lane + 1];
            const int cnt = __popc(w0) + __popc(w1);
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < LK_WAVE; o <<= 1) {
                const int v = __shfl_up(incl, o, LK_WAVE);
                if (lane >= o) incl += v;
            }
            int k = incl - cnt;
            while (w0) {
                active[k++] = s0 + 64 * lane + (__ffs((int)w0) - 1);
                w0 &= w0 - 1u;
            }
            while (w1) {
                active[k++] = s0 + 64 * lane + 32 + (__ffs((int)w1) - 1);
                w1 &= w1 - 1u;
            }
            if (lane == LK_WAVE - 1) n_active = incl;
        }
        __syncthreads();
        const int na = n_active;
        for (int a = (int)blockIdx.x; a < na; a += (int)gridDim.x) {
            const int slot = active[a];
            const LkMap pm = ov_slot_map(ov, (unsigned int)slot);
            const lk_point* pts;
            const int n = ov_pt_src(src, (unsigned int)slot, &pts);
            if (n == 0) continue;
            dev_insert_fallback<false>(pm, pr, filters + slot, pts, (const lk_pt_rec*)nullptr, n, (int)(threadIdx.x >> 6), LK_MB >> 6);
        }
    }
}
#endif

// ---------------------------------------------------------------- residual pass against base grid + overlay
// lk_residual_kernel's body with the overlay root lookup (residual_tile<..., GRID = 3>).
template <bool XID>
__global__ void LK_RES_BOUNDS
    lk_ov_residual_kernel(LkMap base, LkOverlay ov, LkParams pr, const LkFilter* __restrict__ filters, LkPtSrc src, double* __restrict__ partials,
                          size_t part_slot_stride) {
    __shared__ double stage[64 * LK_ROW2];
    const unsigned int slot = blockIdx.y;
    const int lane = threadIdx.x;
    const lk_point* pts;
    const int n = ov_pt_src(src, slot, &pts);
    if ((int)(blockIdx.x * LK_RB) >= n) return;   // (ragged batch: this scan's bucket is shorter than the launch's longest, or it has none)
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    LkOvView ovv;
    ovv.keys = ov.keys + (size_t)slot * ov.hash_cap;
    ovv.hash_mask = ov.hash_cap - 1;
    ovv.match = ov.match + (size_t)slot * ov.nodes_cap;
    ovv.nodes = ov.nodes + (size_t)slot * ov.nodes_cap;
    ovv.bits = ov.bits + (size_t)slot * ov.bit_words;
    ResidualOut out;
    out.h6 = nullptr, out.z = nullptr, out.R = nullptr, out.valid = nullptr, out.world = nullptr, out.ids = nullptr;
    const double acc = residual_tile<false, 3, XID, false, false>(base, pr, bc, reinterpret_cast<const float4*>(pts), blockIdx.x * LK_RB + lane, n, stage, lane, out,
                                                                  (size_t)0, &ovv);
    if (lane < LK_NPART) partials[(size_t)slot * part_slot_stride + (size_t)blockIdx.x * LK_NPART + lane] = (lane < 29) ? acc : 0.0;
}

// ---------------------------------------------------------------- status of all slots after a replay
// out[0] = OR of the slots' error words, out[1..3] = largest node / block / root count of any slot, out[4] = first slot with an error
#ifdef LK_TU_OVERLAY   // compiled in the overlay unit only (lk_internal.h)
__global__ void __launch_bounds__(256) lk_ov_status_kernel(LkOverlay ov, unsigned int n_slots, unsigned int* __restrict__ out) {
    for (unsigned int s = blockIdx.x * 256 + threadIdx.x; s < n_slots; s += gridDim.x * 256) {
        const unsigned int* c = ov.counters + (size_t)s * LK_CTR_COUNT;
        if (c[LK_CTR_ERR]) {
            atomicOr(&out[0], c[LK_CTR_ERR]);
            atomicMin(&out[4], s);
        }
        atomicMax(&out[1], c[LK_CTR_NODES] - ov.hash_cap + c[LK_CTR_ROOTS]);
        atomicMax(&out[2], c[LK_CTR_BLOCKS]);
        atomicMax(&out[3], c[LK_CTR_ROOTS]);
    }
}
#endif
