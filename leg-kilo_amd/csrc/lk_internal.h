// lk_internal.h - what the translation units of liblegkilo_hip.so share: the handle, the error / launch / allocation helpers, and the kernel
// headers.  Round 6: the library is five units compiled side by side - legkilo_hip.hip (LK_TU_MAIN: the C-ABI but for the overlay entries, and every
// kernel but the overlay's and the stream path's own), lk_stream.hip (LK_TU_STREAM: one live scan after the other with the map insert - the per-bucket
// launches, the scan-resident / grid-resident / pipelined kernels, and the KILO-path entries that run them), lk_overlay.hip (LK_TU_OVERLAY: batch replay
// WITH insert - lk_overlay_kernels.h's kernels and the entries that launch them), lk_ovscan.hip (LK_TU_OVSCAN: that replay's scan-resident kernel for small buckets), lk_prim.hip (rocPRIM).  A non-template kernel of a shared header is DEFINED in the main unit; the overlay unit sees its prototype
// (LK_KERNELS_ELSEWHERE) and launches it through the main unit's host stub.  The overlay header's own kernels are compiled in the overlay unit only.
#pragma once
#if !defined(LK_TU_MAIN) && !defined(LK_TU_OVERLAY) && !defined(LK_TU_STREAM) && !defined(LK_TU_OVSCAN)
#error "define LK_TU_MAIN, LK_TU_STREAM, LK_TU_OVERLAY or LK_TU_OVSCAN before including lk_internal.h"
#endif
#if defined(LK_TU_OVERLAY) || defined(LK_TU_STREAM) || defined(LK_TU_OVSCAN)
#define LK_KERNELS_ELSEWHERE 1
#endif
// legkilo_hip.hip — implementation of the C-ABI in include/legkilo_hip.h for gfx950.
// Host side of the shim: owns HBM pools, the HIP stream, and the launch sequences that
// replace KILO::predictUpdatePoint (KILO.cc:108-233) and the bucket loop (KILO.cc:367-396).
// There is no CPU fallback: without a gfx950 device lk_create fails with LK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "lk_prim.h"   // rocPRIM's sorts and scans, instantiated in lk_prim.hip

#include "lk_device.h"
#include "lk_filter_kernels.h"
#include "lk_point_kernels.h"
#include "lk_map_kernels.h"
#ifdef LK_TU_MAIN
#include "lk_pre_kernels.h"   // decode / voxel grid / ragged tables: launched by the main unit only
#include "lk_query_kernels.h" // per-point build_single_residual: launched by the main unit only
#endif
#include "lk_overlay_kernels.h"

static_assert(sizeof(lk_plane_rec) == 256, "plane record must be 256 B");
static_assert(sizeof(lk_node_rec) == 128, "node record must be 128 B");
static_assert(sizeof(lk_pt_rec) == 72, "point record must be 72 B");
static_assert(sizeof(lk_point) == 16, "scan point must be 16 B");

struct ProfEntry {
    uint64_t launches = 0;
    double total_ms = 0.0;
};

struct lk_handle {
    lk_config cfg = {};
    LkParams pr = {};
    LkMap map = {};
    hipStream_t stream = nullptr;
    static constexpr int kMaxGroups = 4;
    hipStream_t side[kMaxGroups - 1] = {};  // extra queues of the slot-group batch replay
    hipEvent_t ev_fork = nullptr, ev_join[kMaxGroups - 1] = {};
    int replay_groups = 3;
    size_t async_n = 0;           // n_scans of the asynchronous batches that may be in flight (0: none since the last lk_synchronize)
    bool wave_update = true;  // batch replay: single-wave update kernel (LEGKILO_UPDATE_CLASSIC=1 selects the 256-thread one)
    double last_slide_position[3] = {0.0, 0.0, 0.0};  // voxel_map.h:201
    unsigned int hash_cap = 0;
    LkFilter* d_filters = nullptr;
    double* d_Q = nullptr;
    double* d_partials = nullptr;
    size_t part_stride = 0;  // doubles per slot
    lk_point* d_scan = nullptr;
    float* d_world = nullptr;
    double* d_rows = nullptr;  // h6 (6n) | z (n) | R (n)
    unsigned char* d_valid = nullptr;
    double* d_tmp = nullptr;   // small scratch for class-surface calls (>= 18*32 doubles + 900*2)
    lk_pose* d_poses = nullptr;
    void* d_ragdev = nullptr;     // lk_batch_replay_scans_dev: flags, ranks, CSR tables, messages (grow-only)
    size_t ragdev_cap = 0;
    void* d_ragtmp = nullptr;     // rocPRIM scan scratch
    size_t ragtmp_cap = 0;
    void* d_rag = nullptr;        // tables of lk_batch_replay_ragged_dev (device copy, pinned staging copy)
    void* h_rag = nullptr;
    size_t rag_cap = 0;
    // pipelined stream path ("spec"): the insert of bucket k on its own stream beside predict + residual of bucket k+1 (enqueue_bucket)
    hipStream_t ins = nullptr;
    hipEvent_t ev_U[2] = {}, ev_D[2] = {}, ev_I = nullptr;
    unsigned int epoch = 16;      // bucket sequence number: stamps of LkMap::dirty / newroot, value of spec[LK_SPEC_DONE]
    unsigned int spec_base = 16;  // first epoch of the open window (stamps below it belong to inserts that were joined)
    bool spec_open = false;       // inserts may still be running on `ins`
    int gridscan_mode = 1;        // scans of large buckets as one grid-resident launch (lk_scan_grid_kernel): 0 never, 1 when every bucket holds
                                  // 513 .. LK_GRIDSCAN_AUTO_MAX points (where it measures faster than the launches), 2 whenever it applies; LEGKILO_GRIDSCAN / lk_stream_grid
    bool resident_enable = true;  // scans of small buckets as one resident launch (lk_scan_stream_kernel); LEGKILO_RESIDENT=0 / lk_stream_resident(h, 0): per-bucket launches
    bool spec_enable = false;     // LEGKILO_SPEC=1 / lk_stream_pipeline(h, 1); measured slower than the sequential order: off by default
    LkFilter* d_snap = nullptr;   // 2 posterior snapshots (dev_snapshot_posterior)
    struct ScanResult {            // what a stream-path scan hands back: written by ONE kernel into host-mapped pinned memory (no copies, one sync)
        lk_pose pose;
        unsigned int ctr[LK_CTR_COUNT];
        int resume[4];               // scan-resident kernel: LkResume's bf, bi, fb_bucket (where the launch stopped), 0
        unsigned int seq, pad_;      // written last: the host may poll it instead of blocking in hipStreamSynchronize
    };
    uint64_t resident_scans = 0, resident_relaunches = 0, grid_scans = 0, grid_relaunches = 0;   // lk_stream_resident_stats
    unsigned int test_stall_ms = 0;   // lk_test_stall: the next resident launches run with this bound and an injected stall (error-path test)
    unsigned int result_seq = 0;
    ScanResult* h_result = nullptr;   // hipHostMalloc(mapped)
    ScanResult* d_result = nullptr;   // its device-side address
    LkFilter* d_fbackup = nullptr;   // filters[0] as it was when the running scan started: what an LK_ERR_TIMEOUT puts back (grid-resident and pipelined paths)
    bool fbackup_valid = false;
    int2* d_ids = nullptr;        // [max_scan] root codes of the speculative residual pass
    uint64_t spec_buckets = 0, spec_tiles = 0, spec_redo_total = 0, res_redo_total = 0;
    unsigned int spec_redo_seen = 0, res_redo_seen = 0;
    double acc_norm = 1.0;
    bool q_diag = true;        // d_Q holds a diagonal matrix (zero-initialised; lk_set_Q re-checks)
    // frozen-map grid of batch replay (LkMap::grid): valid until the map changes
    LkMap fmap = {};           // h->map + the grid fields; h->map itself always has grid_on = 0 (the streaming path mutates the map)
    size_t grid_cap = 0;       // grid cells allocated behind the max_nodes match records of map.match
    bool grid_valid = false;   // the grid describes the current map
    uint64_t map_gen = 0;      // counts the map snapshots batch replays have frozen: frozen_map() bumps it whenever the map had changed since the last one
    uint64_t ov_gen = 0;       // the snapshot the last overlay replay ran against (lk_overlay_export reads base blocks / planes of THAT map)
    bool grid_enable = true;   // LEGKILO_GRID=0 keeps batch replay on the hash table (A/B)
    int* d_grid_mm = nullptr;
    // grow-only scratch of lk_preprocess_scan
    size_t pre_cap = 0, pre_tmp_bytes = 0;
    lk_point *pre_raw = nullptr, *pre_cells = nullptr, *pre_out = nullptr;
    unsigned int *pre_k0 = nullptr, *pre_k1 = nullptr, *pre_flags = nullptr, *pre_pos = nullptr, *pre_misc = nullptr;
    int *pre_v0 = nullptr, *pre_v1 = nullptr, *pre_starts = nullptr;
    void* pre_tmp = nullptr;
    // batch replay with a per-scan insert overlay (lk_overlay_kernels.h): the pools of all slots, grow-only
    LkOverlay ov = {};
    uint32_t ov_slots = 0;                                // slots the pools were allocated for
    uint32_t ov_want_roots = 0, ov_want_nodes = 0, ov_want_blocks = 0;   // lk_overlay_reserve (0: derived from the scan size)
    uint32_t ov_last_slots = 0;                           // slots of the last overlay replay (lk_overlay_export / lk_overlay_stats)
    uint32_t ov_hw_roots = 0, ov_hw_nodes = 0, ov_hw_blocks = 0;   // high-water marks of the last replay (any slot): the next replay's pools are sized from them
    size_t ov_hw_npts = 0;                                // ... which belong to scans of this size
    size_t ov_pool_bytes = 0;                             // bytes the overlay pools hold (lk_overlay_pool_bytes)
    LkFilter* d_ov_priors = nullptr;                      // the batch's priors, kept for the retry after a pool overflow
    size_t ov_priors_cap = 0;
    void* d_query = nullptr;                              // lk_match_points: inputs + outputs of a query, grown on demand
    size_t query_cap = 0;
    int* d_ov_res = nullptr;                              // scan-resident recorded-run replay with insert: [S] next bucket, [S] bucket with fallback items, stopped-scan counter
    size_t ov_res_cap = 0;
    unsigned int ov_res_rounds = 0;                       // launches of the scan-resident kernel in the last such replay
    unsigned int* d_ov_status = nullptr;
    // input order of device-resident batches (lk_batch_order): the batches the frozen-map batch entries have seen, with the library's voxel-ordered copy
    struct OrdEntry {
        const lk_point* src = nullptr;     // the caller's buffer and the shape it was seen with
        size_t n_scans = 0, n_pts = 0, n_buckets = 0;
        uint64_t off_hash = 0;
        bool as_given = false;             // the batch already was in voxel order: replayed where it lies, no copy, no stamp
        lk_point* copy = nullptr;          // voxel-ordered copy (every bucket of every scan sorted by root-voxel key under the priors of the first sight)
        size_t copy_bytes = 0;
        unsigned long long* d_ref = nullptr;   // device: [0] content stamp of src at the replay before, [1] 1 = this replay reads the copy
        bool have_copy = false;            // the copy holds the buffer's content as of the last sort (as far as the host knows: h_seen says what the device found)
        unsigned int* h_seen = nullptr;    // host-mapped, written by the device: replays in a row that found the same content stamp; LK_ORD_SORTED while the copy is current
        unsigned int* d_seen = nullptr;
        uint64_t tick = 0;
    };
    OrdEntry ord[2];
    uint64_t ord_tick = 0, ord_examined = 0, ord_sorted = 0, ord_stale = 0;
    int batch_order_mode = 1;              // LK_BATCH_ORDER_AUTO; LEGKILO_BATCH_ORDER=0 / lk_batch_order(h, 0): replay every batch as given
    int batch_order_after = 2;             // a batch is sorted once this many replays in a row have found the same content in its buffer (the sort pays for itself after ~10)
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, ProfEntry> prof;
    std::string err;
};

extern thread_local std::string g_err;   // defined in legkilo_hip.hip (lk_last_error(NULL) reads it)

static int fail(lk_handle* h, int code, const std::string& msg) {
    g_err = msg;
    if (h) h->err = msg;
    return code;
}
#define HIPCHK(h, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(h, LK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

template <typename F>
static int launch(lk_handle* h, const char* name, F&& f) {
    if (h->profiling) {
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        f();
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipEventSynchronize(h->ev1));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        ProfEntry& p = h->prof[name];
        p.launches += 1;
        p.total_ms += ms;
    } else {
        // LEGKILO_TRACE_LAUNCH=<prefix> (debug aid): every launch whose name starts with the prefix is announced on stderr and waited for -
        // the last name printed before a "Memory access fault by GPU" is the kernel that made it
        static const char* trace = getenv("LEGKILO_TRACE_LAUNCH");
        const bool tr = trace && strncmp(name, trace, strlen(trace)) == 0;
        if (tr) fprintf(stderr, "[launch] %s\n", name), fflush(stderr);
        f();
        if (tr) HIPCHK(h, hipDeviceSynchronize());
    }
    HIPCHK(h, hipGetLastError());
    return LK_OK;
}
#define LAUNCH(h, name, ...)                                   \
    do {                                                       \
        int rc_ = launch(h, name, [&]() { __VA_ARGS__; });     \
        if (rc_ != LK_OK) return rc_;                          \
    } while (0)

// Every device allocation of this library.  LEGKILO_POISON_POOLS=1 (test aid): the fresh memory is filled with 0x5a bytes instead of whatever the
// allocator hands out - in a young process zeros, in a long-lived one somebody's old data - so that a kernel which trusts memory nobody has
// written meets garbage in EVERY run (the whole GPU suite is run that way once per round: tools/gpu_poison_suite.sh)
static hipError_t lk_hip_malloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    static const bool poison = getenv("LEGKILO_POISON_POOLS") != nullptr;
    if (e == hipSuccess && (poison || getenv("LEGKILO_POISON_POOLS")) && bytes) {
        e = hipMemset(*p, 0x5a, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();   // (the fill runs on the null stream, the library's streams do not wait for that one)
    }
    return e;
}
template <typename T>
static hipError_t lk_hip_malloc(T** p, size_t bytes) {
    return lk_hip_malloc(reinterpret_cast<void**>(p), bytes);
}
#define hipMalloc(p, n) lk_hip_malloc((p), (n))

// device temporaries of one call: freed on every return path
struct DevTemps {
    std::vector<void*> ptrs;
    ~DevTemps() {
        for (void* p : ptrs)
            if (p) hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T*)p;
        return e;
    }
};

static unsigned int next_pow2(unsigned int v) {
    unsigned int p = 1;
    while (p < v) p <<= 1;
    return p;
}

static int check_map_errors(lk_handle* h, const unsigned int* fetched = nullptr) {
    unsigned int ctr[LK_CTR_COUNT];
    if (fetched) {
        memcpy(ctr, fetched, sizeof(ctr));
    } else {
        HIPCHK(h, hipMemcpyAsync(ctr, h->map.counters, sizeof(ctr), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (ctr[LK_CTR_ERR] & LK_E_SPEC_TIMEOUT) {
        // not sticky: the word is cleared, so the handle stays usable once its map has been restored
        const unsigned int rest = ctr[LK_CTR_ERR] & ~LK_E_SPEC_TIMEOUT;
        HIPCHK(h, hipMemcpyAsync(h->map.counters + LK_CTR_ERR, &rest, sizeof(rest), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // the filter keeps its PRE-SCAN state on every path: the resident kernels' and the pipelined launches' scans start with a copy of
        // filters[0] that is put back here (a scan-resident launch given up returns before its write-back, but a scan picked up again after
        // fallback items has written the filter once)
        if (h->fbackup_valid) {
            HIPCHK(h, hipMemcpyAsync(h->d_filters, h->d_fbackup, sizeof(LkFilter), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        h->fbackup_valid = false;
        return fail(h, LK_ERR_TIMEOUT, "a bounded device-side wait of the stream path timed out (device fault, or a pre-empted / debugged GPU; "
                                       "LEGKILO_RESIDENT_TIMEOUT_MS raises the bound): the filter keeps its state from before the scan, the map may hold a partial "
                                       "insert - restore it (lk_map_import) and replay the scan");
    }
    h->fbackup_valid = false;
    if (ctr[LK_CTR_ERR]) {
        char buf[160];
        snprintf(buf, sizeof(buf), "device pool overflow (bits 0x%x: 1 hash, 2 nodes, 4 point blocks, 8 scratch, 16 bad blob)", ctr[LK_CTR_ERR]);
        return fail(h, LK_ERR_CAPACITY, buf);
    }
    return LK_OK;
}


#define LK_SCAN_WAVE_MAX 512   // largest bucket of the one-wave-per-scan chains (dev_scan_wave, lk_rag_ov_front_kernel): up to eight tiles, added in tile order
extern "C" int spec_join(lk_handle* h);   // main unit: joins the pipelined stream path's insert stream
#define CHECK_H(h)                                                         \
    do {                                                                   \
        if (!(h)) return fail(nullptr, LK_ERR_INVALID, "null handle");     \
        hipSetDevice((h)->cfg.device_id);                                  \
        if ((h)->spec_open) {                                              \
            int rcj_ = spec_join(h);                                       \
            if (rcj_ != LK_OK) return rcj_;                                \
        }                                                                  \
    } while (0)
#define CHECK_SLOT(h, s)                                                                  \
    do {                                                                                  \
        if ((s) >= (h)->cfg.n_slots) return fail(h, LK_ERR_INVALID, "slot out of range"); \
    } while (0)

#define LK_CTR_GRID_XCC 15   // LkMap.counters[15]: XCC ids (one bit each) the working blocks of the last one-XCD launch of the grid-resident stream kernel ran on
static inline void imu_noise(const lk_config& c, double* Rn) {   // diag of R of an IMU observation (KILO.cc:251-253)
    Rn[0] = Rn[1] = c.imu_acc_meas_noise;
    Rn[2] = c.imu_acc_z_meas_noise;
    Rn[3] = Rn[4] = Rn[5] = c.imu_gyr_meas_noise;
}

// ---- shared between the translation units (all with C linkage: they are defined inside the units' extern "C" regions)
extern "C" {
// main unit (legkilo_hip.hip)
int frozen_map(lk_handle* h, LkMap* out);                    // the handle's map + the frozen-map grid of the batch replays (rebuilt when the map has changed)
int join_side_streams(lk_handle* h);
int zero_scan_counters(lk_handle* h, uint32_t first_slot, uint32_t n_slots);
int fetch_poses(lk_handle* h, lk_pose* out, int n);
int export_map_blob(lk_handle* h, const LkMap& m, unsigned int hash_cap, void* blob, size_t* bytes);
int ragged_replay(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off, const uint32_t* n_buckets, const uint32_t* bucket_off,
                  const double* bucket_dt, const double* t_begin, const uint32_t* n_imu, const void* imus, size_t msg_bytes, lk_pose* out, bool with_insert = false);
__global__ void lk_set_times_kernel(LkFilter* filters, int n, double t);
__global__ void __launch_bounds__(LK_WAVE, 2) lk_rag_advance_kernel(LkFilter* filters, const double* __restrict__ Q, LkRagged rg, int b, int msg_kind);
int upload_xyz_as_points(lk_handle* h, const float* xyz, size_t n);   // n x 3 floats -> h->d_scan as lk_point records
// stream unit (lk_stream.hip)
int run_scan(lk_handle* h, const lk_point* pts, const lk_point* d_pts, size_t n, double t_begin, const lk_imu* imus, size_t n_imu, const lk_kin_imu* kins,
             size_t n_kin, float* xyz_world_out, lk_pose* out);          // the bucket loop of KILO::process on a sorted cloud that is in HBM (and on the host, for the bucket bounds)
// overlay unit (lk_overlay.hip)
void ov_free(lk_handle* h);
// lk_ovscan.hip (LK_TU_OVSCAN: the scan-resident kernel of the recorded-run replay with insert, a unit of its own for the build time): one launch of it
int ov_scan_launch(lk_handle* h, bool xid, int S, hipStream_t st, const LkMap& fmap, const LkOverlay& ov, LkFilter* fl, const LkRagged& rg, const lk_point* d_pts, int msg_kind,
                   int* cur, int* fb_b, unsigned int* pending);
int overlay_ragged_launch(lk_handle* h, const lk_point* d_pts, size_t S, const LkRagged& rg, const double* d_tbegin, int biggest, size_t ldb, const int* max_n,
                          size_t max_scan_pts, int msg_kind, lk_pose* out);
}
