/*
 * legkilo_hip.h — C-ABI of liblegkilo_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of Leg-KILO: the per-time-bucket LiDAR ESKF
 * update, i.e. KILO::predictUpdatePoint (legkilo/src/core/slam/KILO.cc:108-233)
 * and what it calls in eskf.cc / voxel_map.cc, plus the bucket loop around it
 * (KILO.cc:367-396).  The reference has no FFI of its own; the boundary is the
 * C++ class surface KILO consumes (eskf.h:46-109, voxel_map.h:180-244).  The
 * C++ mirror of that surface lives in leg-kilo_amd/host/ and calls only the
 * functions declared here.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative lk_status; nothing
 *     throws across the ABI; lk_last_error() gives the text of the last error.
 *   - plain pointers and sizes only.  Arrays are caller-owned.  Matrices are
 *     ROW-MAJOR fp64 unless a comment says otherwise.  Points are f32.
 *   - a handle = one device + one HIP stream + one voxel map + n_slots filter
 *     slots (slot 0 is "the" filter of the reference; slots >0 exist for batch
 *     replay).  A handle is NOT thread-safe (the reference path is single
 *     threaded: leg_kilo_node.cc:35-38).
 *   - calls are synchronous on return unless the name ends in _async.
 *   - pointers named d_* are DEVICE pointers (HBM-resident); all others host.
 *
 * State vector layout (eskf.h:15-32), LK_STATE_DOUBLES = 36:
 *   x[0..8]  rot_ (3x3 row-major)   x[9..11]  pos_     x[12..14] vel_
 *   x[15..17] ba_   x[18..20] bw_   x[21..23] grav_    x[24..26] imu_a_
 *   x[27..29] imu_w_   x[30..32] bv_   x[33..35] contact_
 * Error-state order (30): rot0 pos3 vel6 ba9 bw12 grav15 imu_a18 imu_w21 bv24 contact27.
 */
#ifndef LEGKILO_HIP_H_
#define LEGKILO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LK_DIM_STATE 30
#define LK_STATE_DOUBLES 36
#define LK_ABI_VERSION 1

typedef enum lk_status {
    LK_OK = 0,
    LK_ERR_INVALID = -1,   /* bad argument */
    LK_ERR_HIP = -2,       /* HIP runtime error (text in lk_last_error) */
    LK_ERR_CAPACITY = -3,  /* a device pool (hash / nodes / point blocks / scan) overflowed */
    LK_ERR_NO_DEVICE = -4, /* no gfx950 device visible; there is NO CPU fallback */
    LK_ERR_STATE = -5,     /* call order violated (e.g. update before set_state) */
    LK_ERR_TIMEOUT = -6    /* a bounded device-side wait of the stream path was given up (fault / pre-empted GPU): the filter keeps its
                              pre-scan state, the map may hold a partial insert - restore it and replay the scan.  Not sticky. */
} lk_status;

/* ESKF::Config (eskf.h:49-65) + VoxelMapConfig (voxel_map.h:41-57) + extrinsics
 * (KILO.cc:74-79) + device capacities.  YAML key names are those of
 * legkilo/config/leg_fusion.yaml. */
typedef struct lk_config {
    /* ESKF::Config, declaration order */
    double vel_process_cov;
    double imu_acc_process_cov;
    double imu_gyr_process_cov;
    double contact_process_cov;
    double acc_bias_process_cov;
    double gyr_bias_process_cov;
    double kin_bias_process_cov;
    double imu_acc_meas_noise;
    double imu_acc_z_meas_noise;
    double imu_gyr_meas_noise;
    double kin_meas_noise;
    double chd_meas_noise;          /* loaded, never used by the reference */
    double contact_meas_noise;      /* loaded, never used by the reference */
    double lidar_point_meas_ratio;
    /* VoxelMapConfig */
    double max_voxel_size;          /* yaml voxel_size */
    double planner_threshold;       /* yaml min_eigen_value */
    double beam_err;                /* degrees */
    double dept_err;                /* metres */
    double sigma_num;
    int32_t max_layer;
    int32_t max_iterations;         /* unused by the reference (voxel_map.h:44) */
    int32_t layer_init_num[5];
    int32_t max_points_num;
    /* extrinsics: p_imu = ext_R * p_lidar + ext_T (KILO.cc:128) */
    double ext_R[9];
    double ext_T[3];
    double gravity;                 /* KILO.cc:50 */
    /* device side */
    int32_t device_id;              /* HIP device ordinal */
    uint32_t n_slots;               /* filter slots (>=1); >1 only for batch replay */
    uint32_t max_roots;             /* root voxels the hash must hold (table = next pow2 of 2x) */
    uint32_t max_nodes;             /* octree nodes (roots + children) */
    uint32_t max_point_blocks;      /* per-leaf point blocks of LK_BLOCK_PTS points */
    uint32_t max_scan_points;       /* points of the largest scan / bucket batch */
} lk_config;

/* PointType fields the path reads (pcl_types.h:11; x,y,z @0 and curvature @36 of
 * the 48-byte pcl::PointXYZINormal): body-frame xyz + per-point time offset. */
typedef struct lk_point {
    float x, y, z;
    float curvature;
} lk_point;

/* sensor_msgs::Imu fields predictUpdateImu reads (KILO.cc:235-258) */
typedef struct lk_imu {
    double stamp;
    double acc[3];
    double gyr[3];
} lk_imu;

/* common::KinImuMeas (sensor_types.hpp:19-27); contact as int32 (bool in the reference) */
typedef struct lk_kin_imu {
    double time_stamp;
    double foot_pos[4][3];
    double foot_vel[4][3];
    int32_t contact[4];
    double acc[3];
    double gyr[3];
} lk_kin_imu;

/* result of one scan / one replay unit */
typedef struct lk_pose {
    double rot[9];   /* row-major */
    double pos[3];
    double vel[3];
    uint64_t n_effect;   /* success_pts_size_out (KILO.cc:180) */
    uint32_t n_buckets;
    uint32_t n_updates;  /* buckets whose N>0 */
} lk_pose;

/* ---- map blob (lk_map_export / lk_map_import; also the RCCL broadcast payload) ----
 * blob = lk_blob_header | roots[n_roots] | nodes[n_nodes] | planes[n_nodes] | blocks[n_blocks]
 * All little-endian PODs below; node ids index nodes[]/planes[]; block ids index blocks[]. */
#define LK_BLOB_MAGIC 0x4C4B4D50u /* 'LKMP' */
#define LK_BLOCK_PTS 52           /* >= max_points_num + 1 (voxel_map.cc:199,232) */

typedef struct lk_blob_header {
    uint32_t magic, version;
    uint32_t n_roots, n_nodes, n_blocks, block_pts;
    double voxel_size;
    int32_t max_layer, max_points_num;
    uint64_t bytes;              /* total blob size */
} lk_blob_header;

typedef struct lk_root_rec {     /* one hash entry: Vec3i key -> root node (voxel_map.h:186) */
    int32_t key[3];
    int32_t node;
} lk_root_rec;

/* VoxelPlane (voxel_map.h:96-119), 256 B, the record the residual kernel streams */
#define LK_PLANE_IS_PLANE 1u
#define LK_PLANE_IS_INIT 2u
typedef struct lk_plane_rec {
    double center[3];
    double normal[3];
    float d;
    float radius;
    uint32_t flags;
    int32_t points_size;
    double plane_var[21];        /* upper triangle of the 6x6, row-major: (0,0)(0,1)..(0,5)(1,1).. */
    float min_eigen_value, mid_eigen_value, max_eigen_value;
    uint32_t pad_[3];
} lk_plane_rec;

/* VoxelOctoTree (voxel_map.h:129-176) minus the plane, 128 B */
#define LK_NODE_INIT_OCTO 1u
#define LK_NODE_UPDATE_ENABLE 2u
#define LK_NODE_OCTO_STATE 4u
#define LK_NODE_PTS_DROPPED 8u   /* >LK_BLOCK_PTS first-frame points: only the count is kept */
typedef struct lk_node_rec {
    int32_t child[8];            /* -1 = nullptr */
    double voxel_center[3];
    float quater_length;
    int32_t layer;
    int32_t npts;                /* temp_points_.size() */
    int32_t new_points;
    uint32_t state;
    int32_t block;               /* point block id, -1 = none */
    int32_t key[3];              /* root key (roots only) */
    int32_t list_head;           /* device scratch, always -1 at rest */
    uint32_t pad_[8];
} lk_node_rec;

typedef struct lk_pt_rec {       /* pointWithVar fields the map keeps: point_w + var (sym) */
    double pw[3];
    double var[6];               /* xx xy xz yy yz zz */
} lk_pt_rec;

typedef struct lk_block_rec {
    lk_pt_rec pts[LK_BLOCK_PTS];
} lk_block_rec;

typedef struct lk_handle lk_handle;

/* ---- lifetime ---- */
int lk_abi_version(void);
int lk_create(const lk_config* cfg, lk_handle** out);                  /* ESKF(const Config&) + VoxelMapManager(VoxelMapConfig&) */
void lk_destroy(lk_handle* h);
const char* lk_last_error(const lk_handle* h);                         /* h may be NULL */

/* ---- ESKF surface (eskf.h:46-109); slot = filter slot, 0 for the reference's single filter ---- */
int lk_set_state(lk_handle* h, uint32_t slot, const double* x36, const double* P900);  /* setState + cov() */
int lk_get_state(lk_handle* h, uint32_t slot, double* x36, double* P900);              /* state(), cov() */
int lk_set_Q(lk_handle* h, const double* Q900);                        /* setQ */
int lk_get_Q(lk_handle* h, double* Q900);                              /* Q() */
int lk_init_process_cov_q(lk_handle* h);                               /* initProcessCovQ, eskf.cc:47-62 */
int lk_set_times(lk_handle* h, uint32_t slot, double last_predict_t, double last_update_t); /* KILO.cc:350-351 */
int lk_get_times(lk_handle* h, uint32_t slot, double* last_predict_t, double* last_update_t);
int lk_set_acc_norm(lk_handle* h, double acc_norm);                    /* KILO.cc:349 */
int lk_get_acc_norm(lk_handle* h, double* acc_norm);                   /* acc_norm_ (KILO.h:60): part of a checkpoint */
int lk_get_fx(lk_handle* h, uint32_t slot, double dt, double* Fx900);   /* getFx, eskf.cc:72-81 */
int lk_get_function_f(lk_handle* h, uint32_t slot, double dt, double* f30); /* getFunctionf, eskf.cc:64-70 */
int lk_predict(lk_handle* h, uint32_t slot, double dt, int prop_state, int prop_cov);   /* predict, eskf.cc:83-89 */
/* updateByPoints(ObsShared&), eskf.cc:91-113; h6 is N x 6 ROW-major */
int lk_update_by_points(lk_handle* h, uint32_t slot, const double* h6, const double* z, const double* R, size_t N);
/* updateByImu / updateByKinImu (eskf.cc:125-145): ki_h is M x 30 row-major */
int lk_update_by_imu(lk_handle* h, uint32_t slot, const double* ki_z6, const double* ki_R6);
int lk_update_by_kin_imu(lk_handle* h, uint32_t slot, const double* ki_h, const double* ki_z, const double* ki_R, size_t M);

/* ---- local map sliding: VoxelMapManager::mapSliding voxel_map.cc:552-569, clearMemOutOfMap voxel_map.cc:571-594
 * (configured at KILO.cc:68-70: sliding_thresh, half_map_size, map_sliding_en - the caller's gate; the reference itself
 * never calls them).  `position` is what the reference keeps in the public member position_last_ (voxel_map.h:199);
 * the handle keeps last_slide_position (initially the origin, voxel_map.h:201).  A slide deletes every root voxel
 * whose key is strictly outside [k - half_map_size, k + half_map_size]^3, k = floor(position / max_voxel_size), and
 * compacts the node / point-block pools so that device memory is bounded by the live map.
 * *slid = the reference's return value; *n_removed = deleted root voxels (either may be NULL). */
int lk_map_slide(lk_handle* h, const double* position /*3*/, double sliding_thresh, int32_t half_map_size, int32_t* slid,
                 uint32_t* n_removed);
int lk_map_clear_outside(lk_handle* h, int32_t x_max, int32_t x_min, int32_t y_max, int32_t y_min, int32_t z_max, int32_t z_min,
                         uint32_t* n_removed);
/* read (set = 0) or write (set != 0) last_slide_position - part of a checkpoint */
int lk_map_slide_position(lk_handle* h, int32_t set, double* last3);

/* ---- VoxelMapManager surface (voxel_map.h:180-244) ---- */
/* BuildVoxelMap(rot, rot_cov, pos_cov) with feats_down_world_/feats_down_body_ = the two clouds
 * (xyz f32, n x 3); rot/rot_cov/pos_cov are taken from slot 0 (KILO.cc:339). */
int lk_map_build(lk_handle* h, const float* xyz_world, const float* xyz_body, size_t n);
/* UpdateVoxelMap(const std::vector<pointWithVar>&): pw n x 3, var n x 9 (row-major 3x3) fp64 */
int lk_map_update(lk_handle* h, const double* pw, const double* var9, size_t n);
/* Residual build of KILO.cc:122-210 with the CURRENT state of slot 0, no predict, no update, no
 * insert (config 2).  Outputs per input point: h6 n x 6 row-major, z, R, valid (0/1). */
int lk_residuals(lk_handle* h, const float* xyz_body, size_t n, double* h6, double* z, double* R, uint8_t* valid);
/* VoxelMapManager::build_single_residual(pv, root voxel at key, layer 0, is_success = false, prob = 0, single_ptpl)
 * (voxel_map.cc:363-427, started the way KILO.cc:149-155 starts it) for n caller-held pointWithVar: keys3 n x 3 voxel keys,
 * pw n x 3 (pv.point_w), var9 n x 9 (pv.var, row-major).  Outputs per point: found (a root voxel exists at the key), success
 * (is_success), prob, and of the winning plane normal n x 3, center n x 3, d, dis_to_plane (signed, float as voxel_map.h:92 stores
 * it) and layer (-1 and zeros when no plane was taken).  The map is not modified. */
int lk_match_points(lk_handle* h, size_t n, const int32_t* keys3, const double* pw, const double* var9, uint8_t* found,
                    uint8_t* success, double* prob, double* normal3, double* center3, double* d, float* dis_to_plane,
                    int32_t* layer);
int lk_map_stats(lk_handle* h, uint32_t* n_roots, uint32_t* n_nodes, uint32_t* n_blocks);
int lk_map_export(lk_handle* h, void* blob, size_t* bytes);            /* blob==NULL: size query */
int lk_map_import(lk_handle* h, const void* blob, size_t bytes);
/* device-resident blob for the RCCL broadcast (no host staging) */
int lk_map_export_dev(lk_handle* h, void* d_blob, size_t* bytes);
int lk_map_import_dev(lk_handle* h, const void* d_blob, size_t bytes);

/* ---- KILO path (KILO.cc) ---- */
/* predictUpdatePoint (KILO.cc:108-233) on slot 0: xyz_body n x 3 f32 (one time bucket),
 * xyz_world_out n x 3 f32 (cloud_down_world), intensity_out n f32 (0 / 255), *n_effect += N. */
int lk_update_points(lk_handle* h, double t, const float* xyz_body, size_t n, float* xyz_world_out,
                     float* intensity_out, size_t* n_effect);
int lk_update_imu(lk_handle* h, const lk_imu* imu);                    /* predictUpdateImu, KILO.cc:235-258 */
int lk_update_kin_imu(lk_handle* h, const lk_kin_imu* kin);            /* predictUpdateKinImu, KILO.cc:260-314 */
/* bucket loop of KILO::process (KILO.cc:367-396) on a time-SORTED scan: buckets are runs of
 * exactly equal curvature; IMU (imu_mode_only) or kin+IMU messages with stamp < bucket time are
 * consumed first.  n_imu>0 and n_kin>0 together is invalid.  world_out may be NULL. */
int lk_process_scan(lk_handle* h, const lk_point* sorted_pts, size_t n, double t_begin, const lk_imu* imus,
                    size_t n_imu, const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out);
/* same with the scan already resident in HBM (d_pts: n x lk_point) and bucket bounds given by the
 * caller: bucket b covers [bucket_off[b], bucket_off[b+1]) at time t_begin + bucket_dt[b]. */
int lk_process_scan_dev(lk_handle* h, const lk_point* d_pts, size_t n, double t_begin, const uint32_t* bucket_off,
                        const double* bucket_dt, size_t n_buckets, lk_pose* out);

/* ---- sensor decode: LidarProcessing::{velodyne,ouster,hesai}Handler, lidar_processing.cc:25-108 ----
 * msg_data = sensor_msgs::PointCloud2::data (n_points x point_step bytes).  Keeps every filter_num-th point that is
 * outside the blind radius (lidar_processing.h:96-98), curvature = round((t - t_first) * 500) / 500 in the arithmetic
 * type of the respective handler, input order preserved.  begin/end = LidarScan::lidar_begin_time_/lidar_end_time_. */
typedef struct lk_cloud_layout {
    uint32_t point_step;
    uint32_t off_x, off_y, off_z;   /* float32 fields */
    uint32_t off_time;              /* Velodyne `time` f32 | Ouster `t` u32 | Hesai `timestamp` f64 */
    int32_t lidar_type;             /* 1 Velodyne, 2 Ouster, 3 Hesai (sensor_types.hpp:34) */
} lk_cloud_layout;
int lk_decode_scan(lk_handle* h, const void* msg_data, size_t n_points, const lk_cloud_layout* layout, double time_scale,
                   int filter_num, float blind, double header_stamp, lk_point* out, size_t* n_out, double* begin_time,
                   double* end_time);
int lk_decode_scan_dev(lk_handle* h, const void* d_msg_data, size_t n_points, const lk_cloud_layout* layout, double time_scale,
                       int filter_num, float blind, double header_stamp, lk_point* d_out, size_t* n_out, double* begin_time,
                       double* end_time);

/* ---- the two steps in front of the path, device-resident ----
 * pcl::VoxelGrid centroid filter as used at KILO.cc:356-360 (leaf = yaml voxel_grid_resolution; centroid of x, y, z
 * AND curvature) followed by the time sort of KILO.cc:369-370 (stable).  Cells are emitted in ascending cell index,
 * points of a cell are summed in input order in float32 (PCL leaves both undefined; oracle/preprocess_oracle.py).
 * out_sorted must hold n_raw points; *n_out receives the number of cells. */
int lk_preprocess_scan(lk_handle* h, const lk_point* raw, size_t n_raw, float leaf, lk_point* out_sorted, size_t* n_out);
int lk_preprocess_scan_dev(lk_handle* h, const lk_point* d_raw, size_t n_raw, float leaf, lk_point* d_out_sorted, size_t* n_out);
/* raw scan -> pose without the cloud leaving HBM: lk_preprocess_scan_dev + the bucket loop of lk_process_scan
 * (only the n_out x 16 B sorted cloud is read back, for the bucket bounds and the IMU interleave). */
int lk_process_raw_scan(lk_handle* h, const lk_point* raw, size_t n_raw, float leaf, double t_begin, const lk_imu* imus,
                        size_t n_imu, const lk_kin_imu* kins, size_t n_kin, size_t* n_down, lk_pose* out);

/* ---- batch replay (config 5): scans are independent units against the handle's FROZEN map ----
 * scan s uses filter slot s (n_scans <= n_slots); all scans have n_pts points laid out
 * d_pts[s * n_pts + i]; bucket bounds are shared by all scans.  Inserts are disabled. */
int lk_batch_set_priors(lk_handle* h, const double* x36, const double* P900, size_t n_scans); /* per-scan priors */
/* same, priors already resident in HBM (d_x36: n_scans x 36, d_P900: n_scans x 900); asynchronous on the handle's
 * stream - a replay loop re-arms its batch without touching the host */
int lk_batch_set_priors_dev(lk_handle* h, const double* d_x36, const double* d_P900, size_t n_scans);
/* Puts every time bucket of every scan of a device-resident batch (layout of lk_batch_replay_dev) into root-voxel order under the slots' PRIOR
 * poses (lk_batch_set_priors(_dev) first): the 64 points of a residual wave then look at a handful of voxels instead of sixty.  Points keep
 * their bucket; the order inside a bucket is one of the legal outcomes of the reference's sort of equal time stamps (KILO.cc:369), so the
 * replay of d_out is a replay of the same scans.  Once per loaded batch.  d_in and d_out must not overlap; needs 16 B of scratch per point. */
int lk_batch_sort_by_voxel_dev(lk_handle* h, const lk_point* d_in, lk_point* d_out, size_t n_scans, size_t n_pts, const uint32_t* bucket_off,
                               size_t n_buckets);
/* Input order of device-resident batches.  The order of the points INSIDE a time bucket is left open by the reference (KILO.cc:369 sorts by time with an
 * unstable sort; a bucket is a run of equal time stamps), but it decides how fast the batch residual kernel runs: in root-voxel order a wave's 64
 * points look at a handful of voxels, in a random order at sixty (1.4 x slower).  LK_BATCH_ORDER_AUTO (default): lk_batch_replay_dev and
 * lk_batch_replay_async_dev keep track of the batches they are given - (device pointer, shape, bucket bounds) and a stamp of 4 096 sampled points,
 * taken on the device ahead of every replay, no host round trip.  A batch that has come back UNCHANGED twice is, at its third replay, sorted ONCE into
 * a library-owned copy unless every bucket already is in voxel order under the slots' priors (the work of lk_batch_sort_by_voxel_dev: 5.6 ms and 16 B per
 * point for 1 024 x 100 000 points; that call synchronises), and the later replays of that batch read the copy: the same scans in another legal order,
 * equal to the replay of the buffer as given up to the order of floating-point sums.  The caller's buffer is never written.  A batch replayed once or
 * twice (new scans streamed through a staging buffer) is never sorted.  New content in a sorted batch's buffer is noticed by the same stamp: that replay
 * reads the buffer as given and the count starts again; after an in-place edit too small for the sample, call lk_batch_changed.  The two most recently
 * used batches are kept.
 * LK_BATCH_ORDER_AS_GIVEN: every batch is replayed where it lies, no copy, no stamp (also: LEGKILO_BATCH_ORDER=0).
 * The entries WITH insert (lk_batch_replay_overlay*_dev) and the ragged entries always replay as given: with the map insert the order inside a bucket
 * decides which points a voxel holds when it is fitted, and the caller's order is the one its checker knows. */
#define LK_BATCH_ORDER_AS_GIVEN 0
#define LK_BATCH_ORDER_AUTO 1
int lk_batch_order(lk_handle* h, int mode);
int lk_batch_changed(lk_handle* h);
/* For a caller who KNOWS a batch will be replayed many times: examine it and make the voxel-ordered copy now, under the priors of slots [0, n_scans)
 * (lk_batch_set_priors(_dev) first), instead of at its third replay.  Synchronous; once per loaded batch (works in either mode). */
int lk_batch_prepare_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, const uint32_t* bucket_off, size_t n_buckets);
/* out3 = { batches examined (replayed unchanged often enough), of those sorted into a copy, sorted batches whose buffer later held other content } since lk_create */
int lk_batch_order_stats(lk_handle* h, uint64_t* out3);
int lk_batch_replay_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, double t_begin,
                        const uint32_t* bucket_off, const double* bucket_dt, size_t n_buckets, lk_pose* out);
/* Config 2 ("voxel kNN + point-to-plane residuals only") for a device-resident batch: the residual build of KILO.cc:122-210 for
 * n_scans x n_pts points in one launch, scan s (layout of lk_batch_replay_dev) under the CURRENT state of filter slot s (lk_batch_set_priors(_dev);
 * no predict, no update, no insert), with the rows MATERIALISED in HBM: d_rows8 [n_scans * n_pts][8] = one 64-byte record per point holding what
 * lk_residuals returns in three arrays - h (1 x 6, KILO.cc:195-197), z (KILO.cc:199), R (KILO.cc:208-209) - and d_valid (0 / 1, the point
 * matched a plane).  Records of unmatched points are zero.  16 B read + 65 B written per point.  d_rows8 must be 16-byte aligned.
 * Asynchronous on the handle's stream (lk_synchronize). */
int lk_batch_residuals_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, double* d_rows8, uint8_t* d_valid);
/* Bulk read-out of the filters a batch replay left in slots [first_slot, first_slot + n): state (n x 36: rot 9, pos, vel, ba, bw,
 * grav, imu_a, imu_w, bv, contact) and covariance (n x 900, row-major) - what ESKF::state() / cov() and getRotCov / getPosCov /
 * getVelCov (blocks (0,0), (3,3), (6,6) of P, eskf.h:46-109) give per filter, for all of them with ONE gather kernel instead of one
 * lk_get_state round trip per slot.  Either pointer may be NULL.  _dev: device pointers, asynchronous on the handle's stream (after
 * everything the batch entries enqueued); the host variant synchronises.  Per scan this is SURVEY 8(e)'s result record: pose, P, and
 * the counters of lk_pose. */
int lk_batch_get_states_dev(lk_handle* h, uint32_t first_slot, size_t n, double* d_x36, double* d_P900);
int lk_batch_get_states(lk_handle* h, uint32_t first_slot, size_t n, double* x36, double* P900);
/* Asynchronous, double-buffered variant: the batch uses filter slots [first_slot, first_slot + n_scans); batches whose
 * slot ranges rotate (first_slot = 0, n_scans, [2 n_scans,] 0, ...) run on up to three streams, so the latency-bound update / predict
 * kernels of one batch overlap the residual launches of the others.  d_x36 / d_P900 (device, n_scans x 36 / 900; both NULL =
 * keep the slots' state) arm the priors on the batch's own stream; the poses are copied into host_out (n_scans records;
 * PINNED host memory for a truly asynchronous copy; may be NULL).  Nothing synchronises: lk_synchronize() completes all
 * enqueued batches.  Buffers must stay valid / untouched until then.  first_slot must be a multiple of n_scans (LK_ERR_INVALID
 * otherwise): the stream a batch runs on is a function of its slot range, so equal-sized ranges never overlap partially; a call
 * with a different n_scans than the batches still in flight first drains them. */
int lk_batch_replay_async_dev(lk_handle* h, const lk_point* d_pts, uint32_t first_slot, size_t n_scans, size_t n_pts, double t_begin,
                              const uint32_t* bucket_off, const double* bucket_dt, size_t n_buckets, const double* d_x36,
                              const double* d_P900, lk_pose* host_out);

/* ---- batch replay WITH the map insert (SURVEY.md 8d, config 5 "scan-local insert overlay") ----
 * What KILO::process does per scan - predict, residual, update AND re-projection + UpdateVoxelMap after every bucket (KILO.cc:108-233,
 * :375-395; voxel_map.cc:336-361), so that buckets 2..n of a scan are matched against the planes their own scan has refitted, cut or
 * created - for n_scans independent scans at once.  Every scan (filter slot s) starts from the handle's map and inserts into its own
 * copy-on-write overlay; the handle's map is not changed, and the overlays are discarded by the next replay.  Same argument meaning
 * as lk_batch_replay_dev (priors from lk_batch_set_priors(_dev); synchronous; out may be NULL).  Per slot the result equals
 * lk_process_scan_dev on a handle that holds a private copy of the map.  LK_ERR_CAPACITY when a scan's overlay outgrows its pools
 * (the message names the slot and the sizes in use). */
int lk_batch_replay_overlay_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, double t_begin,
                                const uint32_t* bucket_off, const double* bucket_dt, size_t n_buckets, lk_pose* out);
/* The same for a RECORDED run's scans (config 1 / config 4 shape): every scan its own size, its own time buckets (runs of equal curvature,
 * KILO.cc:375-378), its own start time and - msg_kind 1: lk_imu, 2: lk_kin_imu, 0: none - its own messages, applied between the buckets
 * as the loop at KILO.cc:379-390 does (n_msg[s] records of scan s, concatenated in msgs, time-sorted per scan).  Table arguments as for
 * lk_batch_replay_ragged_dev.  Bucket index after bucket index over all scans: messages + predict, residual against base map + the scan's
 * overlay, update, re-projection + UpdateVoxelMap into the overlay (KILO.cc:108-233) - what KILO::process computes for each scan alone on a
 * private copy of the map.  Any bucket size; synchronous; out may be NULL. */
int lk_batch_replay_overlay_ragged_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off, const uint32_t* n_buckets,
                                       const uint32_t* bucket_off, const double* bucket_dt, const double* t_begin, const uint32_t* n_msg,
                                       const void* msgs, int msg_kind, lk_pose* out);
/* Per-scan overlay capacities: root voxels a scan's inserts may touch or create, octree nodes and live point blocks of those voxels.
 * 0 (default) = sized by the library: a first guess from the scan size (n_pts / 18 roots), afterwards the previous replay's high-water
 * marks + 25 %, and a scan that outgrows such pools makes them grow and the batch run again (no error).  Capacities set HERE are the
 * caller's word: overflowing them fails the replay with LK_ERR_CAPACITY.  Releases pools of another shape. */
int lk_overlay_reserve(lk_handle* h, uint32_t roots_per_scan, uint32_t nodes_per_scan, uint32_t blocks_per_scan);
/* The voxels scan `slot` of the LAST overlay replay holds privately - every root voxel its inserts touched or created, whole octrees -
 * as a map blob (lk_map_export's format; blob == NULL: size query).  Voxels not in it are the handle's, unchanged.
 * An overlay refers to the handle's map as it was at the replay (old points and unchanged planes of a copied voxel stay there): export
 * BEFORE the map is changed.  After lk_process_scan / lk_update_points / lk_map_update / lk_map_build / lk_map_import / lk_map_slide /
 * lk_map_clear_outside the call fails with LK_ERR_STATE. */
int lk_overlay_export(lk_handle* h, uint32_t slot, void* blob, size_t* bytes);
/* Largest private root / node / point-block count any scan of the last overlay replay reached (any pointer may be NULL). */
int lk_overlay_stats(lk_handle* h, uint32_t* max_roots, uint32_t* max_nodes, uint32_t* max_blocks);
/* What the overlay pools hold right now: total bytes in HBM and the per-scan capacities (root-table entries = root node records,
 * child nodes, point blocks); all 0 before the first overlay replay / after lk_overlay_reserve released them. */
int lk_overlay_pool_bytes(lk_handle* h, uint64_t* bytes, uint32_t* root_entries, uint32_t* child_nodes, uint32_t* blocks);
/* lk_batch_replay_overlay_ragged_dev runs a batch whose buckets all hold <= 512 points (a recorded scan's 2 ms bins, KILO.cc:375-378)
 * SCAN-RESIDENT: one launch carries every scan through its whole bucket chain incl. the insert; a scan whose bucket needs the
 * insert's fallback code stops behind it, the fallback launch runs, and the scans are launched again.  *rounds = launches of the
 * scan kernel in the handle's last such replay (1 + the largest number of stops of any scan), 0 when that replay ran launch by
 * launch (a bucket over 512 points, or LEGKILO_RAG_RESIDENT=0).  Same results either way, bit for bit. */
int lk_overlay_resident_rounds(lk_handle* h, uint32_t* rounds);

/* Ragged batch: the scans of a recorded run differ in size, in their time buckets (KILO.cc:375-378) and in their start
 * time.  Scan s = d_pts[scan_off[s] .. scan_off[s+1]) (scan_off: n_scans + 1 entries) on filter slot s; n_buckets[s] buckets
 * whose bounds (n_buckets[s] + 1 offsets relative to the scan's first point, first 0, last = points in the scan) and time
 * offsets (n_buckets[s] values, added to t_begin[s]) are the next rows of bucket_off / bucket_dt (rows concatenated in
 * scan order).  Empty buckets are skipped.  Frozen map, priors from lk_batch_set_priors(_dev); synchronous; out may be
 * NULL.  What a bucket loop of KILO::process (KILO.cc:375-395) over each scan would give, for all scans at once. */
int lk_batch_replay_ragged_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                               const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                               const double* t_begin, lk_pose* out);

/* The same with the IMU messages of each scan (only_imu_use mode): n_imu[s] messages of scan s, rows concatenated in `imus`,
 * time-sorted per scan; every message stamped before a bucket's time is applied (predictUpdateImu, KILO.cc:235-258) before
 * that bucket, as the loop at KILO.cc:379-383 does.  Requires every bucket to hold <= 512 points (what a recorded scan's
 * 2 ms bins hold); LK_ERR_INVALID otherwise. */
int lk_batch_replay_ragged_imu_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                                   const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                                   const double* t_begin, const uint32_t* n_imu, const lk_imu* imus, lk_pose* out);

/* Leg-fusion mode (only_imu_use: false): the same with each scan's kinematic + IMU messages (common::KinImuMeas): every message
 * stamped before a bucket's time is applied (predictUpdateKinImu, KILO.cc:260-314; updateByKinImu, eskf.cc:137-145) before
 * that bucket, as the loop at KILO.cc:384-390 does.  Same <= 512 points-per-bucket requirement. */
int lk_batch_replay_ragged_kin_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                                   const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                                   const double* t_begin, const uint32_t* n_kin, const lk_kin_imu* kins, lk_pose* out);

/* Recorded-run replay without host-side bucket tables: scan s = d_pts[scan_off[s] .. scan_off[s+1]) (time-sorted, scan_off: n_scans + 1
 * entries, first 0), start time t_begin[s]; the buckets - runs of exactly equal curvature, KILO.cc:375-378 - are found on the
 * device.  msg_kind 0: no messages; 1: n_msg[s] lk_imu records per scan (only_imu_use, KILO.cc:379-383); 2: lk_kin_imu records
 * (leg fusion, KILO.cc:384-390), concatenated in `msgs`.  Equivalent to lk_batch_replay_ragged(_imu/_kin)_dev on tables built
 * from the same scans.  A scan whose curvature decreases somewhere (the sort of KILO.cc:367-370 was skipped) or is NaN is
 * refused with LK_ERR_INVALID before any filter slot is touched. */
int lk_batch_replay_scans_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off, const double* t_begin,
                              int msg_kind, const uint32_t* n_msg, const void* msgs, lk_pose* out);

/* ---- measurement hooks ---- */
int lk_profile_enable(lk_handle* h, int on);                           /* HIP-event timing around each kernel */
int lk_profile_get(lk_handle* h, const char* kernel, uint64_t* launches, double* total_ms);
int lk_profile_reset(lk_handle* h);
int lk_device_malloc(lk_handle* h, void** d_ptr, size_t bytes);
int lk_device_free(lk_handle* h, void* d_ptr);
int lk_memcpy_h2d(lk_handle* h, void* d_dst, const void* src, size_t bytes);
int lk_memcpy_d2h(lk_handle* h, void* dst, const void* d_src, size_t bytes);
int lk_synchronize(lk_handle* h);
/* Pipelined stream path ("insert off the critical chain"): with the pipeline on (on = 1 or LEGKILO_SPEC=1; the
 * default is the sequential order on one stream, which measures faster) the map insert of bucket k (KILO.cc:216-233) runs on a second HIP stream beside
 * the predict + residual pass of bucket k+1, and a verify pass re-evaluates the tiles whose points looked at a root voxel the
 * insert stamped.  Results are those of the sequential order.  lk_stream_stats: out4 = { buckets that went through the
 * HIP-stream pipeline, their residual tiles, tiles its verify pass evaluated again, buckets the scan-resident kernel (below, its own
 * mechanism: LDS flags inside one workgroup) evaluated again after a conflicting insert } since lk_create. */
int lk_stream_pipeline(lk_handle* h, int on);
/* Scan-resident stream kernel: a scan whose time buckets all hold <= 512 points (the reference's own scan shape: 2 ms bins of a dozen
 * points) runs its whole bucket loop - messages, predict, residual, update AND map insert (KILO.cc:375-395, :216-233) - as ONE launch
 * of one resident workgroup instead of several launches per bucket (on by default; on = 0 or LEGKILO_RESIDENT=0: per-bucket
 * launches).  Same device functions, identical results. */
int lk_stream_resident(lk_handle* h, int on);
/* out2 = { scans that went through the scan-resident kernel, launches it needed beyond one per scan } since lk_create.  The generic
 * fallback items of the insert (a voxel that is cut - init_octo_tree / cut_octo_tree, voxel_map.cc:119-183 -, leftovers after a flip to a
 * tree, a root with more than 64 queued points) are not part of the resident kernel: a bucket that produces one ends the launch, the
 * items run as a launch of their own and the resident kernel is launched again from where it stopped (same results; ~30 us per event). */
int lk_stream_resident_stats(lk_handle* h, uint64_t* out2);
/* TEST HOOK (fault injection for the LK_ERR_TIMEOUT path; no product use): bound_ms > 0 - the resident stream kernels' next launches run with
 * this bound on their device-side waits AND one role stops answering at the scan's fourth bucket, so the call fails with LK_ERR_TIMEOUT and
 * must leave the filter at its pre-scan state; 0 - back to normal. */
int lk_test_stall(lk_handle* h, unsigned int bound_ms);
/* Grid-resident stream kernel: a scan whose time buckets all hold > 512 points (and no IMU / kinematic messages between them) runs
 * its whole bucket loop as ONE launch of co-resident workgroups - the phases of the per-bucket launches separated by grid barriers
 * (agent-scope release / acquire hand-offs) instead of kernel boundaries, the rarely needed phases entered only when the device
 * counters ask for them.  Same device functions, identical results.  mode 1 (default): scans whose buckets hold at most 4 096 points
 * (51 two-ms bins of a 100 000-point scan: 2.53 -> 2.34 ms); 2: any size (5 x 20 000: slower than the launches); 0: never.
 * LEGKILO_GRIDSCAN sets the initial mode. */
int lk_stream_grid(lk_handle* h, int mode);
/* Where the last grid-resident launch of up to 32 workgroups ran: one bit per XCC id its working blocks reported (HW_REG_XCC_ID).  Such a
 * launch starts 8 x G blocks of which every eighth works - blocks are observed (not promised) to go to XCD b % 8, so the working ones
 * share one XCD and its L2, and their barriers then need no L2 write-back.  The kernel does not rely on it: the barriers drop the
 * write-back only when this mask, collected on the device behind a full barrier, has exactly one bit.  LEGKILO_GRIDSCAN_XCD=0: G blocks, any XCD. */
int lk_stream_grid_placement(lk_handle* h, uint32_t* xcc_mask);
int lk_stream_stats(lk_handle* h, uint64_t* out4);
void* lk_stream(lk_handle* h);                                         /* the handle's hipStream_t */

#ifdef __cplusplus
}
#endif
#endif /* LEGKILO_HIP_H_ */
