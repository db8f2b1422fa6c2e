// legkilo_rccl.hpp — the multi-GPU side of the C++ host: what leg-kilo_amd/replay.py does over torch.distributed, for a C++ caller
// that owns its RCCL communicator (one rank per GPU: separate processes, or one process driving all GPUs of a node).
//
//   broadcastMap      the voxel map of rank `src` to every rank, HBM to HBM: lk_map_export_dev packs the blob on the device,
//                     RCCL ships it over xGMI, lk_map_import_dev unpacks it - the once-per-snapshot collective of batch replay
//                     (SURVEY 8e).  `ring`: one ncclBroadcast (a ring: bound by ONE of the root's seven xGMI links);
//                     `scatter_allgather`: the root sends a distinct 1/W slice to every peer (ncclSend / ncclRecv in one group: all
//                     its links carry different data), then ncclAllGather of the slices.
//   shardRange        contiguous block partition of the recorded run's scans (replay.shard_range).
//   allGatherPoses    every rank's lk_pose records of its shard, in rank order, on every rank (device buffers).
//   allGatherStates   the same for the result record with covariance (lk_batch_get_states_dev: state 36 + P 900 per scan).
// All calls are enqueued on the handle's own stream (lk_stream), i.e. ordered with the lk_* calls around them; nothing here
// synchronises except where a size has to reach the host.  Header-only over include/legkilo_hip.h + <rccl/rccl.h>.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <stdexcept>
#include <string>

#include "legkilo_hip.h"

namespace legkilo {

inline void rcclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
inline void hipCheckRt(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void lkCheck(lk_handle* h, int rc, const char* what) {
    if (rc != LK_OK) throw std::runtime_error(std::string(what) + ": " + lk_last_error(h));
}

// Device memory of the HANDLE's device (lk_device_malloc selects it: a rank that has made no lk_* call yet may have another device
// current), released on every path out of a scope - an exception between two collectives must not leak the staging buffers.
class DeviceBuf {
  public:
    DeviceBuf(lk_handle* h, size_t bytes) : h_(h) { lkCheck(h, lk_device_malloc(h, &p_, bytes ? bytes : 16), "lk_device_malloc"); }
    ~DeviceBuf() {
        if (p_) lk_device_free(h_, p_);
    }
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    template <typename T>
    T* as() const { return static_cast<T*>(p_); }

  private:
    lk_handle* h_;
    void* p_ = nullptr;
};

inline void shardRange(size_t n_total, int rank, int world, size_t* start, size_t* stop) {
    const size_t base = n_total / (size_t)world, rem = n_total % (size_t)world;
    *start = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
    *stop = *start + base + ((size_t)rank < rem ? 1 : 0);
}

enum class MapTransport { ring, scatter_allgather };

// Returns the blob size in bytes.  Collective: every rank of `comm` calls it with its own handle.  An exception thrown here (a HIP, RCCL
// or lk_* failure on THIS rank) releases this rank's buffers, but the peers may be left inside a collective: treat it as fatal for the
// communicator (ncclCommAbort) - there is no partial map on any rank, lk_map_import_dev replaces a map only after validating the blob.
inline size_t broadcastMap(lk_handle* h, ncclComm_t comm, int rank, int world, int src, MapTransport how = MapTransport::scatter_allgather) {
    hipStream_t st = static_cast<hipStream_t>(lk_stream(h));
    unsigned long long nbytes = 0;
    if (rank == src) {
        size_t b = 0;
        lkCheck(h, lk_map_export_dev(h, nullptr, &b), "lk_map_export_dev (size)");
        nbytes = b;
    }
    {
        DeviceBuf d_n(h, sizeof(nbytes));
        hipCheckRt(hipMemcpyAsync(d_n.as<void>(), &nbytes, sizeof(nbytes), hipMemcpyHostToDevice, st), "hipMemcpyAsync");
        rcclCheck(ncclBroadcast(d_n.as<void>(), d_n.as<void>(), 1, ncclUint64, src, comm, st), "ncclBroadcast (size)");
        hipCheckRt(hipMemcpyAsync(&nbytes, d_n.as<void>(), sizeof(nbytes), hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
        hipCheckRt(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
    if (nbytes == 0) return 0;   // (cannot happen with lk_map_export_dev - an empty map still has a header and its hash table - but a size of 0 must not reach the import)
    const size_t per = (nbytes + (size_t)world - 1) / (size_t)world;   // slice of the scatter + all-gather form
    DeviceBuf blob(h, per * (size_t)world + 16);
    if (rank == src) {
        size_t b = nbytes;
        lkCheck(h, lk_map_export_dev(h, blob.as<void>(), &b), "lk_map_export_dev");
    }
    if (how == MapTransport::ring || world == 1) {
        rcclCheck(ncclBroadcast(blob.as<void>(), blob.as<void>(), nbytes, ncclUint8, src, comm, st), "ncclBroadcast (blob)");
    } else {
        DeviceBuf mine(h, per);
        rcclCheck(ncclGroupStart(), "ncclGroupStart");
        if (rank == src)
            for (int r = 0; r < world; ++r) rcclCheck(ncclSend(blob.as<unsigned char>() + (size_t)r * per, per, ncclUint8, r, comm, st), "ncclSend");
        rcclCheck(ncclRecv(mine.as<void>(), per, ncclUint8, src, comm, st), "ncclRecv");
        rcclCheck(ncclGroupEnd(), "ncclGroupEnd");
        rcclCheck(ncclAllGather(mine.as<void>(), blob.as<void>(), per, ncclUint8, comm, st), "ncclAllGather (blob)");
        hipCheckRt(hipStreamSynchronize(st), "hipStreamSynchronize");   // `mine` is released at the end of this scope
    }
    if (rank != src) lkCheck(h, lk_map_import_dev(h, blob.as<void>(), nbytes), "lk_map_import_dev");
    hipCheckRt(hipStreamSynchronize(st), "hipStreamSynchronize");
    return (size_t)nbytes;
}

// d_local: n_local lk_pose records on the device; d_all: world * n_local records.  Every rank passes the same n_local.
inline void allGatherPoses(lk_handle* h, ncclComm_t comm, const lk_pose* d_local, lk_pose* d_all, size_t n_local) {
    rcclCheck(ncclAllGather(d_local, d_all, n_local * sizeof(lk_pose), ncclUint8, comm, static_cast<hipStream_t>(lk_stream(h))), "ncclAllGather (poses)");
}
// The per-scan result record with covariance: states [world * n_local][36] and covariances [world * n_local][900].
inline void allGatherStates(lk_handle* h, ncclComm_t comm, uint32_t first_slot, size_t n_local, double* d_x_local, double* d_P_local,
                            double* d_x_all, double* d_P_all) {
    hipStream_t st = static_cast<hipStream_t>(lk_stream(h));
    lkCheck(h, lk_batch_get_states_dev(h, first_slot, n_local, d_x_local, d_P_local), "lk_batch_get_states_dev");
    rcclCheck(ncclAllGather(d_x_local, d_x_all, n_local * LK_STATE_DOUBLES, ncclDouble, comm, st), "ncclAllGather (x)");
    rcclCheck(ncclAllGather(d_P_local, d_P_all, n_local * 900, ncclDouble, comm, st), "ncclAllGather (P)");
}

}  // namespace legkilo
