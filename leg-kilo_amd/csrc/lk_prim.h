// lk_prim.h - the three rocPRIM device-wide primitives the library uses, behind plain functions.
// They live in a translation unit of their own (lk_prim.hip): rocPRIM is header-only and its sorts and scans are the slowest templates of the
// build to instantiate; compiled beside legkilo_hip.hip instead of inside it they cost no wall-clock time (make -j2).
// Calling convention = rocPRIM's: tmp == nullptr asks for the temporary storage size in `bytes`, the second call does the work on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

// stable radix sort of (unsigned key, int value) pairs over bits [bit0, bit1)   (BuildVoxelMap's voxel sort, the preprocessing's cell / time sorts)
hipError_t lk_prim_sort_pairs(void* tmp, size_t& bytes, const unsigned int* keys_in, unsigned int* keys_out, const int* vals_in, int* vals_out, size_t n,
                              unsigned int bit0, unsigned int bit1, hipStream_t stream);
// exclusive prefix sum of unsigned counts, starting at 0   (pool compaction of the sliding window, the device-built bucket tables, voxel-grid cells)
hipError_t lk_prim_exclusive_scan(void* tmp, size_t& bytes, const unsigned int* in, unsigned int* out, size_t n, hipStream_t stream);
// stable segmented radix sort of (unsigned key, unsigned value) pairs: segment s = [begin[s], end[s])   (lk_batch_sort_by_voxel_dev: a segment per time bucket)
hipError_t lk_prim_segmented_sort_pairs(void* tmp, size_t& bytes, const unsigned int* keys_in, unsigned int* keys_out, const unsigned int* vals_in,
                                        unsigned int* vals_out, unsigned int n, unsigned int n_segments, const unsigned int* begin, const unsigned int* end,
                                        unsigned int bit0, unsigned int bit1, hipStream_t stream);
