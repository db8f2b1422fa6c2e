// lk_prim.hip - see lk_prim.h.  Nothing but the instantiations.
#include "lk_prim.h"

#include <cstring>   // (rocPRIM's texture iterator calls memset without including it)

#include <rocprim/rocprim.hpp>

hipError_t lk_prim_sort_pairs(void* tmp, size_t& bytes, const unsigned int* keys_in, unsigned int* keys_out, const int* vals_in, int* vals_out, size_t n,
                              unsigned int bit0, unsigned int bit1, hipStream_t stream) {
    return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, bit0, bit1, stream);
}
hipError_t lk_prim_exclusive_scan(void* tmp, size_t& bytes, const unsigned int* in, unsigned int* out, size_t n, hipStream_t stream) {
    return rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<unsigned int>(), stream);
}
hipError_t lk_prim_segmented_sort_pairs(void* tmp, size_t& bytes, const unsigned int* keys_in, unsigned int* keys_out, const unsigned int* vals_in,
                                        unsigned int* vals_out, unsigned int n, unsigned int n_segments, const unsigned int* begin, const unsigned int* end,
                                        unsigned int bit0, unsigned int bit1, hipStream_t stream) {
    return rocprim::segmented_radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, n_segments, begin, end, bit0, bit1, stream);
}
