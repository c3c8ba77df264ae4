// tu_prim.hip -- translation unit of the rocPRIM sorts and scans (bi_prim.h).  See bi_common.h for how the library is split.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "bi_prim.h"

#define BI_SORT_PAIRS(K, V)                                                                                                      \
    hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, size_t n, \
                               unsigned begin_bit, unsigned end_bit, hipStream_t stream) {                                        \
        return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);        \
    }
BI_SORT_PAIRS(double, int32_t)
BI_SORT_PAIRS(uint64_t, int64_t)
BI_SORT_PAIRS(int64_t, int32_t)
#undef BI_SORT_PAIRS

hipError_t prim_inclusive_scan_max(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream) {
    return rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::maximum<int64_t>(), stream);
}
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream) {
    return rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::plus<int64_t>(), stream);
}
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const double* in, double* out, size_t n, hipStream_t stream) {
    return rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::plus<double>(), stream);
}
hipError_t prim_exclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, int64_t init, size_t n, hipStream_t stream) {
    return rocprim::exclusive_scan(tmp, bytes, in, out, init, n, rocprim::plus<int64_t>(), stream);
}
