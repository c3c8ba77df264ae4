// bi_prim.h -- the device-wide sorts and scans the library uses, as plain functions: hand-written for gfx950 in tu_prim.hip
// (rounds 1-4 instantiated rocPRIM there: a kernel per architecture it knows for every algorithm -- ~1000 of the library's ~1250
// kernel symbols and 10 of its 12 MB).  Same argument order and temporary-storage protocol as rocprim:: (tmp == nullptr: only
// the size is returned in `bytes`).  Sorts are stable; every result depends on the input alone (fixed summation order).
#pragma once

hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const double* keys_in, double* keys_out, const int32_t* vals_in, int32_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const int64_t* keys_in, int64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
// keys < key_bound <= 1024 (larger keys count as key_bound - 1): a counting sort, one pass; the same (stable) order as prim_sort_pairs
hipError_t prim_count_sort_pairs(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out,
                                 size_t n, uint64_t key_bound, hipStream_t stream);
hipError_t prim_inclusive_scan_max(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream);
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream);
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const double* in, double* out, size_t n, hipStream_t stream);
hipError_t prim_exclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, int64_t init, size_t n, hipStream_t stream);
