// bi_prim.h -- the device-wide sorts and scans the library uses, as plain functions: rocPRIM is instantiated once, in
// tu_prim.hip (each rocPRIM algorithm brings a kernel per architecture it knows, 13 of them -- ~1000 of the library's ~1250
// kernel symbols and a third of the code object came from including it in the main translation unit).  Same argument order
// and temporary-storage protocol as rocprim:: (tmp == nullptr: only the size is returned in `bytes`).
#pragma once

hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const double* keys_in, double* keys_out, const int32_t* vals_in, int32_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const int64_t* keys_in, int64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
hipError_t prim_inclusive_scan_max(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream);
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream);
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const double* in, double* out, size_t n, hipStream_t stream);
hipError_t prim_exclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, int64_t init, size_t n, hipStream_t stream);
