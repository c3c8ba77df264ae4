// tu_prim.hip -- the device-wide scans and the radix sort the library uses (bi_prim.h), hand-written for gfx950 (round 5).
//
// Rounds 1-4 took these from rocPRIM: correct and fast, but every rocPRIM algorithm instantiates its kernels once per GPU
// architecture it knows (13 of them) -- ~1000 of the library's ~1250 kernel symbols and, through their names, 10 of its 12 MB.
// None of the uses is on the hot path (the device planner's (cell, dataset) sort of a batch, prefix sums of list lengths, the
// cumulative sums behind toy generation, count-ordering of rows at data upload), so plain, deterministic versions do:
//   scan   three launches over tiles of 2048 elements: tile totals, a one-block scan of the totals, the tiles again with
//          their carry.  Sums are grouped thread (8 consecutive elements) -> block tree -> tiles in order: a fixed order, so
//          floating-point results do not depend on the launch.
//   sort   least-significant-digit radix sort of (64-bit key, value) pairs, 4 bits per pass over [begin_bit, end_bit): per pass a
//          histogram of the digits per tile ([digit][tile]), its exclusive scan (the scan above), and a stable scatter -- a
//          thread owns 8 consecutive elements, its rank is (elements of smaller digits in the tile) + (same digit in earlier
//          threads) + (same digit earlier in the thread), from one scan over the [16][256] counters in LDS.  Stable, and
//          independent of everything but the input: every rank of a dealt scan sorts its points into the same order.
//          Doubles and signed keys are compared through the usual order-preserving bit transforms.
// Same argument order and temporary-storage protocol as rocprim:: (tmp == nullptr: only the size is returned in `bytes`).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <type_traits>

#include "bi_prim.h"

namespace {

constexpr int kPT = 256;                 // threads per block
constexpr int kPI = 8;                   // elements per thread
constexpr int kPTile = kPT * kPI;        // 2048 elements per tile

struct OpSum { template <class T> __device__ static T apply(T a, T b) { return a + b; } template <class T> __device__ static T identity() { return (T)0; } };
struct OpMax { template <class T> __device__ static T apply(T a, T b) { return a > b ? a : b; } template <class T> __device__ static T identity() { return (T)INT64_MIN; } };

// inclusive scan of one value per thread over the block (Hillis-Steele in LDS: fixed order); returns the inclusive value, total via *tot
template <class T, class Op>
__device__ __forceinline__ T block_scan_inclusive(T v, T* sh /*[kPT]*/, T* tot) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < kPT; off <<= 1) {
        T u = v;
        if (t >= off) u = Op::apply(sh[t - off], v);
        __syncthreads();
        v = u;
        sh[t] = v;
        __syncthreads();
    }
    if (tot) *tot = sh[kPT - 1];
    __syncthreads();
    return v;
}

template <class T, class Op>
__global__ __launch_bounds__(kPT) void k_scan_tile_totals(const T* __restrict__ in, size_t n, T* __restrict__ totals) {
    __shared__ T sh[kPT];
    const size_t base = (size_t)blockIdx.x * kPTile + (size_t)threadIdx.x * kPI;
    T acc = Op::template identity<T>();
#pragma unroll
    for (int j = 0; j < kPI; ++j)
        if (base + j < n) acc = Op::apply(acc, in[base + j]);
    T tot;
    (void)block_scan_inclusive<T, Op>(acc, sh, &tot);
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

// one block: exclusive scan of the tile totals, in place (carry of tile b = everything before it)
template <class T, class Op>
__global__ __launch_bounds__(kPT) void k_scan_spine(T* __restrict__ totals, size_t nb) {
    __shared__ T sh[kPT];
    const size_t per = (nb + kPT - 1) / kPT;
    const size_t lo = (size_t)threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    T acc = Op::template identity<T>();
    for (size_t i = lo; i < hi; ++i) acc = Op::apply(acc, totals[i]);
    const T incl = block_scan_inclusive<T, Op>(acc, sh, (T*)nullptr);
    // exclusive prefix of this thread's range = inclusive of the previous thread
    __shared__ T prev[kPT];
    prev[threadIdx.x] = incl;
    __syncthreads();
    T run = threadIdx.x ? prev[threadIdx.x - 1] : Op::template identity<T>();
    for (size_t i = lo; i < hi; ++i) {
        const T v = totals[i];
        totals[i] = run;
        run = Op::apply(run, v);
    }
}

template <class T, class Op, bool EXCLUSIVE>
__global__ __launch_bounds__(kPT) void k_scan_apply(const T* __restrict__ in, T* __restrict__ out, size_t n, const T* __restrict__ carry, T init) {
    __shared__ T sh[kPT];
    const size_t base = (size_t)blockIdx.x * kPTile + (size_t)threadIdx.x * kPI;
    T v[kPI];
    T acc = Op::template identity<T>();
#pragma unroll
    for (int j = 0; j < kPI; ++j) {
        v[j] = base + j < n ? in[base + j] : Op::template identity<T>();
        acc = Op::apply(acc, v[j]);
    }
    const T incl = block_scan_inclusive<T, Op>(acc, sh, (T*)nullptr);
    __shared__ T prev[kPT];
    prev[threadIdx.x] = incl;
    __syncthreads();
    T run = Op::apply(carry[blockIdx.x], threadIdx.x ? prev[threadIdx.x - 1] : Op::template identity<T>());
    if (EXCLUSIVE) run = Op::apply(init, run);
#pragma unroll
    for (int j = 0; j < kPI; ++j) {
        if (base + j < n) {
            if (EXCLUSIVE) { out[base + j] = run; run = Op::apply(run, v[j]); }
            else { run = Op::apply(run, v[j]); out[base + j] = run; }
        }
    }
}

// up to kSmallScan elements: ONE block does the whole scan (the radix sort's digit histograms: 16 counters per tile -- 7 824 for 10^6
// pairs; three launches for that were a third of a sorting pass)
constexpr size_t kSmallScan = 32768;
template <class T, class Op, bool EXCLUSIVE>
__global__ __launch_bounds__(kPT) void k_scan_small(const T* __restrict__ in, T* __restrict__ out, size_t n, T init) {
    __shared__ T sh[kPT];
    __shared__ T prev[kPT];
    const size_t per = (n + kPT - 1) / kPT;
    const size_t lo = (size_t)threadIdx.x * per < n ? (size_t)threadIdx.x * per : n, hi = lo + per < n ? lo + per : n;
    T acc = Op::template identity<T>();
    for (size_t i = lo; i < hi; ++i) acc = Op::apply(acc, in[i]);
    const T incl = block_scan_inclusive<T, Op>(acc, sh, (T*)nullptr);
    prev[threadIdx.x] = incl;
    __syncthreads();
    T run = threadIdx.x ? prev[threadIdx.x - 1] : Op::template identity<T>();
    if (EXCLUSIVE) run = Op::apply(init, run);
    for (size_t i = lo; i < hi; ++i) {
        const T v = in[i];                  // (read before the write: in == out is allowed)
        if (EXCLUSIVE) { out[i] = run; run = Op::apply(run, v); }
        else { run = Op::apply(run, v); out[i] = run; }
    }
}

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

template <class T, class Op, bool EXCLUSIVE>
hipError_t scan_impl(void* tmp, size_t& bytes, const T* in, T* out, T init, size_t n, hipStream_t stream) {
    const size_t nb = (n + kPTile - 1) / kPTile;
    const size_t need = align256((nb ? nb : 1) * sizeof(T));
    if (!tmp) { bytes = need; return hipSuccess; }
    if (bytes < need) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (n <= kSmallScan) {
        hipLaunchKernelGGL((k_scan_small<T, Op, EXCLUSIVE>), dim3(1), dim3(kPT), 0, stream, in, out, n, init);
        return hipGetLastError();
    }
    T* totals = (T*)tmp;
    hipLaunchKernelGGL((k_scan_tile_totals<T, Op>), dim3((unsigned)nb), dim3(kPT), 0, stream, in, n, totals);
    hipLaunchKernelGGL((k_scan_spine<T, Op>), dim3(1), dim3(kPT), 0, stream, totals, nb);
    hipLaunchKernelGGL((k_scan_apply<T, Op, EXCLUSIVE>), dim3((unsigned)nb), dim3(kPT), 0, stream, in, out, n, (const T*)totals, init);
    return hipGetLastError();
}

// ---- radix sort -------------------------------------------------------------------------------------------------------------
enum KeyKind { kKeyUnsigned = 0, kKeySigned = 1, kKeyDouble = 2 };

template <int KIND>
__device__ __forceinline__ uint64_t sortable(uint64_t bits) {
    if (KIND == kKeySigned) return bits ^ 0x8000000000000000ull;
    if (KIND == kKeyDouble) return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);   // negative: all bits flipped; else the sign bit set
    return bits;
}

template <int KIND>
__global__ __launch_bounds__(kPT) void k_sort_hist(const uint64_t* __restrict__ keys, size_t n, int shift, size_t n_tiles, int64_t* __restrict__ hist /*[16][n_tiles]*/) {
    __shared__ unsigned cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kPTile + (size_t)threadIdx.x * kPI;
    unsigned local[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) local[d] = 0u;
#pragma unroll
    for (int j = 0; j < kPI; ++j)
        if (base + j < n) {
            const unsigned dg = (unsigned)(sortable<KIND>(keys[base + j]) >> shift) & 15u;
#pragma unroll
            for (int d = 0; d < 16; ++d) local[d] += dg == (unsigned)d ? 1u : 0u;
        }
#pragma unroll
    for (int d = 0; d < 16; ++d)
        if (local[d]) atomicAdd(&cnt[d], local[d]);       // (integer counts: the order of the additions does not matter)
    __syncthreads();
    if (threadIdx.x < 16) hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = (int64_t)cnt[threadIdx.x];
}

template <int KIND, class V>
__global__ __launch_bounds__(kPT) void k_sort_scatter(const uint64_t* __restrict__ keys_in, const V* __restrict__ vals_in, size_t n, int shift,
                                                      size_t n_tiles, const int64_t* __restrict__ gbase /*[16][n_tiles]: exclusive scan of hist*/,
                                                      uint64_t* __restrict__ keys_out, V* __restrict__ vals_out) {
    __shared__ unsigned cnt[16 * kPT];          // [digit][thread]
    __shared__ unsigned sh[kPT];
    __shared__ unsigned prev[kPT];
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * kPTile + (size_t)t * kPI;
    uint64_t k[kPI];
    unsigned dg[kPI];
    unsigned local[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) local[d] = 0u;
#pragma unroll
    for (int j = 0; j < kPI; ++j) {
        k[j] = base + j < n ? keys_in[base + j] : 0ull;
        dg[j] = base + j < n ? (unsigned)(sortable<KIND>(k[j]) >> shift) & 15u : 16u;
#pragma unroll
        for (int d = 0; d < 16; ++d) local[d] += dg[j] == (unsigned)d ? 1u : 0u;
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d * kPT + t] = local[d];
    __syncthreads();
    // exclusive scan over the flattened [16][256] counters: thread u owns entries 16 u .. 16 u + 15
    unsigned mine[16], acc = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) { mine[q] = cnt[t * 16 + q]; acc += mine[q]; }
    sh[t] = acc;
    __syncthreads();
    unsigned v = acc;
#pragma unroll
    for (int off = 1; off < kPT; off <<= 1) {
        unsigned u = v;
        if (t >= off) u = sh[t - off] + v;
        __syncthreads();
        v = u;
        sh[t] = v;
        __syncthreads();
    }
    prev[t] = v;
    __syncthreads();
    unsigned run = t ? prev[t - 1] : 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const unsigned c = mine[q]; cnt[t * 16 + q] = run; run += c; }
    __syncthreads();
    // element j of this thread, digit d: position in the tile's digit-d run = prefix[d][t] - prefix[d][0] + earlier same-digit elements here
    unsigned seen[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) seen[d] = 0u;
#pragma unroll
    for (int j = 0; j < kPI; ++j) {
        if (base + j < n) {
            unsigned within = 0u, start = 0u, mypre = 0u;
#pragma unroll
            for (int d = 0; d < 16; ++d)
                if (dg[j] == (unsigned)d) { within = seen[d]; seen[d] += 1u; }
            start = cnt[dg[j] * kPT];
            mypre = cnt[dg[j] * kPT + t];
            const size_t dst = (size_t)gbase[(size_t)dg[j] * n_tiles + blockIdx.x] + (mypre - start) + within;
            keys_out[dst] = k[j];
            vals_out[dst] = vals_in[base + j];
        }
    }
}

template <int KIND, class V>
hipError_t sort_impl(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const V* vals_in, V* vals_out, size_t n,
                     unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    const size_t n_tiles = (n + kPTile - 1) / kPTile;
    size_t scan_bytes = 0;
    (void)scan_impl<int64_t, OpSum, true>(nullptr, scan_bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, 16 * (n_tiles ? n_tiles : 1), stream);
    const size_t b_keys = align256((n ? n : 1) * sizeof(uint64_t)), b_vals = align256((n ? n : 1) * sizeof(V));
    const size_t b_hist = align256(16 * (n_tiles ? n_tiles : 1) * sizeof(int64_t));
    const size_t need = b_keys + b_vals + 2 * b_hist + scan_bytes;
    if (!tmp) { bytes = need; return hipSuccess; }
    if (bytes < need) return hipErrorInvalidValue;
    if (end_bit > 64 || begin_bit > end_bit) return hipErrorInvalidValue;
    const int passes = (int)((end_bit - begin_bit + 3) / 4);
    if (n == 0) return hipSuccess;
    if (passes == 0) {            // nothing to compare: a copy
        hipError_t e = hipMemcpyAsync(keys_out, keys_in, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(vals_out, vals_in, n * sizeof(V), hipMemcpyDeviceToDevice, stream);
        return e;
    }
    char* p = (char*)tmp;
    uint64_t* tk = (uint64_t*)p; p += b_keys;
    V* tv = (V*)p; p += b_vals;
    int64_t* hist = (int64_t*)p; p += b_hist;
    int64_t* gbase = (int64_t*)p; p += b_hist;
    void* scan_tmp = p;
    const uint64_t* src_k = keys_in;
    const V* src_v = vals_in;
    for (int pass = 0; pass < passes; ++pass) {
        // the last pass must land in (keys_out, vals_out): passes alternate between the temporary pair and the output pair
        const bool to_out = ((passes - 1 - pass) & 1) == 0;
        uint64_t* dst_k = to_out ? keys_out : tk;
        V* dst_v = to_out ? vals_out : tv;
        const int shift = (int)begin_bit + 4 * pass;
        hipLaunchKernelGGL((k_sort_hist<KIND>), dim3((unsigned)n_tiles), dim3(kPT), 0, stream, src_k, n, shift, n_tiles, hist);
        size_t sb = scan_bytes;
        hipError_t e = scan_impl<int64_t, OpSum, true>(scan_tmp, sb, (const int64_t*)hist, gbase, (int64_t)0, 16 * n_tiles, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_sort_scatter<KIND, V>), dim3((unsigned)n_tiles), dim3(kPT), 0, stream, src_k, src_v, n, shift, n_tiles,
                           (const int64_t*)gbase, dst_k, dst_v);
        src_k = dst_k;
        src_v = dst_v;
    }
    return hipGetLastError();
}


// ---- counting sort for keys below a small bound (the device planner's cell keys of a profile scan: 126 values) ---------------
// One pass instead of ceil(bits / 4): a histogram of the KEYS per tile ([key][tile], counters in LDS), its exclusive scan, and a
// stable scatter in which a wave owns a contiguous quarter of the tile and walks it 64 elements at a time: an element's rank is
// (its key in earlier tiles, from the scan) + (in earlier waves of the tile) + (earlier in the wave's walk) + (in lower lanes of
// the same 64, from a match over the key's bits with ballots).  Same order as the radix sort gives (both are stable).
constexpr int kCsMaxKeys = 1024;

__global__ __launch_bounds__(kPT) void k_csort_hist(const uint64_t* __restrict__ keys, size_t n, int K, size_t n_tiles, int64_t* __restrict__ hist /*[K][n_tiles]*/) {
    __shared__ unsigned cnt[kCsMaxKeys];
    for (int i = threadIdx.x; i < K; i += kPT) cnt[i] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kPTile;
#pragma unroll
    for (int j = 0; j < kPI; ++j) {
        const size_t i = base + (size_t)j * kPT + threadIdx.x;
        if (i < n) atomicAdd(&cnt[min((uint64_t)(K - 1), keys[i])], 1u);      // (integer counts: the order of the additions does not matter)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += kPT) hist[(size_t)i * n_tiles + blockIdx.x] = (int64_t)cnt[i];
}

template <class V>
__global__ __launch_bounds__(kPT) void k_csort_scatter(const uint64_t* __restrict__ keys_in, const V* __restrict__ vals_in, size_t n, int K, int nbits,
                                                       size_t n_tiles, const int64_t* __restrict__ gbase /*[K][n_tiles]: exclusive scan of hist*/,
                                                       uint64_t* __restrict__ keys_out, V* __restrict__ vals_out) {
    constexpr int kWaves = kPT / 64, kPerWave = kPTile / kWaves, kRounds = kPerWave / 64;
    __shared__ unsigned cntw[kWaves][kCsMaxKeys];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < kWaves * kCsMaxKeys; i += kPT) (&cntw[0][0])[i] = 0u;
    __syncthreads();
    const size_t wbase = (size_t)blockIdx.x * kPTile + (size_t)wave * kPerWave;
    uint64_t k[kRounds];
    unsigned key[kRounds], rank[kRounds];
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
        const size_t i = wbase + (size_t)j * 64 + lane;
        const bool valid = i < n;
        k[j] = valid ? keys_in[i] : 0ull;
        key[j] = (unsigned)min((uint64_t)(K - 1), k[j]);
        // the lanes of this round that hold my key
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (key[j] >> b) & 1u;
            const unsigned long long with = __ballot(valid && bit);
            same &= bit ? with : ~with;
        }
        const unsigned long long below = same & ((1ull << lane) - 1ull);
        const unsigned old = valid ? cntw[wave][key[j]] : 0u;
        rank[j] = old + (unsigned)__popcll(below);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // every lane has read the counter before the key's first lane moves it on
        if (valid && below == 0ull) cntw[wave][key[j]] = old + (unsigned)__popcll(same);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    for (int q = t; q < K; q += kPT) {                          // counts per wave -> the key's elements in earlier waves of the tile
        unsigned run = 0u;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const unsigned c = cntw[w][q]; cntw[w][q] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
        const size_t i = wbase + (size_t)j * 64 + lane;
        if (i < n) {
            const size_t dst = (size_t)gbase[(size_t)key[j] * n_tiles + blockIdx.x] + cntw[wave][key[j]] + rank[j];
            keys_out[dst] = k[j];
            vals_out[dst] = vals_in[i];
        }
    }
}

template <class V>
hipError_t csort_impl(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const V* vals_in, V* vals_out, size_t n,
                      uint64_t key_bound, hipStream_t stream) {
    const size_t n_tiles = (n + kPTile - 1) / kPTile;
    if (key_bound < 1 || key_bound > (uint64_t)kCsMaxKeys) return hipErrorInvalidValue;
    const int K = (int)key_bound;
    size_t scan_bytes = 0;
    (void)scan_impl<int64_t, OpSum, true>(nullptr, scan_bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)K * (n_tiles ? n_tiles : 1), stream);
    const size_t b_hist = align256((size_t)K * (n_tiles ? n_tiles : 1) * sizeof(int64_t));
    const size_t need = 2 * b_hist + scan_bytes;
    if (!tmp) { bytes = need; return hipSuccess; }
    if (bytes < need) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    int nbits = 0;
    while ((1 << nbits) < K) ++nbits;
    char* p = (char*)tmp;
    int64_t* hist = (int64_t*)p; p += b_hist;
    int64_t* gbase = (int64_t*)p; p += b_hist;
    hipLaunchKernelGGL(k_csort_hist, dim3((unsigned)n_tiles), dim3(kPT), 0, stream, keys_in, n, K, n_tiles, hist);
    size_t sb = scan_bytes;
    hipError_t e = scan_impl<int64_t, OpSum, true>(p, sb, (const int64_t*)hist, gbase, (int64_t)0, (size_t)K * n_tiles, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_csort_scatter<V>), dim3((unsigned)n_tiles), dim3(kPT), 0, stream, keys_in, vals_in, n, K, nbits, n_tiles,
                       (const int64_t*)gbase, keys_out, vals_out);
    return hipGetLastError();
}

}  // namespace

hipError_t prim_count_sort_pairs(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out,
                                 size_t n, uint64_t key_bound, hipStream_t stream) {
    return csort_impl<int64_t>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, key_bound, stream);
}

hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const double* keys_in, double* keys_out, const int32_t* vals_in, int32_t* vals_out, size_t n,
                           unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    return sort_impl<kKeyDouble, int32_t>(tmp, bytes, (const uint64_t*)keys_in, (uint64_t*)keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}
hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out, size_t n,
                           unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    return sort_impl<kKeyUnsigned, int64_t>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}
hipError_t prim_sort_pairs(void* tmp, size_t& bytes, const int64_t* keys_in, int64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, size_t n,
                           unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    return sort_impl<kKeySigned, int32_t>(tmp, bytes, (const uint64_t*)keys_in, (uint64_t*)keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}
hipError_t prim_inclusive_scan_max(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream) {
    return scan_impl<int64_t, OpMax, false>(tmp, bytes, in, out, (int64_t)0, n, stream);
}
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t stream) {
    return scan_impl<int64_t, OpSum, false>(tmp, bytes, in, out, (int64_t)0, n, stream);
}
hipError_t prim_inclusive_scan_sum(void* tmp, size_t& bytes, const double* in, double* out, size_t n, hipStream_t stream) {
    return scan_impl<double, OpSum, false>(tmp, bytes, in, out, 0.0, n, stream);
}
hipError_t prim_exclusive_scan_sum(void* tmp, size_t& bytes, const int64_t* in, int64_t* out, int64_t init, size_t n, hipStream_t stream) {
    return scan_impl<int64_t, OpSum, true>(tmp, bytes, in, out, init, n, stream);
}
