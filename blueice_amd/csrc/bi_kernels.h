// bi_kernels.h -- every __global__ / __device__ function of libblueice_hip (gfx950 only).
// Included once by blueice_hip.hip; see DESIGN.md section 4 for what each kernel is for and what bounds it.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned wave_or(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_down(v, off, 64);
    return v;
}

// Natural logarithm for the per-bin terms, table-driven (the scheme of Tang's table-driven log as used by modern
// libms, laid out for this hardware): x = 2^k z with z in [0.6875, 1.375); the 7 leading mantissa bits pick a
// subinterval with centre c from a 128-entry {1/c, log c hi, log c lo} table held in LDS (bi_log_table.h); then
//     log x = (k ln2_hi + log c_hi) + ( r + (r^2 P(r) + k ln2_lo + log c_lo) ),   r = z / c - 1   (one fma, |r| <= 2^-7)
// where the first bracket is EXACT in one fma (ln2_hi on a 2^-37 grid, log c_hi on a 2^-43 grid), P is log1p's Taylor
// polynomial through r^8, and the sum of the first bracket and r is carried with its rounding error.  16 fp64
// instructions, one conversion and 6 integer ones per call (round 1: 19 + 1 + 6: it also had to recover the rounding of an
// inexact k ln2 + log c), two LDS reads, no division and no transcendental-rate instruction; worst error 0.70 ulp, 98.6 %
// of results correctly rounded
// (tools/gen_log_table.py, tests/test_gpu_golden.py::test_device_log_accuracy).  On a chip where no vector instruction
// executes beside an fp64 MFMA, the logarithm's instruction count is what bounds scans over dense data.
// Every kernel that calls bin_log fills the LDS table first: log_table_load(), or the overlapped form in morph_tiles.
__shared__ double4 s_log_table[128];

__device__ __forceinline__ void log_table_load() {
    if (threadIdx.x < 128) s_log_table[threadIdx.x] = kLogTable[threadIdx.x];
    __syncthreads();
}

__device__ __forceinline__ bool pos_normal(double x) { return __builtin_amdgcn_class(x, 0x100); }
// Factors of the product forms of sum n log mu (k_scan_mfma): up to eight of them, each above 2^-127, multiply to at least
// 2^-1016 -- a normal number -- in any grouping; a comparison with it is false for nan, zero and negative numbers too.
constexpr double kProdFloor = 0x1p-127;

// the core: x must be a positive normal number (anything else gives a meaningless but harmless value);
// k_adjust is added to the binary exponent
__device__ __forceinline__ double log_core(double x, int k_adjust) {
    const unsigned long long ix = __double_as_longlong(x);
    const int hi = (int)(ix >> 32);
    const int t = hi - 0x3FE60000;                  // bits(x) - bits(0.6875), high word
    const int k0 = t >> 20;
    const int k = k0 + k_adjust;
    // high word of z = hi - (k0 << 20), as ONE 24-bit multiply-add (|k0| <= 2^10, 2^20 < 2^23; written as an instruction
    // because the compiler turns the product back into a mask and a subtraction)
    int zhi;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(zhi) : "v"(k0), "s"(-(1 << 20)), "v"(hi));
    const double z = __longlong_as_double(((unsigned long long)(unsigned)zhi << 32) | (ix & 0xFFFFFFFFull));
    const double4 e = s_log_table[(t >> 13) & 127];
    const double kd = (double)k;
    const double r = fma(z, e.x, -1.0);
    const double w = fma(kd, kLn2Hi, e.y);         // exact
    const double tail = fma(kd, kLn2Lo, e.z);
    double p = fma(r, -1.0 / 8.0, e.w);            // e.w = 1/7: arrives in a vector register with the table entry
    p = fma(r, p, -1.0 / 6.0);
    p = fma(r, p, 1.0 / 5.0);
    p = fma(r, p, -1.0 / 4.0);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    const double q = fma(r * r, p, tail);
    // w + r with its rounding error kept (|w| >= |r| wherever w != 0: k != 0, or a subinterval away from the two that touch
    // 1): where log c and r nearly cancel -- arguments a little off 1 -- the plain r + q would cost up to an ulp
    const double h = w + r;
    const double err = (w - h) + r;
    return h + (err + q);
}

// for arguments known to be positive normal numbers
__device__ __forceinline__ double bin_log_fast(double x) { return log_core(x, 0); }

// for any argument, still without a branch: denormals are scaled by 2^54 first; log 0 = -inf, log of a negative
// number or nan = nan, log inf = inf (numpy.log's values)
__device__ __forceinline__ double bin_log(double x) {
    const bool tiny = x < 2.2250738585072014e-308;
    double y = log_core(tiny ? x * 18014398509481984.0 : x, tiny ? -54 : 0);
    if (x == 0.0) y = -__builtin_inf();
    if (!(x >= 0.0)) y = __builtin_nan("");
    if (x == __builtin_inf()) y = x;
    return y;
}

// self-test hook: out[i] = bin_log(x[i])
__global__ void k_selftest_log(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
    log_table_load();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bin_log(x[i]);
}

template <bool NT>
__device__ __forceinline__ double2 stream_load(const double* p) {
    if constexpr (NT) {
        // streamed-once data: nontemporal hint (global_load_dwordx4 ... nt) keeps it from displacing L2 / MALL lines
        double2 v;
        v.x = __builtin_nontemporal_load(p);
        v.y = __builtin_nontemporal_load(p + 1);
        return v;
    } else {
        return *reinterpret_cast<const double2*>(p);
    }
}

// measurement probe: a plain sum over n2 16-byte elements -- the read-only streaming ceiling the morph kernel is
// compared with (bi_measure_read_bandwidth)
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_read_sum(const double* __restrict__ p, int64_t n2, double* __restrict__ sink) {
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 7 * stride < n2; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = stream_load<NT>(p + 2 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) s += p[2 * i] + p[2 * i + 1];
    if (s == 0.123456789) sink[0] = s;      // keeps the loads alive, practically never stores
}

// measurement probe with the morph kernel's access pattern and nothing else: blockIdx.y = item, every item streams
// `rows` template rows of its own (row r of item i starts at element ((first + i * rows + r) % total_rows) * Bp), the
// tiles of a row are walked in the XCD-aware order of morph_tiles, 16 bytes per lane per row, `rows` loads in flight
// per lane in batches of 8 -- the ceiling `k_morph_reduce` can be held against (bi_measure_stream_bandwidth)
// PIECES: 16-byte pieces per lane per row (1 = the morph kernel's own 512-bin tiles; 2 = 1024-bin tiles, i.e. twice the
// contiguous run per row per block -- to see whether a wider tile would raise the ceiling: it does not, by more than 1 %)
template <bool NT, int PIECES>
__global__ __launch_bounds__(kThreads) void k_read_rows(const double* __restrict__ ps, int64_t Bp, int64_t total_rows,
                                                        int64_t first, int rows, int n_tiles, int chunks_in,
                                                        double* __restrict__ sink) {
    const int64_t row0 = first + (int64_t)blockIdx.y * rows;
    const int chunks = (chunks_in > 1 && n_tiles >= 64 * chunks_in) ? chunks_in : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    double s = 0.0;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile * PIECES + threadIdx.x * kBinsPerThread;
#pragma unroll 8
        for (int r = 0; r < rows; ++r) {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const double2 v = stream_load<NT>(ps + ((row0 + r) % total_rows) * Bp + bin0 + q * kTile);
                s += v.x + v.y;
            }
        }
    }
    if (s == 0.123456789) sink[0] = s;      // keeps the loads alive, practically never stores
}

// Poisson log-pmf without the data-only lgamma(n+1) term, scipy semantics
// (scipy/stats/_distn_infrastructure.py logpmf + _discrete_distns.py poisson._logpmf):
//   mu not >= 0 (negative or nan) or n nan -> nan
//   n negative or non-integer             -> -inf
//   else xlogy(n, mu) - mu                  (xlogy(0, mu) = 0, also for mu = 0)
__device__ __forceinline__ double poisson_term(double n, double mu) {
    double t;
    if (n > 0.0) {
        t = n * bin_log(mu) - mu;  // mu = 0 -> -inf; mu < 0 -> nan
    } else {
        t = -mu;
    }
    if (!(mu >= 0.0) || n != n) t = __builtin_nan("");
    else if (n < 0.0 || n != floor(n)) t = -__builtin_inf();
    return t;
}

// The same term without a branch, for unrolled loops (the compiler can then batch the table reads of many terms).
// Only valid where n > 0 implies that mu is a positive normal number -- the caller checks that for the whole wave
// (needs_checked_term) and takes poisson_term otherwise.  Same operations, same bits.
__device__ __forceinline__ bool needs_checked_term(double n, double mu) { return n > 0.0 && !pos_normal(mu); }

__device__ __forceinline__ double poisson_term_fast(double n, double mu) {
    const double lg = bin_log_fast(mu);            // not used where n <= 0
    double t = (n > 0.0) ? n * lg - mu : -mu;
    if (!(mu >= 0.0) || n != n) t = __builtin_nan("");
    else if (n < 0.0 || n != floor(n)) t = -__builtin_inf();
    return t;
}

// ... for a bin column in which no lane has n > 0
__device__ __forceinline__ double poisson_term_nolog(double n, double mu) {
    double t = -mu;
    if (!(mu >= 0.0) || n != n) t = __builtin_nan("");
    else if (n < 0.0 || n != floor(n)) t = -__builtin_inf();
    return t;
}

// Beeston-Barlow roots, evaluated in the reference's own operation order without FMA
// contraction (blueice/likelihood.py:693-712) so that the sign tests behind its two asserts
// see the same rounding.
__device__ __forceinline__ void bb_roots(double a, double p, double U, double d, double& r1, double& r2) {
#pragma clang fp contract(off)
    double U2 = U * U, p2 = p * p, a2 = a * a, d2 = d * d;
    double disc = U2 * p2 + 2 * U2 * p + U2 + 2 * U * a * p2 + 2 * U * a * p - 2 * U * d * p2 - 2 * U * d * p +
                  a2 * p2 + 2 * a * d * p2 + d2 * p2;
    double lead = -U * p - U + a * p + d * p;
    double den = 2 * p * (p + 1);
    double sq = sqrt(disc);
    r1 = (lead - sq) / den;
    r2 = (lead + sq) / den;
}

// ---- in-launch finishing through mailboxes -------------------------------------------------------------------
// A work item's blocks post their partial sums into 8-byte mailbox slots and EXIT; the item's last block in dispatch
// order (blockIdx.x == gridDim.x - 1: every sibling was dispatched before it, so they are running or done) collects
// them, sums them in block order (fixed order => bitwise reproducible) and writes the result -- what the k_finish
// launch did, without the launch.  A slot is one naturally aligned 8-byte granule written by ONE system-scope
// (sc0 sc1, write-through) store and read with system-scope loads (MI355X_MICROARCH.md "Valid forms": sc0 sc1
// stores and loads on both sides need no fence); "empty" is a signalling-NaN bit pattern that no arithmetic result
// can have (posted NaNs are canonicalised), and the collector puts it back as it takes a value, so the slots are
// empty again when the launch ends.  Posting costs a block one store and no wait: round 2 first tried arrival
// tickets (publish, drain, returning atomics) and measured +27 us on a 300 us launch -- every one of 8192 blocks
// held its CU slot for ~4 us of round trips -- and round 1's release fence per block was worse still.
// The collector's wait is bounded (LaunchArgs::mail_timeout ticks of the 100 MHz wall clock, 2 s by default): if a value never arrives it gives
// up, reports BI_ST_INTERNAL and the result is nan -- no wave can spin forever.
constexpr unsigned long long kMailEmpty = 0x7FF4B10E1CE00001ull;
constexpr long long kMailTicksPerMs = 100000;                          // wall_clock64 runs at 100 MHz

__device__ __forceinline__ void mail_post(double* slot, double v) {
    if (v != v) v = __builtin_nan("");                                   // never the "empty" pattern
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// take the value out of a slot (waiting for it), leave the slot empty; *late is set if the wait ran out
__device__ __forceinline__ double mail_take(double* slot, long long deadline, bool* late) {
    unsigned long long bits;
    for (;;) {
        bits = __hip_atomic_load(reinterpret_cast<unsigned long long*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (bits != kMailEmpty) break;
        if ((long long)wall_clock64() > deadline) { *late = true; return __builtin_nan(""); }
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot), kMailEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __longlong_as_double(bits);
}

// The collector's inner step: the values of slots q0, q0 + stride, ... (up to 4, below `total`) added to s in that order.
// The four loads go out together -- a system-scope load takes about a microsecond, and by the time the last block
// collects nearly every sibling has posted, so polling one slot after the other would only add their latencies up
// (measured on a one-item launch of 1954 blocks x 8 columns: 86 us with sequential takes, the kernel proper 45).
__device__ __forceinline__ double mail_take4(double* mail, int q0, int stride, int total, double s, long long deadline, bool* late) {
    unsigned long long bits[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * stride;
        bits[u] = q < total ? __hip_atomic_load(reinterpret_cast<unsigned long long*>(mail + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                            : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * stride;
        if (q >= total) break;
        if (bits[u] == kMailEmpty) {
            s += mail_take(mail + q, deadline, late);            // not there yet: wait for this one
        } else {
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(mail + q), kMailEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s += __longlong_as_double(bits[u]);
        }
    }
    return s;
}

// Beeston-Barlow status bits travel through one word per result slot: a block that has any ORs them in BEFORE it
// posts its partial (returning atomic: performed when it returns), the collector swaps the word for 0 after the
// partials have arrived
__device__ __forceinline__ void flags_post(unsigned* word, unsigned f) {
    if (f) {
        const unsigned old = __hip_atomic_fetch_or(word, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("" ::"v"(old) : "memory");                           // the post below must not be hoisted above the return
    }
}
__device__ __forceinline__ unsigned flags_take(unsigned* word) {
    return __hip_atomic_exchange(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_mail_init(unsigned long long* slots, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = kMailEmpty;
}

struct LaunchArgs {
    const double* ps;       // [rows][Bp]
    const double* nm;       // [A][Bp] (BB) or null
    const double* counts;   // [T][Bp]
    const int64_t* rowoff;  // [items][NS]  element offsets of the stream rows
    const double* coef;     // [items][NS][G]
    const double* aux;      // [items][G][2]  (p_cal, N) for BB
    const int64_t* item_cnt; // [items] element offset of the item's counts row
    const int32_t* item_tiles; // [items] 512-bin tiles of the item's rows (NULL: n_tiles)
    double* partial;        // [items][nbx][G]
    unsigned* pflags;       // [items][nbx][G]
    int64_t B, Bp;
    double outlier;         // MODE 2: likelihood given to events with a non-positive density (0 = none)
    int n0, n1, n2;         // streams into U (or mu), into P_i, into a
    int n_tiles;
    int chunks;             // > 1: consecutive blocks work in `chunks` far-apart regions of the rows
    int n_keep;             // NT kernels: the first n_keep stream rows are loaded with the default (cacheable) policy
    // in-launch finish (k_morph_reduce): when fin_mail != NULL the item's last block collects the item's partials
    // from the mailbox slots and writes the results -- no k_finish launch behind the morph launch
    double* fin_mail;           // [items][nbx][G] mailbox slots, empty on entry and on exit
    unsigned* fin_flags;        // [items][G] status words (Beeston-Barlow), zero on entry and on exit
    const int64_t* fin_perm;    // [items][G]  result index of every slot (-1: unused slot)
    const double* fin_slot_lg;  // [items][G]  constant subtracted from the sum
    double* fin_out;            // results (device or pinned host memory)
    int32_t* fin_status;        // or NULL
    int nan_S;              // MODE 2 with non-finite pdf values: number of sources (streams are [corner][source]); the
                            // sum over sources then skips nan terms -- np.nansum, blueice/likelihood.py:686.  0 = off
    long long mail_timeout = 2000 * kMailTicksPerMs; // in-launch finish: ticks of the 100 MHz wall clock a collector waits (context: mail_timeout_ms;
                            // default 2 s, far beyond any delay a busy, shared GPU causes)
    int skip_post = -1, late_post = -1;   // fault injection (tests): this block never posts / posts after the collector gave up; -1 = off
};

// the post of a block's partial sum with the two injected faults (both -1 in production: two scalar compares per block)
__device__ __forceinline__ void mail_post_checked(const LaunchArgs& a, double* slot, double v) {
    if ((int)blockIdx.x == a.skip_post) return;
    if ((int)blockIdx.x == a.late_post) {
        const long long until = (long long)wall_clock64() + 2 * a.mail_timeout;
        while ((long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
    }
    mail_post(slot, v);
}

// The morph + reduce kernel.  blockIdx.y = item (a cell pass with up to G points),
// blockIdx.x strides over 512-bin tiles.

// The accumulate + per-bin term loop shared by the batched kernel and the single-point kernel: tiles
// tile0, tile0 + tile_step, ... of one work item.
template <int G, bool BB, bool NT, int MODE>
__device__ __forceinline__ void morph_tiles(const LaunchArgs& a, const int64_t* __restrict__ rowoff,
                                            const double* __restrict__ coef, const double* __restrict__ aux_base,
                                            const double* __restrict__ cnt, int n_tiles, int tile0, int tile_step,
                                            double (&sum)[G], unsigned (&flg)[G]) {
    // the log table travels global -> registers -> LDS; the request goes out first and lands under the first tile's
    // row loads, so a block that lives for only a few tiles does not wait for it separately
    double4 tab = {0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < 128) tab = kLogTable[threadIdx.x];
    bool tab_pending = true;
    // XCD-aware tile order: with 8 chunks block b -- dispatched to XCD b % 8 -- streams the b % 8-th contiguous region of
    // every row instead of every 8th tile (measured +6 % on the 113-stream BB pass, +1 % on C2); short rows keep the
    // plain order, where the padded chunk count would cost some blocks a second tile
    const int chunks = (a.chunks > 1 && n_tiles >= 64 * a.chunks) ? a.chunks : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    for (int lt = tile0; lt < per_chunk * chunks; lt += tile_step) {     // (the trip count is the same for a whole block)
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double acc[G][2];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }

        int k0 = 0;
        if constexpr (NT && G == 1 && MODE != 2 && MODE != 3) {
            // rows meant to stay in the Infinity Cache between calls (repeated evaluations in one cell): default policy
            k0 = a.n_keep;
#pragma unroll 8
            for (int k = 0; k < k0; ++k) {
                const double2 v = stream_load<false>(a.ps + rowoff[k] + bin0);
                const double c = coef[k];
                acc[0][0] = fma(c, v.x, acc[0][0]);
                acc[0][1] = fma(c, v.y, acc[0][1]);
            }
        }
        if constexpr (MODE == 2 || MODE == 3) if (a.nan_S > 0) {
            // np.nansum over sources (likelihood.py:686): a source whose morphed density times its rate is nan at an
            // event contributes nothing there.  Per source the corners are summed first (the morph), then the test.
            const int S = a.nan_S, nc = a.n0 / S;
            for (int s = 0; s < S; ++s) {
                double part[G][2];
#pragma unroll
                for (int g = 0; g < G; ++g) { part[g][0] = 0.0; part[g][1] = 0.0; }
                for (int c = 0; c < nc; ++c) {
                    const int k = c * S + s;
                    const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const double cf = coef[k * G + g];
                        part[g][0] = fma(cf, v.x, part[g][0]);
                        part[g][1] = fma(cf, v.y, part[g][1]);
                    }
                }
                if constexpr (MODE == 3) {        // gradient: a source dropped from the value is dropped from its slopes too
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (part[0][j] == part[0][j]) {
#pragma unroll
                            for (int g = 0; g < G; ++g) acc[g][j] += part[g][j];
                        }
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (part[g][0] == part[g][0]) acc[g][0] += part[g][0];
                        if (part[g][1] == part[g][1]) acc[g][1] += part[g][1];
                    }
                }
            }
            k0 = a.n0;
        }
#pragma unroll 8
        for (int k = k0; k < a.n0; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double c = coef[k * G + g];
                acc[g][0] = fma(c, v.x, acc[g][0]);
                acc[g][1] = fma(c, v.y, acc[g][1]);
            }
        }
        double2 nv;
        if constexpr (MODE == 2 || MODE == 3) { nv.x = nv.y = 0.0; } else { nv = *reinterpret_cast<const double2*>(cnt + bin0); }
        if (tab_pending) {
            if (threadIdx.x < 128) s_log_table[threadIdx.x] = tab;
            __syncthreads();
            tab_pending = false;
        }

        if constexpr (MODE == 2) {
            // extended unbinned likelihood (blueice/likelihood.py:678-690): the "bins" are the events,
            // the term is log(sum_s mu_s p_s(x_e)) with the outlier clamp; -sum_s mu_s is added by the host
            bool checked = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (a.outlier != 0.0 && !(acc[g][j] > 0.0)) acc[g][j] = a.outlier;
                    checked |= bin0 + j < a.B && !pos_normal(acc[g][j]);
                }
            }
            const bool fast = __ballot(checked) == 0ull;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double lg = fast ? bin_log_fast(acc[g][j]) : bin_log(acc[g][j]);   // (wave-uniform choice)
                    if (bin0 + j < a.B) sum[g] += lg;
                }
            }
        } else if constexpr (MODE == 3) {
            // value + gradient of the extended unbinned likelihood (blueice/likelihood.py:678-690): column 0 is the event's
            // density lambda_e = sum_s mu_s p_s(x_e), columns 1.. its derivatives; d log(lambda) = d lambda / lambda.  An event
            // that takes the outlier likelihood (lambda not > 0) is a constant: no slope.  -sum_s d mu_s is added by the host.
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double lam = acc[0][j];
                const bool clamped = a.outlier != 0.0 && !(lam > 0.0);
                if (clamped) lam = a.outlier;
                const double lg = bin_log(lam);
                if (bin0 + j < a.B) {
                    sum[0] += lg;
                    const double inv = clamped ? 0.0 : 1.0 / lam;
#pragma unroll
                    for (int g = 1; g < G; ++g) sum[g] += acc[g][j] * inv;
                }
            }
        } else if constexpr (MODE == 1) {
            sum[0] += poisson_term(nv.x, acc[0][0]) + poisson_term(nv.y, acc[0][1]);
            const double f0 = (nv.x != 0.0 ? nv.x / acc[0][0] : 0.0) - 1.0;
            const double f1 = (nv.y != 0.0 ? nv.y / acc[0][1] : 0.0) - 1.0;
#pragma unroll
            for (int g = 1; g < G; ++g) sum[g] += f0 * acc[g][0] + f1 * acc[g][1];
        } else if constexpr (!BB) {
            // per bin column of the wave: no lane has counts -> no logarithm at all (the usual case with sparse data);
            // every lane that needs one has a positive normal mu -> the branch-free form; else the checked form
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double n = j ? nv.y : nv.x;
                bool checked = false;
#pragma unroll
                for (int g = 0; g < G; ++g) checked |= needs_checked_term(n, acc[g][j]);
                if (__ballot(n > 0.0) == 0ull) {
#pragma unroll
                    for (int g = 0; g < G; ++g) sum[g] += poisson_term_nolog(n, acc[g][j]);
                } else if (__ballot(checked) == 0ull) {
#pragma unroll
                    for (int g = 0; g < G; ++g) sum[g] += poisson_term_fast(n, acc[g][j]);
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) sum[g] += poisson_term(n, acc[g][j]);
                }
            }
        } else {
            double pi[G][2], ai[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) { pi[g][0] = pi[g][1] = ai[g][0] = ai[g][1] = 0.0; }
#pragma unroll 8
            for (int k = 0; k < a.n1; ++k) {
                const double2 v = stream_load<NT>(a.ps + rowoff[a.n0 + k] + bin0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // (the reference's own order -- value = value + V * w, blueice/pdf_morphers.py:70 via scipy -- with
                    // separate multiply and add: P_i and a feed the root formula, whose sign tests see last bits)
                    const double c = coef[(a.n0 + k) * G + g];
                    pi[g][0] = __dadd_rn(pi[g][0], __dmul_rn(v.x, c));
                    pi[g][1] = __dadd_rn(pi[g][1], __dmul_rn(v.y, c));
                }
            }
#pragma unroll 8
            for (int k = 0; k < a.n2; ++k) {
                const double2 v = stream_load<NT>(a.nm + rowoff[a.n0 + a.n1 + k] + bin0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double c = coef[(a.n0 + a.n1 + k) * G + g];
                    ai[g][0] = __dadd_rn(ai[g][0], __dmul_rn(v.x, c));
                    ai[g][1] = __dadd_rn(ai[g][1], __dmul_rn(v.y, c));
                }
            }
            const double* __restrict__ aux = aux_base;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double p_cal = aux[g * 2 + 0];
                const double Ntot = aux[g * 2 + 1];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (bin0 + j < a.B) {
                        const double n = j ? nv.y : nv.x;
                        const double U = acc[g][j];
                        const double ab = ai[g][j];
                        // likelihood.py:645-646
                        const double w = pi[g][j] / ab * Ntot;
                        double r1, r2;
                        bb_roots(ab, w * p_cal, U, n, r1, r2);
                        // likelihood.py:649 asserts root1 <= 0 -- evaluated here in the reference's own operation order.
                        // (Where U_b == 0 that root is 0 analytically and its sign is decided by the last bit of the
                        // inputs; see DESIGN.md section 2 for what that means for parity.)
                        if (!(r1 <= 0.0)) flg[g] |= BI_ST_BB_ROOT1;
                        const double A = (U == 0.0) ? (n + ab) / (1.0 + p_cal) : r2;
                        if (!(0.0 <= A)) flg[g] |= BI_ST_BB_NEG;
                        const double mu = U + (A * w) * p_cal;
                        sum[g] += poisson_term(n, mu);
                    }
                }
            }
        }
    }

}

// MODE 2: as MODE 0 for the extended unbinned likelihood (rows hold pdf values at the events).
// MODE 3: as MODE 1 (value + gradient columns of ONE point) for the extended unbinned likelihood.
// MODE 0: G parameter points of one cell.  MODE 1 (gradient): ONE point; column 0 of the coefficient matrix
// gives mu, columns 1.. give d mu / d theta_j (theta = shape parameters, then rate scales), and the per-bin
// chain rule d ll / d theta_j = (n / mu - 1) * d mu / d theta_j is reduced alongside the likelihood.
template <int G, bool BB, bool NT, int MODE = 0>
__global__ __launch_bounds__(kThreads) void k_morph_reduce(LaunchArgs a) {
    const int item = blockIdx.y;
    const int NS = a.n0 + a.n1 + a.n2;
    const int64_t* __restrict__ rowoff = a.rowoff + (int64_t)item * NS;
    const double* __restrict__ coef = a.coef + (int64_t)item * NS * G;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item];
    const int n_tiles = a.item_tiles ? a.item_tiles[item] : a.n_tiles;

    double sum[G];
    unsigned flg[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { sum[g] = 0.0; flg[g] = 0u; }

    morph_tiles<G, BB, NT, MODE>(a, rowoff, coef, a.aux + (int64_t)item * G * 2, cnt, n_tiles, (int)blockIdx.x, (int)gridDim.x, sum, flg);

    __shared__ double s_sum[kThreads / 64][G];
    __shared__ unsigned s_flg[kThreads / 64][G];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double s = wave_sum(sum[g]);
        const unsigned f = BB ? wave_or(flg[g]) : 0u;
        if (lane == 0) { s_sum[wave][g] = s; s_flg[wave][g] = f; }
    }
    __syncthreads();
    const bool fuse = a.fin_mail != nullptr;
    const int nbx = gridDim.x;
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double s = s_sum[0][g];
        unsigned f = s_flg[0][g];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w][g]; f |= s_flg[w][g]; }
        const int64_t o = ((int64_t)item * nbx + blockIdx.x) * G + g;
        if (fuse) {
            if (BB) flags_post(a.fin_flags + (int64_t)item * G + g, f);
            mail_post_checked(a, a.fin_mail + o, s);
        } else {
            a.partial[o] = s;
            a.pflags[o] = f;
        }
    }
    if (!fuse || (int)blockIdx.x != nbx - 1) return;

    // ---- the item's last block does what k_finish would do, in k_finish's summation order ----
    const long long deadline = (long long)wall_clock64() + a.mail_timeout;
    double* __restrict__ mail = a.fin_mail + (int64_t)item * nbx * G;
    bool late = false;
    if (nbx <= 64) {
        // k_finish's 64-lane form: one wave per slot, lane b takes block b's partial
        for (int g = wave; g < G; g += kThreads / 64) {
            double s = lane < nbx ? mail_take(mail + (int64_t)lane * G + g, deadline, &late) : 0.0;
            s = wave_sum(s);
            const bool any_late = __ballot(late) != 0ull;
            const int64_t p = a.fin_perm[(int64_t)item * G + g];
            if (lane == 0) {
                const unsigned f = (BB ? flags_take(a.fin_flags + (int64_t)item * G + g) : 0u) | (any_late ? (unsigned)BI_ST_INTERNAL : 0u);
                if (p >= 0) {
                    a.fin_out[p] = s - a.fin_slot_lg[(int64_t)item * G + g];
                    if (a.fin_status) a.fin_status[p] |= (int32_t)f;
                }
            }
        }
        return;
    }
    // 256-lane form.  The item's slots are contiguous, [block][g]: thread t takes slots t, t + 256, ... -- always column
    // g = t % G, since G divides 256 -- four loads in flight at a time, then the threads of a column are added in a fixed
    // order (G = 1: k_finish's own order -- wave tree, then the four waves; G > 1: through LDS, thread by thread).
    {
        const int total = nbx * G;
        double s = 0.0;
        for (int q0 = threadIdx.x; q0 < total; q0 += 4 * kThreads) s = mail_take4(mail, q0, kThreads, total, s, deadline, &late);
        __shared__ double s_part[kThreads];
        __shared__ unsigned s_late[kThreads / 64];
        const unsigned lt = __ballot(late) != 0ull ? 1u : 0u;
        if constexpr (G == 1) s = wave_sum(s);
        __syncthreads();                           // (s_sum / s_flg above are done with)
        s_part[threadIdx.x] = s;
        if (lane == 0) s_late[wave] = lt;
        __syncthreads();
        if (threadIdx.x < G) {
            const int g = threadIdx.x;
            double t = 0.0;
            if constexpr (G == 1) {
                t = s_part[0];
#pragma unroll
                for (int w = 1; w < kThreads / 64; ++w) t += s_part[w * 64];
            } else {
                for (int j = 0; j < kThreads / G; ++j) t += s_part[g + G * j];
            }
            unsigned any_late = 0u;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) any_late |= s_late[w];
            const int64_t p = a.fin_perm[(int64_t)item * G + g];
            const unsigned f = (BB ? flags_take(a.fin_flags + (int64_t)item * G + g) : 0u) | (any_late ? (unsigned)BI_ST_INTERNAL : 0u);
            if (p >= 0) {
                a.fin_out[p] = t - a.fin_slot_lg[(int64_t)item * G + g];
                if (a.fin_status) a.fin_status[p] |= (int32_t)f;
            }
        }
    }
}

// ---- value + analytic gradient with Beeston-Barlow (bi_eval_grad, bb_source >= 0) ----------------------------------
// mu_b = U_b + A_b p_b with p_b = r_i P_b / a_b (likelihood.py:645-646: w p_cal = P/a N * r_i/N) and A_b the second root
// of the per-bin quadratic (likelihood.py:706-708) -- or, where U_b == 0 exactly, the reference's special case
// A_b = (n_b + a_b) / (1 + p_cal) with the SCALAR p_cal = r_i / N (likelihood.py:652-653).  Everything is smooth in
// (U, P, a, r_i, N), and those are linear in the coefficient columns, so the chain rule runs per bin:
//     d mu = dU + p dA + A dp,   dA = A_a da + A_p dp + A_U dU,   dp = (dr_i P + r_i dP) / a - p da / a.
// Column 0 of the coefficient matrices gives the values (U, P, a), column q >= 1 their derivatives with respect to
// parameter q: all G columns for the U streams, but only the first DZ = 1 + d (padded) for the P and a streams -- rate
// scales do not move the Beeston-Barlow source's template or its Monte-Carlo counts -- which is what keeps the
// accumulators in registers (C5: 11 + 5 + 5 column pairs instead of 3 x 11).  aux[q] = {d r_i, d N} for q >= 1,
// aux[0] = {p_cal, N}.  The value column follows the value kernel's operation order, so ll equals bi_eval's.
template <int G, int DZ, bool NT>
__global__ __launch_bounds__(kThreads) void k_morph_bbgrad(LaunchArgs a) {
    static_assert(DZ <= G, "shape columns are a prefix of all columns");
    const int item = blockIdx.y;
    const int NS = a.n0 + a.n1 + a.n2;
    const int64_t* __restrict__ rowoff = a.rowoff + (int64_t)item * NS;
    // coefficient block of an item: [n0][G] for the U streams, then [n1][DZ] and [n2][DZ]
    const int64_t coef_per_item = (int64_t)a.n0 * G + (int64_t)(a.n1 + a.n2) * DZ;
    const double* __restrict__ cU = a.coef + (int64_t)item * coef_per_item;
    const double* __restrict__ cP = cU + (int64_t)a.n0 * G;
    const double* __restrict__ cA = cP + (int64_t)a.n1 * DZ;
    const double* __restrict__ aux = a.aux + (int64_t)item * G * 2;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item];
    const int n_tiles = a.item_tiles ? a.item_tiles[item] : a.n_tiles;
    log_table_load();

    double sum[G];
    unsigned flg = 0u;
#pragma unroll
    for (int g = 0; g < G; ++g) sum[g] = 0.0;
    const double p_cal = aux[0], Ntot = aux[1];
    const double r_i = p_cal * Ntot;
    const int chunks = (a.chunks > 1 && n_tiles >= 64 * a.chunks) ? a.chunks : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double acc[G][2], pi[DZ][2], ai[DZ][2];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }
#pragma unroll
        for (int g = 0; g < DZ; ++g) { pi[g][0] = pi[g][1] = ai[g][0] = ai[g][1] = 0.0; }
#pragma unroll 4
        for (int k = 0; k < a.n0; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double c = cU[k * G + g];
                acc[g][0] = fma(c, v.x, acc[g][0]);
                acc[g][1] = fma(c, v.y, acc[g][1]);
            }
        }
#pragma unroll 4
        for (int k = 0; k < a.n1; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[a.n0 + k] + bin0);
            // the value column as the value kernel forms it (separate multiply and add, the reference's corner order)
            pi[0][0] = __dadd_rn(pi[0][0], __dmul_rn(v.x, cP[k * DZ]));
            pi[0][1] = __dadd_rn(pi[0][1], __dmul_rn(v.y, cP[k * DZ]));
#pragma unroll
            for (int g = 1; g < DZ; ++g) {
                const double c = cP[k * DZ + g];
                pi[g][0] = fma(c, v.x, pi[g][0]);
                pi[g][1] = fma(c, v.y, pi[g][1]);
            }
        }
#pragma unroll 4
        for (int k = 0; k < a.n2; ++k) {
            const double2 v = stream_load<NT>(a.nm + rowoff[a.n0 + a.n1 + k] + bin0);
            ai[0][0] = __dadd_rn(ai[0][0], __dmul_rn(v.x, cA[k * DZ]));
            ai[0][1] = __dadd_rn(ai[0][1], __dmul_rn(v.y, cA[k * DZ]));
#pragma unroll
            for (int g = 1; g < DZ; ++g) {
                const double c = cA[k * DZ + g];
                ai[g][0] = fma(c, v.x, ai[g][0]);
                ai[g][1] = fma(c, v.y, ai[g][1]);
            }
        }
        const double2 nv = *reinterpret_cast<const double2*>(cnt + bin0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (bin0 + j >= a.B) continue;
            const double n = j ? nv.y : nv.x;
            const double U = acc[0][j], P = pi[0][j], ab = ai[0][j];
            const double w = P / ab * Ntot;                    // likelihood.py:645-646
            const double p = w * p_cal;
            double r1, r2;
            bb_roots(ab, p, U, n, r1, r2);
            if (!(r1 <= 0.0)) flg |= BI_ST_BB_ROOT1;
            const bool special = U == 0.0;
            const double A = special ? (n + ab) / (1.0 + p_cal) : r2;
            if (!(0.0 <= A)) flg |= BI_ST_BB_NEG;
            const double mu = U + (A * w) * p_cal;
            sum[0] += poisson_term(n, mu);
            const double f = (n != 0.0 ? n / mu : 0.0) - 1.0;  // d term / d mu
            const double inv_a = 1.0 / ab;
            // partial derivatives of the second root
            double A_a = 0.0, A_p = 0.0, A_U = 0.0;
            const double t_sp = 1.0 / (1.0 + p_cal);
            if (!special) {
                const double p2 = p * p;
                const double disc = U * U * p2 + 2 * U * U * p + U * U + 2 * U * ab * p2 + 2 * U * ab * p - 2 * U * n * p2 -
                                    2 * U * n * p + ab * ab * p2 + 2 * ab * n * p2 + n * n * p2;
                const double inv_2sq = 0.5 / sqrt(disc);
                const double inv_den = 1.0 / (2 * p * (p + 1));
                const double D_a = 2 * U * p2 + 2 * U * p + 2 * ab * p2 + 2 * n * p2;
                const double D_p = 2 * U * U * p + 2 * U * U + 4 * U * ab * p + 2 * U * ab - 4 * U * n * p - 2 * U * n +
                                   2 * ab * ab * p + 4 * ab * n * p + 2 * n * n * p;
                const double D_U = 2 * U * p2 + 4 * U * p + 2 * U + 2 * ab * p2 + 2 * ab * p - 2 * n * p2 - 2 * n * p;
                A_a = (p + D_a * inv_2sq) * inv_den;
                A_U = (-p - 1.0 + D_U * inv_2sq) * inv_den;
                A_p = (-U + ab + n + D_p * inv_2sq) * inv_den - A * (4 * p + 2) * inv_den;
            }
#pragma unroll
            for (int g = 1; g < G; ++g) {
                const double dU = acc[g][j];
                const double dr = aux[g * 2 + 0];
                double dP = 0.0, da = 0.0;
                if (g < DZ) { dP = pi[g][j]; da = ai[g][j]; }
                const double dp = (dr * P + r_i * dP) * inv_a - p * da * inv_a;
                double dmu;
                if (!special) {
                    const double dA = A_a * da + A_p * dp + A_U * dU;
                    dmu = dU + p * dA + A * dp;
                } else {
                    const double dpc = (dr - p_cal * aux[g * 2 + 1]) / Ntot;
                    dmu = dp * (n + ab) * t_sp + p * da * t_sp - p * (n + ab) * t_sp * t_sp * dpc;
                }
                sum[g] += f * dmu;
            }
        }
    }

    __shared__ double s_sum[kThreads / 64][G];
    __shared__ unsigned s_flg[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double s = wave_sum(sum[g]);
        if (lane == 0) s_sum[wave][g] = s;
    }
    {
        const unsigned f = wave_or(flg);
        if (lane == 0) s_flg[wave] = f;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double s = s_sum[0][g];
        unsigned f = s_flg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w][g]; f |= s_flg[w]; }
        const int64_t o = ((int64_t)item * gridDim.x + blockIdx.x) * G + g;
        a.partial[o] = s;
        a.pflags[o] = g == 0 ? f : 0u;
    }
}

// ---- the single-point kernel: ONE launch from templates to scalar --------------------------------
// The call shape of `lf(**kwargs)` inside a minimizer.  The point's stream descriptors (row offsets and
// coefficients, <= kMaxSingleStreams of them) travel in the kernel-argument block, so the scalar loads hit the
// kernarg segment and no host-to-device copy precedes the launch; and the reduction is finished inside the
// launch: every block posts its partial into a mailbox slot and leaves, the last block in dispatch order collects
// them in block order (fixed order => bitwise reproducible; see mail_post) and writes {ll, status} straight into
// pinned host memory.
constexpr int kMaxSingleStreams = 128;

struct SingleDesc {
    int64_t rowoff[kMaxSingleStreams];
    double coef[kMaxSingleStreams];
    double aux[2];        // Beeston-Barlow: p_cal, N
    double slot_lg;       // constant subtracted from the sum (sum lgamma, empty-bin term, or sum of rates)
    unsigned* flags;      // one status word (Beeston-Barlow bits), zero on entry and on exit
    double* out;          // pinned host
    int32_t* status;      // pinned host
    unsigned long long* done;   // pinned host: receives `seq` after out / status (the host polls it)
    unsigned long long seq;
};

template <bool BB, bool NT, int MODE, bool FUSE>
__global__ __launch_bounds__(kThreads) void k_morph_single(LaunchArgs a, SingleDesc d) {
    double sum[1] = {0.0};
    unsigned flg[1] = {0u};
    morph_tiles<1, BB, NT, MODE>(a, d.rowoff, d.coef, d.aux, a.counts, a.n_tiles, (int)blockIdx.x, (int)gridDim.x, sum, flg);

    __shared__ double s_sum[kThreads / 64];
    __shared__ unsigned s_flg[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const double s = wave_sum(sum[0]);
        const unsigned f = BB ? wave_or(flg[0]) : 0u;
        if (lane == 0) { s_sum[wave] = s; s_flg[wave] = f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = s_sum[0];
        unsigned f = s_flg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w]; f |= s_flg[w]; }
        if constexpr (FUSE) {          // post into the mailbox and leave: the last block collects (see mail_post)
            if (BB) flags_post(d.flags, f);
            mail_post_checked(a, a.partial + blockIdx.x, s);
        } else {                       // a second, tiny launch sums the partials (k_finish_single)
            a.partial[blockIdx.x] = s;
            a.pflags[blockIdx.x] = f;
        }
    }
    if constexpr (!FUSE) return;
    if (blockIdx.x != gridDim.x - 1) return;
    // the last block in dispatch order: collect the partials of all blocks in block order
    const long long deadline = (long long)wall_clock64() + a.mail_timeout;
    bool late = false;
    double s = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 4 * kThreads) s = mail_take4(a.partial, b, kThreads, (int)gridDim.x, s, deadline, &late);
    s = wave_sum(s);
    const unsigned lt = __ballot(late) != 0ull ? 1u : 0u;
    __syncthreads();   // s_sum / s_flg are reused
    if (lane == 0) { s_sum[wave] = s; s_flg[wave] = lt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_sum[0];
        unsigned any_late = s_flg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { t += s_sum[w]; any_late |= s_flg[w]; }
        const unsigned ff = (BB ? flags_take(d.flags) : 0u) | (any_late ? (unsigned)BI_ST_INTERNAL : 0u);
        *d.out = t - d.slot_lg;
        *d.status = (int32_t)ff;
        __hip_atomic_store(d.done, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// second launch of the single-point path when the grid is too large for in-launch finishing to pay
// (an agent-scope release per block costs more than a kernel boundary once there are hundreds of blocks)
__global__ __launch_bounds__(kThreads) void k_finish_single(const double* __restrict__ partial,
                                                            const unsigned* __restrict__ pflags, int nbx, double slot_lg,
                                                            double* __restrict__ out, int32_t* __restrict__ status,
                                                            unsigned long long* done, unsigned long long seq) {
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    double s = 0.0;
    unsigned f = 0u;
#pragma unroll 8
    for (int b = threadIdx.x; b < nbx; b += kThreads) { s += partial[b]; f |= pflags[b]; }
    s = wave_sum(s);
    f = wave_or(f);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shf[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        unsigned ff = shf[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[w]; ff |= shf[w]; }
        *out = t - slot_lg;
        *status = (int32_t)ff;
        __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Sum the per-block partials of every (item, g) in a fixed order, subtract the dataset's
// sum lgamma(n+1), scatter to the caller's point order.  `lanes` (64 or 256) threads per slot.
__global__ __launch_bounds__(kThreads) void k_finish(const double* __restrict__ partial,
                                                     const unsigned* __restrict__ pflags, int nbx, int G, int lanes,
                                                     int64_t n_slots, const int64_t* __restrict__ perm,
                                                     const double* __restrict__ slot_lg, double* __restrict__ out,
                                                     int32_t* __restrict__ status) {
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    const int per_block = kThreads / lanes;
    const int64_t slot = (int64_t)blockIdx.x * per_block + threadIdx.x / lanes;
    const int l = threadIdx.x % lanes;
    const bool live = slot < n_slots;
    const int64_t item = live ? slot / G : 0;
    const int g = live ? (int)(slot % G) : 0;
    const int64_t p = live ? perm[slot] : -1;
    const double lg = live ? slot_lg[slot] : 0.0;
    double s = 0.0;
    unsigned f = 0u;
    if (p >= 0) {
#pragma unroll 8
        for (int b = l; b < nbx; b += lanes) {
            const int64_t o = (item * nbx + b) * G + g;
            s += partial[o];
            f |= pflags[o];
        }
    }
    s = wave_sum(s);
    f = wave_or(f);
    if (lanes == 64) {
        if ((threadIdx.x & 63) == 0 && p >= 0) {
            out[p] = s - lg;
            if (status) status[p] |= (int32_t)f;
        }
        return;
    }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shf[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0 && p >= 0) {
        double t = sh[0];
        unsigned ff = shf[0];
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[w]; ff |= shf[w]; }
        out[p] = t - lg;
        if (status) status[p] |= (int32_t)ff;
    }
}

// OR of the per-point status words of a plan -> one word (bi_plan_status)
__global__ __launch_bounds__(kThreads) void k_status_or(const int32_t* __restrict__ status, int64_t n, int32_t* __restrict__ out) {
    unsigned f = 0u;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) f |= (unsigned)status[i];
    f = wave_or(f);
    if ((threadIdx.x & 63) == 0 && f) atomicOr((unsigned*)out, f);
}

// a resident plan's status words are OR-ed into by every run: the "gave up" bit of one run must not outlive its report
__global__ __launch_bounds__(kThreads) void k_status_clear(int32_t* __restrict__ status, int64_t n, int32_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        if (status[i] & mask) status[i] &= ~mask;
}

__global__ void k_fill_value(double* __restrict__ out, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ void k_fill_const(double* __restrict__ out, const int64_t* __restrict__ idx, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[idx[i]] = v;
}

// sum_b lgamma(n_b + 1) over the valid counts of one dataset chunk -> partial[t][blk]
__global__ __launch_bounds__(kThreads) void k_counts_lgamma(const double* __restrict__ counts, int64_t B, int64_t Bp,
                                                            double* __restrict__ partial, int nblk) {
    const int t = blockIdx.y;
    const double* __restrict__ c = counts + (int64_t)t * Bp;
    double s = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B; b += (int64_t)nblk * kThreads) {
        const double n = c[b];
        if (n > 1.0 && n == floor(n)) s += lgamma(n + 1.0);
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) r += sh[w];
        partial[(int64_t)t * nblk + blockIdx.x] = r;
    }
}

__global__ void k_rows_sum(const double* __restrict__ partial, int nblk, double* __restrict__ out, int64_t T) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[t * nblk + b];
    out[t] = s;
}

// Compatibility morph: out[r][b] = sum_c V[row(c, r)][b] * w_c in the reference's corner order
// with separate multiply and add (scipy _evaluate_linear: `value = value + term`), i.e.
// bit-identical to the CPU path.  rows of `src` have stride Bp, rows of `out` stride B.
__global__ __launch_bounds__(kThreads) void k_morph_store(const double* __restrict__ src,
                                                          const int64_t* __restrict__ rowoff,  // [R][nc]
                                                          const double* __restrict__ w,        // [nc]
                                                          int nc, int64_t B, double* __restrict__ out,
                                                          const int32_t* __restrict__ place = nullptr) {
    const int r = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    double v = 0.0;
    for (int c = 0; c < nc; ++c) {
        const double term = __dmul_rn(src[rowoff[(int64_t)r * nc + c] + b], w[c]);
        v = __dadd_rn(v, term);
    }
    // place: column b of the tensor is the caller's column place[b] (unbinned data whose events were ordered by cell)
    out[(int64_t)r * B + (place ? (int64_t)place[b] : b)] = v;
}

// sum over bins of one padded row -> out[row]  (used for the Beeston-Barlow N table)
__global__ __launch_bounds__(kThreads) void k_row_total(const double* __restrict__ rows, int64_t B, int64_t Bp,
                                                        double* __restrict__ out) {
    const double* __restrict__ r = rows + (int64_t)blockIdx.x * Bp;
    double s = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += kThreads) s += r[b];
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w];
        out[blockIdx.x] = t;
    }
}

// The Beeston-Barlow normalisation N(z) = n_model_events[i].sum() (likelihood.py:645) in NUMPY'S summation order, so
// that single-point calls hand bb_roots the very bits the reference computes (the knife edge at
// U_b == 0).  np.sum over a contiguous array adds, in order, the pairwise sums of 8192-element chunks (its reduction
// buffer); a chunk's pairwise sum (numpy loops_utils.h.src, pairwise_sum) splits in halves down to blocks of 128, and a
// block is summed with 8 strided accumulators r_j = a[j] + a[8 + j] + ..., combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).
// One block per chunk: interpolate a(z) for the chunk in the reference's order (value = value + V*w), reproduce that
// tree.  The (shorter) last chunk is written out as values and summed on the host by the same rule (pairwise_sum_host).
constexpr int kSumChunk = 8192;

// blockIdx.x = point (fastest in dispatch order: the points of one cell read a chunk's rows back to back, out of L2),
// blockIdx.y = chunk.  rowoff / w: [points][nc]; chunk_sum: [points][n_full]; tail: [points][B % 8192].
__global__ __launch_bounds__(kThreads) void k_bb_chunk_sums(const double* __restrict__ nm, const int64_t* __restrict__ rowoff_all,
                                                            const double* __restrict__ w_all, int nc, int64_t B,
                                                            double* __restrict__ chunk_sum_all, double* __restrict__ tail_all) {
    __shared__ double a[kSumChunk / 2];
    __shared__ double r[256];
    __shared__ double leaf[64];
    const int64_t q = blockIdx.x;
    const int64_t* __restrict__ rowoff = rowoff_all + q * nc;
    const double* __restrict__ w = w_all + q * nc;
    double* __restrict__ chunk_sum = chunk_sum_all + q * (B / kSumChunk);
    double* __restrict__ tail = tail_all + q * (B % kSumChunk);
    const int64_t b0 = (int64_t)blockIdx.y * kSumChunk;
    const int n = (int)(B - b0 < (int64_t)kSumChunk ? B - b0 : (int64_t)kSumChunk);
    if (n < kSumChunk) {                              // the last, partial chunk: values out, the host sums them
        for (int i = threadIdx.x; i < n; i += kThreads) {
            double v = 0.0;
            for (int c = 0; c < nc; ++c) v = __dadd_rn(v, __dmul_rn(nm[rowoff[c] + b0 + i], w[c]));
            tail[i] = v;
        }
        return;
    }
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        for (int i = threadIdx.x; i < kSumChunk / 2; i += kThreads) {
            double v = 0.0;
            for (int c = 0; c < nc; ++c) v = __dadd_rn(v, __dmul_rn(nm[rowoff[c] + b0 + half * (kSumChunk / 2) + i], w[c]));
            a[i] = v;
        }
        __syncthreads();
        {   // 32 blocks of 128 x 8 accumulators = 256 (block, j) pairs, one per thread
            const int blk = threadIdx.x >> 3, j = threadIdx.x & 7;
            double t = a[blk * 128 + j];
#pragma unroll
            for (int i = 1; i < 16; ++i) t = __dadd_rn(t, a[blk * 128 + 8 * i + j]);
            r[threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const double* q = r + threadIdx.x * 8;
            leaf[half * 32 + threadIdx.x] = __dadd_rn(__dadd_rn(__dadd_rn(q[0], q[1]), __dadd_rn(q[2], q[3])),
                                                      __dadd_rn(__dadd_rn(q[4], q[5]), __dadd_rn(q[6], q[7])));
        }
    }
    __syncthreads();
    for (int stride = 1; stride < 64; stride <<= 1) {   // the halving recursion over 64 blocks = adjacent pairs, level by level
        if (threadIdx.x < 64 && (threadIdx.x % (2 * stride)) == 0) leaf[threadIdx.x] = __dadd_rn(leaf[threadIdx.x], leaf[threadIdx.x + stride]);
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_sum[blockIdx.y] = leaf[0];
}

// full_output with Beeston-Barlow (likelihood.py:634-658) on already-morphed templates:
// aw[b] = A_b * w_b and per-block partial sums of it.
__global__ __launch_bounds__(kThreads) void k_bb_full(const double* __restrict__ ps_m, const double* __restrict__ a_row,
                                                      const double* __restrict__ counts_row,
                                                      const double* __restrict__ mus, int S, int src, double p_cal,
                                                      double Ntot, int64_t B, double* __restrict__ aw,
                                                      double* __restrict__ partial) {
    double s = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B; b += (int64_t)gridDim.x * kThreads) {
        double U = 0.0;
        for (int k = 0; k < S; ++k) {
            const double e = __dmul_rn(ps_m[(int64_t)k * B + b], k == src ? 0.0 : mus[k]);
            U = k == 0 ? e : __dadd_rn(U, e);
        }
        const double ab = a_row[b];
        const double w = ps_m[(int64_t)src * B + b] / ab * Ntot;
        double r1, r2;
        bb_roots(ab, w * p_cal, U, counts_row[b], r1, r2);
        const double A = (U == 0.0) ? (counts_row[b] + ab) / (1.0 + p_cal) : r2;
        const double v = A * w;
        aw[b] = v;
        s += v;
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void k_bb_normalise(const double* __restrict__ aw, const double* __restrict__ tot, int64_t B,
                               double* __restrict__ row) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) row[b] = aw[b] / tot[0];
}

// per padded row: sum over bins, minimum, and a "has non-finite" flag -> out[row*3 + {0,1,2}]
__global__ __launch_bounds__(kThreads) void k_row_stats(const double* __restrict__ rows, int64_t B, int64_t Bp,
                                                        double* __restrict__ out) {
    const double* __restrict__ r = rows + (int64_t)blockIdx.x * Bp;
    double s = 0.0, mn = __builtin_inf(), bad = 0.0;
    {   // (four loads in flight per thread: with one, the 37 MB tensor of a 9400-event unbinned toy took 147 us)
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        int64_t b = threadIdx.x;
        for (; b + 3 * kThreads < B; b += 4 * kThreads) {
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = r[b + k * kThreads];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s4[k] += v[k];
                mn = fmin(mn, v[k]);
                if (!(fabs(v[k]) < __builtin_inf())) bad = 1.0;
            }
        }
        for (; b < B; b += kThreads) {
            const double v = r[b];
            s4[0] += v;
            mn = fmin(mn, v);
            if (!(fabs(v) < __builtin_inf())) bad = 1.0;
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    __shared__ double sh[3][kThreads / 64];
    s = wave_sum(s);
    bad = wave_sum(bad);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = mn; sh[2][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0][0], m = sh[1][0], f = sh[2][0];
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[0][w]; m = fmin(m, sh[1][w]); f += sh[2][w]; }
        out[(int64_t)blockIdx.x * 3 + 0] = t;
        out[(int64_t)blockIdx.x * 3 + 1] = m;
        out[(int64_t)blockIdx.x * 3 + 2] = f;
    }
}

// ---- non-empty-bin lists (CSR) of the datasets, built in bin order (deterministic) -----------
constexpr int kNzPerThread = 8;
constexpr int kNzChunk = kThreads * kNzPerThread;  // 2048 bins per block

__device__ __forceinline__ bool is_nz(double n) { return n != 0.0; }  // true for nan as well

__global__ __launch_bounds__(kThreads) void k_nz_count(const double* __restrict__ counts, int64_t B, int64_t Bp,
                                                       int32_t* __restrict__ cnt, int nchunks) {
    const double* __restrict__ c = counts + (int64_t)blockIdx.y * Bp;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    int k = 0;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j)
        if (b0 + j < B && is_nz(c[b0 + j])) ++k;
    __shared__ int sh[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(int64_t)blockIdx.y * nchunks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kThreads) void k_nz_scatter(const double* __restrict__ counts, int64_t B, int64_t Bp,
                                                         const int64_t* __restrict__ chunk_off, int nchunks,
                                                         int32_t* __restrict__ nz_idx, double* __restrict__ nz_n) {
    const double* __restrict__ c = counts + (int64_t)blockIdx.y * Bp;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    double v[kNzPerThread];
    int k = 0;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j) {
        v[j] = (b0 + j < B) ? c[b0 + j] : 0.0;
        if (is_nz(v[j])) ++k;
    }
    // exclusive prefix of k over the block, in thread order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __shared__ int sh[kThreads / 64];
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    int64_t pos = chunk_off[(int64_t)blockIdx.y * nchunks + blockIdx.x] + base + incl - k;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j)
        if (is_nz(v[j])) {
            nz_idx[pos] = (int32_t)(b0 + j);
            nz_n[pos] = v[j];
            ++pos;
        }
}

// compacted templates of one dataset: out[row][j] = rows[row][idx[j]] (0 beyond nnz)
__global__ __launch_bounds__(kThreads) void k_gather_rows(const double* __restrict__ rows, int64_t Bp,
                                                          const int32_t* __restrict__ idx, int64_t nnz, int64_t np,
                                                          double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (j >= np) return;
    const int64_t row = blockIdx.y;
    out[row * np + j] = j < nnz ? rows[row * Bp + idx[j]] : 0.0;
}

__global__ void k_pad_copy(const double* __restrict__ src, int64_t n, int64_t np, double* __restrict__ dst) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < np) dst[j] = j < n ? src[j] : 0.0;
}

// toy-MC, CSR form: for dataset t: sum_j xlogy(n_j, mu[idx_j]); one block per dataset
__global__ __launch_bounds__(kThreads) void k_dataset_dot_csr(const int32_t* __restrict__ nz_idx,
                                                              const double* __restrict__ nz_n,
                                                              const int64_t* __restrict__ nz_off,
                                                              const double* __restrict__ logmu, int64_t t0,
                                                              double* __restrict__ partial) {
    const int64_t t = t0 + blockIdx.x;
    const int64_t lo = nz_off[t], hi = nz_off[t + 1];
    double s = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += kThreads) {
        const double n = nz_n[j];
        double term = n * logmu[nz_idx[j]];
        if (n != n) term = __builtin_nan("");
        else if (n < 0.0 || n != floor(n)) term = -__builtin_inf();
        s += term;
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) r += sh[w];
        partial[blockIdx.x] = r;
    }
}

// ---- toy-MC, CSR form, bin tiles staged through LDS --------------------------------------------------------------
// The row kernel above gathers log mu[idx] from an 8 MB table with one 8-byte element per cache line: at C2 (10^4
// datasets x ~9 400 non-empty bins) that gather, not HBM, sets its 0.54 ms.  Here a block owns a TILE of kDotTile bins:
// it stages that part of log mu in LDS once and walks the entries of many datasets that fall into the tile -- a
// dataset's non-empty-bin list is sorted by bin, so they are one contiguous run, found through a per-dataset table
// of tile offsets (k_csr_tile_offsets, built once per data upload).  One wave per (dataset, tile); its sum goes to
// partial[dataset][tile], which k_dataset_finish adds up in tile order: fixed order, reproducible.
constexpr int kDotTile = 8192;      // bins per tile: 64 KB of LDS

__global__ __launch_bounds__(kThreads) void k_csr_tile_offsets(const int32_t* __restrict__ nz_idx, const int64_t* __restrict__ nz_off,
                                                               int n_tl, int32_t* __restrict__ tile_off /*[T][n_tl + 1]*/) {
    const int64_t t = blockIdx.x;
    const int64_t lo = nz_off[t], hi = nz_off[t + 1];
    int32_t* __restrict__ dst = tile_off + t * (n_tl + 1);
    if (hi == lo) {
        for (int tl = threadIdx.x; tl <= n_tl; tl += kThreads) dst[tl] = 0;
        return;
    }
    for (int64_t j = lo + threadIdx.x; j < hi; j += kThreads) {
        const int cur = nz_idx[j] / kDotTile;
        const int prev = j > lo ? nz_idx[j - 1] / kDotTile : -1;
        for (int tl = prev + 1; tl <= cur; ++tl) dst[tl] = (int32_t)(j - lo);     // first entry at or beyond the start of tile tl
        if (j == hi - 1)
            for (int tl = cur + 1; tl <= n_tl; ++tl) dst[tl] = (int32_t)(hi - lo);
    }
}

// sum of a double over the 16 lanes of a DPP row (lanes 16r .. 16r+15): four rotate-and-add steps on the cross-lane
// data path; every lane of the row ends up with the total
__device__ __forceinline__ double row16_sum(double v) {
#define BI_ROR_ADD(N)                                                                                              \
    do {                                                                                                           \
        const unsigned long long u = __double_as_longlong(v);                                                      \
        const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x120 + N, 0xF, 0xF, true);           \
        const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x120 + N, 0xF, 0xF, true);   \
        v += __longlong_as_double(((unsigned long long)hi << 32) | lo);                                            \
    } while (0)
    BI_ROR_ADD(8);
    BI_ROR_ADD(4);
    BI_ROR_ADD(2);
    BI_ROR_ADD(1);
#undef BI_ROR_ADD
    return v;
}

// Tile-major copy of the non-empty-bin lists: all entries of bin tile 0 (dataset 0, 1, ...), then tile 1, ... -- the
// block that owns a tile reads ONE contiguous stream instead of a sub-kilobyte run per dataset scattered over the
// dataset-major lists (which held the first tiled version at 2.6 TB/s).  An entry is 4 bytes: the byte offset of the bin
// within the staged tile (bits 3..16: bin * 8) and the count (15 bits, from bit 17), so that the kernel gets the LDS
// address with one AND and the count with one shift; data whose counts do not fit (or are not positive integers) keep
// the row kernel.  Every (dataset, tile) run is padded to a multiple of FOUR entries with kTmPadEntry -- count 0, and the
// offset of an extra LDS slot behind the tile that holds 0.0, never log mu = -inf -- so that a lane's 16-byte load is
// wholly inside its run or wholly outside it (one test per load, not per entry) and every load is 16-byte aligned.
// Round 4, measured: with the entry loads taken out the kernel runs in 24 us instead of 86 -- the ENTRY STREAM is its time
// (376 MB at C2: 54 us of HBM at best), neither the LDS gathers nor the arithmetic.  So where every count is at most 7 -- toys
// of sparse expectations: all of them -- an entry is TWO bytes: bin * 8 (the LDS byte offset, bits 3..15) | count (bits
// 0..2), eight entries per 16-byte load, runs padded to multiples of eight with 0 (count 0: the term is dropped by a select,
// there is no room for the offset of the extra slot).  Data with a larger count anywhere keeps the 4-byte entries.
constexpr uint32_t kTmPadEntry = (uint32_t)kDotTile << 3;
__global__ void k_tm_counts(const int32_t* __restrict__ tile_off, int64_t T, int n_tl, int64_t* __restrict__ cnt /*[n_tl * T + 1]*/,
                            int group /* entries per 16-byte load: 4 or 8 */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > (int64_t)n_tl * T) return;
    if (i == (int64_t)n_tl * T) { cnt[i] = 0; return; }
    const int64_t tl = i / T, t = i % T;
    const int32_t* __restrict__ o = tile_off + t * (n_tl + 1) + tl;
    cnt[i] = (o[1] - o[0] + group - 1) & ~(group - 1);     // (the run's padded length: the list is pre-filled with padding entries)
}

template <typename ENTRY>
__global__ __launch_bounds__(kThreads) void k_tm_scatter(const int32_t* __restrict__ nz_idx, const double* __restrict__ nz_n,
                                                         const int64_t* __restrict__ nz_off, const int32_t* __restrict__ tile_off,
                                                         int64_t T, int n_tl, const int64_t* __restrict__ tm_off,
                                                         ENTRY* __restrict__ tm_entries, int* __restrict__ bad) {
    const int64_t t = blockIdx.x;
    const int64_t lo = nz_off[t], hi = nz_off[t + 1];
    const int32_t* __restrict__ o = tile_off + t * (n_tl + 1);
    bool any_bad = false;
    for (int64_t j = lo + threadIdx.x; j < hi; j += kThreads) {
        const int idx = nz_idx[j];
        const double n = nz_n[j];
        const int tl = idx / kDotTile;
        if constexpr (sizeof(ENTRY) == 2) {
            if (!(n >= 1.0 && n <= 7.0 && n == floor(n))) any_bad = true;
            tm_entries[tm_off[(int64_t)tl * T + t] + (j - lo - o[tl])] = (ENTRY)(((uint32_t)(idx - tl * kDotTile) << 3) | ((uint32_t)n & 7u));
        } else {
            if (!(n >= 1.0 && n < 32768.0 && n == floor(n))) any_bad = true;
            tm_entries[tm_off[(int64_t)tl * T + t] + (j - lo - o[tl])] = ((uint32_t)(idx - tl * kDotTile) << 3) | ((uint32_t)n << 17);
        }
    }
    if (any_bad) atomicOr(bad, 1);
}

// 1024 threads: 16 waves share one staged tile; every 16-lane row of a wave takes its own dataset, so a wave has four
// (dataset, tile) runs in flight and a block 64.  The runs are short (~80 entries at C2) and each starts with a chain of
// dependent loads (offsets -> entries -> LDS): with one run after the other per row the kernel sat at 134 us for 0.4 GB
// (3 TB/s), every wave waiting a full memory latency two or three times per run.  Now a row keeps the loads of THREE
// runs in flight (ring buffers in registers, the loop unrolled over the ring so that nothing is copied and nothing
// newer than what it needs is waited for): offsets four runs ahead, entries two runs ahead.  Every load is
// unconditional -- a run behind the row's last one is the last one again (requested, never stored), an entry load
// reads up to kDotPad entries past its run (the lists are padded by that much) and its 4-entry sum is dropped afterwards
// (runs are whole 16-byte groups: round 4) -- because a branch around a load makes the compiler wait for ALL outstanding
// loads.  Index arithmetic is 32-bit, relative to the
// block's first entry (the launch keeps a block below 2^31 entries).  Per lane the entries are added in ascending
// order with fma, four at a time (one 16-byte load), the groups' sums then in ascending order: a fixed order, independent of
// the launch geometry.
typedef uint32_t bi_uint4 __attribute__((ext_vector_type(4)));
constexpr int kDotThreads = 1024;
#ifndef BI_DOT_DEPTH
#define BI_DOT_DEPTH 2
#endif
constexpr int kDotPad = 64 * 2 + 16;           // entries behind the lists that the read-ahead may touch (read, masked, never used): the largest of the variants below

// sum over the L lanes of a group inside a 16-lane DPP row (L = 16, or 8: groups start at lanes 0 and 8) on the cross-lane
// data path; every lane of the group ends up with the total.  L = 16: four rotate-and-add steps; L = 8: the mirror image
// within the half row (lane i <-> 7 - i), then the two exchanges within quads -- three steps, in a fixed order.
template <int L>
__device__ __forceinline__ double row_group_sum(double v) {
#define BI_DPP_ADD(CTRL)                                                                                           \
    do {                                                                                                           \
        const unsigned long long u = __double_as_longlong(v);                                                      \
        const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);                \
        const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);        \
        v += __longlong_as_double(((unsigned long long)hi << 32) | lo);                                            \
    } while (0)
    if constexpr (L == 16) {
        BI_DPP_ADD(0x120 + 8);        // row_ror:8
        BI_DPP_ADD(0x120 + 4);
        BI_DPP_ADD(0x120 + 2);
        BI_DPP_ADD(0x120 + 1);
    } else if constexpr (L == 8) {
        BI_DPP_ADD(0x141);            // row_half_mirror
        BI_DPP_ADD(0xB1);             // quad_perm:[1,0,3,2]
        BI_DPP_ADD(0x4E);             // quad_perm:[2,3,0,1]
    } else {
        static_assert(L == 4, "groups of 4, 8 or 16 lanes");
        BI_DPP_ADD(0xB1);             // quad_perm:[1,0,3,2]
        BI_DPP_ADD(0x4E);             // quad_perm:[2,3,0,1]
    }
#undef BI_DPP_ADD
    return v;
}

// L lanes per (dataset, tile) run, AHEAD 16-byte loads per lane requested up front: 4 L AHEAD entry slots per run.
// <16, 2>: 128 slots, 64 runs in flight per block (round 3).  <8, 3>: 96 slots -- a run at C2 is ~76 entries, so 79 % of the
// slots carry an entry instead of 59 % --, 128 runs in flight per block, three rotate-and-add steps instead of four (round 4).
// W: bytes per entry (4, or 2: counts up to 7) -- 16 / W entries per 16-byte load, (16 / W) L AHEAD slots per run.
template <int L, int AHEAD, int W = 4>
__global__ __launch_bounds__(kDotThreads) void k_dataset_dot_tiled(const void* __restrict__ tm_entries_v,
                                                                   const int64_t* __restrict__ tm_off, int64_t T, int n_tl,
                                                                   const double* __restrict__ logmu, int64_t B, int64_t t0,
                                                                   int64_t n, double* __restrict__ partial /*[n_tl][n]*/) {
    typedef typename std::conditional<W == 2, uint16_t, uint32_t>::type entry_t;
    constexpr int EPL = 16 / W;                                        // entries per lane and load
    static_assert(EPL * L * AHEAD + 16 <= kDotPad, "the lists' padding must cover the read-ahead");
    const entry_t* __restrict__ tm_entries = static_cast<const entry_t*>(tm_entries_v);
    __shared__ double s_mu[kDotTile + 1];                              // (+ the slot of the padding entries: 0.0)
    const int tl = blockIdx.x;
    const int64_t bin0 = (int64_t)tl * kDotTile;
    const int row = threadIdx.x / L, gl = threadIdx.x % L;             // kDotThreads / L rows of L lanes
    const int per = (int)((n + gridDim.y - 1) / gridDim.y);
    const int c0u = (int)blockIdx.y * per, c1 = min((int)n, c0u + per);
    const int c0 = min(c0u, (int)n - 1);                               // (clamped: a block without datasets still loads validly)
    const int64_t* __restrict__ off = tm_off + (int64_t)tl * T + t0;
    const int64_t base = off[c0];                                  // block-uniform: the entries of this block start here
    const entry_t* __restrict__ ent = tm_entries + base;
    const uint32_t* __restrict__ off32 = reinterpret_cast<const uint32_t*>(off);   // low words: all a block-relative index needs
    const uint32_t base32 = (uint32_t)base;
    constexpr int kStep = kDotThreads / L, kAhead = AHEAD, kDepth = BI_DOT_DEPTH, kRing = 2 * (kDepth + 1);   // entries kDepth runs ahead, offsets 2 kDepth; the ring a multiple of the entry buffers
    const int q_last = max(c0, c1 - 1);
    // (a step only ISSUES loads into the rings; whatever touches a loaded value -- the subtraction of the base, the mask
    //  of the entries past the run's end -- happens in the step that consumes it, two or four steps later: the compiler
    //  waits where a value is first used, and it does not move that use out of the step it was written in)
    auto load_offsets = [&](int q, uint32_t& ra, uint32_t& rb) {
        const int qc = min(q, q_last);
        ra = off32[2 * qc];
        rb = off32[2 * qc + 2];
    };
    // (a lane takes EPL consecutive entries per load: a group's load covers 16 L contiguous bytes of its run)
    auto load_entries = [&](uint32_t ra, bi_uint4 (&e)[kAhead]) {
        const entry_t* __restrict__ p = ent + (int)(ra - base32) + EPL * gl;
#pragma unroll
        for (int k = 0; k < kAhead; ++k) __builtin_memcpy(&e[k], p + EPL * L * k, 16);   // (16-byte load, 16-byte aligned: runs are whole groups)
    };
    bi_uint4 E[kDepth + 1][kAhead];
    uint32_t RA[kRing], RB[kRing];
    const int q0 = c0 + row;
    // the pipeline's first requests go out before the tile is staged: their latency passes under the staging
#pragma unroll
    for (int u = 0; u < 2 * kDepth; ++u) load_offsets(q0 + u * kStep, RA[u], RB[u]);
    {   // (all loads first, addresses clamped instead of predicated: eight loads in flight per thread, not one at a time)
        double v[kDotTile / kDotThreads];
#pragma unroll
        for (int k = 0; k < kDotTile / kDotThreads; ++k) v[k] = logmu[min(bin0 + threadIdx.x + k * kDotThreads, B - 1)];
#pragma unroll
        for (int k = 0; k < kDotTile / kDotThreads; ++k)
            s_mu[threadIdx.x + k * kDotThreads] = bin0 + threadIdx.x + k * kDotThreads < B ? v[k] : 0.0;
        if (threadIdx.x == 0) s_mu[kDotTile] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < kDepth; ++u) load_entries(RA[u], E[u]);
    __syncthreads();
    if (c0u >= c1) return;
    const int n_iter = (c1 - c0 + kStep - 1) / kStep;             // block-uniform: rows past their last run idle along
    for (int it = 0; it < n_iter; it += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            if (it + u < n_iter) {                               // (scalar condition)
                const int q = q0 + (it + u) * kStep;
                load_offsets(q + 2 * kDepth * kStep, RA[(u + 2 * kDepth) % kRing], RB[(u + 2 * kDepth) % kRing]);
                load_entries(RA[(u + kDepth) % kRing], E[(u + kDepth) % (kDepth + 1)]);
                __builtin_amdgcn_sched_barrier(0);               // the requests go out BEFORE this step's arithmetic, not after it
                const bi_uint4 (&e)[kAhead] = E[u % (kDepth + 1)];
                const int a = (int)(RA[u % kRing] - base32), b = (int)(RB[u % kRing] - base32);
                const int len = b - a - EPL * gl;                // (padded) entries of the run from this lane's first on
                double s = 0.0;
                if constexpr (W == 4) {
                    double lm[4 * kAhead];
#define BI_TM_LOG(x) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_mu) + ((x) & 0x1FFF8u)))
#pragma unroll
                    for (int k = 0; k < 4 * kAhead; ++k) lm[k] = BI_TM_LOG(e[k >> 2][k & 3]);      // LDS reads in flight together
#pragma unroll
                    for (int g = 0; g < kAhead; ++g) {
                        double sg = (double)(e[g][0] >> 17) * lm[4 * g];
#pragma unroll
                        for (int k = 1; k < 4; ++k) sg = __builtin_fma((double)(e[g][k] >> 17), lm[4 * g + k], sg);
                        s += 4 * L * g < len ? sg : 0.0;         // (a load behind the run's end read other runs' entries: dropped whole)
                    }
                    for (int j = a + gl + 4 * L * kAhead; j < b; j += L) {                  // (runs beyond 4 L AHEAD entries)
                        const uint32_t x = ent[j];
                        s = __builtin_fma((double)(x >> 17), BI_TM_LOG(x), s);
                    }
#undef BI_TM_LOG
                } else {
                    // two entries per word: LDS byte offset = entry & 0xFFF8, count = entry & 7 (0 = padding: its term is dropped,
                    // whatever log mu of bin 0 is)
                    double lm[8 * kAhead];
#define BI_TM_LOG16(x) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_mu) + ((x) & 0xFFF8u)))
#pragma unroll
                    for (int k = 0; k < 4 * kAhead; ++k) {
                        const uint32_t x = e[k >> 2][k & 3];
                        lm[2 * k] = BI_TM_LOG16(x);
                        lm[2 * k + 1] = BI_TM_LOG16(x >> 16);
                    }
#pragma unroll
                    for (int g = 0; g < kAhead; ++g) {
                        double sg = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t x = e[g][k];
                            const uint32_t n0 = x & 7u, n1 = (x >> 16) & 7u;
                            sg = __builtin_fma((double)n0, n0 ? lm[8 * g + 2 * k] : 0.0, sg);
                            sg = __builtin_fma((double)n1, n1 ? lm[8 * g + 2 * k + 1] : 0.0, sg);
                        }
                        s += 8 * L * g < len ? sg : 0.0;         // (a load behind the run's end read other runs' entries: dropped whole)
                    }
                    for (int j = a + gl + 8 * L * kAhead; j < b; j += L) {                  // (runs beyond 8 L AHEAD entries)
                        const uint32_t x = ent[j];
                        const uint32_t n0 = x & 7u;
                        s = __builtin_fma((double)n0, n0 ? BI_TM_LOG16(x) : 0.0, s);
                    }
#undef BI_TM_LOG16
                }
                s = row_group_sum<L>(s);
                if (gl == 0 && q < c1) partial[(int64_t)tl * n + q] = s;
            }
        }
    }
}

// ---- toy-MC generation on the device ---------------------------------------------------------
// n_{t,b} ~ Poisson(mu_b): the binned equivalent of Model.simulate (blueice/model.py:69-91: Poisson number of
// events per source, each drawn from the source's pdf) followed by set_data's binning (likelihood.py:603-609).
// Counter-based Philox4x32-10 keyed by the seed, counter = (bin, dataset, attempt): every (dataset, bin) draw
// is independent of launch geometry and can be regenerated, which is what lets the two-pass CSR build
// (count, then scatter) see the same numbers twice.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {  // uniform on [0, 1) with 53 random bits
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// lam < 10: inversion by sequential search with one 53-bit uniform; p0 = exp(-lam) comes from a per-bin table (it is
// the same for every toy)
__device__ __forceinline__ double poisson_small(double lam, double p0, double u) {
    double p = p0, F = p, n = 0.0;
    while (u > F && n < 1000.0) {
        n += 1.0;
        p *= lam / n;
        F += p;
    }
    return n;
}

// lam >= 10: PTRS, Hoermann (1993): transformed rejection with squeeze, as in numpy's random_poisson_ptrs
__device__ double poisson_ptrs(double lam, uint64_t seed, int64_t t, int64_t b) {
    uint32_t r[4];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const double slam = sqrt(lam), loglam = log(lam);
    const double bb = 0.931 + 2.53 * slam, aa = -0.059 + 0.02483 * bb;
    const double invalpha = 1.1239 + 1.1328 / (bb - 3.4), vr = 0.9277 - 3.6224 / (bb - 2.0);
    for (uint32_t attempt = 0; attempt < 4096u; ++attempt) {
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)t, ((uint32_t)(t >> 32) & 0xFFFFu) | ((attempt + 1u) << 16), k0, k1, r);
        const double U = u53(r[0], r[1]) - 0.5, V = u53(r[2], r[3]);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * aa / us + bb) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(aa / (us * us) + bb) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
    }
    return floor(lam);  // unreachable in practice (acceptance > 0.9 per attempt)
}

// The draws of bins b (even) and b + 1 of toy t: one Philox block gives both their uniforms.
__device__ __forceinline__ void poisson_draw_pair(const double* __restrict__ mu, const double* __restrict__ p0, int64_t B,
                                                  uint64_t seed, int64_t t, int64_t b, double& n0, double& n1) {
    n0 = n1 = 0.0;
    if (b >= B) return;
    const double2 lam = *reinterpret_cast<const double2*>(mu + b);      // rows are padded to an even length
    const double2 e = *reinterpret_cast<const double2*>(p0 + b);
    const bool has1 = b + 1 < B;
    const bool small0 = lam.x > 0.0 && lam.x < 10.0, small1 = has1 && lam.y > 0.0 && lam.y < 10.0;
    if (small0 || small1) {
        uint32_t r[4];
        const int64_t pair = b >> 1;
        philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, (uint32_t)seed,
                      (uint32_t)(seed >> 32), r);
        if (small0) n0 = poisson_small(lam.x, e.x, u53(r[0], r[1]));
        if (small1) n1 = poisson_small(lam.y, e.y, u53(r[2], r[3]));
    }
    if (lam.x >= 10.0) n0 = poisson_ptrs(lam.x, seed, t, b);            // mu = 0 or invalid -> no events
    if (has1 && lam.y >= 10.0) n1 = poisson_ptrs(lam.y, seed, t, b + 1);
}

// p0[b] = exp(-mu[b]): once per bin, shared by all toys
__global__ void k_exp_neg(const double* __restrict__ mu, int64_t n, double* __restrict__ p0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p0[i] = exp(-mu[i]);
}

__global__ __launch_bounds__(kThreads) void k_toy_count(const double* __restrict__ mu, const double* __restrict__ p0, int64_t B,
                                                        uint64_t seed, int64_t t0, int32_t* __restrict__ cnt, int nchunks) {
    const int64_t t = t0 + blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    int k = 0;
#pragma unroll 1
    for (int j = 0; j < kNzPerThread; j += 2) {
        double n0, n1;
        poisson_draw_pair(mu, p0, B, seed, t, b0 + j, n0, n1);
        k += (n0 != 0.0) + (n1 != 0.0);
    }
    __shared__ int sh[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(int64_t)blockIdx.y * nchunks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kThreads) void k_toy_scatter(const double* __restrict__ mu, const double* __restrict__ p0, int64_t B,
                                                          uint64_t seed, int64_t t0,
                                                          const int64_t* __restrict__ chunk_off, int nchunks,
                                                          int32_t* __restrict__ nz_idx, double* __restrict__ nz_n,
                                                          double* __restrict__ lg_partial) {
    const int64_t t = t0 + blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    double v[kNzPerThread];
    int k = 0;
    double lg = 0.0;
#pragma unroll
    for (int j = 0; j < kNzPerThread; j += 2) {
        poisson_draw_pair(mu, p0, B, seed, t, b0 + j, v[j], v[j + 1]);
#pragma unroll
        for (int q = j; q < j + 2; ++q)
            if (v[q] != 0.0) { ++k; if (v[q] > 1.0) lg += lgamma(v[q] + 1.0); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int q = __shfl_up(incl, off, 64);
        if (lane >= off) incl += q;
    }
    __shared__ int sh[kThreads / 64];
    __shared__ double shl[kThreads / 64];
    lg = wave_sum(lg);
    if (lane == 63) sh[wave] = incl;
    if (lane == 0) shl[wave] = lg;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    int64_t pos = chunk_off[(int64_t)blockIdx.y * nchunks + blockIdx.x] + base + incl - k;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j)
        if (v[j] != 0.0) {
            nz_idx[pos] = (int32_t)(b0 + j);
            nz_n[pos] = v[j];
            ++pos;
        }
    if (threadIdx.x == 0) lg_partial[(int64_t)blockIdx.y * nchunks + blockIdx.x] = shl[0] + shl[1] + shl[2] + shl[3];
}

// ---- toy-MC generation, event by event (sparse expectations) ----------------------------------------------------
// Independent n_b ~ Poisson(mu_b) is the same law as N ~ Poisson(M = sum_b mu_b) events thrown onto the bins with
// probabilities mu_b / M.  Where M << B (10^4 expected events in 10^6 bins at C2) that is 100 times fewer random numbers
// than one draw per bin: one block per toy draws N, finds every event's bin by bisection in the cumulative sums of mu
// (8 MB, L2), sorts the bin numbers in LDS (bitonic, up to 32 768 keys = 128 KB of the CU's 160 KB) and run-length
// encodes them into the non-empty-bin list of the toy -- sorted by bin, as the per-bin generator leaves it.  Two passes
// with the same counters (Philox4x32-10 keyed by the seed; counter = (event, toy)): the first only counts the non-empty
// bins of every toy, the second writes them behind the offsets a scan made of the counts.
constexpr int kEvThreads = 512;
constexpr uint32_t kEvTag = 0x45564E54u;        // separates the event counters from the per-bin ones

__device__ __forceinline__ int toy_event_count(double M, uint64_t seed, int64_t t) {
    if (M >= 10.0) return (int)poisson_ptrs(M, seed, t, ((int64_t)1 << 40) + 7);          // a "bin" no model has
    uint32_t r[4];
    philox4x32_10(0xFFFFFFFFu, kEvTag, (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (int)poisson_small(M, exp(-M), u53(r[0], r[1]));
}

// events per toy (an upper bound of its non-empty bins: the room it gets in the provisional lists)
__global__ void k_toy_event_counts(double M, uint64_t seed, int64_t t0, int64_t T, int npow2, int64_t* __restrict__ n_ev,
                                   int* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > T) return;
    if (i == T) { n_ev[i] = 0; return; }
    const int n = toy_event_count(M, seed, t0 + i);
    if (n > npow2) atomicOr(overflow, 1);
    n_ev[i] = min(n, npow2);
}

__global__ __launch_bounds__(kEvThreads) void k_toy_events(const double* __restrict__ cdf, int64_t B, double M, uint64_t seed,
                                                           int64_t t0, int npow2, const int64_t* __restrict__ room_off,
                                                           int32_t* __restrict__ idx_out, double* __restrict__ n_out,
                                                           int64_t* __restrict__ nnz_out, double* __restrict__ lgsum) {
    // LDS: radix sort (npow2 <= 16384): two key buffers of npow2 and 16 x 512 counters; bitonic sort: one key buffer;
    // then kEvThreads ints and doubles of scratch
    extern __shared__ uint32_t s_keys[];
    const bool radix = npow2 <= 16384;
    const int n_alloc = npow2;
    int key_bits = 1;
    while (key_bits < 32 && ((int64_t)1 << key_bits) < B) ++key_bits;
    int* s_scan = reinterpret_cast<int*>(s_keys + (radix ? 2 * n_alloc + 16 * kEvThreads / 2 : n_alloc));
    double* s_lg = reinterpret_cast<double*>(s_scan + kEvThreads);
    __shared__ int s_N;
    const int64_t t = t0 + blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) s_N = min(toy_event_count(M, seed, t), npow2);
    __syncthreads();
    const int N = s_N;
    int n2 = 1024;                                             // the power of two this toy needs (the sort is the cost)
    while (n2 < N) n2 <<= 1;
    for (int e = tid; e < n2; e += kEvThreads) {
        uint32_t key = 0xFFFFFFFFu;
        if (e < N) {
            uint32_t r[4];
            philox4x32_10((uint32_t)e, kEvTag, (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            const double target = u53(r[0], r[1]) * M;
            int64_t lo = 0, hi = B;                                         // first bin whose cumulative sum exceeds the target
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (cdf[mid] <= target) lo = mid + 1; else hi = mid;
            }
            key = (uint32_t)min(lo, B - 1);
        }
        s_keys[e] = key;
    }
    __syncthreads();
    if (radix) {
        // LSD radix sort, 4 bits per pass, in LDS: keys ping-pong between two buffers; every thread owns a contiguous chunk
        // and a column of 16 counters, so counting and scattering need no atomics and the sort is stable.  ~100 LDS
        // accesses per thread and pass against the bitonic network's 105 stages over all keys (10^4 toys of C2: 21 -> 5 ms).
        uint32_t* src = s_keys;
        uint32_t* dst = s_keys + n_alloc;
        uint16_t* cnt = reinterpret_cast<uint16_t*>(s_keys + 2 * n_alloc);        // [16][kEvThreads]
        const int chunk = n2 / kEvThreads, c0 = tid * chunk;
        for (int shift = 0; shift < key_bits; shift += 4) {
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[d * kEvThreads + tid] = 0;
            for (int i = 0; i < chunk; ++i) ++cnt[((src[c0 + i] >> shift) & 15u) * kEvThreads + tid];
            __syncthreads();
            // exclusive scan of the 16 x 512 counters in (digit, thread) order: thread t takes elements 16 t .. 16 t + 15
            unsigned local[16], sum = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { local[q] = sum; sum += cnt[tid * 16 + q]; }
            unsigned incl = sum;
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) s_scan[wave] = (int)incl;
            __syncthreads();
            unsigned base = incl - sum;
            for (int w = 0; w < wave; ++w) base += (unsigned)s_scan[w];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) cnt[tid * 16 + q] = (uint16_t)(base + local[q]);
            __syncthreads();
            for (int i = 0; i < chunk; ++i) {
                const uint32_t key = src[c0 + i];
                uint16_t& slot = cnt[((key >> shift) & 15u) * kEvThreads + tid];
                dst[slot] = key;
                ++slot;
            }
            __syncthreads();
            uint32_t* t2 = src; src = dst; dst = t2;
        }
        if (src != s_keys) {                                   // an odd number of passes: bring the result home
            for (int i = tid; i < n2; i += kEvThreads) s_keys[i] = src[i];
            __syncthreads();
        }
    } else {
        for (int k = 2; k <= n2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2; i += kEvThreads) {
                    const int x = i ^ j;
                    if (x > i) {
                        const uint32_t a = s_keys[i], b = s_keys[x];
                        if ((a > b) == ((i & k) == 0)) { s_keys[i] = b; s_keys[x] = a; }
                    }
                }
                __syncthreads();
            }
    }
    // run-length encoding: every thread owns a contiguous segment of the sorted keys
    const int seg = n2 / kEvThreads;
    const int a0 = tid * seg, a1 = a0 + seg;
    int heads = 0;
    for (int i = a0; i < a1 && i < N; ++i) heads += (i == 0 || s_keys[i] != s_keys[i - 1]);
    s_scan[tid] = heads;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < tid; ++q) base += s_scan[q];                        // (512 additions per thread: nothing beside the sort)
    int64_t pos = room_off[blockIdx.x] + base;
    double lg = 0.0;
    for (int i = a0; i < a1 && i < N; ++i) {
        if (i != 0 && s_keys[i] == s_keys[i - 1]) continue;
        int j = i + 1;
        while (j < N && s_keys[j] == s_keys[i]) ++j;
        const double n = (double)(j - i);
        idx_out[pos] = (int32_t)s_keys[i];
        n_out[pos] = n;
        ++pos;
        if (n > 1.0) lg += lgamma(n + 1.0);
    }
    s_lg[tid] = lg;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int q = 0; q < kEvThreads; ++q) tot += s_lg[q];
        lgsum[blockIdx.x] = tot;
        int nn = 0;
        for (int q = 0; q < kEvThreads; ++q) nn += s_scan[q];
        nnz_out[blockIdx.x] = nn;
    }
}

// the provisional lists (room for one entry per EVENT) packed into the final ones (one entry per non-empty bin)
__global__ __launch_bounds__(kThreads) void k_toy_pack(const int64_t* __restrict__ room_off, const int64_t* __restrict__ nz_off,
                                                       const int32_t* __restrict__ idx_in, const double* __restrict__ n_in,
                                                       int32_t* __restrict__ nz_idx, double* __restrict__ nz_n) {
    const int64_t t = blockIdx.x;
    const int64_t src = room_off[t], dst = nz_off[t], n = nz_off[t + 1] - dst;
    for (int64_t j = threadIdx.x; j < n; j += kThreads) { nz_idx[dst + j] = idx_in[src + j]; nz_n[dst + j] = n_in[src + j]; }
}

// set_data on the device: bin events into the analysis space with numpy.histogramdd semantics (what
// Histdd.add does in blueice/likelihood.py:608-609): per axis the bin is searchsorted(edges, x, 'right') - 1,
// the right-most edge is inclusive, events outside any axis range (or nan) are dropped.  Adding 1.0 with an fp64
// atomic is exact, so the result does not depend on the order of arrival.
struct HistArgs {
    int k;
    int n_edges[kMaxDim];
    int edge_off[kMaxDim];
};

__global__ __launch_bounds__(kThreads) void k_histogram(const double* __restrict__ coords /*[k][N]*/, int64_t N, HistArgs h,
                                                        const double* __restrict__ edges, double* __restrict__ counts) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    int64_t bin = 0;
    for (int a = 0; a < h.k; ++a) {
        const double x = coords[(int64_t)a * N + e];
        const double* __restrict__ g = edges + h.edge_off[a];
        const int n = h.n_edges[a];
        if (!(x >= g[0] && x <= g[n - 1])) return;      // out of range or nan
        int lo = 0, hi = n;                               // first index with g[idx] > x  (side = 'right')
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (g[mid] <= x) lo = mid + 1; else hi = mid;
        }
        int b = lo - 1;
        if (b == n - 1) b = n - 2;                        // x == last edge: belongs to the last bin
        bin = bin * (n - 1) + b;
    }
    atomicAdd(&counts[bin], 1.0);
}

// ---- unbinned set_data: histogram pdfs evaluated at the events ------------------------------------------------
// HistogramPdfSource.pdf (blueice/source.py:218-243) for every (anchor, source) row at once.  One thread = one event:
// the per-axis cell and weights are found once, then the thread walks the template rows (blockIdx.y strides over them).
//   method 0: density of the bin holding the event -- numpy.searchsorted(edges, x) - 1, clipped (nan sorts last);
//   method 1: scipy RegularGridInterpolator over the bin centres, its arithmetic: i = largest index with g[i] <= x
//             (at most n - 2), t = (x - g[i]) / (g[i+1] - g[i]); corners in itertools.product order (axis 0 slowest),
//             weight = ((1 * w_0) * w_1) ..., value = value + V * weight from 0 -- no contraction (the build sets
//             -ffp-contract=off), so the bits are scipy's.
struct ScoreArgs {
    int k, method;
    int clip;                    // 'linear': clip the coordinates to [first centre, last centre] HERE, as HistogramPdfSource.pdf does
                                 // (blueice/source.py:231-239) -- events simulated on the device keep their true coordinates
    int n_grid[kMaxDim];
    int grid_off[kMaxDim];
    int64_t stride[kMaxDim];     // bins (C order) per step along the axis
};

// Two passes (round 4).  Round 3's single kernel -- one thread per event looping over all anchors x sources rows -- let its
// blocks drift apart over the rows, so the 10^9-byte-scale gathers of 10^6 events came from a working set of gigabytes:
// every 8-byte value cost a whole memory transaction (10.6 ms for the 4 GB tensor of the C2 shape = 0.05 of the HBM peak).
// Now (1) k_score_locate finds every event's cell and interpolation weights ONCE, and (2) k_score_rows has the ROW as the
// slow grid dimension: blocks are dispatched row by row, so at any time the gathers of the whole chip fall into one or two
// 8 MB histograms that stay in L2 / Infinity Cache, and each histogram is read from HBM once (6.1 ms).  And (3) between the
// two the events are ORDERED BY CELL (radix sort of the cell indices, stable): the events of a block then fall into a few
// consecutive cache lines of the histogram, every line is fetched once per block instead of once per event, and the
// kernel becomes the stream it should be -- 4 GB in, 4 GB out.  The tensor's columns are then in sorted order; `perm`
// (sorted position -> the caller's event) stays with the context and bi_interpolate / bi_eval_full hand per-event values
// back in the caller's order, so the order is visible only to sums over events (1e-10, not bitwise, against a host-scored
// tensor).  Same arithmetic per value in the same order as before: bit-identical values.
__global__ __launch_bounds__(kThreads) void k_score_locate(const double* __restrict__ coords /*[k][N]*/, int64_t N, ScoreArgs a,
                                                           const double* __restrict__ grid, int64_t* __restrict__ base_out,
                                                           double* __restrict__ t_out /*[k][N], method 1 only*/) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    int64_t base = 0;
    for (int ax = 0; ax < a.k; ++ax) {
        double x = coords[(int64_t)ax * N + e];
        const double* __restrict__ g = grid + a.grid_off[ax];
        const int n = a.n_grid[ax];
        if (a.clip && a.method == 1) x = fmin(fmax(x, g[0]), g[n - 1]);
        int lo = 0, hi = n;
        if (a.method == 0) {                       // first index with g[idx] >= x  (side = 'left')
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (g[mid] < x) lo = mid + 1; else hi = mid;
            }
            if (x != x) lo = n;
        } else {                                   // first index with g[idx] > x
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (g[mid] <= x) lo = mid + 1; else hi = mid;
            }
        }
        const int i = min(max(lo - 1, 0), n - 2);
        if (a.method == 1) t_out[(int64_t)ax * N + e] = (x - g[i]) / (g[i + 1] - g[i]);
        base += i * a.stride[ax];
    }
    base_out[e] = base;
}

// events per thread of k_score_rows (their gathers are in flight together; one event where 2^K corners are many already)
constexpr int score_events_per_thread(int K) { return K <= 3 ? 4 : 1; }

// K: analysis dimensions of the 'linear' method (0 = 'piecewise': one value per event and row)
template <int K>
__global__ __launch_bounds__(kThreads) void k_score_rows(const int64_t* __restrict__ base_in, const double* __restrict__ t_in, int64_t N,
                                                         ScoreArgs a, const double* __restrict__ rows, int64_t row_stride, int n_rows,
                                                         double* __restrict__ out, int64_t out_stride,
                                                         const int32_t* __restrict__ perm /*NULL: base_in in event order*/) {
    constexpr int kScoreEvents = score_events_per_thread(K);
    const int64_t e0 = (int64_t)blockIdx.x * (kThreads * kScoreEvents) + threadIdx.x;
    int64_t base[kScoreEvents];
    double t[kScoreEvents][K > 0 ? K : 1];
#pragma unroll
    for (int j = 0; j < kScoreEvents; ++j) {
        const int64_t e = min(e0 + (int64_t)j * kThreads, N - 1);
        base[j] = base_in[e];                        // (sorted keys when perm is given)
        const int64_t ev = perm ? (int64_t)perm[e] : e;
#pragma unroll
        for (int ax = 0; ax < K; ++ax) t[j][ax] = t_in[(int64_t)ax * N + ev];
    }
    for (int r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const double* __restrict__ row = rows + (int64_t)r * row_stride;
        double value[kScoreEvents];
        if constexpr (K == 0) {
#pragma unroll
            for (int j = 0; j < kScoreEvents; ++j) value[j] = row[base[j]];
        } else {
#pragma unroll
            for (int j = 0; j < kScoreEvents; ++j) {
                const double* __restrict__ src = row + base[j];
                double v = 0.0;
                auto add_corner = [&](int corner) {
                    double weight = 1.0;
                    int64_t off = 0;
#pragma unroll
                    for (int ax = 0; ax < K; ++ax) {
                        const bool up = (corner >> (K - 1 - ax)) & 1;
                        weight = weight * (up ? t[j][ax] : 1.0 - t[j][ax]);
                        if (up) off += a.stride[ax];
                    }
                    v = v + src[off] * weight;
                };
                if constexpr (K <= 3) {
#pragma unroll
                    for (int corner = 0; corner < (1 << K); ++corner) add_corner(corner);
                } else {
#pragma unroll 1
                    for (int corner = 0; corner < (1 << K); ++corner) add_corner(corner);
                }
                value[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < kScoreEvents; ++j) {
            const int64_t e = e0 + (int64_t)j * kThreads;
            if (e < N) out[(int64_t)r * out_stride + e] = value[j];
        }
    }
}

// ---- event-level toy Monte Carlo for histogram-pdf sources (bi_simulate_events) ---------------------------------
// Model.simulate (blueice/model.py:69-91): per source N_s ~ Poisson(mu_s) events, each drawn from the source's pdf --
// for a histogram pdf (HistogramPdfSource.simulate, source.py:248-264 -> Histdd.get_random) a bin with probability
// proportional to density x volume, then a uniform position inside the bin.  Here: pmf rows of the morphed densities and
// their running sums are prepared per source (k_sim_pmf + a scan), then one thread per event finds its source from the
// per-source counts, its bin by bisection in that source's cumulative sums and its position in the bin -- Philox4x32-10
// counters (event within its source, source) keyed by the seed, so a toy does not depend on the launch geometry.
struct SimArgs {
    int k;                       // analysis dimensions
    int S;
    int n_edges[kMaxDim];
    int edge_off[kMaxDim];
    int64_t stride[kMaxDim];     // bins (C order) per step along the axis
};
constexpr uint32_t kSimTag = 0x53494D45u;

// dens [S][B] (morphed densities) -> pmf [S][B] = density x bin volume (negative / nan densities count as 0)
__global__ __launch_bounds__(kThreads) void k_sim_pmf(const double* __restrict__ dens, SimArgs a, const double* __restrict__ edges,
                                                      int64_t B, double* __restrict__ pmf) {
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    double vol = 1.0;
    int64_t rem = b;
    for (int ax = 0; ax < a.k; ++ax) {
        const int64_t i = rem / a.stride[ax];
        rem -= i * a.stride[ax];
        const double* __restrict__ e = edges + a.edge_off[ax];
        vol *= e[i + 1] - e[i];
    }
    const double v = dens[(int64_t)blockIdx.y * B + b] * vol;
    pmf[(int64_t)blockIdx.y * B + b] = v > 0.0 ? v : 0.0;
}

// events per source: N_s ~ Poisson(rate_s), counter (source, stream)
__global__ void k_sim_counts(const double* __restrict__ rates, int S, uint64_t seed, int64_t* __restrict__ n_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double M = rates[s];
    n_out[s] = (M > 0.0 && M < 1e15) ? (int64_t)toy_event_count(M, seed ^ 0x9E3779B97F4A7C15ull, (int64_t)s) : 0;
}

__global__ __launch_bounds__(kThreads) void k_sim_events(const double* __restrict__ cdf /*[S][B]*/, int64_t B, SimArgs a,
                                                         const double* __restrict__ edges, const int64_t* __restrict__ first /*[S+1]*/,
                                                         uint64_t seed, int64_t N, double* __restrict__ coords /*[k][N]*/,
                                                         int32_t* __restrict__ source /*[N]*/) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    int s = 0;
    while (s + 1 < a.S && e >= first[s + 1]) ++s;
    const int64_t j = e - first[s];                                      // event j of source s
    const double* __restrict__ F = cdf + (int64_t)s * B;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)s, kSimTag, k0, k1, r);
    const double target = u53(r[0], r[1]) * F[B - 1];
    int64_t lo = 0, hi = B;                                              // first bin whose cumulative sum exceeds the target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (F[mid] <= target) lo = mid + 1; else hi = mid;
    }
    int64_t rem = min(lo, B - 1);
    for (int ax = 0; ax < a.k; ++ax) {
        // the position inside the bin: one uniform per axis, counter (event, source, axis)
        philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)s | ((uint32_t)ax << 24), kSimTag + 1u, k0, k1, r);
        const double u = u53(r[0], r[1]);
        const int64_t i = rem / a.stride[ax];
        rem -= i * a.stride[ax];
        const double* __restrict__ ed = edges + a.edge_off[ax];
        // (uniform inside the bin, as Histdd.get_random draws it, source.py:248-264; a 'linear' pdf clips to the outer bin
        //  centres only when it is EVALUATED, source.py:231-239: k_score_locate does that, the stored events stay as drawn)
        coords[(int64_t)ax * N + e] = ed[i] + u * (ed[i + 1] - ed[i]);
    }
    source[e] = s;
}

// small device -> pinned-host copies as a kernel (bi_memcpy_to_host): a copy-engine transfer of 80 KB costs 15 ... 110 us on
// this runtime, a launch writing through the host mapping about 10
__global__ void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n_words) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// densify one dataset from its non-empty-bin list
__global__ void k_csr_to_dense(const int32_t* __restrict__ idx, const double* __restrict__ n, int64_t nnz,
                               double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) out[idx[j]] = n[j];
}

// ---- toy-MC form: one parameter point, many datasets --------------------------------------
// pass 1: mu_b -> logmu[b] (log mu, or -inf for mu == 0, or nan for invalid mu), partial sum mu
__device__ __forceinline__ void morph_logmu_body(const LaunchArgs& a, const int64_t* __restrict__ rowoff, const double* __restrict__ coef,
                                                 double* __restrict__ logmu, int store_mu) {
    double sum = 0.0;
    unsigned bad = 0u;
    log_table_load();
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double m0 = 0.0, m1 = 0.0;
#pragma unroll 8
        for (int k = 0; k < a.n0; ++k) {
            const double2 v = *reinterpret_cast<const double2*>(a.ps + rowoff[k] + bin0);
            const double c = coef[k];
            m0 = fma(c, v.x, m0);
            m1 = fma(c, v.y, m1);
        }
        double2 l;
        if (store_mu) {  // toy generation wants the expectation itself
            l.x = m0;
            l.y = m1;
        } else {
            l.x = (m0 >= 0.0) ? bin_log(m0) : __builtin_nan("");
            l.y = (m1 >= 0.0) ? bin_log(m1) : __builtin_nan("");
        }
        if (!(m0 >= 0.0) || !(m1 >= 0.0)) bad = 1u;
        *reinterpret_cast<double2*>(logmu + bin0) = l;
        sum += m0 + m1;
    }
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    sum = wave_sum(sum);
    bad = wave_or(bad);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = sum; shf[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        unsigned f = shf[0];
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[w]; f |= shf[w]; }
        a.partial[blockIdx.x] = t;
        a.pflags[blockIdx.x] = f;
    }
}

__global__ __launch_bounds__(kThreads) void k_morph_logmu(LaunchArgs a, double* __restrict__ logmu, int store_mu) {
    morph_logmu_body(a, a.rowoff, a.coef, logmu, store_mu);
}

// The point's descriptors in the kernel arguments (up to kMaxSingleStreams streams): a toy-MC call is ONE point, and a
// host-to-device copy of its 512 bytes ahead of the launch costs more than the launch (10 ... 15 us on this runtime).
struct PointDesc {
    int64_t rowoff[kMaxSingleStreams];
    double coef[kMaxSingleStreams];
};

__global__ __launch_bounds__(kThreads) void k_morph_logmu_desc(LaunchArgs a, PointDesc d, double* __restrict__ logmu, int store_mu) {
    morph_logmu_body(a, d.rowoff, d.coef, logmu, store_mu);
}

// pass 2: for dataset t: sum_b xlogy(n_tb, mu_b).  blockIdx.y = group of kDotGroup datasets, x strides tiles: the
// log mu tile is loaded once per group (it stays in L2 / Infinity Cache: 8 MB), the counts rows stream through once
// with the nontemporal hint; XCD-aware tile order as in morph_tiles.
constexpr int kDotGroup = 8;

__device__ __forceinline__ double xlogy_term(double n, double l) {
    double t = (n > 0.0) ? n * l : 0.0;
    if (n != n) t = __builtin_nan("");
    else if (n < 0.0 || n != floor(n)) t = -__builtin_inf();
    return t;
}

__global__ __launch_bounds__(kThreads) void k_dataset_dot(const double* __restrict__ counts,
                                                          const double* __restrict__ logmu, int64_t Bp, int n_tiles,
                                                          int64_t t0, int64_t n_sets, double* __restrict__ partial) {
    const int64_t d0 = (int64_t)blockIdx.y * kDotGroup;
    const double* __restrict__ c[kDotGroup];
#pragma unroll
    for (int g = 0; g < kDotGroup; ++g) c[g] = counts + (t0 + min(d0 + g, n_sets - 1)) * Bp;   // tail group: repeats the last row
    double s[kDotGroup];
#pragma unroll
    for (int g = 0; g < kDotGroup; ++g) s[g] = 0.0;
    const int chunks = n_tiles >= 64 * 8 ? 8 : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double2 n[kDotGroup];
#pragma unroll
        for (int g = 0; g < kDotGroup; ++g) n[g] = stream_load<true>(c[g] + bin0);
        const double2 l = *reinterpret_cast<const double2*>(logmu + bin0);
#pragma unroll
        for (int g = 0; g < kDotGroup; ++g) s[g] += xlogy_term(n[g].x, l.x) + xlogy_term(n[g].y, l.y);
    }
    __shared__ double sh[kThreads / 64][kDotGroup];
#pragma unroll
    for (int g = 0; g < kDotGroup; ++g) {
        const double w = wave_sum(s[g]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][g] = w;
    }
    __syncthreads();
    if (threadIdx.x < kDotGroup && d0 + threadIdx.x < n_sets) {
        double t = sh[0][threadIdx.x];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w][threadIdx.x];
        partial[(d0 + threadIdx.x) * gridDim.x + blockIdx.x] = t;
    }
}

// out[t] = sum_blocks partial[t][:] - summu - lgsum[t0 + t]   (nan if any mu invalid).  256 threads per block: the
// block first sums the mu partials of pass 1 together (fixed tree), then every thread finishes one dataset.
// (partial[t * t_stride + b * b_stride]: dataset-major from the row kernels, block-major from the tiled one, whose 123
// partials per dataset would otherwise be read with a stride of 123 doubles between neighbouring threads)
__global__ __launch_bounds__(kThreads) void k_dataset_finish(const double* __restrict__ partial, int nbx, int64_t t_stride, int64_t b_stride,
                                                             const double* __restrict__ mu_partial,
                                                             const unsigned* __restrict__ mu_flags, int nmu,
                                                             const double* __restrict__ lgsum, int64_t t0, int64_t n,
                                                             double* __restrict__ out) {
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    double m = 0.0;
    unsigned f = 0u;
    {   // (four loads in flight per thread: every block of this kernel repeats this sum before it can start on its datasets)
        double m4[4] = {0.0, 0.0, 0.0, 0.0};
        int b = threadIdx.x;
        for (; b + 3 * kThreads < nmu; b += 4 * kThreads) {
            double v[4];
            unsigned g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = mu_partial[b + k * kThreads]; g[k] = mu_flags[b + k * kThreads]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { m4[k] += v[k]; f |= g[k]; }
        }
        for (; b < nmu; b += kThreads) { m4[0] += mu_partial[b]; f |= mu_flags[b]; }
        m = (m4[0] + m4[1]) + (m4[2] + m4[3]);
    }
    m = wave_sum(m);
    f = wave_or(f);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = m; shf[threadIdx.x >> 6] = f; }
    __syncthreads();
    m = sh[0];
    f = shf[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) { m += sh[w]; f |= shf[w]; }
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    // eight running sums: the loads of a thread do not wait for one another (123 block-major partials per dataset from
    // the tiled kernel are 123 trips to L2 otherwise); fixed order all the same
    const double* __restrict__ p = partial + t * t_stride;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 7 < nbx; b += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(int64_t)(b + k) * b_stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    for (; b < nbx; ++b) acc[0] += p[(int64_t)b * b_stride];
    const double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    double r = (s - m) - lgsum[t0 + t];
    if (f) r = __builtin_nan("");
    out[t] = r;
}

// The finish of the tiled toy-MC pass, partial[tile][dataset] (block-major): 64 datasets per block, the tiles of a
// dataset split over the block's four waves (k_dataset_finish has one thread walk all ~123 tiles of its dataset, 16 trips
// to L2 one after the other on 40 blocks: 10 ... 15 us for 10^4 datasets; here 157 blocks and four trips).  Fixed order:
// a wave's range in steps of eight with eight running sums, then the four waves' sums ((0 + 1) + (2 + 3)).
// done != NULL: out is pinned host memory; the block that finishes last publishes `seq` there with a system-scope
// release once every block's results are on their way (the host polls the word instead of synchronising the stream).
__global__ __launch_bounds__(kThreads) void k_dataset_finish_tiled(const double* __restrict__ partial, int nbx,
                                                                   const double* __restrict__ mu_partial,
                                                                   const unsigned* __restrict__ mu_flags, int nmu,
                                                                   const double* __restrict__ lgsum, int64_t t0, int64_t n,
                                                                   double* __restrict__ out, unsigned* __restrict__ blocks_done,
                                                                   unsigned long long* done, unsigned long long seq) {
    static_assert(kThreads == 256, "four waves per block");
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    __shared__ double part[kThreads / 64][64];
    double m = 0.0;
    unsigned f = 0u;
    {
        double m4[4] = {0.0, 0.0, 0.0, 0.0};
        int b = threadIdx.x;
        for (; b + 3 * kThreads < nmu; b += 4 * kThreads) {
            double v[4];
            unsigned g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = mu_partial[b + k * kThreads]; g[k] = mu_flags[b + k * kThreads]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { m4[k] += v[k]; f |= g[k]; }
        }
        for (; b < nmu; b += kThreads) { m4[0] += mu_partial[b]; f |= mu_flags[b]; }
        m = (m4[0] + m4[1]) + (m4[2] + m4[3]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 64 + lane;
    const int per = (nbx + 3) / 4;
    const int b0 = wave * per, b1 = min(nbx, b0 + per);
    double s = 0.0;
    if (t < n) {
        const double* __restrict__ p = partial + t;
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int b = b0;
        for (; b + 7 < b1; b += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(int64_t)(b + k) * n];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
        for (; b < b1; ++b) acc[0] += p[(int64_t)b * n];
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    part[wave][lane] = s;
    m = wave_sum(m);
    f = wave_or(f);
    if (lane == 0) { sh[wave] = m; shf[wave] = f; }
    __syncthreads();
    if (wave == 0 && t < n) {
        double mm = sh[0];
        unsigned ff = shf[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { mm += sh[w]; ff |= shf[w]; }
        const double tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        double r = (tot - mm) - lgsum[t0 + t];
        if (ff) r = __builtin_nan("");
        out[t] = r;
    }
    if (!done) return;
    if (wave == 0) {
        __threadfence_system();                        // this wave's results have left before the block is counted
        if (lane == 0) {
            const unsigned before = __hip_atomic_fetch_add(blocks_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before == gridDim.x - 1) {
                __hip_atomic_store(blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (ready for the next call)
                __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

}  // namespace

namespace {

// ---- the scan kernel: many points per grid cell, fp64 matrix cores ---------------------------------------
// For a batch whose points pile up in few grid cells (likelihood scans), mu[point][bin] = sum_k coef[point][k] *
// row[k][bin] is a [points x streams] x [streams x bins] product.  One wave owns a strip of 16 CB bins of the cell's
// 2^d*S template rows, holds it in registers in v_mfma_f64_16x16x4 operand layout (k = lane >> 4,
// bin = lane & 15; loaded once, every 128-byte cache line fully used) and loops over ALL 16-point work items of
// the cell: per item KG coalesced coefficient loads (coef[k][point], point = lane & 15) and, for the CB 16-bin blocks
// of the strip, KG MFMAs each, then the Poisson epilogue of the block.  Two cross-row exchanges leave the 16
// per-point sums in the first 16 lanes, which add them (no-return fp64 atomics) into a partial slot that only this
// wave ever touches, so the result is deterministic.
// Bound: 78.6 TFLOP/s fp64 matrix peak / (2 * 2^d*S * B flop per evaluation) = 1.2 M evaluations/s at C2 for the
// FMA work alone; fp64 MFMA and fp64 VALU share the same units on this chip (measured: tools/micro/
// mfma_valu_overlap.hip), so the epilogue's logarithms add to that rather than hide under it.
// Plain binned likelihood, up to 32 streams (K <= 32); everything else takes k_morph_reduce.
typedef double bi_double4 __attribute__((ext_vector_type(4)));

struct ScanArgs {
    const double* ps;
    const double* counts;
    const int64_t* rowoff;      // [items][NS]   (rows of a group = rows of its first item)
    const double* coef;         // [items][NS][16]
    const int64_t* item_cnt;    // [items]
    const int32_t* item_tiles;  // [items] 512-bin tiles of the item's rows
    const int64_t* grp_first;   // [groups] first item of the group
    const int32_t* grp_items;   // [groups] items in the group
    double* partial;            // [items][nslots][16], zero on entry
    int NS;
    int nslots;                 // waves per group = gridDim.x * 4
    int n_groups;               // k_scan_sorted (one-dimensional grid, dealt to the XCDs in contiguous ranges of blocks)
    int xcd_mode;               // 0 launch order, 1 contiguous ranges (default), 2 group g -> XCD g mod 8
    int share_slow;             // k_scan_sorted: strips of mixed counts are worked by all waves of the cell together
};

// sum of a double over the 4 DPP rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48): one half-row exchange and one half-wave
// exchange (v_permlane16_swap / v_permlane32_swap, gfx950), every lane ends up with the total
__device__ __forceinline__ double rows4_sum(double v) {
#define BI_SWAP_ADD(SWAP)                                                                                          \
    do {                                                                                                           \
        const unsigned long long u = __double_as_longlong(v);                                                      \
        const auto lo = SWAP((unsigned)u, (unsigned)u, false, false);                                              \
        const auto hi = SWAP((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);                              \
        v = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) +                                      \
            __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);                                       \
    } while (0)
    BI_SWAP_ADD(__builtin_amdgcn_permlane16_swap);
    BI_SWAP_ADD(__builtin_amdgcn_permlane32_swap);
#undef BI_SWAP_ADD
    return v;
}

// CB: 16-bin blocks per strip (strip = CB * 16 bins).  KG: groups of 4 streams (4 KG >= NS).  MASK: NS < 4 KG,
// the coefficient operands of the padding streams must be zeroed.
// Operand roles: the template strip is the MFMA's A operand (row i = bin = lane & 15, k = lane >> 4), the coefficients
// its B operand (k = lane >> 4, column j = point = lane & 15), so lane (kq, col) receives mu[bin = 4 r + kq][point = col]
// in accumulator element r: ALL FOUR elements of a lane belong to ONE point.  The per-point sum therefore needs three
// in-lane additions and two cross-row exchanges per item (rows4_sum: ~10 vector instructions) -- with the operands the
// other way round (bins along the lanes of a row) it took four 16-lane rotations per accumulator element, ~60
// instructions per item, a fifth of the kernel's vector work when every bin has data.
// PROD = 1: the rows are the compacted non-empty bins of sparse data -- blocks whose counts are all 1 or 2 take one logarithm of
// the product mu^n over a lane's four bins (a separate instantiation, so that the dense-data kernel keeps its code).
// (Rows ordered by their count -- dense data, or the count-sorted compacted copy -- are k_scan_sorted's, bi_scan_sorted.h.)
// MASK: the ROWS of the padding streams are zeroed once per strip; their coefficient reads are steered to a valid element
// of the item's last stream group (one select on a scalar condition per group, no per-group offset registers).
template <int CB, int KG, bool MASK, int PROD = 0>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(CB == 2 ? 3 : 2))) void k_scan_mfma(ScanArgs a) {
    constexpr int STRIP = CB * 16;
    const int grp = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = blockIdx.x * 4 + wave;
    const int64_t item0 = a.grp_first[grp];
    const int n_items = a.grp_items[grp];
    const int64_t* __restrict__ rowoff = a.rowoff + item0 * a.NS;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item0];
    const int n_strips = a.item_tiles[item0] * (kTile / STRIP);
    // (the group's row offsets: in LDS, read again for every strip -- as loop invariants they would hold 2 KG registers)
    __shared__ int64_t s_rowoff[4 * KG];
    if (threadIdx.x < 4 * KG) s_rowoff[threadIdx.x] = rowoff[min((int)threadIdx.x, a.NS - 1)];
    log_table_load();
    const int kq = lane >> 4, col = lane & 15;
    const int aoff0 = min(kq, a.NS - 1) * 16 + col;          // coefficient of K group kg sits at aoff0 + kg * 64 ...
    const int kg_last = (a.NS - 1) >> 2;                     // ... up to the group that holds stream NS - 1: from there on
    const int alast = min(kg_last * 4 + kq, a.NS - 1) * 16 + col;   // the lane reads this (valid) element instead
#define BI_COEF_AT(kg) ((MASK && (kg) >= kg_last) ? alast : aoff0 + (kg) * 64)

    for (int strip = slot; strip < n_strips; strip += a.nslots) {
        const int64_t bin0 = (int64_t)strip * STRIP;
        double b[KG][CB], n[CB][4];
        int kqo = kq;
        asm volatile("" : "+v"(kqo));           // (opaque: keeps the LDS reads inside the strip loop)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int64_t row = s_rowoff[kg * 4 + kqo];               // streams beyond NS: a valid row, zeroed
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const double v = a.ps[row + bin0 + cb * 16 + col];
                b[kg][cb] = (MASK && kg * 4 + kq >= a.NS) ? 0.0 : v;
            }
        }
        // Everything about the counts is known per bin, once per strip: kind 0 = empty bin (term -mu),
        // 1 = n > 0 (adds n log mu), 2 = negative / non-integer n (-inf), 3 = nan n (nan); scipy's poisson.logpmf
        // (the four kinds of a lane's bins packed into one register, 2 bits each: registers decide the occupancy here)
        int kinds[CB];
        bool special[CB], alldata[CB], ones_twos[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            kinds[cb] = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = cnt[bin0 + cb * 16 + 4 * r + kq];
                n[cb][r] = v;
                kinds[cb] |= ((v != v) ? 3 : ((v < 0.0 || v != floor(v)) ? 2 : (v > 0.0 ? 1 : 0))) << (2 * r);
            }
            special[cb] = __ballot(kinds[cb] != 0) != 0ull;       // wave-uniform: does any bin of this block need more
            alldata[cb] = __ballot(kinds[cb] == 0x55) == ~0ull;   // ... every bin holds a count > 0: the logarithm alone decides
            // ... and every count is 1 or 2 (the non-empty bins of sparse data): sum n log mu = log prod mu^n, and the four
            // bins of a lane belong to one point, so four logarithms become five multiplications and one logarithm.  The
            // factors are positive normal numbers (checked per item); a product that leaves that range takes the bin-wise form.
            bool small = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) small &= n[cb][r] == 1.0 || n[cb][r] == 2.0;
            ones_twos[cb] = PROD == 1 && __ballot(small) == ~0ull;
        }
#define BI_KIND(cb, r) ((kinds[cb] >> (2 * (r))) & 3)

        // coefficient operands: coef[k][point]; streams beyond NS read a valid element (their rows are zero)
        double av[KG];
        {
            const double* __restrict__ coef = a.coef + item0 * a.NS * 16;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) av[kg] = coef[BI_COEF_AT(kg)];
        }
        // (the item's coefficient block and its partial slot advance by fixed steps: pointers, not products per item)
        const double* __restrict__ coef_next = a.coef + item0 * a.NS * 16;
        double* __restrict__ dst = a.partial + (item0 * a.nslots + slot) * 16 + col;
        const int64_t coef_step = (int64_t)a.NS * 16, dst_step = (int64_t)a.nslots * 16;
        for (int it = 0; it < n_items; ++it) {
            if (it + 1 < n_items) coef_next += coef_step;
            double s[4] = {0.0, 0.0, 0.0, 0.0};      // four chains, one point
            double mn = 0.0;                          // running minimum of mu: a negative expectation makes the result nan
            bi_double4 acc[CB];
            // all chains first (no vector instruction executes beside an fp64 MFMA anyway), then the next item's
            // coefficients are requested straight into the registers the chains have just read -- they arrive under the
            // epilogues, and there is neither a second register set nor a rotation
#define BI_CHAIN(cb)                                                                                               \
    do {                                                                                                           \
        acc[cb] = bi_double4{0.0, 0.0, 0.0, 0.0};                                                                  \
        _Pragma("unroll") for (int kg = 0; kg < KG; ++kg)                                                          \
            acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[kg][cb], av[kg], acc[cb], 0, 0, 0);                   \
    } while (0)
    /* (the linear part, -sum_b mu_b = -sum_k coef_k * rowsum_k, is in the per-point constant: k_plan_fill, linear_outside) */ \
#define BI_EPILOGUE(cb)                                                                                            \
    do {                                                                                                           \
        if (alldata[cb]) { /* dense data: n log mu in every bin; mu <= 0 / nan comes out of the checked logarithm */ \
            if (PROD == 1 && ones_twos[cb]) { /* (wave-uniform) product form */                                    \
                bool low = false;                                                                                  \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) low |= !(acc[cb][r] > kProdFloor);                   \
                if (__ballot(low) == 0ull) {                                                                       \
                    /* counts of 1 and 2 only: ONE logarithm of the product of mu^n.  At most eight factors above  */ \
                    /* kProdFloor: no partial product can be subnormal, one that overflows stays +inf to the end   */ \
                    double f[4];                                                                                   \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
                        f[r] = (PROD == 1 && n[cb][r] == 2.0) ? acc[cb][r] * acc[cb][r] : acc[cb][r];              \
                    const double prod = (f[0] * f[1]) * (f[2] * f[3]);                                             \
                    if (__ballot(!pos_normal(prod)) == 0ull) {                                                     \
                        s[cb & 3] += bin_log_fast(prod);                                                           \
                        break;                                                                                     \
                    }                                                                                              \
                }                                                                                                  \
            }                                                                                                      \
            bool checked = false;                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) checked |= !pos_normal(acc[cb][r]);                      \
            if (__ballot(checked) == 0ull) {                                                                       \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) s[r] = fma(n[cb][r], bin_log_fast(acc[cb][r]), s[r]); \
            } else {                                                                                               \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) s[r] += n[cb][r] * bin_log(acc[cb][r]);              \
            }                                                                                                      \
            break;                                                                                                 \
        }                                                                                                          \
        mn = fmin(mn, fmin(fmin(acc[cb][0], acc[cb][1]), fmin(acc[cb][2], acc[cb][3])));                           \
        if (special[cb]) {                                                                                         \
            bool checked = false;                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) checked |= BI_KIND(cb, r) == 1 && !pos_normal(acc[cb][r]);  \
            if (__ballot(checked) == 0ull) {                                                                       \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                    \
                    const double lg = bin_log_fast(acc[cb][r]);                                                    \
                    if (BI_KIND(cb, r) == 1) s[r] += n[cb][r] * lg;                                                   \
                }                                                                                                  \
            } else {                                                                                               \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                      \
                    if (BI_KIND(cb, r) == 1) s[r] += n[cb][r] * bin_log(acc[cb][r]);                                  \
            }                                                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                          \
                if (BI_KIND(cb, r) > 1) s[r] += BI_KIND(cb, r) == 2 ? -__builtin_inf() : __builtin_nan("");              \
        }                                                                                                          \
    } while (0)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) BI_CHAIN(cb);
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) av[kg] = coef_next[BI_COEF_AT(kg)];
            double tot = 0.0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) BI_EPILOGUE(cb);
            tot = (s[0] + s[1]) + (s[2] + s[3]);
            if (mn < 0.0) tot = __builtin_nan("");
#undef BI_CHAIN
#undef BI_EPILOGUE
#undef BI_KIND
            tot = rows4_sum(tot);                     // over the four DPP rows: the 16 bins of the block are spread 4 r + kq
            if (kq == 0) unsafeAtomicAdd(dst, tot);
            dst += dst_step;
        }
    }
#undef BI_COEF_AT
}

// The finish of a scan plan: partial is [items][nslots][16] (point fastest), one wave per item.  Lane l takes point
// l & 15 and the slots l >> 4, l >> 4 + 4, ...: every load instruction of the wave covers 64 consecutive doubles (k_finish
// walks the same array with one 8-byte element per 128-byte line and took 1.3 ms for the 640 MB of a 10^6-point scan;
// this takes 0.1).  Fixed order: slots in steps of four per lane, then the four lane groups -- bitwise reproducible.
__global__ __launch_bounds__(kThreads) void k_finish_scan(const double* __restrict__ partial, int nslots, int64_t n_items,
                                                          const int64_t* __restrict__ perm, const double* __restrict__ slot_lg,
                                                          double* __restrict__ out) {
    const int64_t item = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int lane = threadIdx.x & 63, g = lane & 15;
    const double* __restrict__ src = partial + item * nslots * 16 + g;
    double s = 0.0;
#pragma unroll 4
    for (int b = lane >> 4; b < nslots; b += 4) s += src[(int64_t)b * 16];
    s = rows4_sum(s);
    if (lane < 16) {
        const int64_t p = perm[item * 16 + g];
        if (p >= 0) out[p] = s - slot_lg[item * 16 + g];
    }
}

// ---- the validity pass of a dense scan over sparse data -----------------------------------------------------
// "Every bin visited" with mostly empty data splits into two passes (plan->valid, bi_planning_device.h):
//   (A) the bins WITH data, on the compacted rows: n log mu - mu for those bins, and the linear remainder
//       -sum_{empty b} mu_b = -sum_k coef_k * (row total over the empty bins) from tables -- the non-empty-bin form;
//   (B) this kernel, over ALL bins: mu[point][bin] on the fp64 matrix cores exactly as in k_scan_mfma, and the one
//       thing an empty bin can still do to the result -- scipy's poisson.logpmf is nan where mu is negative or nan
//       (blueice/likelihood.py:674), whatever n is.  So the epilogue is one compare per matrix element, no logarithm,
//       no running sums, no cross-lane reduction; a point with any such bin is flagged and set to nan afterwards.
// With non-negative templates and rates (B) can never fire (that is why (A) alone is the default path); it is what makes
// the split exact for templates or rates of either sign.  Per 16-point item and 64-bin strip: 32 MFMAs (2048 cycles of
// the SIMD's fp64 pipe) + 16 v_cmp -- against ~240 vector instructions in k_scan_mfma, which matter because on this
// chip NO vector instruction executes beside an fp64 MFMA (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r02_scan_pmc.json).
struct ValidArgs {
    const double* ps;
    const int64_t* rowoff;      // [items][NS] element offsets of the FULL rows (rows of a group = rows of its first item)
    const double* coef;         // [items][NS][16]
    const int64_t* grp_first;   // [groups]
    const int32_t* grp_items;   // [groups]
    unsigned* bad;              // [items][16], zero on entry: set to 1 where a point has a bin with mu < 0 or nan
    int NS;
    int nslots;                 // waves per group = gridDim.x * 4
    int n_strips;               // strips of 16 CB bins per full row
};

template <int CB, int KG, bool MASK>
__global__ __launch_bounds__(kThreads) void k_scan_valid(ValidArgs a) {
    constexpr int STRIP = CB * 16;
    const int grp = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = blockIdx.x * 4 + wave;
    const int64_t item0 = a.grp_first[grp];
    const int n_items = a.grp_items[grp];
    const int64_t* __restrict__ rowoff = a.rowoff + item0 * a.NS;
    const int kq = lane >> 4, col = lane & 15;
    const int aoff0 = min(kq, a.NS - 1) * 16 + col;

    for (int strip = slot; strip < a.n_strips; strip += a.nslots) {
        const int64_t bin0 = (int64_t)strip * STRIP + col;
        double b[KG][CB];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int64_t row = rowoff[min(kg * 4 + kq, a.NS - 1)];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) b[kg][cb] = a.ps[row + bin0 + cb * 16];
        }
        double av[KG];
        {
            const double* __restrict__ coef = a.coef + item0 * a.NS * 16;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int k = kg * 4 + kq;
                av[kg] = coef[MASK ? min(k, a.NS - 1) * 16 + col : aoff0 + kg * 64];
                if (MASK && k >= a.NS) av[kg] = 0.0;
            }
        }
        // The next item's coefficients are requested behind the first chain and arrive under the others; two copies of the
        // item body alternate the two register sets (no rotation), and the coefficient block / flag words advance as pointers.
        double an[KG];
        const double* __restrict__ coef_next = a.coef + item0 * a.NS * 16;
        unsigned* __restrict__ bad = a.bad + item0 * 16 + kq;
        const int64_t coef_step = (int64_t)a.NS * 16;
        auto item = [&](double (&cur)[KG], double (&nxt)[KG], bool more) __attribute__((always_inline)) {
            if (more) coef_next += coef_step;
            bi_double4 acc[CB];
            unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};      // per r: lanes whose element is not >= 0
#define BI_VCHAIN(cb)                                                                                              \
    do {                                                                                                           \
        acc[cb] = bi_double4{0.0, 0.0, 0.0, 0.0};                                                                  \
        _Pragma("unroll") for (int kg = 0; kg < KG; ++kg)                                                          \
            acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[kg], b[kg][cb], acc[cb], 0, 0, 0);                  \
    } while (0)
#define BI_VCHECK(cb)                                                                                              \
    do {                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) m[r] |= __ballot(!(acc[cb][r] >= 0.0));                      \
    } while (0)
            BI_VCHAIN(0);
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int k = kg * 4 + kq;
                nxt[kg] = coef_next[MASK ? min(k, a.NS - 1) * 16 + col : aoff0 + kg * 64];
                if (MASK && k >= a.NS) nxt[kg] = 0.0;
            }
#pragma unroll
            for (int cb = 1; cb < CB; ++cb) {
                BI_VCHAIN(cb);
                BI_VCHECK(cb - 1);
            }
            BI_VCHECK(CB - 1);
#undef BI_VCHAIN
#undef BI_VCHECK
            if ((m[0] | m[1] | m[2] | m[3]) != 0ull) {         // rare (never with templates and rates >= 0)
                // element r of lane (kq, col) belongs to point kq + 4 r; lane 16 kq speaks for its row of 16 bins
                if (col == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((m[r] >> (16 * kq)) & 0xFFFFull) atomicOr(bad + 4 * r, 1u);
                }
            }
            bad += 16;
        };
        int it = 0;
        for (; it + 1 < n_items; it += 2) {
            item(av, an, true);
            item(an, av, it + 2 < n_items);
        }
        if (it < n_items) item(av, an, false);
    }
}

// out[perm[slot]] = nan where the validity pass flagged the slot
__global__ __launch_bounds__(kThreads) void k_apply_bad(const unsigned* __restrict__ bad, const int64_t* __restrict__ perm,
                                                        int64_t n_slots, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_slots || !bad[i]) return;
    const int64_t p = perm[i];
    if (p >= 0) out[p] = __builtin_nan("");
}

}  // namespace
