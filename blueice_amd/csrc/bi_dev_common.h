// bi_dev_common.h -- device helpers and launch-argument structs shared by every translation unit of libblueice_hip
// (gfx950 only).  No __global__ function lives here: a kernel is defined in the header of the ONE translation unit that
// launches it (bi_k_*.h), so that the families compile in parallel and no kernel is emitted twice.  DESIGN.md section 4.
#pragma once

// ---- launch-argument structs (global scope: they cross translation units through the tu_launch_* functions) ----

constexpr long long kMailTicksPerMs = 100000;                          // wall_clock64 runs at 100 MHz

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

// k_scan_bb (bi_k_scan_bb.h): Beeston-Barlow scans on the matrix cores
struct BbScanArgs {
    const double* ps;           // template rows
    const double* nm;           // Monte-Carlo count rows of the Beeston-Barlow source
    const double* counts;
    const int64_t* rowoff;      // [items][NS]   (rows of a group = rows of its first item)
    const double* coef;         // [items][NS][16]: n0 streams into U, nc into P, nc into a
    const double* aux;          // [items][16][2]  (p_cal, N) per point
    const int64_t* item_cnt;    // [items] element offset of the item's counts row
    const int64_t* grp_first;   // [groups] first item of the group
    const int32_t* grp_items;   // [groups] items in the group
    double* partial;            // [items][gridDim.x][16]
    unsigned* pflags;           // [items][gridDim.x][16]
    int64_t B;
    int n0, nc;
    int n_tiles;                // tiles of 16 bins covering [0, B)
};

// k_grad_mfma (bi_k_grad_mfma.h)
struct GradMfmaArgs {
    const double* ps;
    const double* counts;
    const int64_t* rowoff;      // [items][NS]   (rows of a group = rows of its first item)
    const double* coef;         // [items][NS][16]  value coefficients w_corner * r_source (unused slots repeat a point)
    const int64_t* item_cnt;    // [items]
    const int32_t* item_tiles;  // [items]
    const int64_t* grp_first;   // [groups]
    const int32_t* grp_items;   // [groups]
    double* part_ll;            // [items][n_slices][16]
    double* part_g;             // [items][n_slices][NSP][16]   NSP = 16 * NB
    int NS, n_slices;
};

namespace {

// ------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// exchange for a transposing reduction: returns a' * b' where a' = [a | b](lower halves / even rows), b' = the others
#define BI_SWAP_MUL(SWAP, a, b, out)                                                                               \
    do {                                                                                                           \
        const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);                       \
        const auto lo = SWAP((unsigned)ua, (unsigned)ub, false, false);                                            \
        const auto hi = SWAP((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);                            \
        out = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) *                                    \
              __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);                                     \
    } while (0)
// (used by k_scan_sorted and k_grad_mfma: four items' products over the four 16-lane rows of a wave, row q keeping item q --
//  BI_SWAP_MUL(permlane32_swap, p0, p2, x); BI_SWAP_MUL(permlane32_swap, p1, p3, y); BI_SWAP_MUL(permlane16_swap, x, y, P))

// a lane's double as a wave-uniform (scalar) value
__device__ __forceinline__ double lane_value(double v, int src_lane) {
    const unsigned long long u = __double_as_longlong(v);
    return __longlong_as_double(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), src_lane) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)u, src_lane));
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

// the post of a block's partial sum with the two injected faults (both -1 in production: two scalar compares per block)
__device__ __forceinline__ void mail_post_checked(const LaunchArgs& a, double* slot, double v) {
    if ((int)blockIdx.x == a.skip_post) return;
    if ((int)blockIdx.x == a.late_post) {
        const long long until = (long long)wall_clock64() + 2 * a.mail_timeout;
        while ((long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
    }
    mail_post(slot, v);
}

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

}  // namespace
