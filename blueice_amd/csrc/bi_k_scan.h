// bi_k_scan.h -- the matrix-core scan kernels over rows in bin order (k_scan_mfma) and the validity pass of split
// scans (k_scan_valid): translation unit tu_scan.hip.  (Rows ordered by count: bi_scan_sorted.h, tu_scan_sorted.hip.)
#pragma once

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

}  // namespace
