// bi_scan_sorted.h -- the matrix-core scan kernel for rows ordered by count (round 4).
// Translation unit tu_scan_sorted.hip (uses the logarithm, ScanArgs and the permlane helpers of bi_dev_common.h).
//
// k_scan_mfma<2,KG,MASK,2> (round 3) took one logarithm per lane, 16-point work item and 32-bin strip and issued 3.8
// vector instructions per MFMA -- on a chip where nothing executes beside an fp64 MFMA that kept the matrix pipe 69 %
// busy.  This kernel works on 64-bin strips and FOUR work items at a time:
//   * a lane's 16 matrix elements (4 blocks x 4 accumulator elements) are 16 bins of ONE point; in count order the whole
//     strip carries one count n, so the strip's term of that point is n log of the product of its 64 expectations;
//   * the lane's 16 factors are multiplied in a tree (15 multiplications); the product over the four 16-lane rows and
//     the hand-over "row q keeps item q" are ONE transposing reduction for four items (v_permlane32_swap /
//     v_permlane16_swap on two different registers: six swaps and three multiplications for four items, where the sum
//     over rows of one item took four swaps, four moves and two additions);
//   * so ONE logarithm serves four items (every row of the wave evaluates the logarithm of another item), and one
//     64-lane atomic instruction adds the four results to their partial slots;
//   * range checks without branches per item: the smallest high word of the 16 factors (v_min3_i32: negative numbers,
//     zeros and subnormals compare low as integers) against 2^-127 -- eight factors above it multiply to a normal
//     number in any grouping --, the lane product inside (2^-255, 2^255) -- four of those multiply to a normal number;
//     nan and inf fail the last test.  The verdicts of a quad are ANDed in scalar registers and tested once.
// Per item and strip: 32 MFMAs and ~36 vector instructions (1.1 per MFMA; round 3: 3.8).
// Expectations far from 1 (rates of 1e45 or 1e-90, a handful of events in 10^6 bins) would leave that window; the wave
// therefore keeps a power-of-two scale 2^s for its template rows (exact; applied once per strip, taken back inside the
// logarithm's exponent), set from the strip's own terms whenever a quad had to take the slow path.
// Strips of empty bins (n = 0) need no logarithm at all: one compare per element (scipy: nan where mu is negative or
// nan).  Everything else -- mixed counts where two runs meet, negative / non-integer / nan counts, expectations of zero --
// takes the slow path: item by item, block by block, bin by bin where it must, with scipy's values for every argument.
#pragma once

namespace {

// log(x * 2^k_adjust) for positive normal x, Horner steps with their constants in scalar registers (the compiler's own
// selection copies each constant into a fresh vector register pair first: seven moves per logarithm)
__device__ __forceinline__ double fma_sc(double x, double p, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(p), "s"(c));
    return r;
}
__device__ __forceinline__ double log_core_s(double x, int k_adjust) {
    const unsigned long long ix = __double_as_longlong(x);
    const int hi = (int)(ix >> 32);
    const int t = hi - 0x3FE60000;
    const int k0 = t >> 20;
    const int k = k0 + k_adjust;
    int zhi;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(zhi) : "v"(k0), "s"(-(1 << 20)), "v"(hi));
    const double z = __longlong_as_double(((unsigned long long)(unsigned)zhi << 32) | (ix & 0xFFFFFFFFull));
    const double4 e = s_log_table[(t >> 13) & 127];
    const double kd = (double)k;
    const double r = fma(z, e.x, -1.0);
    const double w = fma(kd, kLn2Hi, e.y);
    const double tail = fma(kd, kLn2Lo, e.z);
    double p = fma(r, -1.0 / 8.0, e.w);
    p = fma_sc(r, p, -1.0 / 6.0);
    p = fma_sc(r, p, 1.0 / 5.0);
    p = fma_sc(r, p, -1.0 / 4.0);
    p = fma_sc(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    const double q = fma(r * r, p, tail);
    const double h = w + r;
    const double err = (w - h) + r;
    return h + (err + q);
}

constexpr int kFloorHi = 0x38000000;          // high word of 2^-127
constexpr double kLaneLo = 0x1p-255, kLaneHi = 0x1p255;
constexpr int kScaleMax = 900;

template <int KG, bool MASK>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3))) void k_scan_sorted(ScanArgs a) {
    constexpr int CB = 4, STRIP = 64;
    // Blocks are dispatched to the 8 XCDs round-robin in launch order.  The (group, block) pairs, group-major, are cut
    // into eight contiguous ranges, one per XCD: the blocks of a group walk the same coefficient list (4 KB per work item,
    // 4 MB per cell of a 10^6-point scan, read again for every strip), which then lives in ONE XCD's L2 (two, where a range
    // ends inside a group) instead of being fetched into all eight -- and the XCDs carry equal numbers of blocks whatever
    // the number of groups is.
    const unsigned gx = (unsigned)a.nslots / 4, total = gx * (unsigned)a.n_groups;
    const unsigned per_xcd = (total + 7) / 8;
    unsigned linear = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (a.xcd_mode == 0) linear = blockIdx.x;                       // (tuning: plain launch order)
    else if (a.xcd_mode == 2) linear = (((blockIdx.x >> 3) / gx) * 8 + (blockIdx.x & 7)) * gx + (blockIdx.x >> 3) % gx;
    if (linear >= total) return;
    const int grp = (int)(linear / gx);
    // (readfirstlane: everything derived from the wave's number -- its slot, its strips, the item ranges of the shared pass -- is
    //  then scalar for the compiler too; taken for lane-dependent, the item loops' address arithmetic moves to the vector unit)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = (int)(linear % gx) * 4 + wave;
    const int64_t item0 = a.grp_first[grp];
    const int n_items = a.grp_items[grp];
    const int64_t* __restrict__ rowoff = a.rowoff + item0 * a.NS;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item0];
    const int n_strips = a.item_tiles[item0] * (kTile / STRIP);
    const int kq = lane >> 4, col = lane & 15;
    const int NS = a.NS;
    // the group's row offsets sit in LDS and are read again for every strip: as loop invariants they would occupy 2 KG
    // vector registers for the whole kernel (the four waves of a block work on the same group)
    __shared__ int64_t s_rowoff[4 * KG];
    __shared__ double s_cnt[kThreads / 64][64];
    if (threadIdx.x < 4 * KG) s_rowoff[threadIdx.x] = rowoff[min((int)threadIdx.x, NS - 1)];
    log_table_load();
    // coefficient of stream group kg for this lane: element loff[kg] of the item's [NS][16] block (streams beyond NS read
    // a valid element; their ROWS are zeroed instead, once per strip)
    unsigned loff[KG];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) loff[kg] = (unsigned)(((MASK && kg == KG - 1) ? min(kg * 4 + kq, NS - 1) : kg * 4 + kq) * 16 + col);
    const double* __restrict__ coef0 = a.coef + item0 * NS * 16;       // (wave-uniform)
    const int64_t coef_step = (int64_t)NS * 16, dst_step = (int64_t)a.nslots * 16;
    double* __restrict__ part0 = a.partial + (item0 * a.nslots + slot) * 16;   // (wave-uniform) item 0 of the group, this wave's slot
    const unsigned qoff = (unsigned)(kq * (int)dst_step + col);          // row q of a quad adds to item it + q
    int sc = 0;                                                         // the rows in registers are the templates * 2^sc
    int sc_learnt = 0;                                                  // the scale this wave's last slow quad asked for

    // A strip that is worked item by item costs about 25 strips of the product form, and the wave that owns it would still
    // be at it long after the cell's other waves have finished (ONE such strip -- the end of the data next to the padding --
    // was 14 % of a 10^6-point scan of C2's 9930 non-empty bins).  So a wave does all items only of its own UNIFORM strips
    // (first count == last count: in count order that is what uniform means); the others are done afterwards by ALL waves of
    // the cell together, wave w taking items w, w + nslots, ... of each -- into its own partial slot, as always.  Both passes
    // use the same predicate, so every strip is worked exactly once whatever its counts are.
    const bool share = a.nslots > 1 && a.share_slow;
    // (loop state: ONE cursor and the pass -- the kernel is at its limit of scalar registers, and what does not fit is
    //  computed with vector instructions inside the item loops)
    int cursor = slot;
    bool shared = false;
    while (true) {
        int strip;
        if (!shared) {
            if (cursor >= n_strips) {
                if (!share) break;
                shared = true;
                cursor = 0;
                continue;
            }
            strip = cursor;
            cursor += a.nslots;
            if (share && !(cnt[(int64_t)strip * STRIP] == cnt[(int64_t)strip * STRIP + STRIP - 1])) continue;
        } else {
            if (cursor >= n_strips) break;
            // the 64 strips around the cursor, one per lane; the first mixed one at or behind it
            const int batch0 = cursor & ~63;
            const int s = min(batch0 + lane, n_strips - 1);
            const bool mixed = !(cnt[(int64_t)s * STRIP] == cnt[(int64_t)s * STRIP + STRIP - 1]);
            const unsigned long long batch = __builtin_amdgcn_ballot_w64(mixed && batch0 + lane < n_strips && batch0 + lane >= cursor);
            if (batch == 0ull) {
                cursor = batch0 + 64;
                continue;
            }
            strip = batch0 + __builtin_ctzll(batch);
            cursor = strip + 1;
        }
        // items of this strip that this wave works: all of them, or its share
        const int i_first = shared ? slot : 0, i_step = shared ? a.nslots : 1;
        const int64_t bin0 = (int64_t)strip * STRIP;
        double b[KG][CB];
#define BI_LOAD_ROWS()                                                                                             \
    do {                                                                                                           \
        int kqo = kq;                                                                                              \
        asm volatile("" : "+v"(kqo));           /* (opaque: keeps the LDS reads inside the strip loop) */           \
        _Pragma("unroll") for (int kg = 0; kg < KG; ++kg) {                                                        \
            const int k = kg * 4 + kq;                                                                             \
            const int64_t row = s_rowoff[kg * 4 + kqo];                                                            \
            _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) {                                                    \
                const double v = a.ps[row + bin0 + cb * 16 + col];                                                 \
                b[kg][cb] = (MASK && kg == KG - 1 && k >= NS) ? 0.0 : v;                                                       \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
        // 2^s times the rows -- exact, unless an entry would become subnormal: then the strip stays unscaled
#define BI_SCALE_ROWS(s_new)                                                                                       \
    do {                                                                                                           \
        sc = 0;                                                                                                    \
        if ((s_new) != 0) {                                                                                        \
            bool lossy = false;                                                                                    \
            _Pragma("unroll") for (int kg = 0; kg < KG; ++kg)                                                      \
                _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) {                                                \
                    const double v = ldexp(b[kg][cb], (s_new));                                                    \
                    lossy |= (b[kg][cb] != 0.0) && !__builtin_amdgcn_class(v, 0x108);   /* +-normal */                 \
                }                                                                                                  \
            if (__builtin_amdgcn_ballot_w64(lossy) == 0ull) {                                                      \
                _Pragma("unroll") for (int kg = 0; kg < KG; ++kg)                                                  \
                    _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) b[kg][cb] = ldexp(b[kg][cb], (s_new));       \
                sc = (s_new);                                                                                      \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
        BI_LOAD_ROWS();
        // the strip's counts: one per lane; a strip of ONE count (the rule in count order) is class U (n a positive
        // integer) or Z (n = 0); anything else -- two runs meeting, negative / non-integer / nan counts -- is worked
        // item by item
        const double c_lane = cnt[bin0 + lane];
        s_cnt[wave][lane] = c_lane;                 // for the slow path: a lane needs the counts of ITS bins (4 r + kq of every block)
        __builtin_amdgcn_wave_barrier();
        const bool strip_odd = __builtin_amdgcn_ballot_w64(c_lane != c_lane || c_lane < 0.0 || c_lane != floor(c_lane)) != 0ull;
        const double n_strip = lane_value(c_lane, 0);
        const bool one_count = __builtin_amdgcn_ballot_w64(c_lane == n_strip) == ~0ull;
        // 1 = Z, 2 = U, 0 = item by item -- as a scalar, so that the tests inside the item loop are scalar branches
        const int cls = __builtin_amdgcn_readfirstlane(
            !one_count ? 0 : (n_strip == 0.0 ? 1 : ((n_strip > 0.0 && n_strip == floor(n_strip) && n_strip < 0x1p52) ? 2 : 0)));
        const bool cls_z = cls == 1, cls_u = cls == 2, fast = cls != 0;
        sc = 0;
        if (cls_u && sc_learnt != 0) BI_SCALE_ROWS(sc_learnt);
        // Blocks of 16 bins with one count each, for the slow path (a lane's four bins of a block then take one logarithm)
        unsigned blk_uniform = 0;
        double n_blk[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            n_blk[cb] = lane_value(c_lane, cb * 16);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(c_lane == n_blk[cb]);
            const bool whole = ((m >> (16 * cb)) & 0xFFFFull) == 0xFFFFull;
            if (whole && n_blk[cb] > 0.0 && n_blk[cb] == floor(n_blk[cb])) blk_uniform |= 1u << cb;
        }
        blk_uniform = (unsigned)__builtin_amdgcn_readfirstlane((int)blk_uniform);

        double av[KG];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) av[kg] = coef0[loff[kg]];

        int it = 0;
        while (true) {
            int lo = i_first, hi = n_items;
            if (lo >= hi) break;
            if (fast) {
                // one loop per class (compile-time Z): the class test stays out of the item loop
                auto quads = [&](auto ztag) -> int {
                    constexpr bool Z = decltype(ztag)::value;
                    for (; it < n_items; it += 4) {
                        double p[4];
                        unsigned long long okm = ~0ull;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            bi_double4 acc[CB];
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) acc[cb] = bi_double4{0.0, 0.0, 0.0, 0.0};
                            // the next item's coefficients go straight into the registers the chains have just read (the
                            // last item of the group is read again behind the end: its results are masked out below); every
                            // load is pinned behind the last MFMA that reads its register, so that it has the rest of the
                            // chains and the epilogue to arrive
                            const double* __restrict__ cf = coef0 + (int64_t)min(it + j + 1, n_items - 1) * coef_step;
#pragma unroll
                            for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
                                for (int cb = 0; cb < CB; ++cb)
                                    acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[kg][cb], av[kg], acc[cb], 0, 0, 0);
                                av[kg] = cf[loff[kg]];
                                __builtin_amdgcn_sched_group_barrier(0x008, CB, 0);     // CB MFMAs
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one VMEM read
                            }
                            if constexpr (Z) {
                                // empty bins: the term is 0 unless the expectation is negative or nan
#pragma unroll
                                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) okm &= __builtin_amdgcn_ballot_w64(acc[cb][r] >= 0.0);
                            } else {
                                int h[16];
#pragma unroll
                                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) h[cb * 4 + r] = __double2hiint(acc[cb][r]);
#define BI_MIN3(x, y, z) min(min((x), (y)), (z))
                                const int t0 = BI_MIN3(h[0], h[1], h[2]), t1 = BI_MIN3(h[3], h[4], h[5]), t2 = BI_MIN3(h[6], h[7], h[8]);
                                const int t3 = BI_MIN3(h[9], h[10], h[11]), t4 = BI_MIN3(h[12], h[13], h[14]);
                                const int mm = min(BI_MIN3(t0, t1, t2), BI_MIN3(t3, t4, h[15]));
#undef BI_MIN3
                                double q[CB];
#pragma unroll
                                for (int cb = 0; cb < CB; ++cb) q[cb] = (acc[cb][0] * acc[cb][1]) * (acc[cb][2] * acc[cb][3]);
                                // (eight factors per product: q[0] q[1] and q[2] q[3] are normal numbers when mm passes)
                                const double pl = (q[0] * q[1]) * (q[2] * q[3]);
                                okm &= __builtin_amdgcn_ballot_w64(mm >= kFloorHi) & __builtin_amdgcn_ballot_w64(pl > kLaneLo) &
                                       __builtin_amdgcn_ballot_w64(pl < kLaneHi);
                                p[j] = pl;
                            }
                        }
                        if (okm != ~0ull) return it;
                        if constexpr (!Z) {
                            // row q <- the product of item q over the four rows (lanes l, l ^ 16, l ^ 32, l ^ 48)
                            double x, y, P;
                            BI_SWAP_MUL(__builtin_amdgcn_permlane32_swap, p[0], p[2], x);
                            BI_SWAP_MUL(__builtin_amdgcn_permlane32_swap, p[1], p[3], y);
                            BI_SWAP_MUL(__builtin_amdgcn_permlane16_swap, x, y, P);
                            const double term = n_strip * log_core_s(P, -STRIP * sc);
                            if (it + kq < n_items) unsafeAtomicAdd(part0 + (int64_t)it * dst_step + qoff, term);
                        }
                    }
                    return -1;
                };
                const int fail_at = cls_z ? quads(std::true_type{}) : quads(std::false_type{});
                if (fail_at < 0) break;
                lo = fail_at;
                hi = min(fail_at + 4, n_items);
            }
            // ---- slow path: items [lo, hi) one by one on the true rows, scipy's value for every argument ----------
            if (sc != 0) BI_LOAD_ROWS();
            const int sc_old = sc;
            sc = 0;
            double first = __builtin_nan("");
            double sv[KG];
            {
                const double* __restrict__ cf = coef0 + (int64_t)lo * coef_step;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) sv[kg] = cf[loff[kg]];
            }
            const int step = fast ? 1 : i_step;              // (a quad that left the product form: its four items; a mixed strip: this wave's share)
            for (int i = lo; i < hi; i += step) {
                bi_double4 acc[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb] = bi_double4{0.0, 0.0, 0.0, 0.0};
                // (the next item's coefficients behind the chains, as in the fast loop)
                const double* __restrict__ cf = coef0 + (int64_t)(i + step < hi ? i + step : i) * coef_step;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[kg][cb], sv[kg], acc[cb], 0, 0, 0);
                    sv[kg] = cf[loff[kg]];
                    __builtin_amdgcn_sched_group_barrier(0x008, CB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                double tot = 0.0;
                bool invalid = false;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) invalid |= !(acc[cb][r] >= 0.0);
                    if (blk_uniform & (1u << cb)) {          // (wave-uniform) one count in the block: n log of the lane's product
                        const bool ok = acc[cb][0] > kProdFloor && acc[cb][1] > kProdFloor && acc[cb][2] > kProdFloor && acc[cb][3] > kProdFloor;
                        const double q = (acc[cb][0] * acc[cb][1]) * (acc[cb][2] * acc[cb][3]);
                        if (__builtin_amdgcn_ballot_w64(ok && pos_normal(q)) == ~0ull) {
                            tot += n_blk[cb] * bin_log_fast(q);
                            continue;
                        }
                    }
                    double n[4];
                    bool checked = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        n[r] = s_cnt[wave][cb * 16 + 4 * r + kq];
                        checked |= n[r] > 0.0 && !pos_normal(acc[cb][r]);
                    }
                    if (__builtin_amdgcn_ballot_w64(checked) == 0ull) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double lg = bin_log_fast(acc[cb][r]);
                            tot += n[r] > 0.0 ? n[r] * lg : 0.0;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) tot += n[r] > 0.0 ? n[r] * bin_log(acc[cb][r]) : 0.0;
                    }
                    if (strip_odd) {                         // (wave-uniform) negative / non-integer / nan counts in this strip
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (n[r] != n[r]) tot = __builtin_nan("");
                            else if (n[r] < 0.0 || n[r] != floor(n[r])) tot += -__builtin_inf();
                        }
                    }
                }
                if (invalid) tot = __builtin_nan("");
                tot = rows4_sum(tot);
                if (kq == 0) unsafeAtomicAdd(part0 + (int64_t)i * dst_step + col, tot);
                if (i == lo) first = tot;
            }
            if (!fast) break;
            // A quad left the window of the product form: centre the window on this strip's own terms (the first point of
            // the first item: sum over the strip's 64 bins of n log mu) for what follows
            if (cls_u) {
                const double mean_log2 = first / (n_strip * (double)STRIP * 0.6931471805599453);
                const double m0 = __longlong_as_double(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(mean_log2) >> 32)) << 32) |
                                                       (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(mean_log2)));
                int s_new = sc_old;
                if (m0 == m0 && fabs(m0) < 2000.0) s_new = fabs(m0) <= 12.0 ? 0 : max(-kScaleMax, min(kScaleMax, -(int)rint(m0)));
                sc_learnt = s_new;
                BI_SCALE_ROWS(s_new);
            }
            it = hi;
            if (it >= n_items) break;
            // (the fast loop had asked for the coefficients of item `hi` before it gave up; they are read again rather than
            // kept in registers through the slow path)
            {
                const double* __restrict__ cf = coef0 + (int64_t)it * coef_step;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) av[kg] = cf[loff[kg]];
            }
        }
    }
#undef BI_LOAD_ROWS
#undef BI_SCALE_ROWS
}

}  // namespace
