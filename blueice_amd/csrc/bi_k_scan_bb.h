// bi_k_scan_bb.h -- Beeston-Barlow scans on the fp64 matrix cores: k_scan_bb<KGT> (round 5).  Translation unit tu_scan_bb.hip.
//
// What it replaces.  A Beeston-Barlow evaluation (blueice/likelihood.py:618-660, roots :693-712) reads 2^d (S + 1) template rows
// per bin -- the other sources' rows into U_b, the Beeston-Barlow source's own rows into P_b, its Monte-Carlo counts into a_b --
// and k_morph_reduce<G, true> shares one pass over them among G = 8 points of a grid cell (16 spill): a 256-point scan of one
// cell of configs[4] was 32 passes of 5.65 GB, 1.02 ms each.  Here the rows of a bin tile are staged ONCE in LDS for the four
// waves of a block, each wave holds the coefficients of ONE 16-point work item in registers (the MFMA's B operand) for the
// whole kernel, and
//     U[bin][point] = rowsU x coefU,   P[bin][point] = rowsP x w,   a[bin][point] = rowsA x w
// are three chains of v_mfma_f64_16x16x4 per 16-bin tile -- the template tile is the A operand (row = bin), as in the scan
// kernels, so a lane's four accumulator elements are four bins of ONE point.  The rows are read from HBM once per 64 points
// instead of once per 8, the per-bin work -- w = P / a N, the root formula in the reference's operation order without
// contraction, the two assertions as status bits, mu = U + (A w) p_cal, the Poisson term -- runs on the vector ALU on the
// accumulators where they are.
// U, P and a come out of fused multiply-adds here (k_morph_reduce interpolates P and a with separate multiply and add, the
// reference's own bits): the kernel is only given batches in which no bin can have U_b == 0 -- there the first root's sign
// hangs on those last bits, everywhere else the assertions are far from their edge -- which is what
// the device planner checks anyway (kPlanNeedsHost).
// Streams: n0 = 2^d (S - 1) into U, nc = 2^d into P, nc into a; each segment padded to whole groups of four streams (zero
// rows), KGT >= ceil(n0 / 4) + 2 ceil(nc / 4) groups in all (template: 4, 8, ... 32).
#pragma once

namespace {

constexpr int kBbTile = 16;                      // bins per tile = one MFMA block

// KGU > 0: the segments are compile-time -- KGU groups into U, KGP into P, KGP into a (models with 4 KGP = 2^d corners) --
// and a tile's chains are straight-line code: all LDS reads of the A operands go out ahead of the MFMAs that use them.
// KGU == 0: KGT groups in all with run-time segment boundaries (any other shape): the same arithmetic, scalar branches between
// the groups.
template <int KGU, int KGP, int KGT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2))) void k_scan_bb(BbScanArgs a) {
    static_assert(kThreads == 256, "four waves per block: four work items share the staged rows");
    static_assert(KGU == 0 || KGT == KGU + 2 * KGP, "static segments fill the variant");
    constexpr bool STATIC = KGU > 0;
    const int grp = blockIdx.y, quad = blockIdx.z;
    const int n_items = a.grp_items[grp];
    if (quad * 4 >= n_items) return;                                   // (whole block: before any barrier)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, col = lane & 15;
    const int64_t item0 = a.grp_first[grp];
    const int64_t item = item0 + quad * 4 + wave;
    const bool active = quad * 4 + wave < n_items;                      // (wave-uniform) idle waves still stage rows
    const int n0 = a.n0, nc = a.nc, NS = n0 + 2 * nc;
    const int kgu = STATIC ? KGU : (n0 + 3) >> 2, kgp = STATIC ? KGP : (nc + 3) >> 2;   // groups of four streams per segment
    // padded row r = 4 kg + kq  ->  stream of the item's [NS] lists, or -1 (a zero row)
    auto stream_of = [&](int r) -> int {
        const int kg = r >> 2, q = r & 3;
        if (kg < kgu) { const int s = kg * 4 + q; return s < n0 ? s : -1; }
        if (kg < kgu + kgp) { const int c = (kg - kgu) * 4 + q; return c < nc ? n0 + c : -1; }
        if (kg < kgu + 2 * kgp) { const int c = (kg - kgu - kgp) * 4 + q; return c < nc ? n0 + nc + c : -1; }
        return -1;
    };
    __shared__ double s_rows[2][KGT * 4 + 1][kBbTile];                  // (+ the counts of the tile's bins)
    log_table_load();
    // ---- this wave's work item: coefficients (B operand: k = kq, column = point col) and the per-point constants ----
    double bco[KGT];
#pragma unroll
    for (int kg = 0; kg < KGT; ++kg) {
        const int s = stream_of(kg * 4 + kq);
        bco[kg] = (active && s >= 0) ? a.coef[(item * NS + s) * 16 + col] : 0.0;
    }
    const double p_cal = active ? a.aux[(item * 16 + col) * 2 + 0] : 1.0;
    const double Ntot = active ? a.aux[(item * 16 + col) * 2 + 1] : 1.0;
    // ---- staging: thread t fetches elements t, t + 256, ... of the tile's [KGT * 4 + 1][16] block; sources fixed per thread.
    // Every load is unconditional (a zero row reads the group's first row and is masked when it is put into LDS): a branch
    // around a load makes the compiler wait for everything outstanding ----
    constexpr int kElems = (KGT * 4 + 1) * kBbTile;
    constexpr int kPer = (kElems + kThreads - 1) / kThreads;
    const double* src[kPer];
    unsigned live = 0u;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int e = min((int)threadIdx.x + j * kThreads, kElems - 1);
        const int r = e / kBbTile, b = e % kBbTile;
        const double* p = a.ps + a.rowoff[item0 * NS] + b;
        if (r == KGT * 4) { p = a.counts + a.item_cnt[item0] + b; live |= 1u << j; }
        else {
            const int s = stream_of(r);
            if (s >= 0) { p = (s >= n0 + nc ? a.nm : a.ps) + a.rowoff[item0 * NS + s] + b; live |= 1u << j; }
        }
        src[j] = p;
    }
    const int tiles = a.n_tiles;                                        // tiles of 16 bins covering [0, B)
    const int per = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per, t_end = min(tiles, t_begin + per);
    double stage[kPer];
    auto fetch = [&](int tile) {
        const int64_t bin0 = (int64_t)tile * kBbTile;
        // (rows are padded with zeros up to Bp, a multiple of 512 bins: in bounds; bins >= B are masked in the epilogue)
#pragma unroll
        for (int j = 0; j < kPer; ++j) stage[j] = __builtin_nontemporal_load(src[j] + bin0);
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int e = threadIdx.x + j * kThreads;
            if (e < kElems) (&s_rows[buf][0][0])[e] = ((live >> j) & 1u) ? stage[j] : 0.0;
        }
    };
    double sum = 0.0;
    unsigned flg = 0u;
    if (t_begin < t_end) {
        fetch(t_begin);
        put(0);
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        if (tile + 1 < t_end) fetch(tile + 1);                          // the next tile's loads fly under this tile's chains
        if (active) {
            bi_double4 aU = bi_double4{0.0, 0.0, 0.0, 0.0}, aP = aU, aA = aU;
            if constexpr (STATIC) {
                double av[KGT];
#pragma unroll
                for (int kg = 0; kg < KGT; ++kg) av[kg] = s_rows[buf][kg * 4 + kq][col];
                // (P and a first: their short chains finish under the long one, and the epilogue's first divisions can start)
#pragma unroll
                for (int kg = KGU; kg < KGU + KGP; ++kg) aP = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kg], bco[kg], aP, 0, 0, 0);
#pragma unroll
                for (int kg = KGU + KGP; kg < KGT; ++kg) aA = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kg], bco[kg], aA, 0, 0, 0);
#pragma unroll
                for (int kg = 0; kg < KGU; ++kg) aU = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kg], bco[kg], aU, 0, 0, 0);
            } else {
#pragma unroll
                for (int kg = 0; kg < KGT; ++kg) {
                    if (kg < kgu + 2 * kgp) {                           // (scalar tests: the segment a group belongs to)
                        const double av = s_rows[buf][kg * 4 + kq][col];
                        if (kg < kgu) aU = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bco[kg], aU, 0, 0, 0);
                        else if (kg < kgu + kgp) aP = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bco[kg], aP, 0, 0, 0);
                        else aA = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bco[kg], aA, 0, 0, 0);
                    }
                }
            }
            const int64_t bin0 = (int64_t)tile * kBbTile;
            // The four elements of a lane side by side, without a branch: their divisions and square roots overlap.  Same
            // operations in the same order as morph_tiles<.., BB> (likelihood.py:645-646, :693-712, :649-658), minus what cannot
            // change a bit of the result:
            //   * the first root is only ever tested for `r1 <= 0` (likelihood.py:649): with lead - sqrt(disc) <= 0 and a positive
            //     finite denominator the quotient is <= 0 whatever its value, so the division is left to the (wave-uniform) case
            //     where that quick test does not settle it;
            //   * the special case of bins with U_b == 0 (:652-653) costs a division: computed only when some lane has such a bin;
            //   * the logarithm's checked form (zero, subnormal, negative, infinite, nan arguments) only when some lane needs it.
            const bool whole = bin0 + kBbTile <= a.B;                      // (scalar) every bin of the tile is a bin of the model
            //   * a tile WITHOUT events (all 16 counts zero: 98 % of the tiles of configs[4]'s data) -- wave-uniform: the
            //     discriminant's four terms with the count as a factor are +0 (finite operands), and x - 0 = x + 0 = x bit for
            //     bit (the partial sums are positive), likewise `+ n p` of the leading term; and the bin's Poisson term is -mu
            //     without a logarithm (scipy: xlogy(0, mu) - gammaln(1) - mu).  With a non-finite p (a bin without Monte-Carlo
            //     events: a = 0) the skipped 0 x inf would have been nan -- where another term already is (2 U a p^2 = 0 x inf).
            double n4[4], mu4[4], A4[4], lead4[4], sq4[4], den4[4];
            bool uzero = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) n4[r] = s_rows[buf][KGT * 4][4 * r + kq];   // element r of this lane: bin 4 r + kq of the tile, point col
            const bool empty_tile = __builtin_amdgcn_ballot_w64(n4[0] == 0.0 && n4[1] == 0.0 && n4[2] == 0.0 && n4[3] == 0.0) == ~0ull;
            if (empty_tile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double U = aU[r], ab = aA[r];
                    const double w = aP[r] / ab * Ntot;
                    const double p = w * p_cal;
                    double lead, sq, den;
                    {
#pragma clang fp contract(off)
                        const double U2 = U * U, p2 = p * p, a2 = ab * ab;
                        const double disc = U2 * p2 + 2 * U2 * p + U2 + 2 * U * ab * p2 + 2 * U * ab * p + a2 * p2;
                        lead = -U * p - U + ab * p;
                        den = 2 * p * (p + 1);
                        sq = sqrt(disc);
                    }
                    lead4[r] = lead; sq4[r] = sq; den4[r] = den;
                    A4[r] = (lead + sq) / den;
                    uzero |= (U == 0.0);
                    mu4[r] = w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double n = n4[r];
                    const double U = aU[r], ab = aA[r];
                    const double w = aP[r] / ab * Ntot;
                    const double p = w * p_cal;
                    double lead, sq, den;
                    {
#pragma clang fp contract(off)
                        const double U2 = U * U, p2 = p * p, a2 = ab * ab, d2 = n * n;
                        const double disc = U2 * p2 + 2 * U2 * p + U2 + 2 * U * ab * p2 + 2 * U * ab * p - 2 * U * n * p2 - 2 * U * n * p +
                                            a2 * p2 + 2 * ab * n * p2 + d2 * p2;
                        lead = -U * p - U + ab * p + n * p;
                        den = 2 * p * (p + 1);
                        sq = sqrt(disc);
                    }
                    const double r2 = (lead + sq) / den;
                    lead4[r] = lead; sq4[r] = sq; den4[r] = den;
                    A4[r] = r2;
                    uzero |= (U == 0.0);
                    mu4[r] = w;                                              // (w for now: mu once A is final)
                }
            }
            if (__builtin_amdgcn_ballot_w64(uzero) != 0ull) {              // likelihood.py:652-653, scalar p_cal
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (aU[r] == 0.0) A4[r] = (n4[r] + aA[r]) / (1.0 + p_cal);
            }
            unsigned f4 = 0u;
            bool r1_open = false, checked = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = whole || bin0 + 4 * r + kq < a.B;
                const double t1 = lead4[r] - sq4[r];
                // r1 = t1 / den <= 0 for certain when t1 <= 0 and 0 < den < inf (nan operands fail both tests)
                r1_open |= in && !(t1 <= 0.0 && den4[r] > 0.0 && den4[r] < __builtin_inf());
                if (in && !(0.0 <= A4[r])) f4 |= BI_ST_BB_NEG;
                mu4[r] = aU[r] + (A4[r] * mu4[r]) * p_cal;
                checked |= in && needs_checked_term(n4[r], mu4[r]);
            }
            if (__builtin_amdgcn_ballot_w64(r1_open) != 0ull) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool in = whole || bin0 + 4 * r + kq < a.B;
                    const double r1 = (lead4[r] - sq4[r]) / den4[r];
                    if (in && !(r1 <= 0.0)) f4 |= BI_ST_BB_ROOT1;
                }
            }
            double term[4];
            if (empty_tile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) term[r] = poisson_term_nolog(n4[r], mu4[r]);
            } else if (__builtin_amdgcn_ballot_w64(checked) == 0ull) {
#pragma unroll
                for (int r = 0; r < 4; ++r) term[r] = poisson_term_fast(n4[r], mu4[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) term[r] = poisson_term(n4[r], mu4[r]);
            }
            if (!whole) {
#pragma unroll
                for (int r = 0; r < 4; ++r) term[r] = bin0 + 4 * r + kq < a.B ? term[r] : 0.0;
            }
            sum += (term[0] + term[1]) + (term[2] + term[3]);
            flg |= f4;
        }
        if (tile + 1 < t_end) put(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    // a lane's sum belongs to point col: add the four rows (bins 4 r + kq) and hand the block's partial over
    sum = rows4_sum(sum);
    flg |= __shfl_xor(flg, 16, 64);
    flg |= __shfl_xor(flg, 32, 64);
    if (kq == 0) {
        const int64_t o = (item * gridDim.x + blockIdx.x) * 16 + col;
        a.partial[o] = sum;
        a.pflags[o] = flg;
    }
}

}  // namespace
