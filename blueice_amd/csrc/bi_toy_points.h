// bi_toy_points.h -- the toy-MC form over SEVERAL parameter points: bi_eval_datasets_points (round 5).
//
// What it replaces.  The reference's toy-MC users evaluate every toy dataset at many hypotheses -- the loops of
// blueice/inference.py:392-443 (one likelihood call per hypothesis and dataset) around blueice/model.py:69-91 (simulate, set_data).
// bi_eval_datasets takes ONE point: P hypotheses were P calls, each streaming the T datasets' non-empty-bin lists (188 MB at
// BASELINE.json configs[2]) and each paying its own log mu pass (264 MB), its own three launches and its own wait.  Here:
//   (1) k_morph_logmu_multi<GC>: the points are ordered by grid cell and chopped into PASSES of PP = 4 (or 2) points, the passes
//       into groups of two; the points of a GROUP (GC = 2 PP) that share a cell share ONE pass over the cell's 2^d S template
//       rows (hypotheses that differ in their rates only -- the usual signal-strength scan -- always do); log mu is written one
//       row per point, lm[pass][PP][Bp] (coalesced stores whatever the cells; the dot kernel interleaves the PP rows of its tile
//       as it stages them);
//   (2) k_dataset_dot_multi<L, AHEAD, W, PP>: ONE pass over the tile-major entry lists per pass of points: a block stages the
//       log mu of its bin tile for all PP points side by side in LDS (4096 bins x 4 points x 8 B = 128 KB of the CU's 160 KB) and
//       an entry's one LDS address yields PP values (PP / 2 ds_read_b128): the 188 MB entry stream, the offsets, the decode and
//       the run bookkeeping are paid once per PP evaluations of a dataset;
//   (3) k_dataset_finish_multi: per (point, dataset) the tiles' partial sums in tile order, minus sum mu of the point and
//       sum lgamma of the dataset -- a fixed order, bitwise reproducible -- into out[point][dataset] in the caller's point order.
// Algorithmic bytes per call: 8 (2^d S) B per distinct grid cell of a group of passes + 8 PP B (log mu written and staged once) per pass
// + W bytes per list entry per PASS (not per point).
// Included by blueice_hip.hip (main translation unit) behind the single-point toy-MC form.
#pragma once

namespace {

constexpr int kDotTileMulti = 4096;          // bins per tile of the multi-point lists: 4096 x 4 points x 8 B = 128 KB of LDS
constexpr int kToyPointsMaxPP = 4;
constexpr int kPartBlock = 32;               // datasets per block of the per-(tile, dataset) partial sums

// ---- (1) log mu of the points of a group of passes ---------------------------------------------------------------------------
// The table of a group is GC = (passes per group) x PP rows of Bp doubles, row = pass-in-group * PP + column-in-pass.
// blockIdx.y = work item = the points of ONE group that lie in ONE grid cell (1 .. GC of them: the rows of the cell's anchors are
// read once for all of them, across the passes of the group); blockIdx.x strides over the 512-bin tiles in the XCD-aware order
// of morph_tiles.  coef [items][NS][GC]: column g of the matrix is row g of the group's table, zero for the rows that belong to
// another item (another cell); meta [items][4] = {first row, number of rows, 0, 0}.  Same accumulation order over the streams as
// k_morph_logmu (fma, k ascending): the same log mu bits.
// partial / pflags [items][gridDim.x][GC]: sum_b mu_b and the "some mu is negative or nan" flag per row.
template <int GC, bool NT>
__global__ __launch_bounds__(kThreads) void k_morph_logmu_multi(LaunchArgs a, const int32_t* __restrict__ meta, double* __restrict__ lm) {
    const int item = blockIdx.y;
    const int NS = a.n0;
    const int64_t* __restrict__ rowoff = a.rowoff + (int64_t)item * NS;
    const double* __restrict__ coef = a.coef + (int64_t)item * NS * GC;
    const int col0 = meta[4 * item], ncol = meta[4 * item + 1];
    double* __restrict__ out = lm;                                    // row g of the group's table: out + g * Bp
    double sum[GC];
    unsigned bad[GC];
#pragma unroll
    for (int g = 0; g < GC; ++g) { sum[g] = 0.0; bad[g] = 0u; }
    log_table_load();
    const int n_tiles = a.n_tiles;
    const int chunks = (a.chunks > 1 && n_tiles >= 64 * a.chunks) ? a.chunks : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    if (ncol == 1) {
        // (block-uniform) a point alone in its cell: one column -- the other GC - 1 accumulators would be most of the
        // kernel's fp64 work for nothing (random cells: 5.1 -> 6.x TB/s); the same operations in the same order
        double s1 = 0.0;
        unsigned b1 = 0u;
        for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
            const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
            if (tile >= n_tiles) continue;
            const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
            double m0 = 0.0, m1 = 0.0;
#pragma unroll 8
            for (int k = 0; k < NS; ++k) {
                const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
                const double c = coef[k * GC + col0];
                m0 = fma(c, v.x, m0);
                m1 = fma(c, v.y, m1);
            }
            double2 w;
            w.x = (m0 >= 0.0) ? bin_log(m0) : __builtin_nan("");
            w.y = (m1 >= 0.0) ? bin_log(m1) : __builtin_nan("");
            *reinterpret_cast<double2*>(out + (int64_t)col0 * a.Bp + bin0) = w;
            if (!(m0 >= 0.0) || !(m1 >= 0.0)) b1 = 1u;
            s1 += m0 + m1;
        }
#pragma unroll
        for (int g = 0; g < GC; ++g)
            if (g == col0) { sum[g] = s1; bad[g] = b1; }
    } else
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double acc[GC][2];
#pragma unroll
        for (int g = 0; g < GC; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }
#pragma unroll 8
        for (int k = 0; k < NS; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
            for (int g = 0; g < GC; ++g) {
                const double c = coef[k * GC + g];
                acc[g][0] = fma(c, v.x, acc[g][0]);
                acc[g][1] = fma(c, v.y, acc[g][1]);
            }
        }
        double l[2][GC];
#pragma unroll
        for (int g = 0; g < GC; ++g) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double m = acc[g][j];
                l[j][g] = (m >= 0.0) ? bin_log(m) : __builtin_nan("");
                if (!(m >= 0.0)) bad[g] = 1u;
            }
            sum[g] += acc[g][0] + acc[g][1];
        }
#pragma unroll
        for (int g = 0; g < GC; ++g)
            if (g >= col0 && g < col0 + ncol) {                       // (block-uniform) the columns of this item
                double2 w;
                w.x = l[0][g];
                w.y = l[1][g];
                *reinterpret_cast<double2*>(out + (int64_t)g * a.Bp + bin0) = w;
            }
    }
    __shared__ double sh[kThreads / 64][GC];
    __shared__ unsigned shf[kThreads / 64][GC];
#pragma unroll
    for (int g = 0; g < GC; ++g) {
        const double s = wave_sum(sum[g]);
        const unsigned f = wave_or(bad[g]);
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][g] = s; shf[threadIdx.x >> 6][g] = f; }
    }
    __syncthreads();
    if (threadIdx.x < GC) {
        const int g = threadIdx.x;
        double s = sh[0][g];
        unsigned f = shf[0][g];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += sh[w][g]; f |= shf[w][g]; }
        const int64_t o = ((int64_t)item * gridDim.x + blockIdx.x) * GC + g;
        a.partial[o] = s;
        a.pflags[o] = f;
    }
}

// ---- (2) the datasets' entry lists against the log mu tiles of PP points ----------------------------------------------------
// k_dataset_dot_tiled's pipeline (rings of offsets and entries in registers, every load unconditional, runs of whole 16-byte
// groups: see there) with PP accumulators per lane.  blockIdx.x = bin tile of TB bins, blockIdx.y = slice of the datasets,
// blockIdx.z = pass.  LDS: PP / 2 planes of (TB + 1) 16-byte cells (dynamic: more than 64 KB needs the function attribute); the
// extra cell behind a plane holds 0.0 for the padding entries.
// partial [passes][dataset blocks of kPartBlock][n_tl][kPartBlock][PP].
// The block of tile 0 / slice 0 of every pass first adds up sum mu (and the "some mu negative or nan" flags) of the pass's PP points
// from the block partials of their log mu work items -- a fixed order: thread-strided, wave, the 16 waves in turn -- into
// MuTotals::tot / flag [pass in group][PP]: the finish kernel reads PP numbers instead of every one of its blocks adding them up.
struct MuTotals {
    const int32_t* colmap;        // [passes of the group][PP][2] = {work item of the log mu kernel, output row} (-1: no point)
    const double* partial;        // [items][nmu][GC]
    const unsigned* flags;
    int nmu, GC;
    double* tot;
    unsigned* flag;
};

template <int L, int AHEAD, int W, int PP, int TB>
__global__ __launch_bounds__(kDotThreads) void k_dataset_dot_multi(const void* __restrict__ tm_entries_v, const int64_t* __restrict__ tm_off,
                                                                   int64_t T, int n_tl, const double* __restrict__ lm, int64_t B,
                                                                   int64_t Bp, int64_t t0, int64_t n, double* __restrict__ partial, MuTotals mt) {
    typedef typename std::conditional<W == 2, uint16_t, uint32_t>::type entry_t;
    constexpr int EPL = 16 / W;                                        // entries per lane and load
    static_assert(EPL * L * AHEAD + 16 <= kDotPad, "the lists' padding must cover the read-ahead");
    static_assert(PP == 2 || PP == 4, "two or four points per pass");
    // LDS layout: planes of 16-byte cells, one cell per bin -- plane 0 holds (point 0, point 1) of every bin, plane 1 (point 2,
    // point 3) -- so that a wave's 16-byte reads of RANDOM bins spread over all eight 16-byte bank groups (bin & 7).  Round 5's
    // first layout kept a bin's four values adjacent (32 bytes): each of the two ds_read_b128 of an entry then only ever touched
    // every other bank group, and the kernel -- bound by LDS bank conflicts, 7.9 conflict cycles per LDS instruction on top of
    // its 8 -- ran 120 us per pass of four points.
    constexpr int kPlane = (TB + 1) * 16;                              // bytes per plane (+ the zero cell of the padding entries)
    const entry_t* __restrict__ tm_entries = static_cast<const entry_t*>(tm_entries_v);
    extern __shared__ double s_mu[];
    if (blockIdx.x == 0 && blockIdx.y == 0) {                          // (block-uniform)
        __shared__ double sh_m[kDotThreads / 64][PP];
        __shared__ unsigned sh_f[kDotThreads / 64][PP];
        const int pz = blockIdx.z;
#pragma unroll
        for (int g = 0; g < PP; ++g) {
            const int gc = pz * PP + g;
            const int item = mt.colmap[2 * gc];
            double m = 0.0;
            unsigned f = 0u;
            if (item >= 0) {
                const double* __restrict__ mp = mt.partial + (int64_t)item * mt.nmu * mt.GC + gc;
                const unsigned* __restrict__ mf = mt.flags + (int64_t)item * mt.nmu * mt.GC + gc;
                for (int b = threadIdx.x; b < mt.nmu; b += kDotThreads) { m += mp[(int64_t)b * mt.GC]; f |= mf[(int64_t)b * mt.GC]; }
            }
            const double mw = wave_sum(m);
            const unsigned fw = wave_or(f);
            if ((threadIdx.x & 63) == 0) { sh_m[threadIdx.x >> 6][g] = mw; sh_f[threadIdx.x >> 6][g] = fw; }
        }
        __syncthreads();
        if (threadIdx.x < PP) {
            double mm = 0.0;
            unsigned ff = 0u;
#pragma unroll
            for (int w = 0; w < kDotThreads / 64; ++w) { mm += sh_m[w][threadIdx.x]; ff |= sh_f[w][threadIdx.x]; }
            mt.tot[pz * PP + threadIdx.x] = mm;
            mt.flag[pz * PP + threadIdx.x] = ff;
        }
    }
    const int tl = blockIdx.x;
    const int pass = blockIdx.z;
    const int64_t bin0 = (int64_t)tl * TB;
    const int row = threadIdx.x / L, gl = threadIdx.x % L;
    const int per = (int)((n + gridDim.y - 1) / gridDim.y);
    const int c0u = (int)blockIdx.y * per, c1 = min((int)n, c0u + per);
    const int c0 = min(c0u, (int)n - 1);
    const int64_t* __restrict__ off = tm_off + (int64_t)tl * T + t0;
    const int64_t base = off[c0];
    const entry_t* __restrict__ ent = tm_entries + base;
    const uint32_t* __restrict__ off32 = reinterpret_cast<const uint32_t*>(off);
    const uint32_t base32 = (uint32_t)base;
    constexpr int kStep = kDotThreads / L, kAhead = AHEAD, kDepth = BI_DOT_DEPTH, kRing = 2 * (kDepth + 1);
    const int q_last = max(c0, c1 - 1);
    auto load_offsets = [&](int q, uint32_t& ra, uint32_t& rb) {
        const int qc = min(q, q_last);
        ra = off32[2 * qc];
        rb = off32[2 * qc + 2];
    };
    auto load_entries = [&](uint32_t ra, bi_uint4 (&e)[kAhead]) {
        const entry_t* __restrict__ p = ent + (int)(ra - base32) + EPL * gl;
#pragma unroll
        for (int k = 0; k < kAhead; ++k) __builtin_memcpy(&e[k], p + EPL * L * k, 16);
    };
    bi_uint4 E[kDepth + 1][kAhead];
    uint32_t RA[kRing], RB[kRing];
    const int q0 = c0 + row;
#pragma unroll
    for (int u = 0; u < 2 * kDepth; ++u) load_offsets(q0 + u * kStep, RA[u], RB[u]);
    {   // stage the tile: TB bins of each of the pass's PP rows, interleaved in LDS so that a bin's PP values are adjacent (all
        // loads first, addresses clamped instead of predicated; bins beyond the table read as 0.0: never addressed by an entry)
        const double* __restrict__ src = lm + (int64_t)pass * PP * Bp + bin0;
        const int avail = (int)(min(bin0 + TB, Bp) - bin0);           // (the table is Bp bins long, Bp a multiple of 512)
        constexpr int kPer = TB / 2 / kDotThreads;                    // 16-byte pieces per thread and row
        double2 v[PP][kPer];
#pragma unroll
        for (int g = 0; g < PP; ++g)
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i2 = (threadIdx.x + k * kDotThreads) * 2;
                v[g][k] = *reinterpret_cast<const double2*>(src + (int64_t)g * Bp + min(i2, avail - 2));
            }
#pragma unroll
        for (int g = 0; g < PP; ++g)
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i2 = (threadIdx.x + k * kDotThreads) * 2;
                const bool in = i2 < avail;
                double* __restrict__ cell = s_mu + (g >> 1) * (kPlane / 8) + (g & 1);
                cell[i2 * 2] = in ? v[g][k].x : 0.0;
                cell[(i2 + 1) * 2] = in ? v[g][k].y : 0.0;
            }
        if (threadIdx.x < PP) s_mu[(threadIdx.x >> 1) * (kPlane / 8) + TB * 2 + (threadIdx.x & 1)] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < kDepth; ++u) load_entries(RA[u], E[u]);
    __syncthreads();
    if (c0u >= c1) return;
    // partial sums in blocks of kPartBlock datasets, [pass][dataset block][tile][dataset in block][PP]: the finish of a dataset
    // block reads ONE contiguous range (all tiles of its datasets), the dot kernel writes whole 1 KB pieces of it
    const int64_t n_pad = (n + kPartBlock - 1) / kPartBlock * kPartBlock;
    double* __restrict__ pout = partial + (int64_t)pass * n_tl * n_pad * PP + (int64_t)tl * kPartBlock * PP;
    const int n_iter = (c1 - c0 + kStep - 1) / kStep;
    for (int it = 0; it < n_iter; it += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            if (it + u < n_iter) {
                const int q = q0 + (it + u) * kStep;
                load_offsets(q + 2 * kDepth * kStep, RA[(u + 2 * kDepth) % kRing], RB[(u + 2 * kDepth) % kRing]);
                load_entries(RA[(u + kDepth) % kRing], E[(u + kDepth) % (kDepth + 1)]);
                __builtin_amdgcn_sched_barrier(0);
                const bi_uint4 (&e)[kAhead] = E[u % (kDepth + 1)];
                const int a = (int)(RA[u % kRing] - base32), b = (int)(RB[u % kRing] - base32);
                const int len = b - a - EPL * gl;
                double s[PP];
#pragma unroll
                for (int p = 0; p < PP; ++p) s[p] = 0.0;
                // one entry: its PP log mu values (adjacent in LDS) times its count, into the PP sums.  Padding entries carry count 0
                // and the offset of the extra slot behind the tile, which holds 0.0 (both entry widths: tiles of 4096 bins leave
                // a two-byte entry the bit for it): 0 x 0.0, no select -- and every padding lane of a wave reads the SAME LDS
                // address, a broadcast that costs the banks one access
                auto add_entry = [&](uint32_t byte_off, uint32_t cnt, double (&sg)[PP]) {
                    const char* __restrict__ q8 = reinterpret_cast<const char*>(s_mu) + (byte_off << 1);    // bin * 16
                    const double nn = (double)cnt;
#pragma unroll
                    for (int p = 0; p < PP; p += 2) {
                        const double2 lv = *reinterpret_cast<const double2*>(q8 + (p >> 1) * kPlane);
                        sg[p] = __builtin_fma(nn, lv.x, sg[p]);
                        sg[p + 1] = __builtin_fma(nn, lv.y, sg[p + 1]);
                    }
                };
#pragma unroll
                for (int g = 0; g < kAhead; ++g) {
                    // a load behind the run's end read other runs' entries: the whole group is skipped -- its LDS reads would be
                    // as random as live ones (the kernel is bound by LDS bank conflicts: 64 lanes x 32 random bytes per entry)
                    if (EPL * L * g < len) {
                        double sg[PP];
#pragma unroll
                        for (int p = 0; p < PP; ++p) sg[p] = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t x = e[g][k];
                            if constexpr (W == 4) {
                                add_entry(x & 0x1FFF8u, x >> 17, sg);
                            } else {
                                add_entry(x & 0xFFF8u, x & 7u, sg);
                                add_entry((x >> 16) & 0xFFF8u, (x >> 16) & 7u, sg);
                            }
                        }
#pragma unroll
                        for (int p = 0; p < PP; ++p) s[p] += sg[p];
                    }
                }
                for (int j = a + gl + EPL * L * kAhead; j < b; j += L) {                  // (runs beyond EPL L AHEAD entries)
                    const uint32_t x = ent[j];
                    if constexpr (W == 4) add_entry(x & 0x1FFF8u, x >> 17, s);
                    else add_entry(x & 0xFFF8u, x & 7u, s);
                }
#pragma unroll
                for (int p = 0; p < PP; ++p) s[p] = row_group_sum<L>(s[p]);
                if (gl == 0 && q < c1) {
#pragma unroll
                    for (int p = 0; p < PP; p += 2) {
                        double2 w;
                        w.x = s[p];
                        w.y = s[p + 1];
                        *reinterpret_cast<double2*>(pout + ((int64_t)(q / kPartBlock) * n_tl * kPartBlock + q % kPartBlock) * PP + p) = w;
                    }
                }
            }
        }
    }
}

// ---- (3) finish: out[point][dataset] ------------------------------------------------------------------------------------
// mu_tot / mu_flag [passes of the group][PP]: sum mu of every column's point and its flag, added up by the dot kernel's first block
// of the pass (MuTotals); colmap [passes][PP][2] = {work item of the log mu kernel, output row} of every column (-1: no point).
// k_dataset_finish_multi: blockIdx.y = pass, blockIdx.x = block of kPartBlock = 32 datasets.  A block is 32 datasets x 8 SLICES
// of the tiles: a thread adds the partial sums of ITS dataset over its slice of the tiles for all PP columns of the pass (the PP
// values of a (tile, dataset) are adjacent: 16-byte loads; the block's input is ONE contiguous range, [tile][32 datasets][PP]),
// the slices are added in order by the first 32 threads -- a fixed order.
// done != NULL: out is pinned host memory; the block that finishes last publishes `seq` there with a system-scope release.
template <int PP>
__global__ __launch_bounds__(kThreads) void k_dataset_finish_multi(const double* __restrict__ partial, int n_tl, const int32_t* __restrict__ colmap,
                                                                   const double* __restrict__ mu_tot, const unsigned* __restrict__ mu_flag,
                                                                   const double* __restrict__ lgsum, int64_t t0, int64_t n,
                                                                   double* __restrict__ out, int64_t out_stride,
                                                                   unsigned* __restrict__ blocks_done, unsigned long long* done,
                                                                   unsigned long long seq) {
    constexpr int kDs = kPartBlock, kSl = kThreads / kDs;
    __shared__ double part[kSl][PP][kDs];
    const int pass = blockIdx.y;
    const int32_t* __restrict__ cm = colmap + (int64_t)pass * PP * 2;
    const int dsl = threadIdx.x % kDs, sl = threadIdx.x / kDs;
    const int64_t t = (int64_t)blockIdx.x * kDs + dsl;
    const int per = (n_tl + kSl - 1) / kSl;
    const int b0 = min(n_tl, sl * per), b1 = min(n_tl, b0 + per);
    double s[PP];
#pragma unroll
    for (int g = 0; g < PP; ++g) s[g] = 0.0;
    if (t < n) {
        const int64_t n_pad = (n + kPartBlock - 1) / kPartBlock * kPartBlock;
        const double* __restrict__ p = partial + (int64_t)pass * n_tl * n_pad * PP + ((int64_t)blockIdx.x * n_tl * kDs + dsl) * PP;
        const int64_t stride = (int64_t)kDs * PP;
        // four tiles in flight per thread (PP / 2 16-byte loads each), four running sums (measured: 16 datasets x 16 slices with
        // eight in flight is slower, 40 against 33 us per pass of 10^4 datasets -- with the results going to pinned host memory
        // the kernel's floor is their ~320 KB over PCIe)
        constexpr int U = 4;
        double acc[4][PP];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int g = 0; g < PP; ++g) acc[k][g] = 0.0;
        int b = b0;
        for (; b + U - 1 < b1; b += U) {
            double2 vv[U][PP / 2];
#pragma unroll
            for (int k = 0; k < U; ++k)
#pragma unroll
                for (int g = 0; g < PP; g += 2) vv[k][g / 2] = *reinterpret_cast<const double2*>(p + (int64_t)(b + k) * stride + g);
#pragma unroll
            for (int k = 0; k < U; ++k)
#pragma unroll
                for (int g = 0; g < PP; g += 2) { acc[k & 3][g] += vv[k][g / 2].x; acc[k & 3][g + 1] += vv[k][g / 2].y; }
        }
        for (; b < b1; ++b)
#pragma unroll
            for (int g = 0; g < PP; ++g) acc[0][g] += p[(int64_t)b * stride + g];
#pragma unroll
        for (int g = 0; g < PP; ++g) s[g] = (acc[0][g] + acc[1][g]) + (acc[2][g] + acc[3][g]);
    }
#pragma unroll
    for (int g = 0; g < PP; ++g) part[sl][g][dsl] = s[g];
    __syncthreads();
    if (sl == 0 && t < n) {
        const double lg = lgsum[t0 + t];
#pragma unroll
        for (int g = 0; g < PP; ++g) {
            const int orow = cm[2 * g + 1];
            if (orow < 0) continue;
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < kSl; q += 4)
                tot += (part[q][g][dsl] + part[q + 1][g][dsl]) + (part[q + 2][g][dsl] + part[q + 3][g][dsl]);
            double r = (tot - mu_tot[pass * PP + g]) - lg;
            if (mu_flag[pass * PP + g]) r = __builtin_nan("");
            out[(int64_t)orow * out_stride + t] = r;
        }
    }
    if (!done) return;
    __syncthreads();
    if (threadIdx.x < 64) {
        __threadfence_system();
        if (threadIdx.x == 0) {
            const unsigned before = __hip_atomic_fetch_add(blocks_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before == gridDim.x * gridDim.y - 1) {
                __hip_atomic_store(blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// rows of rejected points (outside the anchor box, unphysical rates): -inf, as the reference answers before it looks at any data
__global__ void k_fill_rows(double* __restrict__ out, int64_t out_stride, const int32_t* __restrict__ rows, int n_rows, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int r = 0; r < n_rows; ++r) out[(int64_t)rows[r] * out_stride + i] = v;
}

int eval_datasets_impl(bi_ctx* c, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out, double* out_dev, int32_t* status);

// P points x datasets [t0, t1) -> out [P][t1 - t0] (host) or out_dev (device, same layout); status [P] or NULL
int eval_datasets_points_impl(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out,
                              double* out_dev, int32_t* status) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (c->bb_source >= 0) return fail(c, BI_ERR_INVALID, "bi_eval_datasets_points is not available with Beeston-Barlow");
    if (c->unbinned) return fail(c, BI_ERR_INVALID, "bi_eval_datasets_points needs a binned likelihood");
    if (P < 0 || P > 65535) return fail(c, BI_ERR_INVALID, "P = %lld outside [0, 65535]", (long long)P);
    if (t0 < 0 || t1 > c->T || t0 > t1) return fail(c, BI_ERR_INVALID, "dataset range [%lld,%lld) outside [0,%lld)", (long long)t0, (long long)t1, (long long)c->T);
    if (c->d > 0 && P > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    const int64_t n = t1 - t0;
    if (P > 0 && n > 0 && !out && !out_dev) return fail(c, BI_ERR_INVALID, "out is NULL");
    if (status) std::fill(status, status + P, 0);
    if (P == 0 || n == 0) return BI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int S = c->S, d = c->d;
    // point by point through bi_eval_datasets: what the call means, and the route of everything the multi-point kernels do not
    // cover (dense counts, few datasets, lists whose counts fit neither entry format, models with more streams than a point's
    // descriptors hold)
    auto one_by_one = [&]() -> int {
        for (int64_t p = 0; p < P; ++p) {
            int32_t st = 0;
            const int r1 = eval_datasets_impl(c, z ? z + p * d : nullptr, rate_scale ? rate_scale + p * S : nullptr, t0, t1,
                                              out ? out + p * n : nullptr, out_dev ? out_dev + p * n : nullptr, &st);
            if (r1) return r1;
            if (status) status[p] = st;
        }
        return BI_OK;
    };
    const bool csr = (c->sparse && c->csr_ready) || !c->dense_counts;
    if (csr && !c->csr_ready) return fail(c, BI_ERR_STATE, "no counts resident");
    const int n_tl = (int)((c->B + kDotTileMulti - 1) / kDotTileMulti);
    const int nc = 1 << (int)c->eff_axes.size(), NS = nc * S;
    bool multi = csr && c->dot_tiled && c->toy_points_pp != 1 && P >= 2 && n >= 64 && n_tl >= 4 && c->h_nz_off.size() == (size_t)c->T + 1 &&
                 c->h_nz_off.back() >= (int64_t)8 * c->T * n_tl && (int64_t)c->T * (n_tl + 1) <= ((int64_t)1 << 28) && n <= ((int64_t)1 << 18);
    if (multi && c->tmm_epoch != c->epoch) {
        if ((rc = build_tile_major(c, kDotTileMulti, c->tmm_entries, c->tmm_off, c->tmm_ok, c->tmm_width))) return rc;
        c->tmm_epoch = c->epoch;
    }
    if (!multi || !c->tmm_ok) return one_by_one();

    // ---- per point: geometry, rates, early exits (blueice/likelihood.py:345-347, 397-415) ----
    struct Pt { int64_t cell; int idx; };
    std::vector<Pt> valid;
    std::vector<int32_t> bad_rows;
    std::vector<PointGeom> geom((size_t)P);
    std::vector<double> rates((size_t)P * S);
    for (int64_t p = 0; p < P; ++p) {
        PointGeom& g = geom[(size_t)p];
        if (!point_geometry(c, z ? z + p * d : nullptr, g)) {
            if (status) status[p] = BI_ST_OUT_OF_BOUNDS;
            bad_rows.push_back((int32_t)p);
            continue;
        }
        double* r = rates.data() + (size_t)p * S;
        interp_mus(c, g, r);
        if (rate_scale) for (int s = 0; s < S; ++s) r[s] *= rate_scale[p * S + s];
        if (!rates_physical(c, r)) {
            if (status) status[p] = BI_ST_UNPHYSICAL;
            bad_rows.push_back((int32_t)p);
            continue;
        }
        valid.push_back(Pt{g.cell_anchor, (int)p});
    }
    const int n_valid = (int)valid.size();
    std::stable_sort(valid.begin(), valid.end(), [](const Pt& a, const Pt& b) { return a.cell < b.cell; });
    const int PP = (c->toy_points_pp == 2 || (c->toy_points_pp == 0 && n_valid <= 2)) ? 2 : 4;
    const int n_pass = (n_valid + PP - 1) / PP;

    // ---- work items of the log mu kernel: the points of a GROUP of passes that share a cell ----
    // the passes are worked in groups of kPassGroup (two): the log mu rows and the per-tile partial sums of a group live in
    // scratch buffers that the next group reuses (stream order) -- 32 hypotheses x 10^4 datasets would otherwise want 0.6 GB of
    // partial sums at once, beyond what the context's recycle cache parks.  A work item covers the points of a group in one
    // cell, across its passes: eight hypotheses of one cell read the cell's anchor rows once, not once per pass.
    constexpr int kPassGroup = 2;
    const int GC = PP * kPassGroup;
    std::vector<int64_t> rowoff;
    std::vector<double> coef;
    std::vector<int32_t> meta, colmap((size_t)std::max(n_pass, 1) * PP * 2, -1);
    int n_items = 0;
    bool shared_anchor = false;           // two items read rows of the same anchor model (neighbouring cells share corners): then
    std::vector<char> anchor_used((size_t)c->A, 0);   // the default cache policy wins over the nontemporal hint
    const int n_groups = (n_pass + kPassGroup - 1) / kPassGroup;
    std::vector<int> item_begin((size_t)n_groups + 1, 0);
    for (int gr = 0; gr < n_groups; ++gr) {
        item_begin[(size_t)gr] = n_items;
        const int v0 = gr * GC, v1 = std::min(n_valid, v0 + GC);
        const int rows_of_group = std::min(kPassGroup, n_pass - gr * kPassGroup) * PP;   // rows of the table the dot kernel stages
        int v = v0;
        while (v < v1) {
            int w = v + 1;
            while (w < v1 && valid[(size_t)w].cell == valid[(size_t)v].cell) ++w;
            const PointGeom& g0 = geom[(size_t)valid[(size_t)v].idx];
            const size_t ro = rowoff.size(), co = coef.size();
            rowoff.resize(ro + NS);
            coef.resize(co + (size_t)NS * GC, 0.0);
            int k = 0;
            for (int corner = 0; corner < nc; ++corner)
                for (int s = 0; s < S; ++s, ++k) {
                    rowoff[ro + k] = ((g0.cell_anchor + corner_offset(c, corner)) * S + s) * c->Bp;
                    for (int q = v; q < w; ++q) {
                        const int p = valid[(size_t)q].idx;
                        coef[co + (size_t)k * GC + (q - v0)] = geom[(size_t)p].w[(size_t)corner] * rates[(size_t)p * S + s];
                    }
                }
            // the last rows of a group's last pass without a point repeat nothing: they stay zero (log 0 = -inf in the table, never
            // read back)
            const bool last_item_of_short_group = (w == v1) && (v1 - v0 < rows_of_group);
            meta.insert(meta.end(), {v - v0, last_item_of_short_group ? rows_of_group - (v - v0) : w - v, 0, 0});
            for (int q = v; q < w; ++q) {
                colmap[(size_t)q * 2 + 0] = n_items;          // [pass][column] flat = the point's place in the cell order
                colmap[(size_t)q * 2 + 1] = valid[(size_t)q].idx;
            }
            for (int corner = 0; corner < nc; ++corner) {
                char& used = anchor_used[(size_t)(g0.cell_anchor + corner_offset(c, corner))];
                if (used) shared_anchor = true;
                used = 1;
            }
            ++n_items;
            v = w;
        }
    }
    item_begin[(size_t)n_groups] = n_items;
    int max_group_items = 1;
    for (int gr = 0; gr < n_groups; ++gr) max_group_items = std::max(max_group_items, item_begin[(size_t)gr + 1] - item_begin[(size_t)gr]);
    const double ninf = -std::numeric_limits<double>::infinity();
    const bool host_out = !out_dev && (size_t)P * n * sizeof(double) <= ((size_t)4 << 20);
    DevBuf d_out, d_lm, d_part, d_mu;
    auto cleanup = [&]() { dev_free(d_out); dev_free(d_lm); dev_free(d_part); dev_free(d_mu); };
    const int n_tiles = n_tiles_of(c);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nmu = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (slots + max_group_items - 1) / max_group_items));   // blocks per item of a log mu launch
    PackedUpload pu;
    std::vector<std::pair<const void*, size_t>> parts = {{rowoff.data(), rowoff.size() * sizeof(int64_t)}, {coef.data(), coef.size() * sizeof(double)},
                                                        {meta.data(), meta.size() * sizeof(int32_t)}, {colmap.data(), colmap.size() * sizeof(int32_t)},
                                                        {bad_rows.data(), bad_rows.size() * sizeof(int32_t)}};
    const size_t mu_bytes = (size_t)std::max(n_items, 1) * nmu * GC * sizeof(double);
    // Two groups in flight (toy_points_overlap = 1, more than one group, no per-kernel timing): the dot kernel is bound by LDS, the
    // log mu pass and the finish by HBM -- the next group's log mu pass and the last group's finish run on the context's second
    // (low-priority) stream BESIDE the dot kernel (a block of which leaves 32 KB of LDS and half the wave slots of its CU), each on
    // its own half of the scratch buffers.  OFF by default: measured on 32 hypotheses x 10^4 toys of C2 the kernels do run side by
    // side (rocprofv3 trace), but each slows the other by what it gains -- the dot kernel 200 -> 245-310 us beside a 105 us finish
    // and a 120 us log mu pass (60 and 52 us alone): 1.37-1.41 ms per call against 1.33-1.39 in sequence
    // (tools/probe/toy_points_overlap.py).  The same bits either way.
    const bool overlap = c->toy_points_overlap && n_groups > 1 && !c->profiling;
    const int n_buf = overlap ? 2 : 1;
    const size_t lm_group = (size_t)std::min(n_pass, kPassGroup) * c->Bp * PP;                 // doubles per group
    const size_t part_group = (size_t)std::min(n_pass, kPassGroup) * n_tl * ((n + kPartBlock - 1) / kPartBlock * kPartBlock) * PP;
    if ((rc = packed_upload(c, parts, host_out ? (size_t)P * n * sizeof(double) : 0, pu)) ||
        (n_pass > 0 && (rc = dev_alloc(c, d_lm, (size_t)n_buf * lm_group * sizeof(double)))) ||
        (n_pass > 0 && (rc = dev_alloc(c, d_part, (size_t)n_buf * part_group * sizeof(double)))) ||
        (rc = dev_alloc(c, d_mu, 2 * ((mu_bytes + 63) / 64 * 64) + (size_t)std::max(n_groups, 1) * GC * 16)) ||
        (!host_out && !out_dev && (rc = dev_alloc(c, d_out, (size_t)P * n * sizeof(double))))) {
        cleanup();
        return rc;
    }
    double* res = out_dev ? out_dev : (host_out ? (double*)pu.host_out() : (double*)d_out.p);
    hipError_t e = hipSuccess;
    unsigned long long* done_word = nullptr;
    unsigned long long seq = 0;
    if (n_valid > 0) {
        LaunchArgs a{};
        a.ps = (const double*)c->ps.p;
        a.rowoff = pu.dev<int64_t>(0);
        a.coef = pu.dev<double>(1);
        a.partial = (double*)d_mu.p;
        a.pflags = (unsigned*)((char*)d_mu.p + (mu_bytes + 63) / 64 * 64);
        // per group: sum mu and the flag of every column of its table (MuTotals: the dot kernel's first block of a pass)
        double* mu_tot = (double*)((char*)d_mu.p + 2 * ((mu_bytes + 63) / 64 * 64));
        unsigned* mu_flag = (unsigned*)(mu_tot + (size_t)std::max(n_groups, 1) * GC);
        a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = n_tiles; a.chunks = (int)c->tile_chunks;
        const bool nt = c->nt_loads == 1 || (c->nt_loads == 2 && !shared_anchor);
        // the dot kernel: the datasets split over blockIdx.y so that a pass fills the chip once where it can (one resident block per
        // CU: 128 KB of LDS each)
        const size_t lds = (size_t)(kDotTileMulti + 1) * 16 * (PP / 2);            // planes of 16-byte cells (+ the zero cell)
        const int variant = (c->tmm_width == 2 ? 0 : 2) + (PP == 4 ? 0 : 1);     // {W2 PP4, W2 PP2, W4 PP4, W4 PP2}
        const int lanes = (int)c->toy_points_lanes;
        const unsigned by = (unsigned)std::max<int64_t>({1, std::min<int64_t>((n + 255) / 256, (int64_t)c->prop.multiProcessorCount / n_tl), (n + 262143) / 262144});
        // the completion word in the pinned block (behind the results, if they go there): polled instead of a stream synchronisation,
        // whether the results land in pinned host memory or stay in HBM (out_dev)
        if ((host_out || out_dev) && c->poll_result && !c->profiling && (c->toy_fast_call & 4)) {
            if ((rc = dev_alloc(c, c->toy_blocks_done, 64))) { cleanup(); return rc; }
            if (!c->toy_blocks_done_zeroed) {
                e = hipMemsetAsync(c->toy_blocks_done.p, 0, 64, c->stream);
                c->toy_blocks_done_zeroed = true;
            }
            done_word = (unsigned long long*)((char*)pu.host_out() + (host_out ? ((size_t)P * n * sizeof(double) + 63) / 64 * 64 : 0));
            seq = ++c->toy_seq;
            *(volatile unsigned long long*)done_word = 0ull;
        }
        hipStream_t sA = c->stream, sB = c->stream;
        hipEvent_t *ev_lm = c->tp_ev, *ev_dot = c->tp_ev + 2, *ev_fin = c->tp_ev + 4;
        if (overlap) {
            if (!c->stream2) {                 // the lowest priority: the dot kernel's blocks go first wherever both could
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo);
            }
            for (int k = 0; k < 7 && e == hipSuccess; ++k)
                if (!c->tp_ev[k]) e = hipEventCreateWithFlags(&c->tp_ev[k], hipEventDisableTiming);
            if (e == hipSuccess) {
                sB = c->stream2;
                e = hipEventRecord(c->tp_ev[6], sA);                       // the descriptors' upload (and whatever ran before) first
                if (e == hipSuccess) e = hipStreamWaitEvent(sB, c->tp_ev[6], 0);
            }
        }
        const void* fn = nullptr;
        auto launch_logmu = [&](int gr) {
            const int ib = item_begin[(size_t)gr], ie = item_begin[(size_t)gr + 1];
            LaunchArgs b = a;
            b.rowoff = a.rowoff + (int64_t)ib * NS;
            b.coef = a.coef + (int64_t)ib * NS * GC;
            b.partial = a.partial + (int64_t)ib * nmu * GC;
            b.pflags = a.pflags + (int64_t)ib * nmu * GC;
            const int32_t* meta_dev = pu.dev<int32_t>(2) + 4 * ib;
            double* lm = (double*)d_lm.p + (size_t)(gr % n_buf) * lm_group;
            const dim3 lgrid((unsigned)nmu, (unsigned)(ie - ib));
            EventScope ev(c);
#define BI_LM(GCv) /* GCv: rows of a group's table */                                                                          \
    do {                                                                                                                       \
        if (nt) hipLaunchKernelGGL((k_morph_logmu_multi<GCv, true>), lgrid, dim3(kThreads), 0, sB, b, meta_dev, lm);            \
        else hipLaunchKernelGGL((k_morph_logmu_multi<GCv, false>), lgrid, dim3(kThreads), 0, sB, b, meta_dev, lm);             \
    } while (0)
            if (PP == 2) BI_LM(4); else BI_LM(8);
#undef BI_LM
        };
        if (e == hipSuccess) {
            launch_logmu(0);
            if (overlap) e = hipEventRecord(ev_lm[0], sB);
        }
        for (int gr = 0; gr < n_groups && e == hipSuccess; ++gr) {
            const int g0 = gr * kPassGroup;
            const int np = std::min(kPassGroup, n_pass - g0);
            const bool last = gr + 1 == n_groups;
            const double* lm = (const double*)d_lm.p + (size_t)(gr % n_buf) * lm_group;
            double* part = (double*)d_part.p + (size_t)(gr % n_buf) * part_group;
            if (!last) {                       // the next group's log mu rows: beside this group's dot kernel when the streams differ
                if (overlap && gr >= 1) e = hipStreamWaitEvent(sB, ev_dot[(gr + 1) % 2], 0);      // (its half of the table: read by group gr - 1)
                if (e == hipSuccess && overlap) {
                    launch_logmu(gr + 1);
                    e = hipEventRecord(ev_lm[(gr + 1) % 2], sB);
                }
            }
            if (e == hipSuccess && overlap) e = hipStreamWaitEvent(sA, ev_lm[gr % 2], 0);
            if (e == hipSuccess && overlap && gr >= 2) e = hipStreamWaitEvent(sA, ev_fin[gr % 2], 0);   // (its half of the partial sums: read by group gr - 2's finish)
            if (e != hipSuccess) break;
            // L x AHEAD x (16 / W) entry slots per run; a run holds ~38 entries at configs[2] (tiles of 4096 bins)
#define BI_DM(Lv, Av, Wv, PPv)                                                                                             \
    do {                                                                                                                   \
        fn = (const void*)k_dataset_dot_multi<Lv, Av, Wv, PPv, kDotTileMulti>;                                            \
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
        if (e == hipSuccess) {                                                                                             \
            EventScope ev(c);                                                                                              \
            hipLaunchKernelGGL((k_dataset_dot_multi<Lv, Av, Wv, PPv, kDotTileMulti>), dgrid, dim3(kDotThreads), lds, sA,     \
                               (const void*)c->tmm_entries.p, (const int64_t*)c->tmm_off.p, c->T, n_tl, lm,                \
                               c->B, c->Bp, t0, n, part, mt);                                                              \
        }                                                                                                                  \
    } while (0)
            const dim3 dgrid((unsigned)n_tl, by, (unsigned)np);
            const MuTotals mt{pu.dev<int32_t>(3) + (int64_t)g0 * PP * 2, (const double*)a.partial, (const unsigned*)a.pflags, nmu, GC,
                              mu_tot + (size_t)gr * GC, mu_flag + (size_t)gr * GC};
            if (variant == 0) { if (lanes == 2) BI_DM(2, 3, 2, 4); else if (lanes == 8) BI_DM(8, 1, 2, 4); else BI_DM(4, 2, 2, 4); }
            else if (variant == 1) { if (lanes == 2) BI_DM(2, 3, 2, 2); else if (lanes == 8) BI_DM(8, 1, 2, 2); else BI_DM(4, 2, 2, 2); }
            else if (variant == 2) { if (lanes == 4) BI_DM(4, 3, 4, 4); else BI_DM(8, 2, 4, 4); }
            else { if (lanes == 4) BI_DM(4, 3, 4, 2); else BI_DM(8, 2, 4, 2); }
#undef BI_DM
            if (e == hipSuccess && overlap) {
                e = hipEventRecord(ev_dot[gr % 2], sA);
                if (e == hipSuccess) e = hipStreamWaitEvent(sB, ev_dot[gr % 2], 0);
            }
            if (e == hipSuccess) {
                EventScope ev(c);
                const dim3 fgrid((unsigned)((n + kPartBlock - 1) / kPartBlock), (unsigned)np);
                // (the completion word is published by the LAST group's finish: the second stream runs the finishes in order, and a
                //  group's finish follows its dot kernel)
#define BI_FM(PPv)                                                                                                          \
    hipLaunchKernelGGL((k_dataset_finish_multi<PPv>), fgrid, dim3(kThreads), 0, sB, (const double*)part, n_tl,              \
                       pu.dev<int32_t>(3) + (int64_t)g0 * PPv * 2, (const double*)(mu_tot + (size_t)gr * GC),                     \
                       (const unsigned*)(mu_flag + (size_t)gr * GC), (const double*)c->lgsum.p, t0, n, res, n, (unsigned*)c->toy_blocks_done.p, last ? done_word : (unsigned long long*)nullptr, seq)
                if (PP == 2) BI_FM(2); else BI_FM(4);
#undef BI_FM
                if (overlap) e = hipEventRecord(ev_fin[gr % 2], sB);
            }
            if (e == hipSuccess && !overlap && !last) launch_logmu(gr + 1);
        }
        if (e == hipSuccess && overlap) {      // what follows on the context's stream follows the finishes
            e = hipStreamWaitEvent(sA, ev_fin[(n_groups - 1) % 2], 0);
            if (e == hipSuccess) e = hipStreamWaitEvent(sA, ev_fin[n_groups % 2], 0);
        }
        c->n_toy_points_passes += n_pass;
    }
    if (e == hipSuccess && !bad_rows.empty())
        hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, res, n, pu.dev<int32_t>(4), (int)bad_rows.size(), n, ninf);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && !host_out && !out_dev) e = hipMemcpyAsync(out, d_out.p, (size_t)P * n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    bool arrived = false;
    if (e == hipSuccess && done_word && bad_rows.empty()) {
        const volatile unsigned long long* dw = done_word;
        const auto t_start = std::chrono::steady_clock::now();
        const auto t_spin = t_start + std::chrono::microseconds(30);
        const auto t_end = t_start + std::chrono::microseconds(std::min<int64_t>(50000, 2000 + (int64_t)n_valid * n));
        bool yielding = false;
        for (unsigned spin = 0; !(arrived = (*dw == seq)); ++spin) {
            if (yielding) {
                sched_yield();
                if (std::chrono::steady_clock::now() > t_end) break;
                continue;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spin & 63u) == 63u && std::chrono::steady_clock::now() > t_spin) yielding = true;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        ++c->n_toy_polled;
    }
    if (e == hipSuccess && (!arrived || (seq & 255ull) == 0)) e = hipStreamSynchronize(c->stream);
    else if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); if (c->stream2) (void)hipStreamSynchronize(c->stream2); }
    if (done_word && !arrived) c->toy_blocks_done_zeroed = false;
    if (e == hipSuccess && host_out) memcpy(out, res, (size_t)P * n * sizeof(double));
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_datasets_points: %s", hipGetErrorString(e));
    return BI_OK;
}

}  // namespace
