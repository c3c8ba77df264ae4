// bi_planning_device.h -- the planning of bi_planning.h done on the GPU, for large batches (scans).
// The host version costs ~0.2 us per point even threaded, which caps the non-empty-bin form near 4 M
// evaluations/s; here the per-point geometry, the (cell, dataset) sort (rocPRIM radix sort), the chopping into
// 16-point work items (two prefix scans) and the descriptor fill all run on the device, and the only host
// round trip is one 24-byte read of the item count.  Plain binned / unbinned likelihoods only
// (Beeston-Barlow batches are planned on the host).
#pragma once

namespace {

// Blocks of a matrix-core scan kernel variant that one CU holds at a time (registers decide it), asked from the runtime
// once per variant: the planner sizes the split of a cell's strips so that the blocks run in full rounds.
int scan_resident_blocks(bool valid, int cb, int NS, bool by_count = false) {
    if (by_count) {                      // k_scan_sorted: one variant per number of 4-stream groups (1..8), masked or not
        static std::atomic<int> sorted_cache[8][2];
        const int KG = std::max(1, std::min(8, (NS + 3) / 4)), mask = NS == 4 * KG ? 0 : 1;
        std::atomic<int>& slot = sorted_cache[KG - 1][mask];
        int v = slot.load();
        if (v > 0) return v;
        int blocks = occupancy_scan_sorted(KG, mask != 0);
        if (blocks < 1) blocks = 2;
        slot.store(blocks);
        return blocks;
    }
    static std::atomic<int> cache[2][2][4][2];
    const int kg = NS <= 4 ? 0 : (NS <= 8 ? 1 : (NS <= 16 ? 2 : 3));
    const int mask = NS == (4 << kg) ? 0 : 1;
    std::atomic<int>& slot = cache[valid ? 1 : 0][cb == 2 ? 0 : 1][kg][mask];
    int v = slot.load();
    if (v > 0) return v;
    int blocks = occupancy_scan(valid, cb, kg, mask != 0);
    if (blocks < 1) blocks = 2;
    slot.store(blocks);
    return blocks;
}

__global__ void k_iota32(int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

// The rows of the model with the bins of dataset 0 ordered BY THEIR COUNT (ties: by bin), for scans over dense data on the
// matrix cores: in that order a lane's bins carry one count, and sum n log mu turns into n log prod mu (k_scan_mfma,
// PROD = 2).  A sum over bins does not depend on their order; only the matrix-core scan kernel reads this copy.  Built on
// first use per data upload (a radix sort of the counts and one gather pass over the tensor: 4 GB at C2, ~3 ms), within
// the budget the compacted templates have (`compact_budget`); one dataset only.  -> true when the copy is resident.
bool ensure_sorted_rows(bi_ctx* c) {
    if (c->sorted_epoch == c->epoch) return c->sorted_ok;
    c->sorted_epoch = c->epoch;
    c->sorted_ok = false;
    const int64_t B = c->B, Bp = c->Bp, rows = c->A * c->S;
    if (!c->scan_pow || c->T != 1 || !c->dense_counts || c->unbinned || c->bb_source >= 0 || B < 64 || B > INT32_MAX || rows > 65535) return false;
    if ((rows + 1) * Bp * (int64_t)sizeof(double) > c->compact_budget) return false;
    DevBuf d_iota, d_perm, d_tmp;
    auto drop = [&]() { dev_free(d_iota); dev_free(d_perm); dev_free(d_tmp); };
    size_t tmp_bytes = 0;
    (void)prim_sort_pairs(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                    (size_t)B, 0u, 64u, c->stream);
    if (dev_alloc(c, c->ps_sorted, (size_t)rows * Bp * sizeof(double)) || dev_alloc(c, c->cnt_sorted, (size_t)Bp * sizeof(double)) ||
        dev_alloc(c, d_iota, (size_t)B * sizeof(int32_t)) || dev_alloc(c, d_perm, (size_t)B * sizeof(int32_t)) ||
        dev_alloc(c, d_tmp, std::max<size_t>(tmp_bytes, 256))) {
        drop();
        dev_free(c->ps_sorted); dev_free(c->cnt_sorted);
        return false;
    }
    hipLaunchKernelGGL(k_iota32, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, c->stream, (int32_t*)d_iota.p, B);
    hipError_t e = hipMemsetAsync(c->cnt_sorted.p, 0, (size_t)Bp * sizeof(double), c->stream);
    size_t tb = d_tmp.bytes;
    if (e == hipSuccess) e = prim_sort_pairs(d_tmp.p, tb, (const double*)c->counts.p, (double*)c->cnt_sorted.p, (const int32_t*)d_iota.p,
                                                       (int32_t*)d_perm.p, (size_t)B, 0u, 64u, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((Bp + kThreads - 1) / kThreads), (unsigned)rows), dim3(kThreads), 0, c->stream,
                           (const double*)c->ps.p, Bp, (const int32_t*)d_perm.p, B, Bp, (double*)c->ps_sorted.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) { dev_free(c->ps_sorted); dev_free(c->cnt_sorted); return false; }
    c->sorted_ok = true;
    return true;
}

constexpr int kDevG = 16;   // a device-planned work item has 16 slots (the last of a group is padded); Beeston-Barlow: bb_max_group

struct PlanMeta {
    int d, S, de, nc, unbinned, sparse;
    int64_t T, Bp, n_rows;
    int n_anchor[kMaxDim];
    int grid_off[kMaxDim];
    int64_t astride[kMaxDim];
    int eff_axes[kMaxDim];
    int any_allow_neg;
    const double* grid;        // concatenated anchor z values
    const double* mus;         // [A][S]
    const int64_t* corner_off; // [nc]
    const int32_t* allow_neg;  // [S]
    const double* lgsum;       // [T]
    const int64_t* c_off;      // sparse: [T] element offset of the dataset's compacted templates
    const int64_t* cnt_off;    // sparse: [T]
    const int64_t* c_np;       // sparse: [T] padded non-empty bins
    const double* Tz;          // sparse: [T][n_rows]
    const double* rowsum;      // [n_rows] sum of every template row over all bins
    int share_order;           // results in sorted order (perm = position in the share) instead of the caller's point order
    int linear_outside;        // the batch goes to k_scan_mfma: sum_b mu_b = sum_k coef_k * rowsum_k joins the per-point constant,
                               // and the kernel adds only the n log mu terms
    int G;                     // slots per work item: 16, or bb_max_group for Beeston-Barlow models
    int bb_source;             // -1, or the Beeston-Barlow source: streams [corner][s != bb], then its ps rows, then the n_model rows
    const double* nm_tot;      // Beeston-Barlow: [A] per-anchor totals of the Monte-Carlo counts (N(z) is linear in the weights)
    const double* rowmin;      // [n_rows] smallest entry of every template row (can some bin have U_b == 0 at a point?)
    uint64_t bad_key;          // the sort key of rejected points: A * T, one above every (cell, dataset) key -- the radix sort then
                               // walks only the bits of A * T (C2: 7 of them, ONE pass instead of the eight of a 64-bit key)
};

// The scalar half of likelihood.py:345-415 for ONE point: bounds, cell, the per-axis interpolation coordinates t [d], status.
// Shared by k_plan_geometry (keys) and k_plan_fill (descriptors): the same operations in the same order, so what the second
// kernel rebuilds is what the first one judged.
__device__ __forceinline__ int32_t plan_point_cell(const PlanMeta& m, const double* __restrict__ zrow, int64_t zstride, int64_t ds,
                                                   double (&t)[kMaxDim], int64_t& cell) {
    cell = 0;
    if (ds < 0 || ds >= m.T) return BI_ST_BAD_DATASET;
    for (int i = 0; i < m.d; ++i) {
        const double* g = m.grid + m.grid_off[i];
        const int n = m.n_anchor[i];
        const double zi = zrow[i * zstride];
        if (!(g[0] <= zi && zi <= g[n - 1])) return BI_ST_OUT_OF_BOUNDS;
        int k = 0;
        double ti = 0.0;
        if (n > 1) {
            if (zi == g[n - 1]) {
                k = n - 2;
            } else {
                while (k + 1 < n && g[k + 1] <= zi) ++k;   // last anchor with g[k] <= z
                k = min(k, n - 2);
            }
            ti = (zi - g[k]) / (g[k + 1] - g[k]);
        }
        t[i] = ti;
        cell += (int64_t)k * m.astride[i];
    }
    return 0;
}

// corner weight ((1 * w_0) * w_1) ... in itertools.product order (scipy's _evaluate_linear)
__device__ __forceinline__ double plan_corner_weight(const PlanMeta& m, const double (&t)[kMaxDim], int corner) {
    double wc = 1.0;
    for (int i = 0; i < m.de; ++i) {
        const double ti = t[m.eff_axes[i]];
        wc = wc * (((corner >> (m.de - 1 - i)) & 1) ? ti : (1 - ti));
    }
    return wc;
}

// per point: status and sort key.  Weights and rates are NOT written out (round 4 wrote 96 bytes per point here and gathered
// them back through the sort's permutation in k_plan_fill: 0.3 of that kernel's 0.5 ms): the fill kernel rebuilds them from
// the point's 56 bytes of z and rate_scale.
// MUS_LDS: the anchors' rate table ([A][S] doubles, n_rows of them: 4 KB at C2) is staged in LDS first -- a point reads 2^d S
// entries of it at ITS cell, a gather in which the 64 lanes of a wave touch up to 64 cache lines per load: 32 such loads per
// point were the kernel's time (53 us per 10^6 points of C2; from LDS: see DESIGN.md 5.6).
template <bool MUS_LDS>
__global__ __launch_bounds__(kThreads) void k_plan_geometry(PlanMeta m, int64_t P, const double* __restrict__ z,
                                                            const double* __restrict__ rate_scale,
                                                            const int64_t* __restrict__ dataset,
                                                            uint64_t* __restrict__ keys, int64_t* __restrict__ idx,
                                                            int32_t* __restrict__ status, unsigned long long* __restrict__ n_inf) {
    extern __shared__ double s_geo_mus[];
    if (MUS_LDS) {
        for (int64_t i = threadIdx.x; i < m.n_rows; i += kThreads) s_geo_mus[i] = m.mus[i];
        __syncthreads();
    }
    const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (p >= P) return;
    idx[p] = p;
    const int64_t ds = dataset ? dataset[p] : 0;
    double t[kMaxDim];
    int64_t cell = 0;
    int32_t st = plan_point_cell(m, z + p * m.d, 1, ds, t, cell);
    if (!st) {
        bool any_fin = false, phys = true, inf_rate = false;
        double tot = 0.0;
        auto judge = [&](int s, double v) {
            if (rate_scale) v *= rate_scale[p * m.S + s];
            any_fin |= (v < __builtin_inf());
            inf_rate |= (v == __builtin_inf() || v == -__builtin_inf());
            tot += v;
            if (!m.any_allow_neg) { if (!(v >= 0 && v < __builtin_inf())) phys = false; }
            else if (!(0 <= v) && !m.allow_neg[s]) phys = false;
        };
        constexpr int kFastS = 8;
        if (m.S <= kFastS) {
            // every corner's weight once (not once per source), the sources' sums side by side in registers: the same products
            // added in the same order, a quarter of the multiplications (this kernel is bound by its fp64 instructions)
            double teff[kMaxDim], v[kFastS];
#pragma unroll
            for (int i = 0; i < kMaxDim; ++i) teff[i] = i < m.de ? t[m.eff_axes[i]] : 0.0;
#pragma unroll
            for (int s = 0; s < kFastS; ++s) v[s] = 0.0;
            for (int corner = 0; corner < m.nc; ++corner) {
                double wc = 1.0;
#pragma unroll
                for (int i = 0; i < kMaxDim; ++i)
                    if (i < m.de) wc = wc * (((corner >> (m.de - 1 - i)) & 1) ? teff[i] : (1 - teff[i]));
                const int64_t mrow = (cell + m.corner_off[corner]) * m.S;
#pragma unroll
                for (int s = 0; s < kFastS; ++s)
                    if (s < m.S) { const double term = (MUS_LDS ? s_geo_mus[mrow + s] : m.mus[mrow + s]) * wc; v[s] = v[s] + term; }
            }
#pragma unroll
            for (int s = 0; s < kFastS; ++s)
                if (s < m.S) judge(s, v[s]);
        } else {
            for (int s = 0; s < m.S; ++s) {
                double v = 0.0;
                for (int corner = 0; corner < m.nc; ++corner) {
                    const int64_t mi = (cell + m.corner_off[corner]) * m.S + s;
                    const double term = (MUS_LDS ? s_geo_mus[mi] : m.mus[mi]) * plan_corner_weight(m, t, corner);
                    v = v + term;
                }
                judge(s, v);
            }
        }
        if (m.any_allow_neg && (!any_fin || tot < 0)) phys = false;
        if (!phys) st = BI_ST_UNPHYSICAL;
        // an infinite rate that passes (a source may go negative: likelihood.py:403-415) is answered on the host, the way
        // the reference evaluates it (inf_rate_value): such a batch is not for this planner
        if (phys && m.any_allow_neg && n_inf && inf_rate) atomicAdd(n_inf, 1ull);
    }
    status[p] = st;
    keys[p] = st ? m.bad_key : (uint64_t)(cell * m.T + ds);
}

// number of valid (sorted-to-the-front) points
__global__ void k_plan_count_valid(const uint64_t* __restrict__ keys, int64_t P, uint64_t bad_key, int64_t* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    int64_t lo = 0, hi = P;   // first index with the key of rejected points
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] != bad_key) lo = mid + 1; else hi = mid;
    }
    out[0] = lo;
}

__global__ __launch_bounds__(kThreads) void k_plan_heads(const uint64_t* __restrict__ keys, int64_t n, int64_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? i : 0;
}

__global__ __launch_bounds__(kThreads) void k_plan_item_heads(const int64_t* __restrict__ gstart, int64_t n, int G, int64_t* __restrict__ ihead) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) ihead[i] = ((i - gstart[i]) % G == 0) ? 1 : 0;
}

// ---- group structure from per-key tables (few distinct keys: A * T + 1 <= 65 536) --------------------------------------------
// The sorted keys fall into at most A * T runs.  Where a run starts is found by comparing neighbours (k_plan_key_bounds); what
// the fill kernel needs per sorted position -- its group's start, its work item, its slot -- then follows from three small
// tables over the KEYS, built by one block (k_plan_key_tables): this rank's window [lo, hi) of the valid points, each key's run
// clipped to it, the items and groups before it.  No device-wide scan over the points, one host round trip for the counts.
__global__ __launch_bounds__(kThreads) void k_plan_key_bounds(const uint64_t* __restrict__ keys, int64_t P, int64_t* __restrict__ kstart) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < P && (i == 0 || keys[i] != keys[i - 1])) kstart[keys[i]] = i;
}

// scal: [0] valid points in all, [1] work items of this share, [3] items of its largest group, [6] its groups, [7] / [8] the share's
// window of the sorted list.  tab_start [K]: start of the key's run inside the window (relative to lo), tab_item [K]: items before it;
// grp_first / grp_items [<= K]: the group tables.  kstart [K + 1]: first sorted position of every key (-1: absent), index K = rejected.
__global__ __launch_bounds__(kThreads) void k_plan_key_tables(const int64_t* __restrict__ kstart, int64_t K, int64_t P, int share_rank,
                                                              int share_world, int G, int64_t* __restrict__ tab_start,
                                                              int64_t* __restrict__ tab_item, int64_t* __restrict__ grp_first,
                                                              int32_t* __restrict__ grp_items, int64_t* __restrict__ scal) {
    __shared__ int64_t s_first[kThreads], s_items[kThreads], s_groups[kThreads];
    __shared__ int64_t s_lo, s_hi;
    const int t = threadIdx.x;
    const int64_t n_valid = kstart[K] >= 0 ? kstart[K] : P;
    if (t == 0) {
        const int64_t base = n_valid / share_world, extra = n_valid % share_world;
        s_lo = share_rank * base + (share_rank < extra ? share_rank : extra);
        s_hi = s_lo + base + (share_rank < extra ? 1 : 0);
    }
    const int64_t per = (K + kThreads - 1) / kThreads;
    const int64_t k0 = min((int64_t)t * per, K), k1 = min(k0 + per, K);
    // the first run that starts in this thread's range of keys (starts ascend with the key)
    int64_t first = -1;
    for (int64_t k = k0; k < k1 && first < 0; ++k) first = kstart[k];
    s_first[t] = first;
    __syncthreads();
    const int64_t lo = s_lo, hi = s_hi;
    // where the run behind this thread's range starts: the next thread with a run, or the end of the valid points
    int64_t nxt = n_valid;
    for (int u = t + 1; u < kThreads; ++u)
        if (s_first[u] >= 0) { nxt = s_first[u]; break; }
    // backwards over the range: every run ends where the next one starts; its part inside [lo, hi); items and groups of the range
    int64_t items = 0, groups = 0, largest = 0;
    for (int64_t k = k1 - 1; k >= k0; --k) {
        const int64_t s = kstart[k];
        int64_t s_in = 0, cnt = 0;
        if (s >= 0) {
            const int64_t a = min(max(s, lo), hi), b = min(max(nxt, lo), hi);
            s_in = a - lo;
            cnt = b - a;
            nxt = s;
        }
        tab_start[k] = s_in;
        const int64_t it = (cnt + G - 1) / G;
        tab_item[k] = it;                       // (for now: the run's own items; made a prefix below)
        items += it;
        groups += cnt > 0 ? 1 : 0;
        largest = max(largest, it);
    }
    s_items[t] = items;
    s_groups[t] = groups;
    __syncthreads();
    int64_t item_base = 0, grp_base = 0;
    for (int u = 0; u < t; ++u) { item_base += s_items[u]; grp_base += s_groups[u]; }
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t it = tab_item[k];
        tab_item[k] = item_base;
        if (it > 0) {
            grp_first[grp_base] = item_base;
            grp_items[grp_base] = (int32_t)it;
            ++grp_base;
        }
        item_base += it;
    }
    if (largest > 0) atomicMax((unsigned long long*)(scal + 3), (unsigned long long)largest);
    if (t == kThreads - 1) {
        scal[0] = n_valid;
        scal[1] = item_base;
        scal[6] = grp_base;
        scal[7] = lo;
        scal[8] = hi;
    }
}

// One thread per sorted position.  The block first fetches its points' z and rate_scale rows TOGETHER into LDS -- lanes walk the
// (point, component) pairs, so the d (or S) doubles of a point are fetched by adjacent lanes and a load instruction touches a
// third (a quarter) of the lines it would with one point per lane --, every thread then rebuilds its point's corner weights and
// rates there ([index][thread]: dynamic indices without scratch, no bank conflicts), and writes the descriptors: the
// coefficients of an item are [stream][slot], so the 16 threads of an item write whole 128-byte lines.  Padding slots of a
// group's last item are written here too (perm = -1, a copy of the last point's coefficients): no memset precedes the kernel.
// LDS doubles per thread: d + S (inputs) + nc (weights) + S (rates).
__global__ void k_plan_fill(PlanMeta m, int64_t n, const uint64_t* __restrict__ keys,
                            const int64_t* __restrict__ idx, const int64_t* __restrict__ gstart /* per position, or NULL: tables */,
                            const int64_t* __restrict__ item_incl, const int64_t* __restrict__ tab_start, const int64_t* __restrict__ tab_item,
                            const double* __restrict__ z,
                            const double* __restrict__ rate_scale, int64_t* __restrict__ rowoff,
                            double* __restrict__ coef, int64_t* __restrict__ cnt_off,
                            int32_t* __restrict__ tiles, int64_t* __restrict__ perm,
                            double* __restrict__ slot_lg, unsigned long long* __restrict__ tile_sum,
                            int64_t* __restrict__ rowoff_full /* split scans, else NULL */,
                            double* __restrict__ aux /* Beeston-Barlow: [items][G][2], else NULL */,
                            unsigned long long* __restrict__ n_zero_u /* Beeston-Barlow, else NULL */,
                            int mus_lds /* the rate table [n_rows] staged in LDS behind the per-thread arrays */,
                            int tz_lds /* ... and behind it the per-row constants (row sums, or the T = 1 dataset's Tz) */) {
    // (the items' tile counts are summed per block first: one atomic per work item on ONE address -- 62 500 of them for a
    //  10^6-point scan -- was half of this kernel's 0.5 ms)
    __shared__ unsigned long long s_tiles;
    extern __shared__ double s_pf[];
    const int BD = blockDim.x, tx = threadIdx.x;
    double* __restrict__ s_z = s_pf;                              // [d][BD]
    double* __restrict__ s_rs = s_z + (size_t)m.d * BD;            // [S][BD]   rate scales, then the rates
    double* __restrict__ s_w = s_rs + (size_t)m.S * BD;            // [nc][BD]
    int64_t* __restrict__ s_idx = reinterpret_cast<int64_t*>(s_w + (size_t)m.nc * BD);   // [BD]
    // the small per-row tables every point gathers from at ITS cell (2^d S entries each: up to 64 cache lines per wave-level load
    // from global memory) -- staged once per block where they fit (k_plan_geometry<true> does the same)
    double* __restrict__ s_mus = reinterpret_cast<double*>(s_idx + BD);                  // [n_rows] if mus_lds
    double* __restrict__ s_tz = s_mus + (mus_lds ? m.n_rows : 0);                        // [n_rows] if tz_lds
    const double* __restrict__ tz_src = m.linear_outside ? m.rowsum : m.Tz;             // (tz_lds with Tz only when T == 1)
    if (mus_lds) for (int64_t e = tx; e < m.n_rows; e += BD) s_mus[e] = m.mus[e];
    if (tz_lds) for (int64_t e = tx; e < m.n_rows; e += BD) s_tz[e] = tz_src[e];
    if (tx == 0) s_tiles = 0ull;
    const int64_t i0 = (int64_t)blockIdx.x * BD;
    const int64_t i = i0 + tx;
    const int live = (int)min<int64_t>(BD, n - i0);              // sorted positions of this block
    if (tx < live) s_idx[tx] = idx[i];
    __syncthreads();
    for (int e = tx; e < live * m.d; e += BD) {
        const int q = e / m.d, comp = e - q * m.d;
        s_z[comp * BD + q] = z[s_idx[q] * m.d + comp];
    }
    for (int e = tx; e < live * m.S; e += BD) {
        const int q = e / m.S, comp = e - q * m.S;
        s_rs[comp * BD + q] = rate_scale ? rate_scale[s_idx[q] * m.S + comp] : 1.0;
    }
    __syncthreads();
    [&]() {
    if (i >= n) return;
    const int64_t p = s_idx[tx];
    const int G = m.G;
    int g;
    int64_t item;
    bool last_of_group;
    if (gstart) {                                          // per-position arrays (many distinct keys: the scans' route)
        g = (int)((i - gstart[i]) % G);
        item = item_incl[i] - 1;
        last_of_group = i + 1 == n || gstart[i + 1] == i + 1;
    } else {                                               // tables over the keys
        const int64_t in_group = i - tab_start[keys[i]];
        g = (int)(in_group % G);
        item = tab_item[keys[i]] + in_group / G;
        last_of_group = i + 1 == n || keys[i + 1] != keys[i];
    }
    const int64_t ds = (int64_t)(keys[i] % (uint64_t)m.T);
    const bool bb = m.bb_source >= 0;
    const int n0 = bb ? m.nc * (m.S - 1) : m.nc * m.S;
    const int NS = bb ? n0 + 2 * m.nc : n0;
    const int64_t row_stride = m.sparse ? m.c_np[ds] : m.Bp;
    const int64_t row_base = m.sparse ? m.c_off[ds] : 0;
    // the point's geometry again (plan_point_cell: what k_plan_geometry judged valid), weights and rates into LDS
    double t[kMaxDim];
    int64_t cell = 0;
    (void)plan_point_cell(m, s_z + tx, BD, ds, t, cell);
    for (int corner = 0; corner < m.nc; ++corner) s_w[corner * BD + tx] = plan_corner_weight(m, t, corner);
    double rsum = 0.0;
    for (int s = 0; s < m.S; ++s) {
        double v = 0.0;
        for (int corner = 0; corner < m.nc; ++corner) {
            const int64_t mi = (cell + m.corner_off[corner]) * m.S + s;
            const double term = (mus_lds ? s_mus[mi] : m.mus[mi]) * s_w[corner * BD + tx];
            v = v + term;
        }
        if (rate_scale) v *= s_rs[s * BD + tx];
        s_rs[s * BD + tx] = v;
        rsum += v;
    }
#define W_(c) s_w[(c) * BD + tx]
#define R_(s) s_rs[(s) * BD + tx]
    double zsum = 0.0;
    // the unused slots of a group's last work item repeat its last point (their results are dropped: perm = -1): with
    // coefficients of zero their expectations would be zero, and the product forms of the scan kernels would have to leave
    // their fast path for the whole item
    const int g_end = last_of_group ? G : g + 1;
    int k = 0;
    for (int corner = 0; corner < m.nc; ++corner) {
        const int64_t a = cell + m.corner_off[corner];
        const double wc = W_(corner);
#pragma unroll 4
        for (int s = 0; s < m.S; ++s) {                     // (unrolled: the Tz loads of a corner go out together)
            if (bb && s == m.bb_source) continue;
            const double cf = wc * R_(s);
            const double tz = tz_lds ? s_tz[a * m.S + s]
                                     : (m.linear_outside ? m.rowsum[a * m.S + s] : (m.sparse ? m.Tz[ds * m.n_rows + a * m.S + s] : 0.0));
            for (int gg = g; gg < g_end; ++gg) coef[(item * NS + k) * G + gg] = cf;
            if (m.sparse || m.linear_outside) zsum += cf * tz;
            if (g == 0) {
                rowoff[item * NS + k] = row_base + (a * m.S + s) * row_stride;
                if (rowoff_full) rowoff_full[item * NS + k] = (a * m.S + s) * m.Bp;
            }
            ++k;
        }
    }
    if (bb) {
        // the Beeston-Barlow source's own template rows and its Monte-Carlo counts: coefficient = the corner weight
        // (blueice/likelihood.py:643-646); N(z) = sum_c w_c N_c and p_cal = r_i / N travel per point (aux)
        double Ntot = 0.0;
        for (int corner = 0; corner < m.nc; ++corner) {
            const int64_t a = cell + m.corner_off[corner];
            for (int gg = g; gg < g_end; ++gg) {
                coef[(item * NS + n0 + corner) * G + gg] = W_(corner);
                coef[(item * NS + n0 + m.nc + corner) * G + gg] = W_(corner);
            }
            if (g == 0) {
                rowoff[item * NS + n0 + corner] = (a * m.S + m.bb_source) * m.Bp;
                rowoff[item * NS + n0 + m.nc + corner] = a * m.Bp;
            }
            const double term = m.nm_tot[a] * W_(corner);
            Ntot = Ntot + term;
        }
        for (int gg = g; gg < g_end; ++gg) {
            aux[(item * G + gg) * 2 + 0] = R_(m.bb_source) / Ntot;
            aux[(item * G + gg) * 2 + 1] = Ntot;
        }
        // can some bin have U_b == 0 at this point (bb_zero_u_possible, bi_single.h)?  Then the reference's first-root
        // assertion hangs on the last bits of N, which only the host planner's extra pass reproduces (bb_exact_totals)
        bool zero_u = true;
        for (int s = 0; s < m.S && zero_u; ++s) {
            if (s == m.bb_source || !(R_(s) > 0.0)) continue;
            bool positive = true;
            for (int corner = 0; corner < m.nc && positive; ++corner) {
                if (!(W_(corner) > 0.0)) continue;
                positive = m.rowmin[(cell + m.corner_off[corner]) * m.S + s] > 0.0;
            }
            if (positive) zero_u = false;
        }
        if (zero_u && n_zero_u) atomicAdd(n_zero_u, 1ull);
    }
#undef W_
#undef R_
    if (g == 0) {
        cnt_off[item] = m.sparse ? m.cnt_off[ds] : ds * m.Bp;
        tiles[item] = (int32_t)(row_stride / kTile);
        atomicAdd(&s_tiles, (unsigned long long)(row_stride / kTile));
    }
    slot_lg[item * G + g] = m.unbinned ? rsum : m.lgsum[ds] + zsum;
    perm[item * G + g] = m.share_order ? i : p;
    for (int gg = g + 1; gg < g_end; ++gg) {            // the padding slots behind a group's last point
        slot_lg[item * G + gg] = 0.0;
        perm[item * G + gg] = -1;
    }
    }();
    __syncthreads();
    if (tx == 0 && s_tiles) atomicAdd(tile_sum, s_tiles);
}

// group bookkeeping for the scan kernel: flag[i] = 1 at the first sorted position of every (cell, dataset) group
__global__ __launch_bounds__(kThreads) void k_plan_group_flags(const int64_t* __restrict__ gstart, int64_t n, int64_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) flag[i] = (gstart[i] == i) ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void k_plan_group_first(const int64_t* __restrict__ gstart, const int64_t* __restrict__ gid_incl,
                                                               const int64_t* __restrict__ item_incl, int64_t n,
                                                               int64_t* __restrict__ grp_first) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n && gstart[i] == i) grp_first[gid_incl[i] - 1] = item_incl[i] - 1;
}

__global__ __launch_bounds__(kThreads) void k_plan_group_items(const int64_t* __restrict__ grp_first, int64_t n_groups, int64_t n_items,
                                                               int32_t* __restrict__ grp_items, unsigned long long* __restrict__ max_items /* or NULL */) {
    const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (g >= n_groups) return;
    const int64_t n = (g + 1 < n_groups ? grp_first[g + 1] : n_items) - grp_first[g];
    grp_items[g] = (int32_t)n;
    if (max_items) atomicMax(max_items, (unsigned long long)n);
}

__global__ __launch_bounds__(kThreads) void k_fill_bad_by_status(const int32_t* __restrict__ status, int64_t P, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (p < P && (status[p] & (BI_ST_OUT_OF_BOUNDS | BI_ST_UNPHYSICAL | BI_ST_BAD_DATASET))) out[p] = -__builtin_inf();
}

// model / data tables the planning kernels read, mirrored on the device once per model+data epoch
// ---- what the planner reports back to the host ----------------------------------------------------------------------------------
// The planner needs a few counters (and, for a rank's share of a dealt scan, the group tables) on the host twice per call.  As
// hipMemcpyAsync into pageable memory each piece was a staged copy of ~20 us and the wait a stream synchronisation of another
// ~20; here ONE small kernel writes the pieces through the host mapping of a pinned block and publishes a sequence number behind
// them (system-scope release), which the host polls -- the same hand-over as the single-evaluation path's result word.
struct ReportPiece { const uint32_t* src; uint32_t words; uint32_t dst_word; };
constexpr size_t kPlanHostBytes = 64 * 1024;          // 128 B of counters, 4096 groups x (8 + 4) B, the word at the end
constexpr size_t kPlanFlagWord = (kPlanHostBytes - 64) / 4;

__global__ __launch_bounds__(kThreads) void k_plan_report(ReportPiece a, ReportPiece b, ReportPiece c3, uint32_t* __restrict__ dst,
                                                          unsigned long long* flag, unsigned long long seq) {
    for (uint32_t i = threadIdx.x; i < a.words; i += kThreads) dst[a.dst_word + i] = a.src[i];
    for (uint32_t i = threadIdx.x; i < b.words; i += kThreads) dst[b.dst_word + i] = b.src[i];
    for (uint32_t i = threadIdx.x; i < c3.words; i += kThreads) dst[c3.dst_word + i] = c3.src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// launch the report and wait for it; the pieces are then at ((uint32_t*)c->plan_host) + dst_word
hipError_t plan_report(bi_ctx* c, ReportPiece a, ReportPiece b = ReportPiece{nullptr, 0, 0}, ReportPiece c3 = ReportPiece{nullptr, 0, 0}) {
    hipError_t e = hipSuccess;
    if (!c->plan_host) {
        e = hipHostMalloc(&c->plan_host, kPlanHostBytes, hipHostMallocDefault);
        if (e != hipSuccess) { c->plan_host = nullptr; return e; }
        memset(c->plan_host, 0, kPlanHostBytes);
    }
    if ((size_t)a.dst_word + a.words > kPlanFlagWord || (size_t)b.dst_word + b.words > kPlanFlagWord || (size_t)c3.dst_word + c3.words > kPlanFlagWord)
        return hipErrorInvalidValue;
    unsigned long long* flag = (unsigned long long*)((uint32_t*)c->plan_host + kPlanFlagWord);
    const unsigned long long seq = ++c->plan_seq;
    hipLaunchKernelGGL(k_plan_report, dim3(1), dim3(kThreads), 0, c->stream, a, b, c3, (uint32_t*)c->plan_host, flag, seq);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (c->poll_result && !c->profiling) {
        const volatile unsigned long long* f = flag;
        const auto t_start = std::chrono::steady_clock::now();
        const auto t_spin = t_start + std::chrono::microseconds(100), t_end = t_start + std::chrono::milliseconds(20);
        bool yielding = false;                 // (the planning kernels of a large batch take a while: the core is given away after 100 us)
        for (unsigned spin = 0; *f != seq; ++spin) {
            if (yielding) {
                sched_yield();
                if (std::chrono::steady_clock::now() > t_end) break;                           // (a fault upstream: the synchronisation below reports it)
                continue;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spin & 63u) == 63u && std::chrono::steady_clock::now() > t_spin) yielding = true;
        }
        if (*f == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return hipSuccess;
        }
    }
    return hipStreamSynchronize(c->stream);
}

int ensure_plan_tables(bi_ctx* c) {
    if (c->plan_tables_epoch == c->epoch) return BI_OK;
    int rc;
    std::vector<double> grid;
    for (auto& g : c->grid) grid.insert(grid.end(), g.begin(), g.end());
    const int nc = 1 << (int)c->eff_axes.size();
    std::vector<int64_t> coff((size_t)nc);
    for (int k = 0; k < nc; ++k) coff[(size_t)k] = corner_offset(c, k);
    if ((rc = dev_upload(c, c->pt_grid, grid)) || (rc = dev_upload(c, c->pt_mus, c->h_mus)) ||
        (rc = dev_upload(c, c->pt_coff, coff)) || (rc = dev_upload(c, c->pt_allow, c->allow_neg)) ||
        (rc = dev_upload(c, c->pt_rowsum, c->h_rowsum)) || (rc = dev_upload(c, c->pt_rowmin, c->h_rowmin)) ||
        (c->bb_source >= 0 && (rc = dev_upload(c, c->pt_nm_tot, c->h_nm_tot))))
        return rc;
    if (c->compact_ready && ((rc = dev_upload(c, c->pt_c_off, c->h_c_off)) || (rc = dev_upload(c, c->pt_cnt_off, c->h_cnt_off)) ||
                             (rc = dev_upload(c, c->pt_c_np, c->h_c_np)) || (rc = dev_upload(c, c->pt_Tz, c->h_Tz))))
        return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->plan_tables_epoch = c->epoch;
    return BI_OK;
}

// share_world > 1: this context evaluates only ITS contiguous share of the (cell, dataset)-sorted list of valid points --
// the dealing of a scan over several GPUs done where the sort already happens (every rank plans the same P points and
// gets the same order; rank r takes sorted positions [lo_r, hi_r)).  Its results then come out in sorted order
// (out[0 .. hi - lo)), and the plan keeps the sorted -> original index map for unsort_share.
// the model / data description the planning kernels take by value (tables: ensure_plan_tables)
PlanMeta plan_meta_of(const bi_ctx* c, bool sparse) {
    const int S = c->S, d = c->d, de = (int)c->eff_axes.size();
    PlanMeta m{};
    m.d = d; m.S = S; m.de = de; m.nc = 1 << de; m.unbinned = c->unbinned ? 1 : 0; m.sparse = sparse ? 1 : 0;
    m.T = c->T; m.Bp = c->Bp; m.n_rows = c->A * S;
    int off = 0;
    for (int i = 0; i < d; ++i) { m.n_anchor[i] = c->n_anchor[(size_t)i]; m.grid_off[i] = off; off += c->n_anchor[(size_t)i]; m.astride[i] = c->astride[(size_t)i]; }
    for (int i = 0; i < de; ++i) m.eff_axes[i] = c->eff_axes[(size_t)i];
    for (int s = 0; s < S; ++s) m.any_allow_neg |= (c->allow_neg[(size_t)s] != 0);
    m.grid = (const double*)c->pt_grid.p; m.mus = (const double*)c->pt_mus.p; m.corner_off = (const int64_t*)c->pt_coff.p;
    m.allow_neg = (const int32_t*)c->pt_allow.p; m.lgsum = (const double*)c->lgsum.p;
    m.c_off = (const int64_t*)c->pt_c_off.p; m.cnt_off = (const int64_t*)c->pt_cnt_off.p; m.c_np = (const int64_t*)c->pt_c_np.p;
    m.Tz = (const double*)c->pt_Tz.p;
    m.rowsum = (const double*)c->pt_rowsum.p;
    m.rowmin = (const double*)c->pt_rowmin.p;
    m.nm_tot = (const double*)c->pt_nm_tot.p;
    m.bb_source = c->bb_source;
    m.G = c->bb_source >= 0 ? (int)std::min<int64_t>(c->max_group, c->bb_max_group) : kDevG;
    m.bad_key = (uint64_t)c->A * (uint64_t)std::max<int64_t>(c->T, 1);
    return m;
}

// Gradient mode (bi_eval_grad) for large batches: one work item per point, its descriptors written where they are read.
// The coefficient MATRIX of a point -- [2^d * S stream rows][G columns: value, d/dz_j, d/drate_s] -- is 2 KB at C2; built on
// the host that is 268 MB to fill, stage and copy for 131 072 points (105 of the call's 145 ms), built here it is a few
// microseconds of stores.  One thread per point: the scalar half of likelihood.py:345-415 as in k_plan_geometry, then
// d w_c / d z_i = (+-1/delta_i) prod_{j != i} w^(j), d mus_s / d z_i, and the columns.  Rejected points keep their status,
// get a work item of zero tiles, and are answered on the host.
constexpr int kGradMaxCorners = 1 << 6;          // d_eff <= 6 on this path (the host path has no such limit)
constexpr int kGradFillThreads = 64;
// per-thread work arrays live in LDS, [index][lane] (conflict-free): w[64], dmus[6][16], mus[16], r[16], lg[16], t[8], 1/delta[8].
// In registers they would be indexed dynamically, i.e. sit in scratch -- 1.9 KB per lane, which the runtime provisions for
// every wave slot of the chip (~1 GB) on every launch: 3 ms per call.
constexpr int kGradFillDoubles = kGradMaxCorners + 6 * 16 + 16 + 16 + 16 + kMaxDim + kMaxDim;
__global__ __launch_bounds__(kGradFillThreads) void k_grad_fill(PlanMeta m, int64_t P, const double* __restrict__ z,
                                                                const double* __restrict__ rate_scale, const int64_t* __restrict__ dataset, int G,
                                                                int64_t* __restrict__ rowoff, double* __restrict__ coef, int64_t* __restrict__ cnt_off,
                                                                int32_t* __restrict__ tiles, int64_t* __restrict__ perm, double* __restrict__ slot_lg,
                                                                int32_t* __restrict__ status) {
    extern __shared__ double s_grad[];
    const int lane = threadIdx.x;
#define BI_AT(base, i) s_grad[((base) + (i)) * kGradFillThreads + lane]
#define W_(c) BI_AT(0, c)
#define DMUS_(i, s) BI_AT(kGradMaxCorners, (i) * 16 + (s))
#define MUS_(s) BI_AT(kGradMaxCorners + 96, s)
#define R_(s) BI_AT(kGradMaxCorners + 112, s)
#define LG_(q) BI_AT(kGradMaxCorners + 128, q)
#define T_(i) BI_AT(kGradMaxCorners + 144, i)
#define ID_(i) BI_AT(kGradMaxCorners + 144 + kMaxDim, i)
    const int64_t p = (int64_t)blockIdx.x * kGradFillThreads + threadIdx.x;
    if (p >= P) return;
    const int NS = m.nc * m.S, W = 1 + m.d + m.S;
    int32_t st = 0;
    const int64_t ds = dataset ? dataset[p] : 0;
    if (ds < 0 || ds >= m.T) st = BI_ST_BAD_DATASET;
    int64_t cell = 0;
    if (!st) {
        for (int i = 0; i < m.d; ++i) {
            const double* g = m.grid + m.grid_off[i];
            const int n = m.n_anchor[i];
            const double zi = z[p * m.d + i];
            if (!(g[0] <= zi && zi <= g[n - 1])) { st = BI_ST_OUT_OF_BOUNDS; break; }
            int k = 0;
            double ti = 0.0, id = 0.0;
            if (n > 1) {
                if (zi == g[n - 1]) {
                    k = n - 2;
                } else {
                    while (k + 1 < n && g[k + 1] <= zi) ++k;
                    k = min(k, n - 2);
                }
                ti = (zi - g[k]) / (g[k + 1] - g[k]);
                id = 1.0 / (g[k + 1] - g[k]);
            }
            T_(i) = ti;
            ID_(i) = id;
            cell += (int64_t)k * m.astride[i];
        }
    }
    if (!st) {
        for (int corner = 0; corner < m.nc; ++corner) {
            double wc = 1.0;
            for (int i = 0; i < m.de; ++i) {
                const double ti = T_(m.eff_axes[i]);
                wc = wc * (((corner >> (m.de - 1 - i)) & 1) ? ti : (1 - ti));
            }
            W_(corner) = wc;
        }
        bool any_fin = false, phys = true;
        double tot = 0.0;
        for (int s = 0; s < m.S; ++s) {
            double v = 0.0;
            for (int corner = 0; corner < m.nc; ++corner) {
                const double term = m.mus[(cell + m.corner_off[corner]) * m.S + s] * W_(corner);
                v = v + term;
            }
            MUS_(s) = v;
            if (rate_scale) v *= rate_scale[p * m.S + s];
            R_(s) = v;
            any_fin |= (v < __builtin_inf());
            tot += v;
            if (!m.any_allow_neg) { if (!(v >= 0 && v < __builtin_inf())) phys = false; }
            else if (!(0 <= v) && !m.allow_neg[s]) phys = false;
        }
        if (m.any_allow_neg && (!any_fin || tot < 0)) phys = false;
        if (!phys) st = BI_ST_UNPHYSICAL;
    }
    status[p] = st;
    int64_t* ro = rowoff + p * NS;
    double* co = coef + p * NS * G;
    for (int q = 0; q < G; ++q) { perm[p * G + q] = (!st && q < W) ? p * W + q : -1; slot_lg[p * G + q] = 0.0; }
    if (st) {                                    // a work item that does nothing
        for (int k = 0; k < NS; ++k) ro[k] = 0;
        cnt_off[p] = 0;
        tiles[p] = 0;
        return;
    }
    const int64_t row_stride = m.sparse ? m.c_np[ds] : m.Bp;
    const int64_t row_base = m.sparse ? m.c_off[ds] : 0;
    for (int q = 0; q < W; ++q) LG_(q) = 0.0;
    // d w_c / d z_i = (+-1/delta_i) prod_{j != i} w^(j)
    auto dw = [&](int corner, int i) {
        double v = (((corner >> (m.de - 1 - i)) & 1) ? 1.0 : -1.0) * ID_(m.eff_axes[i]);
        for (int j = 0; j < m.de; ++j) {
            if (j == i) continue;
            const double tj = T_(m.eff_axes[j]);
            v *= ((corner >> (m.de - 1 - j)) & 1) ? tj : (1 - tj);
        }
        return v;
    };
    // d mus_s / d z_i = sum over corners of d w_c / d z_i times the anchor's expectation
    for (int i = 0; i < m.de; ++i) {
        for (int s = 0; s < m.S; ++s) DMUS_(i, s) = 0.0;
        for (int c2 = 0; c2 < m.nc; ++c2) {
            const double v = dw(c2, i);
            for (int s = 0; s < m.S; ++s) DMUS_(i, s) += v * m.mus[(cell + m.corner_off[c2]) * m.S + s];
        }
    }
    int k = 0;
    for (int corner = 0; corner < m.nc; ++corner) {
        const int64_t a = cell + m.corner_off[corner];
        const double wc = W_(corner);
        for (int s = 0; s < m.S; ++s, ++k) {
            const int64_t row = a * m.S + s;
            ro[k] = row_base + row * row_stride;
            double* col = co + (int64_t)k * G;
            for (int q = 0; q < G; ++q) col[q] = 0.0;
            const double rs = rate_scale ? rate_scale[p * m.S + s] : 1.0;
            const double tz = m.sparse ? m.Tz[ds * m.n_rows + row] : 0.0;
            const double c0 = wc * R_(s);
            col[0] = c0;
            LG_(0) += c0 * tz;
            for (int i = 0; i < m.de; ++i) {     // total derivative: through the weights and through mus(z)
                const double v = dw(corner, i) * R_(s) + wc * DMUS_(i, s) * rs;
                col[1 + m.eff_axes[i]] = v;
                LG_(1 + m.eff_axes[i]) += v * tz;
            }
            const double cr = wc * MUS_(s);
            col[1 + m.d + s] = cr;
            LG_(1 + m.d + s) += cr * tz;
        }
    }
    LG_(0) += m.lgsum[ds];
    for (int q = 0; q < W; ++q) slot_lg[p * G + q] = LG_(q);
    cnt_off[p] = m.sparse ? m.cnt_off[ds] : ds * m.Bp;
    tiles[p] = (int32_t)(row_stride / kTile);
#undef BI_AT
#undef W_
#undef DMUS_
#undef MUS_
#undef R_
#undef LG_
#undef T_
#undef ID_
}

// bi_eval_grad for batches planned on the device (plain binned likelihoods): -> ll [P], grad [P][d + S], status [P]
int eval_grad_device(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, bool sparse, int G,
                     double* ll, double* grad, int32_t* status) {
    int rc = ensure_plan_tables(c);
    if (rc) return rc;
    const int S = c->S, d = c->d, W = 1 + d + S, NS = (1 << (int)c->eff_axes.size()) * S;
    PlanMeta m = plan_meta_of(c, sparse);
    DevBuf d_z, d_rs, d_ds, d_row, d_coef, d_cnt, d_tiles, d_perm, d_lg, d_st, d_part, d_flag, d_out;
    auto cleanup = [&]() { dev_free(d_z); dev_free(d_rs); dev_free(d_ds); dev_free(d_row); dev_free(d_coef); dev_free(d_cnt); dev_free(d_tiles);
                           dev_free(d_perm); dev_free(d_lg); dev_free(d_st); dev_free(d_part); dev_free(d_flag); dev_free(d_out); };
    const size_t nP = (size_t)P;
    int max_tiles = n_tiles_of(c);
    if (sparse) max_tiles = (int)(*std::max_element(c->h_c_np.begin(), c->h_c_np.end()) / kTile);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nbx = (int)std::min<int64_t>(max_tiles, std::max<int64_t>(1, (4 * slots + P - 1) / P));
    if ((rc = dev_alloc(c, d_z, nP * std::max(d, 1) * 8)) || (rate_scale && (rc = dev_alloc(c, d_rs, nP * S * 8))) ||
        (dataset && (rc = dev_alloc(c, d_ds, nP * 8))) || (rc = dev_alloc(c, d_row, nP * NS * 8)) || (rc = dev_alloc(c, d_coef, nP * NS * G * 8)) ||
        (rc = dev_alloc(c, d_cnt, nP * 8)) || (rc = dev_alloc(c, d_tiles, nP * 4)) || (rc = dev_alloc(c, d_perm, nP * G * 8)) ||
        (rc = dev_alloc(c, d_lg, nP * G * 8)) || (rc = dev_alloc(c, d_st, nP * 4)) || (rc = dev_alloc(c, d_part, nP * nbx * G * 8)) ||
        (rc = dev_alloc(c, d_flag, nP * nbx * G * 4)) || (rc = dev_alloc(c, d_out, nP * W * 8))) { cleanup(); return rc; }
    hipError_t e = hipSuccess;
    if (d) e = hipMemcpyAsync(d_z.p, z, nP * d * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && rate_scale) e = hipMemcpyAsync(d_rs.p, rate_scale, nP * S * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && dataset) e = hipMemcpyAsync(d_ds.p, dataset, nP * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const size_t lds = (size_t)kGradFillDoubles * kGradFillThreads * sizeof(double);                   // 112 KB of the CU's 160
        e = hipFuncSetAttribute((const void*)k_grad_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            hipLaunchKernelGGL(k_grad_fill, dim3((unsigned)((P + kGradFillThreads - 1) / kGradFillThreads)), dim3(kGradFillThreads), lds, c->stream, m, P, (const double*)d_z.p,
                           rate_scale ? (const double*)d_rs.p : nullptr, dataset ? (const int64_t*)d_ds.p : nullptr, G, (int64_t*)d_row.p,
                           (double*)d_coef.p, (int64_t*)d_cnt.p, (int32_t*)d_tiles.p, (int64_t*)d_perm.p, (double*)d_lg.p, (int32_t*)d_st.p);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { cleanup(); return fail(c, BI_ERR_HIP, "bi_eval_grad (device planning): %s", hipGetErrorString(e)); }
    LaunchArgs a{};
    a.ps = sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
    a.counts = sparse ? (const double*)c->cnt_c.p : (const double*)c->counts.p;
    a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = max_tiles; a.chunks = (int)c->tile_chunks;
    for (int64_t i0 = 0; i0 < P; i0 += 65535) {
        const int64_t ni = std::min<int64_t>(65535, P - i0);
        LaunchArgs b = a;
        b.rowoff = (const int64_t*)d_row.p + i0 * NS;
        b.coef = (const double*)d_coef.p + i0 * NS * G;
        b.item_cnt = (const int64_t*)d_cnt.p + i0;
        b.item_tiles = (const int32_t*)d_tiles.p + i0;
        b.partial = (double*)d_part.p + i0 * nbx * G;
        b.pflags = (unsigned*)d_flag.p + i0 * nbx * G;
        launch_morph_grad(c, G, b, dim3((unsigned)nbx, (unsigned)ni), false);
        const int64_t n_slots = ni * G;
        const int lanes = nbx > 64 ? kThreads : 64;
        const int per_block = kThreads / lanes;
        hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0, c->stream,
                           (const double*)b.partial, (const unsigned*)b.pflags, nbx, G, lanes, n_slots, (const int64_t*)d_perm.p + i0 * G,
                           (const double*)d_lg.p + i0 * G, (double*)d_out.p, (int32_t*)nullptr);
    }
    std::vector<double> h_out(nP * W);
    std::vector<int32_t> h_st(nP);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_out.data(), d_out.p, h_out.size() * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_st.data(), d_st.p, nP * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_grad (device planning): %s", hipGetErrorString(e));
    const double ninf = -std::numeric_limits<double>::infinity(), qnan = std::numeric_limits<double>::quiet_NaN();
    for (int64_t p = 0; p < P; ++p) {
        if (status) status[p] = h_st[(size_t)p];
        if (h_st[(size_t)p]) {
            ll[p] = ninf;
            for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = qnan;
        } else {
            ll[p] = h_out[(size_t)p * W];
            for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = h_out[(size_t)p * W + 1 + j];
        }
    }
    return BI_OK;
}

// grad_mode (eval_grad_mfma): the plan of a gradient batch for the matrix cores -- group tables always, the linear part
// -sum_b mu_b in the per-point constant, no scan / split decisions, no partial-sum buffers, and the device copies of z and
// rate_scale stay with the plan (the finish kernel rebuilds every point's derivative coefficients from them)
int plan_points_device(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, bool sparse,
                       bi_plan** out, int share_rank, int share_world, bool resident, bool grad_mode) {
    int rc = ensure_plan_tables(c);
    if (rc) return rc;
    const int S = c->S, d = c->d, de = (int)c->eff_axes.size(), nc = 1 << de;
    const bool bb = c->bb_source >= 0;
    const int NS = bb ? nc * (S - 1) + 2 * nc : nc * S;      // Beeston-Barlow: the other sources' rows, its own template rows, its Monte-Carlo counts
    // Beeston-Barlow with `bb_exact = 1` wants N(z) in numpy's summation order for every point: the host planner's extra pass
    if (bb && c->bb_exact == 1) return kPlanNeedsHost;
    PlanMeta m = plan_meta_of(c, sparse);                    // (split scans switch m.sparse on below)
    // Beeston-Barlow batches of at least scan_bb_min points: work items of 16 points for the matrix-core kernel (k_scan_bb), if a
    // variant holds the model's streams; smaller batches keep items of bb_max_group points for k_morph_reduce<G, true>
    const int bb_kgt = (bb && c->scan_bb && !grad_mode && !c->unbinned && P >= c->scan_bb_min) ? scan_bb_variant(nc * (S - 1), nc) : 0;
    if (bb_kgt) m.G = 16;
    const int G = m.G;

    bi_plan* plan = new bi_plan();
    plan->P = P; plan->sparse = sparse; plan->epoch = c->epoch; plan->device_planned = true; plan->no_reuse = false;
    DevBuf d_z, d_rs, d_ds, d_keys, d_keys2, d_idx, d_idx2, d_a, d_b, d_tmp, d_scal;
    auto cleanup = [&]() { dev_free(d_z); dev_free(d_rs); dev_free(d_ds); dev_free(d_keys);
                           dev_free(d_keys2); dev_free(d_idx); dev_free(d_idx2); dev_free(d_a); dev_free(d_b); dev_free(d_tmp); dev_free(d_scal); };
    auto abort_plan = [&](int code) { cleanup(); free_plan_buffers(plan); delete plan; return code; };
    const size_t nP = (size_t)P;
    // resident: z / rate_scale / dataset ARE device arrays (bi_plan_points_resident) and are read where they lie
    if ((!resident && (rc = dev_alloc(c, d_z, nP * std::max(d, 1) * sizeof(double)))) ||
        (rc = dev_alloc(c, d_keys, nP * 8)) || (rc = dev_alloc(c, d_keys2, nP * 8)) ||
        (rc = dev_alloc(c, d_idx, nP * 8)) || (rc = dev_alloc(c, d_idx2, nP * 8)) || (rc = dev_alloc(c, d_a, nP * 8)) ||
        (rc = dev_alloc(c, d_b, nP * 8)) || (rc = dev_alloc(c, d_scal, 128)) ||
        (rc = dev_alloc(c, plan->status, nP * sizeof(int32_t))) || (rc = dev_alloc(c, plan->out, nP * sizeof(double))))
        return abort_plan(rc);
    hipError_t e = hipSuccess;
    const double* z_dev = resident ? z : (const double*)d_z.p;
    const double* rs_dev = resident ? rate_scale : nullptr;
    const int64_t* ds_dev = resident ? dataset : nullptr;
    if (d && !resident) e = hipMemcpyAsync(d_z.p, z, nP * d * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && rate_scale && !resident) {
        if ((rc = dev_alloc(c, d_rs, nP * S * sizeof(double)))) return abort_plan(rc);
        e = hipMemcpyAsync(d_rs.p, rate_scale, nP * S * sizeof(double), hipMemcpyHostToDevice, c->stream);
        rs_dev = (const double*)d_rs.p;
    }
    if (e == hipSuccess && dataset && !resident) {
        if ((rc = dev_alloc(c, d_ds, nP * 8))) return abort_plan(rc);
        e = hipMemcpyAsync(d_ds.p, dataset, nP * 8, hipMemcpyHostToDevice, c->stream);
        ds_dev = (const int64_t*)d_ds.p;
    }
    if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning upload: %s", hipGetErrorString(e)));
    const unsigned nblk = (unsigned)((P + kThreads - 1) / kThreads);
    // [0] n_valid  [1] n_items  [2] sum of tiles  [3] largest group (gradient batches)  [4] Beeston-Barlow points at which some
    // bin can have U_b == 0  [5] points with an infinite rate that the reference evaluates (sources that may go negative)
    int64_t* scal = (int64_t*)d_scal.p;
    HIP_TRY(c, hipMemsetAsync(scal, 0, 128, c->stream));
    if ((size_t)m.n_rows * sizeof(double) <= (size_t)32 * 1024)
        hipLaunchKernelGGL(k_plan_geometry<true>, dim3(nblk), dim3(kThreads), (size_t)m.n_rows * sizeof(double), c->stream, m, P, z_dev, rs_dev, ds_dev,
                           (uint64_t*)d_keys.p, (int64_t*)d_idx.p, (int32_t*)plan->status.p, (unsigned long long*)(scal + 5));
    else
        hipLaunchKernelGGL(k_plan_geometry<false>, dim3(nblk), dim3(kThreads), 0, c->stream, m, P, z_dev, rs_dev, ds_dev,
                           (uint64_t*)d_keys.p, (int64_t*)d_idx.p, (int32_t*)plan->status.p, (unsigned long long*)(scal + 5));
    // sort (key, point) pairs: keys are cell * T + dataset < A * T, rejected points carry A * T -- the radix sort walks only the
    // bits that can differ (C2: 7 bits, one pass over the pairs; round 4 sorted all 64 bits of a key whose rejected points were ~0)
    size_t tmp_bytes = 0;
    int end_bit = 1;
    while (end_bit < 64 && (m.bad_key >> end_bit) != 0) ++end_bit;
    // at most 1024 key values (a profile scan's cells: T = 1): the counting sort, one pass whatever the bits
    const bool count_sort = m.bad_key + 1 <= 1024 && c->plan_count_sort;
    if (count_sort)
        (void)prim_count_sort_pairs(nullptr, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const int64_t*)d_idx.p,
                                    (int64_t*)d_idx2.p, (size_t)P, (uint64_t)m.bad_key + 1, c->stream);
    else
        (void)prim_sort_pairs(nullptr, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const int64_t*)d_idx.p,
                              (int64_t*)d_idx2.p, (size_t)P, 0u, (unsigned)end_bit, c->stream);
    size_t scan_bytes = 0, scan_bytes2 = 0;
    (void)prim_inclusive_scan_max(nullptr, scan_bytes, (const int64_t*)d_a.p, (int64_t*)d_b.p, (size_t)P, c->stream);
    (void)prim_inclusive_scan_sum(nullptr, scan_bytes2, (const int64_t*)d_a.p, (int64_t*)d_b.p, (size_t)P, c->stream);
    if ((rc = dev_alloc(c, d_tmp, std::max({tmp_bytes, scan_bytes, scan_bytes2, (size_t)256})))) return abort_plan(rc);
    size_t tb = d_tmp.bytes;
    if (count_sort)
        e = prim_count_sort_pairs(d_tmp.p, tb, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const int64_t*)d_idx.p,
                                  (int64_t*)d_idx2.p, (size_t)P, (uint64_t)m.bad_key + 1, c->stream);
    else
        e = prim_sort_pairs(d_tmp.p, tb, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const int64_t*)d_idx.p,
                            (int64_t*)d_idx2.p, (size_t)P, 0u, (unsigned)end_bit, c->stream);
    if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning sort: %s", hipGetErrorString(e)));
    // few distinct keys (A * T + 1 <= 65 536: every named configuration): the group structure from per-key tables, one round trip
    const int64_t K = (int64_t)m.bad_key;
    const bool by_tables = K <= 65536 && c->plan_tables;
    DevBuf d_kstart, d_tab_start, d_tab_item;
    int64_t h_scal[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, h_inf = 0;
    if (by_tables) {
        if ((rc = dev_alloc(c, d_kstart, (size_t)(K + 1) * 8)) || (rc = dev_alloc(c, d_tab_start, (size_t)K * 8)) || (rc = dev_alloc(c, d_tab_item, (size_t)K * 8)) ||
            (rc = dev_alloc(c, plan->grp_first, (size_t)K * 8)) || (rc = dev_alloc(c, plan->grp_items, (size_t)K * 4))) {
            dev_free(d_kstart); dev_free(d_tab_start); dev_free(d_tab_item);
            return abort_plan(rc);
        }
        e = hipMemsetAsync(d_kstart.p, 0xFF, (size_t)(K + 1) * 8, c->stream);
        hipLaunchKernelGGL(k_plan_key_bounds, dim3(nblk), dim3(kThreads), 0, c->stream, (const uint64_t*)d_keys2.p, P, (int64_t*)d_kstart.p);
        hipLaunchKernelGGL(k_plan_key_tables, dim3(1), dim3(kThreads), 0, c->stream, (const int64_t*)d_kstart.p, K, P, share_rank, share_world, G,
                           (int64_t*)d_tab_start.p, (int64_t*)d_tab_item.p, (int64_t*)plan->grp_first.p, (int32_t*)plan->grp_items.p, scal);
    } else {
        hipLaunchKernelGGL(k_plan_count_valid, dim3(1), dim3(64), 0, c->stream, (const uint64_t*)d_keys2.p, P, m.bad_key, scal);
    }
    // (the tables live until the fill kernel has run: freed with the other planning scratch -- stream order keeps them valid)
    struct TabGuard { DevBuf &a, &b, &c3; ~TabGuard() { dev_free(a); dev_free(b); dev_free(c3); } } tab_guard{d_kstart, d_tab_start, d_tab_item};
    if (e == hipSuccess) e = plan_report(c, ReportPiece{(const uint32_t*)scal, 32, 0});
    if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning: %s", hipGetErrorString(e)));
    memcpy(h_scal, c->plan_host, sizeof(h_scal));
    h_inf = h_scal[5];
    if (h_inf > 0)
        return abort_plan(fail(c, BI_ERR_INVALID, "%lld points carry an infinite rate of a source that may go negative: those are answered on "
                                                  "the host (bi_plan_points / bi_eval with host arrays)", (long long)h_inf));
    const int64_t n_valid_all = h_scal[0];           // (table route: the share's window [7], [8] is the same arithmetic as below)
    plan->n_bad = P - n_valid_all;
    // the share of this context: everything, or a contiguous range of the sorted list (the arrays below are windows on it)
    const bool shared = share_world > 1;
    int64_t share_lo = 0, share_hi = n_valid_all;
    if (shared) {
        const int64_t base = n_valid_all / share_world, extra = n_valid_all % share_world;
        share_lo = share_rank * base + std::min<int64_t>(share_rank, extra);
        share_hi = share_lo + base + (share_rank < extra ? 1 : 0);
        plan->shared = true;
        plan->share_world = share_world;
        plan->share_lo = share_lo;
        plan->share_hi = share_hi;
        plan->n_valid = n_valid_all;
    }
    const int64_t n_valid = share_hi - share_lo;
    const uint64_t* const keys_s = (const uint64_t*)d_keys2.p + share_lo;
    const int64_t* const idx_s = (const int64_t*)d_idx2.p + share_lo;
    m.share_order = shared ? 1 : 0;
    if (n_valid > 0) {
        const unsigned vblk = (unsigned)((n_valid + kThreads - 1) / kThreads);
        int64_t n_groups = 0;
        if (by_tables) {
            h_scal[1] = h_scal[1];                      // n_items, n_groups came with the tables
            n_groups = h_scal[6];
        } else {
        hipLaunchKernelGGL(k_plan_heads, dim3(vblk), dim3(kThreads), 0, c->stream, keys_s, n_valid, (int64_t*)d_a.p);
        tb = d_tmp.bytes;
        (void)prim_inclusive_scan_max(d_tmp.p, tb, (const int64_t*)d_a.p, (int64_t*)d_b.p, (size_t)n_valid, c->stream);  // d_b = group start
        hipLaunchKernelGGL(k_plan_item_heads, dim3(vblk), dim3(kThreads), 0, c->stream, (const int64_t*)d_b.p, n_valid, G, (int64_t*)d_a.p);
        tb = d_tmp.bytes;
        (void)prim_inclusive_scan_sum(d_tmp.p, tb, (const int64_t*)d_a.p, (int64_t*)d_keys.p, (size_t)n_valid, c->stream);               // d_keys = item index + 1
        // groups of items sharing (cell, dataset): their number decides which kernels take the batch
        hipLaunchKernelGGL(k_plan_group_flags, dim3(vblk), dim3(kThreads), 0, c->stream, (const int64_t*)d_b.p, n_valid, (int64_t*)d_a.p);
        tb = d_tmp.bytes;
        (void)prim_inclusive_scan_sum(d_tmp.p, tb, (const int64_t*)d_a.p, (int64_t*)d_idx.p, (size_t)n_valid, c->stream);   // d_idx = group id + 1
        // (the two counts in ONE round trip: every host synchronisation of the planner is ~35 us of a call)
        e = hipMemcpyAsync(h_scal + 1, (const int64_t*)d_keys.p + (n_valid - 1), 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&n_groups, (const int64_t*)d_idx.p + (n_valid - 1), 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning: %s", hipGetErrorString(e)));
        }
        const int64_t n_items = h_scal[1];
        const bool mostly_empty = c->h_nz_off.size() == (size_t)c->T + 1 && c->h_nz_off.back() <= c->T * c->B / 8;
        const bool scan_shape = c->scan_mfma && !bb && !c->unbinned && c->ps_finite && NS <= 32 && n_groups <= 65535;
        // Every bin visited, mostly empty data, several items per cell: split the scan into the non-empty-bin pass (the
        // descriptors below then describe the compacted rows, as for a sparse plan) and a validity pass over all bins on
        // the matrix cores (k_scan_valid).  The two together are exact for templates and rates of either sign.
        const bool split = !grad_mode && !sparse && c->scan_split && scan_shape && c->compact_ready && c->dense_counts &&
                           n_items >= c->scan_min_items * n_groups;
        const bool compacted = sparse || split;
        // the matrix-core scan kernel pays when many items share a cell (it streams a cell's rows once per strip and
        // keeps them in registers): measured against k_morph_reduce 1.2x at 2, 1.3x at 8 and 1.7x at 128 items per cell
        // on sparse data; 1.0x at 4, 1.2x at 8 and 1.3x at 128 on dense data (where the per-bin logarithm is the
        // larger part of the work)
        const bool scan_ok = !grad_mode && !split && scan_shape &&
                             n_items >= c->scan_min_items * (mostly_empty ? 1 : 2) * n_groups &&
                             !(sparse && n_items > c->scan_sparse_max_items * n_groups);   // compacted rows, very long item lists: see scan_sparse_max_items
        m.sparse = compacted ? 1 : 0;
        m.linear_outside = (scan_ok || grad_mode) ? 1 : 0;
        plan->sparse = compacted;
        plan->classes.emplace_back();
        bi_plan::Class& k = plan->classes.back();
        k.G = G;
        k.n_items = n_items;
        const int n_tiles = n_tiles_of(c);
        const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
        int64_t max_tiles = n_tiles;
        if (compacted) max_tiles = *std::max_element(c->h_c_np.begin(), c->h_c_np.end()) / kTile;
        int64_t nbx = std::min<int64_t>(max_tiles, std::max<int64_t>(1, (4 * slots + n_items - 1) / n_items));
        if (n_items == 1) nbx = std::min<int64_t>(max_tiles, slots);
        if (c->xcd_affine && n_items > 1 && nbx > 4 && nbx < max_tiles) nbx = std::min<int64_t>(max_tiles, (nbx + 7) / 8 * 8);
        if (bb_kgt != 0 && n_groups <= 65535) {
            // k_scan_bb: a block = four work items of a group over a range of 16-bin tiles; about three rounds of the resident blocks
            // over all quads, every block at least 64 tiles
            const int64_t quads = (n_items + 3) / 4 + n_groups;
            const int resident = std::max(1, occupancy_scan_bb(bb_kgt));
            const int64_t tiles16 = (c->B + 15) / 16;
            nbx = std::max<int64_t>(1, std::min<int64_t>({(3 * (int64_t)c->prop.multiProcessorCount * resident + quads - 1) / quads, (tiles16 + 63) / 64, (int64_t)65535}));
        }
        k.nbx = (int)nbx;
        const size_t ni = (size_t)n_items;
        if ((rc = dev_alloc(c, k.rowoff, ni * NS * 8)) || (rc = dev_alloc(c, k.coef, ni * NS * G * 8)) || (rc = dev_alloc(c, k.aux, ni * G * 16)) ||
            (rc = dev_alloc(c, k.item_cnt, ni * 8)) || (rc = dev_alloc(c, k.item_tiles, ni * 4)) || (rc = dev_alloc(c, k.perm, ni * G * 8)) ||
            (rc = dev_alloc(c, k.slot_lg, ni * G * 8)) ||
            (!grad_mode && ((rc = dev_alloc(c, k.partial, ni * k.nbx * G * sizeof(double))) ||
                            (rc = dev_alloc(c, k.pflags, ni * k.nbx * G * sizeof(unsigned))))) ||
            (split && (rc = dev_alloc(c, k.rowoff_full, ni * NS * 8))))
            return abort_plan(rc);
        // (no memset of the descriptor arrays: the fill kernel writes every slot of every item, padding slots included --
        //  256 MB of zeros for a 10^6-point scan used to go first)
        {
            // threads per block by the LDS its per-thread arrays take: d + 2 S + nc doubles and the point's index per thread
            const size_t per_thread = (size_t)(d + 2 * S + nc) * sizeof(double) + sizeof(int64_t);
            int bd = kThreads;
            while (bd > 64 && per_thread * bd > (size_t)64 * 1024) bd >>= 1;
            // the rate table and the per-row constants behind them where they fit (16 KB each)
            const size_t table = (size_t)m.n_rows * sizeof(double);
            const int mus_lds = table <= (size_t)16 * 1024;
            const int tz_lds = mus_lds && (m.linear_outside || (m.sparse && c->T == 1));
            const size_t lds = per_thread * bd + (mus_lds ? table : 0) + (tz_lds ? table : 0);
            if (lds > (size_t)48 * 1024) e = hipFuncSetAttribute((const void*)k_plan_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning: %s", hipGetErrorString(e)));
            hipLaunchKernelGGL(k_plan_fill, dim3((unsigned)((n_valid + bd - 1) / bd)), dim3((unsigned)bd), lds, c->stream, m, n_valid, keys_s,
                           idx_s, by_tables ? (const int64_t*)nullptr : (const int64_t*)d_b.p, (const int64_t*)d_keys.p,
                           (const int64_t*)d_tab_start.p, (const int64_t*)d_tab_item.p, z_dev,
                           rs_dev, (int64_t*)k.rowoff.p, (double*)k.coef.p, (int64_t*)k.item_cnt.p,
                           (int32_t*)k.item_tiles.p, (int64_t*)k.perm.p, (double*)k.slot_lg.p, (unsigned long long*)(scal + 2),
                           split ? (int64_t*)k.rowoff_full.p : (int64_t*)nullptr, bb ? (double*)k.aux.p : (double*)nullptr,
                           bb ? (unsigned long long*)(scal + 4) : (unsigned long long*)nullptr, mus_lds, tz_lds);
        }
        e = hipGetLastError();
        bool tables_done = by_tables;          // (k_plan_key_tables wrote them, and the largest group into scal[3])
        auto group_tables = [&]() -> int {
            int rc2;
            if (tables_done) return BI_OK;
            tables_done = true;
            if ((rc2 = dev_alloc(c, plan->grp_first, (size_t)n_groups * 8)) || (rc2 = dev_alloc(c, plan->grp_items, (size_t)n_groups * 4)))
                return rc2;
            hipLaunchKernelGGL(k_plan_group_first, dim3(vblk), dim3(kThreads), 0, c->stream, (const int64_t*)d_b.p, (const int64_t*)d_idx.p,
                               (const int64_t*)d_keys.p, n_valid, (int64_t*)plan->grp_first.p);
            hipLaunchKernelGGL(k_plan_group_items, dim3((unsigned)((n_groups + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                               (const int64_t*)plan->grp_first.p, n_groups, n_items, (int32_t*)plan->grp_items.p,
                               (grad_mode || bb_kgt) ? (unsigned long long*)(scal + 3) : (unsigned long long*)nullptr);
            return BI_OK;
        };
        // (the group tables go out before the read-back, so that a gradient batch's largest group and -- for the scan kernels --
        //  the tables themselves travel with it: every host synchronisation of the planner is ~35 us of a call)
        const bool bb_scan = bb_kgt != 0 && n_groups <= 65535;
        const bool tables_early = grad_mode || split || scan_ok || bb_scan;
        if (e == hipSuccess && tables_early && (rc = group_tables())) return abort_plan(rc);
        std::vector<int64_t> h_grp_first;
        std::vector<int32_t> h_grp_items;
        int64_t h_zero_u = 0, h_max = 0;
        {
            // one report: the counters and -- where the item lists may be cut into chunks below -- the group tables
            const bool want_tables = (split || scan_ok) && c->scan_chunk && n_groups <= 4096;
            const uint32_t ng = want_tables ? (uint32_t)n_groups : 0u;
            if (e == hipSuccess)
                e = plan_report(c, ReportPiece{(const uint32_t*)scal, 32, 0}, ReportPiece{(const uint32_t*)plan->grp_first.p, 2 * ng, 32},
                                ReportPiece{(const uint32_t*)plan->grp_items.p, ng, 32 + 2 * 4096});
            if (e == hipSuccess) {
                const int64_t* hs = (const int64_t*)c->plan_host;
                h_scal[2] = hs[2];
                if (bb) h_zero_u = hs[4];
                if (grad_mode || bb_scan) h_max = hs[3];
                if (want_tables) {
                    h_grp_first.assign((const int64_t*)((const uint32_t*)c->plan_host + 32), (const int64_t*)((const uint32_t*)c->plan_host + 32) + ng);
                    h_grp_items.assign((const int32_t*)c->plan_host + 32 + 2 * 4096, (const int32_t*)c->plan_host + 32 + 2 * 4096 + ng);
                }
            }
        }
        if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning fill: %s", hipGetErrorString(e)));
        // Beeston-Barlow points at which some bin can have U_b == 0: the reference's first-root assertion then hangs on the
        // last bits of N(z), which only the host planner's pass in numpy's summation order reproduces (bb_exact_totals)
        if (bb && c->bb_exact && h_zero_u > 0) return abort_plan(kPlanNeedsHost);
        plan->bytes = (int64_t)sizeof(double) * ((int64_t)NS + 1) * (compacted ? h_scal[2] * kTile : n_items * c->B);
        plan->launches = (n_items + 65534) / 65535;

        if (grad_mode) {
            plan->n_groups = n_groups;
            plan->max_group_items = h_max;
            plan->max_item_tiles = (int)max_tiles;
        }
        if (bb_scan) {
            plan->bb_kgt = bb_kgt;
            plan->n_groups = n_groups;
            plan->max_group_items = h_max;
            plan->launches = 1;
        }
        // A group's strips are spread over W = 4 b waves (b blocks); a block lasts as long as its busiest wave, ceil(strips / W)
        // strips, and the blocks of all groups run in ceil(b * n_groups / resident blocks) rounds -- the last of which is
        // as long as the others however few blocks it holds.  Choose b for the smallest rounds x strips-per-wave (ties: the
        // fewer waves): at C2 the 320 compacted strips of a cell go to 12 blocks of 7 strips in ONE round instead of 20 blocks
        // of 4 strips in two, dense data gets 8 full rounds instead of 2.7.  `resident` = blocks of that kernel variant a CU
        // holds (asked from the runtime); slot_cap bounds the waves where every wave owns a partial slot per item.
        // pipe_bound (k_scan_sorted): the kernel keeps the fp64 matrix pipe ~87 % busy, so the waves of a SIMD SHARE it
        // and a CU's time is the sum of its blocks' strips, not the longest of them: what counts is the busiest CU --
        // ceil(blocks / CUs) blocks when all are resident at once (640 blocks of 4 strips on 256 CUs are 3 x 4 = 12
        // units on half of the CUs, 512 blocks of 5 strips are 10 on all of them: 11.6 instead of 13.8 ms for the 10^6-
        // point scan of C2), the mean load plus a quarter of a block generation's length when they run in several
        // rounds (measured: tools/tune_scan_sorted.py).  Among the splits within 3 % of the best the coarsest is taken
        // (fewer partial slots for the finish to add up).
        auto waves_per_group = [&](int64_t strips, int resident, int64_t slot_cap, bool pipe_bound = false) -> int64_t {
            const int64_t capacity = (int64_t)c->prop.multiProcessorCount * std::max(1, resident);
            const int64_t b_max = std::max<int64_t>(1, std::min<int64_t>(strips / 4, slot_cap / 4));
            if (c->scan_waves_per_cu > 0) {      // forced (tuning): that many waves per CU over all groups, evened out
                int64_t blocks = std::max<int64_t>(1, (c->scan_waves_per_cu * c->prop.multiProcessorCount + 4 * n_groups - 1) / (4 * n_groups));
                blocks = std::min<int64_t>(blocks, b_max);
                const int64_t per_wave = (strips + 4 * blocks - 1) / (4 * blocks);
                return (strips + 4 * per_wave - 1) / (4 * per_wave) * 4;
            }
            auto cost_of = [&](int64_t b) -> double {
                const int64_t per_wave = (strips + 4 * b - 1) / (4 * b);
                if ((strips + 4 * per_wave - 1) / (4 * per_wave) != b) return 1e300;          // not an evened-out split
                return (double)((b * n_groups + capacity - 1) / capacity) * (double)per_wave;
            };
            if (pipe_bound) {
                const int64_t n_cu = std::max(1, c->prop.multiProcessorCount);
                auto pipe_cost = [&](int64_t b) -> double {
                    const int64_t per_wave = (strips + 4 * b - 1) / (4 * b);
                    if ((strips + 4 * per_wave - 1) / (4 * per_wave) != b) return 1e300;
                    const int64_t blocks = b * n_groups;
                    if (blocks <= capacity) {
                        const int64_t per_cu = (blocks + n_cu - 1) / n_cu;
                        // fewer waves on a SIMD than it could hold: less to cover a wave's latencies with (measured on the
                        // dense-data scan of C2: two waves per SIMD 131 ms, three 123 ms; one wave ~ +25 %)
                        const double thin = per_cu >= resident ? 1.0 : (per_cu == 1 ? 1.25 : 1.07);
                        return (double)(per_cu * per_wave) * thin;
                    }
                    return (double)blocks / (double)n_cu * (double)per_wave + 0.25 * (double)resident * (double)per_wave;
                };
                double best = 1e300;
                for (int64_t b = 1; b <= b_max; ++b) best = std::min(best, pipe_cost(b));
                int64_t b_model = 1;
                for (int64_t b = 1; b <= b_max; ++b)
                    if (pipe_cost(b) <= 1.03 * best) { b_model = b; break; }
                // Since the strips of mixed counts are shared among a cell's waves (round 4) the measured optimum is simply
                // about THREE rounds of the resident waves over all cells, with at least four strips per wave (dense data of C2:
                // 48 waves per cell 125.6 ms, 128-384 122.0; 16 cells of 2500 strips: 836 waves per cell 11.1 ms, 252-504 10.4;
                // the 155 compacted strips of C2 keep their 32 waves: tools/tune_scan_split_shapes.py).  The model above decides
                // only where that target cannot be met.
                const int64_t b_cap = std::min<int64_t>(b_max, std::max<int64_t>(1, strips / 16));
                int64_t b_target = std::min<int64_t>(b_cap, std::max<int64_t>(1, (3 * capacity + n_groups / 2) / n_groups));
                // (only where the target means several rounds: within ONE round of resident blocks the model's count of blocks
                //  per CU decides -- 640 blocks on 256 CUs are 3 on half of them: 40 waves per cell 12.1 ms, 32 waves 10.1)
                if (b_target * n_groups > capacity)
                    for (int64_t b = b_target; b >= 1 && b * n_groups > capacity; --b)
                        if (pipe_cost(b) < 1e299) return b * 4;                             // (the nearest evened-out split below the target)
                return b_model * 4;
            }
            double best_cost = 1e300;
            int64_t best_b = 1;
            for (int64_t b = 1; b <= b_max; ++b)
                if (cost_of(b) < best_cost) { best_cost = cost_of(b); best_b = b; }
            // Among the splits within 1 % of the best take the FINEST that still leaves a wave 8 strips per item list:
            // the blocks of one cell are dispatched together, so many short rounds keep few cells -- and their
            // coefficient lists -- in flight at a time (L2), where one long round has every cell's list streaming at
            // once (measured at C2, dense data: 598 k evaluations/s with 8 rounds of 96 blocks per cell against 539 k
            // with one round of 12); below ~8 strips per wave the re-read of the coefficient lists by every wave shows.
            for (int64_t b = b_max; b > best_b; --b)
                if (cost_of(b) <= 1.01 * best_cost && (strips + 4 * b - 1) / (4 * b) >= 8) { best_b = b; break; }
            return best_b * 4;
        };
        // Few groups with long item lists -- a rank's share of a dealt scan: 8 of C2's 64 cells with ~1000 work items each --
        // cannot fill the chip by their strips alone: a group's waves split its STRIPS, every wave walks ALL its items, and a
        // group gives at most strips / 16 blocks (8 cells x 9 blocks = 72 blocks for 256 CUs: the share of one of 8 ranks took
        // 2.1 ms where an eighth of the whole scan is 1.3).  So the item lists are cut as well: chunks of a group's items
        // (multiples of four: the kernels work quads of items) become groups of their own -- same rows, other items, other
        // partial slots; nothing changes for the kernels.  The tables are small (one entry per group): read back, cut, sent again.
        auto chunk_groups = [&](int64_t strips, int resident) -> int {
            if (h_grp_items.empty() || n_groups < 1) return BI_OK;          // (tables of more than 4096 groups are not cut: plenty of blocks)
            const int64_t capacity = (int64_t)c->prop.multiProcessorCount * std::max(1, resident);
            const int64_t b_cap = std::max<int64_t>(1, strips / 16);
            if (b_cap * n_groups >= 2 * capacity) return BI_OK;
            const int64_t pieces = (3 * capacity + b_cap * n_groups - 1) / (b_cap * n_groups);
            const int64_t longest = *std::max_element(h_grp_items.begin(), h_grp_items.end());
            const int64_t chunk = std::max<int64_t>(16, ((longest + pieces - 1) / pieces + 3) / 4 * 4);
            if (chunk >= longest) return BI_OK;
            std::vector<int64_t>& first2 = plan->h_grp_first;                 // (kept with the plan: the upload below is asynchronous)
            std::vector<int32_t>& items2 = plan->h_grp_items;
            first2.clear();
            items2.clear();
            for (int64_t g = 0; g < n_groups; ++g)
                for (int64_t o = 0; o < h_grp_items[(size_t)g]; o += chunk) {
                    first2.push_back(h_grp_first[(size_t)g] + o);
                    items2.push_back((int32_t)std::min<int64_t>(chunk, h_grp_items[(size_t)g] - o));
                }
            if ((int64_t)first2.size() > 65535) return BI_OK;
            int rc2;
            if ((rc2 = dev_upload(c, plan->grp_first, first2)) || (rc2 = dev_upload(c, plan->grp_items, items2))) return rc2;
            n_groups = (int64_t)first2.size();
            return BI_OK;
        };
        if (split) {
            if ((rc = group_tables()) || (rc = dev_alloc(c, plan->bad, ni * kDevG * sizeof(unsigned)))) return abort_plan(rc);
            plan->bytes += (int64_t)sizeof(double) * NS * c->B * n_groups;     // every cell's rows once more, in full
            if ((rc = chunk_groups((int64_t)n_tiles * (kTile / 64), scan_resident_blocks(true, 4, NS)))) return abort_plan(rc);
            plan->valid = true;
            plan->n_groups = n_groups;
            plan->scan_cb = 4;
            plan->valid_nslots = (int)waves_per_group((int64_t)n_tiles * (kTile / 64), scan_resident_blocks(true, 4, NS),
                                                      (int64_t)1 << 40);       // (the pass keeps no per-wave partial sums)
            c->last_valid_nslots = plan->valid_nslots;
            plan->launches += 1;
            if ((e = hipGetLastError()) != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning groups: %s", hipGetErrorString(e)));
        }
        if (scan_ok) {
            if ((rc = group_tables())) return abort_plan(rc);
            // strips of 32 bins where (nearly) every bin needs its logarithm -- dense data, or the compacted rows of the
            // non-empty-bin form --, 64 where most blocks of 16 bins hold no data at all
            const int cb = c->scan_cb ? (int)c->scan_cb : ((mostly_empty && !sparse) ? 4 : 2);
            plan->use_scan = true;
            // dense data, rows in full: the count-sorted copy, if it can be had (every bin still visited, in another order)
            plan->sorted = !compacted && cb == 2 && ensure_sorted_rows(c);
            plan->by_count = plan->sorted || (compacted && cb == 2 && c->compact_sorted);
            // rows in count order go to k_scan_sorted: strips of 64 bins, four work items at a time
            const int strip_cb = plan->by_count ? 4 : cb;
            plan->scan_cb = strip_cb;
            plan->n_groups = n_groups;
            // every wave owns one partial slot per item: the split is bounded by the memory the slots may take (1 GiB)
            const int64_t slot_cap = std::max<int64_t>(4, ((int64_t)1 << 30) / std::max<int64_t>(1, (int64_t)ni * kDevG * (int64_t)sizeof(double)));
            const int resident = scan_resident_blocks(false, strip_cb, NS, plan->by_count);
            if ((rc = chunk_groups(max_tiles * (kTile / (16 * strip_cb)), resident))) return abort_plan(rc);
            plan->n_groups = n_groups;
            k.nbx = (int)waves_per_group(max_tiles * (kTile / (16 * strip_cb)), resident, slot_cap, plan->by_count);
            c->last_scan_nslots = k.nbx;
            c->last_scan_resident = resident;
            dev_free(k.partial);
            dev_free(k.pflags);                     // the scan kernel raises no per-block flags (k_finish_scan reads none)
            if ((rc = dev_alloc(c, k.partial, ni * k.nbx * kDevG * sizeof(double)))) return abort_plan(rc);
            if ((e = hipGetLastError()) != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "device planning groups: %s", hipGetErrorString(e)));
            plan->launches = 1;
        }
    }
    if (shared) {                 // the sorted -> original index map outlives the planning: unsort_share reads it
        plan->sorted_idx = d_idx2;
        d_idx2 = DevBuf{};
    }
    if (grad_mode && !resident) {
        plan->keep_z = d_z; d_z = DevBuf{};
        plan->keep_rs = d_rs; d_rs = DevBuf{};
    }
    cleanup();
    *out = plan;
    return BI_OK;
}

// gathered [share_world][stride]: rank r's results in sorted order -> full [P] in the caller's point order; rejected
// points (sorted behind the valid ones) get -inf, as bi_run_plan gives them
__global__ __launch_bounds__(kThreads) void k_unsort_share(const double* __restrict__ gathered, int64_t stride, int world, int64_t n_valid,
                                                           int64_t P, const int64_t* __restrict__ sorted_idx, double* __restrict__ full) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= P) return;
    double v = -__builtin_inf();
    if (i < n_valid) {
        const int64_t base = n_valid / world, extra = n_valid % world;
        // ranks [0, extra) hold base + 1 positions, the others base
        const int64_t cut = extra * (base + 1);
        const int64_t r = i < cut ? i / (base + 1) : extra + (base ? (i - cut) / base : 0);
        const int64_t lo = r * base + (r < extra ? r : extra);
        v = gathered[r * stride + (i - lo)];
    }
    full[sorted_idx[i]] = v;
}

}  // namespace
