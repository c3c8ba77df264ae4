// bi_planning.h -- host half of a batched evaluation (bi_plan_points): per point the scalar work of
// blueice/likelihood.py:345-415 (bounds, morph weights, rates, unphysical-rate exit), then grouping by
// (grid cell, dataset) and packing into work items of G in {1,2,4,8,16} points that share one pass over the
// cell's template rows.  Everything per point / per item is independent, so both phases run on a few host
// threads for large batches (0.47 us per point single-threaded would otherwise cap the non-empty-bin form).
#pragma once

namespace {

// Host threads the planner may start per call.  One process per GPU: eight ranks on a node each starting 16 threads are
// 128 threads planning at once, on a pool that kills runs with large worker pools -- so the default is what THIS process
// may run on (its scheduling affinity, at most 16), and a launcher that knows the rank count narrows it further
// (bi_set_param("host_threads", n); process-wide: the pool is per call, not per context).
inline std::atomic<int>& host_threads_setting() {
    static std::atomic<int> v{0};
    return v;
}
inline int host_threads() {
    const int forced = host_threads_setting().load();
    if (forced > 0) return forced;
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::max(1u, std::thread::hardware_concurrency());
    return std::min(n, 16);
}

template <class F>
void parallel_for(int64_t n, int64_t grain, F body) {  // body(begin, end)
    const int64_t want = std::min<int64_t>((int64_t)host_threads(), (n + grain - 1) / grain);
    if (want <= 1) { body(0, n); return; }
    std::vector<std::thread> pool;
    const int64_t step = (n + want - 1) / want;
    for (int64_t b = 0; b < n; b += step) pool.emplace_back(body, b, std::min(n, b + step));
    for (auto& t : pool) t.join();
}

struct PlanItem {
    int cls;          // index into kClassG
    int take;         // live points in the item (<= G)
    int64_t first;    // index of its first point in the sorted point list
    int64_t slot;     // index of the item within its class
};

constexpr int kClassG[5] = {1, 2, 4, 8, 16};

constexpr int kPlanNeedsHost = 1;   // plan_points_device: this batch needs the host planner (exact Beeston-Barlow totals) -- internal, not a code of the ABI

int plan_points_device(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, bool sparse,
                       bi_plan** out, int share_rank, int share_world, bool resident, bool grad_mode);

// transient: the plan is run once and destroyed inside the calling entry point (bi_eval); small ones then travel in
// one packed copy and deliver their results to pinned host memory.
int plan_points(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, bi_plan** out,
                bool transient = false, int share_rank = 0, int share_world = 1) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!out) return fail(c, BI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (P < 0) return fail(c, BI_ERR_INVALID, "P < 0");
    if (c->d > 0 && P > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    HIP_TRY(c, hipSetDevice(c->device));

    const int S = c->S, d = c->d;
    const bool bb = c->bb_source >= 0;
    const int nc = 1 << (int)c->eff_axes.size();
    const int n0 = bb ? nc * (S - 1) : nc * S;
    const int n1 = bb ? nc : 0, n2 = bb ? nc : 0;
    const int NS = n0 + n1 + n2;
    bool any_neg = false;
    for (int q = 0; q < S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    const bool sparse = c->sparse && c->compact_ready && c->ps_nonneg && !bb && !any_neg && !c->unbinned;
    const int64_t n_rows = c->A * S;
    if (!sparse && !c->dense_counts)
        return fail(c, BI_ERR_STATE, "the datasets exist only as non-empty-bin lists (device-generated toys): point "
                                     "evaluations need the compacted templates (sparse mode, budget) or bi_eval_datasets");

    // large plain batches are planned on the device (bi_planning_device.h) unless groups would be tiny
    // (every device-planned item has 16 slots): expected points per (cell, dataset) group >= 8
    bool inf_scale = false;      // infinite rates are answered point by point on the host (inf_rate_value): host planner
    if (any_neg && rate_scale)
        for (int64_t i = 0; i < P * S && !inf_scale; ++i) inf_scale = std::isinf(rate_scale[i]);
    if (share_world > 1) {       // a share of a dealt scan: the dealing IS the device planner's sort
        if (share_rank < 0 || share_rank >= share_world) return fail(c, BI_ERR_INVALID, "share %d outside [0,%d)", share_rank, share_world);
        if (inf_scale || P > (int64_t)1 << 30)
            return fail(c, BI_ERR_INVALID, "shares of a scan are planned on the device: not available with infinite rate scales");
        rc = plan_points_device(c, P, z, rate_scale, dataset, sparse, out, share_rank, share_world, false, false);
        if (rc == kPlanNeedsHost)
            return fail(c, BI_ERR_INVALID, "shares of a scan are planned on the device: this Beeston-Barlow batch has points at which some bin can "
                                           "have U_b == 0 (or bb_exact = 1) and needs the host planner's exact totals");
        return rc;
    }
    if (!inf_scale && c->device_plan_min > 0 && P >= c->device_plan_min && P <= (int64_t)1 << 30) {
        int64_t cells = 1;
        for (int ax : c->eff_axes) cells *= c->n_anchor[(size_t)ax] - 1;
        const int64_t groups = cells * (dataset ? c->T : 1);
        // (Beeston-Barlow work items hold bb_max_group points: the device planner from 4 expected points per group on)
        if (groups <= P / (bb ? 4 : 8)) {
            rc = plan_points_device(c, P, z, rate_scale, dataset, sparse, out, 0, 1, false, false);
            if (rc != kPlanNeedsHost) return rc;         // (else: exact Beeston-Barlow totals wanted -- the host planner below)
        }
    }

    // ---- phase 1: per point geometry, rates, early exits (parallel) ---------------------------------
    std::vector<int32_t> st((size_t)P, 0);
    std::vector<int64_t> cell((size_t)P, -1);
    std::vector<double> wts((size_t)P * nc), rates((size_t)P * S);
    parallel_for(P, 4096, [&](int64_t lo, int64_t hi) {
        PointGeom g;
        for (int64_t p = lo; p < hi; ++p) {
            const int64_t ds = dataset ? dataset[p] : 0;
            if (ds < 0 || ds >= c->T) { st[(size_t)p] = BI_ST_BAD_DATASET; continue; }
            if (!point_geometry(c, z ? z + p * d : nullptr, g)) { st[(size_t)p] = BI_ST_OUT_OF_BOUNDS; continue; }
            double* r = &rates[(size_t)p * S];
            interp_mus(c, g, r);
            if (rate_scale) for (int s = 0; s < S; ++s) r[s] *= rate_scale[p * S + s];
            if (!rates_physical(c, r)) { st[(size_t)p] = BI_ST_UNPHYSICAL; continue; }
            std::copy(g.w.begin(), g.w.end(), wts.begin() + (size_t)p * nc);
            cell[(size_t)p] = g.cell_anchor;
        }
    });

    // ---- phase 2: group by (cell, dataset) ----------------------------------------------------------
    struct Pt { int64_t key, idx; };
    std::vector<Pt> pts;
    std::vector<int64_t> bad, nanv;      // points answered without a launch: -inf / nan
    pts.reserve((size_t)P);
    const bool answer_inf = !bb && !c->unbinned && c->ps_finite && c->dense_counts;
    for (int64_t p = 0; p < P; ++p) {
        if (st[(size_t)p]) { bad.push_back(p); continue; }
        if (answer_inf && any_neg && has_infinite_rate(&rates[(size_t)p * S], S)) {
            PointGeom g;
            point_geometry(c, z ? z + p * d : nullptr, g);
            double v = 0.0;
            if ((rc = inf_rate_value(c, g, &rates[(size_t)p * S], dataset ? dataset[p] : 0, &v))) return rc;
            (v != v ? nanv : bad).push_back(p);
            continue;
        }
        pts.push_back({cell[(size_t)p] * c->T + (dataset ? dataset[p] : 0), p});
    }
    std::sort(pts.begin(), pts.end(), [](const Pt& a, const Pt& b) { return a.key < b.key || (a.key == b.key && a.idx < b.idx); });

    // ---- phase 3: chop groups into items, detect row reuse (serial, O(items * corners)) -------------
    const int maxg = bb ? (int)std::min<int64_t>(c->max_group, c->bb_max_group) : (int)c->max_group;  // (G = 16 with BB: 310 VGPRs + 54 AGPRs)
    std::vector<int64_t> corner_off((size_t)nc);
    for (int k = 0; k < nc; ++k) corner_off[(size_t)k] = corner_offset(c, k);
    std::vector<PlanItem> items;
    int64_t n_in_class[5] = {0, 0, 0, 0, 0};
    std::vector<char> anchor_used((size_t)c->A, 0);
    bool reuse = false;
    for (size_t i = 0; i < pts.size();) {
        size_t j = i;
        while (j < pts.size() && pts[j].key == pts[i].key) ++j;
        for (size_t n = j - i; n > 0;) {
            const int G = pick_class((int)std::min<size_t>(n, (size_t)maxg), maxg);
            const int take = (int)std::min<size_t>(n, (size_t)G);
            int ci = 0;
            while (kClassG[ci] != G) ++ci;
            items.push_back({ci, take, (int64_t)i, n_in_class[ci]++});
            const int64_t ca = cell[(size_t)pts[i].idx];
            for (int corner = 0; corner < nc; ++corner) {
                char& u = anchor_used[(size_t)(ca + corner_off[(size_t)corner])];
                if (u) reuse = true;
                u = 1;
            }
            i += (size_t)take;
            n -= (size_t)take;
        }
    }

    // ---- phase 4: fill the per-class descriptor arrays (parallel over items) --------------------------
    struct HostClass { std::vector<int64_t> rowoff, cnt_off, perm; std::vector<double> coef, aux, slot_lg; std::vector<int32_t> tiles; };
    HostClass hc[5];
    for (int ci = 0; ci < 5; ++ci) {
        const size_t n = (size_t)n_in_class[ci], G = (size_t)kClassG[ci];
        hc[ci].rowoff.resize(n * NS);
        hc[ci].coef.assign(n * NS * G, 0.0);
        hc[ci].aux.assign(n * G * 2, 1.0);
        hc[ci].cnt_off.resize(n);
        hc[ci].tiles.resize(n);
        hc[ci].perm.assign(n * G, -1);
        hc[ci].slot_lg.resize(n * G);
    }
    std::vector<double*> aux_of(bb ? (size_t)P : 0, nullptr);      // Beeston-Barlow: where every point's {p_cal, N} went
    parallel_for((int64_t)items.size(), 512, [&](int64_t lo, int64_t hi) {
        for (int64_t it = lo; it < hi; ++it) {
            const PlanItem& I = items[(size_t)it];
            HostClass& h = hc[I.cls];
            const int G = kClassG[I.cls];
            const int64_t p0 = pts[(size_t)I.first].idx;
            const int64_t ca = cell[(size_t)p0];
            const int64_t ds = pts[(size_t)I.first].key % c->T;
            const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
            const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
            int64_t* rowoff = &h.rowoff[(size_t)I.slot * NS];
            double* coef = &h.coef[(size_t)I.slot * NS * G];
            // stream rows: [n0] (corner, source != bb) ; [n1] (corner, bb source) ; [n2] n_model corner rows
            int k0 = 0;
            for (int corner = 0; corner < nc; ++corner)
                for (int s = 0; s < S; ++s) {
                    if (bb && s == c->bb_source) continue;
                    rowoff[k0++] = row_base + ((ca + corner_off[(size_t)corner]) * S + s) * row_stride;
                }
            for (int corner = 0; corner < n1; ++corner) rowoff[n0 + corner] = ((ca + corner_off[(size_t)corner]) * S + c->bb_source) * c->Bp;
            for (int corner = 0; corner < n2; ++corner) rowoff[n0 + n1 + corner] = (ca + corner_off[(size_t)corner]) * c->Bp;
            h.cnt_off[(size_t)I.slot] = sparse ? c->h_cnt_off[(size_t)ds] : ds * c->Bp;
            h.tiles[(size_t)I.slot] = (int32_t)(row_stride / kTile);
            for (int g = 0; g < G; ++g) h.slot_lg[(size_t)I.slot * G + g] = c->h_lgsum[(size_t)ds];
            for (int g = 0; g < I.take; ++g) {
                const int64_t p = pts[(size_t)I.first + g].idx;
                const double* w = &wts[(size_t)p * nc];
                const double* r = &rates[(size_t)p * S];
                int k = 0;
                double zsum = 0.0;
                for (int corner = 0; corner < nc; ++corner)
                    for (int s = 0; s < S; ++s) {
                        if (bb && s == c->bb_source) continue;
                        const double cf = w[corner] * r[s];
                        coef[(size_t)k * G + g] = cf;
                        // minus sum_k coef_k * (sum of row k over the empty bins of this dataset)
                        if (sparse) zsum += cf * c->h_Tz[(size_t)(ds * n_rows + (ca + corner_off[(size_t)corner]) * S + s)];
                        ++k;
                    }
                for (int corner = 0; corner < n1; ++corner) coef[(size_t)(n0 + corner) * G + g] = w[corner];
                for (int corner = 0; corner < n2; ++corner) coef[(size_t)(n0 + n1 + corner) * G + g] = w[corner];
                if (bb) {
                    double Ntot = 0.0;
                    for (int corner = 0; corner < nc; ++corner) {
                        const double term = c->h_nm_tot[(size_t)(ca + corner_off[(size_t)corner])] * w[corner];
                        Ntot = Ntot + term;
                    }
                    h.aux[((size_t)I.slot * G + g) * 2 + 0] = r[c->bb_source] / Ntot;  // p_calibration, likelihood.py:645
                    h.aux[((size_t)I.slot * G + g) * 2 + 1] = Ntot;
                    aux_of[(size_t)p] = &h.aux[((size_t)I.slot * G + g) * 2];
                }
                double& lg = h.slot_lg[(size_t)I.slot * G + g];
                if (c->unbinned) {   // ll = -sum_s mu_s + sum_e log(...)   (likelihood.py:690)
                    double rsum = 0.0;
                    for (int s = 0; s < S; ++s) rsum += r[s];
                    lg = rsum;
                }
                lg += zsum;
                h.perm[(size_t)I.slot * G + g] = p;
            }
        }
    });

    // Beeston-Barlow points at which some bin can have U_b == 0: N(z) in numpy's own summation order instead of the
    // per-anchor totals, so that the root formula sees the reference's bits there (bb_exact_totals)
    if (bb && c->bb_exact) {
        std::vector<int64_t> who, anchors;
        std::vector<double> ws;
        for (const Pt& q : pts) {
            const int64_t p = q.idx;
            if (!aux_of[(size_t)p]) continue;
            if (c->bb_exact == 1 || bb_zero_u_possible(c, cell[(size_t)p], &wts[(size_t)p * nc], &rates[(size_t)p * S])) {
                who.push_back(p);
                anchors.push_back(cell[(size_t)p]);
                ws.insert(ws.end(), wts.begin() + (size_t)p * nc, wts.begin() + (size_t)(p + 1) * nc);
            }
        }
        if (!who.empty()) {
            std::vector<double> Nx(who.size());
            if ((rc = bb_exact_totals(c, (int64_t)who.size(), anchors.data(), ws.data(), Nx.data()))) return rc;
            for (size_t i = 0; i < who.size(); ++i) {
                double* aux = aux_of[(size_t)who[i]];
                aux[0] = rates[(size_t)who[i] * S + c->bb_source] / Nx[i];
                aux[1] = Nx[i];
            }
        }
    }

    // ---- upload; grid shape: enough blocks to fill the chip, few enough that partial buffers stay small ----
    bi_plan* plan = new bi_plan();
    plan->P = P;
    plan->sparse = sparse;
    plan->epoch = c->epoch;
    plan->no_reuse = !reuse;
    plan->h_status = st;
    auto abort_plan = [&](int code) { free_plan_buffers(plan); delete plan; return code; };
    const int n_tiles = n_tiles_of(c);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int64_t total_items = (int64_t)items.size();
    size_t small_bytes = (size_t)P * 16 + bad.size() * 8 + nanv.size() * 8 + 64;
    for (int ci = 0; ci < 5; ++ci) {
        const HostClass& h = hc[ci];
        small_bytes += h.rowoff.size() * 8 + h.coef.size() * 8 + h.aux.size() * 8 + h.cnt_off.size() * 8 + h.tiles.size() * 4 +
                       h.perm.size() * 8 + h.slot_lg.size() * 8 + 7 * 64;
    }
    const bool packed = transient && small_bytes <= (size_t)256 * 1024;
    std::vector<std::pair<const void*, size_t>> parts;
    for (int ci = 0; ci < 5; ++ci) {
        HostClass& h = hc[ci];
        if (h.tiles.empty()) continue;
        plan->classes.emplace_back();
        bi_plan::Class& k = plan->classes.back();
        k.G = kClassG[ci];
        k.n_items = (int64_t)h.tiles.size();
        const int64_t max_tiles = sparse ? *std::max_element(h.tiles.begin(), h.tiles.end()) : n_tiles;
        int64_t nbx = std::min<int64_t>(max_tiles, std::max<int64_t>(1, (4 * slots + total_items - 1) / total_items));
        if (total_items == 1) nbx = std::min<int64_t>(max_tiles, slots);
        // Blocks are dealt round-robin over the 8 XCDs, so with nbx a multiple of 8 the tile chunk blockIdx.x of
        // EVERY item lands on XCD x % 8: items that share template rows (same or neighbouring cell) then find
        // them in that XCD's L2 instead of each XCD fetching every tile (speed only, never correctness).
        if (c->xcd_affine && total_items > 1 && nbx > 4 && nbx < max_tiles) nbx = std::min<int64_t>(max_tiles, (nbx + 7) / 8 * 8);
        k.nbx = (int)nbx;
        if (packed) {
            parts.push_back({h.rowoff.data(), h.rowoff.size() * sizeof(int64_t)});
            parts.push_back({h.coef.data(), h.coef.size() * sizeof(double)});
            parts.push_back({h.aux.data(), h.aux.size() * sizeof(double)});
            parts.push_back({h.cnt_off.data(), h.cnt_off.size() * sizeof(int64_t)});
            parts.push_back({h.tiles.data(), h.tiles.size() * sizeof(int32_t)});
            parts.push_back({h.perm.data(), h.perm.size() * sizeof(int64_t)});
            parts.push_back({h.slot_lg.data(), h.slot_lg.size() * sizeof(double)});
        } else if ((rc = dev_upload(c, k.rowoff, h.rowoff)) || (rc = dev_upload(c, k.coef, h.coef)) ||
                   (rc = dev_upload(c, k.aux, h.aux)) || (rc = dev_upload(c, k.item_cnt, h.cnt_off)) ||
                   (rc = dev_upload(c, k.item_tiles, h.tiles)) || (rc = dev_upload(c, k.perm, h.perm)) ||
                   (rc = dev_upload(c, k.slot_lg, h.slot_lg)))
            return abort_plan(rc);
        if ((rc = dev_alloc(c, k.partial, (size_t)k.n_items * k.nbx * k.G * sizeof(double))) ||
            (rc = dev_alloc(c, k.pflags, (size_t)k.n_items * k.nbx * k.G * sizeof(unsigned))))
            return abort_plan(rc);
        for (int32_t t : h.tiles) plan->bytes += (int64_t)sizeof(double) * ((int64_t)NS + 1) * (sparse ? (int64_t)t * kTile : c->B);
        plan->launches += (k.n_items + 65534) / 65535;
    }
    plan->n_bad = (int64_t)bad.size();
    plan->n_nan = (int64_t)nanv.size();
    if (packed) {
        // one copy for every descriptor array; out and status live in the pinned block behind them, where the finish
        // kernel writes them directly (the caller reads them there after its stream sync)
        parts.push_back({bad.data(), bad.size() * sizeof(int64_t)});
        parts.push_back({nanv.data(), nanv.size() * sizeof(int64_t)});
        const size_t out_bytes = ((size_t)std::max<int64_t>(P, 1) * sizeof(double) + 63) / 64 * 64;
        PackedUpload pu;
        if ((rc = packed_upload(c, parts, out_bytes + (size_t)std::max<int64_t>(P, 1) * sizeof(int32_t), pu, &plan->slab)))
            return abort_plan(rc);
        auto view = [&](DevBuf& b, size_t part) {
            b.p = pu.dev_base + pu.off[part];
            b.bytes = parts[part].second;
            b.view = true;
        };
        size_t part = 0;
        for (auto& k : plan->classes) {
            view(k.rowoff, part++); view(k.coef, part++); view(k.aux, part++); view(k.item_cnt, part++);
            view(k.item_tiles, part++); view(k.perm, part++); view(k.slot_lg, part++);
        }
        view(plan->bad_idx, part++);
        view(plan->nan_idx, part);
        plan->host_results = true;
        plan->out.p = pu.host_out();
        plan->out.bytes = out_bytes;
        plan->out.view = true;
        plan->status.p = (char*)pu.host_out() + out_bytes;
        plan->status.bytes = (size_t)std::max<int64_t>(P, 1) * sizeof(int32_t);
        plan->status.view = true;
        if (P) memcpy(plan->status.p, plan->h_status.data(), (size_t)P * sizeof(int32_t));
        *out = plan;
        return BI_OK;
    }
    if ((rc = dev_upload(c, plan->bad_idx, bad)) || (rc = dev_upload(c, plan->nan_idx, nanv)) || (rc = dev_alloc(c, plan->out, (size_t)std::max<int64_t>(P, 1) * sizeof(double))) ||
        (rc = dev_upload(c, plan->status, plan->h_status)))
        return abort_plan(rc);
    hipError_t e = hipStreamSynchronize(c->stream);  // the host staging vectors die with this scope
    if (e != hipSuccess) return abort_plan(fail(c, BI_ERR_HIP, "plan upload: %s", hipGetErrorString(e)));
    *out = plan;
    return BI_OK;
}

}  // namespace
