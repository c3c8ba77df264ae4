// bi_grad_bb.h -- host half of bi_eval_grad for Beeston-Barlow models: per point the coefficient COLUMNS of the value
// and of its derivatives (k_morph_bbgrad, bi_k_bbgrad.h), one work item per point, k_finish for the sums.
#pragma once

namespace {

int eval_grad_bb(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, double* ll,
                 double* grad, int32_t* status) {
    const int S = c->S, d = c->d, bbs = c->bb_source;
    const int W = 1 + d + S;
    const int de = (int)c->eff_axes.size();
    const int nc = 1 << de;
    const int n0 = nc * (S - 1), n1 = nc, n2 = nc, NS = n0 + n1 + n2;
    const int G = W <= 8 ? 8 : 16;
    const int DZ = 1 + d <= 4 ? 4 : 8;
    if (W > 16 || 1 + d > 8) return fail(c, BI_ERR_INVALID, "Beeston-Barlow gradient needs 1 + d + S <= 16 and d <= 7 (got d = %d, S = %d)", d, S);
    if (!c->dense_counts) return fail(c, BI_ERR_STATE, "dataset counts are not resident in dense form");
    const double ninf = -std::numeric_limits<double>::infinity();
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    const int64_t coef_per_item = (int64_t)n0 * G + (int64_t)(n1 + n2) * DZ;

    std::vector<int64_t> rowoff, cnt_off, perm, live;
    std::vector<double> coef, aux, slot_lg;
    std::vector<double> ones((size_t)S, 1.0), mus((size_t)S), dmus((size_t)S * std::max(d, 1)), dw((size_t)nc * std::max(de, 1));
    std::vector<int64_t> corner_off((size_t)nc);
    for (int k = 0; k < nc; ++k) corner_off[(size_t)k] = corner_offset(c, k);
    std::vector<PointGeom> geoms;
    std::vector<double> rates_all;
    for (int64_t p = 0; p < P; ++p) {
        if (status) status[p] = 0;
        ll[p] = ninf;
        for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = qnan;
        const int64_t ds = dataset ? dataset[p] : 0;
        if (ds < 0 || ds >= c->T) { if (status) status[p] = BI_ST_BAD_DATASET; continue; }
        PointGeom g;
        if (!point_geometry(c, z ? z + p * d : nullptr, g)) { if (status) status[p] = BI_ST_OUT_OF_BOUNDS; continue; }
        interp_mus(c, g, mus.data());
        const double* rs = rate_scale ? rate_scale + p * S : ones.data();
        std::vector<double> r((size_t)S);
        for (int s = 0; s < S; ++s) r[(size_t)s] = mus[(size_t)s] * rs[s];
        if (!rates_physical(c, r.data())) { if (status) status[p] = BI_ST_UNPHYSICAL; continue; }
        // d w_c / d z_i over the effective axes, d mus_s / d z_i, d N / d z_i
        for (int corner = 0; corner < nc; ++corner)
            for (int i = 0; i < de; ++i) {
                const int ax = c->eff_axes[(size_t)i];
                double v = (((corner >> (de - 1 - i)) & 1) ? 1.0 : -1.0) * g.inv_delta[ax];
                for (int j = 0; j < de; ++j) {
                    if (j == i) continue;
                    const double t = g.t[c->eff_axes[(size_t)j]];
                    v *= ((corner >> (de - 1 - j)) & 1) ? t : (1 - t);
                }
                dw[(size_t)corner * de + i] = v;
            }
        for (int i = 0; i < de; ++i)
            for (int s = 0; s < S; ++s) {
                double v = 0.0;
                for (int corner = 0; corner < nc; ++corner)
                    v += dw[(size_t)corner * de + i] * c->h_mus[(size_t)((g.cell_anchor + corner_off[(size_t)corner]) * S + s)];
                dmus[(size_t)i * S + s] = v;
            }
        const size_t ro = rowoff.size(), co = coef.size(), po = perm.size();
        rowoff.resize(ro + NS);
        coef.resize(co + (size_t)coef_per_item, 0.0);
        aux.resize(po * 2 + (size_t)G * 2, 0.0);
        perm.resize(po + G, -1);
        slot_lg.resize(po + G, 0.0);
        double* cU = &coef[co];
        double* cP = cU + (size_t)n0 * G;
        double* cA = cP + (size_t)n1 * DZ;
        int k = 0;
        double Ntot = 0.0;
        for (int corner = 0; corner < nc; ++corner) {
            const int64_t a = g.cell_anchor + corner_off[(size_t)corner];
            const double w = g.w[(size_t)corner];
            for (int s = 0; s < S; ++s) {
                if (s == bbs) continue;
                rowoff[ro + k] = (a * S + s) * c->Bp;
                double* col = cU + (size_t)k * G;
                col[0] = w * r[(size_t)s];
                for (int i = 0; i < de; ++i)
                    col[1 + c->eff_axes[(size_t)i]] = dw[(size_t)corner * de + i] * r[(size_t)s] + w * dmus[(size_t)i * S + s] * rs[s];
                col[1 + d + s] = w * mus[(size_t)s];
                ++k;
            }
            rowoff[ro + n0 + corner] = (a * S + bbs) * c->Bp;
            rowoff[ro + n0 + n1 + corner] = a * c->Bp;
            cP[(size_t)corner * DZ] = w;
            cA[(size_t)corner * DZ] = w;
            for (int i = 0; i < de; ++i) {
                cP[(size_t)corner * DZ + 1 + c->eff_axes[(size_t)i]] = dw[(size_t)corner * de + i];
                cA[(size_t)corner * DZ + 1 + c->eff_axes[(size_t)i]] = dw[(size_t)corner * de + i];
            }
            const double term = c->h_nm_tot[(size_t)a] * w;
            Ntot = Ntot + term;
        }
        double* ax_ = &aux[po * 2];
        ax_[0] = 0.0; ax_[1] = Ntot;                      // p_cal is set below, once N is final (bb_exact)
        for (int i = 0; i < de; ++i) {
            const int axq = 1 + c->eff_axes[(size_t)i];
            double dN = 0.0;
            for (int corner = 0; corner < nc; ++corner)
                dN += dw[(size_t)corner * de + i] * c->h_nm_tot[(size_t)(g.cell_anchor + corner_off[(size_t)corner])];
            ax_[axq * 2 + 0] = dmus[(size_t)i * S + bbs] * rs[bbs];
            ax_[axq * 2 + 1] = dN;
        }
        ax_[(1 + d + bbs) * 2 + 0] = mus[(size_t)bbs];
        slot_lg[po] = c->h_lgsum[(size_t)ds];
        for (int q = 0; q < W; ++q) perm[po + q] = (int64_t)live.size() * W + q;
        cnt_off.push_back(ds * c->Bp);
        live.push_back(p);
        geoms.push_back(g);
        rates_all.insert(rates_all.end(), r.begin(), r.end());
    }
    const int64_t n_items = (int64_t)live.size();
    if (n_items == 0) return BI_OK;
    int rc;
    // N(z) in numpy's summation order where some bin can have U_b == 0 (as the value path does: same bits, same asserts)
    for (int64_t i = 0; i < n_items; ++i) {
        double& N = aux[(size_t)i * G * 2 + 1];
        const double* r = &rates_all[(size_t)i * S];
        if (c->bb_exact == 1 || (c->bb_exact == 2 && bb_zero_u_possible(c, geoms[(size_t)i], r))) {
            if ((rc = bb_exact_total(c, geoms[(size_t)i], &N))) return rc;
        }
        aux[(size_t)i * G * 2 + 0] = r[bbs] / N;
    }
    const int n_tiles = n_tiles_of(c);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nbx = (int)std::min<int64_t>(n_tiles, n_items == 1 ? slots : std::max<int64_t>(1, (4 * slots + n_items - 1) / n_items));
    DevBuf d_part, d_flag;
    auto cleanup = [&]() { dev_free(d_part); dev_free(d_flag); };
    PackedUpload pu;
    const size_t out_bytes = (size_t)n_items * W * sizeof(double) + (size_t)n_items * W * sizeof(int32_t) + 64;
    if ((rc = packed_upload(c, {{rowoff.data(), rowoff.size() * sizeof(int64_t)}, {coef.data(), coef.size() * sizeof(double)},
                                {aux.data(), aux.size() * sizeof(double)}, {cnt_off.data(), cnt_off.size() * sizeof(int64_t)},
                                {perm.data(), perm.size() * sizeof(int64_t)}, {slot_lg.data(), slot_lg.size() * sizeof(double)}},
                            out_bytes, pu)) ||
        (rc = dev_alloc(c, d_part, (size_t)n_items * nbx * G * sizeof(double))) ||
        (rc = dev_alloc(c, d_flag, (size_t)n_items * nbx * G * sizeof(unsigned)))) {
        cleanup();
        return rc;
    }
    double* h_out = (double*)pu.host_out();
    int32_t* h_st = (int32_t*)((char*)pu.host_out() + ((size_t)n_items * W * sizeof(double) + 63) / 64 * 64);
    memset(h_st, 0, (size_t)n_items * W * sizeof(int32_t));
    LaunchArgs a{};
    a.ps = (const double*)c->ps.p;
    a.nm = (const double*)c->nm.p;
    a.counts = (const double*)c->counts.p;
    a.B = c->B; a.Bp = c->Bp; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.n_tiles = n_tiles; a.chunks = (int)c->tile_chunks;
    const bool huge = (int64_t)sizeof(double) * (NS + 1) * c->Bp > ((int64_t)1 << 30);
    const bool nt = c->nt_loads == 1 || (c->nt_loads == 2 && (n_items == 1 || huge));
    for (int64_t i0 = 0; i0 < n_items; i0 += 65535) {
        const int64_t ni = std::min<int64_t>(65535, n_items - i0);
        LaunchArgs b = a;
        b.rowoff = pu.dev<int64_t>(0) + i0 * NS;
        b.coef = pu.dev<double>(1) + i0 * coef_per_item;
        b.aux = pu.dev<double>(2) + i0 * G * 2;
        b.item_cnt = pu.dev<int64_t>(3) + i0;
        b.partial = (double*)d_part.p + i0 * nbx * G;
        b.pflags = (unsigned*)d_flag.p + i0 * nbx * G;
        if ((rc = launch_morph_bbgrad(c, G, DZ, b, dim3((unsigned)nbx, (unsigned)ni), nt))) { cleanup(); return fail(c, rc, "no Beeston-Barlow gradient kernel for %d x %d columns", G, DZ); }
        const int64_t n_slots = ni * G;
        const int lanes = nbx > 64 ? kThreads : 64;
        const int per_block = kThreads / lanes;
        hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0, c->stream,
                           (const double*)b.partial, (const unsigned*)b.pflags, nbx, G, lanes, n_slots,
                           pu.dev<int64_t>(4) + i0 * G, pu.dev<double>(5) + i0 * G, h_out, h_st);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_grad (Beeston-Barlow): %s", hipGetErrorString(e));
    for (int64_t i = 0; i < n_items; ++i) {
        const int64_t p = live[(size_t)i];
        ll[p] = h_out[(size_t)i * W];
        for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = h_out[(size_t)i * W + 1 + j];
        if (status) status[p] |= h_st[(size_t)i * W];          // the Beeston-Barlow assertion bits ride on the value slot
    }
    return BI_OK;
}

}  // namespace
