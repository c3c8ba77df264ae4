// bi_params.h -- the ONE table behind bi_set_param / bi_get_param / bi_list_params: tunables (read-write), counters
// (read-only) and triggers (write-only) of a context, each with its range check.  A name that is not in the table is an
// error in both directions (bi_set_param: BI_ERR_INVALID; bi_get_param: INT64_MIN, which no parameter can hold).
#pragma once

namespace {

enum : int { kParamRead = 1, kParamWrite = 2, kParamRW = 3 };

struct ParamDef {
    const char* name;
    int access;
    int64_t (*get)(bi_ctx*);
    int (*set)(bi_ctx*, int64_t);
};

#define BI_P_GET(expr) [](bi_ctx* c) -> int64_t { return (int64_t)(expr); }
#define BI_P_SET(stmt) [](bi_ctx* c, int64_t v) -> int { stmt; return BI_OK; }
#define BI_P_RANGE(lo, hi, field, what)                                                                           \
    [](bi_ctx* c, int64_t v) -> int {                                                                             \
        if (v < (lo) || v > (hi)) return fail(c, BI_ERR_INVALID, what);                                           \
        c->field = v;                                                                                             \
        return BI_OK;                                                                                             \
    }
#define BI_P_FLAG(field) BI_P_SET(c->field = v ? 1 : 0)
#define BI_P_RO(nm, expr) {nm, kParamRead, BI_P_GET(expr), nullptr}

const ParamDef kParams[] = {
    // ---- tunables --------------------------------------------------------------------------------------------
    {"sparse", kParamRW, BI_P_GET(c->sparse), BI_P_RANGE(0, 2, sparse, "sparse: 0 = off, 1 = auto, 2 = whenever exact")},
    {"max_group", kParamRW, BI_P_GET(c->max_group),
     [](bi_ctx* c, int64_t v) -> int {
         if (v < 1 || v > kMaxG || (v & (v - 1))) return fail(c, BI_ERR_INVALID, "max_group must be a power of two in [1,%d]", kMaxG);
         c->max_group = v;
         return BI_OK;
     }},
    {"bb_max_group", kParamRW, BI_P_GET(c->bb_max_group), BI_P_SET(c->bb_max_group = v >= 16 ? 16 : (v >= 8 ? 8 : (v >= 4 ? 4 : (v >= 2 ? 2 : 1))))},
    {"blocks_per_cu", kParamRW, BI_P_GET(c->blocks_per_cu), BI_P_RANGE(1, 32, blocks_per_cu, "blocks_per_cu in [1,32]")},
    {"nt_loads", kParamRW, BI_P_GET(c->nt_loads), BI_P_RANGE(0, 2, nt_loads, "nt_loads: 0 = never, 1 = always, 2 = auto")},
    {"tile_chunks", kParamRW, BI_P_GET(c->tile_chunks), BI_P_SET(c->tile_chunks = v < 1 ? 1 : v)},
    {"single_kernel", kParamRW, BI_P_GET(c->single_kernel), BI_P_FLAG(single_kernel)},
    {"fuse_max_blocks", kParamRW, BI_P_GET(c->fuse_max_blocks), BI_P_SET(c->fuse_max_blocks = v)},
    {"single_blocks_per_cu", kParamRW, BI_P_GET(c->single_blocks_per_cu),
     BI_P_RANGE(1, 32, single_blocks_per_cu, "single_blocks_per_cu in [1,32]")},
    {"fuse_finish", kParamRW, BI_P_GET(c->fuse_finish), BI_P_FLAG(fuse_finish)},
    {"keep_rows", kParamRW, BI_P_GET(c->keep_rows), BI_P_SET(c->keep_rows = v < 0 ? -1 : v)},
    {"poll_result", kParamRW, BI_P_GET(c->poll_result), BI_P_FLAG(poll_result)},
    {"xcd_affine", kParamRW, BI_P_GET(c->xcd_affine), BI_P_FLAG(xcd_affine)},
    {"device_plan_min", kParamRW, BI_P_GET(c->device_plan_min), BI_P_SET(c->device_plan_min = v)},
    {"scan_mfma", kParamRW, BI_P_GET(c->scan_mfma), BI_P_FLAG(scan_mfma)},
    {"scan_min_items", kParamRW, BI_P_GET(c->scan_min_items), BI_P_SET(c->scan_min_items = v < 1 ? 1 : v)},
    {"scan_cb", kParamRW, BI_P_GET(c->scan_cb), BI_P_SET(c->scan_cb = (v == 2 || v == 4) ? v : 0)},
    {"scan_waves_per_cu", kParamRW, BI_P_GET(c->scan_waves_per_cu), BI_P_SET(c->scan_waves_per_cu = v < 0 ? 0 : v)},
    {"host_threads", kParamRW, [](bi_ctx*) -> int64_t { return host_threads(); },
     [](bi_ctx* c, int64_t v) -> int { if (v < 0 || v > 256) return fail(c, BI_ERR_INVALID, "host_threads in [0, 256] (0 = by the process's affinity, at most 16)"); host_threads_setting().store((int)v); return BI_OK; }},
    {"toy_points_pp", kParamRW, BI_P_GET(c->toy_points_pp),
     [](bi_ctx* c, int64_t v) -> int {
         if (v != 0 && v != 1 && v != 2 && v != 4) return fail(c, BI_ERR_INVALID, "toy_points_pp: 0 = by the batch, 1 = point by point, 2 or 4 points per pass");
         c->toy_points_pp = v;
         return BI_OK;
     }},
    {"toy_points_overlap", kParamRW, BI_P_GET(c->toy_points_overlap), BI_P_FLAG(toy_points_overlap)},
    {"toy_points_lanes", kParamRW, BI_P_GET(c->toy_points_lanes),
     [](bi_ctx* c, int64_t v) -> int {
         if (v != 0 && v != 2 && v != 4 && v != 8) return fail(c, BI_ERR_INVALID, "toy_points_lanes: 0 = default, or 2 / 4 / 8");
         c->toy_points_lanes = v;
         return BI_OK;
     }},
    {"grad_mfma", kParamRW, BI_P_GET(c->grad_mfma), BI_P_FLAG(grad_mfma)},
    {"grad_mfma_min", kParamRW, BI_P_GET(c->grad_mfma_min), BI_P_RANGE(1, (int64_t)1 << 40, grad_mfma_min, "grad_mfma_min >= 1")},
    {"grad_slices", kParamRW, BI_P_GET(c->grad_slices), BI_P_RANGE(0, 4096, grad_slices, "grad_slices in [0, 4096]")},
    {"scan_xcd", kParamRW, BI_P_GET(c->scan_xcd), BI_P_RANGE(0, 2, scan_xcd, "scan_xcd: 0 launch order, 1 contiguous ranges, 2 one XCD per group")},
    {"scan_share_slow", kParamRW, BI_P_GET(c->scan_share_slow), BI_P_FLAG(scan_share_slow)},
    {"scan_chunk", kParamRW, BI_P_GET(c->scan_chunk), BI_P_FLAG(scan_chunk)},
    {"plan_tables", kParamRW, BI_P_GET(c->plan_tables), BI_P_FLAG(plan_tables)},
    {"plan_count_sort", kParamRW, BI_P_GET(c->plan_count_sort), BI_P_FLAG(plan_count_sort)},
    {"scan_bb", kParamRW, BI_P_GET(c->scan_bb), BI_P_FLAG(scan_bb)},
    {"scan_bb_min", kParamRW, BI_P_GET(c->scan_bb_min), BI_P_RANGE(1, (int64_t)1 << 40, scan_bb_min, "scan_bb_min >= 1")},
    {"scan_sparse_max_items", kParamRW, BI_P_GET(c->scan_sparse_max_items), BI_P_SET(c->scan_sparse_max_items = v < 0 ? 0 : v)},
    {"scan_split", kParamRW, BI_P_GET(c->scan_split), BI_P_FLAG(scan_split)},
    {"scan_pow", kParamRW, BI_P_GET(c->scan_pow), BI_P_SET(c->scan_pow = v ? 1 : 0; c->sorted_epoch = -1)},
    {"bb_exact", kParamRW, BI_P_GET(c->bb_exact), BI_P_RANGE(0, 2, bb_exact, "bb_exact: 0 never, 1 always, 2 auto")},
    {"toy_events", kParamRW, BI_P_GET(c->toy_events), BI_P_FLAG(toy_events)},
    {"dot_tiled", kParamRW, BI_P_GET(c->dot_tiled), BI_P_FLAG(dot_tiled)},
    {"score_sorted", kParamRW, BI_P_GET(c->score_sorted), BI_P_FLAG(score_sorted)},
    {"toy_fast_call", kParamRW, BI_P_GET(c->toy_fast_call), BI_P_RANGE(0, 7, toy_fast_call, "toy_fast_call: bits 1 | 2 | 4")},
    {"dot_entry16", kParamRW, BI_P_GET(c->dot_entry16), [](bi_ctx* c, int64_t v) -> int { c->dot_entry16 = v ? 1 : 0; c->nz_tile_epoch = -1; c->tmm_epoch = -1; return BI_OK; }},
    {"dot_blocks_per_cu", kParamRW, BI_P_GET(c->dot_blocks_per_cu), BI_P_RANGE(0, 16, dot_blocks_per_cu, "dot_blocks_per_cu in [0, 16]")},
    {"dot_lanes", kParamRW, BI_P_GET(c->dot_lanes), BI_P_SET(c->dot_lanes = (v == 16 || v == 8 || v == 4) ? v : 0)},
    {"compact_budget", kParamRW, BI_P_GET(c->compact_budget), BI_P_SET(c->compact_budget = v)},
    {"toy_offset", kParamRW, BI_P_GET(c->toy_offset), BI_P_RANGE(0, INT64_MAX, toy_offset, "toy_offset >= 0")},
    {"mail_timeout_ms", kParamRW, BI_P_GET(c->mail_timeout_ms),
     BI_P_RANGE(1, 60000, mail_timeout_ms, "mail_timeout_ms in [1, 60000]")},
    // ---- triggers (write-only) ---------------------------------------------------------------------------------
    {"single_timing_reset", kParamWrite, nullptr,
     BI_P_SET(c->single_ns[0] = c->single_ns[1] = c->single_ns[2] = 0; c->single_calls = 0; (void)v)},
    {"drop_recycle_cache", kParamWrite, nullptr, BI_P_SET((void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); drop_recycle_cache(c); (void)v)},
    // fault injection for the in-launch finish, consumed by the NEXT launch that finishes through the mailbox: block
    // `v` of every work item never posts its partial sum (debug_skip_post), or posts it only after the collector's
    // wait has run out (debug_late_post); -1 = off
    {"debug_skip_post", kParamWrite, nullptr, BI_P_SET(c->debug_skip_post = v < 0 ? -1 : v)},
    {"debug_late_post", kParamWrite, nullptr, BI_P_SET(c->debug_late_post = v < 0 ? -1 : v)},
    // ---- counters and state (read-only) ------------------------------------------------------------------------
    BI_P_RO("tile_bins", kTile),
    BI_P_RO("padded_bins", c->Bp),
    BI_P_RO("n_scan_launches", c->n_scan_launches),
    BI_P_RO("n_toy_polled", c->n_toy_polled),
    BI_P_RO("n_toy_points_passes", c->n_toy_points_passes),
    BI_P_RO("n_bb_scan_launches", c->n_bb_scan_launches),
    BI_P_RO("tmm_entry_bytes", c->tmm_ok ? c->tmm_width : 0),
    BI_P_RO("tm_entry_bytes", c->tm_width),
    BI_P_RO("events_sorted", c->ev_sorted ? 1 : 0),
    BI_P_RO("n_grad_mfma_launches", c->n_grad_mfma_launches),
    BI_P_RO("n_valid_launches", c->n_valid_launches),
    BI_P_RO("n_sorted_scans", c->n_sorted_scans),
    BI_P_RO("n_bb_exact", c->n_bb_exact),
    BI_P_RO("n_mail_resets", c->n_mail_resets),
    // "ready" = prepared AND in use as an evaluation path (with sparse = 0 at upload they serve split scans only)
    BI_P_RO("csr_ready", (c->csr_ready && (c->sparse_at_upload != 0 || !c->dense_counts)) ? 1 : 0),
    BI_P_RO("compact_ready", (c->compact_ready && c->ps_nonneg && (c->sparse_at_upload != 0 || !c->dense_counts)) ? 1 : 0),
    BI_P_RO("split_ready", (c->compact_ready && c->dense_counts) ? 1 : 0),
    BI_P_RO("ps_nonneg", c->ps_nonneg ? 1 : 0),
    BI_P_RO("compact_sorted", (c->compact_ready && c->compact_sorted) ? 1 : 0),
    BI_P_RO("nnz_total", c->csr_ready ? c->h_nz_off.back() : -1),
    BI_P_RO("last_scan_nslots", c->last_scan_nslots),
    BI_P_RO("last_valid_nslots", c->last_valid_nslots),
    BI_P_RO("last_scan_resident", c->last_scan_resident),
    BI_P_RO("last_toy_method", c->last_toy_method),
    BI_P_RO("single_calls", c->single_calls),
    BI_P_RO("single_ns_host", c->single_ns[0]),
    BI_P_RO("single_ns_launch", c->single_ns[1]),
    BI_P_RO("single_ns_wait", c->single_ns[2]),
    BI_P_RO("user_allocations", (int64_t)c->user_allocs.size()),
    BI_P_RO("recycle_cache_bytes", (int64_t)c->cache_bytes),
};

#undef BI_P_GET
#undef BI_P_SET
#undef BI_P_RANGE
#undef BI_P_FLAG
#undef BI_P_RO

const ParamDef* find_param(const char* name) {
    for (const ParamDef& p : kParams)
        if (!strcmp(p.name, name)) return &p;
    return nullptr;
}

}  // namespace
