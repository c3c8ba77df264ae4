// bi_context.h -- the context / plan objects behind the opaque C handles, error and device-memory helpers.
#pragma once

struct bi_ctx;

constexpr int kThreads = 256;           // 4 wave64 per block
constexpr int kBinsPerThread = 2;       // one 16-byte load per stream per lane
constexpr int kTile = kThreads * kBinsPerThread;  // 512 bins = 4 KiB per stream per block tile
constexpr int kMaxDim = 8;              // shape parameters
constexpr int kMaxG = 16;               // points per cell pass

// (the library is several translation units -- one per heavy kernel family, compiled in parallel: types that cross them
//  sit at global scope, state that must exist once is `inline`)
inline thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    struct bi_ctx* owner = nullptr;    // context whose recycle cache takes the buffer back on dev_free
    bool view = false;                 // a window into somebody else's allocation: dev_free only forgets it
};

struct bi_plan {
    int64_t P = 0;
    struct Class {
        int G = 0;
        int64_t n_items = 0;
        int nbx = 0;
        DevBuf rowoff, coef, aux, item_cnt, item_tiles, perm, slot_lg, partial, pflags;
        DevBuf rowoff_full;        // split scans: offsets of the FULL rows of every item (validity pass)
    };
    std::vector<Class> classes;
    DevBuf bad_idx;            // points answered on the host side with -inf
    int64_t n_bad = 0;
    DevBuf nan_idx;            // ... and with nan (infinite rates whose expectation is nan somewhere, or meets data)
    int64_t n_nan = 0;
    DevBuf out, status;        // internal result buffers [P]
    std::vector<int32_t> h_status;
    int64_t epoch = 0;         // ctx->epoch at creation: a plan dies with the model / data it was made for
    DevBuf slab;                  // transient small plans: every descriptor array is a view into this one buffer
    bool host_results = false;    // ... and out / status are views into the context's pinned block
    bool use_scan = false;        // evaluated by the matrix-core scan kernel (groups of items per cell)
    int64_t n_groups = 0;
    int scan_cb = 4;              // its strip width in 16-bin blocks
    DevBuf grp_first, grp_items;
    std::vector<int64_t> h_grp_first;   // host copies of cut group tables (device planner: their upload is asynchronous)
    std::vector<int32_t> h_grp_items;
    int bb_kgt = 0;               // Beeston-Barlow batch on the matrix cores (k_scan_bb<bb_kgt>): items of 16 points, group tables
    bool valid = false;           // split dense scan: classes hold the non-empty-bin pass, k_scan_valid checks every bin
    int valid_nslots = 0;         // its waves per group
    DevBuf bad;                   // [items][16] flags it raises
    bool shared = false;          // a share of a scan dealt over several contexts (plan_points_device, share_world > 1): results
    int share_world = 1;          // come out in sorted order, out[0 .. share_hi - share_lo); sorted_idx [P] maps sorted positions
    int64_t share_lo = 0, share_hi = 0, n_valid = 0;   // back to the caller's point indices
    DevBuf sorted_idx;
    DevBuf keep_z, keep_rs;       // gradient batches on the matrix cores: the points' z / rate_scale stay on the device for the finish kernel
    int64_t max_group_items = 0;  // ... the largest number of work items a (cell, dataset) group holds
    int max_item_tiles = 0;       // ... and the tiles of the longest rows
    bool device_planned = false;  // built by plan_points_device: rejected points are found through the status array
    bool no_reuse = false;     // no anchor model is touched by two items of the plan
    bool sparse = false;       // rows / counts refer to the compacted (non-empty-bin) copies
    bool sorted = false;       // rows / counts refer to the count-sorted copy of all bins (dense-data scans, ensure_sorted_rows)
    bool by_count = false;     // the rows the scan kernel reads are ordered by count (sorted, or a count-sorted compacted copy)
    int64_t bytes = 0;         // algorithmic HBM bytes per run
    int64_t launches = 0;
};

struct bi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    std::string err;

    // model
    bool model_ready = false, model_open = false;
    int d = 0, S = 0;
    int64_t B = 0, Bp = 0, A = 0;
    std::vector<int> n_anchor;
    std::vector<std::vector<double>> grid;
    std::vector<int64_t> astride;  // anchor-index stride per axis
    std::vector<int> eff_axes;     // axes with >= 2 anchors
    int bb_source = -1;
    std::vector<int32_t> allow_neg;
    DevBuf ps, nm, nm_tot;
    std::vector<double> h_mus;     // [A][S]
    std::vector<double> h_nm_tot;  // [A]
    std::vector<char> anchor_set;

    // analysis space (bin edges) for device-side binning of events
    int space_k = 0;
    std::vector<int32_t> space_n_edges;
    DevBuf space_edges;

    // data
    bool unbinned = false;      // extended unbinned likelihood: rows are pdf values at the events
    double outlier = 0.0;
    bool ps_finite = true;
    bool data_ready = false;
    bool dense_counts = false;  // counts [T][Bp] resident (false for device-generated toys: CSR lists only)
    int64_t T = 0;
    DevBuf counts, lgsum;
    std::vector<double> h_lgsum;

    // model statistics (for the sparse forms)
    std::vector<double> h_rowsum;  // [A*S] sum over bins of every ps row
    std::vector<double> h_rowmin;  // [A*S] smallest entry of every ps row
    bool ps_nonneg = false;        // every ps entry is finite and >= 0

    // sparse forms of the data: CSR lists of the non-empty bins, and per-dataset compacted templates
    bool csr_ready = false, compact_ready = false;
    bool compact_sorted = false;               // the compacted copy holds the non-empty bins ordered by their count (build_compact_templates)
    DevBuf nz_idx, nz_n, nz_off, ps_c, cnt_c;
    DevBuf tm_entries, tm_off;                // tile-major copy of the non-empty-bin lists (k_dataset_dot_tiled): 4-byte entries, [n_tiles * T + 1] offsets
    int64_t nz_tile_epoch = -1;               // data epoch the copy was built for
    bool tm_ok = false;                       // ... and whether every count fits its 19 bits
    // the same lists over tiles of kDotTileMulti bins, for the toy-MC call over several parameter points (bi_eval_datasets_points:
    // the log mu tiles of four points side by side in LDS)
    DevBuf tmm_entries, tmm_off;
    int64_t tmm_epoch = -1;
    bool tmm_ok = false;
    int tmm_width = 4;
    int64_t toy_points_pp = 0;                // parameter: points per pass of that call (0 = by the batch: 4, or 2 for two points; 1 = point by point)
    int64_t toy_points_lanes = 0;             // parameter: lanes per (dataset, tile) run of its kernel (0 = measured default)
    int64_t toy_points_overlap = 0;           // parameter: the log mu pass and the finish of neighbouring pass groups run on a second stream beside the dot kernel (measured: no gain)
    hipStream_t stream2 = nullptr;            // ... that stream and its events (created on first use)
    hipEvent_t tp_ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t n_toy_points_passes = 0;          // read-only: passes over the entry lists the multi-point kernel has made
    int64_t dot_tiled = 1;                    // parameter: 0 = always the row kernel
    int64_t score_sorted = 1;                 // parameter: bi_score_events / bi_simulate_events order the events by cell before the gathers
    DevBuf ev_perm;                           // [events] sorted position -> the caller's event (valid while ev_sorted)
    bool ev_sorted = false;                   // the unbinned tensor's columns are in sorted order
    int64_t toy_fast_call = 7;                // parameter, bits: 1 descriptors in the kernel arguments, 2 parallel finish (k_dataset_finish_tiled), 4 poll the completion word
    DevBuf toy_blocks_done;                   // the finish kernel's block counter (zero between calls)
    bool toy_blocks_done_zeroed = false;
    unsigned long long toy_seq = 0;
    int64_t n_toy_polled = 0;                 // calls that returned on the completion word
    int64_t dot_entry16 = 1;                  // parameter: two-byte entries in the tile-major lists where every count is <= 7
    int tm_width = 4;                         // bytes per entry of the lists as built (2 or 4)
    int64_t dot_blocks_per_cu = 0;            // parameter: blocks of the tiled kernel per CU the dataset split aims at (0 = as many as are resident)
    int64_t dot_lanes = 0;                    // parameter: lanes per (dataset, tile) run of the tiled kernel: 0 = by entry width (8 with four-byte entries, 4 with two-byte ones), or 4 / 8 / 16
    int64_t toy_events = 1;                   // parameter: toys of sparse expectations are drawn event by event (0 = always one draw per bin)
    int64_t last_toy_method = 0;              // read-only: 1 = the last bi_generate_toys drew event by event, 0 = bin by bin
    std::vector<int64_t> h_nz_off;            // [T+1]
    std::vector<int64_t> h_c_off, h_cnt_off;  // [T] element offsets into ps_c / cnt_c
    std::vector<int64_t> h_c_np;              // [T] padded non-empty bins per dataset
    std::vector<double> h_Tz;                 // [T][A*S] sum of every ps row over the EMPTY bins of the dataset

    // persistent single-point slot (the lf(**kw) call shape): no allocation, one H2D, one D2H per call
    DevBuf slot_dev, slot_partial, slot_pflags, slot_counter;
    DevBuf mail, mail_flags;    // mailbox slots / status words of in-launch finishing: empty / zero between launches
    void* pack_host = nullptr;  // pinned staging of packed_upload (small-batch descriptors in, results out)
    void* bounce_host = nullptr;  // pinned bounce buffer of bi_memcpy_to_host / _to_device for copies of up to kBounceBytes
    size_t pack_host_bytes = 0;
    DevBuf pack_dev;
    void* plan_host = nullptr;  // pinned: what the device planner reports back (counters, group tables) + the word the host polls for it
    unsigned long long plan_seq = 0;
    void* slot_host = nullptr;  // pinned staging: descriptors in, {ll, status} out
    size_t slot_host_bytes = 0;
    unsigned long long slot_seq = 0;  // sequence number of single-point calls (the kernel echoes it when done)
    int64_t single_ns[3] = {0, 0, 0}; // accumulated wall time of single-point calls: host half, launch calls, wait
    int64_t single_calls = 0;
    int pending = 0;                  // bi_eval_begin without its bi_eval_end: 1 = launch in flight, 2 = answer parked
    unsigned long long pending_seq = 0;
    double pending_ll = 0.0;
    int32_t pending_status = 0;

    // device mirrors of the small tables the planning kernels read (bi_planning_device.h)
    DevBuf pt_grid, pt_mus, pt_coff, pt_allow, pt_c_off, pt_cnt_off, pt_c_np, pt_Tz, pt_rowsum, pt_rowmin, pt_nm_tot;
    int64_t plan_tables_epoch = -1;
    bool plan_tables_sparse = false;
    int64_t n_valid_launches = 0;                // how often the validity pass of a split scan ran
    int64_t n_scan_launches = 0;                 // how often the matrix-core scan kernel ran (observability)
    int64_t last_scan_nslots = 0, last_scan_resident = 0, last_valid_nslots = 0;   // what the planner chose last (read-only parameters)
    int64_t grad_mfma = 1;                       // bi_eval_grad: large single-dataset batches of plain binned likelihoods on the matrix cores (k_grad_mfma)
    int64_t grad_mfma_min = 2048;                // ... from this many points on (below, the planning of the batch costs more than the kernel saves)
    int64_t grad_slices = 0;                     // ... slices a cell's 16-bin blocks are split into (0 = by the batch)
    int64_t n_grad_mfma_launches = 0;
    int64_t scan_xcd = 1;                        // k_scan_sorted: how the (group, block) pairs are dealt to the 8 XCDs (0 launch order, 1 contiguous ranges, 2 group g -> XCD g mod 8)
    int64_t scan_share_slow = 1;                 // parameter: k_scan_sorted deals the items of mixed-count strips over all waves of the cell
    int64_t plan_count_sort = 1;                 // parameter: device planner: the one-pass counting sort where the keys take at most 1024 values (0: radix sort)
    int64_t plan_tables = 1;                     // parameter: device planner: group structure from per-key tables where the keys are few (0: scans over the points)
    int64_t scan_chunk = 1;                      // parameter: few groups with long item lists are cut into chunks of items, each a group of its own
    int64_t scan_waves_per_cu = 0;               // scan kernels: 0 = the planner sizes the split by occupancy; > 0 forces that many waves per CU
    int64_t keep_rows = -1;                      // single dense evaluations in a repeated cell: rows that keep the default cache policy (-1: as many as fit the Infinity Cache, 0: none)
    int64_t last_single_cell = -1, last_single_ds = -1;
    int64_t poll_result = 1;                     // single evaluations: poll the pinned result word instead of a stream sync
    int64_t tile_chunks = 8;                     // blocks walk the tiles in this many far-apart regions: block b (XCD b % 8) streams region b % 8
    int64_t scan_min_items = 4;                  // ... at least this many 16-point items per cell on average (x2: dense data)
    int64_t toy_offset = 0;                      // bi_generate_toys: toy t of the call is dataset toy_offset + t of the seed's stream
    int64_t scan_cb = 0;                         // scan kernel strip width in 16-bin blocks: 2, 4, or 0 = by the data
    int64_t bb_exact = 2;                        // single-point Beeston-Barlow calls: N(z) in numpy's summation order: 0 never, 1 always, 2 when some bin can have U_b == 0
    int64_t n_bb_exact = 0;                      // how often that pass ran
    int64_t scan_sparse_max_items = (int64_t)1 << 40;   // compacted rows: items per cell up to which the matrix-core scan kernel is used (round 1: 384;
                                                 // since the kernel leaves the linear term to the host it wins at every size: 46 vs 40 M evaluations/s at 10^6 points)
    int64_t scan_split = 1;                      // dense scans over mostly empty data: non-empty-bin pass + matrix-core validity pass
    int64_t sparse_at_upload = 1;                // value of `sparse` when the resident data were uploaded
    int64_t scan_mfma = 1;                       // scans: fp64 matrix-core kernel when many points share a cell
    int64_t scan_pow = 1;                        // dense-data scans run on a copy of the rows with the bins ordered by their count, where n log mu
                                                 // over a lane's bins becomes n log of their product (ensure_sorted_rows, k_scan_mfma PROD = 2)
    DevBuf ps_sorted, cnt_sorted;                // that copy: [A*S][Bp] rows and [Bp] counts of dataset 0
    int64_t sorted_epoch = -1;                   // data epoch it was (or could not be) built for
    bool sorted_ok = false;
    int64_t n_sorted_scans = 0;                  // how often a scan ran on it (observability)
    int64_t mail_timeout_ms = 2000;              // in-launch finish: how long a collecting block waits for a partial sum before BI_ST_INTERNAL
    int64_t debug_skip_post = -1, debug_late_post = -1;   // fault injection for the next mailbox launch (bi_params.h)
    int64_t n_mail_resets = 0;                   // how often the mailbox had to be emptied after a collector gave up
    std::vector<void*> user_allocs;              // bi_device_alloc buffers still alive: freed with the context
    int64_t bb_max_group = 8;                    // points per Beeston-Barlow work item (16: one wave per SIMD, accumulators partly in AGPRs)
    int64_t scan_bb = 1;                         // parameter: device-planned Beeston-Barlow batches on the matrix cores (k_scan_bb)
    int64_t scan_bb_min = 64;                    // ... from this many points on
    int64_t n_bb_scan_launches = 0;              // read-only
    int64_t device_plan_min = 512;               // batches at least this large are planned on the device

    // the events of the last bi_simulate_events into this context: coordinates [k][N], source index [N]
    DevBuf sim_coords, sim_source;
    int sim_k = 0;
    int64_t sim_n = -1;

    // scratch
    DevBuf scratch, scratch2, logmu;

    // recycle cache for the small transient buffers of a call (descriptors, partials): hipMalloc / hipFree
    // cost tens of microseconds each and every call needs about ten of them.  All work of a context is
    // ordered on its one stream, so handing a buffer to the next call is safe without a sync.
    std::vector<DevBuf> cache;
    size_t cache_bytes = 0;

    int64_t epoch = 0;  // bumped by every model / data upload

    // tunables
    int64_t blocks_per_cu = 8;
    int64_t max_group = kMaxG;
    int64_t xcd_affine = 1;                      // multi-item launches: tile chunks keep their XCD across items
    int64_t fuse_max_blocks = (int64_t)1 << 20;  // single evaluations: finish inside the launch up to this many blocks
    int64_t single_blocks_per_cu = 4;            // single evaluations: blocks per CU the launch shape aims for
    int64_t fuse_finish = 1;                     // batched launches of few items: the last block of an item finishes it (no k_finish launch)
    int64_t single_kernel = 1;                   // bi_eval(P = 1): one fused launch (0: two-kernel fallback)
    int64_t nt_loads = 2;                        // nontemporal template loads: 0 never, 1 always, 2 when no reuse
    int64_t sparse = 1;                          // use the sparse forms when they are exactly equivalent
    int64_t compact_budget = (int64_t)16 << 30;  // bytes of HBM the compacted templates may take

    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
};

namespace {

int fail(bi_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), BI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr size_t kCacheMaxBuf = (size_t)1 << 30;      // only buffers up to 1 GiB are recycled (a 10^6-point scan: 256 + 640 MB)
constexpr size_t kCacheMaxTotal = (size_t)4 << 30;    // at most 4 GiB parked (given back when an allocation fails)
constexpr size_t kCacheMaxEntries = 256;

void dev_free(DevBuf& b) {
    if (b.p && !b.view) {
        bi_ctx* c = b.owner;
        if (c && b.bytes <= kCacheMaxBuf && c->cache.size() < kCacheMaxEntries && c->cache_bytes + b.bytes <= kCacheMaxTotal) {
            c->cache.push_back(b);
            c->cache_bytes += b.bytes;
        } else {
            (void)hipFree(b.p);
        }
    }
    b.p = nullptr;
    b.bytes = 0;
    b.view = false;
}

void drop_recycle_cache(bi_ctx* c) {
    (void)hipGetLastError();
    for (auto& q : c->cache) (void)hipFree(q.p);
    c->cache.clear();
    c->cache_bytes = 0;
}

int dev_alloc(bi_ctx* c, DevBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return BI_OK;
    if (b.p) dev_free(b);
    if (bytes == 0) bytes = 16;
    if (bytes <= kCacheMaxBuf) {  // smallest parked buffer that fits without wasting more than 4x
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c->cache.size(); ++i)
            if (c->cache[i].bytes >= bytes && c->cache[i].bytes <= 4 * bytes + 4096 &&
                (best == (size_t)-1 || c->cache[i].bytes < c->cache[best].bytes))
                best = i;
        if (best != (size_t)-1) {
            b = c->cache[best];
            c->cache_bytes -= b.bytes;
            c->cache[best] = c->cache.back();
            c->cache.pop_back();
            return BI_OK;
        }
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        drop_recycle_cache(c);                  // give parked memory back and retry once
        e = hipMalloc(&b.p, bytes);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, BI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    b.bytes = bytes;
    b.owner = c;
    return BI_OK;
}

template <class T>
int dev_upload(bi_ctx* c, DevBuf& b, const std::vector<T>& h) {
    int rc = dev_alloc(c, b, h.size() * sizeof(T));
    if (rc) return rc;
    if (!h.empty()) HIP_TRY(c, hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return BI_OK;
}

// Several small host arrays -> the device with ONE copy: they are laid out back to back (64-byte aligned) in the
// context's pinned staging block and moved with a single hipMemcpyAsync into one device slab.  `out_bytes` more
// bytes of the pinned block are reserved behind them for results a kernel writes straight to host memory.
// Sub-pointers: dev(i), host_out().  The block is owned by the context and reused by the next call, so the caller
// synchronises the stream before returning.
struct PackedUpload {
    std::vector<size_t> off;
    size_t in_bytes = 0, out_off = 0;
    char* dev_base = nullptr;
    char* host_base = nullptr;
    template <typename T>
    const T* dev(size_t i) const { return reinterpret_cast<const T*>(dev_base + off[i]); }
    void* host_out() const { return host_base + out_off; }
};

inline int packed_upload(bi_ctx* c, const std::vector<std::pair<const void*, size_t>>& parts, size_t out_bytes, PackedUpload& pu,
                         DevBuf* slab = nullptr) {   // slab: a caller-owned device buffer instead of the context's
    pu.off.clear();
    size_t total = 0;
    for (const auto& part : parts) {
        pu.off.push_back(total);
        total += (part.second + 63) / 64 * 64;
    }
    pu.in_bytes = total;
    pu.out_off = total;
    const size_t need = total + (out_bytes + 63) / 64 * 64 + 64;
    if (c->pack_host_bytes < need) {
        if (c->pack_host) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipHostFree(c->pack_host)); c->pack_host = nullptr; c->pack_host_bytes = 0; }
        const size_t grow = std::max<size_t>(2 * need, 65536);
        HIP_TRY(c, hipHostMalloc(&c->pack_host, grow, hipHostMallocDefault));
        c->pack_host_bytes = grow;
    }
    DevBuf& dst = slab ? *slab : c->pack_dev;
    int rc = dev_alloc(c, dst, std::max<size_t>(total, 64));
    if (rc) return rc;
    pu.host_base = (char*)c->pack_host;
    pu.dev_base = (char*)dst.p;
    for (size_t i = 0; i < parts.size(); ++i)
        if (parts[i].second) memcpy(pu.host_base + pu.off[i], parts[i].first, parts[i].second);
    if (total) HIP_TRY(c, hipMemcpyAsync(pu.dev_base, pu.host_base, total, hipMemcpyHostToDevice, c->stream));
    return BI_OK;
}

void free_plan_buffers(bi_plan* p) {
    for (auto& k : p->classes) {
        dev_free(k.rowoff); dev_free(k.coef); dev_free(k.aux); dev_free(k.item_cnt); dev_free(k.item_tiles);
        dev_free(k.perm); dev_free(k.slot_lg); dev_free(k.partial); dev_free(k.pflags); dev_free(k.rowoff_full);
    }
    dev_free(p->bad_idx); dev_free(p->nan_idx); dev_free(p->out); dev_free(p->status); dev_free(p->grp_first); dev_free(p->grp_items);
    dev_free(p->slab); dev_free(p->bad); dev_free(p->sorted_idx); dev_free(p->keep_z); dev_free(p->keep_rs);
}

}  // namespace
