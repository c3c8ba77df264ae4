// bi_launch.h -- kernel launch dispatch (template instantiation tables), HIP-event scopes, state checks.
#pragma once

namespace {

struct EventScope {
    bi_ctx* c;
    size_t idx = (size_t)-1;
    explicit EventScope(bi_ctx* ctx) : c(ctx) {
        if (!c->profiling) return;
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            c->ev_pool.emplace_back(a, b);
        }
        idx = c->ev_used++;
        (void)hipEventRecord(c->ev_pool[idx].first, c->stream);
    }
    ~EventScope() {
        if (idx != (size_t)-1) (void)hipEventRecord(c->ev_pool[idx].second, c->stream);
    }
};

template <int G>
void launch_morph(bi_ctx* c, const LaunchArgs& a, dim3 grid, bool bb, bool nt) {
    if (c->unbinned) {
        if (nt) hipLaunchKernelGGL((k_morph_reduce<G, false, true, 2>), grid, dim3(kThreads), 0, c->stream, a);
        else hipLaunchKernelGGL((k_morph_reduce<G, false, false, 2>), grid, dim3(kThreads), 0, c->stream, a);
        return;
    }
    if (bb && nt) hipLaunchKernelGGL((k_morph_reduce<G, true, true>), grid, dim3(kThreads), 0, c->stream, a);
    else if (bb) hipLaunchKernelGGL((k_morph_reduce<G, true, false>), grid, dim3(kThreads), 0, c->stream, a);
    else if (nt) hipLaunchKernelGGL((k_morph_reduce<G, false, true>), grid, dim3(kThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((k_morph_reduce<G, false, false>), grid, dim3(kThreads), 0, c->stream, a);
}

void launch_morph_grad(bi_ctx* c, int G, const LaunchArgs& a, dim3 grid, bool nt) {
    EventScope ev(c);
#define BI_GRAD_CASE(GG)                                                                                          \
    case GG:                                                                                                      \
        if (c->unbinned) {                                                                                        \
            if (nt) hipLaunchKernelGGL((k_morph_reduce<GG, false, true, 3>), grid, dim3(kThreads), 0, c->stream, a); \
            else hipLaunchKernelGGL((k_morph_reduce<GG, false, false, 3>), grid, dim3(kThreads), 0, c->stream, a); \
        } else if (nt) hipLaunchKernelGGL((k_morph_reduce<GG, false, true, 1>), grid, dim3(kThreads), 0, c->stream, a); \
        else hipLaunchKernelGGL((k_morph_reduce<GG, false, false, 1>), grid, dim3(kThreads), 0, c->stream, a);   \
        break;
    switch (G) {
        BI_GRAD_CASE(2)
        BI_GRAD_CASE(4)
        BI_GRAD_CASE(8)
        default:
            BI_GRAD_CASE(16)
    }
#undef BI_GRAD_CASE
}

// value + gradient with Beeston-Barlow: G columns in all (padded 1 + d + S), DZ of them (padded 1 + d) for the P / a streams
int launch_morph_bbgrad(bi_ctx* c, int G, int DZ, const LaunchArgs& a, dim3 grid, bool nt) {
    EventScope ev(c);
#define BI_BBG(GG, ZZ)                                                                                             \
    do {                                                                                                            \
        if (nt) hipLaunchKernelGGL((k_morph_bbgrad<GG, ZZ, true>), grid, dim3(kThreads), 0, c->stream, a);          \
        else hipLaunchKernelGGL((k_morph_bbgrad<GG, ZZ, false>), grid, dim3(kThreads), 0, c->stream, a);            \
    } while (0)
    if (G == 8 && DZ == 4) BI_BBG(8, 4);
    else if (G == 8 && DZ == 8) BI_BBG(8, 8);
    else if (G == 16 && DZ == 4) BI_BBG(16, 4);
    else if (G == 16 && DZ == 8) BI_BBG(16, 8);
    else return BI_ERR_INVALID;
#undef BI_BBG
    return BI_OK;
}

// nt: the launch streams its template rows exactly once (no two items touch the same anchor), so the loads
// carry the nontemporal hint: +8 % HBM rate on gfx950; with shared rows the default policy (L2 / MALL) wins.
void launch_morph_g(bi_ctx* c, int G, const LaunchArgs& a, dim3 grid, bool bb, bool nt) {
    EventScope ev(c);
    switch (G) {
        case 1: launch_morph<1>(c, a, grid, bb, nt); break;
        case 2: launch_morph<2>(c, a, grid, bb, nt); break;
        case 4: launch_morph<4>(c, a, grid, bb, nt); break;
        case 8: launch_morph<8>(c, a, grid, bb, nt); break;
        default: launch_morph<16>(c, a, grid, bb, nt); break;
    }
}

// the mailbox of in-launch finishing: allocated and emptied once (every collector leaves its slots empty again)
constexpr int64_t kMailSlots = (int64_t)1 << 20;       // 8 MB
constexpr int64_t kMailFlagWords = (int64_t)1 << 16;

int ensure_mail(bi_ctx* c) {
    if (c->mail.p) return BI_OK;
    void* p = nullptr;
    void* f = nullptr;
    // not from the recycle cache: these must keep their contents between calls
    if (hipMalloc(&p, (size_t)kMailSlots * sizeof(double)) != hipSuccess) return BI_ERR_NOMEM;
    if (hipMalloc(&f, (size_t)kMailFlagWords * sizeof(unsigned)) != hipSuccess) { (void)hipFree(p); return BI_ERR_NOMEM; }
    hipLaunchKernelGGL(k_mail_init, dim3((unsigned)((kMailSlots + 255) / 256)), dim3(256), 0, c->stream, (unsigned long long*)p, kMailSlots);
    if (hipGetLastError() != hipSuccess || hipMemsetAsync(f, 0, (size_t)kMailFlagWords * sizeof(unsigned), c->stream) != hipSuccess) {
        (void)hipFree(p);
        (void)hipFree(f);
        return BI_ERR_HIP;
    }
    c->mail.p = p; c->mail.bytes = (size_t)kMailSlots * sizeof(double); c->mail.owner = nullptr;
    c->mail_flags.p = f; c->mail_flags.bytes = (size_t)kMailFlagWords * sizeof(unsigned); c->mail_flags.owner = nullptr;
    return BI_OK;
}

// a launch that finishes through the mailbox: the collector's patience, and the injected faults (consumed here)
void arm_mail(bi_ctx* c, LaunchArgs& a) {
    a.mail_timeout = c->mail_timeout_ms * kMailTicksPerMs;
    a.skip_post = (int)c->debug_skip_post;
    a.late_post = (int)c->debug_late_post;
    c->debug_skip_post = c->debug_late_post = -1;
}

// after a collector gave up (BI_ST_INTERNAL) the mailbox may hold values nobody took: empty it again
void reset_mail(bi_ctx* c) {
    if (!c->mail.p) return;
    ++c->n_mail_resets;
    (void)hipStreamSynchronize(c->stream);
    hipLaunchKernelGGL(k_mail_init, dim3((unsigned)((kMailSlots + 255) / 256)), dim3(256), 0, c->stream, (unsigned long long*)c->mail.p, kMailSlots);
    (void)hipMemsetAsync(c->mail_flags.p, 0, (size_t)kMailFlagWords * sizeof(unsigned), c->stream);
    (void)hipStreamSynchronize(c->stream);
}

int check_ready(bi_ctx* c, bool need_data) {
    if (!c) return BI_ERR_INVALID;
    if (c->pending) return fail(c, BI_ERR_STATE, "a bi_eval_begin is outstanding on this context: call bi_eval_end first");
    if (!c->model_ready) return fail(c, BI_ERR_STATE, "no model uploaded (prepare() first)");
    if (need_data && !c->data_ready) return fail(c, BI_ERR_STATE, "no data uploaded (set_data() first)");
    return BI_OK;
}

int n_tiles_of(const bi_ctx* c) { return (int)(c->Bp / kTile); }

}  // namespace
