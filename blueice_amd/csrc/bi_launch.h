// bi_launch.h -- mailbox set-up of in-launch finishing and state checks (main translation unit).  The instantiation tables of
// the heavy kernel families are in tu_*.hip, declared in bi_common.h.
#pragma once

namespace {

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
