// tu_morph.hip -- translation unit of the morph + reduce kernels: k_morph_reduce<G, BB, NT, MODE> (bi_k_morph.h: batched
// values, gradients, the unbinned likelihood) and k_morph_single (the synchronous single-point call), with their
// instantiation tables.  See bi_common.h for how the library is split.
#include "bi_common.h"
#include "bi_k_morph.h"

namespace {

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

}  // namespace

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


void launch_morph_single(bi_ctx* c, bool bb, bool nt, bool fuse, dim3 grid, const LaunchArgs& a, const SingleDesc& d) {
    const dim3 block(kThreads);
#define BI_SINGLE(BBv, MODEv)                                                                                          \
    do {                                                                                                               \
        if (nt && fuse) hipLaunchKernelGGL((k_morph_single<BBv, true, MODEv, true>), grid, block, 0, c->stream, a, d);    \
        else if (nt) hipLaunchKernelGGL((k_morph_single<BBv, true, MODEv, false>), grid, block, 0, c->stream, a, d);      \
        else if (fuse) hipLaunchKernelGGL((k_morph_single<BBv, false, MODEv, true>), grid, block, 0, c->stream, a, d);    \
        else hipLaunchKernelGGL((k_morph_single<BBv, false, MODEv, false>), grid, block, 0, c->stream, a, d);             \
    } while (0)
    if (c->unbinned) BI_SINGLE(false, 2);
    else if (bb) BI_SINGLE(true, 0);
    else BI_SINGLE(false, 0);
#undef BI_SINGLE
}
