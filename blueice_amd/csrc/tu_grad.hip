// tu_grad.hip -- translation unit of the gradient kernels with the largest register footprints: k_grad_mfma (value + gradient
// of large batches on the fp64 matrix cores, bi_k_grad_mfma.h) and k_morph_bbgrad (value + gradient with Beeston-Barlow,
// bi_k_bbgrad.h).  See bi_common.h for how the library is split.
#include "bi_common.h"
#include "bi_k_bbgrad.h"
#include "bi_k_grad_mfma.h"

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

void launch_grad_mfma(bi_ctx* c, int NS, dim3 grid, const GradMfmaArgs& ga) {
    const int kg = NS <= 4 ? 1 : (NS <= 8 ? 2 : (NS <= 16 ? 4 : 8));
#define BI_GM(KG)                                                                                                 \
    do {                                                                                                          \
        if (NS == 4 * KG) hipLaunchKernelGGL((k_grad_mfma<KG, false>), grid, dim3(kThreads), 0, c->stream, ga);   \
        else hipLaunchKernelGGL((k_grad_mfma<KG, true>), grid, dim3(kThreads), 0, c->stream, ga);                 \
    } while (0)
    if (kg == 1) BI_GM(1); else if (kg == 2) BI_GM(2); else if (kg == 4) BI_GM(4); else BI_GM(8);
#undef BI_GM
}
