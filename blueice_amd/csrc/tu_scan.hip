// tu_scan.hip -- translation unit of the matrix-core scan kernels over rows in bin order: k_scan_mfma<CB, KG, MASK, PROD> and
// the validity pass of split scans, k_scan_valid<4, KG, MASK> (bi_k_scan.h).  See bi_common.h for how the library is split.
#include "bi_common.h"
#include "bi_k_scan.h"

void launch_scan_mfma(bi_ctx* c, int cb, bool prod, int NS, dim3 sgrid, const ScanArgs& sa) {
    const int kg = NS <= 4 ? 1 : (NS <= 8 ? 2 : (NS <= 16 ? 4 : 8));
#define BI_SCAN(CB, KG)                                                                                           \
    do {                                                                                                          \
        if (CB == 2 && prod) { /* compacted rows in bin order: blocks of counts 1 and 2 take the logarithm of the product mu^n */ \
            if (NS == 4 * KG) hipLaunchKernelGGL((k_scan_mfma<2, KG, false, 1>), sgrid, dim3(kThreads), 0, c->stream, sa); \
            else hipLaunchKernelGGL((k_scan_mfma<2, KG, true, 1>), sgrid, dim3(kThreads), 0, c->stream, sa);     \
        } else if (NS == 4 * KG) hipLaunchKernelGGL((k_scan_mfma<CB, KG, false>), sgrid, dim3(kThreads), 0, c->stream, sa); \
        else hipLaunchKernelGGL((k_scan_mfma<CB, KG, true>), sgrid, dim3(kThreads), 0, c->stream, sa);              \
    } while (0)
#define BI_SCAN_KG(CB)                                                                                            \
    do {                                                                                                          \
        if (kg == 1) BI_SCAN(CB, 1); else if (kg == 2) BI_SCAN(CB, 2); else if (kg == 4) BI_SCAN(CB, 4); else BI_SCAN(CB, 8); \
    } while (0)
    if (cb == 2) BI_SCAN_KG(2); else BI_SCAN_KG(4);
#undef BI_SCAN_KG
#undef BI_SCAN
}

void launch_scan_valid(bi_ctx* c, int NS, dim3 vgrid, const ValidArgs& va) {
    const int kg = NS <= 4 ? 1 : (NS <= 8 ? 2 : (NS <= 16 ? 4 : 8));
#define BI_VALID(KG)                                                                                              \
    do {                                                                                                          \
        if (NS == 4 * KG) hipLaunchKernelGGL((k_scan_valid<4, KG, false>), vgrid, dim3(kThreads), 0, c->stream, va); \
        else hipLaunchKernelGGL((k_scan_valid<4, KG, true>), vgrid, dim3(kThreads), 0, c->stream, va);              \
    } while (0)
    if (kg == 1) BI_VALID(1); else if (kg == 2) BI_VALID(2); else if (kg == 4) BI_VALID(4); else BI_VALID(8);
#undef BI_VALID
}

// resident blocks per CU of the variant for (valid, CB, 4-stream groups 1 << kg, masked), 0 if the runtime cannot say
int occupancy_scan(bool valid, int cb, int kg, bool mask) {
    const void* f = nullptr;
#define BI_PICK(KERNEL, CB)                                                                                        \
    do {                                                                                                           \
        switch (kg * 2 + (mask ? 1 : 0)) {                                                                         \
            case 0: f = (const void*)KERNEL<CB, 1, false>; break;                                                  \
            case 1: f = (const void*)KERNEL<CB, 1, true>; break;                                                   \
            case 2: f = (const void*)KERNEL<CB, 2, false>; break;                                                  \
            case 3: f = (const void*)KERNEL<CB, 2, true>; break;                                                   \
            case 4: f = (const void*)KERNEL<CB, 4, false>; break;                                                  \
            case 5: f = (const void*)KERNEL<CB, 4, true>; break;                                                   \
            case 6: f = (const void*)KERNEL<CB, 8, false>; break;                                                  \
            default: f = (const void*)KERNEL<CB, 8, true>; break;                                                  \
        }                                                                                                          \
    } while (0)
    if (valid) BI_PICK(k_scan_valid, 4);
    else if (cb == 2) BI_PICK(k_scan_mfma, 2);
    else BI_PICK(k_scan_mfma, 4);
#undef BI_PICK
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, f, kThreads, 0) != hipSuccess) return 0;
    return blocks;
}
