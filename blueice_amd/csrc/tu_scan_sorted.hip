// tu_scan_sorted.hip -- translation unit of k_scan_sorted<KG, MASK> (bi_scan_sorted.h): the matrix-core scan over rows ordered
// by count, one variant per number of 4-stream groups.  See bi_common.h for how the library is split.
#include "bi_common.h"
#include "bi_scan_sorted.h"

void launch_scan_sorted(bi_ctx* c, int NS, dim3 sgrid, const ScanArgs& sa) {
#define BI_SORTED(KG)                                                                                             \
    do {                                                                                                          \
        if (NS == 4 * KG) hipLaunchKernelGGL((k_scan_sorted<KG, false>), sgrid, dim3(kThreads), 0, c->stream, sa); \
        else hipLaunchKernelGGL((k_scan_sorted<KG, true>), sgrid, dim3(kThreads), 0, c->stream, sa);              \
    } while (0)
    switch ((NS + 3) / 4) {           // the exact number of 4-stream groups: no matrix work on padding
        case 1: BI_SORTED(1); break; case 2: BI_SORTED(2); break; case 3: BI_SORTED(3); break; case 4: BI_SORTED(4); break;
        case 5: BI_SORTED(5); break; case 6: BI_SORTED(6); break; case 7: BI_SORTED(7); break; default: BI_SORTED(8); break;
    }
#undef BI_SORTED
}

// resident blocks per CU of the variant with KG (1 .. 8) 4-stream groups, 0 if the runtime cannot say
int occupancy_scan_sorted(int KG, bool mask) {
    const void* f = nullptr;
    switch ((KG - 1) * 2 + (mask ? 1 : 0)) {
#define BI_CASE(K) case ((K) - 1) * 2: f = (const void*)k_scan_sorted<K, false>; break; case ((K) - 1) * 2 + 1: f = (const void*)k_scan_sorted<K, true>; break;
        BI_CASE(1) BI_CASE(2) BI_CASE(3) BI_CASE(4) BI_CASE(5) BI_CASE(6) BI_CASE(7) BI_CASE(8)
#undef BI_CASE
    }
    int blocks = 0;
    if (!f || hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, f, kThreads, 0) != hipSuccess) return 0;
    return blocks;
}
