// tu_scan_bb.hip -- translation unit of k_scan_bb (bi_k_scan_bb.h): Beeston-Barlow scans on the fp64 matrix cores.  Variants
// with compile-time stream segments for models with 8 or 16 corners (3 or 4 shape parameters with two or more anchors) and up
// to seven other sources; one generic variant per number of 4-stream groups (rounded up to a multiple of four) for every other
// shape.  See bi_common.h for how the library is split.
#include "bi_common.h"
#include "bi_k_scan_bb.h"

namespace {

// (corner groups KGP, other sources SO) -> the static variant; 0 = none
#define BI_BB_STATIC(X) \
    X(2, 1) X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 6) X(2, 7) \
    X(4, 1) X(4, 2) X(4, 3) X(4, 4) X(4, 5) X(4, 6)

bool has_static(int kgp, int so) {
#define X(P, S) if (kgp == P && so == S) return true;
    BI_BB_STATIC(X)
#undef X
    return false;
}

}  // namespace

// The variant for (streams into U, corners): 1000 * KGP + SO for a static one (n0 = 4 KGP * SO, nc = 4 KGP), else KGT = the
// smallest multiple of 4 that is >= ceil(n0 / 4) + 2 ceil(nc / 4) for the generic kernel; 0 = none fits (more than 128 padded streams)
int scan_bb_variant(int n0, int nc) {
    if (nc >= 8 && nc % 4 == 0 && n0 % nc == 0 && has_static(nc / 4, n0 / nc)) return 1000 * (nc / 4) + n0 / nc;
    const int need = (n0 + 3) / 4 + 2 * ((nc + 3) / 4);
    const int kgt = (need + 3) / 4 * 4;
    return kgt <= 32 ? kgt : 0;
}

namespace {

const void* bb_kernel(int variant) {
    if (variant >= 1000) {
#define X(P, S) if (variant == 1000 * P + S) return (const void*)k_scan_bb<P * S, P, P * S + 2 * P>;
        BI_BB_STATIC(X)
#undef X
        return nullptr;
    }
    switch (variant) {
#define BI_BB(K) case K: return (const void*)k_scan_bb<0, 0, K>;
        BI_BB(4) BI_BB(8) BI_BB(12) BI_BB(16) BI_BB(20) BI_BB(24) BI_BB(28) BI_BB(32)
#undef BI_BB
        default: return nullptr;
    }
}

}  // namespace

void launch_scan_bb(bi_ctx* c, int variant, dim3 grid, const BbScanArgs& a) {
    if (variant >= 1000) {
#define X(P, S) if (variant == 1000 * P + S) { hipLaunchKernelGGL((k_scan_bb<P * S, P, P * S + 2 * P>), grid, dim3(kThreads), 0, c->stream, a); return; }
        BI_BB_STATIC(X)
#undef X
        return;
    }
    switch (variant) {
#define BI_BB(K) case K: hipLaunchKernelGGL((k_scan_bb<0, 0, K>), grid, dim3(kThreads), 0, c->stream, a); break;
        BI_BB(4) BI_BB(8) BI_BB(12) BI_BB(16) BI_BB(20) BI_BB(24) BI_BB(28) BI_BB(32)
#undef BI_BB
        default: break;
    }
}

// resident blocks per CU of the variant, 0 if the runtime cannot say
int occupancy_scan_bb(int variant) {
    const void* f = bb_kernel(variant);
    int blocks = 0;
    if (!f || hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, f, kThreads, 0) != hipSuccess) return 0;
    return blocks;
}
