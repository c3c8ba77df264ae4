// bi_common.h -- what every translation unit of libblueice_hip includes: the runtime, the context / plan objects, the
// device helpers, and the launchers through which the main translation unit (blueice_hip.hip: the C ABI, planning, the
// small kernels) reaches the heavy template families, each of which is compiled in its own translation unit so that a
// clean build runs on several cores (blueice_amd/build.py):
//     tu_morph.hip        k_morph_reduce (values, gradients, unbinned), k_morph_single        bi_k_morph.h
//     tu_scan.hip         k_scan_mfma, k_scan_valid                                          bi_k_scan.h
//     tu_scan_sorted.hip  k_scan_sorted                                                      bi_scan_sorted.h
//     tu_grad.hip         k_grad_mfma, k_morph_bbgrad                                        bi_k_grad_mfma.h, bi_k_bbgrad.h
//     tu_scan_bb.hip      k_scan_bb (Beeston-Barlow scans on the matrix cores)               bi_k_scan_bb.h
//     tu_prim.hip         the rocPRIM sorts and scans (instantiated once, behind plain functions)
// gfx950 only; no kernel is defined in two translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/blueice_hip.h"

#define BI_VERSION "blueice_hip 0.1 (gfx950)"

#include "bi_context.h"
#include "bi_log_table.h"
#include "bi_dev_common.h"

namespace {

struct EventScope {
    bi_ctx* c;
    size_t idx = (size_t)-1;
    explicit EventScope(bi_ctx* ctx) : c(ctx) {       // (kernels on the context's second stream are only launched when profiling is off)
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

}  // namespace

// ---- launchers of the kernel families (defined in tu_*.hip) ----------------------------------------------------------
// k_morph_reduce<G, BB, NT, MODE>: G points per cell pass (1, 2, 4, 8, 16); MODE 0 binned / 2 unbinned by the context
void launch_morph_g(bi_ctx* c, int G, const LaunchArgs& a, dim3 grid, bool bb, bool nt);
// ... MODE 1 / 3: value + gradient columns of one point (G = 2, 4, 8, 16 columns)
void launch_morph_grad(bi_ctx* c, int G, const LaunchArgs& a, dim3 grid, bool nt);
// k_morph_single<BB, NT, MODE, FUSE>: the single-point call
void launch_morph_single(bi_ctx* c, bool bb, bool nt, bool fuse, dim3 grid, const LaunchArgs& a, const SingleDesc& d);
// k_morph_bbgrad<G, DZ, NT>: value + gradient with Beeston-Barlow; BI_ERR_INVALID for a column count without a variant
int launch_morph_bbgrad(bi_ctx* c, int G, int DZ, const LaunchArgs& a, dim3 grid, bool nt);
// k_scan_mfma<CB, KG, MASK, PROD>: rows in bin order; prod = the compacted rows' product form (CB = 2 only)
void launch_scan_mfma(bi_ctx* c, int cb, bool prod, int NS, dim3 grid, const ScanArgs& a);
// k_scan_valid<4, KG, MASK>: the validity pass of split scans
void launch_scan_valid(bi_ctx* c, int NS, dim3 grid, const ValidArgs& a);
// k_scan_sorted<KG, MASK>: rows ordered by count, KG = ceil(NS / 4) in 1 .. 8
void launch_scan_sorted(bi_ctx* c, int NS, dim3 grid, const ScanArgs& a);
// blocks of a scan kernel variant one CU holds at a time (hipOccupancyMaxActiveBlocksPerMultiprocessor; 0 = unknown):
// (valid, CB, 4-stream groups 1 << kg, masked) / k_scan_sorted with KG groups
int occupancy_scan(bool valid, int cb, int kg, bool mask);
int occupancy_scan_sorted(int KG, bool mask);
// k_grad_mfma<KG, MASK>
void launch_grad_mfma(bi_ctx* c, int NS, dim3 grid, const GradMfmaArgs& a);
// k_scan_bb<KGT>: Beeston-Barlow scans on the matrix cores; scan_bb_variant: the KGT for (streams into U, corners), 0 = none fits
int scan_bb_variant(int n0, int nc);
void launch_scan_bb(bi_ctx* c, int kgt, dim3 grid, const BbScanArgs& a);
int occupancy_scan_bb(int kgt);
