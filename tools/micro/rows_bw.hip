// micro-benchmark: the morph kernel's access pattern -- ITEMS work items, each streaming ROWS rows of 8 MB concurrently
// -- without arithmetic, for several bytes-per-lane-per-row: does a wider contiguous run per wave raise the ceiling?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kThreads = 256;
template <int W>   // W doubles per lane per row (2 = 16 B, 4 = 32 B, 8 = 64 B)
__global__ __launch_bounds__(kThreads) void k_rows(const double* __restrict__ ps, int64_t Bp, int64_t total_rows, int64_t first, int rows,
                                                   int n_tiles, int chunks_in, double* __restrict__ sink) {
    const int64_t row0 = first + (int64_t)blockIdx.y * rows;
    const int chunks = (chunks_in > 1 && n_tiles >= 64 * chunks_in) ? chunks_in : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    double s = 0.0;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kThreads * W + threadIdx.x * 2;     // lanes interleave 16-byte pieces: piece q at + q * 512 doubles
#pragma unroll 8
        for (int r = 0; r < rows; ++r) {
            const double* p = ps + ((row0 + r) % total_rows) * Bp + bin0;
#pragma unroll
            for (int q = 0; q < W / 2; ++q) {
                s += __builtin_nontemporal_load(p + q * kThreads * 2) + __builtin_nontemporal_load(p + q * kThreads * 2 + 1);
            }
        }
    }
    if (s == 0.123456789) sink[0] = s;
}
template <int W>
void run(const double* ps, int64_t Bp, int64_t total_rows, int items, int rows, int bpc, double* sink) {
    const int n_tiles = (int)(Bp / (kThreads * W));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    const int nbx = 256 * bpc / items;
    for (int r = 0; r < 6; ++r) {
        const int64_t first = ((int64_t)r * items * rows * 7) % total_rows;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rows<W>, dim3(nbx, items), dim3(kThreads), 0, 0, ps, Bp, total_rows, first, rows, n_tiles, 8, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r) best = std::max(best, (double)items * rows * Bp * 8.0 / (ms * 1e6));
    }
    printf("  %2d B per lane per row, %2d blocks/CU (%d blocks per item): %.0f GB/s\n", W * 8, bpc, nbx, best);
}
int main() {
    const int64_t Bp = 1000448 / 1024 * 1024 + 1024, total_rows = 500;      // ~8 MB rows, 4 GB
    double* ps; (void)hipMalloc(&ps, total_rows * Bp * 8); (void)hipMemset(ps, 0, total_rows * Bp * 8);
    double* sink; (void)hipMalloc(&sink, 8);
    for (int rows : {32, 16, 113})
        for (int items : {8, 1}) {
            printf("%d items x %d concurrent rows\n", items, rows);
            for (int bpc : {8, 16, 32}) { run<2>(ps, Bp, total_rows, items, rows, bpc, sink); run<4>(ps, Bp, total_rows, items, rows, bpc, sink); run<8>(ps, Bp, total_rows, items, rows, bpc, sink); }
        }
    return 0;
}
