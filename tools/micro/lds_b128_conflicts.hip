// micro-benchmark: cycles per ds_read_b128 of a wave versus the pattern of the 64 lanes' 16-byte cells -- which lanes share a
// conflict domain, and what a random pattern costs against a conflict-free one.  One block of 1024 threads per CU (the toy dot
// kernel's shape), 128 KB of LDS as 8192 cells of 16 bytes; every lane reads cell (base_k + residue(lane)) with base_k a multiple
// of 8 that changes per read (so nothing is a broadcast).
//   hipcc -O3 --offload-arch=gfx950 -o lds_b128 tools/micro/lds_b128_conflicts.hip && ./lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
constexpr int kCells = 8192, kReads = 64;
__global__ __launch_bounds__(1024) void k(const uint16_t* __restrict__ idx /*[patterns][kReads][64]*/, int pattern, int iters, double* out) {
    extern __shared__ double2 s[];
    for (int i = threadIdx.x; i < kCells; i += 1024) s[i] = double2{(double)i, 1.0};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint16_t my[kReads];
#pragma unroll
    for (int r = 0; r < kReads; ++r) my[r] = idx[((size_t)pattern * kReads + r) * 64 + lane];
    double ax = 0.0, ay = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < kReads; ++r) {
            const double2 v = s[(my[r] + it) & (kCells - 1)];
            ax += v.x;
            ay += v.y;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = ax + ay;
}
int main() {
    const char* names[] = {"distinct within aligned 8 lanes (lane % 8)", "same within aligned 8, distinct across (lane / 8 % 8)",
                           "distinct within aligned 16 lanes only ((lane % 16) / 2)", "pairs of lanes (2 per cell group) lane % 4 * 2",
                           "random", "all one residue", "distinct within lanes {l, l+8, l+16, ...} stride 8 (lane / 8)",
                           "quads of one run, distinct in quad, random between quads", "broadcast (all lanes one cell)",
                           "distinct within aligned 8, only 5 of 8 lanes active (others broadcast cell 0)"};
    const int n_pat = 10;
    std::vector<uint16_t> h((size_t)n_pat * kReads * 64);
    std::mt19937 rng(1);
    for (int p = 0; p < n_pat; ++p)
        for (int r = 0; r < kReads; ++r) {
            int quad_res[16][4];
            for (int q = 0; q < 16; ++q) { int perm[8] = {0,1,2,3,4,5,6,7}; for (int i = 7; i > 0; --i) std::swap(perm[i], perm[rng() % (i + 1)]); for (int i = 0; i < 4; ++i) quad_res[q][i] = perm[i]; }
            for (int l = 0; l < 64; ++l) {
                const int base = (int)(rng() % (kCells / 8 - 8)) * 8;
                int res = 0, cell;
                switch (p) {
                    case 0: res = l % 8; break;
                    case 1: res = (l / 8) % 8; break;
                    case 2: res = (l % 16) / 2; break;
                    case 3: res = (l % 4) * 2; break;
                    case 4: res = (int)(rng() % 8); break;
                    case 5: res = 3; break;
                    case 6: res = l / 8; break;
                    case 7: res = quad_res[l / 4][l % 4]; break;
                    default: res = l % 8; break;
                }
                cell = base + res;
                if (p == 8) cell = 800;
                if (p == 9 && (l % 8) >= 5) cell = 0;
                h[((size_t)p * kReads + r) * 64 + l] = (uint16_t)cell;
            }
        }
    uint16_t* d; double* out;
    hipMalloc(&d, h.size() * 2); hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 1024 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kCells * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    for (int p = 0; p < n_pat; ++p) {
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), kCells * 16, 0, d, p, iters, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), kCells * 16, 0, d, p, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 16 waves x iters x kReads wave-level reads
        const double reads = 16.0 * iters * kReads;
        printf("%-92s %6.2f cycles per wave-level ds_read_b128 (at 2.4 GHz)\n", names[p], ms * 1e-3 * 2.4e9 / reads);
    }
    return 0;
}
