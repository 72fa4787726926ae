// tile_order_sweep.hip — the product gemm4d on the launch shapes of the headline step, by tile order and group size.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zett_amd/csrc tools/experiments/tile_order_sweep.hip -o tools/tile_order_sweep
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm4d.hip.h"
using namespace zett;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(f16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; p[i] = (f16_t)(((float)(x & 0xffff) / 32768.f - 1.f) * scale); }
}
int main() {
    struct Shape { int M, N, K, epi; const char* what; };
    const Shape shapes[] = {{77450, 12288, 4096, 0, "QKV"}, {77450, 4096, 4096, 5, "O-proj (residual)"}, {77450, 8192, 4096, 1, "FFN up (erf-GELU)"},
                            {77450, 4096, 8192, 5, "FFN down (residual)"}, {29187, 4096, 8192, 0, "input projection"}, {32768, 4096, 4096, 0, "heads"}};
    for (const Shape& sh : shapes) {
        f16_t *A, *W, *C; float *res, *cf, *bias;
        CK(hipMalloc(&A, (size_t)sh.M * sh.K * 2)); CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&C, (size_t)sh.M * sh.N * 2));
        CK(hipMalloc(&res, (size_t)sh.M * sh.N * 4)); CK(hipMalloc(&cf, (size_t)sh.M * sh.N * 4)); CK(hipMalloc(&bias, sh.N * 4));
        fill<<<2048, 256>>>(A, (size_t)sh.M * sh.K, 1, 1.0f); fill<<<2048, 256>>>(W, (size_t)sh.N * sh.K, 2, 0.05f);
        CK(hipMemset(res, 0, (size_t)sh.M * sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4));
        printf("%-22s M=%d N=%d K=%d:", sh.what, sh.M, sh.N, sh.K);
        struct Cfg { int order, group; };
        const Cfg cfgs[] = {{0, 4}, {0, 2}, {0, 8}, {0, 16}, {1, 4}, {1, 2}, {1, 8}};
        std::vector<std::vector<float>> t(7);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int round = 0; round < 5; ++round)
            for (int c = 0; c < 7; ++c) {
                GemmArgs<f16_t> g{}; g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.M = sh.M; g.N = sh.N; g.K = sh.K; g.epi.split_col = 0x7fffffff;
                g.tile_order = cfgs[c].order; g.group = cfgs[c].group;
                if (sh.epi == 5) { g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = sh.N; g.epi.out_f32 = cf; g.epi.ld_f32 = sh.N; }
                else if (sh.epi == 1) { g.epi.bias = bias; g.epi.act = ACT_GELU_ERF; g.epi.out_lo = C; g.epi.ld_lo = sh.N; }
                else { g.epi.bias = bias; g.epi.out_lo = C; g.epi.ld_lo = sh.N; }
                CK(launch_gemm4d<f16_t>(g, 0));
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 3; ++i) CK(launch_gemm4d<f16_t>(g, 0));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t[c].push_back(ms / 3);
            }
        for (int c = 0; c < 7; ++c) {
            std::sort(t[c].begin(), t[c].end());
            printf("  %s%d %.0f", cfgs[c].order ? "row" : "col", cfgs[c].group, 2.0 * sh.M * sh.N * sh.K / t[c][2] / 1e9);
        }
        printf("  TF (median of 5)\n");
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(res)); CK(hipFree(cf)); CK(hipFree(bias));
    }
    return 0;
}
