// trace4r.hip — where the four-wave experiment kernel waits (s_memtime stamps around the vmcnt wait and the barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DG4R_TRACE -I zett_amd/csrc tools/experiments/trace4r.hip -o tools/trace4r
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm4r.hip.h"
using namespace zett;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(bf16_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; p[i] = f32_to_bf16(((float)(x & 0xffff) / 32768.f - 1.f) * 0.1f); }
}
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    bf16_t *A, *W, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    fill<<<2048, 256>>>(A, (size_t)M * K, 1); fill<<<2048, 256>>>(W, (size_t)N * K, 2);
    GemmArgs<bf16_t> g{}; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.epi.split_col = 0x7fffffff;
    g.epi.out_lo = C; g.epi.ld_lo = N;
    if (argc > 4 && atoi(argv[4]) == 2) {
        float *res, *cf, *bias;
        CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMalloc(&cf, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4));
        CK(hipMemset(res, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, N * 4));
        g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = N; g.epi.out_f32 = cf; g.epi.ld_f32 = N;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(launch_gemm4r<bf16_t>(g, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) CK(launch_gemm4r<bf16_t>(g, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("M=%d N=%d K=%d: %.3f ms  %.0f TF\n", M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
    std::vector<unsigned long long> tr(4096 * 32);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g4r_trace), tr.size() * 8));
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    double loop = 0, wv = 0, wb = 0, rt = 0; int n = 0;
    for (int b = 0; b < tiles && b < 4096; ++b) for (int w = 0; w < 4; ++w) { const unsigned long long* o = &tr[(b * 4 + w) * 8]; loop += o[0]; wv += o[1]; wb += o[2]; rt += o[4]; ++n; }
    const double steps = tr[3];
    printf("shader clock during the K loops: %.0f MHz (s_memtime ticks per 10 ns s_memrealtime tick x 100)\n", loop / rt * 100.0);
    printf("per K step (shader clocks, mean over %d waves): loop %.0f  vmcnt-wait %.0f  barrier-wait %.0f  (64 MFMAs = 2048 clk at full rate)\n", n, loop / n / steps, wv / n / steps, wb / n / steps);
    {   // per-tile timeline of CU slots: prologue, K loop, epilogue (10 ns ticks), and the gap to the next workgroup on the same CU is not visible here
        double pro = 0, lp = 0, epi = 0; int m = 0;
        for (int b = 0; b < tiles && b < 4096; ++b) { const unsigned long long* o = &tr[(b * 4) * 8]; pro += o[5]; lp += o[4]; epi += (double)o[7] - (double)o[6] - (double)o[5] - (double)o[4]; ++m; }
        printf("per tile (us): prologue %.2f  K loop %.2f  epilogue %.2f   (kernel %.3f ms over %.1f tile rounds = %.2f us per round)\n", pro / m / 100, lp / m / 100, epi / m / 100, ms, tiles / 256.0, ms * 1000 / (tiles / 256.0));
    }
    for (int b : {0, 1, 300}) for (int w = 0; w < 4; ++w) { const unsigned long long* o = &tr[(b * 4 + w) * 8]; printf("  block %d wave %d: loop/step %.0f vm %.0f bar %.0f\n", b, w, o[0] / steps, o[1] / steps, o[2] / steps); }
    return 0;
}
