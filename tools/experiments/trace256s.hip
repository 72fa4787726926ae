// trace256s.hip — per-tile timeline of the persistent experiment kernel (wall_clock64 stamps, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DG256S_TRACE -I zett_amd/csrc tools/experiments/trace256s.hip -o tools/trace256s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm256s.hip.h"
using namespace zett;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(bf16_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; p[i] = f32_to_bf16(((float)(x & 0xffff) / 32768.f - 1.f) * 0.1f); }
}
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), epi = atoi(argv[4]);
    bf16_t *A, *W, *C; float *res, *cf, *bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMalloc(&cf, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4));
    fill<<<2048, 256>>>(A, (size_t)M * K, 1); fill<<<2048, 256>>>(W, (size_t)N * K, 2);
    CK(hipMemset(res, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, N * 4));
    GemmArgs<bf16_t> g{}; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.epi.split_col = 0x7fffffff;
    g.epi.out_lo = C; g.epi.ld_lo = N;
    if (epi == 2) { g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = N; g.epi.out_f32 = cf; g.epi.ld_f32 = N; }
    int skew = getenv("SKEW") ? atoi(getenv("SKEW")) : 0, mode = getenv("SKEWMODE") ? atoi(getenv("SKEWMODE")) : 0;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g256s_skew_ticks), &skew, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g256s_skew_mode), &mode, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(launch_gemm256s<bf16_t>(g, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) CK(launch_gemm256s<bf16_t>(g, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("M=%d N=%d K=%d epi=%d skew=%d mode=%d: %.3f ms  %.0f TF\n", M, N, K, epi, skew, mode, ms, 2.0 * M * N * K / ms / 1e9);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> tr(256 * 64 * 4);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g256s_trace), tr.size() * 8));
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), per = (tiles + 255) / 256;
    for (int b : {0, 1, 7, 8}) {
        printf("block %d (units of 10 ns: loop, barrier, epilogue issue, gap to next loop start):\n", b);
        for (int t = 0; t < per && t < 64; ++t) {
            const unsigned long long* e = &tr[(b * 64 + t) * 4];
            const unsigned long long nxt = t + 1 < per ? tr[(b * 64 + t + 1) * 4] : e[3];
            printf("  tile %2d: start %8llu  kloop %5llu  bar %4llu  epi %5llu  gap %4llu\n", t, e[0] - tr[0], e[1] - e[0], e[2] - e[1], e[3] - e[2], nxt - e[3]);
        }
    }
    return 0;
}
