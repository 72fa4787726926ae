// trace4d.hip — per-tile timeline of gemm4d (wall_clock64 stamps, 100 MHz): prologue / K loop / epilogue per tile,
// idle gap between consecutive tiles of a CU, and how many CUs are inside an epilogue at the same time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DG4D_TRACE -I zett_amd/csrc -I tools/experiments tools/experiments/trace4d.hip -o tools/trace4d
//   tools/trace4d M N K epi      (epi: 0 = bf16 out, 1 = bias + GELU(erf) bf16 out, 5 = bias + residual, fp32 out)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "gemm4p.hip.h"
using namespace zett;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(bf16_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; p[i] = f32_to_bf16(((float)(x & 0xffff) / 32768.f - 1.f) * 0.1f); }
}
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), epi = atoi(argv[4]);
    bf16_t *A, *W, *C; float *res, *cf, *bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMalloc(&cf, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4));
    fill<<<2048, 256>>>(A, (size_t)M * K, 1); fill<<<2048, 256>>>(W, (size_t)N * K, 2);
    CK(hipMemset(res, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, N * 4));
    GemmArgs<bf16_t> g{}; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.epi.split_col = 0x7fffffff;
    if (epi == 5) { g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = N; g.epi.out_f32 = cf; g.epi.ld_f32 = N; }
    else if (epi == 1) { g.epi.bias = bias; g.epi.act = ACT_GELU_ERF; g.epi.out_lo = C; g.epi.ld_lo = N; }
    else { g.epi.out_lo = C; g.epi.ld_lo = N; }
    {   // STAGGER=<us> spreads the first round's start over that many microseconds (MODE 0: by XCD, 1: by CU within the XCD)
        int ticks = getenv("STAGGER") ? atoi(getenv("STAGGER")) * 100 : 0, mode = getenv("MODE") ? atoi(getenv("MODE")) : 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g4d_stagger_ticks), &ticks, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g4d_stagger_mode), &mode, 4));
        printf("stagger %d us mode %d\n", ticks / 100, mode);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool persistent = getenv("PERSISTENT") != nullptr;      // PERSISTENT=1: gemm4p (the timeline then has no separate prologue, and the fine-grained epilogue milestones are gemm4d's)
    auto launch = [&]() { return persistent ? launch_gemm4p<bf16_t>(g, 0) : launch_gemm4d<bf16_t>(g, 0); };
    for (int i = 0; i < 3; ++i) CK(launch());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) CK(launch());
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("M=%d N=%d K=%d epi=%d: %.3f ms  %.0f TF\n", M, N, K, epi, ms, 2.0 * M * N * K / ms / 1e9);
    CK(hipDeviceSynchronize());
    const int tiles = std::min(32768, ((M + 255) / 256) * ((N + 255) / 256));
    std::vector<unsigned long long> tr((size_t)tiles * 8);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g4d_trace), tr.size() * 8));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < tiles; ++b) { t0 = std::min(t0, tr[b * 8]); t1 = std::max(t1, tr[b * 8 + 3]); }
    printf("launch span %.1f us, %d tiles\n", (t1 - t0) / 100.0, tiles);
    std::vector<double> pro, loop, ep;
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < tiles; ++b) {
        const unsigned long long* o = &tr[b * 8];
        pro.push_back((o[1] - o[0]) / 100.0); loop.push_back((o[2] - o[1]) / 100.0); ep.push_back((o[3] - o[2]) / 100.0);
        by_cu[(unsigned)(((o[4] >> 8) & 0xff) | ((o[5] & 0xf) << 8))].push_back(b);
    }
    auto stat = [](std::vector<double> v, const char* n) {
        std::sort(v.begin(), v.end());
        double s = 0; for (double x : v) s += x;
        printf("  %-9s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", n, s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    stat(pro, "prologue"); stat(loop, "k-loop"); stat(ep, "epilogue");
    std::vector<double> gap;
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return tr[a * 8] < tr[b * 8]; });
        for (size_t i = 1; i < v.size(); ++i) gap.push_back(((double)tr[v[i] * 8] - (double)tr[v[i - 1] * 8 + 3]) / 100.0);
    }
    printf("  CUs seen: %zu\n", by_cu.size());
    if (!gap.empty()) stat(gap, "gap");
    // concurrency: for 2000 sample instants, how many tiles are in their epilogue
    std::vector<int> conc;
    for (int s = 0; s < 2000; ++s) {
        const unsigned long long t = t0 + (t1 - t0) * (unsigned long long)s / 2000;
        int c = 0;
        for (int b = 0; b < tiles; ++b) c += (tr[b * 8 + 2] <= t && t < tr[b * 8 + 3]);
        conc.push_back(c);
    }
    std::sort(conc.begin(), conc.end());
    double cs = 0; for (int c : conc) cs += c;
    printf("  CUs in an epilogue at a random instant: mean %.1f  p50 %d  p90 %d  p99 %d  max %d\n", cs / conc.size(), conc[1000], conc[1800], conc[1980], conc.back());
    {
        std::vector<unsigned long long> te((size_t)tiles * 16);
        CK(hipMemcpyFromSymbol(te.data(), HIP_SYMBOL(g4d_trace_epi), te.size() * 8));
        const char* names[10] = {"bar+cols", "p0 res issued", "p0 staged", "p0 vmcnt0", "p0 drained", "p1 res issued", "p1 staged", "p1 vmcnt0", "p1 drained", "stores acked"};
        printf("  wave 0 epilogue milestones (us after the K loop, mean over tiles):\n");
        for (int k = 0; k < 10; ++k) { double s = 0; for (int b = 0; b < tiles; ++b) s += te[b * 16 + k]; printf("    %-14s %6.2f\n", names[k], s / tiles / 100.0); }
    }
    // first CU's timeline
    auto& v0 = by_cu.begin()->second;
    for (size_t i = 0; i < v0.size() && i < 24; ++i) {
        const unsigned long long* o = &tr[v0[i] * 8];
        printf("    cu0 tile %5d: start %8.2f  pro %5.2f  loop %7.2f  epi %6.2f\n", v0[i], (o[0] - t0) / 100.0, (o[1] - o[0]) / 100.0, (o[2] - o[1]) / 100.0, (o[3] - o[2]) / 100.0);
    }
    return 0;
}
