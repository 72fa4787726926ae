// gemm_bench.hip — standalone microbenchmark / A-B harness for the GEMM kernels.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zett_amd/csrc -I tools/experiments tools/gemm_bench.hip -o tools/gemm_bench
//   ./tools/gemm_bench [M N K]...
//
// Random bf16 operands (uniform [-1,1) scaled), every variant checked against the
// 128x128 register-staged kernel, timings from HIP events over interleaved rounds.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gemm.hip.h"
// -DGEMM_BENCH_LEAN: only the product kernels (128x128, gemm8r, gemm384, gemm4d, gemm2w) — compiles in a minute instead of seven
#ifndef GEMM_BENCH_LEAN
#include "gemm256.hip.h"
#include "gemm256p.hip.h"
#include "gemm256r.hip.h"
#include "gemm256e.hip.h"
#include "gemm256s.hip.h"
#include "gemm4w.hip.h"
#include "gemm256l.hip.h"
#include "gemm4r.hip.h"
#include "gemm8x.hip.h"
#include "gemm4dx.hip.h"
#include "gemm8p.hip.h"
#endif
#include "gemm8r.hip.h"
#include "gemm384.hip.h"
#include "gemm4d.hip.h"
#include "gemm2w.hip.h"

using namespace zett;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(bf16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale;
        p[i] = f32_to_bf16(f);
    }
}
// standard normal (Box-Muller) times scale: what torch.randn gives the hipBLASLt yardstick
__global__ void fill_bf16_normal(bf16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        uint32_t y = x * 0x9e3779b9u + 0x7f4a7c15u;
        y ^= y >> 16; y *= 0x7feb352du; y ^= y >> 15; y *= 0x846ca68bu; y ^= y >> 16;
        const float u1 = ((float)(x >> 8) + 1.0f) / 16777217.0f, u2 = (float)(y >> 8) / 16777216.0f;
        p[i] = f32_to_bf16(sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * scale);
    }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale;
    }
}
__global__ void max_diff(const bf16_t* a, const bf16_t* b, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.f;
    for (; i < n; i += stride) m = fmaxf(m, fabsf(bf16_to_f32(a[i]) - bf16_to_f32(b[i])));
    atomicMax((int*)out, __float_as_int(m));
}

struct Variant {
    const char* name;
    hipError_t (*launch)(const GemmArgs<bf16_t>&, hipStream_t);
};

int main(int argc, char** argv) {
    std::vector<std::array<int, 3>> shapes;
    for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
    if (shapes.empty())
        shapes = {{8192, 8192, 8192}, {65536, 12288, 4096}, {65536, 4096, 4096}, {65536, 8192, 4096},
                  {65536, 4096, 8192}, {29187, 4096, 8192}, {32768, 4096, 4096}, {5111, 4096, 4096}};
    CK(hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
#ifdef GEMM_BENCH_LEAN
    std::vector<Variant> variants = {{"g128", launch_gemm<bf16_t>}};
    variants.push_back({"g384", launch_gemm384<bf16_t>});
    variants.push_back({"g8r", launch_gemm8r<bf16_t>});
    variants.push_back({"p4d", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm4d<bf16_t>(g, st, false); }});      // the product kernel, streamlined epilogues
    variants.push_back({"p4dg", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm4d<bf16_t>(g, st, true); }});      // the product kernel, generic drain
    variants.push_back({"g2w", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm2w<bf16_t>(g, st, false); }});      // 128x256, two workgroups per CU
    variants.push_back({"g2wg", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm2w<bf16_t>(g, st, true); }});      // ... with the generic drain
#else
    std::vector<Variant> variants = {{"g128", launch_gemm<bf16_t>}, {"g256", launch_gemm256<bf16_t, 0>}};
    variants.push_back({"g256_spread", launch_gemm256<bf16_t, 1>});
    variants.push_back({"spread_skew", launch_gemm256<bf16_t, 17>});
    CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
    variants.push_back({"g256p", launch_gemm256p<bf16_t, 0>});
    CK(hipFuncSetAttribute((const void*)gemm256r_tn_kernel<bf16_t, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * R_SLOT_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm256r_tn_kernel<bf16_t, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * R_SLOT_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm256e_tn_kernel<bf16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm256e_tn_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS_BYTES));
    variants.push_back({"g256e", launch_gemm256e<bf16_t, 0>});
    CK(hipFuncSetAttribute((const void*)gemm256e_tn_kernel<bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS_BYTES));
    variants.push_back({"g256e_latewait", launch_gemm256e<bf16_t, 2>});
    variants.push_back({"g384", launch_gemm384<bf16_t>});
    variants.push_back({"g256s", launch_gemm256s<bf16_t>});
    variants.push_back({"g4w", launch_gemm4w<bf16_t>});
    variants.push_back({"g256l", launch_gemm256l<bf16_t>});
    variants.push_back({"g4r", launch_gemm4r<bf16_t>});
    variants.push_back({"g8r", launch_gemm8r<bf16_t>});
    variants.push_back({"g8p", launch_gemm8p<bf16_t>});
    variants.push_back({"g8x", launch_gemm8x<bf16_t>});
    variants.push_back({"p4d", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm4d<bf16_t>(g, st, false); }});      // the product kernel, streamlined epilogues
    variants.push_back({"p4dg", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm4d<bf16_t>(g, st, true); }});      // the product kernel, generic drain
    variants.push_back({"g2w", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm2w<bf16_t>(g, st, false); }});      // 128x256, two workgroups per CU
    variants.push_back({"g2wg", [](const GemmArgs<bf16_t>& g, hipStream_t st) { return launch_gemm2w<bf16_t>(g, st, true); }});      // ... with the generic drain
    variants.push_back({"g4d", launch_gemm4dx<bf16_t>});
    variants.push_back({"g4dt4", launch_gemm4dx<bf16_t, 4>});
    variants.push_back({"g4ds1", launch_gemm4dx<bf16_t, 101>});
    variants.push_back({"g4ds2", launch_gemm4dx<bf16_t, 102>});
    variants.push_back({"g4dp", launch_gemm4dx<bf16_t, 200>});
    variants.push_back({"g4dps", launch_gemm4dx<bf16_t, 201>});
    variants.push_back({"g4dq", launch_gemm4dx<bf16_t, 210>});
    variants.push_back({"g4dqs", launch_gemm4dx<bf16_t, 211>});
    variants.push_back({"g4dr", launch_gemm4dx<bf16_t, 220>});
    variants.push_back({"g4dqn", launch_gemm4dx<bf16_t, 212>});
    variants.push_back({"g4dqm", launch_gemm4dx<bf16_t, 213>});
    variants.push_back({"g4dqm2", launch_gemm4dx<bf16_t, 214>});
    variants.push_back({"g4dqm1", launch_gemm4dx<bf16_t, 215>});
    variants.push_back({"g4dqm6", launch_gemm4dx<bf16_t, 216>});
    variants.push_back({"g4dv", launch_gemm4dx<bf16_t, 230>});
    variants.push_back({"g4d1b", launch_gemm4dx<bf16_t, 240>});
    {
        int cfg[2] = {getenv("G4DX_GROUP_M") ? atoi(getenv("G4DX_GROUP_M")) : 4, getenv("G4DX_MAP") ? atoi(getenv("G4DX_MAP")) : 0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(zett::g4dx_cfg), cfg, sizeof(cfg)));
    }
    if (getenv("RING")) { variants.push_back({"g256r4", launch_gemm256r<bf16_t, 4>}); variants.push_back({"g256r5", launch_gemm256r<bf16_t, 5>}); }
    CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES + 2048));
    CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES + 2048));
    CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES + 2048));
    if (getenv("PF")) { variants.push_back({"g256p_pf2", launch_gemm256p<bf16_t, 0, 2>}); variants.push_back({"g256p_pf3", launch_gemm256p<bf16_t, 0, 3>}); }
    if (getenv("ABL")) variants.push_back({"nomfma_pf2", launch_gemm256p<bf16_t, 2, 2>});
    if (getenv("ABL")) {
        variants.push_back({"p_nodma", launch_gemm256p<bf16_t, 1>});
        variants.push_back({"p_nomfma", launch_gemm256p<bf16_t, 2>});
        CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
        variants.push_back({"nomfma_2xdma", launch_gemm256p<bf16_t, 3>});
        variants.push_back({"nomfma_halfdma", launch_gemm256p<bf16_t, 4>});
        CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
        variants.push_back({"mfma_only", launch_gemm256p<bf16_t, 5>});
        CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
        variants.push_back({"mfma_nobar", launch_gemm256p<bf16_t, 6>});
        CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
        variants.push_back({"full_halfdma", launch_gemm256p<bf16_t, 7>});
        CK(hipFuncSetAttribute((const void*)gemm256p_tn_kernel<bf16_t, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
        variants.push_back({"Adma_Wvgpr", launch_gemm256p<bf16_t, 9>});
    }
#endif
    if (const char* only = getenv("ONLY")) {      // comma-separated names; the reference variant 0 always stays
        std::vector<Variant> kept = {variants[0]};
        std::string o = std::string(",") + only + ",";
        for (size_t v = 1; v < variants.size(); ++v)
            if (o.find(std::string(",") + variants[v].name + ",") != std::string::npos) kept.push_back(variants[v]);
        variants = kept;
    }
    const char* epi_env = getenv("EPI");     // 0 = bf16 out only, 1 = bias+gelu_erf bf16 out, 2 = bias+residual f32+bf16 out, 3 = bias+gelu_tanh, 4 = bias+scale/shift f32+bf16 out, 5 = bias+residual f32 out only, 6 = bias+gelu_tanh+residual f32 out, 7 = bias+scale/shift f32 out
    const int epi_mode = epi_env ? atoi(epi_env) : 0;
    const int rounds = getenv("ROUNDS") ? atoi(getenv("ROUNDS")) : 5;
    const int burst = getenv("BURST") ? atoi(getenv("BURST")) : 3;      // back-to-back launches per timing
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        bf16_t *A, *W, *C0, *C1;
        float *bias, *res, *cf, *stats;
        float2* parts;
        const int ldpad = getenv("LDPAD") ? atoi(getenv("LDPAD")) : 0;      // extra elements per operand row (channel spread)
        const int ldk = K + ldpad;
        CK(hipMalloc(&A, ((size_t)M + 512) * ldk * 2)); CK(hipMalloc(&W, (size_t)N * ldk * 2));
        CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
        CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMalloc(&cf, (size_t)M * N * 4));
        CK(hipMalloc(&stats, (size_t)M * 8)); CK(hipMalloc(&parts, (size_t)M * (N / 128 + 1) * 8));
        fill_f32<<<64, 256>>>(stats, (size_t)M * 2, 5, 1.0f);
        if (getenv("ZERO")) {               // all-zero operands: the rate of the schedule without the power limit
            CK(hipMemset(A, 0, (size_t)M * ldk * 2)); CK(hipMemset(W, 0, (size_t)N * ldk * 2));
        } else if (getenv("NORMAL")) {
            fill_bf16_normal<<<2048, 256>>>(A, (size_t)M * ldk, 1, 1.0f);
            fill_bf16_normal<<<2048, 256>>>(W, (size_t)N * ldk, 2, 1.0f);
        } else {
            fill_bf16<<<2048, 256>>>(A, (size_t)M * ldk, 1, 1.0f);
            fill_bf16<<<2048, 256>>>(W, (size_t)N * ldk, 2, 0.05f);
        }
        fill_f32<<<64, 256>>>(bias, N, 3, 0.1f);
        fill_f32<<<2048, 256>>>(res, (size_t)M * N, 4, 1.0f);
        CK(hipDeviceSynchronize());
        auto make = [&](bf16_t* out) {
            GemmArgs<bf16_t> g{};
            g.A = A; g.lda = getenv("LDA0") ? 0 : ldk; g.W = W; g.ldw = getenv("LDA0") ? 0 : ldk; g.M = M; g.N = N; g.K = K;
            g.epi.split_col = 0x7fffffff;
            if (epi_mode == 0) { g.epi.out_lo = out; g.epi.ld_lo = N; }
            else if (epi_mode == 1) { g.epi.bias = bias; g.epi.act = ACT_GELU_ERF; g.epi.out_lo = out; g.epi.ld_lo = N; }
            else if (epi_mode == 3) { g.epi.bias = bias; g.epi.act = ACT_GELU_TANH; g.epi.out_lo = out; g.epi.ld_lo = N; }
            else if (epi_mode == 5) { g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = N; g.epi.out_f32 = cf; g.epi.ld_f32 = N; }   // the library's residual launches: fp32 out only
            else if (epi_mode == 6) { g.epi.bias = bias; g.epi.act = ACT_GELU_TANH; g.epi.residual = res; g.epi.ld_res = N; g.epi.out_f32 = cf; g.epi.ld_f32 = N; }   // ProjectorBlock dense2
            else if (epi_mode == 7) { g.epi.bias = bias; g.epi.scale = res; g.epi.shift = bias; g.epi.out_f32 = cf; g.epi.ld_f32 = N; }   // output head with Rescaler
            else if (epi_mode == 8) {   // LayerNorm-fold producer (attention-output / FFN-down of an encoder layer): LN'd fp32 residual, fp32 + 16-bit out, partial statistics
                g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = N; g.epi.res_stats = stats; g.epi.res_gamma = bias; g.epi.res_beta = bias;
                g.epi.out_f32 = cf; g.epi.ld_f32 = N; g.epi.out_lo = out; g.epi.ld_lo = N; g.epi.stats_part = parts; g.epi.ld_part = M; }
            else if (epi_mode == 9) {   // LayerNorm-fold consumer with erf-GELU (FFN up)
                g.epi.bias = bias; g.epi.act = ACT_GELU_ERF; g.epi.out_lo = out; g.epi.ld_lo = N; g.epi.fold_stats = stats; g.epi.fold_c = bias; }
            else if (epi_mode == 4) { g.epi.bias = bias; g.epi.scale = res; g.epi.shift = bias; g.epi.out_f32 = cf; g.epi.ld_f32 = N; g.epi.out_lo = out; g.epi.ld_lo = N; }
            else { g.epi.bias = bias; g.epi.residual = res; g.epi.ld_res = N; g.epi.out_f32 = cf; g.epi.ld_f32 = N; g.epi.out_lo = out; g.epi.ld_lo = N; }
            return g;
        };
        float* dmax; CK(hipMalloc(&dmax, 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        GemmArgs<bf16_t> gref = make(C0);
        CK(variants[0].launch(gref, 0)); CK(hipDeviceSynchronize());
        std::vector<std::vector<float>> times(variants.size());
        std::vector<float> diffs(variants.size(), 0.f);
        for (size_t v = 1; v < variants.size(); ++v) {
            CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
            GemmArgs<bf16_t> g = make(C1);
            const int stress = getenv("STRESS") ? atoi(getenv("STRESS")) : 1;      // repeat launch + compare (race hunting)
            for (int it = 0; it < stress; ++it) {
                if (it) CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
                CK(variants[v].launch(g, 0)); CK(hipDeviceSynchronize());
                CK(hipMemset(dmax, 0, 4));
                max_diff<<<1024, 256>>>(C0, C1, (size_t)M * N, dmax);
                float d; CK(hipMemcpy(&d, dmax, 4, hipMemcpyDeviceToHost));
                diffs[v] = std::max(diffs[v], d);
            }
        }
        for (int r = 0; r < rounds; ++r) {
            for (size_t v = 0; v < variants.size(); ++v) {
                GemmArgs<bf16_t> g = make(C1);
                CK(hipEventRecord(e0, 0));
                for (int it = 0; it < burst; ++it) CK(variants[v].launch(g, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                times[v].push_back(ms / burst);
            }
        }
        printf("M=%d N=%d K=%d epi=%d :", M, N, K, epi_mode);
        for (size_t v = 0; v < variants.size(); ++v) {
            std::sort(times[v].begin(), times[v].end());
            const float med = times[v][times[v].size() / 2], best = times[v][0];
            const double fl = 2.0 * M * N * K;
            printf("  %s %.0f TF (best %.0f, %.3f ms, maxdiff %.3g)", variants[v].name, fl / med / 1e9, fl / best / 1e9, med, diffs[v]);
        }
        printf("\n");
        fflush(stdout);
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C0)); CK(hipFree(C1)); CK(hipFree(bias)); CK(hipFree(res)); CK(hipFree(cf)); CK(hipFree(dmax)); CK(hipFree(stats)); CK(hipFree(parts));
    }
    return 0;
}
