// gemm4d_ablate.hip — standalone harness for VERDICT r4 item 5a: the PRODUCT K loop of zett::gemm4d_tn_kernel (csrc/gemm4d.hip.h,
// compiled here with -DG4D_ABLATE=<level>) on RANDOM f16 operands, piece by piece:
//   level 1 = the MFMAs alone, 2 = + fragment reads, 3 = + LDS-DMA requests, 4 = + waits and barriers (the whole K loop),
//   level 5 = + the 16-bit-output epilogue (the product kernel).
// At every level the LDS stages hold the random tiles the prologue loaded, so the MFMAs toggle real data (the zero-operand table
// of NOTEBOOK R3 hid exactly the power coupling this is after).  Runs the launch back to back for <seconds> and prints one JSON
// line: TFLOP/s over HIP-event time.  tools/ablate.sh samples socket power / shader clock beside it and takes the PMC pass.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DG4D_ABLATE=4 -I zett_amd/csrc tools/gemm4d_ablate.hip -o tools/_ablate/gemm4d_ablate_L4
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "gemm.hip.h"
#include "gemm_tile.hip.h"
#include "gemm4d.hip.h"

using namespace zett;

// (the HALF instantiations are never launched here; the launcher template references them)
namespace zett {
hipError_t launch_gemm_4d_half(const GemmArgs<f16_t>&, hipStream_t, int) { return hipErrorInvalidValue; }
hipError_t launch_gemm_4d_half(const GemmArgs<bf16_t>&, hipStream_t, int) { return hipErrorInvalidValue; }
}

__global__ void fill_random_f16(f16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float u = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f;          // uniform in [-0.5, 0.5)
        p[i] = (f16_t)(u * 2.0f * scale);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 77450, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 4096;
    const double seconds = argc > 4 ? atof(argv[4]) : 5.0;
    const int zero = argc > 5 ? atoi(argv[5]) : 0;          // 1 = all-zero operands (the old table's condition)
    f16_t *A, *W, *C;
    float* bias;
    CK(hipMalloc((void**)&A, (size_t)M * K * 2));
    CK(hipMalloc((void**)&W, (size_t)N * K * 2));
    CK(hipMalloc((void**)&C, (size_t)M * N * 2));
    CK(hipMalloc((void**)&bias, (size_t)N * 4));
    CK(hipMemset(bias, 0, (size_t)N * 4));
    if (zero) { CK(hipMemset(A, 0, (size_t)M * K * 2)); CK(hipMemset(W, 0, (size_t)N * K * 2)); }
    else {
        hipLaunchKernelGGL(fill_random_f16, dim3(4096), dim3(256), 0, 0, A, (size_t)M * K, 1u, 1.0f);        // activations ~ U(-1, 1)
        hipLaunchKernelGGL(fill_random_f16, dim3(4096), dim3(256), 0, 0, W, (size_t)N * K, 7u, 0.05f);       // weights ~ U(-0.05, 0.05)
    }
    CK(hipDeviceSynchronize());
    GemmEpilogue<f16_t> e{};
    e.split_col = 0x7fffffff;
    e.bias = bias; e.out_lo = C; e.ld_lo = N;
    GemmArgs<f16_t> g{A, K, W, K, M, N, K, e};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK((launch_gemm4d_inst<f16_t, ACT_NONE, false, G4D_EPI_LO, false>(g, 0)));
    CK(hipDeviceSynchronize());
    double ms_total = 0.0; long launches = 0;
    const auto t_start = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < seconds) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) CK((launch_gemm4d_inst<f16_t, ACT_NONE, false, G4D_EPI_LO, false>(g, 0)));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms; launches += 20;
    }
    const double ms = ms_total / launches;
    printf("{\"level\": %d, \"operands\": \"%s\", \"m\": %d, \"n\": %d, \"k\": %d, \"launches\": %ld, \"ms_per_launch\": %.4f, \"tflops\": %.1f, \"frac_of_2500\": %.4f}\n",
           (int)G4D_ABLATE, zero ? "zero" : "random", M, N, K, launches, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, 2.0 * M * N * K / (ms * 1e-3) / 1e12 / 2500.0);
    return 0;
}
