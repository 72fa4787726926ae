// idle_transient.hip — does a short idle gap change the rate of the GEMM launches that follow it?
// Per-launch HIP-event times of a train of identical gemm4d launches, with a host-side pause (stream idle) of IDLE_US
// microseconds before every train.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zett_amd/csrc tools/experiments/idle_transient.hip -o tools/idle_transient
//   tools/idle_transient M N K train idle_us [filler_us]     (filler: a streaming kernel of that length instead of idling)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm4d.hip.h"
using namespace zett;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(bf16_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; p[i] = f32_to_bf16(((float)(x & 0xffff) / 32768.f - 1.f) * 0.1f); }
}
__global__ void stream_copy(const float4* a, float4* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), train = atoi(argv[4]), idle_us = atoi(argv[5]);
    const int filler_us = argc > 6 ? atoi(argv[6]) : 0;
    bf16_t *A, *W, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    float4 *s0, *s1; const size_t sn = (size_t)64 << 20;      // 1 GiB each way
    CK(hipMalloc(&s0, sn * 16)); CK(hipMalloc(&s1, sn * 16));
    fill<<<2048, 256>>>(A, (size_t)M * K, 1); fill<<<2048, 256>>>(W, (size_t)N * K, 2);
    GemmArgs<bf16_t> g{}; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.epi.split_col = 0x7fffffff;
    g.epi.out_lo = C; g.epi.ld_lo = N;
    std::vector<hipEvent_t> ev(train + 1);
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (int i = 0; i < 30; ++i) CK(launch_gemm4d<bf16_t>(g, 0));      // warm
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        if (idle_us > 0) usleep(idle_us);
        if (filler_us > 0) {      // ~5 TB/s of copy traffic: 2 GiB moved takes ~400 us
            const int reps = filler_us / 400 + 1;
            for (int i = 0; i < reps; ++i) stream_copy<<<4096, 256>>>(s0, s1, sn);
        }
        for (int i = 0; i < train; ++i) { CK(hipEventRecord(ev[i], 0)); CK(launch_gemm4d<bf16_t>(g, 0)); }
        CK(hipEventRecord(ev[train], 0));
        CK(hipEventSynchronize(ev[train]));
        printf("idle %d us filler %d us:", idle_us, filler_us);
        for (int i = 0; i < train; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); printf(" %.0f", 2.0 * M * N * K / ms / 1e9); }
        printf("  TF per launch\n");
    }
    return 0;
}
