// mfma_power.hip — which MFMA shape does more FLOP/s inside the power envelope?  Register-only loops of
// v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16 on random operands, all 256 CUs, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ u32x4 rnd(uint32_t s) {
    u32x4 v;
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; uint32_t lo = (s >> 9) & 0x7fff; uint32_t hi = (s >> 1) & 0x7fff0000;
        v[i] = (lo | hi | 0x3c003c00u) & 0xbfffbfffu ^ ((s & 1) << 15) ^ ((s & 2) << 30); }   // bf16 pairs of magnitude ~1, random mantissa and sign
    return v;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const uint32_t seed = blockIdx.x * 977 + threadIdx.x * 31;
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(seed + i); b[i] = rnd(seed * 7 + i); }
    float sum = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][7];
    } else {
        f32x4 acc[8][4];      // the same 128x64 wave tile as 8x4 tiles of 16x16: 32 MFMAs of half the FLOPs
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 512 * 4 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep)
        for (int shape : {32, 16}) for (int threads : {256, 512}) {
            auto launch = [&]() { if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(threads), 0, 0, out, iters); else hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, out, iters); };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); launch(); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 2;
            const double flops = 256.0 * (threads / 64) * iters * (shape == 32 ? 8 * 32768.0 : 32 * 16384.0);
            printf("mfma %dx%d, %d waves/CU: %.3f ms  %.0f TFLOP/s\n", shape, shape, threads / 64, ms, flops / ms / 1e9);
        }
    return 0;
}
