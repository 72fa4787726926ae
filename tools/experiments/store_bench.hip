// store_bench.hip — how fast can ONE CU issue the global stores / loads of a GEMM epilogue?
// 256-thread workgroups (one per CU when grid <= 256), every wave issues REPS 16-byte-per-lane accesses back to back.
// Patterns (what the 64 lanes of one instruction touch):
//   0: 4 rows x 256 B   (the bf16 drain of EpiDrain: 16 lanes per row)
//   1: 1 row  x 1 KiB   (fully contiguous)
//   2: 16 rows x 64 B   (a direct-from-accumulator epilogue: 4 lanes per row)
//   3: 2 rows x 512 B
//   4: 4 rows x 512 B with 16-B holes (the fp32 drain: lane stride 32 B)
// Flavours: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1.   op: 0 store, 1 load
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/store_bench.hip -o tools/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FL> __device__ __forceinline__ void st16(float* p, f32x4 v) {
    if (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int FL> __device__ __forceinline__ f32x4 ld16(const float* p) {
    f32x4 v;
    if (FL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if (FL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if (FL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// every workgroup owns a private region of `wg_bytes`; row pitch `ld` bytes
template <int PAT, int FL, int OP>
__global__ __launch_bounds__(256) void k(float* base, size_t wg_bytes, int ld, int reps, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wg = (char*)base + (size_t)blockIdx.x * wg_bytes;
    // lane's offset inside one instruction's footprint, and the footprint's row count
    int rows, off;
    if (PAT == 0) { rows = 4; off = (lane >> 4) * ld + (lane & 15) * 16; }
    else if (PAT == 1) { rows = 1; off = lane * 16; }
    else if (PAT == 2) { rows = 16; off = (lane & 15) * ld + (lane >> 4) * 16; }
    else if (PAT == 3) { rows = 2; off = (lane >> 5) * ld + (lane & 31) * 16; }
    else { rows = 4; off = (lane >> 4) * ld + (lane & 15) * 32; }
    // a wave walks down its own column block of the region: footprints stacked by rows, 4 waves side by side
    const int wave_cols = PAT == 1 ? 1024 : PAT == 3 ? 512 : PAT == 4 ? 512 : PAT == 2 ? 64 : 256;
    char* p = wg + wave * wave_cols + off;
    const size_t step = (size_t)rows * ld;
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        if (OP == 0) st16<FL>((float*)(p + r * step), v);
        else { f32x4 x = ld16<FL>((const float*)(p + r * step)); asm volatile("" :: "v"(x)); (void)acc; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int PAT, int FL, int OP>
void run(const char* name, float* buf, size_t wg_bytes, int ld, int reps, int grid, unsigned long long* clk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<PAT, FL, OP><<<grid, 256>>>(buf, wg_bytes, ld, reps, clk);
    CK(hipEventRecord(e0, 0));
    k<PAT, FL, OP><<<grid, 256>>>(buf, wg_bytes, ld, reps, clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, clk, sizeof(unsigned long long) * (grid < 256 ? grid : 256), hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < (grid < 256 ? grid : 256); ++i) mean += h[i];
    mean /= (grid < 256 ? grid : 256);
    const double bytes = 4.0 * reps * 1024;      // per workgroup
    printf("  %-28s grid %3d: %7.2f us per WG  -> %6.1f GB/s per CU, %5.0f ns per instruction per wave, chip %6.2f TB/s (event %.1f us)\n", name, grid, mean / 100.0,
           bytes / (mean * 10.0), mean * 10.0 / reps, bytes * grid / (mean * 10.0) / 1e3, ms * 1e3);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 256;          // 256 instructions per wave = 1 MiB per workgroup
    const int ld = argc > 2 ? atoi(argv[2]) : 8192;           // row pitch in bytes (N = 4096 bf16)
    const size_t wg_bytes = (size_t)16 * reps * ld + 65536;   // worst case: 16 rows per instruction
    float* buf; CK(hipMalloc(&buf, wg_bytes * 256));
    CK(hipMemset(buf, 0, wg_bytes * 256));
    unsigned long long* clk; CK(hipMalloc(&clk, 256 * 8));
    for (int grid : {8, 64, 256}) {
        printf("grid %d\n", grid);
        run<0, 0, 0>("store 4x256B plain", buf, wg_bytes, ld, reps, grid, clk);
        run<0, 1, 0>("store 4x256B nt", buf, wg_bytes, ld, reps, grid, clk);
        run<0, 2, 0>("store 4x256B sc1", buf, wg_bytes, ld, reps, grid, clk);
        run<0, 3, 0>("store 4x256B sc0sc1", buf, wg_bytes, ld, reps, grid, clk);
        run<1, 0, 0>("store 1x1KiB plain", buf, wg_bytes, ld, reps, grid, clk);
        run<1, 1, 0>("store 1x1KiB nt", buf, wg_bytes, ld, reps, grid, clk);
        run<3, 0, 0>("store 2x512B plain", buf, wg_bytes, ld, reps, grid, clk);
        run<2, 0, 0>("store 16x64B plain", buf, wg_bytes, ld, reps, grid, clk);
        run<2, 1, 0>("store 16x64B nt", buf, wg_bytes, ld, reps, grid, clk);
        run<4, 0, 0>("store 4x512B holes plain", buf, wg_bytes, ld, reps, grid, clk);
        run<4, 1, 0>("store 4x512B holes nt", buf, wg_bytes, ld, reps, grid, clk);
        run<0, 0, 1>("load 4x256B plain", buf, wg_bytes, ld, reps, grid, clk);
        run<0, 1, 1>("load 4x256B nt", buf, wg_bytes, ld, reps, grid, clk);
        run<1, 0, 1>("load 1x1KiB plain", buf, wg_bytes, ld, reps, grid, clk);
        run<2, 0, 1>("load 16x64B plain", buf, wg_bytes, ld, reps, grid, clk);
        run<4, 0, 1>("load 4x512B holes plain", buf, wg_bytes, ld, reps, grid, clk);
    }
    return 0;
}
