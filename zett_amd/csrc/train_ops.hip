// train_ops.hip — C ABI of the training-time primitives (include/zett_hip.h, "training use"): what
// zett_amd/autograd.py composes the differentiable forward and its backward from (reference call
// sites train.py:1007-1013, 1191-1197: the hypernetwork forward inside the loss of a training / evaluation step).
//
// First slice: fp32 arithmetic, the as-written (dense [N, L', H]) layout of the reference.  The dense contractions —
// forward, dgrad and wgrad alike — go through the library's TN GEMM family (fp32 MFMA, gemm_launch.hip.h):
//     forward   y[M,N]  = x[M,K]  · W[N,K]ᵀ                     (A = x,    W-operand = W)
//     dgrad     dx[M,K] = dy[M,N] · (Wᵀ)[K,N]ᵀ                  (A = dy,   W-operand = Wᵀ: zett_op_transpose)
//     wgrad     dW[N,K] = (dyᵀ)[N,M] · (xᵀ)[K,M]ᵀ               (A = dyᵀ,  W-operand = xᵀ; M zero-padded to the K step)
// — the same contraction with swapped / transposed operands, so one kernel family serves all three.  Everything else is a
// row kernel here: LayerNorm forward / backward, the two GELUs and their derivatives, dense masked attention forward /
// backward (eager semantics: finfo.min on masked keys, uniform rows when every key is masked), the source-embedding
// gather with in_scaler / fallback select and its backward, column sums, transposes and three element-wise forms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>

#include "../../include/zett_hip.h"
#include "common.hip.h"
#include "gemm.hip.h"
#include "gemm_launch.hip.h"

using namespace zett;

namespace {

__device__ __forceinline__ float t_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float t_block_sum(float v, float* red /* [4] */) {      // 256 threads
    v = t_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- transposes / reductions / element-wise ----------------------------------------------------------------------
// out[c, r] = in[r, c] for r < R; out[c, r] = 0 for R <= r < Rpad   (ld_out >= Rpad)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int ld_out,
                                                        int R, int C, int Rpad) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < C && r < Rpad) out[(size_t)c * ld_out + r] = tile[tx][k];
    }
}

// 16-bit operands of the training GEMMs: out[r, c] = lo(in[r, c]) (columns zero-padded to cols_padded), and the transposed
// form out[c, r] = lo(in[r, c]) (rows zero-padded to rows_padded): conversion fused with the layout change
template <typename T>
__global__ void convert_lo_kernel(const float* __restrict__ in, int ld_in, T* __restrict__ out, int ld_out, int64_t rows, int cols, int cols_padded) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = rows * cols_padded, stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int64_t r = i / cols_padded;
        const int c = (int)(i % cols_padded);
        out[r * ld_out + c] = to_lo<T>(c < cols ? in[r * ld_in + c] : 0.f);
    }
}
// the common case (no column padding, widths multiples of 4): 16-byte loads, 8-byte stores, one row per workgroup pass
template <typename T>
__global__ __launch_bounds__(256) void convert_lo4_kernel(const float* __restrict__ in, int ld_in, T* __restrict__ out, int ld_out, int64_t rows, int cols) {
    const int c4 = cols >> 2;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float4* src = (const float4*)(in + r * ld_in);
        uint2* dst = (uint2*)(out + r * ld_out);
        for (int c = threadIdx.x; c < c4; c += 256) {
            const float4 v = src[c];
            dst[c] = make_uint2(pack2_lo<T>(v.x, v.y), pack2_lo<T>(v.z, v.w));
        }
    }
}
__device__ __forceinline__ float gelu_fwd1(float x, int kind) { return kind == 1 ? gelu_tanh_f(x) : gelu_erf_f(x); }
__device__ __forceinline__ float gelu_grad1(float x, int kind) {
    if (kind == 1) {       // d/dx [0.5 x (1 + tanh u)], u = sqrt(2/pi) (x + 0.044715 x^3)
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        const float t = tanhf(u);
        return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
    }
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);       // Phi(x) + x phi(x)
}

// four consecutive values of a row as floats: fp32 (16-byte load) or 16-bit (8-byte load) storage; element-wise tail at the edges
template <typename TIn>
__device__ __forceinline__ float4 load4(const TIn* __restrict__ row, int c, int C, bool vec_ok);
template <>
__device__ __forceinline__ float4 load4<float>(const float* __restrict__ row, int c, int C, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < C && vec_ok) return *(const float4*)(row + c);
    if (c < C) v.x = row[c];
    if (c + 1 < C) v.y = row[c + 1];
    if (c + 2 < C) v.z = row[c + 2];
    if (c + 3 < C) v.w = row[c + 3];
    return v;
}
template <typename T>
__device__ __forceinline__ float4 load4_lo(const T* __restrict__ row, int c, int C, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < C && vec_ok) {
        const uint2 u = *(const uint2*)(row + c);
        unpack2_lo<T>(u.x, v.x, v.y);
        unpack2_lo<T>(u.y, v.z, v.w);
        return v;
    }
    if (c < C) v.x = lo_to_f32<T>(row[c]);
    if (c + 1 < C) v.y = lo_to_f32<T>(row[c + 1]);
    if (c + 2 < C) v.z = lo_to_f32<T>(row[c + 2]);
    if (c + 3 < C) v.w = lo_to_f32<T>(row[c + 3]);
    return v;
}
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* __restrict__ row, int c, int C, bool vec_ok) { return load4_lo<bf16_t>(row, c, C, vec_ok); }
template <> __device__ __forceinline__ float4 load4<f16_t>(const f16_t* __restrict__ row, int c, int C, bool vec_ok) { return load4_lo<f16_t>(row, c, C, vec_ok); }

// 64 x 64 tiles through LDS: wide loads along the input rows (fp32 or, TIn = T, an operand that is already 16-bit), 8-byte stores
// (four converted values) along the output rows.
// One read of a gradient serves everything a Linear's backward needs from it: `plain` (nullable) gets the un-transposed 16-bit
// copy (dgrad's A operand), `colpart` (nullable, [ceil(R / 64), C]) the column sums of each 64-row band (their sum is the bias
// gradient: a deterministic two-level reduction, no atomics).  With `act_z` (nullable: the pre-activation of a GELU whose OUTPUT
// gradient `in` is) the value used everywhere is in * gelu'(act_z): the activation's backward costs no pass of its own and its
// fp32 result is never written.
template <typename T, typename TIn>
__global__ __launch_bounds__(256) void transpose_lo_kernel(const TIn* __restrict__ in, int ld_in, T* __restrict__ out, int ld_out, int R, int C, int Rpad,
                                                           T* __restrict__ plain, int ld_plain, float* __restrict__ colpart,
                                                           const float* __restrict__ act_z, int ld_z, int act_kind) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    const bool in_vec = (ld_in & 3) == 0, z_vec = (ld_z & 3) == 0;
    for (int k = t; k < 64 * 16; k += 256) {           // 64 rows x 16 groups of four columns
        const int rr = k >> 4, cc = (k & 15) << 2;
        const int r = r0 + rr, c = c0 + cc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) {
            v = load4<TIn>(in + (size_t)r * ld_in, c, C, in_vec);
            if (act_z) {
                const float4 z = load4<float>(act_z + (size_t)r * ld_z, c, C, z_vec);
                v.x *= gelu_grad1(z.x, act_kind); v.y *= gelu_grad1(z.y, act_kind); v.z *= gelu_grad1(z.z, act_kind); v.w *= gelu_grad1(z.w, act_kind);
            }
            if (plain) {
                if (c + 3 < C && ((ld_plain & 3) == 0)) *(uint2*)(plain + (size_t)r * ld_plain + c) = make_uint2(pack2_lo<T>(v.x, v.y), pack2_lo<T>(v.z, v.w));
                else {
                    if (c < C) plain[(size_t)r * ld_plain + c] = to_lo<T>(v.x);
                    if (c + 1 < C) plain[(size_t)r * ld_plain + c + 1] = to_lo<T>(v.y);
                    if (c + 2 < C) plain[(size_t)r * ld_plain + c + 2] = to_lo<T>(v.z);
                    if (c + 3 < C) plain[(size_t)r * ld_plain + c + 3] = to_lo<T>(v.w);
                }
            }
        }
        tile[rr][cc] = v.x; tile[rr][cc + 1] = v.y; tile[rr][cc + 2] = v.z; tile[rr][cc + 3] = v.w;
    }
    __syncthreads();
    if (colpart && r0 < R && t < 64 && c0 + t < C) {   // (bands past R exist only as zero padding of the transposed output)
        float s = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < 64; ++rr) s += tile[rr][t];
        colpart[(size_t)blockIdx.y * C + c0 + t] = s;
    }
    for (int k = t; k < 64 * 16; k += 256) {           // 64 output rows (= input columns) x 16 groups of four input rows
        const int cc = k >> 4, rr = (k & 15) << 2;
        const int c = c0 + cc, r = r0 + rr;
        if (c >= C || r >= Rpad) continue;
        if (r + 3 < Rpad && ((ld_out & 3) == 0))
            *(uint2*)(out + (size_t)c * ld_out + r) = make_uint2(pack2_lo<T>(tile[rr][cc], tile[rr + 1][cc]), pack2_lo<T>(tile[rr + 2][cc], tile[rr + 3][cc]));
        else
            for (int j = 0; j < 4 && r + j < Rpad; ++j) out[(size_t)c * ld_out + r + j] = to_lo<T>(tile[rr + j][cc]);
    }
}

// out[c] (+)= sum_r in[r, c]: one workgroup per 64 columns, rows strided over the four waves
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int ld, int R, int C, float* __restrict__ out, int accumulate) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int r = w; r < R; r += 4) s += in[(size_t)r * ld + c];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// op 0: out = a + b;  1: out = a * b;  2: out = a * vec[col] + vec2[col] (vec null: 1, vec2 null: 0);  3: out = a + s[row] * vec[col];
// 4: out = a * s[row] (a may be null: out = s[row] * vec[col]).  One thread per four consecutive elements of a row (cols % 4 == 0:
// 16-byte accesses, one division per four values) or per element (V = 1).
template <int V>
__global__ __launch_bounds__(256) void elementwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ vec,
                                                          const float* __restrict__ vec2, const float* __restrict__ s, float* __restrict__ out, int64_t n, int cols) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (; i < n; i += stride) {
        float av[V], bv[V], ov[V];
        if (V == 4) {
            if (a) *(float4*)av = *(const float4*)(a + i);
            if (b) *(float4*)bv = *(const float4*)(b + i);
        } else {
            if (a) av[0] = a[i];
            if (b) bv[0] = b[i];
        }
        int col = 0;
        int64_t row = 0;
        if (op >= 2) { row = i / cols; col = (int)(i - row * cols); }
        const float sr = (op >= 3) ? s[row] : 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            switch (op) {
                case 0: ov[e] = av[e] + bv[e]; break;
                case 1: ov[e] = av[e] * bv[e]; break;
                case 2: ov[e] = (vec ? av[e] * vec[col + e] : av[e]) + (vec2 ? vec2[col + e] : 0.f); break;
                case 3: ov[e] = av[e] + sr * vec[col + e]; break;
                default: ov[e] = a ? av[e] * sr : sr * vec[col + e]; break;
            }
        }
        if (V == 4) *(float4*)(out + i) = *(float4*)ov;
        else out[i] = ov[0];
    }
}

// out[r] = sum_c a[r, c] * w[c] + (b ? b[0] : 0)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, int ld, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ out, int R, int C) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    if (r >= R) return;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += a[(size_t)r * ld + c] * w[c];
    const float t = t_block_sum(s, red);
    if (threadIdx.x == 0) out[r] = t + (b ? b[0] : 0.f);
}

// ---- LayerNorm ------------------------------------------------------------------------------------------------------
// y = (x - mean) * rstd * gamma + beta; stats[r] = (mean, rstd)   (two-pass variance, as torch.nn.LayerNorm).  A row is read
// once, into registers: a thread owns columns 4 (tid + 256 j) ... + 3, j < J (H <= 1024 J, H % 4 == 0).  y_lo (nullable): the
// same values as a 16-bit operand of the next contraction, written while they are in registers.
template <int J, typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ y, float* __restrict__ stats, int R, int H, T* __restrict__ y_lo) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        float4 v[J];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = 4 * (tid + 256 * j);
            v[j] = c < H ? *(const float4*)(x + (size_t)r * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = t_block_sum(s, red) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (4 * (tid + 256 * j) < H) {
                const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(t_block_sum(q, red) / (float)H + eps);
        if (tid == 0) { stats[2 * (size_t)r] = mean; stats[2 * (size_t)r + 1] = rstd; }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = 4 * (tid + 256 * j);
            if (c < H) {
                const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
                const float4 o = make_float4(ln_affine(v[j].x, mean, rstd, g.x, b.x), ln_affine(v[j].y, mean, rstd, g.y, b.y),
                                             ln_affine(v[j].z, mean, rstd, g.z, b.z), ln_affine(v[j].w, mean, rstd, g.w, b.w));
                *(float4*)(y + (size_t)r * H + c) = o;
                if (y_lo) *(uint2*)(y_lo + (size_t)r * H + c) = make_uint2(pack2_lo<T>(o.x, o.y), pack2_lo<T>(o.z, o.w));      // the next contraction's operand
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd, dy = dy1 (+ dy2: the gradient
// arriving over the residual branch, added here instead of in a pass of its own).  The parameter gradients ride along:
// workgroup b walks rows b, b + gridDim.x, ... and keeps, per thread, the running sums of dy * xhat (-> dgamma) and dy (-> dbeta)
// of its columns; they leave as partials[b] = (dgamma part [H], dbeta part [H]) and one small column sum over the gridDim.x
// partials finishes them — the same result for the same grid, no atomics, and no [R, H] product written or read.
// A thread owns columns 4 (tid + 256 j) ... + 3, j < J (H <= 1024 J, H % 4 == 0); a row is read once, into registers.
template <int J>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy1, const float* __restrict__ dy2, const float* __restrict__ x, int ld,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma, float* __restrict__ dx,
                                                     float* __restrict__ partials, int R, int H) {
    __shared__ float red[2][2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 gam[J], ag[J], ab[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = 4 * (tid + 256 * j);
        gam[j] = c < H ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int par = 0;
    for (int r = blockIdx.x; r < R; r += gridDim.x, par ^= 1) {
        const float mean = stats[2 * (size_t)r], rstd = stats[2 * (size_t)r + 1];
        float4 xh[J], d[J];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = 4 * (tid + 256 * j);
            xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            d[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < H) {
                const float4 xv = *(const float4*)(x + (size_t)r * ld + c);
                d[j] = *(const float4*)(dy1 + (size_t)r * H + c);
                if (dy2) {
                    const float4 e = *(const float4*)(dy2 + (size_t)r * H + c);
                    d[j].x += e.x; d[j].y += e.y; d[j].z += e.z; d[j].w += e.w;
                }
                xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                const float4 g = make_float4(d[j].x * gam[j].x, d[j].y * gam[j].y, d[j].z * gam[j].z, d[j].w * gam[j].w);
                sg += (g.x + g.y) + (g.z + g.w);
                sgx += (g.x * xh[j].x + g.y * xh[j].y) + (g.z * xh[j].z + g.w * xh[j].w);
            }
        }
        sg = t_wave_sum(sg);
        sgx = t_wave_sum(sgx);
        if (lane == 0) { red[par][0][wave] = sg; red[par][1][wave] = sgx; }
        __syncthreads();                               // (the other parity's slots are rewritten only after the next barrier)
        const float mg = ((red[par][0][0] + red[par][0][1]) + (red[par][0][2] + red[par][0][3])) / (float)H;
        const float mgx = ((red[par][1][0] + red[par][1][1]) + (red[par][1][2] + red[par][1][3])) / (float)H;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = 4 * (tid + 256 * j);
            if (c < H) {
                float4 o;
                o.x = rstd * (d[j].x * gam[j].x - mg - xh[j].x * mgx);
                o.y = rstd * (d[j].y * gam[j].y - mg - xh[j].y * mgx);
                o.z = rstd * (d[j].z * gam[j].z - mg - xh[j].z * mgx);
                o.w = rstd * (d[j].w * gam[j].w - mg - xh[j].w * mgx);
                *(float4*)(dx + (size_t)r * H + c) = o;
                ag[j].x += d[j].x * xh[j].x; ag[j].y += d[j].y * xh[j].y; ag[j].z += d[j].z * xh[j].z; ag[j].w += d[j].w * xh[j].w;
                ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = 4 * (tid + 256 * j);
        if (c < H) {
            *(float4*)(partials + ((size_t)blockIdx.x * 2) * H + c) = ag[j];
            *(float4*)(partials + ((size_t)blockIdx.x * 2 + 1) * H + c) = ab[j];
        }
    }
}

// ---- GELU -----------------------------------------------------------------------------------------------------------
// kind 1: F.gelu(approximate="tanh") (ProjectorBlock), 2: erf form (RobertaIntermediate) — the forward functions of gemm.hip.h
// V = 4: 16-byte accesses (n % 4 == 0, aligned), V = 1 otherwise
template <int V>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ z, float* __restrict__ h, int64_t n, int kind) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (; i < n; i += stride) {
        if (V == 4) {
            const float4 v = *(const float4*)(z + i);
            *(float4*)(h + i) = make_float4(gelu_fwd1(v.x, kind), gelu_fwd1(v.y, kind), gelu_fwd1(v.z, kind), gelu_fwd1(v.w, kind));
        } else {
            h[i] = gelu_fwd1(z[i], kind);
        }
    }
}
template <int V>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dh, float* __restrict__ dz, int64_t n, int kind) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (; i < n; i += stride) {
        if (V == 4) {
            const float4 v = *(const float4*)(z + i), g = *(const float4*)(dh + i);
            *(float4*)(dz + i) = make_float4(g.x * gelu_grad1(v.x, kind), g.y * gelu_grad1(v.y, kind), g.z * gelu_grad1(v.z, kind), g.w * gelu_grad1(v.w, kind));
        } else {
            dz[i] = dh[i] * gelu_grad1(z[i], kind);
        }
    }
}

// the 16-bit form of the forward: the activation as the next contraction's operand (its fp32 value is never stored)
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_lo_kernel(const float* __restrict__ z, T* __restrict__ h, int64_t n, int kind) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *(const float4*)(z + i);
            *(uint2*)(h + i) = make_uint2(pack2_lo<T>(gelu_fwd1(v.x, kind), gelu_fwd1(v.y, kind)), pack2_lo<T>(gelu_fwd1(v.z, kind), gelu_fwd1(v.w, kind)));
        } else {
            for (int64_t j = i; j < n; ++j) h[j] = to_lo<T>(gelu_fwd1(z[j], kind));
        }
    }
}

// ---- masked attention (eager semantics), dense or packed ------------------------------------------------------------------
// One workgroup of 64 lanes per (vocabulary row n, head).  The positions of row n are rows [t0, t1) of k / v (and of q / ctx
// unless cls_only): t0 = row_offset[n], t1 = row_offset[n + 1] (packed: only the positions the row keeps), or n * seq and
// (n + 1) * seq when row_offset is null (the reference's dense layout).  mask[t] = 1: position t is visible as a key.
// cls_only: the query is position 0 only, q and ctx hold ONE row per vocabulary row (the position-0-only last layer).
// probs [n_rows, heads, seq, seq] (row-major in the padded seq) is kept for the backward.  t1 - t0 <= seq <= ATT_MAX_L.
constexpr int ATT_MAX_L = 32;

// One WAVE per (vocabulary row, head), four heads per workgroup; no LDS, no barrier.  A lane owns head-dim columns lane + 64 e
// (e < DV: head dims up to 64 DV); the row's keys and values live in registers (LMAX positions, loops over positions fully
// unrolled with wave-uniform guards, so the arrays are never indexed dynamically), every load of the row is in flight before
// the first reduction, and each query position is one pass: LMAX dot products (butterfly sums leave every lane with the
// result), the softmax computed redundantly by all lanes, the weighted sum of the values.
template <int LMAX, int DV>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                       const uint8_t* __restrict__ mask, const int32_t* __restrict__ row_offset, int seq, int heads, int d,
                                                       float scaling, int cls_only, float* __restrict__ ctx, int ld_ctx, float* __restrict__ probs,
                                                       void* __restrict__ ctx_lo, int lo_kind) {
    const int groups = (heads + 3) >> 2;
    const int n = blockIdx.x / groups, hd = (blockIdx.x % groups) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (hd >= heads) return;
    const size_t base = row_offset ? (size_t)row_offset[n] : (size_t)n * seq;
    const int L = row_offset ? row_offset[n + 1] - row_offset[n] : seq;
    const int nq = cls_only ? 1 : L;
    const size_t qbase = cls_only ? (size_t)n : base;
    float kr[LMAX][DV], vr[LMAX][DV], bias[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        bias[j] = 0.f;
#pragma unroll
        for (int e = 0; e < DV; ++e) { kr[j][e] = 0.f; vr[j][e] = 0.f; }
        if (j < L) {
            bias[j] = mask[base + j] ? 0.f : -FLT_MAX;             // finfo(float32).min on masked keys
#pragma unroll
            for (int e = 0; e < DV; ++e) {
                const int c = lane + 64 * e;
                if (c < d) { kr[j][e] = k[(base + j) * ld + hd * d + c]; vr[j][e] = v[(base + j) * ld + hd * d + c]; }
            }
        }
    }
    for (int i = 0; i < nq; ++i) {
        float qr[DV];
#pragma unroll
        for (int e = 0; e < DV; ++e) { const int c = lane + 64 * e; qr[e] = c < d ? q[(qbase + i) * ldq + hd * d + c] : 0.f; }
        float sc[LMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < LMAX; ++j) {
            sc[j] = 0.f;
            if (j < L) {
                float p = 0.f;
#pragma unroll
                for (int e = 0; e < DV; ++e) p += qr[e] * kr[j][e];
                sc[j] = t_wave_sum(p) * scaling + bias[j];
                mx = fmaxf(mx, sc[j]);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
        float acc[DV];
#pragma unroll
        for (int e = 0; e < DV; ++e) acc[e] = 0.f;
        float* prow = probs + (((size_t)n * heads + hd) * seq + i) * seq;
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) {
                const float p = sc[j] / sum;
                if (lane == 0) prow[j] = p;
#pragma unroll
                for (int e = 0; e < DV; ++e) acc[e] += p * vr[j][e];
            }
#pragma unroll
        for (int e = 0; e < DV; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const size_t o = (qbase + i) * ld_ctx + hd * d + c;
                // lo_kind != 0: the context only feeds a contraction (and its transposed twin in the backward): written as that 16-bit
                // operand, its fp32 form is never stored
                if (lo_kind == 0) ctx[o] = acc[e];
                else if (lo_kind == 1) ((bf16_t*)ctx_lo)[o] = f32_to_bf16(acc[e]);
                else ((f16_t*)ctx_lo)[o] = (f16_t)acc[e];
            }
        }
    }
}

// dv_j = sum_i p_ij dctx_i;  dp_ij = dctx_i . v_j;  ds_ij = p_ij (dp_ij - sum_j' p_ij' dp_ij');
// dq_i = scaling sum_j ds_ij k_j;  dk_j = scaling sum_i ds_ij q_i.   Same layout as the forward: keys, values and the running
// dk / dv of the row in registers, one pass per query position.
template <int LMAX, int DV>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ dctx, int ld_ctx, const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                       const float* __restrict__ v, int ld, const float* __restrict__ probs, const int32_t* __restrict__ row_offset,
                                                       int seq, int heads, int d, float scaling, int cls_only, float* __restrict__ dq, int ld_dq,
                                                       float* __restrict__ dk, float* __restrict__ dv, int ld_d) {
    const int groups = (heads + 3) >> 2;
    const int n = blockIdx.x / groups, hd = (blockIdx.x % groups) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (hd >= heads) return;
    const size_t base = row_offset ? (size_t)row_offset[n] : (size_t)n * seq;
    const int L = row_offset ? row_offset[n + 1] - row_offset[n] : seq;
    const int nq = cls_only ? 1 : L;
    const size_t qbase = cls_only ? (size_t)n : base;
    float kr[LMAX][DV], vr[LMAX][DV], dkr[LMAX][DV], dvr[LMAX][DV];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
#pragma unroll
        for (int e = 0; e < DV; ++e) { kr[j][e] = 0.f; vr[j][e] = 0.f; dkr[j][e] = 0.f; dvr[j][e] = 0.f; }
        if (j < L) {
#pragma unroll
            for (int e = 0; e < DV; ++e) {
                const int c = lane + 64 * e;
                if (c < d) { kr[j][e] = k[(base + j) * ld + hd * d + c]; vr[j][e] = v[(base + j) * ld + hd * d + c]; }
            }
        }
    }
    for (int i = 0; i < nq; ++i) {
        float qr[DV], gr[DV];
#pragma unroll
        for (int e = 0; e < DV; ++e) {
            const int c = lane + 64 * e;
            qr[e] = c < d ? q[(qbase + i) * ldq + hd * d + c] : 0.f;
            gr[e] = c < d ? dctx[(qbase + i) * ld_ctx + hd * d + c] : 0.f;
        }
        const float* prow = probs + (((size_t)n * heads + hd) * seq + i) * seq;
        float pr[LMAX], dp[LMAX];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < LMAX; ++j) {
            pr[j] = 0.f; dp[j] = 0.f;
            if (j < L) {
                pr[j] = prow[j];
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < DV; ++e) a += gr[e] * vr[j][e];
                dp[j] = t_wave_sum(a);
                dot += pr[j] * dp[j];
            }
        }
        float aq[DV];
#pragma unroll
        for (int e = 0; e < DV; ++e) aq[e] = 0.f;
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) {
                const float dsv = pr[j] * (dp[j] - dot);
#pragma unroll
                for (int e = 0; e < DV; ++e) {
                    aq[e] += dsv * kr[j][e];
                    dkr[j][e] += dsv * qr[e];
                    dvr[j][e] += pr[j] * gr[e];
                }
            }
#pragma unroll
        for (int e = 0; e < DV; ++e) { const int c = lane + 64 * e; if (c < d) dq[(qbase + i) * ld_dq + hd * d + c] = aq[e] * scaling; }
    }
#pragma unroll
    for (int j = 0; j < LMAX; ++j)
        if (j < L) {
#pragma unroll
            for (int e = 0; e < DV; ++e) {
                const int c = lane + 64 * e;
                if (c < d) { dk[(base + j) * ld_d + hd * d + c] = dkr[j][e] * scaling; dv[(base + j) * ld_d + hd * d + c] = dvr[j][e]; }
            }
        }
}

// ---- indexed rows: out[r] = (a ? a[r] : 0) + src[idx[r]];  dst[idx[r]] += src[r] -------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ a, const float* __restrict__ src, int ld_src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int cols) {
    const size_t r = blockIdx.x;
    const float* s = src + (size_t)idx[r] * ld_src;
    for (int c = threadIdx.x; c < cols; c += 256) out[r * cols + c] = (a ? a[r * cols + c] : 0.f) + s[c];
}
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ dst, int ld_dst, const int32_t* __restrict__ idx, const float* __restrict__ src,
                                                               int cols) {
    const size_t r = blockIdx.x;
    float* d = dst + (size_t)idx[r] * ld_dst;
    for (int c = threadIdx.x; c < cols; c += 256) atomicAdd(d + c, src[r * cols + c]);
}

// ---- source-embedding gather (A2 + A3) and its backward -----------------------------------------------------------------
template <int SD> __device__ __forceinline__ float load_src1(const void* base, size_t e);
template <> __device__ __forceinline__ float load_src1<0>(const void* base, size_t e) { return ((const float*)base)[e]; }
template <> __device__ __forceinline__ float load_src1<1>(const void* base, size_t e) { return (float)((const _Float16*)base)[e]; }
template <> __device__ __forceinline__ float load_src1<2>(const void* base, size_t e) { return __uint_as_float(((uint32_t)((const uint16_t*)base)[e]) << 16); }

// x[t] = id < V0 ? sw * src[id] + sb : fallback[id - V0]
template <int SD>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const int32_t* __restrict__ ids, int64_t T, const void* __restrict__ src, int e_in, int v0,
                                                         const float* __restrict__ fallback, const float* __restrict__ sw, const float* __restrict__ sb,
                                                         float* __restrict__ x) {
    const int64_t t = blockIdx.x;
    if (t >= T) return;
    const int id = ids[t];
    for (int c = threadIdx.x; c < e_in; c += 256) {
        float v;
        if (id >= v0) v = fallback[(size_t)(id - v0) * e_in + c];
        else { v = load_src1<SD>(src, (size_t)id * e_in + c); if (sw) v = sw[c] * v + sb[c]; }
        x[(size_t)t * e_in + c] = v;
    }
}
// dfallback[id - V0] += dx[t] (atomics: few rows); prod[t] = dx[t] * src[id] and keep[t] = dx[t] for source rows, 0 for fallback
// rows — their column sums are d in_scaler.w and d in_scaler.b
template <int SD>
__global__ __launch_bounds__(256) void gather_bwd_kernel(const int32_t* __restrict__ ids, int64_t T, const void* __restrict__ src, int e_in, int v0,
                                                         const float* __restrict__ dx, float* __restrict__ dfallback, float* __restrict__ prod,
                                                         float* __restrict__ keep) {
    const int64_t t = blockIdx.x;
    if (t >= T) return;
    const int id = ids[t];
    for (int c = threadIdx.x; c < e_in; c += 256) {
        const float g = dx[(size_t)t * e_in + c];
        if (id >= v0) {
            atomicAdd(dfallback + (size_t)(id - v0) * e_in + c, g);
            prod[(size_t)t * e_in + c] = 0.f; keep[(size_t)t * e_in + c] = 0.f;
        } else {
            prod[(size_t)t * e_in + c] = g * load_src1<SD>(src, (size_t)id * e_in + c);
            keep[(size_t)t * e_in + c] = g;
        }
    }
}

int grid_for(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 65535); }

}  // namespace

// 16-bit MFMA operands, fp32 accumulate and output: the tile choice of the inference path (gemm4d from K = 512, gemm8r below,
// 128x128 for small or unaligned outputs); prec = ZETT_PREC_BF16 | ZETT_PREC_F16
template <typename T>
static int gemm_lo(const T* a, int lda, const T* w, int ldw, int64_t m, int n, int k, const float* bias, int act, const float* residual, int ld_res,
                   float* out, int ld_out, hipStream_t st) {
    GemmArgs<T> g{};
    g.A = a; g.lda = lda; g.W = w; g.ldw = ldw; g.M = (int)m; g.N = n; g.K = k;
    g.epi.split_col = 0x7fffffff;
    g.epi.bias = bias; g.epi.act = act; g.epi.residual = residual; g.epi.ld_res = ld_res; g.epi.out_f32 = out; g.epi.ld_f32 = ld_out;
    const bool wide_ok = n % 8 == 0 && ld_out % 4 == 0 && (!residual || ld_res % 4 == 0);
    const int variant = (m > 128 && n > 128 && wide_ok) ? (k >= 512 ? 7 : 2) : 1;
    const hipError_t e = launch_gemm_variant(variant, g, st);
    if (e != hipSuccess) return fail(ZETT_E_HIP, "gemm launch failed: %s", hipGetErrorString(e));
    return 0;
}

template <typename T, typename TIn>
static void transpose_lo_go(const void* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* plain, int32_t ld_plain,
                            float* colpart, const float* act_z, int32_t ld_z, int32_t act_kind, hipStream_t st) {
    const dim3 grid((cols + 63) / 64, (unsigned)((rows_padded + 63) / 64));
    hipLaunchKernelGGL((transpose_lo_kernel<T, TIn>), grid, dim3(256), 0, st, (const TIn*)in, ld_in, (T*)out, ld_out, (int)rows, cols, (int)rows_padded, (T*)plain, ld_plain,
                       colpart, act_z, ld_z, act_kind);
}

template <int J>
static void ln_fwd_go(int32_t prec, dim3 grid, hipStream_t st, const float* x, int32_t ld, const float* gamma, const float* beta, float eps, float* y, float* stats,
                      int rows, int32_t h, void* y_lo) {
    if (prec == ZETT_PREC_F16) hipLaunchKernelGGL((ln_fwd_kernel<J, f16_t>), grid, dim3(256), 0, st, x, ld, gamma, beta, eps, y, stats, rows, h, (f16_t*)y_lo);
    else hipLaunchKernelGGL((ln_fwd_kernel<J, bf16_t>), grid, dim3(256), 0, st, x, ld, gamma, beta, eps, y, stats, rows, h, (bf16_t*)y_lo);
}

// the register layout of the attention kernels: positions rounded up to 2 / 4 / 8 / 16 / 32, 64-column slices of the head dim
template <int DV, typename... Args>
static void attn_fwd_go(int lmax, dim3 grid, hipStream_t st, Args... a) {
    if (lmax <= 2) hipLaunchKernelGGL((attn_fwd_kernel<2, DV>), grid, dim3(256), 0, st, a...);
    else if (lmax <= 4) hipLaunchKernelGGL((attn_fwd_kernel<4, DV>), grid, dim3(256), 0, st, a...);
    else if (lmax <= 8) hipLaunchKernelGGL((attn_fwd_kernel<8, DV>), grid, dim3(256), 0, st, a...);
    else if (lmax <= 16) hipLaunchKernelGGL((attn_fwd_kernel<16, DV>), grid, dim3(256), 0, st, a...);
    else hipLaunchKernelGGL((attn_fwd_kernel<32, DV>), grid, dim3(256), 0, st, a...);
}
template <int DV, typename... Args>
static void attn_bwd_go(int lmax, dim3 grid, hipStream_t st, Args... a) {
    if (lmax <= 2) hipLaunchKernelGGL((attn_bwd_kernel<2, DV>), grid, dim3(256), 0, st, a...);
    else if (lmax <= 4) hipLaunchKernelGGL((attn_bwd_kernel<4, DV>), grid, dim3(256), 0, st, a...);
    else if (lmax <= 8) hipLaunchKernelGGL((attn_bwd_kernel<8, DV>), grid, dim3(256), 0, st, a...);
    else if (lmax <= 16) hipLaunchKernelGGL((attn_bwd_kernel<16, DV>), grid, dim3(256), 0, st, a...);
    else if constexpr (DV < 4) hipLaunchKernelGGL((attn_bwd_kernel<32, DV>), grid, dim3(256), 0, st, a...);      // (DV = 4 would not fit its registers: refused by the caller)
}

extern "C" {

int zett_op_gemm_f32(const float* a, int32_t lda, const float* w, int32_t ldw, int64_t m, int32_t n, int32_t k, const float* bias, int32_t act,
                     const float* residual, int32_t ld_res, float* out, int32_t ld_out, void* stream) {
    if (!a || !w || !out) return fail(ZETT_E_INVALID, "null argument");
    if (m <= 0 || n <= 0) return 0;
    if (k <= 0 || k % 32) return fail(ZETT_E_INVALID, "contraction width %d is not a positive multiple of 32", k);
    if (lda % 4 || ldw % 4) return fail(ZETT_E_INVALID, "operand leading dimensions must be multiples of 4 floats");
    if (m >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many rows");
    GemmArgs<float> g{};
    g.A = a; g.lda = lda; g.W = w; g.ldw = ldw; g.M = (int)m; g.N = n; g.K = k;
    g.epi.split_col = 0x7fffffff;
    g.epi.bias = bias; g.epi.act = act; g.epi.residual = residual; g.epi.ld_res = ld_res; g.epi.out_f32 = out; g.epi.ld_f32 = ld_out;
    // the 256x256 register-staged tile where it pays and its 16-byte drains apply, the 128x128 tile otherwise (identical bits)
    const bool wide_ok = n % 8 == 0 && ld_out % 4 == 0 && (!residual || ld_res % 4 == 0);
    const int variant = (m > 128 && n > 128 && wide_ok) ? 2 : 1;
    const hipError_t e = launch_gemm_variant(variant, g, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ZETT_E_HIP, "gemm launch failed: %s", hipGetErrorString(e));
    return 0;
}

int zett_op_gemm_lo(int32_t prec, const void* a, int32_t lda, const void* w, int32_t ldw, int64_t m, int32_t n, int32_t k, const float* bias, int32_t act,
                    const float* residual, int32_t ld_res, float* out, int32_t ld_out, void* stream) {
    if (!a || !w || !out) return fail(ZETT_E_INVALID, "null argument");
    if (prec != ZETT_PREC_BF16 && prec != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "zett_op_gemm_lo takes ZETT_PREC_BF16 or ZETT_PREC_F16");
    if (m <= 0 || n <= 0) return 0;
    if (k <= 0 || k % 64) return fail(ZETT_E_INVALID, "contraction width %d is not a positive multiple of 64", k);
    if (lda % 8 || ldw % 8) return fail(ZETT_E_INVALID, "operand leading dimensions must be multiples of 8 elements");
    if (m >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many rows");
    if (prec == ZETT_PREC_F16) return gemm_lo<f16_t>((const f16_t*)a, lda, (const f16_t*)w, ldw, m, n, k, bias, act, residual, ld_res, out, ld_out, (hipStream_t)stream);
    return gemm_lo<bf16_t>((const bf16_t*)a, lda, (const bf16_t*)w, ldw, m, n, k, bias, act, residual, ld_res, out, ld_out, (hipStream_t)stream);
}

int zett_op_convert_lo(int32_t prec, const float* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int32_t cols_padded, void* stream) {
    if (!in || !out || cols_padded < cols || ld_out < cols_padded) return fail(ZETT_E_INVALID, "bad conversion arguments");
    if (prec != ZETT_PREC_BF16 && prec != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "zett_op_convert_lo takes ZETT_PREC_BF16 or ZETT_PREC_F16");
    if (rows <= 0 || cols <= 0) return 0;
    if (cols_padded == cols && cols % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0) {
        const int grid = (int)std::min<int64_t>(rows, 65535);
        if (prec == ZETT_PREC_F16) hipLaunchKernelGGL(convert_lo4_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, (f16_t*)out, ld_out, rows, cols);
        else hipLaunchKernelGGL(convert_lo4_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, (bf16_t*)out, ld_out, rows, cols);
    } else {
        const int grid = grid_for(rows * cols_padded);
        if (prec == ZETT_PREC_F16) hipLaunchKernelGGL(convert_lo_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, (f16_t*)out, ld_out, rows, cols, cols_padded);
        else hipLaunchKernelGGL(convert_lo_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, (bf16_t*)out, ld_out, rows, cols, cols_padded);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// in_lo: the input is already a 16-bit operand of type `prec` (plain transposition), otherwise fp32
static int transpose_lo_launch(int32_t prec, bool in_lo, const void* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded,
                               void* plain, int32_t ld_plain, float* colpart, const float* act_z, int32_t ld_z, int32_t act_kind, void* stream) {
    if (!in || !out || rows_padded < rows || ld_out < rows_padded) return fail(ZETT_E_INVALID, "bad transpose arguments");
    if (prec != ZETT_PREC_BF16 && prec != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "the 16-bit transposes take ZETT_PREC_BF16 or ZETT_PREC_F16");
    if (rows <= 0 || cols <= 0) return 0;
    if (plain && ld_plain < cols) return fail(ZETT_E_INVALID, "bad leading dimension of the plain copy");
    if (act_z && (act_kind != 1 && act_kind != 2)) return fail(ZETT_E_INVALID, "activation kind must be 1 (tanh-GELU) or 2 (erf-GELU)");
    if (act_z && ld_z < cols) return fail(ZETT_E_INVALID, "bad leading dimension of the pre-activation");
    if (((uintptr_t)in & (in_lo ? 7 : 15)) != 0 || ((uintptr_t)out & 7) != 0 || ((uintptr_t)plain & 7) != 0 || ((uintptr_t)act_z & 15) != 0)
        return fail(ZETT_E_INVALID, "the 16-bit transposes need 16-byte aligned fp32 inputs and 8-byte aligned 16-bit buffers");
    hipStream_t st = (hipStream_t)stream;
    if (prec == ZETT_PREC_F16) {
        if (in_lo) transpose_lo_go<f16_t, f16_t>(in, ld_in, out, ld_out, rows, cols, rows_padded, plain, ld_plain, colpart, act_z, ld_z, act_kind, st);
        else transpose_lo_go<f16_t, float>(in, ld_in, out, ld_out, rows, cols, rows_padded, plain, ld_plain, colpart, act_z, ld_z, act_kind, st);
    } else {
        if (in_lo) transpose_lo_go<bf16_t, bf16_t>(in, ld_in, out, ld_out, rows, cols, rows_padded, plain, ld_plain, colpart, act_z, ld_z, act_kind, st);
        else transpose_lo_go<bf16_t, float>(in, ld_in, out, ld_out, rows, cols, rows_padded, plain, ld_plain, colpart, act_z, ld_z, act_kind, st);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_transpose_lo(int32_t prec, const float* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* stream) {
    return transpose_lo_launch(prec, false, in, ld_in, out, ld_out, rows, cols, rows_padded, nullptr, 0, nullptr, nullptr, 0, 0, stream);
}

int zett_op_transpose_lo16(int32_t prec, const void* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* stream) {
    return transpose_lo_launch(prec, true, in, ld_in, out, ld_out, rows, cols, rows_padded, nullptr, 0, nullptr, nullptr, 0, 0, stream);
}

int zett_op_grad_operands_lo(int32_t prec, const float* dy, int32_t ld, const float* act_z, int32_t ld_z, int32_t act_kind, int64_t rows, int32_t cols,
                             int64_t rows_padded, void* dy_lo, int32_t ld_lo, void* dy_t, int32_t ld_t, float* colsum_part, void* stream) {
    if (!dy_lo || !colsum_part) return fail(ZETT_E_INVALID, "null argument");
    return transpose_lo_launch(prec, false, dy, ld, dy_t, ld_t, rows, cols, rows_padded, dy_lo, ld_lo, colsum_part, act_z, ld_z, act_kind, stream);
}

int zett_op_transpose_f32(const float* in, int32_t ld_in, float* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* stream) {
    if (!in || !out || rows_padded < rows || ld_out < rows_padded) return fail(ZETT_E_INVALID, "bad transpose arguments");
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (unsigned)((rows_padded + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       in, ld_in, out, ld_out, (int)rows, cols, (int)rows_padded);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_colsum_f32(const float* in, int32_t ld, int64_t rows, int32_t cols, float* out, int32_t accumulate, void* stream) {
    if (!in || !out) return fail(ZETT_E_INVALID, "null argument");
    if (cols <= 0) return 0;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, ld, (int)rows, cols, out, accumulate);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_elementwise_f32(int32_t op, const float* a, const float* b, const float* vec, const float* vec2, const float* s, float* out,
                            int64_t n, int32_t cols, void* stream) {
    if (!out || op < 0 || op > 4 || cols <= 0) return fail(ZETT_E_INVALID, "bad elementwise arguments");
    if (n <= 0) return 0;
    const bool wide = cols % 4 == 0 && n % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    if (wide) hipLaunchKernelGGL(elementwise_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, op, a, b, vec, vec2, s, out, n, cols);
    else hipLaunchKernelGGL(elementwise_kernel<1>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, vec, vec2, s, out, n, cols);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_rowdot_f32(const float* a, int32_t ld, const float* w, const float* b, float* out, int64_t rows, int32_t cols, void* stream) {
    if (!a || !w || !out) return fail(ZETT_E_INVALID, "null argument");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, ld, w, b, out, (int)rows, cols);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_layernorm_fwd_f32(const float* x, int32_t ld, const float* gamma, const float* beta, float eps, float* y, float* stats,
                              int64_t rows, int32_t h, void* y_lo, int32_t prec, void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return fail(ZETT_E_INVALID, "null argument");
    if (h < 4 || h % 4 || h > 8192 || ld % 4) return fail(ZETT_E_INVALID, "LayerNorm: 4 <= h <= 8192, h and ld multiples of 4");
    if ((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15) != 0 || ((uintptr_t)y_lo & 7) != 0)
        return fail(ZETT_E_INVALID, "LayerNorm needs 16-byte aligned rows");
    if (y_lo && prec != ZETT_PREC_BF16 && prec != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "the 16-bit copy takes ZETT_PREC_BF16 or ZETT_PREC_F16");
    if (rows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::min<int64_t>(rows, 16384));
    const int j = (h + 1023) / 1024;
    if (j <= 1) ln_fwd_go<1>(prec, grid, st, x, ld, gamma, beta, eps, y, stats, (int)rows, h, y_lo);
    else if (j <= 2) ln_fwd_go<2>(prec, grid, st, x, ld, gamma, beta, eps, y, stats, (int)rows, h, y_lo);
    else if (j <= 4) ln_fwd_go<4>(prec, grid, st, x, ld, gamma, beta, eps, y, stats, (int)rows, h, y_lo);
    else ln_fwd_go<8>(prec, grid, st, x, ld, gamma, beta, eps, y, stats, (int)rows, h, y_lo);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_layernorm_bwd_f32(const float* dy, const float* dy2, const float* x, int32_t ld, const float* stats, const float* gamma, float* dx,
                              float* partials, int32_t n_part, int64_t rows, int32_t h, void* stream) {
    if (!dy || !x || !stats || !gamma || !dx || !partials) return fail(ZETT_E_INVALID, "null argument");
    if (n_part < 1 || h < 4 || h % 4 || h > 8192 || ld % 4) return fail(ZETT_E_INVALID, "LayerNorm backward: 4 <= h <= 8192, h and ld multiples of 4, n_part >= 1");
    if ((((uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)partials) & 15) != 0)
        return fail(ZETT_E_INVALID, "LayerNorm backward needs 16-byte aligned rows");
    if (rows < 0) rows = 0;                            // (no rows: the partials are still written, as zeros)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(n_part), block(256);
    const int j = (h + 1023) / 1024;
    if (j <= 1) hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, block, 0, st, dy, dy2, x, ld, stats, gamma, dx, partials, (int)rows, h);
    else if (j <= 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, 0, st, dy, dy2, x, ld, stats, gamma, dx, partials, (int)rows, h);
    else if (j <= 4) hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, block, 0, st, dy, dy2, x, ld, stats, gamma, dx, partials, (int)rows, h);
    else hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, block, 0, st, dy, dy2, x, ld, stats, gamma, dx, partials, (int)rows, h);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_gelu_fwd_f32(const float* z, float* h, int64_t n, int32_t kind, void* stream) {
    if (!z || !h || (kind != 1 && kind != 2)) return fail(ZETT_E_INVALID, "bad gelu arguments");
    if (n <= 0) return 0;
    if (n % 4 == 0 && (((uintptr_t)z | (uintptr_t)h) & 15) == 0) hipLaunchKernelGGL(gelu_fwd_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, z, h, n, kind);
    else hipLaunchKernelGGL(gelu_fwd_kernel<1>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, h, n, kind);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_gelu_fwd_lo(int32_t prec, const float* z, void* h_lo, int64_t n, int32_t kind, void* stream) {
    if (!z || !h_lo || (kind != 1 && kind != 2)) return fail(ZETT_E_INVALID, "bad gelu arguments");
    if (prec != ZETT_PREC_BF16 && prec != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "zett_op_gelu_fwd_lo takes ZETT_PREC_BF16 or ZETT_PREC_F16");
    if ((((uintptr_t)z) & 15) != 0 || (((uintptr_t)h_lo) & 7) != 0) return fail(ZETT_E_INVALID, "zett_op_gelu_fwd_lo needs a 16-byte aligned input and an 8-byte aligned output");
    if (n <= 0) return 0;
    const int grid = grid_for((n + 3) / 4);
    if (prec == ZETT_PREC_F16) hipLaunchKernelGGL(gelu_fwd_lo_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, (f16_t*)h_lo, n, kind);
    else hipLaunchKernelGGL(gelu_fwd_lo_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, (bf16_t*)h_lo, n, kind);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_gelu_bwd_f32(const float* z, const float* dh, float* dz, int64_t n, int32_t kind, void* stream) {
    if (!z || !dh || !dz || (kind != 1 && kind != 2)) return fail(ZETT_E_INVALID, "bad gelu arguments");
    if (n <= 0) return 0;
    if (n % 4 == 0 && (((uintptr_t)z | (uintptr_t)dh | (uintptr_t)dz) & 15) == 0)
        hipLaunchKernelGGL(gelu_bwd_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, z, dh, dz, n, kind);
    else hipLaunchKernelGGL(gelu_bwd_kernel<1>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, dh, dz, n, kind);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_attention_fwd_f32(const float* q, int32_t ldq, const float* k, const float* v, int32_t ld, const uint8_t* mask, const int32_t* row_offset,
                              int64_t n_rows, int32_t seq, int32_t heads, int32_t head_dim, int32_t cls_only, float* ctx, int32_t ld_ctx, float* probs,
                              void* ctx_lo, int32_t prec, void* stream) {
    if (!q || !k || !v || !mask || (!ctx && !ctx_lo) || !probs) return fail(ZETT_E_INVALID, "null argument");
    if (ctx_lo && prec != ZETT_PREC_BF16 && prec != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "the 16-bit context takes ZETT_PREC_BF16 or ZETT_PREC_F16");
    if (seq < 1 || seq > ATT_MAX_L) return fail(ZETT_E_INVALID, "training attention handles 1 <= L <= %d positions, got %d", ATT_MAX_L, seq);
    if (heads < 1 || head_dim < 1 || head_dim > 256) return fail(ZETT_E_INVALID, "training attention handles head dims up to 256, got %d", head_dim);
    if (n_rows <= 0) return 0;
    const dim3 grid((unsigned)(n_rows * ((heads + 3) / 4)));
    const float scaling = 1.0f / sqrtf((float)head_dim);
    const int lo_kind = !ctx_lo ? 0 : (prec == ZETT_PREC_BF16 ? 1 : 2);
    hipStream_t st = (hipStream_t)stream;
    if (head_dim <= 64) attn_fwd_go<1>(seq, grid, st, q, ldq, k, v, ld, mask, row_offset, seq, heads, head_dim, scaling, cls_only, ctx, ld_ctx, probs, ctx_lo, lo_kind);
    else if (head_dim <= 128) attn_fwd_go<2>(seq, grid, st, q, ldq, k, v, ld, mask, row_offset, seq, heads, head_dim, scaling, cls_only, ctx, ld_ctx, probs, ctx_lo, lo_kind);
    else attn_fwd_go<4>(seq, grid, st, q, ldq, k, v, ld, mask, row_offset, seq, heads, head_dim, scaling, cls_only, ctx, ld_ctx, probs, ctx_lo, lo_kind);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_attention_bwd_f32(const float* dctx, int32_t ld_ctx, const float* q, int32_t ldq, const float* k, const float* v, int32_t ld, const float* probs,
                              const int32_t* row_offset, int64_t n_rows, int32_t seq, int32_t heads, int32_t head_dim, int32_t cls_only, float* dq, int32_t ld_dq,
                              float* dk, float* dv, int32_t ld_d, void* stream) {
    if (!dctx || !q || !k || !v || !probs || !dq || !dk || !dv) return fail(ZETT_E_INVALID, "null argument");
    if (seq < 1 || seq > ATT_MAX_L) return fail(ZETT_E_INVALID, "training attention handles 1 <= L <= %d positions, got %d", ATT_MAX_L, seq);
    if (heads < 1 || head_dim < 1 || head_dim > 256) return fail(ZETT_E_INVALID, "training attention handles head dims up to 256, got %d", head_dim);
    if (head_dim > 128 && seq > 16) return fail(ZETT_E_INVALID, "training attention backward: head dims above 128 are handled for up to 16 positions, got %d", seq);
    if (n_rows <= 0) return 0;
    const dim3 grid((unsigned)(n_rows * ((heads + 3) / 4)));
    const float scaling = 1.0f / sqrtf((float)head_dim);
    hipStream_t st = (hipStream_t)stream;
    if (head_dim <= 64) attn_bwd_go<1>(seq, grid, st, dctx, ld_ctx, q, ldq, k, v, ld, probs, row_offset, seq, heads, head_dim, scaling, cls_only, dq, ld_dq, dk, dv, ld_d);
    else if (head_dim <= 128) attn_bwd_go<2>(seq, grid, st, dctx, ld_ctx, q, ldq, k, v, ld, probs, row_offset, seq, heads, head_dim, scaling, cls_only, dq, ld_dq, dk, dv, ld_d);
    else attn_bwd_go<4>(seq, grid, st, dctx, ld_ctx, q, ldq, k, v, ld, probs, row_offset, seq, heads, head_dim, scaling, cls_only, dq, ld_dq, dk, dv, ld_d);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_gather_rows_f32(const float* a, const float* src, int32_t ld_src, const int32_t* idx, float* out, int64_t rows, int32_t cols, void* stream) {
    if (!src || !idx || !out) return fail(ZETT_E_INVALID, "null argument");
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, src, ld_src, idx, out, cols);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_scatter_add_rows_f32(float* dst, int32_t ld_dst, const int32_t* idx, const float* src, int64_t rows, int32_t cols, void* stream) {
    if (!dst || !idx || !src) return fail(ZETT_E_INVALID, "null argument");
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dst, ld_dst, idx, src, cols);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_gather_fwd_f32(const int32_t* ids, int64_t n_tokens, const void* src, int32_t src_dtype, int32_t e_in, int32_t v0, const float* fallback,
                           const float* sw, const float* sb, float* x, void* stream) {
    if (!ids || !src || !fallback || !x || src_dtype < ZETT_F32 || src_dtype > ZETT_BF16) return fail(ZETT_E_INVALID, "bad gather arguments");
    if (n_tokens <= 0) return 0;
    const dim3 grid((unsigned)n_tokens), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == ZETT_F32) hipLaunchKernelGGL(gather_fwd_kernel<0>, grid, block, 0, st, ids, n_tokens, src, e_in, v0, fallback, sw, sb, x);
    else if (src_dtype == ZETT_F16) hipLaunchKernelGGL(gather_fwd_kernel<1>, grid, block, 0, st, ids, n_tokens, src, e_in, v0, fallback, sw, sb, x);
    else hipLaunchKernelGGL(gather_fwd_kernel<2>, grid, block, 0, st, ids, n_tokens, src, e_in, v0, fallback, sw, sb, x);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_op_gather_bwd_f32(const int32_t* ids, int64_t n_tokens, const void* src, int32_t src_dtype, int32_t e_in, int32_t v0, const float* dx,
                           float* dfallback, float* prod, float* keep, void* stream) {
    if (!ids || !src || !dx || !dfallback || !prod || !keep || src_dtype < ZETT_F32 || src_dtype > ZETT_BF16) return fail(ZETT_E_INVALID, "bad gather arguments");
    if (n_tokens <= 0) return 0;
    const dim3 grid((unsigned)n_tokens), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == ZETT_F32) hipLaunchKernelGGL(gather_bwd_kernel<0>, grid, block, 0, st, ids, n_tokens, src, e_in, v0, dx, dfallback, prod, keep);
    else if (src_dtype == ZETT_F16) hipLaunchKernelGGL(gather_bwd_kernel<1>, grid, block, 0, st, ids, n_tokens, src, e_in, v0, dx, dfallback, prod, keep);
    else hipLaunchKernelGGL(gather_bwd_kernel<2>, grid, block, 0, st, ids, n_tokens, src, e_in, v0, dx, dfallback, prod, keep);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
