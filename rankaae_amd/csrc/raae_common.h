// Shared device helpers for librankaae_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rankaae_hip.h"

#define RAAE_WAVE 64

#define RAAE_CHECK_ARG(cond) do { if (!(cond)) return RAAE_EINVAL; } while (0)
#define RAAE_LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return (int)e_; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace raae {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of a double (blockDim.x multiple of 64, <= 1024). `sh` holds >= 16 doubles.
// Result valid in every thread.  Fixed order => deterministic.
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

__device__ __forceinline__ float prelu(float x, float a) { return x > 0.f ? x : a * x; }

// softplus with beta=2, threshold=20 (torch.nn.Softplus(beta=2): model.py:389,446,535)
__device__ __forceinline__ float softplus2(float x) {
    const float bx = 2.f * x;
    return bx > 20.f ? x : 0.5f * log1pf(expf(bx));
}

// Reduce BatchNorm partial sums -> mean / rstd for C channels into shared arrays.
// Every thread of the block must call it; ends with __syncthreads().
// Train mode: biased variance for normalisation; block 0 (when update_running) applies
// running = (1-m) running + m * {mean, unbiased var}  (torch.nn.BatchNorm1d semantics).
__device__ __forceinline__ void bn_prologue(const raae_bn_t& bn, int C, float* s_mean, float* s_rstd,
                                            bool is_block0) {
    if (bn.partials != nullptr) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            double s = 0.0, q = 0.0;
            const double* p = bn.partials + 2 * (size_t)c;
            for (int i = 0; i < bn.nparts; ++i) { s += p[0]; q += p[1]; p += 2 * (size_t)C; }
            const double n = (double)bn.count;
            const double mean = s / n;
            double var = q / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[c] = (float)mean;
            s_rstd[c] = (float)(1.0 / sqrt(var + (double)bn.eps));
            if (is_block0 && bn.update_running && bn.running_mean != nullptr) {
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                bn.running_mean[c] = (float)((1.0 - bn.momentum) * (double)bn.running_mean[c] + bn.momentum * mean);
                bn.running_var[c] = (float)((1.0 - bn.momentum) * (double)bn.running_var[c] + bn.momentum * unb);
            }
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            s_mean[c] = bn.running_mean[c];
            s_rstd[c] = 1.0f / sqrtf(bn.running_var[c] + bn.eps);
        }
    }
    __syncthreads();
}

// Column sums {sum g, sum g*y} partials -> per-channel mean terms for BN backward.
__device__ __forceinline__ void bnbwd_prologue(const double* partials, int nparts, int C, float count,
                                               float* s_m1, float* s_m2) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0.0, q = 0.0;
        const double* p = partials + 2 * (size_t)c;
        for (int i = 0; i < nparts; ++i) { s += p[0]; q += p[1]; p += 2 * (size_t)C; }
        s_m1[c] = (float)(s / (double)count);
        s_m2[c] = (float)(q / (double)count);
    }
    __syncthreads();
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace raae
