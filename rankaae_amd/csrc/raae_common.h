// Shared device helpers for librankaae_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rankaae_hip.h"

#define RAAE_WAVE 64

#define RAAE_CHECK_ARG(cond) do { if (!(cond)) return RAAE_EINVAL; } while (0)
#define RAAE_LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return (int)e_; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef RAAE_STAMPS
extern __device__ long long d_stamps[4][3][16];
extern __device__ unsigned long long d_stage_sum[4][3][16], d_stage_cnt[4][3][16];
extern __device__ long long d_stage_prev[4];
extern __device__ unsigned long long d_kind_sum[8][4][16], d_kind_cnt[8][4][16];
#endif
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

// Fixed-order reduction of per-workgroup partials [nparts][C][2] (doubles) into per-channel
// totals, spread over the first 256 threads: thread (c, j) sums parts j, j+J, ... (independent
// loads, all in flight together), then thread c adds the J slices in order.  C <= 256.
// Results land in LDS arrays tot_s / tot_q (doubles, >= C).  Ends with __syncthreads().
__device__ __forceinline__ void reduce_partials(const double* partials, int nparts, int C, double* tot_s,
                                                double* tot_q) {
    __shared__ double scr_s[256], scr_q[256];
    int Cp = 1;
    while (Cp < C) Cp <<= 1;
    const int J = 256 / Cp;                 // >= 1 because C <= 256
    const int t = threadIdx.x;
    if (t < 256) {
        const int c = t & (Cp - 1), j = t / Cp;
        double s = 0.0, q = 0.0;
        if (c < C) {
            const double2* p = reinterpret_cast<const double2*>(partials) + c;
            int i = j;
            for (; i + 3 * J < nparts; i += 4 * J) {
                const double2 v0 = p[(size_t)i * C], v1 = p[(size_t)(i + J) * C];
                const double2 v2 = p[(size_t)(i + 2 * J) * C], v3 = p[(size_t)(i + 3 * J) * C];
                s += v0.x; q += v0.y; s += v1.x; q += v1.y; s += v2.x; q += v2.y; s += v3.x; q += v3.y;
            }
            for (; i < nparts; i += J) { const double2 v = p[(size_t)i * C]; s += v.x; q += v.y; }
        }
        scr_s[t] = s; scr_q[t] = q;
    }
    __syncthreads();
    if (t < C) {
        double s = 0.0, q = 0.0;
        for (int j = 0; j < J; ++j) { s += scr_s[j * Cp + t]; q += scr_q[j * Cp + t]; }
        tot_s[t] = s; tot_q[t] = q;
    }
    __syncthreads();
}

// BatchNorm statistics for a consumer: mean / rstd of C channels into LDS arrays.
// Every thread of the block must call it (blockDim >= 256 or C <= blockDim); ends with a barrier.
// Train mode: biased variance for normalisation; block 0 (when update_running) applies
// running = (1-m) running + m * {mean, unbiased var}  (torch.nn.BatchNorm1d semantics).
__device__ __forceinline__ void bn_prologue(const raae_bn_t& bn, int C, float* s_mean, float* s_rstd,
                                            bool is_block0) {
    if (bn.partials != nullptr) {
        __shared__ double tot_s[256], tot_q[256];
        reduce_partials(bn.partials, bn.nparts, C, tot_s, tot_q);
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const double n = (double)bn.count;
            const double mean = tot_s[c] / n;
            double var = tot_q[c] / n - mean * mean;
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
    __shared__ double tot_s[256], tot_q[256];
    reduce_partials(partials, nparts, C, tot_s, tot_q);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        s_m1[c] = (float)(tot_s[c] / (double)count);
        s_m2[c] = (float)(tot_q[c] / (double)count);
    }
    __syncthreads();
}



// Sum NV (power of two) doubles per lane over groups of `width` consecutive lanes (power of two, NV <= width
// <= 64) with a halving butterfly: log2(NV) exchange steps that halve the number of live values, then plain
// xor steps.  Afterwards lane l holds in v[0] the group total of value index group_multi_slot<NV>(l)
// (NV + log2(width/NV) shuffles instead of NV * log2(width)).
// (compile-time recursion: with a runtime loop variable for the number of live values the array is indexed
// dynamically and ends up in scratch memory -- 14 MB of HBM writes per launch of the weight-gradient kernel)
template <int NV, int N>
struct MultiSumLevel {
    static __device__ __forceinline__ void run(double (&v)[NV], int lane) {
        constexpr int half = N / 2, o = NV / N;
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const double send = up ? v[i] : v[i + half];
            const double keep = up ? v[i + half] : v[i];
            v[i] = keep + __shfl_xor(send, o, 64);
        }
        MultiSumLevel<NV, half>::run(v, lane);
    }
};
template <int NV>
struct MultiSumLevel<NV, 1> {
    static __device__ __forceinline__ void run(double (&)[NV], int) {}
};
template <int NV>
__device__ __forceinline__ void group_multi_sum(double (&v)[NV], int lane, int width) {
    MultiSumLevel<NV, NV>::run(v, lane);
    for (int o = NV; o < width; o <<= 1) v[0] += __shfl_xor(v[0], o, 64);
}
template <int NV>
__device__ __forceinline__ int group_multi_slot(int lane) {       // bit k of the lane selects the half at level k
    int idx = 0;
    for (int o = 1, half = NV >> 1; o < NV; o <<= 1, half >>= 1) if (lane & o) idx += half;
    return idx;
}

// Copy the kernel's by-value argument struct (first and only parameter, offset 0 of the kernarg segment)
// into LDS with ONE coalesced vector load per thread.  Scalar s_loads of a 600-byte struct come out as a
// chain of dependent cache-line misses (the kernarg buffer is fresh for every launch): several
// microseconds at the head of a kernel that runs for ten.  All threads call it; ends with a barrier.
// The same for an argument block that starts `byte_off` bytes (a multiple of 4) into the kernarg segment.
template <typename T>
__device__ __forceinline__ const T& args_to_lds_at(T* slot, int byte_off) {
    typedef __attribute__((address_space(4))) const unsigned* kptr_t;
    const kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr() + byte_off / 4;
    unsigned* dst = reinterpret_cast<unsigned*>(slot);
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 4); i += blockDim.x) dst[i] = kp[i];
    __syncthreads();
    return *slot;
}
template <typename T>
__device__ __forceinline__ const T& args_to_lds(T* slot) {
    typedef __attribute__((address_space(4))) const unsigned* kptr_t;
    const kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned* dst = reinterpret_cast<unsigned*>(slot);
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 4); i += blockDim.x) dst[i] = kp[i];
    __syncthreads();
    return *slot;
}

// ---- several BatchNorm statistic reductions in ONE pass (one pair of barriers for all of them) ----
// kind 0: partials {sum x, sum x^2}        -> o1 = mean,        o2 = rstd   (eval mode when partials == NULL)
// kind 1: partials {sum g, sum g*y}        -> o1 = mean(g),     o2 = mean(g*y)
struct StatJob {
    const double* partials; int nparts; int C; float count; float eps; int kind;
    float* o1; float* o2; float* rm; float* rv; float momentum; int update;
};
__device__ __forceinline__ StatJob stat_job_bn(const raae_bn_t& bn, int C, float* mean, float* rstd, bool update) {
    StatJob j;
    j.partials = bn.partials; j.nparts = bn.nparts; j.C = C; j.count = bn.count; j.eps = bn.eps; j.kind = 0;
    j.o1 = mean; j.o2 = rstd; j.rm = bn.running_mean; j.rv = bn.running_var; j.momentum = bn.momentum;
    j.update = (update && bn.update_running) ? 1 : 0;
    return j;
}
__device__ __forceinline__ StatJob stat_job_bwd(const double* partials, int nparts, int C, float count, float* m1, float* m2) {
    StatJob j;
    j.partials = partials; j.nparts = nparts; j.C = C; j.count = count; j.eps = 0.f; j.kind = 1;
    j.o1 = m1; j.o2 = m2; j.rm = nullptr; j.rv = nullptr; j.momentum = 0.f; j.update = 0;
    return j;
}
__device__ __forceinline__ StatJob stat_job_none() {
    StatJob j;
    j.partials = nullptr; j.nparts = 0; j.C = 0; j.count = 1.f; j.eps = 0.f; j.kind = 2; j.o1 = j.o2 = j.rm = j.rv = nullptr;
    j.momentum = 0.f; j.update = 0;
    return j;
}
// All threads of a 256-thread block call it.  C <= 64 per job.
// One WAVE per job (the k-th active job runs on wave k & 3): its 64 lanes sweep the partial rows eight
// loads deep, lanes that share a channel meet in an xor-shuffle tree, and lanes < C finish the statistic.
// Jobs proceed side by side on the four SIMDs; nothing goes through LDS and there is one barrier.
// (Measured on MI355X: the loads return in 0.2 us -- what costs is dependent shuffle chains and issue
// slots, so they are spread over waves instead of repeated in each.)
template <int N, int MAXC = 64>
__device__ __forceinline__ void stat_jobs(const StatJob (&jobs)[N], bool is_block0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const StatJob& jb = jobs[n];
        if (jb.kind == 2) continue;
        const bool mine = (rank & 3) == wave;
        ++rank;
        if (!mine) continue;
        if (jb.partials == nullptr) {              // eval mode: running statistics
            if (lane < jb.C) {
                jb.o1[lane] = jb.rm[lane];
                jb.o2[lane] = 1.0f / sqrtf(jb.rv[lane] + jb.eps);
            }
            continue;
        }
        // workgroup 0 updates the running statistics: their loads are issued BEFORE the sweep of the partial rows and
        // used after it (as a read-modify-write behind the reduction they were one more dependent round trip of the
        // workgroup that ends the kernel last)
        const bool upd = is_block0 && jb.update && jb.rm != nullptr && lane < jb.C;
        const float rm0 = upd ? jb.rm[lane] : 0.f, rv0 = upd ? jb.rv[lane] : 0.f;
        int cp = 1;
        while (cp < jb.C) cp <<= 1;
        const int J = 64 / cp, c = lane & (cp - 1), j = lane / cp;
        const double2* p = reinterpret_cast<const double2*>(jb.partials) + (c < jb.C ? c : 0);
        const int last = jb.nparts - 1;
        double s = 0.0, q = 0.0;
        for (int i = j; i < jb.nparts; i += 8 * J) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {          // clamped address, value dropped below: no branch per load
                const int r = i + u * J;
                v[u] = p[(size_t)(r < last ? r : last) * jb.C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = i + u * J <= last;
                s += ok ? v[u].x : 0.0; q += ok ? v[u].y : 0.0;
            }
        }
        if (c >= jb.C) { s = 0.0; q = 0.0; }
        for (int o = 32; o >= cp; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (lane < jb.C) {
            const double inv = 1.0 / (double)jb.count;
            if (jb.kind == 0) {
                const double mean = s * inv;
                double var = q * inv - mean * mean;
                if (var < 0.0) var = 0.0;
                jb.o1[lane] = (float)mean;
                // double like ATen's CPU accumulate type: a float rstd moves PReLU-slope gradients of the
                // compact fixture by 2% (ill-conditioned sums), see DESIGN.md
                jb.o2[lane] = (float)(1.0 / sqrt(var + (double)jb.eps));
                if (upd) {
                    const double n_ = (double)jb.count;
                    const double unb = n_ > 1.0 ? var * n_ / (n_ - 1.0) : var;
                    jb.rm[lane] = (float)((1.0 - jb.momentum) * (double)rm0 + jb.momentum * mean);
                    jb.rv[lane] = (float)((1.0 - jb.momentum) * (double)rv0 + jb.momentum * unb);
                }
            } else {
                jb.o1[lane] = (float)(s * inv);
                jb.o2[lane] = (float)(q * inv);
            }
        }
    }
    __syncthreads();
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- batched launches over independent TRIALS (the reference's workload is `trials: 8` small models) ----
// A kernel that supports it exists twice: K(Args) for one trial and K_m(const Args* table), launched with
// gridDim.z = T, whose workgroups of plane z take table[z] -- the SAME body, the same blockIdx.x / gridDim.x, so a
// trial's arithmetic is bit for bit what it is alone.  raae::launch() launches K and, while the calling thread is
// recording (raae_record_begin), logs {K_m, grid, block, LDS, the argument block}; raae_multi_build() checks that the
// logs of T trials agree launch by launch and uploads the T argument blocks of every launch as one device table;
// raae_multi_launch() replays the program with gridDim.z = T (capturable into a hipGraph).
void record_launch(const void* multi_fn, dim3 grid, dim3 block, size_t lds, const void* args, size_t nbytes);
void record_unsupported();          // a launch without a batched form: a recording in progress becomes invalid
#define RAAE_PLAIN_LAUNCH(...) do { raae::record_unsupported(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
template <typename A>
inline void launch(void (*single)(A), void (*multi)(const A*), dim3 grid, dim3 block, size_t lds, hipStream_t st, const A& a) {
    record_launch(reinterpret_cast<const void*>(multi), grid, block, lds, &a, sizeof(A));
    hipLaunchKernelGGL(single, grid, block, lds, st, a);
}
// plane z's argument block -> LDS in one coalesced load (the table counterpart of args_to_lds)
template <typename T>
__device__ __forceinline__ const T& args_from_ptr(T* slot, const T* entry) {
    const unsigned* src = reinterpret_cast<const unsigned*>(entry);
    unsigned* dst = reinterpret_cast<unsigned*>(slot);
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 4); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    return *slot;
}
template <typename T>
__device__ __forceinline__ const T& args_from_table(T* slot, const T* table) { return args_from_ptr(slot, table + blockIdx.z); }

// ---- dropout multipliers from a counter-based hash (rng_mode "philox") ---------------------------------------------
// A dropout multiplier carries ONE bit.  Instead of a Philox block per four multipliers written to a fp32 tape and read
// back by every consumer (146 MB per step at 4096 rows), element e of the step's random tape is
//     keep(e) = lowbias32(lowbias32(e + k1) ^ k2) < keep * 2^32,      (k1, k2) = f(seed, step counter)
// (lowbias32: C. Wellons' 2-multiply avalanche hash, bias 0.17; two rounds keyed differently decorrelate the steps).
// ~12 integer instructions per element, any access granularity, so a consumer regenerates its multipliers while it
// loads the tensor they scale.  raae_rng_fill evaluates THE SAME function for the slots that still live on the tape
// (consumers without in-kernel generation, `inline_masks: false`): tape and in-kernel masks are bit-identical
// (tests/test_engine_gpu.py::test_inline_masks_equal_tape_masks).  Gaussian slots stay Philox4x32-10 + Box-Muller.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct MaskGen { uint32_t k1, k2, thr, off; float inv; };
__device__ __forceinline__ MaskGen mask_gen_make(unsigned long long seed, unsigned long long ctr, uint32_t off, float keep) {
    MaskGen m;
    m.k1 = lowbias32((uint32_t)seed ^ lowbias32((uint32_t)ctr + 0x9E3779B9u));
    m.k2 = lowbias32((uint32_t)(seed >> 32) + 0x85EBCA6Bu + lowbias32((uint32_t)(ctr >> 32) ^ m.k1));
    const double t = (double)keep * 4294967296.0;
    m.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    m.inv = 1.f / keep;
    m.off = off;
    return m;
}
// the step's keys as raae_step_begin / raae_step_tick stored them (two words: no hashing, no counter arithmetic here)
__device__ __forceinline__ MaskGen mask_gen_from(const raae_maskgen_t& g) {
    MaskGen m;
    m.k1 = g.keys[0]; m.k2 = g.keys[1]; m.thr = g.thr; m.off = g.offset; m.inv = g.inv;
    return m;
}
// ... and what stores them: words [2] of the {counter, seed, keys} state
__device__ __forceinline__ void mask_keys_store(unsigned long long* state, unsigned long long seed, unsigned long long ctr) {
    const MaskGen m = mask_gen_make(seed, ctr, 0u, 1.f);
    reinterpret_cast<unsigned*>(state + 2)[0] = m.k1;
    reinterpret_cast<unsigned*>(state + 2)[1] = m.k2;
}
__device__ __forceinline__ bool mask_keep(const MaskGen& m, uint32_t e) {     // e: element index inside the slot
    return lowbias32(lowbias32(e + m.off + m.k1) ^ m.k2) < m.thr;
}
__device__ __forceinline__ float mask_val(const MaskGen& m, uint32_t e) { return mask_keep(m, e) ? m.inv : 0.f; }
__device__ __forceinline__ float4 mask_val4(const MaskGen& m, uint32_t e) {
    return make_float4(mask_val(m, e), mask_val(m, e + 1), mask_val(m, e + 2), mask_val(m, e + 3));
}

// ---- one to three BatchNorm statistic reductions spread over ALL FOUR waves of a 256-thread block -------------------
// (stat_jobs gives each statistic one wave: right for the conv kernels with their 4-8 channels and several jobs; a
// dense layer has 64 channels = one per lane, so a single wave walked every partial row: 32 dependent trips at 256
// rows -- 15 of the 20 us of a dense forward at 4096 rows.)  Wave w takes row phases w*J .. w*J + J - 1 of 4J
// (J = 64 / pow2(C) lanes share a channel); the first batch of every job's loads is issued before any is used; the
// four waves' sums meet in LDS in wave order (fixed order => deterministic).  C <= 64.  All threads call it; ends
// with a barrier.
template <int N>
__device__ __forceinline__ void stat_jobs_wide(const StatJob (&jobs)[N], bool is_block0) {
    __shared__ double2 scr[N][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int D = 4;                              // loads in flight per job and lane (first batch)
    double2 v[N][D];
    int cpv[N], cv[N], phv[N];
    // running statistics of the job this thread will finish (wave n finishes job n): loaded now, used at the end
    float rm0 = 0.f, rv0 = 0.f;
    bool upd = false;
#pragma unroll
    for (int n = 0; n < N; ++n)
        if (n == wave && jobs[n].kind == 0 && jobs[n].partials != nullptr && is_block0 && jobs[n].update &&
            jobs[n].rm != nullptr && lane < jobs[n].C) {
            upd = true; rm0 = jobs[n].rm[lane]; rv0 = jobs[n].rv[lane];
        }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const StatJob& jb = jobs[n];
        if (jb.kind == 2 || jb.partials == nullptr) continue;
        int cp = 1;
        while (cp < jb.C) cp <<= 1;
        const int J = 64 / cp, c = lane & (cp - 1);
        cpv[n] = cp; cv[n] = c; phv[n] = wave * J + lane / cp;
        const double2* p = reinterpret_cast<const double2*>(jb.partials) + (c < jb.C ? c : 0);
        const int last = jb.nparts - 1, nph = 4 * J;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int r = phv[n] + u * nph;
            v[n][u] = p[(size_t)(r < last ? r : last) * jb.C];
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const StatJob& jb = jobs[n];
        if (jb.kind == 2 || jb.partials == nullptr) continue;
        const int cp = cpv[n], c = cv[n], nph = 4 * (64 / cp), last = jb.nparts - 1;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const bool ok = phv[n] + u * nph <= last;
            s += ok ? v[n][u].x : 0.0; q += ok ? v[n][u].y : 0.0;
        }
        const double2* p = reinterpret_cast<const double2*>(jb.partials) + (c < jb.C ? c : 0);
        for (int i = phv[n] + D * nph; i < jb.nparts; i += 8 * nph) {
            double2 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = i + u * nph;
                w[u] = p[(size_t)(r < last ? r : last) * jb.C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = i + u * nph <= last;
                s += ok ? w[u].x : 0.0; q += ok ? w[u].y : 0.0;
            }
        }
        if (c >= jb.C) { s = 0.0; q = 0.0; }
        for (int o = 32; o >= cp; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (lane < jb.C) scr[n][wave][lane] = make_double2(s, q);
    }
    __syncthreads();
    // thread (n, c) = (tid / 64, tid % 64) finishes statistic c of job n  (compile-time n: a runtime index into the
    // job array would put it in scratch memory)
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if (n == wave) {
            const StatJob& jb = jobs[n];
            if (jb.kind != 2 && lane < jb.C) {
                if (jb.partials == nullptr) {          // eval mode: running statistics
                    jb.o1[lane] = jb.rm[lane];
                    jb.o2[lane] = 1.0f / sqrtf(jb.rv[lane] + jb.eps);
                } else {
                    double s = 0.0, q = 0.0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) { s += scr[n][w][lane].x; q += scr[n][w][lane].y; }
                    const double inv = 1.0 / (double)jb.count;
                    if (jb.kind == 0) {
                        const double mean = s * inv;
                        double var = q * inv - mean * mean;
                        if (var < 0.0) var = 0.0;
                        jb.o1[lane] = (float)mean;
                        jb.o2[lane] = (float)(1.0 / sqrt(var + (double)jb.eps));
                        if (upd) {
                            const double n_ = (double)jb.count;
                            const double unb = n_ > 1.0 ? var * n_ / (n_ - 1.0) : var;
                            jb.rm[lane] = (float)((1.0 - jb.momentum) * (double)rm0 + jb.momentum * mean);
                            jb.rv[lane] = (float)((1.0 - jb.momentum) * (double)rv0 + jb.momentum * unb);
                        }
                    } else {
                        jb.o1[lane] = (float)(s * inv);
                        jb.o2[lane] = (float)(q * inv);
                    }
                }
            }
        }
    }
    __syncthreads();
}

}  // namespace raae
