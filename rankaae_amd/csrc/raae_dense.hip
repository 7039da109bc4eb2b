// Fused dense layers on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32 -- exact fp32,
// bit-for-bit a k-ordered fmaf chain), forward and backward.
//
// Replaces, per call, the ATen chain  PReLU -> BatchNorm1d(affine=False) -> Dropout ->
// Linear (+ output activation) of FCEncoder / FCDecoder / DiscriminatorFC
// (reference sc/clustering/model.py:346-371, 540-563, 635-653) and its autograd backward.
//
// Data layout: activations row-major [B][features]; weights [N][K] (torch nn.Linear).
// A workgroup = 4 waves; it owns 16-row tiles of the batch (grid-stride) and, in the
// forward, a group of 64 output columns (one 16x16 MFMA tile per wave).  BatchNorm
// statistics and parameter gradients leave the kernel as fixed-order per-workgroup
// partials (double) / slabs (float): no atomics, bitwise reproducible.
#include "raae_common.h"
#include <stddef.h>
#include <hip/hip_bf16.h>

#ifdef RAAE_STAMPS
// per-instance stage clocks of workgroup 0 (100 MHz), summed over all launches: tools/dense_stamps.py
__device__ unsigned long long d_dense_sum[8][12], d_dense_cnt[8][12];
__device__ long long d_dense_prev[8];
#define DSTAMP(slot, i) do { __syncthreads(); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { \
        const long long now_ = wall_clock64(); \
        if ((i) > 0) { d_dense_sum[slot][i] += (unsigned long long)(now_ - d_dense_prev[slot]); d_dense_cnt[slot][i] += 1ull; } \
        d_dense_prev[slot] = now_; } } while (0)
extern "C" int raae_debug_dense_stamps(unsigned long long* sum, unsigned long long* cnt) {
    hipError_t e = hipMemcpyFromSymbol(sum, HIP_SYMBOL(d_dense_sum), sizeof(unsigned long long) * 8 * 12);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(cnt, HIP_SYMBOL(d_dense_cnt), sizeof(unsigned long long) * 8 * 12);
}
#else
#define DSTAMP(slot, i) do { } while (0)
#endif

namespace {

using raae::prelu;

struct DenseFwdArgs {
    const float* x; int B; int K; int in_kind; const float* slope; raae_bn_t bn; const float* mask;
    const float* w; const float* bias; int N; float* z; int out_kind; const float* out_slope;
    double* out_partials; int pitch; int storage; float mask_scale; raae_maskgen_t gen;
};

// ---- bf16 STORAGE of activations and dropout multipliers (`precision: bf16`, BASELINE configs[4]) ----------------
// RAAE_ST_X / _MASK / _Z: the layer input / its dropout multipliers / the layer's raw output live in memory as bf16
// (the pointers are still typed float*).  Everything is converted to fp32 on load and all arithmetic, statistics and
// accumulators stay fp32 / double; the output is rounded to bf16 (round to nearest even: v_cvt_pk_bf16_f32) BEFORE its
// BatchNorm statistics are taken, so that the statistics describe the values the next layer will read.  A bf16 mask
// holds {0, 1}; the fp32 1/(1-p) multiplies it here (mask_scale).  With storage == 0 no instruction of the fp32 path
// changes.
__device__ __forceinline__ float bf16_at(const float* p, size_t i) {
    return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
}
__device__ __forceinline__ float4 bf16x4_at(const float* p, size_t i) {        // i a multiple of 4
    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + i);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float bf16_round(float v) { return __bfloat162float(__float2bfloat16(v)); }
__device__ __forceinline__ void bf16_store(float* p, size_t i, float v) {
    reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v);
}

// Input transform of one element (PReLU -> BatchNorm -> Dropout scale), statistics from LDS.
__device__ __forceinline__ float in_transform(float v, int k, int in_kind, const float* s_slope, const float* s_mean,
                                              const float* s_rstd) {
    if (in_kind != RAAE_IN_NONE) {
        v = prelu(v, s_slope[k]);
        if (in_kind == RAAE_IN_PRELU_BN_DROP) v = (v - s_mean[k]) * s_rstd[k];
    }
    return v;
}

// Where a layer's dropout multipliers come from: a tensor (fp32 {0, 1/keep} or bf16 {0, 1} x scale), or the in-kernel
// hash generator (raae_common.h), or nowhere.
struct MaskSrc {
    const float* ptr; bool bf; float scale; bool gen; raae::MaskGen g;
    __device__ __forceinline__ bool any() const { return gen || ptr != nullptr; }
    __device__ __forceinline__ void key() {}
    __device__ __forceinline__ float4 at4(size_t o) const {        // o a multiple of 4
        if (gen) return raae::mask_val4(g, (uint32_t)o);
        if (bf) { const float4 m = bf16x4_at(ptr, o); return make_float4(m.x * scale, m.y * scale, m.z * scale, m.w * scale); }
        return *reinterpret_cast<const float4*>(ptr + o);
    }
    __device__ __forceinline__ float at(size_t o) const {
        if (gen) return raae::mask_val(g, (uint32_t)o);
        return bf ? bf16_at(ptr, o) * scale : ptr[o];
    }
};
__device__ __forceinline__ MaskSrc mask_src(const float* mask, int in_kind, int storage, float scale, const raae_maskgen_t& gen) {
    MaskSrc m;
    m.ptr = in_kind != RAAE_IN_NONE ? mask : nullptr;
    m.bf = (storage & RAAE_ST_MASK) != 0;
    m.scale = scale != 0.f ? scale : 1.f;
    m.gen = in_kind != RAAE_IN_NONE && gen.keys != nullptr;
    if (m.gen) m.g = raae::mask_gen_from(gen);         // two words; first used behind the statistic prologue
    return m;
}

// A tile of ROWS = 16 RT rows of a layer input as NV float4 per thread (K % 4 == 0): thread t holds float4 number
// t + 256 u of the tile, i.e. (row f / kq, columns 4 (f % kq) ..).  The raw values and their dropout multipliers are
// LOADED here and transformed / stored to LDS later, so that the loads of a tile are in flight during the statistic
// prologue (first tile) or the previous tile's matrix work (following tiles).
// (K: the row stride of x and of the multipliers' numbering; k0: first column of this workgroup's slice of the row)
template <int NV>
struct RawTile { float4 x[NV]; float4 m[NV]; };

template <int NV>
__device__ __forceinline__ void tile_coords(int kq, int (&rr)[NV], int (&cc)[NV]) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = threadIdx.x + 256 * u;
        rr[u] = f / kq;
        cc[u] = (f - rr[u] * kq) << 2;
    }
}
template <int NV>
__device__ __forceinline__ void tile_load(RawTile<NV>& t, const int (&rr)[NV], const int (&cc)[NV], int rows,
                                          const float* x, bool xb, const MaskSrc& ms, int row0, int B, int K, int k0 = 0) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int row = row0 + rr[u];
        t.x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        t.m[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (rr[u] < rows && row < B) {
            const size_t o = (size_t)row * K + k0 + cc[u];
            t.x[u] = xb ? bf16x4_at(x, o) : *reinterpret_cast<const float4*>(x + o);
            if (ms.ptr != nullptr) t.m[u] = ms.at4(o);       // (generated multipliers: tile_store, no memory involved)
        }
    }
}
// transform -> Xs (x * mask); optionally the pre-dropout value -> Ys and the multiplier -> Ms (backward epilogue)
template <int NV>
__device__ __forceinline__ void tile_store(const RawTile<NV>& t, const int (&rr)[NV], const int (&cc)[NV], int rows,
                                           float* Xs, float* Ys, float* Ms, int pitch, int in_kind, const float* s_slope,
                                           const float* s_mean, const float* s_rstd, const MaskSrc& ms, int row0, int B,
                                           int K, int k0 = 0) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (rr[u] >= rows) continue;
        const int k = cc[u];
        float4 v = t.x[u];
        v.x = in_transform(v.x, k, in_kind, s_slope, s_mean, s_rstd);
        v.y = in_transform(v.y, k + 1, in_kind, s_slope, s_mean, s_rstd);
        v.z = in_transform(v.z, k + 2, in_kind, s_slope, s_mean, s_rstd);
        v.w = in_transform(v.w, k + 3, in_kind, s_slope, s_mean, s_rstd);
        const int o = rr[u] * pitch + k;               // pitch even => 8-byte aligned
        if (Ys != nullptr) {
            reinterpret_cast<float2*>(Ys + o)[0] = make_float2(v.x, v.y);
            reinterpret_cast<float2*>(Ys + o)[1] = make_float2(v.z, v.w);
        }
        if (ms.any()) {
            float4 m = t.m[u];
            if (ms.gen) m = (row0 + rr[u] < B) ? ms.at4((size_t)(row0 + rr[u]) * K + k0 + k) : make_float4(1.f, 1.f, 1.f, 1.f);
            if (Ms != nullptr) {
                reinterpret_cast<float2*>(Ms + o)[0] = make_float2(m.x, m.y);
                reinterpret_cast<float2*>(Ms + o)[1] = make_float2(m.z, m.w);
            }
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        reinterpret_cast<float2*>(Xs + o)[0] = make_float2(v.x, v.y);
        reinterpret_cast<float2*>(Xs + o)[1] = make_float2(v.z, v.w);
    }
}

// The same for any K (a decoder's first layer has K = nstyle = 6): element by element, zero padded to KP columns.
__device__ __forceinline__ void stage_rows_scalar(float* Xs, float* Ys, float* Ms, int pitch, int rows, const float* x, bool xb,
                                                  const MaskSrc& ms, int row0, int B, int K, int KP, int in_kind,
                                                  const float* s_slope, const float* s_mean, const float* s_rstd) {
    for (int idx = threadIdx.x; idx < rows * KP; idx += 256) {
        const int r = idx / KP, k = idx - r * KP;
        const int row = row0 + r;
        float v = 0.f, m = 1.f;
        if (row < B && k < K) {
            const size_t o = (size_t)row * K + k;
            v = in_transform(xb ? bf16_at(x, o) : x[o], k, in_kind, s_slope, s_mean, s_rstd);
            if (ms.any()) m = ms.at(o);
        }
        if (Ys != nullptr) Ys[r * pitch + k] = v;
        if (Ms != nullptr) Ms[r * pitch + k] = m;
        Xs[r * pitch + k] = v * m;
    }
}

// dynamic LDS: s_mean[K4] s_rstd[K4] s_slope[K4] | Xs[16 RT][pitch].  The wave's 16 output columns of W
// stay in registers (KQ floats per lane = the B operands of all K/4 MFMA steps) across its row tiles.
// (bx, by) of a (gx, *) grid: the workgroup's tile / column-block index (the whole grid of dense_fwd_kernel, one of
// the two ranges of dense_fwd2_kernel)
// RT: 16-row tiles per pass.  RT = 4 (batches of >= 2048 rows): 64 rows are loaded, staged and multiplied together --
//     four independent accumulator chains per wave instead of one, one exposed memory round trip per 64 rows instead of
//     one per 16 (a tile takes less time than the prefetch of the next one needs), a quarter of the workgroups, i.e. of
//     the partial-statistic rows every consumer reduces.
// ST: instance that honours a.storage (bf16 storage); the fp32 instances (ST false) carry none of its branches
template <int KQ, int RT, bool ST>
__device__ __forceinline__ void dense_fwd_tiles(const DenseFwdArgs& a, const int bx, const int by, const int gx, float* smem,
                                                MaskSrc& ms) {
    const int st = ST ? a.storage : 0;
    constexpr int NV = (KQ >= 64 ? KQ / 16 : 1) * RT;      // float4s per thread of a 16 RT-row tile (K <= 4 KQ)
    constexpr int ROWS = 16 * RT;
    const int K4 = (a.K + 3) & ~3;
    float* s_mean = smem;
    float* s_rstd = s_mean + K4;
    float* s_slope = s_rstd + K4;
    float* Xs = s_slope + K4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = by * 64 + wv * 16 + (lane & 15);
    const bool vec = (a.K & 3) == 0, xb = (st & RAAE_ST_X) != 0;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    constexpr int SLOT = KQ == 4 ? 0 : (KQ == 16 ? 1 : (KQ == 64 ? 2 : 3));
    (void)SLOT;
    DSTAMP(SLOT, 0);

    // every global operand of the prologue is requested before any is used: the workgroup's weight block, the first
    // tile's raw input (and multipliers), the PReLU slopes of the input, bias and output slope -- and, inside
    // stat_jobs_wide, the partial statistics and the running statistics workgroup 0 updates.
    // Weights: the 64 x K block of the workgroup's output columns goes through LDS (KQ <= 64) -- 16-byte loads of
    // consecutive addresses; a lane fetching its own B operands W[col][4 q + lane / 16] straight from memory made
    // every load instruction touch 16 rows (7.7 of the 13 us of the 256 -> 64 first layer at 256 rows).  The 512-column
    // first layer (KQ = 128: 132 KB) keeps the register form.
    constexpr bool WLDS = KQ <= 64;
    constexpr int NW = WLDS ? (KQ >= 16 ? KQ / 4 : 1) : 1;      // float4s of the weight block per thread
    const int pitchW = K4 + 4;
    float* Ws = Xs + ROWS * a.pitch;
    float wreg[WLDS ? 1 : KQ];
    float4 wraw[NW];
    if (!WLDS) {
#pragma unroll
        for (int q = 0; q < (WLDS ? 1 : KQ); ++q) {
            const int k = 4 * q + (lane >> 4);
            wreg[q] = (col < a.N && k < a.K) ? a.w[(size_t)col * a.K + k] : 0.f;
        }
    } else if (vec) {
        const int kq = a.K >> 2;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int f = tid + 256 * u, c = f / kq, k4 = f - c * kq;
            wraw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < 64 && by * 64 + c < a.N) wraw[u] = *reinterpret_cast<const float4*>(a.w + (size_t)(by * 64 + c) * a.K + 4 * k4);
        }
    }
    int rr[NV], cc[NV];
    RawTile<NV> raw;
    if (vec) {
        tile_coords<NV>(a.K >> 2, rr, cc);
        tile_load<NV>(raw, rr, cc, ROWS, a.x, xb, ms, bx * ROWS, a.B, a.K);
    }
    float slope_pre[2] = {0.f, 0.f};
    if (a.in_kind != RAAE_IN_NONE) {
        if (tid < a.K) slope_pre[0] = a.slope[tid];
        if (tid + 256 < a.K) slope_pre[1] = a.slope[tid + 256];
    }
    const float bias = (col < a.N) ? a.bias[col] : 0.f;
    const float oslope = (a.out_kind == RAAE_OUT_STATS_PRELU && col < a.N) ? a.out_slope[col] : 1.f;
    if (a.in_kind == RAAE_IN_PRELU_BN_DROP) {
        if (a.K <= 64) {            // all four waves sweep the partial rows (raae_common.h)
            const raae::StatJob jobs[1] = {raae::stat_job_bn(a.bn, a.K, s_mean, s_rstd, true)};
            raae::stat_jobs_wide<1>(jobs, bx == 0 && by == 0);
        } else {
            raae::bn_prologue(a.bn, a.K, s_mean, s_rstd, bx == 0 && by == 0);
        }
    }
    if (a.in_kind != RAAE_IN_NONE) {
        if (tid < a.K) s_slope[tid] = slope_pre[0];
        if (tid + 256 < a.K) s_slope[tid + 256] = slope_pre[1];
    }
    if (WLDS) {
        if (vec) {
            const int kq = a.K >> 2;
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                const int f = tid + 256 * u, c = f / kq, k4 = f - c * kq;
                if (c < 64) *reinterpret_cast<float4*>(Ws + c * pitchW + 4 * k4) = wraw[u];
            }
        } else {
            for (int idx = tid; idx < 64 * K4; idx += 256) {
                const int c = idx / K4, k = idx - c * K4;
                Ws[c * pitchW + k] = (by * 64 + c < a.N && k < a.K) ? a.w[(size_t)(by * 64 + c) * a.K + k] : 0.f;
            }
        }
    }
    ms.key();
    __syncthreads();
    DSTAMP(SLOT, 1);

    double s_acc = 0.0, q_acc = 0.0;
    const float* xa = Xs + (lane & 15) * a.pitch + (lane >> 4);
    const float* wb = Ws + (wv * 16 + (lane & 15)) * pitchW + (lane >> 4);

    for (int tile = bx; tile < ntiles; tile += gx) {
        const int row0 = tile * ROWS;
        if (vec) {
            tile_store<NV>(raw, rr, cc, ROWS, Xs, nullptr, nullptr, a.pitch, a.in_kind, s_slope, s_mean, s_rstd, ms,
                                 row0, a.B, a.K);
            if (tile + gx < ntiles)       // the next tile's loads fly during this tile's matrix work and stores
                tile_load<NV>(raw, rr, cc, ROWS, a.x, xb, ms, (tile + gx) * ROWS, a.B, a.K);
        } else {
            stage_rows_scalar(Xs, nullptr, nullptr, a.pitch, ROWS, a.x, xb, ms, row0, a.B, a.K, K4, a.in_kind, s_slope, s_mean, s_rstd);
        }
        __syncthreads();
        DSTAMP(SLOT, 2);
        f32x4 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
            if (4 * q < K4) {
                const float bq = WLDS ? wb[4 * q] : wreg[WLDS ? 0 : q];
#pragma unroll
                for (int t = 0; t < RT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t * 16 * a.pitch + 4 * q], bq, acc[t], 0, 0, 0);
            }
        __syncthreads();
        DSTAMP(SLOT, 3);
        // epilogue: lane holds rows row0 + 16 t + (lane>>4)*4 + j of column `col`
        if (col < a.N) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = row0 + t * 16 + (lane >> 4) * 4 + j;
                    if (row < a.B) {
                        float zv = acc[t][j] + bias;
                        if (st & RAAE_ST_Z) zv = bf16_round(zv);
                        float o = zv;
                        if (a.out_kind == RAAE_OUT_SOFTPLUS) o = raae::softplus2(zv);
                        else if (a.out_kind == RAAE_OUT_RELU) o = fmaxf(zv, 0.f);
                        if (st & RAAE_ST_Z) bf16_store(a.z, (size_t)row * a.N + col, o);
                        else a.z[(size_t)row * a.N + col] = o;
                        if (a.out_kind == RAAE_OUT_STATS_PRELU || a.out_kind == RAAE_OUT_STATS_RAW) {
                            const float v = (a.out_kind == RAAE_OUT_STATS_PRELU) ? prelu(zv, oslope) : zv;
                            s_acc += (double)v;
                            q_acc += (double)v * (double)v;
                        }
                    }
                }
            }
        }
    }
    DSTAMP(SLOT, 4);
    if (a.out_kind == RAAE_OUT_STATS_PRELU || a.out_kind == RAAE_OUT_STATS_RAW) {
        s_acc += __shfl_xor(s_acc, 16, 64); q_acc += __shfl_xor(q_acc, 16, 64);
        s_acc += __shfl_xor(s_acc, 32, 64); q_acc += __shfl_xor(q_acc, 32, 64);
        if (lane < 16 && col < a.N) {
            double* p = a.out_partials + ((size_t)bx * a.N + col) * 2;
            p[0] = s_acc; p[1] = q_acc;
        }
    }
    DSTAMP(SLOT, 5);
}

template <int KQ, int RT = 1, bool ST = false>
__device__ __forceinline__ void dense_fwd_body(const DenseFwdArgs& a, const int bx, const int by, const int gx, float* smem) {
    MaskSrc ms = mask_src(a.mask, a.in_kind, ST ? a.storage : 0, a.mask_scale, a.gen);
    dense_fwd_tiles<KQ, RT, ST>(a, bx, by, gx, smem, ms);
}

// (the by-value argument block goes to LDS in one coalesced vector load, raae::args_to_lds: scalar loads of a 200-300-byte
// struct out of a fresh kernarg buffer are a chain of dependent misses at the head of a 6-us kernel -- the fused block
// kernels have done this since round 1)
template <int KQ, int RT = 1, bool ST = false>
__global__ __launch_bounds__(256) void dense_fwd_kernel(DenseFwdArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ DenseFwdArgs sa;
    const DenseFwdArgs& a = raae::args_to_lds(&sa);
    dense_fwd_body<KQ, RT, ST>(a, blockIdx.x, blockIdx.y, gridDim.x, smem);
}

template <int KQ, int RT = 1, bool ST = false>
__global__ __launch_bounds__(256) void dense_fwd_kernel_m(const DenseFwdArgs* table) {     // one trial per grid plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ DenseFwdArgs sa;
    const DenseFwdArgs& a = raae::args_from_table(&sa, table);
    dense_fwd_body<KQ, RT, ST>(a, blockIdx.x, blockIdx.y, gridDim.x, smem);
}

// Two independent layers (one of the encoder, one of the decoder: the forward chain whose result the reference
// discards beside one that is needed) in ONE launch: workgroups [0, n1) run the first, the rest the second.
struct DenseFwd2Args { DenseFwdArgs x; DenseFwdArgs y; int n1; int gx1; int gx2; };
template <int Q1, int Q2>
__global__ __launch_bounds__(256, 1) void dense_fwd2_kernel(DenseFwd2Args k) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ DenseFwdArgs sa;
    int b = blockIdx.x;
    if (b < k.n1) {
        const DenseFwdArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(DenseFwd2Args, x));
        dense_fwd_body<Q1>(a, b % k.gx1, b / k.gx1, k.gx1, smem);
    } else {
        b -= k.n1;
        const DenseFwdArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(DenseFwd2Args, y));
        dense_fwd_body<Q2>(a, b % k.gx2, b / k.gx2, k.gx2, smem);
    }
}
template <int Q1, int Q2>
__global__ __launch_bounds__(256, 1) void dense_fwd2_kernel_m(const DenseFwd2Args* table) {     // one trial per grid plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ DenseFwdArgs sa;
    const DenseFwd2Args* k = table + blockIdx.z;
    const int n1 = k->n1, gx1 = k->gx1, gx2 = k->gx2;
    int b = blockIdx.x;
    if (b < n1) {
        const DenseFwdArgs& a = raae::args_from_ptr(&sa, &k->x);
        dense_fwd_body<Q1>(a, b % gx1, b / gx1, gx1, smem);
    } else {
        b -= n1;
        const DenseFwdArgs& a = raae::args_from_ptr(&sa, &k->y);
        dense_fwd_body<Q2>(a, b % gx2, b / gx2, gx2, smem);
    }
}

struct DenseBwdArgs {
    const float* g; int g_kind; const double* g_partials; int g_nparts; const float* zout;
    const float* out_slope; raae_bn_t out_bn; int B; int N;
    const float* x; int K; int in_kind; const float* slope; raae_bn_t bn; const float* mask; const float* w;
    float* dw; float* db; float* dslope; long slab_stride; float* dx; double* dx_partials;
    int pitch_g; int pitch_x; int storage;      // RAAE_ST_X: x, RAAE_ST_MASK: mask, RAAE_ST_Z: zout are bf16
    float mask_scale; raae_maskgen_t gen;
    int kw;       // input columns per workgroup: blockIdx.y owns columns [y kw, (y + 1) kw) of x, dW and dx (kw == K: all)
};

// Stage a 16-row tile of the layer input into LDS (transform applied, zero padded to K4 columns).
// Vector path (K % 4 == 0): one float4 per thread per pass, no integer division in the loop.
// (backward kernel; Kl columns from column k0 of rows of stride K; ms: where the dropout multipliers come from)
__device__ __forceinline__ void stage_rows(float* Xs, int pitch, const float* x, const MaskSrc& ms, int row0, int B,
                                           int K, int k0, int Kl, int K4, int in_kind, const float* s_slope,
                                           const float* s_mean, const float* s_rstd, int storage = 0) {
    const int tid = threadIdx.x;
    const bool xb = (storage & RAAE_ST_X) != 0;
    if ((Kl & 3) == 0 && (K & 3) == 0) {
        const int kq = Kl >> 2;                      // float4s per row
        int r = tid / kq, c4 = tid - r * kq;
        const int dr = 256 / kq, dc = 256 - dr * kq;
        for (; r < 16; ) {
            const int row = row0 + r, k = c4 << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < B) {
                const size_t o = (size_t)row * K + k0 + k;
                v = xb ? bf16x4_at(x, o) : *reinterpret_cast<const float4*>(x + o);
                v.x = in_transform(v.x, k, in_kind, s_slope, s_mean, s_rstd);
                v.y = in_transform(v.y, k + 1, in_kind, s_slope, s_mean, s_rstd);
                v.z = in_transform(v.z, k + 2, in_kind, s_slope, s_mean, s_rstd);
                v.w = in_transform(v.w, k + 3, in_kind, s_slope, s_mean, s_rstd);
                if (ms.any()) {
                    const float4 m = ms.at4(o);
                    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
                }
            }
            float2* dst = reinterpret_cast<float2*>(Xs + r * pitch + k);     // pitch even => 8-B aligned
            dst[0] = make_float2(v.x, v.y);
            dst[1] = make_float2(v.z, v.w);
            r += dr; c4 += dc;
            if (c4 >= kq) { c4 -= kq; ++r; }
        }
    } else {
        for (int idx = tid; idx < 16 * K4; idx += 256) {
            const int r = idx / K4, k = idx - r * K4;
            const int row = row0 + r;
            float v = 0.f;
            if (row < B && k < Kl) {
                const size_t o = (size_t)row * K + k0 + k;
                v = in_transform(xb ? bf16_at(x, o) : x[o], k, in_kind, s_slope, s_mean, s_rstd);
                if (ms.any()) v *= ms.at(o);
            }
            Xs[r * pitch + k] = v;
        }
    }
}


// TPW: 16x16 dW tiles per wave; KT4: 16-column dx tiles per wave.
// LDS: o_mean[N16] o_rstd[N16] o_slope[N16] m1[N16] m2[N16] | i_mean[K16] i_rstd[K16] i_slope[K16]
//      | Gs[16][pitch_g] | Xs[16][pitch_x] | red[2][256]
// (round 3: a rewrite that carried every operand one tile ahead in registers, kept the dx weights in registers and
// batched four row tiles measured SLOWER in the step -- 11.9 against 10.5 us for the 64 x 64 layer at 256 rows, 26.7
// against 24.3 at 4096 -- and is not kept; what is kept from it: the column slices of a first layer, dropout
// multipliers generated in the kernel, slopes requested before the statistic prologue.)
// A first layer (K = 256 / 512 input points, no input transform) is split over blockIdx.y in slices of kw columns: more
// workgroups for the launch-bound batches and a dW tile set of 8 tiles per wave (19.4 -> 12.0 us at 256 rows).
template <int TPW, int KT4, bool ST>
__device__ __forceinline__ void dense_bwd_body(const DenseBwdArgs& a, float* smem) {
    const int st = ST ? a.storage : 0;
    const int k0 = blockIdx.y * a.kw, Kl = a.kw;      // this workgroup's input columns [k0, k0 + Kl)
    const int N16 = (a.N + 15) & ~15, K16 = (Kl + 15) & ~15;
    MaskSrc ms = mask_src(a.mask, a.in_kind, st, a.mask_scale, a.gen);
    float* o_mean = smem;
    float* o_rstd = o_mean + N16;
    float* o_slope = o_rstd + N16;
    float* m1 = o_slope + N16;
    float* m2 = m1 + N16;
    float* i_mean = m2 + N16;
    float* i_rstd = i_mean + K16;
    float* i_slope = i_rstd + K16;
    float* Gs = i_slope + K16;
    float* Xs = Gs + 16 * a.pitch_g;
    float* red = Xs + 16 * a.pitch_x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NT = N16 >> 4, KT = K16 >> 4;

    const bool g_bn = a.g_kind == RAAE_G_PRELU_BN, i_bn = a.in_kind == RAAE_IN_PRELU_BN_DROP;
    if ((!g_bn || a.N <= 64) && (!i_bn || a.K <= 64)) {
        // the three statistic reductions side by side, one wave each (they used to run one after the other)
        const raae::StatJob jobs[3] = {
            g_bn ? raae::stat_job_bn(a.out_bn, a.N, o_mean, o_rstd, false) : raae::stat_job_none(),
            g_bn ? raae::stat_job_bwd(a.g_partials, a.g_nparts, a.N, a.out_bn.count, m1, m2) : raae::stat_job_none(),
            i_bn ? raae::stat_job_bn(a.bn, a.K, i_mean, i_rstd, false) : raae::stat_job_none()};
        raae::stat_jobs<3>(jobs, false);
    } else {
        if (g_bn) {
            raae::bn_prologue(a.out_bn, a.N, o_mean, o_rstd, false);
            raae::bnbwd_prologue(a.g_partials, a.g_nparts, a.N, a.out_bn.count, m1, m2);
        }
        if (i_bn) raae::bn_prologue(a.bn, a.K, i_mean, i_rstd, false);
    }
    if (a.g_kind == RAAE_G_PRELU_BN || a.g_kind == RAAE_G_PRELU)
        for (int n = tid; n < a.N; n += 256) o_slope[n] = a.out_slope[n];
    if (a.in_kind != RAAE_IN_NONE)
        for (int k = tid; k < Kl; k += 256) i_slope[k] = a.slope[k0 + k];
    ms.key();
    __syncthreads();

    f32x4 wacc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) wacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    double dxs[KT4], dxq[KT4];
#pragma unroll
    for (int i = 0; i < KT4; ++i) { dxs[i] = 0.0; dxq[i] = 0.0; }
    // G staging: thread owns column(s) n = tid (+256) when N > 64, else n = tid % 64 with row phase tid/64
    const bool wideN = a.N > 64;
    const int gcol0 = wideN ? tid : (tid & 63);
    const int grow0 = wideN ? 0 : (tid >> 6);
    const int grstep = wideN ? 1 : 4;
    double db_acc[2] = {0.0, 0.0}, ds_acc[2] = {0.0, 0.0};

    const bool zb = (st & RAAE_ST_Z) != 0;
    const int ntiles = (a.B + 15) >> 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile << 4;
        // ---- 1. dL/dz tile -> Gs (zero padded) ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = gcol0 + h * 256;
            if (h == 1 && !(wideN && n < N16)) break;
            if (n >= N16) continue;
            if constexpr (TPW == 16) {       // (not the 32-tile instance: 64 more live registers spill there)
                // Wide output layers (N > 64: the decoder's last layer, 64 -> 256): a thread owns a whole 16-row column
                // of the tile.  Row by row that was 16 dependent round trips per tile (25 us for the layer at 256 rows
                // against 11.5 us for a 64 x 64 layer); here the 32 operands are requested before the first is used.
                // Same values, same order of the column's bias / slope sums.
                if (wideN) {
                    float gq[16], zq[16];
                    const bool colok = n < a.N, needz = a.g_kind != RAAE_G_DIRECT;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + r;
                        const bool ok = colok && row < a.B;
                        const size_t o = (size_t)(ok ? row : 0) * a.N + (colok ? n : 0);
                        gq[r] = ok ? a.g[o] : 0.f;
                        zq[r] = (ok && needz) ? (zb ? bf16_at(a.zout, o) : a.zout[o]) : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float dz = 0.f;
                        if (colok && row0 + r < a.B) {
                            const float gv = gq[r], zv = zq[r];
                            if (a.g_kind == RAAE_G_DIRECT) dz = gv;
                            else if (a.g_kind == RAAE_G_SOFTPLUS) dz = gv * (1.f - expf(-2.f * zv));
                            else if (a.g_kind == RAAE_G_RELU) dz = zv > 0.f ? gv : 0.f;
                            else {
                                float da = gv;
                                if (a.g_kind == RAAE_G_PRELU_BN) {
                                    const float y = (prelu(zv, o_slope[n]) - o_mean[n]) * o_rstd[n];
                                    da = o_rstd[n] * (gv - m1[n] - y * m2[n]);
                                }
                                if (zv > 0.f) dz = da;
                                else { dz = da * o_slope[n]; ds_acc[h] += (double)da * (double)zv; }
                            }
                            db_acc[h] += (double)dz;
                        }
                        Gs[r * a.pitch_g + n] = dz;
                    }
                    continue;
                }
            }
            for (int r = grow0; r < 16; r += grstep) {
                const int row = row0 + r;
                float dz = 0.f;
                if (row < a.B && n < a.N) {
                    const size_t o = (size_t)row * a.N + n;
                    const float gv = a.g[o];
                    if (a.g_kind == RAAE_G_DIRECT) dz = gv;
                    else if (a.g_kind == RAAE_G_SOFTPLUS) dz = gv * (1.f - expf(-2.f * a.zout[o]));
                    else if (a.g_kind == RAAE_G_RELU) dz = a.zout[o] > 0.f ? gv : 0.f;
                    else {
                        const float zv = zb ? bf16_at(a.zout, o) : a.zout[o];
                        float da = gv;
                        if (a.g_kind == RAAE_G_PRELU_BN) {
                            const float y = (prelu(zv, o_slope[n]) - o_mean[n]) * o_rstd[n];
                            da = o_rstd[n] * (gv - m1[n] - y * m2[n]);
                        }
                        if (zv > 0.f) dz = da;
                        else { dz = da * o_slope[n]; ds_acc[h] += (double)da * (double)zv; }
                    }
                    db_acc[h] += (double)dz;
                }
                Gs[r * a.pitch_g + n] = dz;
            }
        }
        // ---- 2. layer input tile -> Xs (transform applied) ----
        stage_rows(Xs, a.pitch_x, a.x, ms, row0, a.B, a.K, k0, Kl, K16, a.in_kind, i_slope, i_mean, i_rstd, st);
        __syncthreads();
        // ---- 3. dW[n][k] += sum_rows dz[row][n] * xin[row][k]; wave owns tiles t = wv + 4 i ----
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wv + 4 * i;
            if (t < NT * KT) {
                const int tn = t / KT, tk = t - tn * KT;
                const float* ga = Gs + (lane >> 4) * a.pitch_g + tn * 16 + (lane & 15);
                const float* xb = Xs + (lane >> 4) * a.pitch_x + tk * 16 + (lane & 15);
#pragma unroll
                for (int r4 = 0; r4 < 16; r4 += 4)
                    wacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[r4 * a.pitch_g], xb[r4 * a.pitch_x], wacc[i], 0, 0, 0);
            }
        }
        // ---- 4. dx[row][k] = (sum_n dz[row][n] W[n][k]) * mask ; partial sums for the input's BN ----
        if (a.dx != nullptr) {
#pragma unroll
            for (int i = 0; i < KT4; ++i) {
                const int tk = wv + 4 * i;
                if (tk < KT) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const int kcol = tk * 16 + (lane & 15);
                    const float* ga = Gs + (lane & 15) * a.pitch_g + (lane >> 4);
                    const bool kok = kcol < Kl;
                    // B operand W[n][kcol]: 16 independent loads in flight per group of 16 MFMA steps
                    const float* wp = a.w + (size_t)(lane >> 4) * a.K + k0 + kcol;
                    int nn = 0;
                    for (; nn + 64 <= N16; nn += 64) {
                        float bb[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int n = nn + 4 * u + (lane >> 4);
                            bb[u] = (kok && n < a.N) ? wp[(size_t)(nn + 4 * u) * a.K] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[nn + 4 * u], bb[u], acc, 0, 0, 0);
                    }
                    for (; nn < N16; nn += 16) {
                        float bb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int n = nn + 4 * u + (lane >> 4);
                            bb[u] = (kok && n < a.N) ? wp[(size_t)(nn + 4 * u) * a.K] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[nn + 4 * u], bb[u], acc, 0, 0, 0);
                    }
                    if (kok) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int row = row0 + (lane >> 4) * 4 + j;
                            if (row < a.B) {
                                const size_t o = (size_t)row * a.K + k0 + kcol;
                                float d = acc[j];
                                if (ms.any()) d *= ms.at(o);
                                a.dx[o] = d;
                                if (a.dx_partials != nullptr) {
                                    const float xv = (st & RAAE_ST_X) ? bf16_at(a.x, o) : a.x[o];
                                    const float y = (prelu(xv, i_slope[kcol]) - i_mean[kcol]) * i_rstd[kcol];
                                    dxs[i] += (double)d;
                                    dxq[i] += (double)d * (double)y;
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- write this workgroup's slabs ----
    const size_t slab = (size_t)blockIdx.x * (size_t)a.slab_stride;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wv + 4 * i;
        if (t < NT * KT) {
            const int tn = t / KT, tk = t - tn * KT;
            const int k = tk * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tn * 16 + (lane >> 4) * 4 + j;
                if (n < a.N && k < Kl) a.dw[slab + (size_t)n * a.K + k0 + k] = wacc[i][j];
            }
        }
    }
    // db / dslope: combine the row-phase copies of each column in fixed order (the first column slice writes them)
    if (blockIdx.y != 0) {
    } else if (wideN) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = tid + h * 256;
            if (n < a.N) {
                a.db[slab + n] = (float)db_acc[h];
                if (a.dslope != nullptr) a.dslope[slab + n] = (float)ds_acc[h];
            }
        }
    } else {
        red[tid] = (float)db_acc[0];
        red[256 + tid] = (float)ds_acc[0];
        __syncthreads();
        if (tid < a.N) {
            a.db[slab + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
            if (a.dslope != nullptr)
                a.dslope[slab + tid] = (red[256 + tid] + red[320 + tid]) + (red[384 + tid] + red[448 + tid]);
        }
    }
    if (a.dx != nullptr && a.dx_partials != nullptr) {
#pragma unroll
        for (int i = 0; i < KT4; ++i) {
            const int tk = wv + 4 * i;
            double s = dxs[i], q = dxq[i];
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
            const int kcol = tk * 16 + (lane & 15);
            if (tk < KT && lane < 16 && kcol < Kl) {
                double* p = a.dx_partials + ((size_t)blockIdx.x * a.K + k0 + kcol) * 2;
                p[0] = s; p[1] = q;
            }
        }
    }
}
template <int TPW, int KT4, bool ST = false>
__global__ __launch_bounds__(256) void dense_bwd_kernel(DenseBwdArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ DenseBwdArgs sa;
    const DenseBwdArgs& a = raae::args_to_lds(&sa);       // (see dense_fwd_kernel)
    dense_bwd_body<TPW, KT4, ST>(a, smem);
}
template <int TPW, int KT4, bool ST = false>
__global__ __launch_bounds__(256) void dense_bwd_kernel_m(const DenseBwdArgs* table) {     // one trial per grid plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ DenseBwdArgs sa;
    const DenseBwdArgs& a = raae::args_from_table(&sa, table);
    dense_bwd_body<TPW, KT4, ST>(a, smem);
}

// Workgroups (= partial-statistic rows / gradient slabs) of a dense launch over B rows.  Every consumer reduces all
// partial rows in its prologue, in every workgroup: rows x workgroups grows with the square of the grid, so large
// batches use FEWER, longer-running workgroups -- 16-row tiles one after the other, the next tile's loads in flight
// during the current tile's matrix work (256 workgroups of one tile each at 4096 rows meant 256 KB of partials per
// workgroup: 15 of the 20 us of a 64-wide layer).
// rt: 16-row tiles per pass of a workgroup (1, or 4 for batches of >= 2048 rows where the instance exists)
int pick_rt(int B) { return B >= 2048 ? 4 : 1; }
int pick_grid(int B, int rt = 1) {
    const int ntiles = (B + 16 * rt - 1) / (16 * rt);
    int cap = ntiles / 8;
    if (cap > RAAE_MAX_PARTS) cap = RAAE_MAX_PARTS;
    if (cap < 64) cap = 64;
    return ntiles < cap ? ntiles : cap;
}

}  // namespace

static int prep_dense_fwd(const float* x, int B, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                          const float* mask, const float* w, const float* bias, int N, float* z, int out_kind,
                          const float* out_slope, double* out_partials, DenseFwdArgs& a, dim3& grid, size_t& lds, int& kq,
                          bool allow_rt = true) {
    RAAE_CHECK_ARG(x && w && bias && z && B > 0 && K > 0 && N > 0 && K <= 512);
    RAAE_CHECK_ARG(in_kind >= 0 && in_kind <= 2 && out_kind >= 0 && out_kind <= 4);
    RAAE_CHECK_ARG(in_kind == RAAE_IN_NONE || slope);
    RAAE_CHECK_ARG(in_kind != RAAE_IN_PRELU_BN_DROP || (bn && (bn->partials || (bn->running_mean && bn->running_var))));
    RAAE_CHECK_ARG(!(out_kind == RAAE_OUT_STATS_PRELU) || out_slope);
    RAAE_CHECK_ARG(!(out_kind == RAAE_OUT_STATS_PRELU || out_kind == RAAE_OUT_STATS_RAW) || out_partials);
    a.x = x; a.B = B; a.K = K; a.in_kind = in_kind; a.slope = slope; a.mask = mask;
    if (bn) a.bn = *bn; else { raae_bn_t z0 = {}; a.bn = z0; }
    RAAE_CHECK_ARG(a.bn.nparts >= 0 && a.bn.nparts <= RAAE_MAX_PARTS);
    a.w = w; a.bias = bias; a.N = N; a.z = z; a.out_kind = out_kind; a.out_slope = out_slope;
    a.out_partials = out_partials;
    a.storage = 0;
    a.mask_scale = 1.f;
    a.gen.keys = nullptr; a.gen.offset = 0; a.gen.thr = 0xFFFFFFFFu; a.gen.inv = 1.f;
    const int K4 = (K + 3) & ~3;
    RAAE_CHECK_ARG(K4 <= 512 && (in_kind != RAAE_IN_PRELU_BN_DROP || K <= 256));
    a.pitch = K4 + 2;
    kq = K4 <= 16 ? 4 : K4 <= 64 ? 16 : K4 <= 256 ? 64 : 128;
    // 64-row passes for the large batches (not the 512-column layer: its tile would not fit the registers)
    const int rt = (allow_rt && kq <= 64) ? pick_rt(B) : 1;
    lds = sizeof(float) * (3 * (size_t)K4 + 16 * (size_t)rt * (size_t)a.pitch + (kq <= 64 ? 64 * ((size_t)K4 + 4) : 0));
    grid = dim3(pick_grid(B, rt), (N + 63) / 64, rt);       // (grid.z carries rt to launch_dense_fwd; launched with z = 1)
    return 0;
}

static void launch_dense_fwd(const DenseFwdArgs& a, dim3 grid, size_t lds, int kq, hipStream_t st) {
    const int rt = (int)grid.z;
    grid.z = 1;
#define RAAE_FWD(KQ_, RT_) do { if (a.storage) raae::launch(dense_fwd_kernel<KQ_, RT_, true>, dense_fwd_kernel_m<KQ_, RT_, true>, grid, dim3(256), lds, st, a); \
                               else raae::launch(dense_fwd_kernel<KQ_, RT_, false>, dense_fwd_kernel_m<KQ_, RT_, false>, grid, dim3(256), lds, st, a); } while (0)
    // bf16 storage (a.storage): the hidden layers of the dense networks (K <= 64 -> KQ 16; first layer KQ 64 / 128)
    if (rt == 4) {
        if (kq == 4) RAAE_FWD(4, 4);
        else if (kq == 16) RAAE_FWD(16, 4);
        else RAAE_FWD(64, 4);
    } else {
        if (kq == 4) RAAE_FWD(4, 1);
        else if (kq == 16) RAAE_FWD(16, 1);
        else if (kq == 64) RAAE_FWD(64, 1);
        else RAAE_FWD(128, 1);
    }
#undef RAAE_FWD
}

extern "C" int raae_dense_fwd(const float* x, int B, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                              const float* mask, const float* w, const float* bias, int N, float* z, int out_kind,
                              const float* out_slope, double* out_partials, int* out_nparts, void* stream) {
    DenseFwdArgs a;
    dim3 grid;
    size_t lds;
    int kq;
    const int rc = prep_dense_fwd(x, B, K, in_kind, slope, bn, mask, w, bias, N, z, out_kind, out_slope, out_partials,
                                  a, grid, lds, kq);
    if (rc) return rc;
    if (out_nparts) *out_nparts = (int)grid.x;
    launch_dense_fwd(a, grid, lds, kq, (hipStream_t)stream);
    RAAE_LAUNCH_RET();
}

// struct form of raae_dense_fwd; honours p->storage (bf16 storage of x / mask / z)
extern "C" int raae_dense_fwd_s(const raae_dense_fwd_t* p, int* out_nparts, void* stream) {
    RAAE_CHECK_ARG(p && (p->storage & ~(RAAE_ST_X | RAAE_ST_MASK | RAAE_ST_Z)) == 0);
    RAAE_CHECK_ARG(!(p->storage & RAAE_ST_X) || (p->K & 3) == 0);
    DenseFwdArgs a;
    dim3 grid;
    size_t lds;
    int kq;
    const int rc = prep_dense_fwd(p->x, p->B, p->K, p->in_kind, p->slope, p->has_bn ? &p->bn : nullptr, p->mask, p->w,
                                  p->bias, p->N, p->z, p->out_kind, p->out_slope, p->out_partials, a, grid, lds, kq);
    if (rc) return rc;
    RAAE_CHECK_ARG(!(p->gen.keys && p->mask) && !(p->gen.keys && !(p->gen.inv >= 1.f)));
    a.storage = p->storage; a.mask_scale = p->mask_scale; a.gen = p->gen;
    if (out_nparts) *out_nparts = (int)grid.x;
    launch_dense_fwd(a, grid, lds, kq, (hipStream_t)stream);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_dense_fwd2(const raae_dense_fwd_t* p, const raae_dense_fwd_t* q, int* nparts_p, int* nparts_q,
                               void* stream) {
    RAAE_CHECK_ARG(p && q);
    DenseFwd2Args k;
    dim3 g1, g2;
    size_t l1, l2;
    int q1, q2;
    int rc = prep_dense_fwd(p->x, p->B, p->K, p->in_kind, p->slope, p->has_bn ? &p->bn : nullptr, p->mask, p->w, p->bias,
                            p->N, p->z, p->out_kind, p->out_slope, p->out_partials, k.x, g1, l1, q1, false);
    if (rc) return rc;
    rc = prep_dense_fwd(q->x, q->B, q->K, q->in_kind, q->slope, q->has_bn ? &q->bn : nullptr, q->mask, q->w, q->bias,
                        q->N, q->z, q->out_kind, q->out_slope, q->out_partials, k.y, g2, l2, q2, false);
    if (rc) return rc;
    RAAE_CHECK_ARG(!(p->gen.keys && p->mask) && !(q->gen.keys && q->mask));
    k.x.storage = p->storage; k.y.storage = q->storage;
    k.x.mask_scale = p->mask_scale; k.y.mask_scale = q->mask_scale;
    k.x.gen = p->gen; k.y.gen = q->gen;
    if (nparts_p) *nparts_p = (int)g1.x;
    if (nparts_q) *nparts_q = (int)g2.x;
    hipStream_t st = (hipStream_t)stream;
    k.n1 = (int)(g1.x * g1.y); k.gx1 = (int)g1.x; k.gx2 = (int)g2.x;
    const dim3 grid(k.n1 + g2.x * g2.y);
    const size_t lds = l1 > l2 ? l1 : l2;
    g1.z = 1; g2.z = 1;
    // instances: the layer pairs of the 256-point dense networks (first layers 256 -> 64 beside 6 -> 64, then 64-wide
    // layers beside each other); anything else: two launches
    if (!(p->storage | q->storage) && q1 == 64 && q2 == 4) raae::launch(dense_fwd2_kernel<64, 4>, dense_fwd2_kernel_m<64, 4>, grid, dim3(256), lds, st, k);
    else if (!(p->storage | q->storage) && q1 == 16 && q2 == 16) raae::launch(dense_fwd2_kernel<16, 16>, dense_fwd2_kernel_m<16, 16>, grid, dim3(256), lds, st, k);
    else {
        launch_dense_fwd(k.x, g1, l1, q1, st);
        launch_dense_fwd(k.y, g2, l2, q2, st);
    }
    RAAE_LAUNCH_RET();
}

extern "C" int raae_dense_bwd(const float* g, int g_kind, const double* g_partials, int g_nparts, const float* zout,
                              const float* out_slope, const raae_bn_t* out_bn, int B, int N,
                              const float* x, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                              const float* mask, const float* w,
                              float* dw, float* db, float* dslope, long slab_stride, int* nslab,
                              float* dx, double* dx_partials, void* stream) {
    return raae_dense_bwd_st(g, g_kind, g_partials, g_nparts, zout, out_slope, out_bn, B, N, x, K, in_kind, slope, bn, mask, w,
                             dw, db, dslope, slab_stride, nslab, dx, dx_partials, 0, stream);
}

extern "C" int raae_dense_bwd_st(const float* g, int g_kind, const double* g_partials, int g_nparts, const float* zout,
                                 const float* out_slope, const raae_bn_t* out_bn, int B, int N,
                                 const float* x, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                                 const float* mask, const float* w,
                                 float* dw, float* db, float* dslope, long slab_stride, int* nslab,
                                 float* dx, double* dx_partials, int storage, void* stream) {
    raae_dense_bwd_t p = {};
    p.g = g; p.g_kind = g_kind; p.g_partials = g_partials; p.g_nparts = g_nparts; p.zout = zout; p.out_slope = out_slope;
    p.has_out_bn = out_bn ? 1 : 0;
    if (out_bn) p.out_bn = *out_bn;
    p.B = B; p.N = N; p.x = x; p.K = K; p.in_kind = in_kind; p.slope = slope;
    p.has_bn = bn ? 1 : 0;
    if (bn) p.bn = *bn;
    p.mask = mask; p.w = w; p.dw = dw; p.db = db; p.dslope = dslope; p.slab_stride = slab_stride; p.dx = dx;
    p.dx_partials = dx_partials; p.storage = storage; p.mask_scale = 1.f;
    return raae_dense_bwd_s(&p, nslab, stream);
}

extern "C" int raae_dense_bwd_s(const raae_dense_bwd_t* p, int* nslab, void* stream) {
    RAAE_CHECK_ARG(p);
    const int storage = p->storage, B = p->B, N = p->N, K = p->K, g_kind = p->g_kind, in_kind = p->in_kind;
    const raae_bn_t* out_bn = p->has_out_bn ? &p->out_bn : nullptr;
    const raae_bn_t* bn = p->has_bn ? &p->bn : nullptr;
    RAAE_CHECK_ARG((storage & ~(RAAE_ST_X | RAAE_ST_MASK | RAAE_ST_Z)) == 0 && (!(storage & RAAE_ST_X) || (K & 3) == 0));
    RAAE_CHECK_ARG(p->g && p->x && p->w && p->dw && p->db && B > 0 && N > 0 && K > 0 && N <= 512 && K <= 512);
    RAAE_CHECK_ARG(g_kind >= 0 && g_kind <= 4 && in_kind >= 0 && in_kind <= 2);
    RAAE_CHECK_ARG(g_kind == RAAE_G_DIRECT || p->zout);
    RAAE_CHECK_ARG(!(g_kind == RAAE_G_PRELU_BN) || (p->g_partials && out_bn && p->out_slope && p->g_nparts > 0 && p->g_nparts <= RAAE_MAX_PARTS));
    RAAE_CHECK_ARG(!(g_kind == RAAE_G_PRELU) || p->out_slope);
    RAAE_CHECK_ARG(in_kind == RAAE_IN_NONE || p->slope);
    RAAE_CHECK_ARG(in_kind != RAAE_IN_PRELU_BN_DROP || bn);
    RAAE_CHECK_ARG(!(p->dx && in_kind == RAAE_IN_PRELU_BN_DROP) || p->dx_partials);
    RAAE_CHECK_ARG(!(p->gen.keys && p->mask) && !(p->gen.keys && !(p->gen.inv >= 1.f)));
    DenseBwdArgs a;
    a.g = p->g; a.g_kind = g_kind; a.g_partials = p->g_partials; a.g_nparts = p->g_nparts; a.zout = p->zout;
    a.out_slope = p->out_slope;
    raae_bn_t z0 = {};
    a.out_bn = out_bn ? *out_bn : z0;
    a.B = B; a.N = N; a.x = p->x; a.K = K; a.in_kind = in_kind; a.slope = p->slope;
    a.bn = bn ? *bn : z0;
    RAAE_CHECK_ARG(a.bn.nparts <= RAAE_MAX_PARTS && a.out_bn.nparts <= RAAE_MAX_PARTS);
    a.mask = p->mask; a.w = p->w; a.dw = p->dw; a.db = p->db; a.dslope = p->dslope; a.slab_stride = p->slab_stride;
    a.dx = p->dx; a.dx_partials = (in_kind == RAAE_IN_PRELU_BN_DROP) ? p->dx_partials : nullptr;
    // first layers (no input transform, hence no statistics over K) with 256 / 512 input points: slices of 128 columns
    const bool split = in_kind == RAAE_IN_NONE && N <= 64 && K >= 256 && (K % 128) == 0;
    a.kw = split ? 128 : K;
    const int N16 = (N + 15) & ~15, K16 = (a.kw + 15) & ~15;
    a.pitch_g = N16 + 2; a.pitch_x = K16 + 2;
    a.storage = storage; a.mask_scale = p->mask_scale; a.gen = p->gen;
    const int tiles = (N16 / 16) * (K16 / 16);
    const int tpw = (tiles + 3) / 4, kt4 = (K16 / 16 + 3) / 4;
    const size_t lds = sizeof(float) * (5 * (size_t)N16 + 3 * (size_t)K16 + 16 * (size_t)(a.pitch_g + a.pitch_x) + 512);
    RAAE_CHECK_ARG(lds <= 160 * 1024);
    // workgroups = gradient slabs = partial rows of dx: few and long-running at large batches (pick_grid).  Measured at
    // 4096 rows, dense networks: 64 workgroups of four tiles each make THIS kernel slower (31.7 against 24.3 us for the
    // 64 x 64 layer with 256 one-tile workgroups) and the step faster (408-442 against 341-356 steps/s): every consumer
    // of its partial rows sweeps a quarter of them.  Cap the slabs of large layers: the Adam kernel re-reads each.
    int gx = pick_grid(B);
    if ((long)N * K >= 8192 && gx > 64) gx = 64;
    if (nslab) *nslab = gx;
    dim3 grid(gx, K / a.kw), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool need_dx = p->dx != nullptr;
#define RAAE_BWD(TPW_, KT4_) do { if (storage) raae::launch(dense_bwd_kernel<TPW_, KT4_, true>, dense_bwd_kernel_m<TPW_, KT4_, true>, grid, block, lds, st, a); \
                                 else raae::launch(dense_bwd_kernel<TPW_, KT4_, false>, dense_bwd_kernel_m<TPW_, KT4_, false>, grid, block, lds, st, a); } while (0)
    if (tpw <= 1 && kt4 <= 1) RAAE_BWD(1, 1);
    else if (tpw <= 4 && kt4 <= 1) RAAE_BWD(4, 1);
    else if (tpw <= 8 && kt4 <= 2) RAAE_BWD(8, 2);               // a 128-column slice of a first layer
    else if (tpw <= 16 && (kt4 <= 1 || !need_dx)) RAAE_BWD(16, 1);
    else if (tpw <= 16 && kt4 <= 4) RAAE_BWD(16, 4);
    else if (tpw <= 32 && (kt4 <= 1 || !need_dx)) RAAE_BWD(32, 1);
    else if (tpw <= 32 && kt4 <= 8) RAAE_BWD(32, 8);
    else return RAAE_EINVAL;
#undef RAAE_BWD
    RAAE_LAUNCH_RET();
}
