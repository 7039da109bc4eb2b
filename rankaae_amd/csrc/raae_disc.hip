// The whole adversarial branch of one step in ONE launch: discriminator input (Gaussian-prior samples and encoder
// styles, plus input noise), the three Linear / PReLU / Dropout layers of DiscriminatorFC forward
// (sc/clustering/model.py:631-663), the two-sided BCE-with-logits loss (sc/utils/functions.py:109-132), the
// backward of all three layers with their weight / bias / PReLU-slope gradients, and the reversed gradient
// -alpha * dL/dx of the encoder styles (GradientReversalLayer, model.py:8-22).
//
// The discriminator has no BatchNorm, so its rows only meet in the loss mean and in the parameter-gradient sums:
// a workgroup takes tiles of 16 rows through forward and backward entirely in LDS and writes one slab of
// parameter gradients (summed by raae_adam_step like every other slab) and one partial of the loss; the last
// workgroup to finish (ticket) adds the partials in index order.  This replaces nine launches of the per-layer path
// (raae_disc_input, 3 x raae_dense_fwd, raae_bce_pair_fwd_bwd, 3 x raae_dense_bwd, raae_scale_by_dev), which at
// 256-row batches cost ~50 us for ~7 MFLOP.  Instance: hidden width 64, three layers, nstyle <= 16.
#include "raae_common.h"
#include <stdlib.h>

namespace {

constexpr int DH = 64;      // hidden width
constexpr int DT = 16;      // rows per tile
constexpr int DNS = 16;     // max nstyle (row pitch of the input tile)
constexpr int AP = DH + 1;  // row pitch of the activation tiles (the four rows a wave touches sit in different banks)

struct DiscArgs {
    const float* z_real; const float* styles; const float* noise; float sigma;
    const float* m1; const float* m2;                       // dropout multipliers [n][64] or NULL
    const float* w1; const float* b1; const float* s1;      // [64][ns], [64], [64]
    const float* w2; const float* b2; const float* s2;      // [64][64], [64], [64]
    const float* w3; const float* b3;                       // [1][64], [1]
    const float* alpha;                                     // device scalar of the gradient reversal
    int n_real, n_fake, ns;
    float* dw1; float* db1; float* ds1; float* dw2; float* db2; float* ds2; float* dw3; float* db3;   // slab 0
    long slab_stride;
    float* dstyles;                                         // [n_fake][ns]
    double* partial; unsigned* ticket; float* loss;
};

__device__ __forceinline__ double softplus_dd(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }

// MM: the three 64 x 64 contractions of a tile (layer 2 forward, dW2 += G2^T A1, dA1 = G2 W2) run on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32; operand a = A[row = lane & 15][k = lane >> 4], b = B[k = lane >> 4][col =
// lane & 15], result c[j] = C[row = 4 (lane >> 4) + j][col = lane & 15]); each wave owns 16 of the 64 output columns
// (rows of dW2).  Used from 2048 rows up, where the discriminator is a real contraction (BASELINE configs[2]:
// 8192 x 64 x 64 per layer); below that the step is launch-bound and the VALU form keeps its thread mapping from
// load to epilogue without the extra LDS round trip.
template <bool MM>
__device__ __forceinline__ void disc_fused_body(const DiscArgs& a) {
    __shared__ float W1[DH * DNS], W2[DH * (DH + 1)], W3[DH], B1[DH], B2[DH], S1[DH], S2[DH];
    __shared__ float X[DT * DNS], Z1[DT * AP], A1[DT * AP], Z2[DT * AP], A2[DT * AP], G2[DT * AP], G1[DT * AP];
    __shared__ float DL[DT];
    __shared__ double shd[16];
    __shared__ unsigned s_last;
    const int t = threadIdx.x, ns = a.ns, n = a.n_real + a.n_fake;
    // weights: W1 [o][k] pitch DNS (zero padded), W2 [o][k] pitch 65 (both the row- and the column-wise walk are
    // conflict free)
    for (int i = t; i < DH * DNS; i += 256) { const int o = i / DNS, k = i - o * DNS; W1[i] = k < ns ? a.w1[o * ns + k] : 0.f; }
    for (int i = t; i < DH * DH; i += 256) { const int o = i >> 6, k = i & 63; W2[o * (DH + 1) + k] = a.w2[i]; }
    if (t < DH) { W3[t] = a.w3[t]; B1[t] = a.b1[t]; B2[t] = a.b2[t]; S1[t] = a.s1[t]; S2[t] = a.s2[t]; }
    const float b3 = a.b3[0];
    const float neg_alpha = -a.alpha[0];
    // per-thread accumulators of the parameter gradients over this workgroup's tiles
    float aw2[16], aw1[4];
    f32x4 mw2[4];                                         // MM: dW2 tiles (rows 16 wave + 4 (lane >> 4) + j, columns 16 tk + (lane & 15))
#pragma unroll
    for (int i = 0; i < 4; ++i) mw2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) aw2[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) aw1[i] = 0.f;
    float ab1 = 0.f, ab2 = 0.f, as1 = 0.f, as2 = 0.f, aw3 = 0.f, ab3 = 0.f;
    double lacc = 0.0;
    const int ntiles = (n + DT - 1) / DT;
    const int r4 = t >> 4, c4 = (t & 15) * 4;             // [16 x 64] outputs: row r4, columns c4 .. c4+3
    const int wo = t >> 2, wk = (t & 3) * 16;             // dW2: row wo, columns wk .. wk+15
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * DT;
        // dropout multipliers of this thread's four columns of both layers: requested up front, used forward and backward
        float m1v[4] = {1.f, 1.f, 1.f, 1.f}, m2v[4] = {1.f, 1.f, 1.f, 1.f};
        if (row0 + r4 < n) {
            if (a.m1) { const float4 v = *reinterpret_cast<const float4*>(a.m1 + (size_t)(row0 + r4) * DH + c4); m1v[0] = v.x; m1v[1] = v.y; m1v[2] = v.z; m1v[3] = v.w; }
            if (a.m2) { const float4 v = *reinterpret_cast<const float4*>(a.m2 + (size_t)(row0 + r4) * DH + c4); m2v[0] = v.x; m2v[1] = v.y; m2v[2] = v.z; m2v[3] = v.w; }
        }
        // ---- input tile: prior sample / style (+ sigma * noise)
        for (int i = t; i < DT * DNS; i += 256) {
            const int r = i / DNS, k = i - r * DNS, row = row0 + r;
            float v = 0.f;
            if (row < n && k < ns) {
                v = row < a.n_real ? a.z_real[(size_t)row * ns + k] : a.styles[(size_t)(row - a.n_real) * ns + k];
                if (a.noise) v += a.sigma * a.noise[(size_t)row * ns + k];
            }
            X[i] = v;
        }
        __syncthreads();
        // ---- layer 1: Z1 = X W1^T + b1 ; A1 = PReLU(Z1) * mask1
        {
            const int row = row0 + r4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = c4 + j;
                float acc = B1[o];
                for (int k = 0; k < ns; ++k) acc = fmaf(X[r4 * DNS + k], W1[o * DNS + k], acc);
                Z1[r4 * AP + o] = acc;
                float v = acc > 0.f ? acc : acc * S1[o];
                v *= m1v[j];
                A1[r4 * AP + o] = row < n ? v : 0.f;
            }
        }
        __syncthreads();
        // ---- layer 2
        {
            const int row = row0 + r4;
            float acc[4] = {B2[c4], B2[c4 + 1], B2[c4 + 2], B2[c4 + 3]};
            if constexpr (MM) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                const float* pa = A1 + l15 * AP + l4;
                const float* pb = W2 + (16 * wave + l15) * (DH + 1) + l4;
#pragma unroll
                for (int q = 0; q < DH / 4; ++q) c = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * q], pb[4 * q], c, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) Z2[(4 * l4 + j) * AP + 16 * wave + l15] = c[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += Z2[r4 * AP + c4 + j];
            } else {
#pragma unroll 4
                for (int k = 0; k < DH; ++k) {
                    const float x = A1[r4 * AP + k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(x, W2[(c4 + j) * (DH + 1) + k], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = c4 + j;
                Z2[r4 * AP + o] = acc[j];
                float v = acc[j] > 0.f ? acc[j] : acc[j] * S2[o];
                v *= m2v[j];
                A2[r4 * AP + o] = row < n ? v : 0.f;
            }
        }
        __syncthreads();
        // ---- layer 3 + loss: 16 lanes per row
        {
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) p = fmaf(A2[r4 * AP + c4 + j], W3[c4 + j], p);
            p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 8, 64);
            const int row = row0 + r4;
            if ((t & 15) == 0) {
                float d = 0.f;
                if (row < n) {
                    const float o = p + b3;
                    const float sg = 1.f / (1.f + expf(-o));
                    if (row < a.n_real) { lacc += softplus_dd(-(double)o) / (double)a.n_real; d = (sg - 1.f) / (float)a.n_real; }
                    else { lacc += softplus_dd((double)o) / (double)a.n_fake; d = sg / (float)a.n_fake; }
                }
                DL[r4] = d;
            }
        }
        __syncthreads();
        // ---- backward of layer 3: dA2 = dlogit * w3 ; through mask2 and PReLU2 -> G2 = dZ2
        {
            const float d = DL[r4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = c4 + j;
                float g = d * W3[o];
                g *= m2v[j];
                const float z = Z2[r4 * AP + o];
                G2[r4 * AP + o] = z > 0.f ? g : g * S2[o];
                Z2[r4 * AP + o] = z > 0.f ? 0.f : g * z;   // slope-gradient term of this element
            }
        }
        __syncthreads();
        if (t < DH) {
            float w = 0.f, b = 0.f, s = 0.f;
#pragma unroll 4
            for (int r = 0; r < DT; ++r) { w = fmaf(A2[r * AP + t], DL[r], w); b += G2[r * AP + t]; s += Z2[r * AP + t]; }
            aw3 += w; ab2 += b; as2 += s;
        }
        if (t == 64) { float b = 0.f; for (int r = 0; r < DT; ++r) b += DL[r]; ab3 += b; }
        // dW2[o][k] += sum_r G2[r][o] A1[r][k]
        if constexpr (MM) {
            // A = G2^T: a = G2[r = 4 q + (lane >> 4)][o = 16 wave + (lane & 15)], b = A1[r][k = 16 tk + (lane & 15)]
#pragma unroll
            for (int q = 0; q < DT / 4; ++q) {
                const float ga = G2[(4 * q + l4) * AP + 16 * wave + l15];
#pragma unroll
                for (int tk = 0; tk < 4; ++tk)
                    mw2[tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, A1[(4 * q + l4) * AP + 16 * tk + l15], mw2[tk], 0, 0, 0);
            }
        } else {
#pragma unroll 2
            for (int r = 0; r < DT; ++r) {
                const float g = G2[r * AP + wo];
#pragma unroll
                for (int j = 0; j < 16; ++j) aw2[j] = fmaf(g, A1[r * AP + wk + j], aw2[j]);
            }
        }
        // dA1 = G2 W2 ; through mask1 and PReLU1 -> G1 = dZ1 (Z1 turns into the slope-gradient terms)
        {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MM) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                const float* pa = G2 + l15 * AP + l4;                       // A[row r][k = o]
                const float* pb = W2 + l4 * (DH + 1) + 16 * wave + l15;     // B[k = o][col = input feature]
#pragma unroll
                for (int q = 0; q < DH / 4; ++q)
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * q], pb[4 * q * (DH + 1)], c, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) G1[(4 * l4 + j) * AP + 16 * wave + l15] = c[j];
                __syncthreads();                         // dA1 tile complete; dW2 above has read A1
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = G1[r4 * AP + c4 + j];
            } else {
#pragma unroll 4
                for (int o = 0; o < DH; ++o) {
                    const float g = G2[r4 * AP + o];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(g, W2[o * (DH + 1) + c4 + j], acc[j]);
                }
            }
            __syncthreads();                             // dW2 above still reads A1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = c4 + j;
                float g = acc[j];
                g *= m1v[j];
                const float z = Z1[r4 * AP + k];
                G1[r4 * AP + k] = z > 0.f ? g : g * S1[k];
                Z1[r4 * AP + k] = z > 0.f ? 0.f : g * z;
            }
        }
        __syncthreads();
        if (t < DH) {
            float b = 0.f, s = 0.f;
            for (int r = 0; r < DT; ++r) { b += G1[r * AP + t]; s += Z1[r * AP + t]; }
            ab1 += b; as1 += s;
        }
        // dW1[o][k] += sum_r G1[r][o] X[r][k]   (64 * ns <= 1024 elements: up to 4 per thread)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = t + 256 * j;
            if (idx < DH * ns) {
                const int o = idx / ns, k = idx - o * ns;
                float w = 0.f;
#pragma unroll 4
                for (int r = 0; r < DT; ++r) w = fmaf(G1[r * AP + o], X[r * DNS + k], w);
                aw1[j] += w;
            }
        }
        // dX = G1 W1 ; the fake rows leave as -alpha * dX (gradient reversal)
        for (int i = t; i < DT * ns; i += 256) {
            const int r = i / ns, k = i - r * ns, row = row0 + r;
            if (row >= a.n_real && row < n) {
                float d = 0.f;
#pragma unroll 4
                for (int o = 0; o < DH; ++o) d = fmaf(G1[r * AP + o], W1[o * DNS + k], d);
                a.dstyles[(size_t)(row - a.n_real) * ns + k] = neg_alpha * d;
            }
        }
        __syncthreads();
    }
    // ---- this workgroup's slab of parameter gradients
    const size_t slab = (size_t)blockIdx.x * (size_t)a.slab_stride;
    if constexpr (MM) {
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int j = 0; j < 4; ++j) a.dw2[slab + (16 * wave + 4 * l4 + j) * DH + 16 * tk + l15] = mw2[tk][j];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) a.dw2[slab + wo * DH + wk + j] = aw2[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int idx = t + 256 * j; if (idx < DH * ns) a.dw1[slab + idx] = aw1[j]; }
    if (t < DH) {
        a.db1[slab + t] = ab1; a.ds1[slab + t] = as1; a.db2[slab + t] = ab2; a.ds2[slab + t] = as2; a.dw3[slab + t] = aw3;
    }
    if (t == 64) a.db3[slab] = ab3;
    // ---- loss: partial per workgroup, summed in index order by the last one to arrive
    const double tot = raae::block_sum(lacc, shd);
    if (t == 0) {
        a.partial[blockIdx.x] = tot;
        __threadfence();
        s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {                             // uniform per workgroup: all its threads add the partials (fixed-order tree)
        __threadfence();
        double sum = 0.0;
        for (unsigned i = t; i < gridDim.x; i += 256) sum += ((volatile double*)a.partial)[i];
        sum = raae::block_sum(sum, shd);
        if (t == 0) {
            a.loss[0] = (float)sum;
            *a.ticket = 0u;
        }
    }
}
template <bool MM>
__global__ __launch_bounds__(256) void disc_fused_kernel(DiscArgs a) { disc_fused_body<MM>(a); }
template <bool MM>
__global__ __launch_bounds__(256) void disc_fused_kernel_m(const DiscArgs* t) {     // one trial per grid plane (raae_common.h)
    const DiscArgs a = t[blockIdx.z];
    disc_fused_body<MM>(a);
}

}  // namespace

extern "C" int raae_disc_fused(const raae_disc_fused_t* in, int* nslab, void* stream) {
    RAAE_CHECK_ARG(in && in->z_real && in->styles && in->w1 && in->b1 && in->s1 && in->w2 && in->b2 && in->s2 && in->w3 &&
                   in->b3 && in->alpha && in->n_real > 0 && in->n_fake > 0 && in->ns >= 1 && in->ns <= DNS);
    RAAE_CHECK_ARG(in->hidden == DH && in->dw1 && in->db1 && in->ds1 && in->dw2 && in->db2 && in->ds2 && in->dw3 && in->db3 &&
                   in->dstyles && in->partial && in->ticket && in->loss);
    DiscArgs a;
    a.z_real = in->z_real; a.styles = in->styles; a.noise = in->noise; a.sigma = in->sigma; a.m1 = in->mask1; a.m2 = in->mask2;
    a.w1 = in->w1; a.b1 = in->b1; a.s1 = in->s1; a.w2 = in->w2; a.b2 = in->b2; a.s2 = in->s2; a.w3 = in->w3; a.b3 = in->b3;
    a.alpha = in->alpha; a.n_real = in->n_real; a.n_fake = in->n_fake; a.ns = in->ns;
    a.dw1 = in->dw1; a.db1 = in->db1; a.ds1 = in->ds1; a.dw2 = in->dw2; a.db2 = in->db2; a.ds2 = in->ds2;
    a.dw3 = in->dw3; a.db3 = in->db3; a.slab_stride = in->slab_stride; a.dstyles = in->dstyles;
    a.partial = in->partial; a.ticket = in->ticket; a.loss = in->loss;
    const int ntiles = (in->n_real + in->n_fake + DT - 1) / DT;
    const int grid = ntiles < 256 ? ntiles : 256;
    if (nslab) *nslab = grid;
    // from 2048 rows (real + fake) the 64 x 64 layers go through the matrix cores; RAAE_DISC_MFMA=0/1 forces a form
    static const int force = [] { const char* e = getenv("RAAE_DISC_MFMA"); return e ? atoi(e) : -1; }();
    const bool mm = force >= 0 ? force != 0 : (in->n_real + in->n_fake >= 2048);
    if (mm) raae::launch(disc_fused_kernel<true>, disc_fused_kernel_m<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else raae::launch(disc_fused_kernel<false>, disc_fused_kernel_m<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}
