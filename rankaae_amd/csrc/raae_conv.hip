// 1-D convolutional encoder/decoder kernels (ae_form: compact): Conv1d / ConvTranspose1d /
// length-axis Linear / block sum, forward and backward, over "views" (raw tensor + PReLU +
// BatchNorm + dropout scale applied on load) -- reference sc/clustering/model.py:24-174, 264-295,
// 430-474 and their autograd.
//
// Layout: activations [B][C][L] fp32 contiguous (coalesced along L).  C <= 64, L <= 512.
// Mapping: a workgroup owns ONE channel and a contiguous slice of the (b, l) index space, so
// BatchNorm partial sums are per-workgroup scalars (fixed order, no atomics); parameter
// gradients are produced one workgroup per element as final values.
#ifdef RAAE_STAMPS
__device__ long long d_stamps[4][3][16];
__device__ unsigned long long d_stage_sum[4][3][16], d_stage_cnt[4][3][16];   // per kernel: ticks spent before stamp i, visits
__device__ long long d_stage_prev[4];
__device__ unsigned long long d_kind_sum[8][4][16], d_kind_cnt[8][4][16];     // the same per block shape (7 = generic), grids of > 128 workgroups
#endif
#include "raae_common.h"
#include <string.h>
#include <stdlib.h>
#include <stddef.h>

namespace {

using raae::prelu;

#define CV_MAXC 64
#define RAAE_BIG_ROWS 1024     // batches from here up run the BIG kernel instances (16-byte staging / elementwise paths)

struct ViewStats { float mean[CV_MAXC]; float rstd[CV_MAXC]; };
struct GradStats { float mean[CV_MAXC]; float rstd[CV_MAXC]; float m1[CV_MAXC]; float m2[CV_MAXC]; };

// All conv kernels run 256-thread blocks with <= CV_MAXC channels: the merged statistic pass applies.
__device__ __forceinline__ raae::StatJob view_job(const raae_view_t& v, int C, ViewStats* st, bool update) {
    return v.has_bn ? raae::stat_job_bn(v.bn, C, st->mean, st->rstd, update) : raae::stat_job_none();
}
__device__ __forceinline__ void view_prologue(const raae_view_t& v, int C, ViewStats* st, bool block0) {
    const raae::StatJob jobs[1] = {view_job(v, C, st, true)};
    raae::stat_jobs<1, CV_MAXC>(jobs, block0);
}
__device__ __forceinline__ void grad_prologue(const raae_grad_t& g, int C, GradStats* st) {
    const raae::StatJob jobs[2] = {
        g.has_bn ? raae::stat_job_bn(g.bn, C, st->mean, st->rstd, false) : raae::stat_job_none(),
        g.has_bn ? raae::stat_job_bwd(g.g_partials, g.g_nparts, C, g.bn.count, st->m1, st->m2) : raae::stat_job_none()};
    raae::stat_jobs<2, CV_MAXC>(jobs, false);
}
// statistics of a view and of a gradient in one pass
__device__ __forceinline__ void view_grad_prologue(const raae_view_t& v, int Cv, ViewStats* vs, const raae_grad_t& g,
                                                   int Cg, GradStats* gs) {
    const raae::StatJob jobs[3] = {
        view_job(v, Cv, vs, false),
        g.has_bn ? raae::stat_job_bn(g.bn, Cg, gs->mean, gs->rstd, false) : raae::stat_job_none(),
        g.has_bn ? raae::stat_job_bwd(g.g_partials, g.g_nparts, Cg, g.bn.count, gs->m1, gs->m2) : raae::stat_job_none()};
    raae::stat_jobs<3, CV_MAXC>(jobs, false);
}

// value of a view at flat element idx of channel c
__device__ __forceinline__ float view_at(const raae_view_t& v, const ViewStats* st, int c, size_t idx) {
    float x = v.raw[idx];
    if (v.slope) x = prelu(x, v.slope[c]);
    if (v.has_bn) x = (x - st->mean[c]) * st->rstd[c];
    if (v.mask) x *= v.mask[idx];
    return x;
}
// BatchNorm output of a view (before the dropout scale) -- "y" of the BN backward formula
__device__ __forceinline__ float view_y(const raae_view_t& v, const ViewStats* st, int c, size_t idx) {
    float x = v.raw[idx];
    if (v.slope) x = prelu(x, v.slope[c]);
    return (x - st->mean[c]) * st->rstd[c];
}
// dL/d raw at element idx of channel c; *da = gradient w.r.t. the PReLU output (for dslope)
__device__ __forceinline__ float grad_at(const raae_grad_t& g, const GradStats* st, int c, size_t idx, float* da_out,
                                         float* raw_out) {
    const float gv = g.g[idx];
    const float raw = g.raw ? g.raw[idx] : 0.f;
    float da = gv;
    if (g.has_bn) {
        const float u = g.u ? g.u[idx] : (g.slope ? prelu(raw, g.slope[c]) : raw);
        const float y = (u - st->mean[c]) * st->rstd[c];
        da = st->rstd[c] * (gv - st->m1[c] - y * st->m2[c]);
    }
    float dr = da;
    if (g.slope) dr = raw > 0.f ? da : da * g.slope[c];
    else if (g.act == RAAE_OUT_SOFTPLUS) dr = da * (1.f - expf(-2.f * raw));
    else if (g.act == RAAE_OUT_RELU) dr = raw > 0.f ? da : 0.f;
    if (da_out) *da_out = da;
    if (raw_out) *raw_out = raw;
    return dr;
}

// slice of the (b, l) space owned by this workgroup: [lo, hi)
__device__ __forceinline__ void slice_range(long total, int nsl, int sl, long* lo, long* hi) {
    const long per = (total + nsl - 1) / nsl;
    *lo = (long)sl * per;
    *hi = *lo + per < total ? *lo + per : total;
    if (*lo > total) *lo = total;
}

// ------------------------------------------------------------------ conv forward
struct ConvFwdArgs {
    raae_view_t in; int B; raae_conv_t cv; const float* w; const float* bias; float* out;
    int stats_kind; const float* out_slope; double* out_partials; int act; int nsl;
};

__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvFwdArgs a) {
    __shared__ ViewStats vs;
    __shared__ float s_w[CV_MAXC * 16];
    __shared__ double shd[16];
    const raae_conv_t& cv = a.cv;
    const int co = blockIdx.x % cv.Cout, sl = blockIdx.x / cv.Cout;
    const int cig = cv.Cin / cv.groups, cog = cv.Cout / cv.groups;
    const int grp = co / cog, col = co - grp * cog;
    view_prologue(a.in, cv.Cin, &vs, blockIdx.x == 0);
    // weights of this output channel -> LDS as [ci_local][t]
    for (int i = threadIdx.x; i < cig * cv.K; i += 256) {
        const int cil = i / cv.K, t = i - cil * cv.K;
        s_w[i] = cv.transposed ? a.w[((size_t)(grp * cig + cil) * cog + col) * cv.K + t]
                               : a.w[((size_t)co * cig + cil) * cv.K + t];
    }
    __syncthreads();
    const float bias = a.bias[co];
    const float oslope = (a.stats_kind == RAAE_OUT_STATS_PRELU) ? a.out_slope[co] : 1.f;
    long lo, hi;
    slice_range((long)a.B * cv.Lout, a.nsl, sl, &lo, &hi);
    double s_acc = 0.0, q_acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / cv.Lout), l = (int)(i - (long)b * cv.Lout);
        float acc = bias;
        for (int cil = 0; cil < cig; ++cil) {
            const int ci = grp * cig + cil;
            const size_t base = ((size_t)b * cv.Cin + ci) * cv.Lin;
            if (cv.transposed) {
                const int li = l / cv.stride, t = l - li * cv.stride;
                acc += s_w[cil * cv.K + t] * view_at(a.in, &vs, ci, base + li);
            } else {
                for (int t = 0; t < cv.K; ++t) {
                    int p = l * cv.stride + t - cv.pad;
                    if (p < 0 || p >= cv.Lin) {
                        if (!cv.pad_replicate) continue;
                        p = p < 0 ? 0 : cv.Lin - 1;
                    }
                    acc += s_w[cil * cv.K + t] * view_at(a.in, &vs, ci, base + p);
                }
            }
        }
        float o = acc;
        if (a.act == RAAE_OUT_SOFTPLUS) o = raae::softplus2(acc);
        else if (a.act == RAAE_OUT_RELU) o = fmaxf(acc, 0.f);
        a.out[((size_t)b * cv.Cout + co) * cv.Lout + l] = o;
        if (a.stats_kind != RAAE_OUT_RAW) {
            const float v = (a.stats_kind == RAAE_OUT_STATS_PRELU) ? prelu(acc, oslope) : acc;
            s_acc += (double)v; q_acc += (double)v * (double)v;
        }
    }
    if (a.stats_kind != RAAE_OUT_RAW) {
        const double s = raae::block_sum(s_acc, shd);
        const double q = raae::block_sum(q_acc, shd);
        if (threadIdx.x == 0) {
            double* p = a.out_partials + ((size_t)sl * cv.Cout + co) * 2;
            p[0] = s; p[1] = q;
        }
    }
}

// ------------------------------------------------------------------ conv backward (data)
struct ConvBwdDataArgs {
    raae_grad_t go; int B; raae_conv_t cv; const float* w; raae_view_t in; float* din; int accumulate;
    double* din_partials; int nsl;
};

__global__ __launch_bounds__(256) void conv_bwd_data_kernel(ConvBwdDataArgs a) {
    __shared__ ViewStats vs;
    __shared__ GradStats gs;
    __shared__ float s_w[CV_MAXC * 16];          // [co_local][t] for this input channel
    __shared__ double shd[16];
    const raae_conv_t& cv = a.cv;
    const int ci = blockIdx.x % cv.Cin, sl = blockIdx.x / cv.Cin;
    const int cig = cv.Cin / cv.groups, cog = cv.Cout / cv.groups;
    const int grp = ci / cig, cil = ci - grp * cig;
    view_grad_prologue(a.in, cv.Cin, &vs, a.go, cv.Cout, &gs);
    for (int i = threadIdx.x; i < cog * cv.K; i += 256) {
        const int col = i / cv.K, t = i - col * cv.K;
        s_w[i] = cv.transposed ? a.w[((size_t)ci * cog + col) * cv.K + t]
                               : a.w[((size_t)(grp * cog + col) * cig + cil) * cv.K + t];
    }
    __syncthreads();
    long lo, hi;
    slice_range((long)a.B * cv.Lin, a.nsl, sl, &lo, &hi);
    double s_acc = 0.0, q_acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / cv.Lin), li = (int)(i - (long)b * cv.Lin);
        float acc = 0.f;
        for (int col = 0; col < cog; ++col) {
            const int co = grp * cog + col;
            const size_t obase = ((size_t)b * cv.Cout + co) * cv.Lout;
            if (cv.transposed) {
                for (int t = 0; t < cv.K; ++t)
                    acc += s_w[col * cv.K + t] * grad_at(a.go, &gs, co, obase + (size_t)li * cv.stride + t, nullptr, nullptr);
            } else {
                // padded positions p that read input element li
                int p0 = li + cv.pad, p1 = li + cv.pad;
                if (cv.pad_replicate) {
                    if (li == 0) p0 = 0;
                    if (li == cv.Lin - 1) p1 = cv.Lin - 1 + 2 * cv.pad;
                }
                for (int p = p0; p <= p1; ++p)
                    for (int t = 0; t < cv.K; ++t) {
                        const int num = p - t;
                        if (num < 0) break;
                        if (num % cv.stride) continue;
                        const int l = num / cv.stride;
                        if (l >= cv.Lout) continue;
                        acc += s_w[col * cv.K + t] * grad_at(a.go, &gs, co, obase + l, nullptr, nullptr);
                    }
            }
        }
        const size_t idx = ((size_t)b * cv.Cin + ci) * cv.Lin + li;
        if (a.in.mask) acc *= a.in.mask[idx];
        if (a.accumulate) acc += a.din[idx];
        a.din[idx] = acc;
        if (a.din_partials) {
            const float y = view_y(a.in, &vs, ci, idx);
            s_acc += (double)acc; q_acc += (double)acc * (double)y;
        }
    }
    if (a.din_partials) {
        const double s = raae::block_sum(s_acc, shd);
        const double q = raae::block_sum(q_acc, shd);
        if (threadIdx.x == 0) {
            double* p = a.din_partials + ((size_t)sl * cv.Cin + ci) * 2;
            p[0] = s; p[1] = q;
        }
    }
}

// ------------------------------------------------------------------ conv backward (parameters)
struct ConvBwdWArgs {
    raae_grad_t go; int B; raae_conv_t cv; raae_view_t in; float* dw; float* dbias; float* dslope; int nw;
};

// workgroup e < nw: weight element e (torch layout); nw <= e < nw+Cout: dbias; then dslope.
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(ConvBwdWArgs a) {
    __shared__ ViewStats vs;
    __shared__ GradStats gs;
    __shared__ double shd[16];
    const raae_conv_t& cv = a.cv;
    const int cig = cv.Cin / cv.groups, cog = cv.Cout / cv.groups;
    view_grad_prologue(a.in, cv.Cin, &vs, a.go, cv.Cout, &gs);
    const int e = blockIdx.x;
    double acc = 0.0;
    if (e < a.nw) {
        int co, ci, t;
        if (cv.transposed) {        // [Cin][cog][K]
            t = e % cv.K; const int col = (e / cv.K) % cog; ci = e / (cv.K * cog);
            co = (ci / cig) * cog + col;
        } else {                    // [Cout][cig][K]
            t = e % cv.K; const int cil = (e / cv.K) % cig; co = e / (cv.K * cig);
            ci = (co / cog) * cig + cil;
        }
        const long total = (long)a.B * (cv.transposed ? cv.Lin : cv.Lout);
        for (long i = threadIdx.x; i < total; i += 256) {
            if (cv.transposed) {
                const int b = (int)(i / cv.Lin), li = (int)(i - (long)b * cv.Lin);
                const float x = view_at(a.in, &vs, ci, ((size_t)b * cv.Cin + ci) * cv.Lin + li);
                const float g = grad_at(a.go, &gs, co, ((size_t)b * cv.Cout + co) * cv.Lout + (size_t)li * cv.stride + t,
                                        nullptr, nullptr);
                acc += (double)x * (double)g;
            } else {
                const int b = (int)(i / cv.Lout), l = (int)(i - (long)b * cv.Lout);
                int p = l * cv.stride + t - cv.pad;
                if (p < 0 || p >= cv.Lin) {
                    if (!cv.pad_replicate) continue;
                    p = p < 0 ? 0 : cv.Lin - 1;
                }
                const float x = view_at(a.in, &vs, ci, ((size_t)b * cv.Cin + ci) * cv.Lin + p);
                const float g = grad_at(a.go, &gs, co, ((size_t)b * cv.Cout + co) * cv.Lout + l, nullptr, nullptr);
                acc += (double)x * (double)g;
            }
        }
        const double t2 = raae::block_sum(acc, shd);
        if (threadIdx.x == 0) a.dw[e] = (float)t2;
    } else {
        const bool is_bias = e < a.nw + cv.Cout;
        const int co = is_bias ? e - a.nw : e - a.nw - cv.Cout;
        if (!is_bias && a.dslope == nullptr) return;
        const long total = (long)a.B * cv.Lout;
        for (long i = threadIdx.x; i < total; i += 256) {
            const int b = (int)(i / cv.Lout), l = (int)(i - (long)b * cv.Lout);
            float da, raw;
            const float g = grad_at(a.go, &gs, co, ((size_t)b * cv.Cout + co) * cv.Lout + l, &da, &raw);
            if (is_bias) acc += (double)g;
            else if (raw <= 0.f) acc += (double)da * (double)raw;
        }
        const double t2 = raae::block_sum(acc, shd);
        if (threadIdx.x == 0) { if (is_bias) a.dbias[co] = (float)t2; else a.dslope[co] = (float)t2; }
    }
}

// ------------------------------------------------------------------ length-axis Linear
struct LenLinFwdArgs {
    raae_view_t in; int B; int C; int Lin; const float* w; const float* bias; int E; float* out;
    int stats_kind; const float* out_slope; double* out_partials; int nsl;
};

__global__ __launch_bounds__(256) void lenlin_fwd_kernel(LenLinFwdArgs a) {
    __shared__ ViewStats vs;
    __shared__ double shd[16];
    const int c = blockIdx.x % a.C, sl = blockIdx.x / a.C;
    view_prologue(a.in, a.C, &vs, blockIdx.x == 0);
    const float oslope = (a.stats_kind == RAAE_OUT_STATS_PRELU) ? a.out_slope[c] : 1.f;
    long lo, hi;
    slice_range((long)a.B * a.E, a.nsl, sl, &lo, &hi);
    double s_acc = 0.0, q_acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / a.E), e = (int)(i - (long)b * a.E);
        const size_t base = ((size_t)b * a.C + c) * a.Lin;
        float acc = a.bias[e];
        for (int l = 0; l < a.Lin; ++l) acc += a.w[(size_t)e * a.Lin + l] * view_at(a.in, &vs, c, base + l);
        a.out[((size_t)b * a.C + c) * a.E + e] = acc;
        if (a.stats_kind != RAAE_OUT_RAW) {
            const float v = (a.stats_kind == RAAE_OUT_STATS_PRELU) ? prelu(acc, oslope) : acc;
            s_acc += (double)v; q_acc += (double)v * (double)v;
        }
    }
    if (a.stats_kind != RAAE_OUT_RAW) {
        const double s = raae::block_sum(s_acc, shd);
        const double q = raae::block_sum(q_acc, shd);
        if (threadIdx.x == 0) {
            double* p = a.out_partials + ((size_t)sl * a.C + c) * 2;
            p[0] = s; p[1] = q;
        }
    }
}

struct LenLinBwdDataArgs {
    raae_grad_t go; int B; int C; int E; const float* w; raae_view_t in; int Lin; float* din; int accumulate;
    double* din_partials; int nsl;
};

__global__ __launch_bounds__(256) void lenlin_bwd_data_kernel(LenLinBwdDataArgs a) {
    __shared__ ViewStats vs;
    __shared__ GradStats gs;
    __shared__ double shd[16];
    const int c = blockIdx.x % a.C, sl = blockIdx.x / a.C;
    view_grad_prologue(a.in, a.C, &vs, a.go, a.C, &gs);
    long lo, hi;
    slice_range((long)a.B * a.Lin, a.nsl, sl, &lo, &hi);
    double s_acc = 0.0, q_acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / a.Lin), l = (int)(i - (long)b * a.Lin);
        const size_t obase = ((size_t)b * a.C + c) * a.E;
        float acc = 0.f;
        for (int e = 0; e < a.E; ++e)
            acc += a.w[(size_t)e * a.Lin + l] * grad_at(a.go, &gs, c, obase + e, nullptr, nullptr);
        const size_t idx = ((size_t)b * a.C + c) * a.Lin + l;
        if (a.in.mask) acc *= a.in.mask[idx];
        if (a.accumulate) acc += a.din[idx];
        a.din[idx] = acc;
        if (a.din_partials) {
            const float y = view_y(a.in, &vs, c, idx);
            s_acc += (double)acc; q_acc += (double)acc * (double)y;
        }
    }
    if (a.din_partials) {
        const double s = raae::block_sum(s_acc, shd);
        const double q = raae::block_sum(q_acc, shd);
        if (threadIdx.x == 0) {
            double* p = a.din_partials + ((size_t)sl * a.C + c) * 2;
            p[0] = s; p[1] = q;
        }
    }
}

struct LenLinBwdWArgs {
    raae_grad_t go; int B; int C; int E; raae_view_t in; int Lin; float* dw; float* dbias; float* dslope;
};

// workgroup x < E*Lin: dW[e][l]; then E workgroups dbias[e]; then C workgroups dslope[c]
__global__ __launch_bounds__(256) void lenlin_bwd_weight_kernel(LenLinBwdWArgs a) {
    __shared__ ViewStats vs;
    __shared__ GradStats gs;
    __shared__ double shd[16];
    view_grad_prologue(a.in, a.C, &vs, a.go, a.C, &gs);
    const int x = blockIdx.x, nw = a.E * a.Lin;
    double acc = 0.0;
    if (x < nw) {
        const int e = x / a.Lin, l = x - e * a.Lin;
        const long total = (long)a.B * a.C;
        for (long i = threadIdx.x; i < total; i += 256) {
            const int c = (int)(i % a.C);
            const float g = grad_at(a.go, &gs, c, (size_t)i * a.E + e, nullptr, nullptr);
            acc += (double)g * (double)view_at(a.in, &vs, c, (size_t)i * a.Lin + l);
        }
        const double t = raae::block_sum(acc, shd);
        if (threadIdx.x == 0) a.dw[x] = (float)t;
    } else if (x < nw + a.E) {
        const int e = x - nw;
        const long total = (long)a.B * a.C;
        for (long i = threadIdx.x; i < total; i += 256)
            acc += (double)grad_at(a.go, &gs, (int)(i % a.C), (size_t)i * a.E + e, nullptr, nullptr);
        const double t = raae::block_sum(acc, shd);
        if (threadIdx.x == 0) a.dbias[e] = (float)t;
    } else {
        if (a.dslope == nullptr) return;
        const int c = x - nw - a.E;
        const long total = (long)a.B * a.E;
        for (long i = threadIdx.x; i < total; i += 256) {
            const int b = (int)(i / a.E), e = (int)(i - (long)b * a.E);
            float da, raw;
            grad_at(a.go, &gs, c, ((size_t)b * a.C + c) * a.E + e, &da, &raw);
            if (raw <= 0.f) acc += (double)da * (double)raw;
        }
        const double t = raae::block_sum(acc, shd);
        if (threadIdx.x == 0) a.dslope[c] = (float)t;
    }
}

// ------------------------------------------------------------------ block sum / grad materialise
struct Sum3Args { raae_view_t a, b, c; int B; int C; int L; float* y; double* out_partials; int nsl; };

__global__ __launch_bounds__(256) void sum3_kernel(Sum3Args a) {
    __shared__ ViewStats va, vb, vc;
    __shared__ double shd[16];
    const int c = blockIdx.x % a.C, sl = blockIdx.x / a.C;
    view_prologue(a.a, a.C, &va, false);
    view_prologue(a.b, a.C, &vb, blockIdx.x == 0);
    view_prologue(a.c, a.C, &vc, false);
    long lo, hi;
    slice_range((long)a.B * a.L, a.nsl, sl, &lo, &hi);
    double s_acc = 0.0, q_acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / a.L), l = (int)(i - (long)b * a.L);
        const size_t idx = ((size_t)b * a.C + c) * a.L + l;
        const float v = view_at(a.a, &va, c, idx) + view_at(a.b, &vb, c, idx) + view_at(a.c, &vc, c, idx);
        a.y[idx] = v;
        s_acc += (double)v; q_acc += (double)v * (double)v;
    }
    if (a.out_partials) {
        const double s = raae::block_sum(s_acc, shd);
        const double q = raae::block_sum(q_acc, shd);
        if (threadIdx.x == 0) {
            double* p = a.out_partials + ((size_t)sl * a.C + c) * 2;
            p[0] = s; p[1] = q;
        }
    }
}

struct GradMatArgs { raae_grad_t go; int B; int C; int L; float* draw; float* dslope; int accumulate; };

// one workgroup per channel: writes dRaw (if draw) and the final dslope[c] (if dslope)
__global__ __launch_bounds__(256) void grad_materialize_kernel(GradMatArgs a) {
    __shared__ GradStats gs;
    __shared__ double shd[16];
    const int c = blockIdx.x;
    grad_prologue(a.go, a.C, &gs);
    double acc = 0.0;
    const long total = (long)a.B * a.L;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int b = (int)(i / a.L), l = (int)(i - (long)b * a.L);
        const size_t idx = ((size_t)b * a.C + c) * a.L + l;
        float da, raw;
        const float dr = grad_at(a.go, &gs, c, idx, &da, &raw);
        if (a.draw) a.draw[idx] = a.accumulate ? a.draw[idx] + dr : dr;
        if (raw <= 0.f) acc += (double)da * (double)raw;
    }
    if (a.dslope) {
        const double t = raae::block_sum(acc, shd);
        if (threadIdx.x == 0) a.dslope[c] = (float)t;
    }
}

#include "raae_block_shapes.inc"     // constexpr table of the residual-block shapes the fused kernels are instantiated for
__host__ __device__ constexpr int clg2(int v) { int s = -1; if (v > 0 && !(v & (v - 1))) { s = 0; while ((1 << s) < v) ++s; } return s; }
#include "raae_conv_tiled.inc"
#include "raae_block_fused.inc"
#include "raae_conv_strip.inc"
#include "raae_head.inc"

int slices_for(long per_channel, int C) {
    long n = (per_channel + 255) / 256;
    long cap = RAAE_MAX_PARTS;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    (void)C;
    return (int)n;
}

bool view_ok(const raae_view_t* v, int C) {
    return v && v->raw && C <= CV_MAXC && (!v->has_bn || (v->bn.nparts <= RAAE_MAX_PARTS &&
                                                         (v->bn.partials || (v->bn.running_mean && v->bn.running_var))));
}
bool grad_ok(const raae_grad_t* g, int C) {
    return g && g->g && C <= CV_MAXC && (g->raw || (!g->slope && g->act == RAAE_OUT_RAW && (!g->has_bn || g->u)))
           && (!g->has_bn || (g->g_partials && g->g_nparts > 0 && g->g_nparts <= RAAE_MAX_PARTS && g->bn.partials));
}
bool conv_ok(const raae_conv_t* cv) {
    if (!cv || cv->Cin < 1 || cv->Cout < 1 || cv->K < 1 || cv->K > 16 || cv->stride < 1 || cv->groups < 1) return false;
    if (cv->Cin % cv->groups || cv->Cout % cv->groups || cv->Cin > CV_MAXC || cv->Cout > CV_MAXC) return false;
    if (cv->transposed) return cv->K == cv->stride && cv->pad == 0 && cv->Lout == cv->Lin * cv->stride;
    return cv->Lout == (cv->Lin + 2 * cv->pad - cv->K) / cv->stride + 1;
}
int conv_nw(const raae_conv_t* cv) {
    return cv->transposed ? cv->Cin * (cv->Cout / cv->groups) * cv->K : cv->Cout * (cv->Cin / cv->groups) * cv->K;
}
thread_local int g_tile_mult = 1;
// samples per tile: enough outputs to occupy 256 threads, bounded by an LDS budget
int pick_S(long floats_per_sample, long outputs_per_sample, int B, long lds_budget_floats, int min_out) {
    long S = (min_out + outputs_per_sample - 1) / outputs_per_sample;
    if (S < 1) S = 1;
    const long cap = lds_budget_floats / (floats_per_sample > 0 ? floats_per_sample : 1);
    // large batches: about two groups per workgroup of a 512-workgroup grid, up to 4x the samples per group
    // (fewer barriers, more loads in flight per staging pass; measured at B = 4096: 180 -> 201 steps/s, 8x: 192)
    const long Bh = (long)B * g_tile_mult;      // raae_tile_hint: size the groups for a caller that batches trials
    long want = (Bh + 1023) / 1024;
    // small samples (a few KB per sample): ONE group per workgroup -- such kernels are a chain
    // of barrier-separated stages of a few microseconds each, and a second group repeats the chain
    // (measured at B = 4096: 217 -> 223 steps/s with the threshold anywhere between 1400 and 4400 floats per sample)
    static const long small_floats = getenv("RAAE_PICK_SMALL") ? atol(getenv("RAAE_PICK_SMALL")) : 2200;
    static const long small_mult = getenv("RAAE_PICK_MULT") ? atol(getenv("RAAE_PICK_MULT")) : 8;
    // ... of a 256-workgroup grid (RAAE_PICK_DIV): half as many partial-statistic rows for every consumer's prologue and
    // half as many prologues; measured 1024 / 2048 / 4096 / 8192 rows: 521 -> 547, 390 -> 405, 272.5 -> 279.6,
    // 170.5 -> 168.9 steps/s against 512 workgroups
    static const long small_div = getenv("RAAE_PICK_DIV") ? atol(getenv("RAAE_PICK_DIV")) : 256;
    if (floats_per_sample <= small_floats) want = (Bh + small_div - 1) / small_div;
    if (want > small_mult * S && floats_per_sample <= small_floats) want = small_mult * S;
    else
    if (want > 4 * S) want = 4 * S;
    if (S < want) S = want;
    if (S > cap) S = cap;
    if (S > B) S = B;
    if (S < 1) S = 1;
    return (int)S;
}
int lg2(int v) { if (v <= 0 || (v & (v - 1))) return -1; int s = 0; while ((1 << s) < v) ++s; return s; }
const long kTileBudget = 10 * 1024;    // floats (40 KB) of dynamic LDS for staged tiles

}  // namespace

extern "C" int raae_tile_hint(int rows_multiplier) {
    RAAE_CHECK_ARG(rows_multiplier >= 1 && rows_multiplier <= 64);
    g_tile_mult = rows_multiplier;
    return 0;
}

extern "C" int raae_conv_fwd(const raae_view_t* in, int B, const raae_conv_t* cv, const float* w, const float* bias,
                             float* out, int stats_kind, const float* out_slope, double* out_partials, int* out_nparts,
                             int act, void* stream) {
    RAAE_CHECK_ARG(conv_ok(cv) && view_ok(in, cv->Cin) && w && bias && out && B > 0);
    RAAE_CHECK_ARG(stats_kind == RAAE_OUT_RAW || out_partials);
    RAAE_CHECK_ARG(stats_kind != RAAE_OUT_STATS_PRELU || out_slope);
    ConvFwdArgs a;
    a.in = *in; a.B = B; a.cv = *cv; a.w = w; a.bias = bias; a.out = out; a.stats_kind = stats_kind;
    a.out_slope = out_slope; a.out_partials = out_partials; a.act = act;
    if (stats_kind == RAAE_OUT_RAW && head_shape_ok(cv, in) && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        // the decoder's head: BatchNorm -> Conv1d(C, 1, 1) -> activation as one streaming pass (raae_head.inc)
        HeadFwdArgs h;
        h.in = *in; h.B = B; h.L = cv->Lin; h.w = w; h.bias = bias; h.out = out; h.act = act; h.nq = B * (cv->Lin >> 2);
        const int grid = head_grid(h.nq, cv->Cin <= 4 ? kHeadU : kHeadU / 2);
        if (out_nparts) *out_nparts = 0;
        if (cv->Cin == 4) raae::launch(head_fwd_kernel<4>, head_fwd_kernel_m<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h);
        else raae::launch(head_fwd_kernel<8>, head_fwd_kernel_m<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h);
        RAAE_LAUNCH_RET();
    }
    if (conv_fwd_strip(a, out_nparts, (hipStream_t)stream)) RAAE_LAUNCH_RET();
    const long per_in = (long)cv->Cin * (cv->Lin + 2 * (cv->transposed ? 0 : cv->pad));
    if (conv_nw(cv) <= 1024 && (stats_kind == RAAE_OUT_RAW || cv->Cout <= CT_MAXCH) && per_in <= kTileBudget) {
        ConvFwdTArgs t;
        t.a = a; t.sh_in = lg2(cv->Lin); t.sh_out = lg2(cv->Lout);
        t.S = pick_S(per_in, (long)cv->Cout * cv->Lout, B, kTileBudget, 256);
        t.ngroups = (B + t.S - 1) / t.S;
        const int grid = t.ngroups < RAAE_MAX_PARTS ? t.ngroups : RAAE_MAX_PARTS;
        t.a.nsl = grid;
        if (out_nparts) *out_nparts = grid;
        if (B >= RAAE_BIG_ROWS) RAAE_PLAIN_LAUNCH(conv_fwd_tiled_kernel<true>, dim3(grid), dim3(256), sizeof(float) * t.S * per_in,
                           (hipStream_t)stream, t);
        else RAAE_PLAIN_LAUNCH(conv_fwd_tiled_kernel<false>, dim3(grid), dim3(256), sizeof(float) * t.S * per_in,
                           (hipStream_t)stream, t);
        RAAE_LAUNCH_RET();
    }
    a.nsl = slices_for((long)B * cv->Lout, cv->Cout);
    if (out_nparts) *out_nparts = a.nsl;
    RAAE_PLAIN_LAUNCH(conv_fwd_kernel, dim3(a.nsl * cv->Cout), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_head_bwd_supported(const raae_grad_t* go, int B, const raae_conv_t* cv, const raae_view_t* in) {
    return go && cv && in && B > 0 && conv_ok(cv) && head_shape_ok(cv, in) && !go->has_bn && !go->slope && go->g &&
           (go->act == RAAE_OUT_RAW || go->raw) && (reinterpret_cast<uintptr_t>(go->g) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(go->raw) & 15) == 0;
}

extern "C" int raae_head_bwd(const raae_grad_t* go, int B, const raae_conv_t* cv, const float* w, const raae_view_t* in,
                             float* din, double* din_partials, int* din_nparts, float* dw, float* dbias,
                             long slab_stride, int* nslab, void* stream) {
    RAAE_CHECK_ARG(raae_head_bwd_supported(go, B, cv, in) && w && din && din_partials && dw && dbias && slab_stride > 0 &&
                   (reinterpret_cast<uintptr_t>(din) & 15) == 0);
    HeadBwdArgs h;
    h.go = *go; h.in = *in; h.B = B; h.L = cv->Lin; h.w = w; h.din = din; h.din_partials = din_partials; h.dw = dw;
    h.dbias = dbias; h.slab_stride = slab_stride; h.nq = B * (cv->Lin >> 2);
    const int grid = head_grid(h.nq, 2);
    if (din_nparts) *din_nparts = grid;
    if (nslab) *nslab = grid;
    if (cv->Cin == 4) raae::launch(head_bwd_kernel<4>, head_bwd_kernel_m<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h);
    else raae::launch(head_bwd_kernel<8>, head_bwd_kernel_m<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_conv_bwd_data(const raae_grad_t* go, int B, const raae_conv_t* cv, const float* w,
                                  const raae_view_t* in, float* din, int accumulate, double* din_partials,
                                  int* din_nparts, void* stream) {
    RAAE_CHECK_ARG(conv_ok(cv) && grad_ok(go, cv->Cout) && view_ok(in, cv->Cin) && w && din && B > 0);
    RAAE_CHECK_ARG(!din_partials || in->has_bn);
    ConvBwdDataArgs a;
    a.go = *go; a.B = B; a.cv = *cv; a.w = w; a.in = *in; a.din = din; a.accumulate = accumulate;
    a.din_partials = din_partials;
    const long per_g = (long)cv->Cout * cv->Lout;
    if (conv_nw(cv) <= 1024 && (!din_partials || cv->Cin <= CT_MAXCH) && per_g <= kTileBudget) {
        ConvBwdDataTArgs t;
        t.a = a; t.sh_in = lg2(cv->Lin); t.sh_out = lg2(cv->Lout);
        t.S = pick_S(per_g, (long)cv->Cin * cv->Lin, B, kTileBudget, 256);
        t.ngroups = (B + t.S - 1) / t.S;
        const int grid = t.ngroups < RAAE_MAX_PARTS ? t.ngroups : RAAE_MAX_PARTS;
        t.a.nsl = grid;
        if (din_nparts) *din_nparts = grid;
        if (B >= RAAE_BIG_ROWS) RAAE_PLAIN_LAUNCH(conv_bwd_data_tiled_kernel<true>, dim3(grid), dim3(256), sizeof(float) * t.S * per_g,
                           (hipStream_t)stream, t);
        else RAAE_PLAIN_LAUNCH(conv_bwd_data_tiled_kernel<false>, dim3(grid), dim3(256), sizeof(float) * t.S * per_g,
                           (hipStream_t)stream, t);
        RAAE_LAUNCH_RET();
    }
    a.nsl = slices_for((long)B * cv->Lin, cv->Cin);
    if (din_nparts) *din_nparts = a.nsl;
    RAAE_PLAIN_LAUNCH(conv_bwd_data_kernel, dim3(a.nsl * cv->Cin), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_conv_bwd_weight(const raae_grad_t* go, int B, const raae_conv_t* cv, const raae_view_t* in,
                                    float* dw, float* dbias, float* dslope, long slab_stride, int* nslab,
                                    void* stream) {
    RAAE_CHECK_ARG(conv_ok(cv) && grad_ok(go, cv->Cout) && view_ok(in, cv->Cin) && dw && dbias && B > 0);
    RAAE_CHECK_ARG(!dslope || go->slope);
    ConvBwdWArgs a;
    a.go = *go; a.B = B; a.cv = *cv; a.in = *in; a.dw = dw; a.dbias = dbias; a.dslope = dslope;
    a.nw = conv_nw(cv);
    const long per = (long)(dslope ? 2 : 1) * cv->Cout * cv->Lout + (long)cv->Cin * (cv->Lin + 2 * (cv->transposed ? 0 : cv->pad));
    if (a.nw <= 1024 && cv->Cout <= 8 && per <= kTileBudget) {
        ConvBwdWTArgs t;
        t.a = a; t.slab_stride = slab_stride; t.sh_in = lg2(cv->Lin); t.sh_out = lg2(cv->Lout);
        const long span = cv->transposed ? cv->Lin : cv->Lout;      // inner-loop trip count per sample
        t.S = pick_S(per, span, B, kTileBudget, 256);
        t.ngroups = (B + t.S - 1) / t.S;
        // 128 workgroups like the tasks of raae_block_wgrad (rocprofv3, decoder head at B=256: 18.2 us with 64, 13.6 us
        // with 128 or 256); samples per group follow
        { const int scap = (B + 127) / 128; if (t.S > scap) { t.S = scap; t.ngroups = (B + t.S - 1) / t.S; } }
        const int grid = t.ngroups < 128 ? t.ngroups : 128;
        if (nslab) *nslab = grid;
        if (B >= RAAE_BIG_ROWS) RAAE_PLAIN_LAUNCH(conv_bwd_weight_tiled_kernel<true>, dim3(grid), dim3(256), sizeof(float) * t.S * per,
                           (hipStream_t)stream, t);
        else RAAE_PLAIN_LAUNCH(conv_bwd_weight_tiled_kernel<false>, dim3(grid), dim3(256), sizeof(float) * t.S * per,
                           (hipStream_t)stream, t);
        RAAE_LAUNCH_RET();
    }
    if (nslab) *nslab = 1;
    RAAE_PLAIN_LAUNCH(conv_bwd_weight_kernel, dim3(a.nw + 2 * cv->Cout), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_lenlin_fwd(const raae_view_t* in, int B, int C, int Lin, const float* w, const float* bias, int E,
                               float* out, int stats_kind, const float* out_slope, double* out_partials,
                               int* out_nparts, void* stream) {
    RAAE_CHECK_ARG(view_ok(in, C) && w && bias && out && B > 0 && C > 0 && Lin > 0 && E > 0);
    RAAE_CHECK_ARG(stats_kind == RAAE_OUT_RAW || out_partials);
    RAAE_CHECK_ARG(stats_kind != RAAE_OUT_STATS_PRELU || out_slope);
    LenLinFwdArgs a;
    a.in = *in; a.B = B; a.C = C; a.Lin = Lin; a.w = w; a.bias = bias; a.E = E; a.out = out;
    a.stats_kind = stats_kind; a.out_slope = out_slope; a.out_partials = out_partials;
    const long wfl = (long)E * Lin + E, per = (long)C * Lin;
    if ((stats_kind == RAAE_OUT_RAW || C <= CT_MAXCH) && wfl <= 2048 && per <= kTileBudget - wfl) {
        LenLinFwdTArgs t;
        t.a = a; t.sh_in = lg2(Lin); t.sh_e = lg2(E);
        t.S = pick_S(per, (long)C * E, B, kTileBudget - wfl, Lin >= 64 ? 16 : 256);
        t.ngroups = (B + t.S - 1) / t.S;
        const int grid = t.ngroups < RAAE_MAX_PARTS ? t.ngroups : RAAE_MAX_PARTS;
        t.a.nsl = grid;
        if (out_nparts) *out_nparts = grid;
        RAAE_PLAIN_LAUNCH(lenlin_fwd_tiled_kernel<false>, dim3(grid), dim3(256), sizeof(float) * (t.S * per + wfl),
                           (hipStream_t)stream, t);
        RAAE_LAUNCH_RET();
    }
    a.nsl = slices_for((long)B * E, C);
    if (out_nparts) *out_nparts = a.nsl;
    RAAE_PLAIN_LAUNCH(lenlin_fwd_kernel, dim3(a.nsl * C), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_lenlin_bwd_data(const raae_grad_t* go, int B, int C, int E, const float* w, const raae_view_t* in,
                                    int Lin, float* din, int accumulate, double* din_partials, int* din_nparts,
                                    void* stream) {
    RAAE_CHECK_ARG(grad_ok(go, C) && view_ok(in, C) && w && din && B > 0 && E > 0 && Lin > 0);
    RAAE_CHECK_ARG(!din_partials || in->has_bn);
    LenLinBwdDataArgs a;
    a.go = *go; a.B = B; a.C = C; a.E = E; a.w = w; a.in = *in; a.Lin = Lin; a.din = din; a.accumulate = accumulate;
    a.din_partials = din_partials;
    const long wfl = (long)E * Lin, per = (long)C * E;
    if ((!din_partials || C <= CT_MAXCH) && wfl <= 2048 && per <= kTileBudget - wfl) {
        LenLinBwdDataTArgs t;
        t.a = a; t.sh_in = lg2(Lin); t.sh_e = lg2(E);
        t.S = pick_S(per, (long)C * Lin, B, kTileBudget - wfl, E >= 64 ? 16 : 256);
        t.ngroups = (B + t.S - 1) / t.S;
        const int grid = t.ngroups < RAAE_MAX_PARTS ? t.ngroups : RAAE_MAX_PARTS;
        t.a.nsl = grid;
        if (din_nparts) *din_nparts = grid;
        RAAE_PLAIN_LAUNCH(lenlin_bwd_data_tiled_kernel<false>, dim3(grid), dim3(256), sizeof(float) * (t.S * per + wfl),
                           (hipStream_t)stream, t);
        RAAE_LAUNCH_RET();
    }
    a.nsl = slices_for((long)B * Lin, C);
    if (din_nparts) *din_nparts = a.nsl;
    RAAE_PLAIN_LAUNCH(lenlin_bwd_data_kernel, dim3(a.nsl * C), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_lenlin_bwd_weight(const raae_grad_t* go, int B, int C, int E, const raae_view_t* in, int Lin,
                                      float* dw, float* dbias, float* dslope, long slab_stride, int* nslab,
                                      void* stream) {
    RAAE_CHECK_ARG(grad_ok(go, C) && view_ok(in, C) && dw && dbias && B > 0 && E > 0 && Lin > 0);
    RAAE_CHECK_ARG(!dslope || go->slope);
    LenLinBwdWArgs a;
    a.go = *go; a.B = B; a.C = C; a.E = E; a.in = *in; a.Lin = Lin; a.dw = dw; a.dbias = dbias; a.dslope = dslope;
    const long per = (long)(dslope ? 2 : 1) * C * E + (long)C * Lin;
    if ((long)E * Lin <= 1024 && E <= 256 && C <= CT_MAXCH && per <= kTileBudget) {
        LenLinBwdWTArgs t;
        t.a = a; t.slab_stride = slab_stride; t.sh_in = lg2(Lin); t.sh_e = lg2(E);
        t.S = pick_S(per, C, B, kTileBudget, 64);             // >= 64 rows (s, c) per staging
        t.ngroups = (B + t.S - 1) / t.S;
        const int grid = t.ngroups < 64 ? t.ngroups : 64;
        if (nslab) *nslab = grid;
        RAAE_PLAIN_LAUNCH(lenlin_bwd_weight_tiled_kernel<false>, dim3(grid), dim3(256), sizeof(float) * t.S * per,
                           (hipStream_t)stream, t);
        RAAE_LAUNCH_RET();
    }
    if (nslab) *nslab = 1;
    RAAE_PLAIN_LAUNCH(lenlin_bwd_weight_kernel, dim3(E * Lin + E + C), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_sum3_fwd(const raae_view_t* a_, const raae_view_t* b_, const raae_view_t* c_, int B, int C, int L,
                             float* y, double* out_partials, int* out_nparts, void* stream) {
    RAAE_CHECK_ARG(view_ok(a_, C) && view_ok(b_, C) && view_ok(c_, C) && y && B > 0 && L > 0);
    Sum3Args a;
    a.a = *a_; a.b = *b_; a.c = *c_; a.B = B; a.C = C; a.L = L; a.y = y; a.out_partials = out_partials;
    a.nsl = slices_for((long)B * L, C);
    if (out_nparts) *out_nparts = a.nsl;
    RAAE_PLAIN_LAUNCH(sum3_kernel, dim3(a.nsl * C), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_grad_materialize(const raae_grad_t* go, int B, int C, int L, float* draw, int accumulate,
                                     float* dslope, long slab_stride, int* nslab, void* stream) {
    RAAE_CHECK_ARG(grad_ok(go, C) && (draw || dslope) && B > 0 && L > 0 && C > 0);
    RAAE_CHECK_ARG(!dslope || go->slope);
    GradMatTArgs t;
    t.a.go = *go; t.a.B = B; t.a.C = C; t.a.L = L; t.a.draw = draw; t.a.dslope = dslope; t.a.accumulate = accumulate;
    long n = ((long)B * L + 1023) / 1024;
    if (n > 64) n = 64;
    if (n < 1) n = 1;
    t.nsl = (int)n; t.slab_stride = slab_stride;
    if (nslab) *nslab = t.nsl;
    RAAE_PLAIN_LAUNCH(grad_materialize_sliced_kernel, dim3(t.nsl * C), dim3(256), 0, (hipStream_t)stream, t);
    RAAE_LAUNCH_RET();
}

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? (int)strtol(e, nullptr, 0) : dflt; }
// Which block shapes run their large-batch (`BIG`) instance from RAAE_BIG_ROWS rows on: bit k = block shape k of
// raae_block_shapes.inc, bit 7 = the generic instance.  Chosen on the STEP rate at 4096 rows, not on the kernels alone:
// the 16-byte paths pay for the 64- and 256-point rows of shapes 0, 5, 6 (phase A backward of the last decoder block
// 55 us against 89 plain); for the short rows the BIG instances (170-200 registers) were slower alone while the small
// shapes ran 512 workgroups (phase B backward of decoder block 1 75 us against 31), and since those shapes run 256
// workgroups with twice the samples they are faster alone almost everywhere (that kernel 19.5 us against 28.8) -- yet
// with all of them BIG the step is slower (272-275 steps/s against 280; 388 against 400 at 2048 rows): what a
// main-chain kernel gains alone it takes from the weight-gradient branch running beside it.
// Environment overrides for tuning: RAAE_BIG_MASK_{FWD_A,FWD_B,BWD_B,BWD_A,WGRAD}.
enum { kFamFwdA, kFamFwdB, kFamBwdB, kFamBwdA, kFamWgrad };
static const int kBigMask[5] = {env_int("RAAE_BIG_MASK_FWD_A", 0xe1), env_int("RAAE_BIG_MASK_FWD_B", 0xe1),
                                env_int("RAAE_BIG_MASK_BWD_B", 0xe1), env_int("RAAE_BIG_MASK_BWD_A", 0xf1),
                                env_int("RAAE_BIG_MASK_WGRAD", 0xf1)};
static bool use_big(int B, int kind, int family) {
    return B >= RAAE_BIG_ROWS && ((kBigMask[family] >> (kind < 0 ? 7 : kind)) & 1);
}
// ---- shape-specialised instances of the fused block kernels (raae_block_shapes.inc)
// (`big`: the instance for batches of >= 1024 rows, which carries the 16-byte staging / elementwise paths)
// (the launch-bound instances go through raae::launch: they have the batched form KERNEL_m, one trial per grid plane; the
// large-batch instances do not -- a recording that meets one is refused, raae::record_unsupported)
#define RAAE_LAUNCH_KIND_BIG(KERNEL, ...) if (big) switch (kind) { \
        case 0: RAAE_PLAIN_LAUNCH((KERNEL<0, true>), __VA_ARGS__); break; case 1: RAAE_PLAIN_LAUNCH((KERNEL<1, true>), __VA_ARGS__); break; \
        case 2: RAAE_PLAIN_LAUNCH((KERNEL<2, true>), __VA_ARGS__); break; case 3: RAAE_PLAIN_LAUNCH((KERNEL<3, true>), __VA_ARGS__); break; \
        case 4: RAAE_PLAIN_LAUNCH((KERNEL<4, true>), __VA_ARGS__); break; case 5: RAAE_PLAIN_LAUNCH((KERNEL<5, true>), __VA_ARGS__); break; \
        case 6: RAAE_PLAIN_LAUNCH((KERNEL<6, true>), __VA_ARGS__); break; default: RAAE_PLAIN_LAUNCH((KERNEL<-1, true>), __VA_ARGS__); } \
    else RAAE_LAUNCH_KIND(KERNEL, __VA_ARGS__)
#define RAAE_LAUNCH_KIND(KERNEL, ...) switch (kind) { \
        case 0: raae::launch(KERNEL<0, false>, KERNEL##_m<0>, __VA_ARGS__); break; case 1: raae::launch(KERNEL<1, false>, KERNEL##_m<1>, __VA_ARGS__); break; \
        case 2: raae::launch(KERNEL<2, false>, KERNEL##_m<2>, __VA_ARGS__); break; case 3: raae::launch(KERNEL<3, false>, KERNEL##_m<3>, __VA_ARGS__); break; \
        case 4: raae::launch(KERNEL<4, false>, KERNEL##_m<4>, __VA_ARGS__); break; case 5: raae::launch(KERNEL<5, false>, KERNEL##_m<5>, __VA_ARGS__); break; \
        case 6: raae::launch(KERNEL<6, false>, KERNEL##_m<6>, __VA_ARGS__); break; default: raae::launch(KERNEL<-1, false>, KERNEL##_m<-1>, __VA_ARGS__); }
static_assert(kNumBlkShapes == 7, "RAAE_LAUNCH_KIND enumerates the shape table");
static bool same_conv(const raae_conv_t& x, const raae_conv_t& y) { return !memcmp(&x, &y, sizeof(raae_conv_t)); }
// phase A kernels see (Cin, Cout, Lin, L1, Lout, E, cv1, cvs); phase B kernels (Cin, Cout, L1, Lout, cv2, cve)
static int blk_kind_a(int Cin, int Cout, int Lin, int L1, int Lout, int E, const raae_conv_t& cv1, int has_short,
                      const raae_conv_t& cvs, int has_excit, bool check_excit) {
    for (int k = 0; k < kNumBlkShapes; ++k) {
        const BlkShape& b = kBlk[k];
        if (b.Cin == Cin && b.Cout == Cout && b.Lin == Lin && b.L1 == L1 && b.Lout == Lout && b.E == E &&
            b.has_short == (has_short != 0) && (!check_excit || b.has_excit == (has_excit != 0)) && same_conv(b.cv1, cv1) &&
            (!has_short || same_conv(b.cvs, cvs))) return k;
    }
    return -1;
}
static int blk_kind_b(int Cin, int Cout, int L1, int Lout, const raae_conv_t& cv2, int has_short, int has_excit,
                      const raae_conv_t& cve) {
    for (int k = 0; k < kNumBlkShapes; ++k) {
        const BlkShape& b = kBlk[k];
        if (b.Cin == Cin && b.Cout == Cout && b.L1 == L1 && b.Lout == Lout && b.has_short == (has_short != 0) &&
            b.has_excit == (has_excit != 0) && same_conv(b.cv2, cv2) && (!has_excit || same_conv(b.cve, cve))) return k;
    }
    return -1;
}

static int prep_block_fwd_a(const raae_block_fwd_a_t* in, raae_block_fwd_a_t& a, int& grid, size_t& lds, int& kind) {
    RAAE_CHECK_ARG(in && in->B > 0 && in->Cin >= 1 && in->Cin <= CT_MAXCH && in->Cout >= 1 && in->Cout <= CT_MAXCH);
    RAAE_CHECK_ARG(view_ok(&in->in, in->Cin) && !in->in.mask && conv_ok(&in->cv1) && (!in->has_short || conv_ok(&in->cvs)));
    RAAE_CHECK_ARG(in->T1 && in->E1 && in->E2 && in->pT1 && (!in->has_short || in->Sh));
    RAAE_CHECK_ARG(in->cv1.Cin == in->Cin && in->cv1.Cout == in->Cout && in->cv1.Lin == in->Lin && in->cv1.Lout == in->L1);
    RAAE_CHECK_ARG(!in->has_short || (in->cvs.Lin == in->Lin && in->cvs.Lout == in->Lout && in->cvs.pad == 0));
    a = *in;
    a.halo = tile_halo(a.cv1);
    const long wfl = conv_nw(&a.cv1) + (a.has_short ? conv_nw(&a.cvs) : 0) + 2L * a.E * (a.Lin > a.Lout ? a.Lin : a.Lout);
    const long per = (long)a.Cin * (a.Lin + 2 * a.halo) + (long)a.Cin * a.E;
    RAAE_CHECK_ARG(wfl <= 4096 && per <= kTileBudget);
    long outs = (long)a.Cout * a.L1;
    if ((long)a.Cout * a.Lout > outs) outs = (long)a.Cout * a.Lout;
    if ((long)a.Cin * a.Lout > outs) outs = (long)a.Cin * a.Lout;
    a.S = pick_S(per, outs, a.B, kTileBudget, 256);
    a.ngroups = (a.B + a.S - 1) / a.S;
    a.sh_lin = lg2(a.Lin); a.sh_l1 = lg2(a.L1); a.sh_lout = lg2(a.Lout); a.sh_e = lg2(a.E);
    grid = a.ngroups < RAAE_MAX_PARTS ? a.ngroups : RAAE_MAX_PARTS;
    lds = sizeof(float) * ((size_t)a.S * per + conv_nw(&a.cv1) + (a.has_short ? conv_nw(&a.cvs) : 0) +
                                        (size_t)a.E * a.Lin + (size_t)a.Lout * a.E);
    kind = blk_kind_a(a.Cin, a.Cout, a.Lin, a.L1, a.Lout, a.E, a.cv1, a.has_short, a.cvs, 0, false);
    return 0;
}

extern "C" int raae_block_fwd_a(const raae_block_fwd_a_t* in, int* nparts, void* stream) {
    raae_block_fwd_a_t a;
    int grid, kind;
    size_t lds;
    const int rc = prep_block_fwd_a(in, a, grid, lds, kind);
    if (rc) return rc;
    if (nparts) *nparts = grid;
    const bool big = use_big(a.B, kind, kFamFwdA);
    RAAE_LAUNCH_KIND_BIG(block_fwd_a_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a)
    RAAE_LAUNCH_RET();
}

static int prep_block_fwd_b(const raae_block_fwd_b_t* in, raae_block_fwd_b_t& a, int& grid, size_t& lds, int& kind) {
    RAAE_CHECK_ARG(in && in->B > 0 && in->Cin >= 1 && in->Cin <= CT_MAXCH && in->Cout >= 1 && in->Cout <= CT_MAXCH);
    RAAE_CHECK_ARG(view_ok(&in->vT1, in->Cout) && view_ok(&in->vE2, in->Cin) && conv_ok(&in->cv2));
    RAAE_CHECK_ARG(in->has_short ? (in->Sh && in->ss) : (view_ok(&in->vR, in->Cin) && in->Cin == in->Cout));
    RAAE_CHECK_ARG(in->has_excit ? (conv_ok(&in->cve) && in->E3 && in->cve.K == 1) : in->Cin == in->Cout);
    RAAE_CHECK_ARG(in->T2 && in->Y && in->pY && in->cv2.Lin == in->L1 && in->cv2.Lout == in->Lout);
    a = *in;
    a.halo2 = tile_halo(a.cv2);
    RAAE_CHECK_ARG(a.cv2.transposed || !a.cv2.pad_replicate);
    const long per = (long)a.Cout * (a.L1 + 2 * a.halo2) + (a.has_excit ? (long)a.Cin * a.Lout : 0);
    RAAE_CHECK_ARG(per <= kTileBudget);
    a.S = pick_S(per, (long)a.Cout * a.Lout, a.B, kTileBudget, 256);
    a.ngroups = (a.B + a.S - 1) / a.S;
    a.sh_l1 = lg2(a.L1); a.sh_lout = lg2(a.Lout);
    grid = a.ngroups < RAAE_MAX_PARTS ? a.ngroups : RAAE_MAX_PARTS;
    lds = sizeof(float) * ((size_t)a.S * per + conv_nw(&a.cv2) + (a.has_excit ? conv_nw(&a.cve) : 0));
    kind = blk_kind_b(a.Cin, a.Cout, a.L1, a.Lout, a.cv2, a.has_short, a.has_excit, a.cve);
    return 0;
}

extern "C" int raae_block_fwd_b(const raae_block_fwd_b_t* in, int* nparts, void* stream) {
    raae_block_fwd_b_t a;
    int grid, kind;
    size_t lds;
    const int rc = prep_block_fwd_b(in, a, grid, lds, kind);
    if (rc) return rc;
    if (nparts) *nparts = grid;
    const bool big = use_big(a.B, kind, kFamFwdB);
    RAAE_LAUNCH_KIND_BIG(block_fwd_b_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a)
    RAAE_LAUNCH_RET();
}

// The same forward phase of TWO residual blocks that do not depend on each other (one of the encoder, one of the
// decoder: the forward chain whose result the reference throws away runs beside a forward chain that is needed)
// in ONE launch: workgroups [0, n1) run the first block, the rest the second.
struct FwdA2Args { BlockFwdAArgs x; BlockFwdAArgs y; int n1; };
struct FwdB2Args { BlockFwdBArgs x; BlockFwdBArgs y; int n1; };
template <int K1, int K2>
__global__ __launch_bounds__(256) void block_fwd_a2_kernel(FwdA2Args k) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ BlockFwdAArgs sa;
    const int n1 = k.n1;
    if ((int)blockIdx.x < n1) {
        const BlockFwdAArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(FwdA2Args, x));
        block_fwd_a_body<K1>(a, blockIdx.x, n1, dyn);
    } else {
        const BlockFwdAArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(FwdA2Args, y));
        block_fwd_a_body<K2>(a, blockIdx.x - n1, gridDim.x - n1, dyn);
    }
}
template <int K1, int K2>
__global__ __launch_bounds__(256) void block_fwd_b2_kernel(FwdB2Args k) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ BlockFwdBArgs sa;
    const int n1 = k.n1;
    if ((int)blockIdx.x < n1) {
        const BlockFwdBArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(FwdB2Args, x));
        block_fwd_b_body<K1>(a, blockIdx.x, n1, dyn);
    } else {
        const BlockFwdBArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(FwdB2Args, y));
        block_fwd_b_body<K2>(a, blockIdx.x - n1, gridDim.x - n1, dyn);
    }
}

template <int K1, int K2>
__global__ __launch_bounds__(256) void block_fwd_a2_kernel_m(const FwdA2Args* table) {      // one trial per grid plane
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ BlockFwdAArgs sa;
    const FwdA2Args* k = table + blockIdx.z;
    const int n1 = k->n1;
    if ((int)blockIdx.x < n1) {
        const BlockFwdAArgs& a = raae::args_from_ptr(&sa, &k->x);
        block_fwd_a_body<K1>(a, blockIdx.x, n1, dyn);
    } else {
        const BlockFwdAArgs& a = raae::args_from_ptr(&sa, &k->y);
        block_fwd_a_body<K2>(a, blockIdx.x - n1, gridDim.x - n1, dyn);
    }
}
template <int K1, int K2>
__global__ __launch_bounds__(256) void block_fwd_b2_kernel_m(const FwdB2Args* table) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ BlockFwdBArgs sa;
    const FwdB2Args* k = table + blockIdx.z;
    const int n1 = k->n1;
    if ((int)blockIdx.x < n1) {
        const BlockFwdBArgs& a = raae::args_from_ptr(&sa, &k->x);
        block_fwd_b_body<K1>(a, blockIdx.x, n1, dyn);
    } else {
        const BlockFwdBArgs& a = raae::args_from_ptr(&sa, &k->y);
        block_fwd_b_body<K2>(a, blockIdx.x - n1, gridDim.x - n1, dyn);
    }
}

// instances: encoder block i beside decoder block i of the 256-point networks; anything else: two launches
#define RAAE_FWD_PAIRS(KERNEL) \
    if (k1 == 0 && k2 == 3) { raae::launch(KERNEL<0, 3>, KERNEL##_m<0, 3>, grid, dim3(256), lds, (hipStream_t)stream, k); RAAE_LAUNCH_RET(); } \
    if (k1 == 1 && k2 == 4) { raae::launch(KERNEL<1, 4>, KERNEL##_m<1, 4>, grid, dim3(256), lds, (hipStream_t)stream, k); RAAE_LAUNCH_RET(); } \
    if (k1 == 2 && k2 == 5) { raae::launch(KERNEL<2, 5>, KERNEL##_m<2, 5>, grid, dim3(256), lds, (hipStream_t)stream, k); RAAE_LAUNCH_RET(); }

extern "C" int raae_block_fwd_a2(const raae_block_fwd_a_t* x, const raae_block_fwd_a_t* y, int* nparts_x, int* nparts_y,
                                 void* stream) {
    static thread_local FwdA2Args k;
    int g1, g2, k1, k2;
    size_t l1, l2;
    int rc = prep_block_fwd_a(x, k.x, g1, l1, k1);
    if (rc) return rc;
    rc = prep_block_fwd_a(y, k.y, g2, l2, k2);
    if (rc) return rc;
    if (nparts_x) *nparts_x = g1;
    if (nparts_y) *nparts_y = g2;
    k.n1 = g1;
    const size_t lds = l1 > l2 ? l1 : l2;
    const dim3 grid(g1 + g2);
    RAAE_FWD_PAIRS(block_fwd_a2_kernel)
    { const int kind = k1; const raae_block_fwd_a_t& a = k.x; const bool big = use_big(a.B, kind, kFamFwdA);
      RAAE_LAUNCH_KIND_BIG(block_fwd_a_kernel, dim3(g1), dim3(256), l1, (hipStream_t)stream, a) }
    { const int kind = k2; const raae_block_fwd_a_t& a = k.y; const bool big = use_big(a.B, kind, kFamFwdA);
      RAAE_LAUNCH_KIND_BIG(block_fwd_a_kernel, dim3(g2), dim3(256), l2, (hipStream_t)stream, a) }
    RAAE_LAUNCH_RET();
}

extern "C" int raae_block_fwd_b2(const raae_block_fwd_b_t* x, const raae_block_fwd_b_t* y, int* nparts_x, int* nparts_y,
                                 void* stream) {
    static thread_local FwdB2Args k;
    int g1, g2, k1, k2;
    size_t l1, l2;
    int rc = prep_block_fwd_b(x, k.x, g1, l1, k1);
    if (rc) return rc;
    rc = prep_block_fwd_b(y, k.y, g2, l2, k2);
    if (rc) return rc;
    if (nparts_x) *nparts_x = g1;
    if (nparts_y) *nparts_y = g2;
    k.n1 = g1;
    const size_t lds = l1 > l2 ? l1 : l2;
    const dim3 grid(g1 + g2);
    RAAE_FWD_PAIRS(block_fwd_b2_kernel)
    { const int kind = k1; const raae_block_fwd_b_t& a = k.x; const bool big = use_big(a.B, kind, kFamFwdB);
      RAAE_LAUNCH_KIND_BIG(block_fwd_b_kernel, dim3(g1), dim3(256), l1, (hipStream_t)stream, a) }
    { const int kind = k2; const raae_block_fwd_b_t& a = k.y; const bool big = use_big(a.B, kind, kFamFwdB);
      RAAE_LAUNCH_KIND_BIG(block_fwd_b_kernel, dim3(g2), dim3(256), l2, (hipStream_t)stream, a) }
    RAAE_LAUNCH_RET();
}
#undef RAAE_FWD_PAIRS

// checks + launch geometry of backward phase B (shared by raae_block_bwd_b and raae_block_bwd_b_wgrad)
static int prep_block_bwd_b(const raae_block_bwd_b_t* in, raae_block_bwd_b_t& a, int& grid, size_t& lds, int& kind) {
    RAAE_CHECK_ARG(in && in->B > 0 && in->Cin >= 1 && in->Cin <= CT_MAXCH && in->Cout >= 1 && in->Cout <= CT_MAXCH);
    RAAE_CHECK_ARG(in->gy.g && (!in->gy.has_bn || (in->gy.u && in->gy.g_partials && in->gy.bn.partials &&
                                                  in->gy.g_nparts > 0 && in->gy.g_nparts <= RAAE_MAX_PARTS)));
    RAAE_CHECK_ARG(view_ok(&in->vT1, in->Cout) && in->vT1.has_bn && conv_ok(&in->cv2));
    RAAE_CHECK_ARG(!in->has_excit || (view_ok(&in->vE2, in->Cin) && in->vE2.has_bn && conv_ok(&in->cve) && in->dBnE && in->pdBnE));
    RAAE_CHECK_ARG(in->has_excit || in->Cin == in->Cout);
    RAAE_CHECK_ARG(in->T2 && in->Ex && in->dT2 && in->dSh && in->dEx && in->dBn2 && in->pdBn2 && in->dslope2 && in->dslope_e);
    RAAE_CHECK_ARG(!in->has_short || (in->Sh && in->ss && in->dslope_s));
    a = *in;
    kind = blk_kind_b(a.Cin, a.Cout, a.L1, a.Lout, a.cv2, a.has_short, a.has_excit, a.cve);
    // strip instances (block_bwd_b_body: kStripB) keep an 8-float zero margin on both sides of every dT2 row
    const int gm = (kind >= 0 && strip_conv_ok(kBlk[kind].cv2)) ? kStripMargin : 0;
    const long per = (long)a.Cout * (a.Lout + 2 * gm) + (a.has_excit ? (long)a.Cout * a.Lout : 0);
    RAAE_CHECK_ARG(per <= kTileBudget);
    long widest = (long)a.Cout * a.Lout;
    if ((long)a.Cout * a.L1 > widest) widest = (long)a.Cout * a.L1;
    a.S = pick_S(per, widest, a.B, kTileBudget, 256);
    a.ngroups = (a.B + a.S - 1) / a.S;
    a.sh_l1 = lg2(a.L1); a.sh_lout = lg2(a.Lout);
    grid = a.ngroups < 512 ? a.ngroups : 512;
    lds = sizeof(float) * ((size_t)a.S * per + conv_nw(&a.cv2) + (a.has_excit ? conv_nw(&a.cve) : 0));
    return 0;
}

extern "C" int raae_block_bwd_b(const raae_block_bwd_b_t* in, int* nparts, void* stream) {
    raae_block_bwd_b_t a;
    int grid, kind;
    size_t lds;
    const int rc = prep_block_bwd_b(in, a, grid, lds, kind);
    if (rc) return rc;
    if (nparts) *nparts = grid;
    const bool big = use_big(a.B, kind, kFamBwdB);
    RAAE_LAUNCH_KIND_BIG(block_bwd_b_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a)
    RAAE_LAUNCH_RET();
}

extern "C" int raae_block_bwd_a(const raae_block_bwd_a_t* in, int* nparts, void* stream) {
    RAAE_CHECK_ARG(in && in->B > 0 && in->Cin >= 1 && in->Cin <= CT_MAXCH && in->Cout >= 1 && in->Cout <= CT_MAXCH);
    RAAE_CHECK_ARG(grad_ok(&in->g1, in->Cout) && in->g1.has_bn && in->g1.raw && in->g1.slope);
    RAAE_CHECK_ARG(!in->has_excit || (grad_ok(&in->ge, in->Cin) && in->ge.raw && in->ge.slope && in->dslope_e2));
    RAAE_CHECK_ARG(view_ok(&in->in, in->Cin) && !in->in.mask && conv_ok(&in->cv1) && (!in->has_short || conv_ok(&in->cvs)));
    RAAE_CHECK_ARG(in->has_short || (in->Cin == in->Cout && in->Lin == in->Lout));
    RAAE_CHECK_ARG(in->E1 && in->dSh && in->dT1 && in->dE2 && in->dE1 && in->dslope1 && in->dslope_e1 && in->se1);
    RAAE_CHECK_ARG(!in->pdR || (in->dR && in->in.has_bn));
    raae_block_bwd_a_t a = *in;
    const int kind = blk_kind_a(a.Cin, a.Cout, a.Lin, a.L1, a.Lout, a.E, a.cv1, a.has_short, a.cvs, a.has_excit, true);
    // strip instances (block_bwd_a_kernel: kStripA) keep an 8-float zero margin on both sides of every dT1 row
    const int gm = (kind >= 0 && strip_conv_ok(kBlk[kind].cv1) && !kBlk[kind].has_short) ? kStripMargin : 0;
    const long wfl = conv_nw(&a.cv1) + (a.has_short ? conv_nw(&a.cvs) : 0) + (long)a.E * a.Lin + (long)a.Lout * a.E;
    const long per = (long)a.Cout * (a.L1 + 2 * gm) + (long)a.Cout * a.Lout + (long)a.Cin * a.Lout + (long)a.Cin * a.E;
    RAAE_CHECK_ARG(wfl <= 4096 && per <= kTileBudget);
    long widest = (long)a.Cin * a.Lin;
    if ((long)a.Cout * a.L1 > widest) widest = (long)a.Cout * a.L1;
    if ((long)a.Cout * a.Lout > widest) widest = (long)a.Cout * a.Lout;
    if ((long)a.Cin * a.Lout > widest) widest = (long)a.Cin * a.Lout;
    a.S = pick_S(per, widest, a.B, kTileBudget, 256);
    a.ngroups = (a.B + a.S - 1) / a.S;
    a.sh_lin = lg2(a.Lin); a.sh_l1 = lg2(a.L1); a.sh_lout = lg2(a.Lout); a.sh_e = lg2(a.E);
    const int grid = a.ngroups < 512 ? a.ngroups : 512;
    if (nparts) *nparts = grid;
    const size_t lds = sizeof(float) * ((size_t)a.S * per + wfl);
    const bool big = use_big(a.B, kind, kFamBwdA);
    RAAE_LAUNCH_KIND_BIG(block_bwd_a_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a)
    RAAE_LAUNCH_RET();
}

// workgroups per weight-gradient task (= slabs it writes); 64 -> 128: +2 % at B=256, +18 % at B=4096
static const int kWgradTaskGridDefault = 128;
// tuning knobs (environment, read once): RAAE_WGRAD_GRID workgroups per task, RAAE_WGRAD_S samples per group (0: automatic)
static const int kWgradTaskGrid = env_int("RAAE_WGRAD_GRID", kWgradTaskGridDefault);
static const int kWgradForceS = env_int("RAAE_WGRAD_S", 0);
// checks + task table + launch geometry of a block's weight-gradient tasks; `m` is filled
static int prep_block_wgrad(const raae_block_wgrad_t* in, int* nslab, WgradMultiArgs& m, int& total_out, size_t& dyn_out,
                            int& kind_out) {
    RAAE_CHECK_ARG(in && nslab && in->B > 0 && in->n_conv >= 0 && in->n_conv <= 4 && in->n_lin >= 0 && in->n_lin <= 2 &&
                   in->n_conv + in->n_lin > 0);
    m.ntask = 0;
    int total = 0;
    size_t dyn = 0;
    // raae_tile_hint: a caller that launches T trials at once gets T times the workgroups anyway
    const int kWgradTaskGrid = ::kWgradTaskGrid / g_tile_mult > 16 ? ::kWgradTaskGrid / g_tile_mult : 16;
    for (int i = 0; i < in->n_conv; ++i) {
        const raae_wgrad_conv_t& c = in->conv[i];
        const raae_conv_t* cv = &c.cv;
        RAAE_CHECK_ARG(conv_ok(cv) && grad_ok(&c.go, cv->Cout) && !c.go.has_bn && !c.go.slope && view_ok(&c.in, cv->Cin) &&
                       c.dw && c.dbias && cv->Cout <= 8);
        ConvBwdWTArgs& t = m.conv[i];
        t.a.go = c.go; t.a.B = in->B; t.a.cv = *cv; t.a.in = c.in; t.a.dw = c.dw; t.a.dbias = c.dbias; t.a.dslope = nullptr;
        t.a.nw = conv_nw(cv);
        const long per = (long)cv->Cout * cv->Lout + (long)cv->Cin * (cv->Lin + 2 * (cv->transposed ? 0 : cv->pad));
        RAAE_CHECK_ARG(t.a.nw <= 1024 && per <= kTileBudget);
        t.slab_stride = in->slab_stride; t.sh_in = lg2(cv->Lin); t.sh_out = lg2(cv->Lout);
        t.S = pick_S(per, cv->transposed ? cv->Lin : cv->Lout, in->B, kTileBudget, 256);
        if (kWgradForceS > 0 && in->B >= RAAE_BIG_ROWS) { t.S = kWgradForceS; const long cap = (36 * 1024) / per; if (t.S > cap) t.S = (int)cap; if (t.S < 1) t.S = 1; }
        { const int scap = (in->B + kWgradTaskGrid - 1) / kWgradTaskGrid; if (t.S > scap) t.S = scap; }   // parallelism from workgroups, not from samples per group
        t.ngroups = (in->B + t.S - 1) / t.S;
        const int grid = t.ngroups < kWgradTaskGrid ? t.ngroups : kWgradTaskGrid;
        m.first[m.ntask] = total; m.is_conv[m.ntask] = 1; m.idx[m.ntask] = i; nslab[m.ntask] = grid;
        total += grid; ++m.ntask;
        const size_t d = sizeof(float) * (size_t)t.S * per;
        if (d > dyn) dyn = d;
    }
    for (int i = 0; i < in->n_lin; ++i) {
        const raae_wgrad_lin_t& c = in->lin[i];
        RAAE_CHECK_ARG(grad_ok(&c.go, c.C) && !c.go.has_bn && !c.go.slope && view_ok(&c.in, c.C) && c.dw && c.dbias &&
                       c.C >= 1 && c.C <= CT_MAXCH && c.E >= 1 && c.E <= 256 && c.Lin >= 1 && (long)c.E * c.Lin <= 1024);
        LenLinBwdWTArgs& t = m.lin[i];
        t.a.go = c.go; t.a.B = in->B; t.a.C = c.C; t.a.E = c.E; t.a.in = c.in; t.a.Lin = c.Lin; t.a.dw = c.dw;
        t.a.dbias = c.dbias; t.a.dslope = nullptr;
        const long per = (long)c.C * c.E + (long)c.C * c.Lin;
        RAAE_CHECK_ARG(per <= kTileBudget);
        t.slab_stride = in->slab_stride; t.sh_in = lg2(c.Lin); t.sh_e = lg2(c.E);
        t.S = pick_S(per, c.C, in->B, kTileBudget, 64);
        if (kWgradForceS > 0 && in->B >= RAAE_BIG_ROWS) { t.S = kWgradForceS; const long cap = (36 * 1024) / per; if (t.S > cap) t.S = (int)cap; if (t.S < 1) t.S = 1; }
        { const int scap = (in->B + kWgradTaskGrid - 1) / kWgradTaskGrid; if (t.S > scap) t.S = scap; }   // parallelism from workgroups, not from samples per group
        t.ngroups = (in->B + t.S - 1) / t.S;
        const int grid = t.ngroups < kWgradTaskGrid ? t.ngroups : kWgradTaskGrid;
        m.first[m.ntask] = total; m.is_conv[m.ntask] = 0; m.idx[m.ntask] = i; nslab[m.ntask] = grid;
        total += grid; ++m.ntask;
        const size_t d = sizeof(float) * (size_t)t.S * per;
        if (d > dyn) dyn = d;
    }
    m.first[m.ntask] = total;
    // shape-specialised instance: every conv task must be one of the convs of a known block shape
    int kind = -1;
    for (int k = 0; k < kNumBlkShapes && kind < 0; ++k) {
        const BlkShape& b = kBlk[k];
        bool all = in->n_conv > 0;
        for (int i = 0; i < in->n_conv && all; ++i) {
            const raae_conv_t& cv = in->conv[i].cv;
            const int w = same_conv(cv, b.cv1) ? 1 : same_conv(cv, b.cv2) ? 2 : (b.has_excit && same_conv(cv, b.cve)) ? 3 :
                          (b.has_short && same_conv(cv, b.cvs)) ? 4 : 0;
            m.which[i] = w;
            all = w != 0;
        }
        if (all) kind = k;
    }
    if (kind < 0) for (int i = 0; i < 4; ++i) m.which[i] = 0;
    total_out = total; dyn_out = dyn; kind_out = kind;
    return 0;
}

extern "C" int raae_block_wgrad(const raae_block_wgrad_t* in, int* nslab, void* stream) {
    static thread_local WgradMultiArgs m;   // large (kernarg by value); per-thread host scratch, filled per call
    int total, kind;
    size_t dyn;
    const int rc = prep_block_wgrad(in, nslab, m, total, dyn, kind);
    if (rc) return rc;
    const bool big = use_big(in->B, kind, kFamWgrad);
    RAAE_LAUNCH_KIND_BIG(wgrad_multi_kernel, dim3(total), dim3(256), dyn, (hipStream_t)stream, m)
    RAAE_LAUNCH_RET();
}

// Backward phase B of one residual block and ALL weight-gradient tasks of the block after it (whose data
// gradients are complete) in ONE launch: workgroups [0, nb) run block_bwd_b, the rest the weight-gradient tasks.
// The two are independent, so they overlap on the chip like parallel graph branches would -- without the
// fork/join edges, which at 256-row batches cost as much as the kernels.
struct BwdBWgradArgs { BlockBwdBArgs b; WgradMultiArgs w; int nb; };
template <int KB, int KW>
__global__ __launch_bounds__(256) void block_bwd_b_wgrad_kernel(BwdBWgradArgs k) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int nb = k.nb;
    if ((int)blockIdx.x < nb) {
        __shared__ BlockBwdBArgs sa;
        const BlockBwdBArgs& a = raae::args_to_lds_at(&sa, (int)offsetof(BwdBWgradArgs, b));
        block_bwd_b_body<KB>(a, blockIdx.x, nb, dyn);
    } else {
        __shared__ WgradMultiArgs sm;
        const WgradMultiArgs& m = raae::args_to_lds_at(&sm, (int)offsetof(BwdBWgradArgs, w));
        wgrad_multi_body<KW>(m, blockIdx.x - nb, dyn);
    }
}

template <int KB, int KW>
__global__ __launch_bounds__(256) void block_bwd_b_wgrad_kernel_m(const BwdBWgradArgs* table) {     // one trial per grid plane
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const BwdBWgradArgs* k = table + blockIdx.z;
    const int nb = k->nb;
    if ((int)blockIdx.x < nb) {
        __shared__ BlockBwdBArgs sa;
        const BlockBwdBArgs& a = raae::args_from_ptr(&sa, &k->b);
        block_bwd_b_body<KB>(a, blockIdx.x, nb, dyn);
    } else {
        __shared__ WgradMultiArgs sm;
        const WgradMultiArgs& m = raae::args_from_ptr(&sm, &k->w);
        wgrad_multi_body<KW>(m, blockIdx.x - nb, dyn);
    }
}

extern "C" int raae_block_bwd_b_wgrad(const raae_block_bwd_b_t* bin, const raae_block_wgrad_t* win, int* nparts,
                                      int* nslab, void* stream) {
    static thread_local BwdBWgradArgs k;
    int gridb, kindb, total, kindw;
    size_t ldsb, dynw;
    int rc = prep_block_bwd_b(bin, k.b, gridb, ldsb, kindb);
    if (rc) return rc;
    rc = prep_block_wgrad(win, nslab, k.w, total, dynw, kindw);
    if (rc) return rc;
    if (nparts) *nparts = gridb;
    k.nb = gridb;
    const size_t lds = ldsb > dynw ? ldsb : dynw;
    const dim3 grid(gridb + total), block(256);
    // instances: phase B of block i-1 beside the weight gradients of block i, for the block sequences of the
    // 256-point networks (encoder 0,1,2; decoder 3,4,5,6); anything else runs the generic pair
#define RAAE_PAIR(KB_, KW_) if (kindb == KB_ && kindw == KW_) { \
        raae::launch(block_bwd_b_wgrad_kernel<KB_, KW_>, block_bwd_b_wgrad_kernel_m<KB_, KW_>, grid, block, lds, (hipStream_t)stream, k); RAAE_LAUNCH_RET(); }
    RAAE_PAIR(0, 1) RAAE_PAIR(1, 2) RAAE_PAIR(3, 4) RAAE_PAIR(4, 5) RAAE_PAIR(5, 6)
    RAAE_PAIR(2, 3) RAAE_PAIR(6, 0)      // across the networks: encoder's last block beside decoder block 0's tasks, and back
    RAAE_PAIR(6, -1)                     // the decoder's head conv (a lone generic task) beside its last block
#undef RAAE_PAIR
    raae::launch(block_bwd_b_wgrad_kernel<-1, -1>, block_bwd_b_wgrad_kernel_m<-1, -1>, grid, block, lds, (hipStream_t)stream, k);
    RAAE_LAUNCH_RET();
}
