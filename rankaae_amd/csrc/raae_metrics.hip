// Model-selection metrics of the validation styles on the device: the Shapiro-Wilk W of every
// style column and Spearman's rho of every column pair, which the reference obtains per epoch with
// scipy.stats.shapiro / scipy.stats.spearmanr on a host copy (sc/clustering/trainer.py:286-292).
//
// Two launches.  style_rank_kernel gives every element its average rank and its slot in the sorted column by
// counting (exact for ties, no sort network, any n); style_stat_kernel then forms, one workgroup per column or
// column pair, the W statistic with the arithmetic of Royston's AS R94 as scipy 1.15.3 runs it (squared
// correlation of the sorted, range-scaled sample with the coefficient vector, reported as 1 - (1 - W)) and the
// Pearson correlation of the rank vectors.  All sums are double, in a fixed order (strided per thread, then a
// tree), so a replay is bitwise repeatable.
#include "raae_common.h"

namespace {

constexpr int kTile = 2048;

// grid (ceil(n / 256), k).  rank[c][i] = #{x_j < x_i} + (#{x_j == x_i} + 1) / 2; the element's slot in the sorted
// column is #{x_j < x_i} + #{j < i : x_j == x_i}; sorted[c][slot] = x_i - pivot_c (scipy subtracts x[n / 2],
// "the median or a nearby value", before the W arithmetic).
struct StyleRankArgs { const float* z; int n; int k; double* rank; double* sorted; };
__device__ __forceinline__ void style_rank_body(const float* __restrict__ z, int n, int k,
                                                double* __restrict__ rank, double* __restrict__ sorted) {
    __shared__ float tile[kTile];
    const int c = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const float xi = i < n ? z[(size_t)i * k + c] : 0.f;
    int less = 0, eq = 0, eqb = 0;
    for (int j0 = 0; j0 < n; j0 += kTile) {
        const int m = min(kTile, n - j0);
        __syncthreads();
        for (int t = tid; t < m; t += 256) tile[t] = z[(size_t)(j0 + t) * k + c];
        __syncthreads();
        const int before = min(max(i - j0, 0), m);        // elements of this tile with index < i
        int jj = 0;
        for (; jj + 4 <= m; jj += 4) {
            const float4 v = *reinterpret_cast<const float4*>(tile + jj);
            less += (v.x < xi) + (v.y < xi) + (v.z < xi) + (v.w < xi);
            const int e0 = v.x == xi, e1 = v.y == xi, e2 = v.z == xi, e3 = v.w == xi;
            eq += e0 + e1 + e2 + e3;
            eqb += (e0 & (jj < before)) + (e1 & (jj + 1 < before)) + (e2 & (jj + 2 < before)) + (e3 & (jj + 3 < before));
        }
        for (; jj < m; ++jj) {
            const float v = tile[jj];
            less += v < xi;
            const int e = v == xi;
            eq += e;
            eqb += e & (jj < before);
        }
    }
    if (i < n) {
        const double pivot = (double)z[(size_t)(n / 2) * k + c];
        rank[(size_t)c * n + i] = (double)less + 0.5 * (double)(eq + 1);
        sorted[(size_t)c * n + less + eqb] = (double)xi - pivot;
    }
}

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red) {
    const int tid = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NV; ++u) red[u * 256 + tid] = v[u];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int u = 0; u < NV; ++u) red[u * 256 + tid] += red[u * 256 + tid + o];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u] = red[u * 256];
}

__global__ __launch_bounds__(256) void style_rank_kernel(StyleRankArgs a) { style_rank_body(a.z, a.n, a.k, a.rank, a.sorted); }
__global__ __launch_bounds__(256) void style_rank_kernel_m(const StyleRankArgs* t) {       // one trial per grid plane
    const StyleRankArgs a = t[blockIdx.z];
    style_rank_body(a.z, a.n, a.k, a.rank, a.sorted);
}

// grid (k + k (k - 1) / 2).  Workgroups < k: W of column blockIdx.x; the others: rho of pair (p, q), p < q, in
// itertools.combinations order.  out = [W_0 .. W_{k-1}, rho_(0,1), rho_(0,2), ...].
struct StyleStatArgs { const double* rank; const double* sorted; const double* a; int n; int k; double* out; };
__device__ __forceinline__ void style_stat_body(const double* __restrict__ rank,
                                                const double* __restrict__ sorted,
                                                const double* __restrict__ a, int n, int k,
                                                double* __restrict__ out) {
    __shared__ double red[3 * 256];
    const int tid = threadIdx.x, w = blockIdx.x;
    if (w < k) {
        const double* y = sorted + (size_t)w * n;
        const double range = y[n - 1] - y[0];
        if (range < 1e-19) {                          // AS R94 ifault 6: W is reported as 1
            if (tid == 0) out[w] = 1.0;
            return;
        }
        // antisymmetric coefficient of order statistic i: -a[i] below the middle, +a[n-1-i] above, 0 at it
        auto coef = [&](int i) -> double {
            const int j = n - 1 - i;
            return i < j ? -a[i] : (i > j ? a[j] : 0.0);
        };
        double s[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += 256) { s[0] += y[i] / range; s[1] += coef(i); }
        block_sum<2>(s, red);
        const double sx = s[0] / (double)n, sa = s[1] / (double)n;
        double t[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < n; i += 256) {
            const double asa = coef(i) - sa, xsx = y[i] / range - sx;
            t[0] += asa * asa; t[1] += xsx * xsx; t[2] += asa * xsx;
        }
        block_sum<3>(t, red);
        if (tid == 0) {
            const double ssassx = sqrt(t[0] * t[1]);
            const double w1 = (ssassx - t[2]) * (ssassx + t[2]) / (t[0] * t[1]);
            out[w] = 1.0 - w1;
        }
        return;
    }
    int p = 0, rest = w - k;
    while (rest >= k - 1 - p) { rest -= k - 1 - p; ++p; }
    const int q = p + 1 + rest;
    const double* rp = rank + (size_t)p * n;
    const double* rq = rank + (size_t)q * n;
    const double mean = 0.5 * (double)(n + 1);        // average ranks always sum to n (n + 1) / 2
    double t[3] = {0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += 256) {
        const double u = rp[i] - mean, v = rq[i] - mean;
        t[0] += u * u; t[1] += v * v; t[2] += u * v;
    }
    block_sum<3>(t, red);
    if (tid == 0) {
        double r = t[2] / sqrt(t[0]) / sqrt(t[1]);     // numpy.corrcoef: divide by each deviation, then clip
        r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
        out[w] = r;
    }
}

// out[g][l] = mean_s x[g][s][l]: the n_sampling average of the report's decoder sweeps (sc/report/analysis.py:78-86).
// One thread per output column element; the sum runs in sample order in double.
__global__ __launch_bounds__(256) void style_stat_kernel(StyleStatArgs a) { style_stat_body(a.rank, a.sorted, a.a, a.n, a.k, a.out); }
__global__ __launch_bounds__(256) void style_stat_kernel_m(const StyleStatArgs* t) {
    const StyleStatArgs a = t[blockIdx.z];
    style_stat_body(a.rank, a.sorted, a.a, a.n, a.k, a.out);
}

__global__ __launch_bounds__(256) void group_mean_kernel(const float* __restrict__ x, int groups, int per, int L,
                                                         float* __restrict__ out) {
    const long n = (long)groups * L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long g = i / L, l = i - g * L;
        const float* p = x + g * (long)per * L + l;
        double acc = 0.0;
        for (int s = 0; s < per; ++s) acc += (double)p[(long)s * L];
        out[i] = (float)(acc / (double)per);
    }
}

}  // namespace

extern "C" int raae_group_mean(const float* x, int groups, int per, int L, float* out, void* stream) {
    RAAE_CHECK_ARG(x && out && groups > 0 && per > 0 && L > 0);
    const long n = (long)groups * L;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    RAAE_PLAIN_LAUNCH(group_mean_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, groups, per, L, out);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_style_metrics(const float* z, int n, int k, const double* a_coef, double* work, double* out,
                                  void* stream) {
    RAAE_CHECK_ARG(z && a_coef && work && out && n >= 3 && k >= 1 && k <= 64);
    double* rank = work;
    double* sorted = work + (size_t)k * n;
    const StyleRankArgs ra = {z, n, k, rank, sorted};
    raae::launch(style_rank_kernel, style_rank_kernel_m, dim3(raae::cdiv(n, 256), k), dim3(256), 0, (hipStream_t)stream, ra);
    const StyleStatArgs sa = {rank, sorted, a_coef, n, k, out};
    raae::launch(style_stat_kernel, style_stat_kernel_m, dim3(k + k * (k - 1) / 2), dim3(256), 0, (hipStream_t)stream, sa);
    RAAE_LAUNCH_RET();
}
