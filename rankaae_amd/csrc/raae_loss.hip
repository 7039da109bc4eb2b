// Loss kernels (forward + closed-form gradient in one pass each) and the encoder's
// final BatchNorm.  Each replaces an ATen chain plus its autograd backward in
// reference sc/utils/functions.py (cited per kernel).
#include "raae_common.h"

namespace {

// ------------------------------------------------------------------ style BatchNorm
// styles = BN(z) for z [B][C] (C = nstyle <= 64).  grid-stride over rows.
struct StyleFwdArgs { const float* z; int B; int C; raae_bn_t bn; float* out; };
__device__ __forceinline__ void style_bn_fwd_body(const float* z, int B, int C, const raae_bn_t& bn, float* out) {
    __shared__ float s_mean[64], s_rstd[64];
    raae::bn_prologue(bn, C, s_mean, s_rstd, blockIdx.x == 0);
    const long n = (long)B * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        out[i] = (z[i] - s_mean[c]) * s_rstd[c];
    }
}

__global__ __launch_bounds__(256) void style_bn_fwd_kernel(StyleFwdArgs a) { style_bn_fwd_body(a.z, a.B, a.C, a.bn, a.out); }
__global__ __launch_bounds__(256) void style_bn_fwd_kernel_m(const StyleFwdArgs* t) {
    const StyleFwdArgs a = t[blockIdx.z];
    style_bn_fwd_body(a.z, a.B, a.C, a.bn, a.out);
}

// dz = rstd * (g - mean(g) - y * mean(g*y)), g = scale * dstyles.  Single workgroup: the
// column sums over the whole batch are formed in fixed order (deterministic).
struct StyleBwdArgs { const float* dy; const float* y; int B; int C; raae_bn_t bn; float scale; float* dz; };
__device__ __forceinline__ void style_bn_bwd_body(const float* dy, const float* y, int B, int C,
                                                  const raae_bn_t& bn, float scale, float* dz) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ double s_s[64], s_q[64];
    __shared__ double red[2][1024];
    raae::bn_prologue(bn, C, s_mean, s_rstd, false);
    // thread -> column c = tid % Cp, row phase = tid / Cp
    const int tid = threadIdx.x;
    const int nphase = 1024 / C;
    const int c = tid % C, ph = tid / C;
    double s = 0.0, q = 0.0;
    if (ph < nphase) {
        // four rows per trip, their loads issued together (one dependent round trip per row was 20 us at 4096 rows)
        int r = ph;
        for (; r + 3 * nphase < B; r += 4 * nphase) {
            float g[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { g[u] = dy[(size_t)(r + u * nphase) * C + c]; yv[u] = y[(size_t)(r + u * nphase) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float gg = scale * g[u]; s += (double)gg; q += (double)gg * (double)yv[u]; }
        }
        for (; r < B; r += nphase) {
            const float g = scale * dy[(size_t)r * C + c];
            s += (double)g;
            q += (double)g * (double)y[(size_t)r * C + c];
        }
    }
    red[0][tid] = s; red[1][tid] = q;
    __syncthreads();
    // per column: 8 threads add every 8th phase partial, then one thread adds the 8 (fixed order)
    __shared__ double red2[2][64 * 8];
    if (tid < C * 8) {
        const int cc = tid >> 3, part = tid & 7;
        double ts = 0.0, tq = 0.0;
        for (int p = part; p < nphase; p += 8) { ts += red[0][p * C + cc]; tq += red[1][p * C + cc]; }
        red2[0][tid] = ts; red2[1][tid] = tq;
    }
    __syncthreads();
    if (tid < C) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) { ts += red2[0][tid * 8 + p]; tq += red2[1][tid * 8 + p]; }
        s_s[tid] = ts / (double)B; s_q[tid] = tq / (double)B;
    }
    __syncthreads();
    if (ph < nphase) {
        const float m1 = (float)s_s[c], m2 = (float)s_q[c], rs = s_rstd[c];
        int r = ph;
        for (; r + 3 * nphase < B; r += 4 * nphase) {
            float g[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { g[u] = dy[(size_t)(r + u * nphase) * C + c]; yv[u] = y[(size_t)(r + u * nphase) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) dz[(size_t)(r + u * nphase) * C + c] = rs * (scale * g[u] - m1 - yv[u] * m2);
        }
        for (; r < B; r += nphase) {
            const size_t o = (size_t)r * C + c;
            dz[o] = rs * (scale * dy[o] - m1 - y[o] * m2);
        }
    }
}

__global__ __launch_bounds__(1024) void style_bn_bwd_kernel(StyleBwdArgs a) { style_bn_bwd_body(a.dy, a.y, a.B, a.C, a.bn, a.scale, a.dz); }
__global__ __launch_bounds__(1024) void style_bn_bwd_kernel_m(const StyleBwdArgs* t) {
    const StyleBwdArgs a = t[blockIdx.z];
    style_bn_bwd_body(a.dy, a.y, a.B, a.C, a.bn, a.scale, a.dz);
}

// ------------------------------------------------------------------ rank ("Kendall") loss
// functions.py:37-79.  One pass over the B^2 pairs; per (i,k): g+ = sum_{j: p>0} s, g- = sum_{j: p<0} s;
// per k: n+, n-, S+ = sum_{p>0} p, S- = sum_{p<0} p.  Thread owns one row i for ALL k (n_aux <= 16)
// and streams the j rows through LDS tiles; counts are integers (exact), sums fp32 per thread,
// double across threads.
#define RANK_MAXK 16
#define RANK_TJ 256
struct RankWork {   // per-workgroup partial: [k]{n_pos, n_neg} ints and {S_pos, S_neg} doubles
    long long n_pos[RANK_MAXK], n_neg[RANK_MAXK];
    double s_pos[RANK_MAXK], s_neg[RANK_MAXK];
};

#define RANK_JC 8      // lanes that share one (row, descriptor) and split the j range
// Thread = (row i, descriptor k, j-chunk c); R rows per thread (register blocking for large B).
// The 8 chunk lanes of a (row, k) are adjacent lanes: their partial results meet through
// wavefront shuffles (xor 1, 2, 4).  Workgroups grid-stride over row groups.
// 2-D tiling for large batches: workgroup = (row-group slice ib, column block jb); block jb streams the columns
// [jb * jchunk, (jb + 1) * jchunk) and leaves per-(row, k) partial g+- for that block, which the finalize adds in
// block order (fixed order => deterministic; the counts are integers and exact either way).  At B = 4096 this is
// 171 row groups x 6 column blocks = 1026 workgroups for the 256 CUs instead of 171.
// Rows are [row0, row0 + nrows) of arrays with n_all rows, columns ALL n_all rows: nrows == n_all is the
// reference's loss over one batch; nrows < n_all is one rank's share of the global pairs under data parallelism.
struct RankPairsArgs { const float* d; int ldd; const float* z; int ldz; int n_all; int row0; int nrows; int nj; int jchunk;
                       RankWork* part; float* gpos; float* gneg; };
template <int KA, int R>
__device__ __forceinline__ void rank_pairs_body(const float* d, int ldd, const float* z, int ldz, int n_all,
                                                int row0, int nrows, int nj, int jchunk,
                                                RankWork* part, float* gpos, float* gneg) {
    constexpr int IPB = 32 / KA;            // (row, k) slots per block = IPB * KA <= 32
    __shared__ float sd[RANK_TJ * KA], sz[RANK_TJ * KA];
    __shared__ double red_s[2][32];
    __shared__ long long red_n[2][32];
    const int tid = threadIdx.x, c = tid & (RANK_JC - 1), q = tid >> 3;
    const int il = q / KA, k = q - il * KA;
    const bool slot = il < IPB;
    double tsp = 0.0, tsn = 0.0;            // block totals of this (il, k) slot over all its row groups
    long long tnp = 0, tnn = 0;
    const int rows_per_group = IPB * R;
    const int ngroups = (nrows + rows_per_group - 1) / rows_per_group;
    const int jb = blockIdx.x % nj, ib = blockIdx.x / nj, ni_wg = gridDim.x / nj;
    const int jlo = jb * jchunk, jhi = min(n_all, jlo + jchunk);
    for (int grp = ib; grp < ngroups; grp += ni_wg) {
        float di[R], zi[R], gp[R], gn[R], sp[R], sn[R];
        int np[R], nn[R], irow[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            irow[r] = grp * rows_per_group + r * IPB + il;           // local row index
            const bool ok = slot && irow[r] < nrows;
            di[r] = ok ? d[(size_t)(row0 + irow[r]) * ldd + k] : 0.f;
            zi[r] = ok ? z[(size_t)(row0 + irow[r]) * ldz + k] : 0.f;
            gp[r] = gn[r] = sp[r] = sn[r] = 0.f; np[r] = nn[r] = 0;
        }
        for (int j0 = jlo; j0 < jhi; j0 += RANK_TJ) {
            const int ntile = min(RANK_TJ, jhi - j0);
            __syncthreads();
            for (int idx = tid; idx < ntile * KA; idx += 256) {
                const int jj = idx / KA, kk = idx - jj * KA;
                sd[idx] = d[(size_t)(j0 + jj) * ldd + kk];
                sz[idx] = z[(size_t)(j0 + jj) * ldz + kk];
            }
            __syncthreads();
            if (slot) {
                for (int jj = c; jj < ntile; jj += RANK_JC) {
                    const float dj = sd[jj * KA + k], zj = sz[jj * KA + k];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float dd = di[r] - dj;
                        const float sg = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
                        const float p = (zi[r] - zj) * sg;
                        if (p > 0.f) { np[r]++; sp[r] += p; gp[r] += sg; }
                        else if (p < 0.f) { nn[r]++; sn[r] += p; gn[r] += sg; }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // rows beyond B hold di = zi = 0 against real j rows: discard them
            const bool ok = slot && irow[r] < nrows;
            float a = gp[r], b2 = gn[r];
            double e = (double)sp[r], f = (double)sn[r];
            int g = np[r], h = nn[r];
#pragma unroll
            for (int o = 1; o < RANK_JC; o <<= 1) {
                a += __shfl_xor(a, o, 64); b2 += __shfl_xor(b2, o, 64);
                e += __shfl_xor(e, o, 64); f += __shfl_xor(f, o, 64);
                g += __shfl_xor(g, o, 64); h += __shfl_xor(h, o, 64);
            }
            if (ok && c == 0) {
                gpos[((size_t)jb * nrows + irow[r]) * KA + k] = a; gneg[((size_t)jb * nrows + irow[r]) * KA + k] = b2;
                tsp += e; tsn += f; tnp += g; tnn += h;
            }
        }
    }
    // block totals per k: slots (il, k) combine in fixed order
    __syncthreads();
    if (c == 0) { red_s[0][q] = tsp; red_s[1][q] = tsn; red_n[0][q] = tnp; red_n[1][q] = tnn; }
    __syncthreads();
    if (tid < KA) {
        double a = 0.0, b2 = 0.0; long long g = 0, h = 0;
        for (int i2 = 0; i2 < IPB; ++i2) {
            a += red_s[0][i2 * KA + tid]; b2 += red_s[1][i2 * KA + tid];
            g += red_n[0][i2 * KA + tid]; h += red_n[1][i2 * KA + tid];
        }
        part[blockIdx.x].s_pos[tid] = a; part[blockIdx.x].s_neg[tid] = b2;
        part[blockIdx.x].n_pos[tid] = g; part[blockIdx.x].n_neg[tid] = h;
    }
}

template <int KA, int R>
__global__ __launch_bounds__(256) void rank_pairs_kernel(RankPairsArgs a) {
    rank_pairs_body<KA, R>(a.d, a.ldd, a.z, a.ldz, a.n_all, a.row0, a.nrows, a.nj, a.jchunk, a.part, a.gpos, a.gneg);
}
template <int KA, int R>
__global__ __launch_bounds__(256) void rank_pairs_kernel_m(const RankPairsArgs* t) {
    const RankPairsArgs a = t[blockIdx.z];
    rank_pairs_body<KA, R>(a.d, a.ldd, a.z, a.ldz, a.n_all, a.row0, a.nrows, a.nj, a.jchunk, a.part, a.gpos, a.gneg);
}

// finalize: c_k, loss, dz[i][k] = -(2/norm)(c_k g+ + g-).  Every workgroup re-derives c_k from the
// partials (16 slices per descriptor, fixed order), then grid-strides over dz.
// `totals` != NULL: the per-descriptor totals {n+, n-, S+, S-} [4][16] doubles are given (summed over the ranks of a
// data-parallel run: global pairs) and `part` is not read.  norm = (n_all^2 - n_all) * KA; `scale` multiplies dz
// (the number of ranks, whose gradients the engine AVERAGES).
struct RankFinArgs { const RankWork* part; int nparts; const double* totals; int n_all; int nrows; int nj; int KA; int activate;
                     float scale; const float* gpos; const float* gneg; float* loss; float* dz; int ldz; };
__device__ __forceinline__ void rank_finalize_body(const RankWork* part, int nparts, const double* totals, int n_all,
                                                   int nrows, int nj, int KA, int activate, float scale,
                                                   const float* gpos, const float* gneg, float* loss,
                                                   float* dz, int ldz) {
    __shared__ float s_c[RANK_MAXK];
    __shared__ double s_loss[RANK_MAXK];
    __shared__ double r_s[2][256];
    __shared__ long long r_n[2][256];
    const int tid = threadIdx.x;
    const double norm = ((double)n_all * (double)n_all - (double)n_all) * (double)KA;
    if (totals == nullptr) {
        const int k = tid & 15, sl = tid >> 4;          // 16 slices
        long long np = 0, nn = 0; double sp = 0.0, sn = 0.0;
        if (k < KA) {
            // eight records' loads in flight per trip (up to 2048 records: 128 dependent round trips per lane otherwise,
            // 36 us at 4096 rows); the sums keep their order
#pragma unroll 8
            for (int p = sl; p < nparts; p += 16) {
                np += part[p].n_pos[k]; nn += part[p].n_neg[k];
                sp += part[p].s_pos[k]; sn += part[p].s_neg[k];
            }
        }
        r_s[0][tid] = sp; r_s[1][tid] = sn; r_n[0][tid] = np; r_n[1][tid] = nn;
    }
    __syncthreads();
    if (tid < KA) {
        long long np = 0, nn = 0; double sp = 0.0, sn = 0.0;
        if (totals == nullptr) {
            for (int sl = 0; sl < 16; ++sl) {
                sp += r_s[0][sl * 16 + tid]; sn += r_s[1][sl * 16 + tid];
                np += r_n[0][sl * 16 + tid]; nn += r_n[1][sl * 16 + tid];
            }
        } else {
            np = (long long)totals[tid]; nn = (long long)totals[16 + tid];      // exact: counts < 2^53
            sp = totals[32 + tid]; sn = totals[48 + tid];
        }
        double c = 1.0;
        if (activate) {
            const double n_same = (double)(np > 1 ? np : 1), n_opp = (double)(nn > 1 ? nn : 1);
            c = n_opp / (n_same > n_opp ? n_same : n_opp);
        }
        s_c[tid] = (float)c;
        s_loss[tid] = c * sp + sn;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        double t = 0.0;
        for (int k = 0; k < KA; ++k) t += s_loss[k];
        loss[0] = (float)(-t / norm);
    }
    if (dz != nullptr) {
        const float f = (float)(-2.0 / norm) * scale;
        const long n = (long)nrows * ldz;
        const size_t blk = (size_t)nrows * KA;
        for (long idx = (long)blockIdx.x * 256 + tid; idx < n; idx += (long)gridDim.x * 256) {
            const int i = (int)(idx / ldz), k = (int)(idx - (long)i * ldz);
            float v = 0.f;
            if (k < KA) {
                float gp = 0.f, gn = 0.f;
                for (int b = 0; b < nj; ++b) { gp += gpos[b * blk + (size_t)i * KA + k]; gn += gneg[b * blk + (size_t)i * KA + k]; }
                v = f * (s_c[k] * gp + gn);
            }
            dz[idx] = v;
        }
    }
}

__global__ __launch_bounds__(256) void rank_finalize_kernel(RankFinArgs a) {
    rank_finalize_body(a.part, a.nparts, a.totals, a.n_all, a.nrows, a.nj, a.KA, a.activate, a.scale, a.gpos, a.gneg, a.loss, a.dz, a.ldz);
}
__global__ __launch_bounds__(256) void rank_finalize_kernel_m(const RankFinArgs* t) {
    const RankFinArgs a = t[blockIdx.z];
    rank_finalize_body(a.part, a.nparts, a.totals, a.n_all, a.nrows, a.nj, a.KA, a.activate, a.scale, a.gpos, a.gneg, a.loss, a.dz, a.ldz);
}

// per-descriptor totals {n+, n-, S+, S-} of this rank's pairs as [4][16] doubles (fixed order): what a data-parallel
// run all-reduces between the pair pass and the finalize
__global__ __launch_bounds__(256) void rank_totals_kernel(const RankWork* part, int nparts, int KA, double* totals) {
    __shared__ double r_s[2][256];
    __shared__ long long r_n[2][256];
    const int tid = threadIdx.x, k = tid & 15, sl = tid >> 4;
    long long np = 0, nn = 0; double sp = 0.0, sn = 0.0;
    if (k < KA) {
#pragma unroll 8
        for (int p = sl; p < nparts; p += 16) {
            np += part[p].n_pos[k]; nn += part[p].n_neg[k];
            sp += part[p].s_pos[k]; sn += part[p].s_neg[k];
        }
    }
    r_s[0][tid] = sp; r_s[1][tid] = sn; r_n[0][tid] = np; r_n[1][tid] = nn;
    __syncthreads();
    if (tid < 16) {
        np = 0; nn = 0; sp = 0.0; sn = 0.0;
        if (tid < KA)
            for (int s2 = 0; s2 < 16; ++s2) {
                sp += r_s[0][s2 * 16 + tid]; sn += r_s[1][s2 * 16 + tid];
                np += r_n[0][s2 * 16 + tid]; nn += r_n[1][s2 * 16 + tid];
            }
        totals[tid] = (double)np; totals[16 + tid] = (double)nn; totals[32 + tid] = sp; totals[48 + tid] = sn;
    }
}

// ------------------------------------------------------------------ reconstruction loss
// functions.py:81-107.  One wave per row; L <= 1024.
// Optional finalisation inside the loss kernel: after thread 0 has stored this workgroup's partial, the LAST workgroup
// to arrive (ticket) adds all partials in index order and writes the loss -- what raae_loss_finalize does in a launch
// of its own.  All threads call it; `f.ticket == NULL` leaves the partials to the caller.
struct LossFin { float scale; float* out; int slot; int acc_slot; unsigned* ticket; };
__device__ __forceinline__ void loss_fin_last_block(const LossFin& f, double* partial) {
    __shared__ unsigned s_fin_last;
    __shared__ double s_fin_red[16];
    if (f.ticket == nullptr) return;
    if (threadIdx.x == 0) {
        __threadfence();
        s_fin_last = atomicAdd(f.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_fin_last) return;                 // uniform per workgroup
    // the whole last workgroup adds the partials: thread t takes partials t, t + blockDim, ... and the per-thread sums
    // meet in a fixed-order tree (one thread walking 512 partials was 512 dependent L2 round trips: 80 us of the
    // 95-us smoothness kernel at 4096 rows, 10 of 20 us at 256 rows)
    __threadfence();
    double t = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) t += ((volatile double*)partial)[i];
    t = raae::block_sum(t, s_fin_red);
    if (threadIdx.x == 0) {
        const float v = (float)(t * (double)f.scale);
        f.out[f.slot] = v;
        if (f.acc_slot >= 0) f.out[f.acc_slot] += v;
        *f.ticket = 0u;
    }
}
static LossFin make_fin(const raae_loss_fin_t* fin) {
    LossFin f = {1.f, nullptr, 0, -1, nullptr};
    if (fin && fin->ticket) { f.scale = fin->scale; f.out = fin->out; f.slot = fin->slot; f.acc_slot = fin->acc_slot; f.ticket = fin->ticket; }
    return f;
}

struct ReconArgs { const float* x; const float* y; int B; int L; int scale; double* partial; float* dy; LossFin fin; };
__device__ __forceinline__ void recon_body(const float* x, const float* y, int B, int L, int scale,
                                           double* partial, float* dy, const LossFin& fin) {
    __shared__ double shd[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 0.0;
    for (int row = blockIdx.x * 4 + wv; row < B; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * L;
        const float* yr = y + (size_t)row * L;
        float c = 1.f, gscale = 0.f;
        if (scale) {
            float sx = 0.f, sy = 0.f;
            for (int l = lane; l < L; l += 64) { sx += xr[l]; sy += yr[l]; }
            sx = raae::wave_sum(sx); sy = raae::wave_sum(sy);
            const float mx = sx / (float)L, my = sy / (float)L;
            const float r = fabsf(my) / fabsf(mx);
            c = fminf(fmaxf(r, 0.7f), 1.3f);
            if (lane == 0) acc += 0.1 * (double)(r - 1.f) * (double)(r - 1.f) / (double)B;
            const float sg = my > 0.f ? 1.f : (my < 0.f ? -1.f : 0.f);
            gscale = 0.2f * (r - 1.f) * sg / (fabsf(mx) * (float)L * (float)B);
        }
        const float inv = 1.f / ((float)B * (float)L);
        float se = 0.f;
        for (int l = lane; l < L; l += 64) {
            const float e = yr[l] - xr[l] * c;
            se += e * e;
            if (dy) dy[(size_t)row * L + l] = 2.f * e * inv + gscale;
        }
        se = raae::wave_sum(se);
        if (lane == 0) acc += (double)se / ((double)B * (double)L);
    }
    const double t = raae::block_sum(acc, shd);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
    loss_fin_last_block(fin, partial);
}

__global__ __launch_bounds__(256) void recon_kernel(ReconArgs a) { recon_body(a.x, a.y, a.B, a.L, a.scale, a.partial, a.dy, a.fin); }
__global__ __launch_bounds__(256) void recon_kernel_m(const ReconArgs* t) {
    const ReconArgs a = t[blockIdx.z];
    recon_body(a.x, a.y, a.B, a.L, a.scale, a.partial, a.dy, a.fin);
}

// ------------------------------------------------------------------ smoothness loss
// functions.py:194-212 + GaussianSmoothing (model.py:177-229).  One wave per row, row in LDS.
// loss = mean((x - Gx)^2); dL/dx = (2/N) (e - G^T e), e = x - Gx, G = replicate-pad Gaussian.
#define SM_MAXT 33
struct Taps { float w[SM_MAXT]; int n; };

// NT > 0: the tap count at compile time (17 for the reference's smoothing): the tap loops unroll and the weights
// stay in scalar registers -- with a run-time count every tap was a scalar load from the kernarg segment with its
// own wait (15.7 us for 256 rows; NT = 0 keeps that generic form).
struct SmoothArgs { const float* x; int B; int L; Taps tp; double* partial; float* dx; LossFin fin; };
template <int NT>
__device__ __forceinline__ void smooth_body(const float* x, int B, int L, const Taps& tp, double* partial, float* dx, const LossFin& fin) {
    const int ntap = NT > 0 ? NT : tp.n;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2][L]
    __shared__ double shd[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* xs = smem + (size_t)wv * 2 * L;
    float* es = xs + L;
    const int half = (ntap - 1) / 2;
    double acc = 0.0;
    const float inv = 1.f / ((float)B * (float)L);
    for (int row = blockIdx.x * 4 + wv; row < B; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * L;
        for (int l = lane; l < L; l += 64) xs[l] = xr[l];
        __builtin_amdgcn_wave_barrier();
        float se = 0.f;
        for (int l = lane; l < L; l += 64) {
            float g = 0.f;
#pragma unroll
            for (int t = 0; t < ntap; ++t) {
                int j = l + t - half;
                j = j < 0 ? 0 : (j > L - 1 ? L - 1 : j);
                g += tp.w[t] * xs[j];
            }
            const float e = xs[l] - g;
            es[l] = e;
            se += e * e;
        }
        se = raae::wave_sum(se);
        if (lane == 0) acc += (double)se / ((double)B * (double)L);
        __builtin_amdgcn_wave_barrier();
        if (dx) {
            // (G^T e)_j = sum over padded positions q = l + t - half that clamp to j
            for (int j = lane; j < L; j += 64) {
                float gt = 0.f;
                // interior contribution: l = j - t + half, any l in [0, L)
#pragma unroll
                for (int t = 0; t < ntap; ++t) {
                    const int l = j - t + half;
                    if (l >= 0 && l < L) gt += tp.w[t] * es[l];
                }
                if (j == 0) {            // padded positions q < 0 clamp to 0: l + t - half < 0
                    for (int l = 0; l < half && l < L; ++l)
                        for (int t = 0; t < half - l; ++t) gt += tp.w[t] * es[l];
                }
                if (j == L - 1) {        // q > L-1: l + t - half > L-1
                    for (int l = max(0, L - half); l < L; ++l)
                        for (int t = L - l + half; t < ntap; ++t) gt += tp.w[t] * es[l];
                }
                dx[(size_t)row * L + j] = 2.f * inv * (es[j] - gt);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const double t = raae::block_sum(acc, shd);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
    loss_fin_last_block(fin, partial);
}

template <int NT>
__global__ __launch_bounds__(256) void smooth_kernel(SmoothArgs a) { smooth_body<NT>(a.x, a.B, a.L, a.tp, a.partial, a.dx, a.fin); }
template <int NT>
__global__ __launch_bounds__(256) void smooth_kernel_m(const SmoothArgs* t) {
    __shared__ SmoothArgs slot;      // the taps are indexed at run time: a private copy would live in scratch memory
    const SmoothArgs& a = raae::args_from_table(&slot, t);
    smooth_body<NT>(a.x, a.B, a.L, a.tp, a.partial, a.dx, a.fin);
}

// ------------------------------------------------------------------ MSE (mutual-info loss)
struct MseArgs { const float* a; const float* b; long n; double* partial; float* da; LossFin fin; };
__device__ __forceinline__ void mse_body(const float* a, const float* b, long n, double* partial, float* da, const LossFin& fin) {
    __shared__ double shd[16];
    double acc = 0.0;
    const float inv = 1.f / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float e = a[i] - b[i];
        acc += (double)e * (double)e;
        if (da) da[i] = 2.f * e * inv;
    }
    const double t = raae::block_sum(acc, shd);
    if (threadIdx.x == 0) partial[blockIdx.x] = t / (double)n;
    loss_fin_last_block(fin, partial);
}

__global__ __launch_bounds__(256) void mse_kernel(MseArgs a) { mse_body(a.a, a.b, a.n, a.partial, a.da, a.fin); }
__global__ __launch_bounds__(256) void mse_kernel_m(const MseArgs* t) {
    const MseArgs a = t[blockIdx.z];
    mse_body(a.a, a.b, a.n, a.partial, a.da, a.fin);
}

// ------------------------------------------------------------------ BCE-with-logits pair
// functions.py:119-130: mean_i softplus(-o_i) over real + mean_i softplus(o_i) over fake.
__device__ __forceinline__ double softplus_d(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }

struct BceArgs { const float* o; int n_real; int n_fake; float* loss; float* d; };
__device__ __forceinline__ void bce_pair_body(const float* o, int n_real, int n_fake, float* loss, float* d) {
    __shared__ double shd[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_real + n_fake; i += 1024) {
        const float v = o[i];
        const float sg = 1.f / (1.f + expf(-v));
        if (i < n_real) { acc += softplus_d(-(double)v) / (double)n_real; if (d) d[i] = (sg - 1.f) / (float)n_real; }
        else { acc += softplus_d((double)v) / (double)n_fake; if (d) d[i] = sg / (float)n_fake; }
    }
    const double t = raae::block_sum(acc, shd);
    if (threadIdx.x == 0) loss[0] = (float)t;
}

__global__ __launch_bounds__(1024) void bce_pair_kernel(BceArgs a) { bce_pair_body(a.o, a.n_real, a.n_fake, a.loss, a.d); }
__global__ __launch_bounds__(1024) void bce_pair_kernel_m(const BceArgs* t) {
    const BceArgs a = t[blockIdx.z];
    bce_pair_body(a.o, a.n_real, a.n_fake, a.loss, a.d);
}

struct DiscInArgs { const float* z_real; const float* styles; const float* noise; float sigma; int n_real; int n_fake; int C; float* out; };
__device__ __forceinline__ void disc_input_body(const float* z_real, const float* styles, const float* noise, float sigma,
                                                int n_real, int n_fake, int C, float* out) {
    const long n = (long)(n_real + n_fake) * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long split = (long)n_real * C;
        float v = i < split ? z_real[i] : styles[i - split];
        if (noise) v += sigma * noise[i];
        out[i] = v;
    }
}

__global__ void disc_input_kernel(DiscInArgs a) { disc_input_body(a.z_real, a.styles, a.noise, a.sigma, a.n_real, a.n_fake, a.C, a.out); }
__global__ void disc_input_kernel_m(const DiscInArgs* t) {
    const DiscInArgs a = t[blockIdx.z];
    disc_input_body(a.z_real, a.styles, a.noise, a.sigma, a.n_real, a.n_fake, a.C, a.out);
}

__global__ void scale_by_dev_kernel(const float* src, const float* dev_scale, float sign, long n, float* dst) {
    const float s = sign * dev_scale[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = s * src[i];
}

// the same fixed-order tree as the in-kernel finish (loss_fin_last_block): 256 threads, thread t adds partials
// t, t + 256, ..., then raae::block_sum -- the two ways of finishing a loss give the same bits
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* partial, int n, float scale, float* out, int slot, int acc_slot) {
    __shared__ double red[16];
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) t += partial[i];
    t = raae::block_sum(t, red);
    if (threadIdx.x == 0) {
        const float v = (float)(t * (double)scale);
        out[slot] = v;
        if (acc_slot >= 0) out[acc_slot] += v;
    }
}

__global__ void gather_batch_kernel(const float* spec, const float* aux, const long* idx_all, const int* cursor,
                                    const float* noise, float spec_noise, int B, int L, int n_aux, float* spec_out,
                                    float* aux_out) {
    // the step tick has already advanced the cursor past this batch: rows [cursor-B, cursor)
    const long* idx = idx_all + (cursor ? (cursor[0] - B) : 0);
    const long n = (long)B * L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / L), l = (int)(i - (long)b * L);
        float v = spec[(size_t)idx[b] * L + l];
        if (noise) v += noise[i] * spec_noise;
        spec_out[i] = v;
    }
    const long na = (long)B * n_aux;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na; i += (long)gridDim.x * 256) {
        const int b = (int)(i / n_aux), k = (int)(i - (long)b * n_aux);
        aux_out[i] = aux[(size_t)idx[b] * n_aux + k];
    }
}

// ------------------------------------------------------------------ collapse of partial statistics (experiment)
// one workgroup per statistic adds the <= 512 partial rows {sum, sum of squares} once, so that
// every consumer's prologue reads ONE row.  Fixed order (reduce_partials) => deterministic.
__global__ __launch_bounds__(256) void stat_collapse_kernel(const double* p1, int n1, int C1, double* o1,
                                                            const double* p2, int n2, int C2, double* o2) {
    __shared__ double tot_s[256], tot_q[256];
    const double* p = blockIdx.x == 0 ? p1 : p2;
    const int n = blockIdx.x == 0 ? n1 : n2, C = blockIdx.x == 0 ? C1 : C2;
    double* o = blockIdx.x == 0 ? o1 : o2;
    raae::reduce_partials(p, n, C, tot_s, tot_q);
    if ((int)threadIdx.x < C) { o[2 * threadIdx.x] = tot_s[threadIdx.x]; o[2 * threadIdx.x + 1] = tot_q[threadIdx.x]; }
}

int grid_for(long n, int per_block, int cap) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" int raae_style_bn_fwd(const float* z, int B, int C, const raae_bn_t* bn, float* styles, void* stream) {
    RAAE_CHECK_ARG(z && bn && styles && B > 0 && C > 0 && C <= 64 && bn->nparts <= RAAE_MAX_PARTS);
    const StyleFwdArgs a = {z, B, C, *bn, styles};
    raae::launch(style_bn_fwd_kernel, style_bn_fwd_kernel_m, dim3(grid_for((long)B * C, 256, 256)), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_style_bn_bwd(const float* dstyles, const float* styles, int B, int C, const raae_bn_t* bn,
                                 float scale, float* dz, void* stream) {
    RAAE_CHECK_ARG(dstyles && styles && bn && dz && B > 0 && C > 0 && C <= 64 && bn->nparts <= RAAE_MAX_PARTS);
    const StyleBwdArgs a = {dstyles, styles, B, C, *bn, scale, dz};
    raae::launch(style_bn_bwd_kernel, style_bn_bwd_kernel_m, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_stat_collapse2(const double* p1, int n1, int C1, double* o1, const double* p2, int n2, int C2, double* o2,
                                   void* stream) {
    RAAE_CHECK_ARG(p1 && o1 && n1 > 0 && n1 <= RAAE_MAX_PARTS && C1 > 0 && C1 <= 256);
    RAAE_CHECK_ARG(!p2 || (o2 && n2 > 0 && n2 <= RAAE_MAX_PARTS && C2 > 0 && C2 <= 256));
    RAAE_PLAIN_LAUNCH(stat_collapse_kernel, dim3(p2 ? 2 : 1), dim3(256), 0, (hipStream_t)stream, p1, n1, C1, o1, p2, n2, C2, o2);
    RAAE_LAUNCH_RET();
}

#define RANK_MAXWG 2048       // workgroups (= RankWork partials) of the pair pass
#define RANK_MAXNJ 16         // column blocks
struct RankGrid { int R, nwg, nj, jchunk; };
static RankGrid rank_grid(int n_all, int nrows, int n_aux) {
    RankGrid g;
    const int ipb = 32 / n_aux;
    g.R = nrows > 1024 ? 4 : 1;
    const long groups = (nrows + (long)ipb * g.R - 1) / ((long)ipb * g.R);
    // column blocks only where the pair pass is long enough to need them (>= 1024 workgroups for 256 CUs); the
    // launch-bound small batches keep one block, i.e. one pass over the columns per row group
    g.nj = 1;
    if (n_all > 1024 && groups < 1024) {
        g.nj = (int)((1024 + groups - 1) / groups);
        if (g.nj > RANK_MAXNJ) g.nj = RANK_MAXNJ;
        if (g.nj > n_all / RANK_TJ) g.nj = n_all / RANK_TJ;
        if (g.nj < 1) g.nj = 1;
    }
    g.jchunk = ((n_all + g.nj - 1) / g.nj + RANK_TJ - 1) / RANK_TJ * RANK_TJ;
    long ni = groups < RANK_MAXWG / g.nj ? groups : RANK_MAXWG / g.nj;
    g.nwg = (int)ni * g.nj;
    return g;
}
static size_t rank_part_bytes() { return ((size_t)RANK_MAXWG * sizeof(RankWork) + 255) & ~(size_t)255; }

extern "C" long raae_rank_loss_work_bytes(int B, int n_aux) {
    return (long)rank_part_bytes() + 2L * RANK_MAXNJ * B * n_aux * (long)sizeof(float) + 64 * (long)sizeof(double) + 256;
}

static int rank_pairs_launch(const float* d, int ldd, const float* z, int ldz, int n_all, int row0, int nrows, int n_aux,
                             void* work, RankGrid& g, float*& gpos, float*& gneg, hipStream_t st) {
    g = rank_grid(n_all, nrows, n_aux);
    RankWork* part = (RankWork*)work;
    gpos = (float*)((char*)work + rank_part_bytes());
    gneg = gpos + (size_t)RANK_MAXNJ * nrows * n_aux;
    const RankPairsArgs pa = {d, ldd, z, ldz, n_all, row0, nrows, g.nj, g.jchunk, part, gpos, gneg};
#define RANK_CASE(KA) case KA: \
        if (g.R == 1) raae::launch(rank_pairs_kernel<KA, 1>, rank_pairs_kernel_m<KA, 1>, dim3(g.nwg), dim3(256), 0, st, pa); \
        else raae::launch(rank_pairs_kernel<KA, 4>, rank_pairs_kernel_m<KA, 4>, dim3(g.nwg), dim3(256), 0, st, pa); \
        break;
    switch (n_aux) {
        RANK_CASE(1) RANK_CASE(2) RANK_CASE(3) RANK_CASE(4) RANK_CASE(5) RANK_CASE(6) RANK_CASE(7) RANK_CASE(8)
        RANK_CASE(9) RANK_CASE(10) RANK_CASE(11) RANK_CASE(12) RANK_CASE(13) RANK_CASE(14) RANK_CASE(15) RANK_CASE(16)
        default: return RAAE_EINVAL;
    }
#undef RANK_CASE
    return (int)hipGetLastError();
}

extern "C" int raae_rank_loss_fwd_bwd(const float* d, int ldd, const float* z, int ldz, int B, int n_aux, int activate,
                                      void* work, float* loss, float* dz, void* stream) {
    RAAE_CHECK_ARG(d && z && work && loss && B > 1 && n_aux >= 1 && n_aux <= RANK_MAXK && ldd >= n_aux && ldz >= n_aux);
    hipStream_t st = (hipStream_t)stream;
    RankGrid g;
    float *gpos, *gneg;
    const int rc = rank_pairs_launch(d, ldd, z, ldz, B, 0, B, n_aux, work, g, gpos, gneg, st);
    if (rc) return rc;
    const int gf = dz ? grid_for((long)B * ldz, 256, 256) : 1;
    const RankFinArgs fa = {(const RankWork*)work, g.nwg, (const double*)nullptr, B, B, g.nj, n_aux, activate, 1.f, gpos, gneg, loss, dz, ldz};
    raae::launch(rank_finalize_kernel, rank_finalize_kernel_m, dim3(gf), dim3(256), 0, st, fa);
    RAAE_LAUNCH_RET();
}

// One rank's share of the GLOBAL pairs of a data-parallel batch (rows [row0, row0 + nrows) against all n_all rows):
// pair pass + this rank's totals; after the caller has summed `totals` over the ranks, raae_rank_rows_finish forms
// the global loss and the gradient of this rank's rows.
extern "C" int raae_rank_rows_pairs(const float* d_all, int ldd, const float* z_all, int ldz, int n_all, int row0, int nrows,
                                    int n_aux, void* work, double* totals, void* stream) {
    RAAE_CHECK_ARG(d_all && z_all && work && totals && n_all > 1 && nrows >= 1 && row0 >= 0 && row0 + nrows <= n_all &&
                   n_aux >= 1 && n_aux <= RANK_MAXK && ldd >= n_aux && ldz >= n_aux);
    hipStream_t st = (hipStream_t)stream;
    RankGrid g;
    float *gpos, *gneg;
    const int rc = rank_pairs_launch(d_all, ldd, z_all, ldz, n_all, row0, nrows, n_aux, work, g, gpos, gneg, st);
    if (rc) return rc;
    RAAE_PLAIN_LAUNCH(rank_totals_kernel, dim3(1), dim3(256), 0, st, (const RankWork*)work, g.nwg, n_aux, totals);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_rank_rows_finish(const double* totals, int n_all, int nrows, int n_aux, int activate, float scale,
                                     void* work, float* loss, float* dz, int ldz, void* stream) {
    RAAE_CHECK_ARG(totals && work && loss && n_all > 1 && nrows >= 1 && nrows <= n_all && n_aux >= 1 && n_aux <= RANK_MAXK &&
                   (!dz || ldz >= n_aux));
    const RankGrid g = rank_grid(n_all, nrows, n_aux);
    float* gpos = (float*)((char*)work + rank_part_bytes());
    float* gneg = gpos + (size_t)RANK_MAXNJ * nrows * n_aux;
    const int gf = dz ? grid_for((long)nrows * ldz, 256, 256) : 1;
    const RankFinArgs fa = {(const RankWork*)nullptr, 0, totals, n_all, nrows, g.nj, n_aux, activate, scale, gpos, gneg, loss, dz, ldz};
    raae::launch(rank_finalize_kernel, rank_finalize_kernel_m, dim3(gf), dim3(256), 0, (hipStream_t)stream, fa);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_recon_loss_fwd_bwd(const float* spec_in, const float* spec_out, int B, int L, int scale,
                                       double* partial, int* nparts, float* dout, const raae_loss_fin_t* fin, void* stream) {
    RAAE_CHECK_ARG(spec_in && spec_out && partial && B > 0 && L > 0 && (!fin || !fin->ticket || (fin->out && fin->slot >= 0)));
    const int g = grid_for(B, 4, RAAE_MAX_PARTS);
    if (nparts) *nparts = g;
    const ReconArgs a = {spec_in, spec_out, B, L, scale, partial, dout, make_fin(fin)};
    raae::launch(recon_kernel, recon_kernel_m, dim3(g), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_smooth_loss_fwd_bwd(const float* x, int B, int L, const float* taps, int ntaps,
                                        double* partial, int* nparts, float* dx, const raae_loss_fin_t* fin, void* stream) {
    RAAE_CHECK_ARG((!fin || !fin->ticket || (fin->out && fin->slot >= 0)) && x && taps && partial && B > 0 && L > 1 && ntaps >= 1 && ntaps <= SM_MAXT && (ntaps & 1) && L <= 4096);
    Taps tp; tp.n = ntaps;
    for (int i = 0; i < ntaps; ++i) tp.w[i] = taps[i];   // `taps` is a HOST pointer (17 floats)
    const int g = grid_for(B, 4, RAAE_MAX_PARTS);
    if (nparts) *nparts = g;
    const SmoothArgs a = {x, B, L, tp, partial, dx, make_fin(fin)};
    if (ntaps == 17)
        raae::launch(smooth_kernel<17>, smooth_kernel_m<17>, dim3(g), dim3(256), sizeof(float) * 8 * (size_t)L, (hipStream_t)stream, a);
    else
        raae::launch(smooth_kernel<0>, smooth_kernel_m<0>, dim3(g), dim3(256), sizeof(float) * 8 * (size_t)L, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_mse_fwd_bwd(const float* a, const float* b, long n, double* partial, int* nparts, float* da,
                                const raae_loss_fin_t* fin, void* stream) {
    RAAE_CHECK_ARG(a && b && partial && n > 0 && (!fin || !fin->ticket || (fin->out && fin->slot >= 0)));
    const int g = grid_for(n, 1024, RAAE_MAX_PARTS);
    if (nparts) *nparts = g;
    const MseArgs ma = {a, b, n, partial, da, make_fin(fin)};
    raae::launch(mse_kernel, mse_kernel_m, dim3(g), dim3(256), 0, (hipStream_t)stream, ma);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_bce_pair_fwd_bwd(const float* logits, int n_real, int n_fake, float* loss, float* dlogits, void* stream) {
    RAAE_CHECK_ARG(logits && loss && n_real > 0 && n_fake > 0);
    const BceArgs a = {logits, n_real, n_fake, loss, dlogits};
    raae::launch(bce_pair_kernel, bce_pair_kernel_m, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_disc_input(const float* z_real, const float* styles, const float* noise, float sigma,
                               int n_real, int n_fake, int C, float* out, void* stream) {
    RAAE_CHECK_ARG(z_real && styles && out && n_real > 0 && n_fake > 0 && C > 0);
    const long n = (long)(n_real + n_fake) * C;
    const DiscInArgs a = {z_real, styles, noise, sigma, n_real, n_fake, C, out};
    raae::launch(disc_input_kernel, disc_input_kernel_m, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_scale_by_dev(const float* src, const float* dev_scale, float sign, long n, float* dst, void* stream) {
    RAAE_CHECK_ARG(src && dev_scale && dst && n > 0);
    RAAE_PLAIN_LAUNCH(scale_by_dev_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, src, dev_scale, sign, n, dst);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_loss_finalize(const double* partial, int n, float scale, float* out, int slot, int acc_slot, void* stream) {
    RAAE_CHECK_ARG(partial && out && n > 0 && slot >= 0);
    RAAE_PLAIN_LAUNCH(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, scale, out, slot, acc_slot);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_gather_batch(const float* spec, const float* aux, const long* idx, const int* cursor,
                                 const float* noise, float spec_noise, int B, int L, int n_aux, float* spec_out,
                                 float* aux_out, void* stream) {
    RAAE_CHECK_ARG(spec && aux && idx && spec_out && aux_out && B > 0 && L > 0 && n_aux > 0);
    RAAE_PLAIN_LAUNCH(gather_batch_kernel, dim3(grid_for((long)B * L, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       spec, aux, idx, cursor, noise, spec_noise, B, L, n_aux, spec_out, aux_out);
    RAAE_LAUNCH_RET();
}
