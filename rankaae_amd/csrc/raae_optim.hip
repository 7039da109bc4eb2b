// Fused multi-tensor Adam/AdamW over the flat parameter arena, the per-step tick,
// the Philox random tape, and HIP stream/graph/event plumbing.
#include "raae_common.h"
#include <string.h>
#include <stdlib.h>

namespace {

// torch.optim.Adam / AdamW single-tensor update (torch/optim/adam.py::_single_tensor_adam),
// operation order mirrored in fp32; scalars formed in double like Python floats.
// Gradient of element i = fixed-order sum of seg_nslab[i/64] slabs; 0 slabs => parameter is
// skipped (the reference skips params whose .grad is None, trainer.py:318-321).
struct AdamArgs { float* p; float* m; float* v; const float* g_slabs; long slab_stride; const unsigned short* seg_nslab;
                  long n; const double* hyper; const int* step; int decoupled; };
__device__ __forceinline__ void adam_body(float* p, float* m, float* v, const float* g_slabs, long slab_stride,
                                          const unsigned short* seg_nslab, long n, const double* hyper,
                                          const int* step, int decoupled) {
    __shared__ float s_sc[8];
    if (threadIdx.x == 0) {
        const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
        const double t = (double)step[0];
        const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
        s_sc[0] = (float)(1.0 - lr * wd);      // AdamW decay factor
        s_sc[1] = (float)(1.0 - b1);           // lerp weight
        s_sc[2] = (float)b2;
        s_sc[3] = (float)(1.0 - b2);
        s_sc[4] = (float)(-(lr / bc1));        // -step_size
        s_sc[5] = (float)sqrt(bc2);
        s_sc[6] = (float)eps;
        s_sc[7] = (float)wd;
    }
    __syncthreads();
    const float decay = s_sc[0], w1 = s_sc[1], b2f = s_sc[2], omb2 = s_sc[3], nstep = s_sc[4], bc2s = s_sc[5],
                epsf = s_sc[6], wdf = s_sc[7];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ns = seg_nslab[i >> 6];
        if (ns == 0) continue;
        // fixed-order slab sum, 8 independent loads in flight at a time
        float g = 0.f;
        int s = 0;
        for (; s + 8 <= ns; s += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = g_slabs[(size_t)(s + u) * slab_stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += t[u];
        }
        for (; s < ns; ++s) g += g_slabs[(size_t)s * slab_stride + i];
        float pv = p[i];
        if (decoupled) pv = pv * decay; else if (wdf != 0.f) g = g + wdf * pv;
        float mv = m[i], vv = v[i];
        mv = mv + w1 * (g - mv);
        vv = vv * b2f;
        vv = vv + (omb2 * g) * g;
        const float denom = sqrtf(vv) / bc2s + epsf;
        pv = pv + (nstep * mv) / denom;
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    adam_body(a.p, a.m, a.v, a.g_slabs, a.slab_stride, a.seg_nslab, a.n, a.hyper, a.step, a.decoupled);
}
__global__ __launch_bounds__(256) void adam_kernel_m(const AdamArgs* t) {
    const AdamArgs a = t[blockIdx.z];
    adam_body(a.p, a.m, a.v, a.g_slabs, a.slab_stride, a.seg_nslab, a.n, a.hyper, a.step, a.decoupled);
}

// Same update with the slabs of an element spread over 8 lanes: for ranges whose tensors have many slabs.
__device__ __forceinline__ void adam_wide_body(float* p, float* m, float* v, const float* g_slabs, long slab_stride,
                                               const unsigned short* seg_nslab, long n, const double* hyper,
                                               const int* step, int decoupled) {
    __shared__ float s_sc[8];
    if (threadIdx.x == 0) {
        const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
        const double t = (double)step[0];
        const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
        s_sc[0] = (float)(1.0 - lr * wd);      // AdamW decay factor
        s_sc[1] = (float)(1.0 - b1);           // lerp weight
        s_sc[2] = (float)b2;
        s_sc[3] = (float)(1.0 - b2);
        s_sc[4] = (float)(-(lr / bc1));        // -step_size
        s_sc[5] = (float)sqrt(bc2);
        s_sc[6] = (float)eps;
        s_sc[7] = (float)wd;
    }
    __syncthreads();
    const float decay = s_sc[0], w1 = s_sc[1], b2f = s_sc[2], omb2 = s_sc[3], nstep = s_sc[4], bc2s = s_sc[5],
                epsf = s_sc[6], wdf = s_sc[7];
    // A wave owns 8 consecutive elements; lane = chunk*8 + element: the slabs of an element are spread over
    // 8 lanes (chunk c sums slabs c, c+8, ... eight loads deep), joined by a fixed xor-shuffle tree.  With one
    // thread per element the 256 slabs of a conv weight were 32 dependent round trips (28 us per step phase).
    const int lane = threadIdx.x & 63, el = lane & 7, ch = lane >> 3;
    const long wave0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    for (long base = wave0; base < n; base += (long)gridDim.x * 32) {
        const long i = base + el;                       // n is a multiple of 64: i < n whenever base < n
        const int ns = seg_nslab[i >> 6];
        if (ns == 0) continue;                          // uniform over the wave (8 elements share a segment)
        float g = 0.f;
        for (int s = ch; s < ns; s += 64) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = s + 8 * u;
                t[u] = g_slabs[(size_t)(r < ns ? r : ch) * slab_stride + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) g += (s + 8 * u < ns) ? t[u] : 0.f;
        }
        g += __shfl_xor(g, 8, 64);
        g += __shfl_xor(g, 16, 64);
        g += __shfl_xor(g, 32, 64);
        if (ch != 0) continue;
        float pv = p[i];
        if (decoupled) pv = pv * decay; else if (wdf != 0.f) g = g + wdf * pv;
        float mv = m[i], vv = v[i];
        mv = mv + w1 * (g - mv);
        vv = vv * b2f;
        vv = vv + (omb2 * g) * g;
        const float denom = sqrtf(vv) / bc2s + epsf;
        pv = pv + (nstep * mv) / denom;
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

__global__ __launch_bounds__(256) void adam_wide_kernel(AdamArgs a) {
    adam_wide_body(a.p, a.m, a.v, a.g_slabs, a.slab_stride, a.seg_nslab, a.n, a.hyper, a.step, a.decoupled);
}
__global__ __launch_bounds__(256) void adam_wide_kernel_m(const AdamArgs* t) {
    const AdamArgs a = t[blockIdx.z];
    adam_wide_body(a.p, a.m, a.v, a.g_slabs, a.slab_stride, a.seg_nslab, a.n, a.hyper, a.step, a.decoupled);
}

__global__ void tick_kernel(int* steps, int n, unsigned mask, unsigned long long* rng_counter, int* cursor,
                            int cursor_inc) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < n; ++i) if (mask & (1u << i)) steps[i] += 1;
        if (rng_counter) {
            rng_counter[0] += 1ull;
            raae::mask_keys_store(rng_counter, rng_counter[1], rng_counter[0]);       // {counter, seed, keys}
        }
        if (cursor) cursor[0] += cursor_inc;
    }
}

// ---- Philox4x32-10 (Salmon et al. 2011) ----
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c[0], c[1], c[2], c[3], k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four N(0, 1) values: Philox block `qg` (= position in the numbering of the step's Gaussian elements / 4) of step `ctr`
__device__ __forceinline__ void normal4(long qg, unsigned long long ctr, unsigned long long seed, float (&o)[4]) {
    uint32_t c[4] = {(uint32_t)qg, (uint32_t)(qg >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32)};
    philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float r0 = sqrtf(-2.f * logf(u01(c[0]))), r1 = sqrtf(-2.f * logf(u01(c[2])));
    const float a0 = 6.283185307179586f * u01(c[1]), a1 = 6.283185307179586f * u01(c[3]);
    o[0] = r0 * cosf(a0); o[1] = r0 * sinf(a0); o[2] = r1 * cosf(a1); o[3] = r1 * sinf(a1);
}

// floats [4 q, 4 q + 4) of the tape (segments start at multiples of 4 floats)
__device__ __forceinline__ void fill_quad(float* tape, const int* seg_desc, const float* seg_scale, int nseg, long total,
                                          unsigned long long seed, unsigned long long ctr, long q) {
    const long e0 = q * 4;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {                       // last segment whose offset <= e0
        const int mid = (lo + hi + 1) >> 1;
        if ((long)seg_desc[mid * 4] <= e0) lo = mid; else hi = mid - 1;
    }
    const int kind = seg_desc[lo * 4 + 2];
    const long send = (long)seg_desc[lo * 4] + seg_desc[lo * 4 + 1];
    float o[4];
    if (kind == 0) {
        // Philox counter = (position of these four floats in the numbering of the step's Gaussian elements) / 4
        normal4(((long)seg_desc[lo * 4 + 3] + (e0 - seg_desc[lo * 4])) >> 2, ctr, seed, o);
    } else if (kind == 1) {
        // dropout multipliers {0, 1/keep}: the counter-based hash of raae_common.h -- the function a kernel with a
        // raae_maskgen_t evaluates for a slot it generates itself (hash index = the slot's position in the numbering
        // of ALL dropout elements of the step, seg_desc[.][3], + the element's index in the slot: the same whether the
        // slot lives on the tape or in its consumer)
        const raae::MaskGen g = raae::mask_gen_make(seed, ctr, (uint32_t)seg_desc[lo * 4 + 3], seg_scale[lo]);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = raae::mask_val(g, (uint32_t)(e0 + j - seg_desc[lo * 4]));
    } else {
        // kind 2 (`precision: bf16`): keep flags {0, 1} stored as bf16 -- these four floats hold eight of them (bf16
        // element i of the slot is hashed at index i of the slot); the dense kernels multiply by the fp32 1/keep
        const raae::MaskGen g = raae::mask_gen_make(seed, ctr, (uint32_t)seg_desc[lo * 4 + 3], seg_scale[lo]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t e = 2u * (uint32_t)(e0 + j - seg_desc[lo * 4]);
            const uint32_t lo16 = raae::mask_keep(g, e) ? 0x3F80u : 0u, hi16 = raae::mask_keep(g, e + 1u) ? 0x3F80u : 0u;
            o[j] = __uint_as_float(lo16 | (hi16 << 16));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (e0 + j < send && e0 + j < total) tape[e0 + j] = o[j];
}

struct RngFillArgs { float* tape; const int* seg_desc; const float* seg_scale; int nseg; long total; unsigned long long seed;
                     const unsigned long long* counter; };
__device__ __forceinline__ void rng_fill_body(const RngFillArgs& a) {
    const unsigned long long ctr = a.counter ? a.counter[0] : 0ull;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q * 4 < a.total; q += (long)gridDim.x * 256)
        fill_quad(a.tape, a.seg_desc, a.seg_scale, a.nseg, a.total, a.seed, ctr, q);
}
__global__ __launch_bounds__(256) void rng_fill_kernel(RngFillArgs a) { rng_fill_body(a); }
__global__ __launch_bounds__(256) void rng_fill_kernel_m(const RngFillArgs* t) {       // one trial per grid plane
    const RngFillArgs a = t[blockIdx.z];
    rng_fill_body(a);
}

// ---- the head of a training step in ONE launch (was: tick, tape fill, batch gather -- 19 us at 256 rows) ----
// Every workgroup reads the step counter and the row cursor AS THE PREVIOUS STEP LEFT THEM and works with counter + 1
// and cursor + stride; the last workgroup to finish (ticket) stores the advanced values, and the Adam step counts, for
// the kernels that follow.  Work: (a) gather the batch rows perm[cursor - B, cursor) (+ spectral noise: N(0, 1) from
// the Philox block of the element's position in the step's Gaussian numbering, or from the tape in parity mode),
// (b) fill the resident slots of the random tape.
struct StepBeginArgs {
    int* steps; int nsteps; unsigned step_mask; unsigned long long* rng_state; int* cursor; int stride; unsigned* ticket;
    const float* spec; const float* aux; const long* idx; int B; int L; int n_aux; float spec_noise;
    const float* noise_tape;      // parity mode: the noise slot on the (host-filled) tape; NULL: generated here
    long noise_goff;              // position of the noise slot in the Gaussian numbering (multiple of 4)
    float* spec_out; float* aux_out;
    float* tape; const int* seg_desc; const float* seg_scale; int nseg; long total;     // nseg == 0: no fill
};
__device__ __forceinline__ void step_begin_body(const StepBeginArgs& a) {
    // Thread 0 reads the old counters, hands them to the workgroup through LDS and only then takes the workgroup's
    // ticket -- at the START (the value comes back while the workgroup works): whoever draws the last one knows that
    // every workgroup has READ the counters (the LDS stores need the loaded values, and precede the ticket in program
    // order) and publishes the advanced ones at its end.  (Tickets of one address serialise at ~40 ns each: 2048
    // workgroups taking them at their END made this kernel 84 us long; the grid is now 64 ... 1024 workgroups.)
    __shared__ unsigned long long s_ctr, s_seed;
    __shared__ int s_cur;
    unsigned my_ticket = 0u;
    if (threadIdx.x == 0) {
        s_ctr = a.rng_state[0] + 1ull;
        s_seed = a.rng_state[1];
        s_cur = a.cursor[0] + a.stride;
        __threadfence();
        my_ticket = atomicAdd(a.ticket, 1u);
    }
    __syncthreads();
    const unsigned long long ctr = s_ctr, seed = s_seed;
    const int cur = s_cur;
    const long* idx = a.idx + (cur - a.B);
    const long nthreads = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)a.B * a.L;
    if ((a.L & 3) == 0) {
        for (long i4 = t0; i4 * 4 < n; i4 += nthreads) {
            const long i = i4 * 4;
            const int b = (int)(i / a.L), l = (int)(i - (long)b * a.L);
            float4 v = *reinterpret_cast<const float4*>(a.spec + (size_t)idx[b] * a.L + l);
            if (a.spec_noise != 0.f) {
                float z[4];
                if (a.noise_tape != nullptr) { const float4 t = *reinterpret_cast<const float4*>(a.noise_tape + i); z[0] = t.x; z[1] = t.y; z[2] = t.z; z[3] = t.w; }
                else normal4((a.noise_goff + i) >> 2, ctr, seed, z);
                v.x += z[0] * a.spec_noise; v.y += z[1] * a.spec_noise; v.z += z[2] * a.spec_noise; v.w += z[3] * a.spec_noise;
            }
            *reinterpret_cast<float4*>(a.spec_out + i) = v;
        }
    } else {
        for (long i = t0; i < n; i += nthreads) {
            const int b = (int)(i / a.L), l = (int)(i - (long)b * a.L);
            float v = a.spec[(size_t)idx[b] * a.L + l];
            if (a.spec_noise != 0.f) {
                if (a.noise_tape != nullptr) v += a.noise_tape[i] * a.spec_noise;
                else { float z[4]; normal4((a.noise_goff + i) >> 2, ctr, seed, z); v += z[(a.noise_goff + i) & 3] * a.spec_noise; }
            }
            a.spec_out[i] = v;
        }
    }
    const long na = (long)a.B * a.n_aux;
    for (long i = t0; i < na; i += nthreads) {
        const int b = (int)(i / a.n_aux), k = (int)(i - (long)b * a.n_aux);
        a.aux_out[i] = a.aux[(size_t)idx[b] * a.n_aux + k];
    }
    if (a.nseg > 0) {
        // the segment table goes to LDS once per workgroup: the binary search of every quad is then six LDS reads, not
        // six dependent global round trips (a thread fills ~16 quads here, not one as in rng_fill_kernel)
        __shared__ int s_desc[4 * 256];
        __shared__ float s_scale[256];
        const int* desc = a.seg_desc;
        const float* scale = a.seg_scale;
        if (a.nseg <= 256) {
            for (int i = threadIdx.x; i < 4 * a.nseg; i += 256) s_desc[i] = a.seg_desc[i];
            for (int i = threadIdx.x; i < a.nseg; i += 256) s_scale[i] = a.seg_scale[i];
            __syncthreads();
            desc = s_desc; scale = s_scale;
        }
        for (long q = t0; q * 4 < a.total; q += nthreads)
            fill_quad(a.tape, desc, scale, a.nseg, a.total, seed, ctr, q);
    }
    // the workgroup that STARTED last publishes the advanced counters (nobody in this launch reads them again)
    if (threadIdx.x == 0 && my_ticket == gridDim.x - 1) {
        for (int i = 0; i < a.nsteps; ++i) if (a.step_mask & (1u << i)) a.steps[i] += 1;
        a.rng_state[0] = ctr;
        raae::mask_keys_store(a.rng_state, seed, ctr);
        a.cursor[0] = cur;
        *a.ticket = 0u;
    }
}
__global__ __launch_bounds__(256) void step_begin_kernel(StepBeginArgs a) { step_begin_body(a); }
__global__ __launch_bounds__(256) void step_begin_kernel_m(const StepBeginArgs* t) {
    const StepBeginArgs a = t[blockIdx.z];
    step_begin_body(a);
}

// Hand-over of a step's deferred tail (StepEngine, `overlap_steps`): the styles the tail's decoder reads are copied out of
// the encoder's workspace (the next step's first forward overwrites it), the dropout hash keys of the step out of the
// counter block (the next step's head rewrites them), and the tail optimizer's step count advances -- one small launch
// at the end of the step's head.
__global__ __launch_bounds__(256) void tail_prepare_kernel(const float* src, float* dst, long n, int* step_counter,
                                                           const unsigned* keys_src, unsigned* keys_dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
    if (i == 0) {
        if (step_counter) *step_counter += 1;
        if (keys_src && keys_dst) { keys_dst[0] = keys_src[0]; keys_dst[1] = keys_src[1]; }
    }
}

}  // namespace

extern "C" int raae_tail_prepare(const float* src, float* dst, long n, int* step_counter, const unsigned long long* rng_state,
                                 unsigned long long* tail_state, void* stream) {
    RAAE_CHECK_ARG(src && dst && n > 0 && n <= (1l << 30));
    RAAE_PLAIN_LAUNCH(tail_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n,
                      step_counter, rng_state ? (const unsigned*)(rng_state + 2) : nullptr,
                      tail_state ? (unsigned*)(tail_state + 2) : nullptr);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_adam_step(float* p, float* m, float* v, const float* g_slabs, long slab_stride,
                              const unsigned short* seg_nslab, long n, const double* hyper, const int* step,
                              int decoupled, int max_nslab, void* stream) {
    RAAE_CHECK_ARG(p && m && v && g_slabs && seg_nslab && hyper && step && n > 0 && (n % 64) == 0 && max_nslab >= 0);
    const AdamArgs a = {p, m, v, g_slabs, slab_stride, seg_nslab, n, hyper, step, decoupled};
    if (max_nslab > 16) {
        long g = (n + 31) / 32;                 // 32 elements per workgroup (8 lanes per element)
        if (g > 4096) g = 4096;
        raae::launch(adam_wide_kernel, adam_wide_kernel_m, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        long g = (n + 255) / 256;               // one element per thread
        if (g > 4096) g = 4096;
        raae::launch(adam_kernel, adam_kernel_m, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a);
    }
    RAAE_LAUNCH_RET();
}

extern "C" int raae_step_tick(int* steps, int n, unsigned mask, unsigned long long* rng_counter, int* cursor,
                              int cursor_inc, void* stream) {
    RAAE_CHECK_ARG(steps && n >= 0 && n <= 32);
    RAAE_PLAIN_LAUNCH(tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, steps, n, mask, rng_counter, cursor, cursor_inc);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_rng_fill(float* tape, const int* seg_desc, const float* seg_scale, int nseg, long total,
                             unsigned long long seed, const unsigned long long* counter, void* stream) {
    RAAE_CHECK_ARG(tape && seg_desc && seg_scale && nseg > 0 && total > 0);
    long g = (total / 4 + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    const RngFillArgs a = {tape, seg_desc, seg_scale, nseg, total, seed, counter};
    raae::launch(rng_fill_kernel, rng_fill_kernel_m, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

extern "C" int raae_step_begin(const raae_step_begin_t* p, void* stream) {
    RAAE_CHECK_ARG(p && p->steps && p->nsteps >= 0 && p->nsteps <= 32 && p->rng_state && p->cursor && p->ticket);
    RAAE_CHECK_ARG(p->spec && p->aux && p->idx && p->spec_out && p->aux_out && p->B > 0 && p->L > 0 && p->n_aux > 0);
    RAAE_CHECK_ARG(p->nseg == 0 || (p->tape && p->seg_desc && p->seg_scale && p->total > 0));
    RAAE_CHECK_ARG((p->noise_goff & 3) == 0 && p->noise_goff >= 0);
    StepBeginArgs a;
    a.steps = p->steps; a.nsteps = p->nsteps; a.step_mask = p->step_mask; a.rng_state = p->rng_state; a.cursor = p->cursor;
    a.stride = p->stride; a.ticket = p->ticket; a.spec = p->spec; a.aux = p->aux; a.idx = p->idx; a.B = p->B; a.L = p->L;
    a.n_aux = p->n_aux; a.spec_noise = p->spec_noise; a.noise_tape = p->noise_tape; a.noise_goff = p->noise_goff;
    a.spec_out = p->spec_out; a.aux_out = p->aux_out; a.tape = p->tape; a.seg_desc = p->seg_desc; a.seg_scale = p->seg_scale;
    a.nseg = p->nseg; a.total = p->nseg > 0 ? p->total : 0;
    long work = (long)p->B * p->L / 4;
    if (a.total / 4 > work) work = a.total / 4;
    long g = (work + 4095) / 4096;          // ~16 quads per thread ...
    if (g > 1024) g = 1024;                 // ... up to 1024 workgroups (their tickets, ~40 us, come back during ~45 us of fill at 4096 rows)
    if (g < 64) g = 64;
    raae::launch(step_begin_kernel, step_begin_kernel_m, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a);
    RAAE_LAUNCH_RET();
}

// ---------------------------------------------------------------- runtime plumbing
extern "C" int raae_graph_begin(void* stream) {
    return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
}
extern "C" int raae_graph_end(void* stream, void** graph_exec) {
    RAAE_CHECK_ARG(graph_exec);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess) return (int)e;
    if (const char* dot = getenv("RAAE_GRAPH_DOT")) (void)hipGraphDebugDotPrint(graph, dot, hipGraphDebugDotFlagsVerbose);
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *graph_exec = (void*)ex;
    return 0;
}
extern "C" int raae_graph_launch(void* graph_exec, void* stream) {
    return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
}
extern "C" int raae_graph_destroy(void* graph_exec) { return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec); }
extern "C" int raae_event_create(void** ev) {
    RAAE_CHECK_ARG(ev);
    hipEvent_t e; hipError_t r = hipEventCreate(&e);
    *ev = (void*)e; return (int)r;
}
extern "C" int raae_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
extern "C" int raae_event_elapsed_ms(void* start, void* stop, float* ms) {
    RAAE_CHECK_ARG(ms);
    hipError_t r = hipEventSynchronize((hipEvent_t)stop);
    if (r != hipSuccess) return (int)r;
    return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
extern "C" int raae_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
extern "C" int raae_stream_sync(void* stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }
extern "C" const char* raae_error_string(int code) {
    if (code == RAAE_EINVAL) return "raae: invalid argument (host-side shape validation failed; nothing launched)";
    return hipGetErrorString((hipError_t)code);
}
extern "C" int raae_device_info(int* cu_count, int* lds_bytes, char* name, int name_len) {
    hipDeviceProp_t prop; int dev = 0;
    hipError_t r = hipGetDevice(&dev);
    if (r != hipSuccess) return (int)r;
    r = hipGetDeviceProperties(&prop, dev);
    if (r != hipSuccess) return (int)r;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (name && name_len > 0) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return 0;
}
extern "C" int raae_abi_version(void) { return RAAE_ABI_VERSION; }
#ifndef RAAE_SOURCE_DIGEST
#define RAAE_SOURCE_DIGEST "unknown"
#endif
extern "C" const char* raae_source_digest(void) { return RAAE_SOURCE_DIGEST; }

// ---------------------------------------------------------------- data-parallel helper
namespace {
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* g_slabs, long slab_stride,
                                                          const unsigned short* seg_nslab, long n, float* out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ns = seg_nslab[i >> 6];
        float g = 0.f;
        int s = 0;
        for (; s + 8 <= ns; s += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = g_slabs[(size_t)(s + u) * slab_stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += t[u];
        }
        for (; s < ns; ++s) g += g_slabs[(size_t)s * slab_stride + i];
        out[i] = g;
    }
}
// the summation tree of adam_wide_kernel (8 lanes per element), so that the data-parallel path adds the
// slabs of a rank in exactly the order the single-GPU update does
__global__ __launch_bounds__(256) void slab_reduce_wide_kernel(const float* g_slabs, long slab_stride,
                                                               const unsigned short* seg_nslab, long n, float* out) {
    const int lane = threadIdx.x & 63, el = lane & 7, ch = lane >> 3;
    const long wave0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    for (long base = wave0; base < n; base += (long)gridDim.x * 32) {
        const long i = base + el;
        const int ns = seg_nslab[i >> 6];
        float g = 0.f;
        for (int s = ch; s < ns; s += 64) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = s + 8 * u;
                t[u] = g_slabs[(size_t)(r < ns ? r : ch) * slab_stride + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) g += (s + 8 * u < ns) ? t[u] : 0.f;
        }
        g += __shfl_xor(g, 8, 64);
        g += __shfl_xor(g, 16, 64);
        g += __shfl_xor(g, 32, 64);
        if (ch == 0) out[i] = g;
    }
}
}  // namespace

extern "C" int raae_slab_reduce(const float* g_slabs, long slab_stride, const unsigned short* seg_nslab, long n,
                                float* out, int max_nslab, void* stream) {
    RAAE_CHECK_ARG(g_slabs && seg_nslab && out && n > 0 && (n % 64) == 0 && max_nslab >= 0);
    if (max_nslab > 16) {
        long g = (n + 31) / 32;
        if (g > 4096) g = 4096;
        RAAE_PLAIN_LAUNCH(slab_reduce_wide_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, g_slabs, slab_stride,
                           seg_nslab, n, out);
    } else {
        long g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        RAAE_PLAIN_LAUNCH(slab_reduce_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, g_slabs, slab_stride,
                           seg_nslab, n, out);
    }
    RAAE_LAUNCH_RET();
}


// ---------------------------------------------------------------- batched launches over trials (raae_common.h)
#include <vector>
namespace {
struct LaunchRec { const void* fn; dim3 grid, block; unsigned lds, nbytes; unsigned char args[4096]; };
struct Recording { std::vector<LaunchRec> recs; bool bad = false; };
thread_local Recording* g_recording = nullptr;
struct MultiProgram { std::vector<LaunchRec> recs; std::vector<size_t> off; unsigned char* table = nullptr; int T = 0; };
}  // namespace
void raae::record_launch(const void* multi_fn, dim3 grid, dim3 block, size_t lds, const void* args, size_t nbytes) {
    Recording* r = g_recording;
    if (!r) return;
    LaunchRec rec;
    if (nbytes > sizeof(rec.args) || grid.z != 1) { r->bad = true; return; }
    rec.fn = multi_fn; rec.grid = grid; rec.block = block; rec.lds = (unsigned)lds; rec.nbytes = (unsigned)nbytes;
    memcpy(rec.args, args, nbytes);
    r->recs.push_back(rec);
}
void raae::record_unsupported() {
    if (g_recording) g_recording->bad = true;
}
extern "C" int raae_record_begin(void) {
    if (g_recording) return RAAE_EINVAL;
    g_recording = new Recording();
    return 0;
}
extern "C" int raae_record_end(void** handle, int* n_launches) {
    RAAE_CHECK_ARG(handle && g_recording);
    Recording* r = g_recording;
    g_recording = nullptr;
    if (r->bad) { delete r; return RAAE_EINVAL; }
    if (n_launches) *n_launches = (int)r->recs.size();
    *handle = r;
    return 0;
}
extern "C" int raae_record_free(void* handle) { delete (Recording*)handle; return 0; }
extern "C" int raae_multi_build(void* const* handles, int T, void** program) {
    RAAE_CHECK_ARG(handles && program && T >= 1 && T <= 64);
    const Recording* r0 = (const Recording*)handles[0];
    RAAE_CHECK_ARG(r0 && !r0->recs.empty());
    MultiProgram* mp = new MultiProgram();
    mp->T = T; mp->recs = r0->recs;
    size_t total = 0;
    for (size_t i = 0; i < r0->recs.size(); ++i) {
        mp->off.push_back(total);
        total += ((size_t)T * r0->recs[i].nbytes + 255) & ~(size_t)255;
    }
    std::vector<unsigned char> host(total, 0);
    for (int t = 0; t < T; ++t) {
        const Recording* r = (const Recording*)handles[t];
        if (!r || r->recs.size() != r0->recs.size()) { delete mp; return RAAE_EINVAL; }
        for (size_t i = 0; i < r->recs.size(); ++i) {
            const LaunchRec &a = r0->recs[i], &b = r->recs[i];
            // the trials must be structurally identical: same kernel instance, geometry and LDS at every launch
            if (a.fn != b.fn || a.grid.x != b.grid.x || a.grid.y != b.grid.y || a.block.x != b.block.x || a.lds != b.lds ||
                a.nbytes != b.nbytes) { delete mp; return RAAE_EINVAL; }
            memcpy(host.data() + mp->off[i] + (size_t)t * a.nbytes, b.args, a.nbytes);
        }
    }
    hipError_t e = hipMalloc((void**)&mp->table, total);
    if (e != hipSuccess) { delete mp; return (int)e; }
    e = hipMemcpy(mp->table, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(mp->table); delete mp; return (int)e; }
    *program = mp;
    return 0;
}
extern "C" int raae_multi_launch(void* program, void* stream) {
    RAAE_CHECK_ARG(program);
    MultiProgram* mp = (MultiProgram*)program;
    for (size_t i = 0; i < mp->recs.size(); ++i) {
        const LaunchRec& r = mp->recs[i];
        void* tptr = mp->table + mp->off[i];
        void* params[1] = {&tptr};
        const hipError_t e = hipLaunchKernel(r.fn, dim3(r.grid.x, r.grid.y, mp->T), r.block, params, r.lds, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
extern "C" int raae_multi_count(void* program) { return program ? (int)((MultiProgram*)program)->recs.size() : 0; }
extern "C" int raae_multi_free(void* program) {
    if (!program) return 0;
    MultiProgram* mp = (MultiProgram*)program;
    if (mp->table) (void)hipFree(mp->table);
    delete mp;
    return 0;
}
