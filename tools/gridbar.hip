// Micro-benchmark behind DESIGN.md's "persistent kernel" paragraph: what does a stage boundary cost as
//   (A) a device-wide barrier inside ONE persistent kernel (256 workgroups, release/acquire at agent scope), or
//   (B) a kernel boundary between two nodes of a captured hipGraph ?
// Each stage does what a fused residual-block kernel does around its BatchNorm: every workgroup publishes a partial
// statistic, then every workgroup reads ALL partials.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double read_all(const double* part, int n) {
    double s = 0.0;
    for (int i = threadIdx.x & 63; i < n; i += 64) s += part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

// one stage as its own kernel: read the previous stage's partials, publish this stage's
__global__ __launch_bounds__(256) void stage_kernel(const double* in, double* out, int n) {
    const double s = read_all(in, n);
    if (threadIdx.x == 0) out[blockIdx.x] = s * 1e-3 + (double)blockIdx.x;
}

// all stages in one kernel; bounded spin (a barrier that cannot complete sets *err and gives up)
template <bool SLEEP>
__global__ __launch_bounds__(256) void persistent_kernel(double* buf, int n, int nstage, unsigned* counter, int* err) {
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const double* in = buf + (size_t)(st & 1) * n;
        double* out = buf + (size_t)((st + 1) & 1) * n;
        const double s = read_all(in, n);
        if (threadIdx.x == 0) out[blockIdx.x] = s * 1e-3 + (double)blockIdx.x;
        // ---- device-wide barrier: release own stores, arrive, wait for all, acquire
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(st + 1) * gridDim.x;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1L << 22)) { *err = 1; s_bad = 1; break; }
                if (SLEEP) __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();
        }
        __syncthreads();
        if (s_bad) return;
    }
}

int main() {
    const int G = 256, NST = 200;
    double* buf; unsigned* counter; int* err;
    CHECK(hipMalloc(&buf, sizeof(double) * 2 * G));
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&err, 4));
    CHECK(hipMemset(buf, 0, sizeof(double) * 2 * G)); CHECK(hipMemset(err, 0, 4));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NST; ++i) hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(256), 0, st, buf + (size_t)(i & 1) * G, buf + (size_t)((i + 1) & 1) * G, G);
    CHECK(hipStreamEndCapture(st, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, st)); CHECK(hipGraphLaunch(ge, st)); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of %d stage kernels (%d workgroups): %.2f us per stage\n", NST, G, 1e3 * ms / NST);
    }
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipMemsetAsync(counter, 0, 4, st));
        CHECK(hipEventRecord(e0, st));
        if (rep < 3) hipLaunchKernelGGL(persistent_kernel<true>, dim3(G), dim3(256), 0, st, buf, G, NST, counter, err);
        else hipLaunchKernelGGL(persistent_kernel<false>, dim3(G), dim3(256), 0, st, buf, G, NST, counter, err);
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        int herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("persistent kernel (%s), %d device-wide barriers (%d workgroups): %.2f us per stage%s\n",
               rep < 3 ? "s_sleep in the spin" : "busy spin", NST, G, 1e3 * ms / NST, herr ? "  [a barrier timed out]" : "");
    }
    return 0;
}
