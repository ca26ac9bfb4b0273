// Microbenchmark (round 6): what could ONE persistent launch with two grid barriers save over THREE dependent
// launches at the reference's own batch size (B = 100: 112 / 50 / 129 workgroups of short latency chains)?  Both forms run the
// same three phases of trivial dependent work (each workgroup reads a few KB the previous phase wrote, writes a few KB); the
// difference is what separates the phases: a kernel boundary (end of launch, dispatch, first-wave ramp) or an in-kernel grid barrier
// (arrival counter + sc1 polling, the hand-off of MI355X_MICROARCH.md).  Prints us per "step" for both.  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void phase(const float *in, float *out, int wg, int nwg, int kb) {
    // every workgroup reads kb KiB spread over what ALL workgroups of the previous phase wrote, reduces, writes 1 KiB
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < kb; ++i) {
        const int src = (wg * 7 + i * 13) % nwg;
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(in + (size_t)src * 256 + (threadIdx.x & 63) * 4) : "memory");
        acc += v;
    }
    if (threadIdx.x < 64) {
        float *p = out + (size_t)wg * 256 + threadIdx.x * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(acc) : "memory");
    }
}

__global__ __launch_bounds__(256) void k_phase(const float *in, float *out, int nwg, int kb) {
    if ((int)blockIdx.x < nwg) phase(in, out, blockIdx.x, nwg, kb);
}

// RELEASE != 0: agent-scope release on the arrival (an L2 write-back); 0: relaxed -- enough here, every handed-off byte is an sc1
// store drained (s_waitcnt vmcnt(0)) before the barrier, the form the library's own in-launch hand-offs use
template <int RELEASE>
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (RELEASE) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int RELEASE>
__global__ __launch_bounds__(256) void k_persist(float *a, float *b, float *c, float *d, unsigned *ctr, unsigned gen, int n1, int n2, int n3,
                                                 int kb) {
    const unsigned G = gridDim.x;
    if ((int)blockIdx.x < n1) phase(a, b, blockIdx.x, n1, kb);
    grid_barrier<RELEASE>(ctr, (2 * gen + 1) * G);
    if ((int)blockIdx.x < n2) phase(b, c, blockIdx.x, n2, kb);
    grid_barrier<RELEASE>(ctr, (2 * gen + 2) * G);
    if ((int)blockIdx.x < n3) phase(c, d, blockIdx.x, n3, kb);
}

int main() {
    const int n1 = 112, n2 = 50, n3 = 129, N = 129, steps = 2000;
    float *a, *b, *c, *d;
    unsigned *ctr;
    CK(hipMalloc(&a, N * 1024)); CK(hipMalloc(&b, N * 1024)); CK(hipMalloc(&c, N * 1024)); CK(hipMalloc(&d, N * 1024));
    CK(hipMalloc(&ctr, 64));
    CK(hipMemset(a, 0, N * 1024)); CK(hipMemset(b, 0, N * 1024)); CK(hipMemset(c, 0, N * 1024)); CK(hipMemset(ctr, 0, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kb : {4, 16, 64}) {
        float ms3 = 0, ms1 = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < steps; ++i) {
                hipLaunchKernelGGL(k_phase, dim3(n1), dim3(256), 0, st, a, b, n1, kb);
                hipLaunchKernelGGL(k_phase, dim3(n2), dim3(256), 0, st, b, c, n2, kb);
                hipLaunchKernelGGL(k_phase, dim3(n3), dim3(256), 0, st, c, d, n3, kb);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms3, e0, e1));
        }
        float ms1r = 0;
        for (int rel = 0; rel < 2; ++rel) {
            CK(hipMemset(ctr, 0, 64));
            CK(hipDeviceSynchronize());
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < steps; ++i) {
                    if (rel) hipLaunchKernelGGL(k_persist<1>, dim3(N), dim3(256), 0, st, a, b, c, d, ctr, (unsigned)(rep * steps + i), n1, n2, n3, kb);
                    else hipLaunchKernelGGL(k_persist<0>, dim3(N), dim3(256), 0, st, a, b, c, d, ctr, (unsigned)(rep * steps + i), n1, n2, n3, kb);
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(rel ? &ms1r : &ms1, e0, e1));
            }
        }
        // one phase alone, back to back (the per-launch floor of this box)
        float msk = 0;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(k_phase, dim3(n1), dim3(256), 0, st, a, b, n1, kb);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&msk, e0, e1));
        printf("dependent loads per workgroup and phase %2d: three launches %.2f us/step | one persistent launch + two grid barriers %.2f us/step "
               "(relaxed arrivals; %.2f with agent-scope release) | one such launch alone %.2f us\n",
               kb, 1e3 * ms3 / steps, 1e3 * ms1 / steps, 1e3 * ms1r / steps, 1e3 * msk / steps);
    }
    return 0;
}
