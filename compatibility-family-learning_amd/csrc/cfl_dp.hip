// cfl_dp.hip -- one-shot gradient exchange of the data-parallel pair step (new functionality: the reference is
// single-device).  Opt-in alternative (CFL_DP_EXCHANGE=oneshot) to the RCCL ring all-reduce of
// cfl/engine.py: at 1.57 MB per step a ring over 8 GPUs is 14 latency-bound hops; here every rank PUSHES its flat
// buffer [gradient | 16 scalars] into a slot of every peer's exchange buffer over the 7 point-to-point xGMI links
// at once (peer memory mapped through hipIpc by the host side, cfl/dp_exchange.py), and the Adam launch itself sums
// the N slots it finds in LOCAL memory in rank order -- deterministic, identical on every rank, no reduction tree:
//
//   cfl_dp_push   src -> slot[rank] of every peer (plain 16-byte stores), __threadfence_system() by every block,
//                 last block (device-scope ticket) releases at system scope and writes the step's generation number
//                 into flag[rank] of every peer;
//   cfl_dp_wait   ONE wave polls the `world` local flags (system-scope loads, bounded) -- a single workgroup, so that
//                 ranks which share a GPU (the functional tests) can never starve each other of CUs --, then acquires;
//   cfl_dp_adam   reads the N slots with system-scope (sc0 sc1) loads -- peers wrote them into this GPU's memory past
//                 its L2 --, sums them in rank order, scales by 1/N, stores the summed buffer (gradient + scalars,
//                 what every rank logs) and applies TF-Adam to the parameter part.
// Double buffering by step parity makes the flags sufficient: a rank overwrites slot parity p at step t + 2 only
// after its own step t + 1, which waited for every peer's step-t + 1 flag, which a peer raises after its step-t Adam
// (the reader of parity p) in stream order.
// Exercised functionally by two processes on one GPU (tests/test_data_parallel_gpu.py); no multi-GPU node was
// available to this build, so cross-device visibility rests on the system-scope release / acquire above.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/cfl_hip.h"

extern int cfl_set_err(int code, const char *fmt, ...);

typedef float dp_f32x4 __attribute__((ext_vector_type(4)));

#define CFL_DP_MAX_WORLD 16
#define CFL_DP_SPIN_LIMIT (1 << 24)

struct DpPeers {
    float *slot[CFL_DP_MAX_WORLD];        // this rank's slot inside every peer's exchange buffer (current parity)
    unsigned *flag[CFL_DP_MAX_WORLD];     // this rank's flag word inside every peer's flag array (current parity)
    int world;
};

__global__ __launch_bounds__(256) void cfl_dp_push_kernel(const float *src, long long n4, DpPeers p, unsigned gen,
                                                          unsigned *ticket) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const dp_f32x4 v = ((const dp_f32x4 *)src)[i];
        for (int r = 0; r < p.world; ++r) ((dp_f32x4 *)p.slot[r])[i] = v;
    }
    __threadfence_system();                 // this block's stores are visible system-wide ...
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) {                             // ... and so are all blocks' once the last one has arrived
        __threadfence_system();
        if ((int)threadIdx.x < p.world)
            __hip_atomic_store(p.flag[threadIdx.x], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *ticket = 0;  // next launch on this stream starts from zero
    }
}

__global__ __launch_bounds__(64) void cfl_dp_wait_kernel(const unsigned *flags, int world, unsigned gen, int *lost) {
    const int lane = threadIdx.x;
    int spins = 0;
    for (;;) {
        const unsigned v = lane < world ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : gen;
        if (__builtin_amdgcn_ballot_w64(v != gen) == 0) break;
        if (++spins > CFL_DP_SPIN_LIMIT) {   // a peer never arrived: poison the update instead of hanging the GPU
            if (lane == 0) *lost = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __threadfence_system();
}

// 4 x 16 bytes straight from memory (system scope: not from this GPU's L2, which peers' writes bypass); ONE statement
// that ends with its own wait (the compiler does not track loads issued inside inline asm)
__device__ __forceinline__ void load_sys16x4(const float *p0, const float *p1, const float *p2, const float *p3,
                                             dp_f32x4 (&o)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

__global__ __launch_bounds__(256) void cfl_dp_adam_kernel(float *theta, float *m, float *v, const float *slots,
                                                          int world, long long n4, long long nadam4, float *sum_out,
                                                          float lr_t, float b1, float b2, float eps, const int *lost) {
    const long long stride = (long long)gridDim.x * 256;
    const float scale = 1.f / (float)world;
    const bool bad = *lost != 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        dp_f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int r0 = 0; r0 < world; r0 += 4) {   // four slots in flight, added in rank order
            dp_f32x4 s[4];
            const float *q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = slots + ((long long)(r0 + k < world ? r0 + k : 0) * n4 + i) * 4;
            load_sys16x4(q[0], q[1], q[2], q[3], s);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r0 + k < world) g = (r0 + k == 0) ? s[k] : g + s[k];
        }
        if (bad) g = (dp_f32x4){NAN, NAN, NAN, NAN};
        ((dp_f32x4 *)sum_out)[i] = g;       // the all-reduced buffer (gradient sums | scalar sums), as RCCL would leave it
        if (i < nadam4) {
            g *= scale;
            dp_f32x4 mm = ((dp_f32x4 *)m)[i], vv = ((dp_f32x4 *)v)[i], th = ((dp_f32x4 *)theta)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // TF-1.x Adam, the same operations as adam1() of cfl_hip.hip
                mm[e] = fmaf(b1, mm[e], (1.f - b1) * g[e]);
                vv[e] = fmaf(b2, vv[e], ((1.f - b2) * g[e]) * g[e]);
                th[e] -= lr_t * mm[e] / (sqrtf(vv[e]) + eps);
            }
            ((dp_f32x4 *)m)[i] = mm;
            ((dp_f32x4 *)v)[i] = vv;
            ((dp_f32x4 *)theta)[i] = th;
        }
    }
}

extern "C" int cfl_dp_push(const float *src, int64_t n, float *const *peer_slots, uint32_t *const *peer_flags,
                           int32_t world, uint32_t generation, uint32_t *ticket, cfl_stream_t stream) {
    if (!src || !peer_slots || !peer_flags || !ticket) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_push: NULL pointer");
    if (world < 1 || world > CFL_DP_MAX_WORLD) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_push: world %d out of range", world);
    if (n <= 0 || n % 4) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_push: n=%lld must be a positive multiple of 4", (long long)n);
    DpPeers p;
    memset(&p, 0, sizeof(p));
    p.world = world;
    for (int r = 0; r < world; ++r) {
        if (!peer_slots[r] || !peer_flags[r] || ((uintptr_t)peer_slots[r] & 15))
            return cfl_set_err(CFL_E_SHAPE, "cfl_dp_push: peer %d slot / flag NULL or misaligned", r);
        p.slot[r] = peer_slots[r];
        p.flag[r] = peer_flags[r];
    }
    const long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(cfl_dp_push_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, n4, p, generation, ticket);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_dp_push launch failed");
}

extern "C" int cfl_dp_wait(const uint32_t *flags, int32_t world, uint32_t generation, int32_t *lost, cfl_stream_t stream) {
    if (!flags || !lost) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_wait: NULL pointer");
    if (world < 1 || world > CFL_DP_MAX_WORLD) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_wait: world %d out of range", world);
    hipLaunchKernelGGL(cfl_dp_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flags, world, generation, lost);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_dp_wait launch failed");
}

extern "C" int cfl_dp_adam(float *theta, float *m, float *v, const float *slots, int32_t world, int64_t n,
                           int64_t n_adam, float *sum_out, float lr_t, float beta1, float beta2, float eps,
                           const int32_t *lost, cfl_stream_t stream) {
    if (!theta || !m || !v || !slots || !sum_out || !lost) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_adam: NULL pointer");
    if (world < 1 || world > CFL_DP_MAX_WORLD) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_adam: world %d out of range", world);
    if (n <= 0 || n % 4 || n_adam < 0 || n_adam % 4 || n_adam > n)
        return cfl_set_err(CFL_E_SHAPE, "cfl_dp_adam: n=%lld n_adam=%lld", (long long)n, (long long)n_adam);
    const long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cfl_dp_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, m, v, slots, world, n4,
                       (long long)(n_adam / 4), sum_out, lr_t, beta1, beta2, eps, (const int *)lost);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_dp_adam launch failed");
}
