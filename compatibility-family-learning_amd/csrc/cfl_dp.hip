// cfl_dp.hip -- one-shot gradient exchange of the data-parallel pair step (new functionality: the reference is
// single-device).  Opt-in alternative (CFL_DP_EXCHANGE=oneshot) to the RCCL ring all-reduce of
// cfl/engine.py, written for what xGMI is -- seven point-to-point links per GPU, no switch: a ring all-reduce of the
// 1.57 MB buffer [gradient | 16 scalars] over 8 GPUs is 14 latency-bound hops; here every byte crosses exactly one link,
// every link carries 1/N of the buffer twice, and every rank applies Adam to 1/N of the parameters:
//
//   cfl_dp_rs_push    (reduce-scatter, send side) slice s of this rank's buffer -> row `rank` of rank s's slot array,
//                     all N - 1 links at once; every block drains its stores, the last block (device-scope ticket)
//                     raises flag A[rank] = generation in every peer;
//   cfl_dp_rs_adam    waits (bounded by wall-clock time) until the N local A flags carry the generation, sums the N rows
//                     of ITS slice in rank order (system-scope loads; deterministic, no reduction tree), applies TF-Adam
//                     to its slice of theta / m / v -- the Adam slots are sharded, each rank keeps 1/N of them current --
//                     and pushes the UPDATED theta slice (and, past the parameters, the summed scalars) into every
//                     peer's stage buffer; the last block raises flag B[rank] in every peer;
//   cfl_dp_rs_gather  (all-gather, receive side) waits for the N - 1 B flags and copies the peers' slices from the local
//                     stage buffer into theta (and the scalar sums into the caller's buffer).
// Round 6: (1) the push is FUSED into the weight-gradient launch (csrc/pair_grad.h: the tile finishers store their entries into
// the owners' slots, the launch's last workgroup raises the A flags) -- cfl_dp_rs_push stays for plans without that form; (2) the
// sharded Adam and the all-gather run in ONE launch (cfl_dp_adam_gather_kernel; CFL_DP_SPLIT_ADAM=1: the two launches below);
// (3) no __threadfence_system() anywhere: write-through system-scope stores + drains (see "Memory model" below).  A step is
// proj, mid, grad(+push), adam_gather: FOUR launches behind one library call (cfl_pair_dp_step[s]_idx_planes, cfl_hip.hip).
// Every rank ends the step with bit-identical parameters by construction (each slice has ONE writer; the others copy).
// Per step and link: 2 x (n / N) floats (0.39 MB at the headline shape and N = 8, against 1.57 MB for the push-everything
// form of round 3 and 2 x 7/8 x 1.57 MB around a ring).
// Slots, stage buffers and flags live in FINE-GRAINED device memory (cfl_dp_alloc): HIP only promises that a peer's
// stores become visible to a RUNNING kernel of the owner for such allocations.  Double buffering by step parity makes the
// flags sufficient (a rank overwrites parity p of a peer at step t + 2 only after its own step t + 1 gather, which waited
// for that peer's step-t + 1 B flag, raised after the peer's step-t + 1 Adam, i.e. after every reader of parity p of step t).
// The waits are bounded by TIME (s_memrealtime, tens of seconds by default): a rank that is merely late -- a chief writing a
// checkpoint -- is waited for; a peer that never arrives sets *lost, poisons the update with NaN, and the host raises at its
// next read-back (cfl/dp_exchange.py: check()).
// Exercised functionally by two processes on one GPU (tests/test_data_parallel_gpu.py); no multi-GPU node was
// available to this build.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/cfl_hip.h"
#include "theta_planes.h"

extern int cfl_set_err(int code, const char *fmt, ...);
extern void *cfl_prof_scope_begin(void *stream, int kind);   // cfl_profile_enable's event pairs (cfl_hip.hip)
extern void cfl_prof_scope_end(void *scope);
struct DpProf {
    void *h;
    DpProf(cfl_stream_t st) : h(cfl_prof_scope_begin(st, CFL_K_DP)) {}
    ~DpProf() { if (h) cfl_prof_scope_end(h); }
};

typedef float dp_f32x4 __attribute__((ext_vector_type(4)));

struct DpPeers {
    float *slot[CFL_DP_MAX_WORLD];        // per peer: where this rank writes (row of the slot array / stage buffer, current parity)
    unsigned *flag[CFL_DP_MAX_WORLD];     // per peer: this rank's flag word (current parity)
    int world;
};

// Memory model of the exchange (round 6: no __threadfence_system() anywhere -- on gfx950 it writes back / invalidates the whole L2
// per calling wave, and hundreds of workgroups doing that at the end of a launch cost a step tens of microseconds, measured).
//  * every byte handed to another rank is stored with a SYSTEM-SCOPE WRITE-THROUGH store (sc0 sc1) into fine-grained memory --
//    never dirty in an L2 -- and every storing wave drains its stores (s_waitcnt vmcnt(0): acknowledged by the memory system)
//    before its workgroup counts itself on the launch's ticket;
//  * the workgroup that sees the last count raises this rank's flag word in every peer (system-scope store): the flag leaves the
//    GPU behind every data store of the launch, and stores of one GPU to one peer are delivered in order;
//  * the receiver polls its flag words with system-scope loads and reads the handed-off bytes with system-scope (sc0 sc1) loads
//    to registers: nothing it reads can come from a stale line of its own caches.
// The in-launch hand-offs of the weight-gradient kernels use the same form one scope further in (sc1; MI355X_MICROARCH.md).
__device__ __forceinline__ void dp_store_sys16(float *p, dp_f32x4 v) {
    // (one asm statement, with the wait states a VALU write to the data registers of a > 64-bit store needs behind it: the
    // compiler's hazard recogniser does not see inside inline asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

// last block of a launch: every block drained its stores before it counted; raise this rank's flag in every peer
__device__ __forceinline__ void dp_signal_peers(const DpPeers &p, unsigned gen, unsigned *ticket) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) {
        if ((int)threadIdx.x < p.world)
            __hip_atomic_store(p.flag[threadIdx.x], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *ticket = 0;  // the next launch on this stream starts from zero
    }
}

// wave 0 polls the `world` local flags (one per lane, system scope) until all carry `gen` or `ticks` of the 100 MHz
// real-time counter have passed; returns (to every thread of the block) whether they all arrived
__device__ __forceinline__ bool dp_wait_flags(const unsigned *flags, int world, int skip, unsigned gen, unsigned long long ticks,
                                              int *lost) {
    __shared__ int ok_s;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        for (;;) {
            const unsigned v = (lane < world && lane != skip)
                                   ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : gen;
            if (__builtin_amdgcn_ballot_w64(v != gen) == 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) {
            ok_s = ok;
            if (!ok) *lost = 1;
        }
    }
    __syncthreads();
    return ok_s != 0;
}

__device__ __forceinline__ dp_f32x4 load_sys16(const float *p) {   // system scope: not from this GPU's L2, which peers' writes bypass
    dp_f32x4 o;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(p) : "memory");
    return o;
}

// 4 x 16 bytes straight from memory; ONE statement that ends with its own wait (the compiler does not track loads issued
// inside inline asm)
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

// ... eight rows in ONE round trip (the merged Adam + gather kernel: a rank of an 8-GPU job sums eight rows per element; two
// statements of four were two dependent system-scope round trips)
__device__ __forceinline__ void load_sys16x8(const float *const (&p)[8], dp_f32x4 (&o)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}

__global__ __launch_bounds__(256) void cfl_dp_rs_push_kernel(const float *src, long long n4, long long slice4, DpPeers p,
                                                             unsigned gen, unsigned *ticket) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int s = (int)(i / slice4);                       // owner of element i
        dp_store_sys16(p.slot[s] + 4 * (i - (long long)s * slice4), ((const dp_f32x4 *)src)[i]);
    }
    dp_signal_peers(p, gen, ticket);
}

__global__ __launch_bounds__(256) void cfl_dp_rs_adam_kernel(float *theta, float *m, float *v, const float *gslots,
                                                             const unsigned *flags, int world, int rank, long long n4,
                                                             long long nadam4, long long slice4, float *sum_out, DpPeers p,
                                                             float lr_t, float b1, float b2, float eps, unsigned gen, int *lost,
                                                             unsigned long long ticks, unsigned *ticket) {
    const bool ok = dp_wait_flags(flags, world, -1, gen, ticks, lost);
    const long long lo = (long long)rank * slice4, hi = lo + slice4 < n4 ? lo + slice4 : n4;
    const float scale = 1.f / (float)world;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) {
        dp_f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int r0 = 0; r0 < world; r0 += 4) {   // four rows in flight, added in rank order
            dp_f32x4 s[4];
            const float *q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = gslots + ((long long)(r0 + k < world ? r0 + k : 0) * slice4 + (i - lo)) * 4;
            load_sys16x4(q[0], q[1], q[2], q[3], s);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r0 + k < world) g = (r0 + k == 0) ? s[k] : g + s[k];
        }
        if (!ok) g = (dp_f32x4){NAN, NAN, NAN, NAN};
        ((dp_f32x4 *)sum_out)[i] = g;       // this rank's slice of the summed buffer (gradient sums | scalar sums)
        dp_f32x4 out = g;
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
            out = th;
        }
        for (int r = 0; r < world; ++r)
            if (r != rank) dp_store_sys16(p.slot[r] + 4 * i, out);   // all-gather, send side: one store per link
    }
    dp_signal_peers(p, gen, ticket);
}

__global__ __launch_bounds__(256) void cfl_dp_rs_gather_kernel(float *theta, float *sum_out, const float *stage,
                                                               const unsigned *flags, int world, int rank, long long n4,
                                                               long long nadam4, long long slice4, unsigned gen, int *lost,
                                                               unsigned long long ticks, ThetaPlaneRegions pr, float *scalars_copy) {
    const bool ok = dp_wait_flags(flags, world, rank, gen, ticks, lost);
    const long long lo = (long long)rank * slice4, hi = lo + slice4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        // the step's global scalar SUMS (the 16 floats behind the parameters) also go to the caller's second destination -- a
        // pinned host ring slot of the training loop: no copy command on the stream
        const bool sc = scalars_copy && i >= nadam4 && i < nadam4 + 4;
        if (i >= lo && i < hi) {            // own slice: written by cfl_dp_rs_adam_kernel (the launch before this one)
            if (pr.planes && i < nadam4) theta_planes_store4(pr, i * 4, ((const dp_f32x4 *)theta)[i]);
            if (sc) ((dp_f32x4 *)scalars_copy)[i - nadam4] = ((const dp_f32x4 *)sum_out)[i];
            continue;
        }
        dp_f32x4 x = load_sys16(stage + i * 4);
        if (!ok) x = (dp_f32x4){NAN, NAN, NAN, NAN};
        if (i < nadam4) {
            ((dp_f32x4 *)theta)[i] = x;
            theta_planes_store4(pr, i * 4, x);
        } else {
            ((dp_f32x4 *)sum_out)[i] = x;
            if (sc) ((dp_f32x4 *)scalars_copy)[i - nadam4] = x;
        }
    }
}

// cfl_dp_rs_adam + cfl_dp_rs_gather in ONE launch (round 6).  The two phases of a rank do not depend on each other -- the gather of
// rank r reads what the OTHER ranks' Adam phases pushed -- so no grid barrier is needed between them: a workgroup does its share
// of the sharded Adam (waits for the A flags only if it has a share), counts itself (the last one raises this rank's B flags),
// then waits for the peers' B flags and copies its share of their slices.  No deadlock: phase 1 of every rank waits only for
// weight-gradient launches (A flags), phase 2 only for peers' phase 1.  One launch and one dispatch less per step.
__global__ __launch_bounds__(256) void cfl_dp_adam_gather_kernel(float *theta, float *m, float *v, const float *gslots,
                                                                 const unsigned *flags_a, const unsigned *flags_b, const float *stage,
                                                                 int world, int rank, long long n4, long long nadam4, long long slice4,
                                                                 float *sum_out, DpPeers p, float lr_t, float b1, float b2, float eps,
                                                                 unsigned gen, int *lost, unsigned long long ticks, unsigned *ticket,
                                                                 ThetaPlaneRegions pr, float *scalars_copy) {
    const long long lo = (long long)rank * slice4, hi = lo + slice4 < n4 ? lo + slice4 : n4;
    const long long stride = (long long)gridDim.x * 256;
    // ---- phase 1: this rank's slice -- rank-ordered sum of the pushed rows, TF-Adam, planes, push of the updated slice ----------
    if (lo + (long long)blockIdx.x * 256 < hi) {        // (uniform per workgroup: it owns part of the slice)
        const bool ok = dp_wait_flags(flags_a, world, -1, gen, ticks, lost);
        const float scale = 1.f / (float)world;
        for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) {
            // the local operands first (ordinary loads, in flight while the system-scope row loads below wait for theirs)
            const bool upd = i < nadam4;
            dp_f32x4 mm = {0.f, 0.f, 0.f, 0.f}, vv = mm, th = mm;
            if (upd) { mm = ((dp_f32x4 *)m)[i]; vv = ((dp_f32x4 *)v)[i]; th = ((dp_f32x4 *)theta)[i]; }
            dp_f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (world > 4) {
                for (int r0 = 0; r0 < world; r0 += 8) {   // eight rows per round trip, added in rank order
                    dp_f32x4 s8[8];
                    const float *q[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) q[k] = gslots + ((long long)(r0 + k < world ? r0 + k : 0) * slice4 + (i - lo)) * 4;
                    load_sys16x8(q, s8);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (r0 + k < world) g = (r0 + k == 0) ? s8[k] : g + s8[k];
                }
            } else if (world == 1) {              // (a one-rank group: one row, no padding loads of uncached memory)
                g = load_sys16(gslots + (i - lo) * 4);
            } else {
                dp_f32x4 s4[4];
                const float *q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = gslots + ((long long)(k < world ? k : 0) * slice4 + (i - lo)) * 4;
                load_sys16x4(q[0], q[1], q[2], q[3], s4);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < world) g = (k == 0) ? s4[k] : g + s4[k];
            }
            if (!ok) g = (dp_f32x4){NAN, NAN, NAN, NAN};
            ((dp_f32x4 *)sum_out)[i] = g;
            if (scalars_copy && i >= nadam4 && i < nadam4 + 4) ((dp_f32x4 *)scalars_copy)[i - nadam4] = g;
            dp_f32x4 out = g;
            if (upd) {
                g *= scale;
#pragma unroll
                for (int e = 0; e < 4; ++e) {   // TF-1.x Adam, the same operations as adam1() of cfl_hip.hip
                    mm[e] = fmaf(b1, mm[e], (1.f - b1) * g[e]);
                    vv[e] = fmaf(b2, vv[e], ((1.f - b2) * g[e]) * g[e]);
                    th[e] -= lr_t * mm[e] / (sqrtf(vv[e]) + eps);
                }
                ((dp_f32x4 *)m)[i] = mm;
                ((dp_f32x4 *)v)[i] = vv;
                ((dp_f32x4 *)theta)[i] = th;
                theta_planes_store4(pr, i * 4, th);
                out = th;
            }
            for (int r = 0; r < world; ++r)
                if (r != rank) dp_store_sys16(p.slot[r] + 4 * i, out);
        }
    }
    dp_signal_peers(p, gen, ticket);                    // every workgroup counts; the last one raises this rank's B flags
    // ---- phase 2: the peers' slices, from the local stage buffer -------------------------------------------------------------
    if (world > 1) {
        const bool ok = dp_wait_flags(flags_b, world, rank, gen, ticks, lost);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            if (i >= lo && i < hi) continue;
            dp_f32x4 x = load_sys16(stage + i * 4);
            if (!ok) x = (dp_f32x4){NAN, NAN, NAN, NAN};
            if (i < nadam4) {
                ((dp_f32x4 *)theta)[i] = x;
                theta_planes_store4(pr, i * 4, x);
            } else {
                ((dp_f32x4 *)sum_out)[i] = x;
                if (scalars_copy && i < nadam4 + 4) ((dp_f32x4 *)scalars_copy)[i - nadam4] = x;
            }
        }
    }
}

// ---- exchange memory: fine-grained device allocations shared between the ranks' processes through hipIpc ----------------
// HIP promises that stores of a peer GPU become visible to a RUNNING kernel of the owner only for fine-grained
// (hipDeviceMallocFinegrained) memory; the slots and flags of the exchange are polled by running kernels, so they live
// there.  The caller owns what it allocates here (cfl_dp_free) -- the library keeps no reference.
extern "C" int cfl_dp_alloc(void **ptr, size_t bytes, int32_t fine_grained) {
    if (!ptr || bytes == 0) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_alloc: NULL pointer or zero size");
    void *p = nullptr;
    hipError_t e = fine_grained ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
    if (e != hipSuccess) return cfl_set_err(CFL_E_HIP, "cfl_dp_alloc(%zu bytes, fine_grained=%d): %s", bytes, fine_grained, hipGetErrorString(e));
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return cfl_set_err(CFL_E_HIP, "cfl_dp_alloc: memset: %s", hipGetErrorString(e)); }
    *ptr = p;
    return CFL_OK;
}

extern "C" int cfl_dp_free(void *ptr) {
    if (!ptr) return CFL_OK;
    hipError_t e = hipFree(ptr);
    return e == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_dp_free: %s", hipGetErrorString(e));
}

extern "C" int cfl_dp_ipc_export(void *ptr, void *handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!ptr || !handle64) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_ipc_export: NULL pointer");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return cfl_set_err(CFL_E_HIP, "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
    memcpy(handle64, &h, 64);
    return CFL_OK;
}

extern "C" int cfl_dp_ipc_open(const void *handle64, void **ptr) {
    if (!ptr || !handle64) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_ipc_open: NULL pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return cfl_set_err(CFL_E_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *ptr = p;
    return CFL_OK;
}

extern "C" int cfl_dp_ipc_close(void *ptr) {
    if (!ptr) return CFL_OK;
    hipError_t e = hipIpcCloseMemHandle(ptr);
    return e == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
}

static int dp_fill_peers(DpPeers *p, float *const *slots, uint32_t *const *flags, int world, const char *who) {
    memset(p, 0, sizeof(*p));
    if (world < 1 || world > CFL_DP_MAX_WORLD) return cfl_set_err(CFL_E_SHAPE, "%s: world %d out of range", who, world);
    p->world = world;
    for (int r = 0; r < world; ++r) {
        if (!slots[r] || !flags[r] || ((uintptr_t)slots[r] & 15))
            return cfl_set_err(CFL_E_SHAPE, "%s: peer %d slot / flag NULL or misaligned", who, r);
        p->slot[r] = slots[r];
        p->flag[r] = flags[r];
    }
    return CFL_OK;
}

// Workgroups of the polling kernels.  Every block of cfl_dp_rs_adam / cfl_dp_rs_gather polls flags before it works, so ranks that
// SHARE a GPU (the functional tests: 2 or 8 processes on one device) must not fill it with waiting blocks: CFL_DP_MAX_BLOCKS caps
// the launches (the tests set 64).  One rank per GPU (the default, 512): one float4 per thread wherever the slice allows -- a
// thread's loads are system-scope round trips, and every extra trip of the loop is one more of them in series.
static int dp_max_blocks(void) {
    const char *v = getenv("CFL_DP_MAX_BLOCKS");
    const int n = v ? atoi(v) : 0;
    return n > 0 ? n : 512;
}

static unsigned long long dp_ticks(double timeout_s) {
    if (!(timeout_s > 0.0)) timeout_s = 30.0;
    return (unsigned long long)(timeout_s * 1e8);   // s_memrealtime counts at 100 MHz
}

static int dp_check_sizes(const char *who, int64_t n, int64_t n_adam, int64_t slice, int world, int rank) {
    if (n <= 0 || n % 4 || n_adam < 0 || n_adam % 4 || n_adam > n)
        return cfl_set_err(CFL_E_SHAPE, "%s: n=%lld n_adam=%lld must be multiples of 4, n_adam <= n", who, (long long)n, (long long)n_adam);
    if (slice <= 0 || slice % 4 || slice * world < n)
        return cfl_set_err(CFL_E_SHAPE, "%s: slice=%lld must be a multiple of 4 with slice * world >= n", who, (long long)slice);
    if (rank < 0 || rank >= world) return cfl_set_err(CFL_E_SHAPE, "%s: rank %d of %d", who, rank, world);
    return CFL_OK;
}

extern "C" int cfl_dp_rs_push(const float *src, int64_t n, int64_t slice, float *const *peer_rows,
                              uint32_t *const *peer_flags, int32_t world, uint32_t generation, uint32_t *ticket,
                              cfl_stream_t stream) {
    if (!src || !peer_rows || !peer_flags || !ticket) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_push: NULL pointer");
    if (((uintptr_t)src & 15)) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_push: src must be 16-byte aligned");
    DpPeers p;
    int rc = dp_fill_peers(&p, peer_rows, peer_flags, world, "cfl_dp_rs_push");
    if (rc) return rc;
    rc = dp_check_sizes("cfl_dp_rs_push", n, 0, slice, world, 0);
    if (rc) return rc;
    const long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256) blocks = 256;
    {
        DpProf prof(stream);
        hipLaunchKernelGGL(cfl_dp_rs_push_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, n4, (long long)(slice / 4), p,
                           generation, ticket);
    }
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_dp_rs_push launch failed");
}

extern "C" int cfl_dp_rs_adam(float *theta, float *m, float *v, const float *gslots, const uint32_t *flags, int32_t world,
                              int32_t rank, int64_t n, int64_t n_adam, int64_t slice, float *sum_out,
                              float *const *peer_stage, uint32_t *const *peer_flags, float lr_t, float beta1, float beta2,
                              float eps, uint32_t generation, int32_t *lost, double timeout_s, uint32_t *ticket,
                              cfl_stream_t stream) {
    if (!theta || !m || !v || !gslots || !flags || !sum_out || !peer_stage || !peer_flags || !lost || !ticket)
        return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_adam: NULL pointer");
    if (((uintptr_t)theta | (uintptr_t)m | (uintptr_t)v | (uintptr_t)gslots | (uintptr_t)sum_out) & 15)
        return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_adam: theta / m / v / slots / sum_out must be 16-byte aligned");
    DpPeers p;
    int rc = dp_fill_peers(&p, peer_stage, peer_flags, world, "cfl_dp_rs_adam");
    if (rc) return rc;
    rc = dp_check_sizes("cfl_dp_rs_adam", n, n_adam, slice, world, rank);
    if (rc) return rc;
    int blocks = (int)((slice / 4 + 255) / 256);
    if (blocks > dp_max_blocks()) blocks = dp_max_blocks();
    {
        DpProf prof(stream);
        hipLaunchKernelGGL(cfl_dp_rs_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, m, v, gslots, flags,
                           world, rank, (long long)(n / 4), (long long)(n_adam / 4), (long long)(slice / 4), sum_out, p, lr_t, beta1,
                           beta2, eps, generation, (int *)lost, dp_ticks(timeout_s), ticket);
    }
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_dp_rs_adam launch failed");
}

static int dp_gather_launch(const CflShape *shape, float *theta, float *sum_out, const float *stage,
                            const uint32_t *flags, int32_t world, int32_t rank, int64_t n, int64_t n_adam,
                            int64_t slice, uint32_t generation, int32_t *lost, double timeout_s,
                            CflThetaPlanes *planes, float *scalars_copy, cfl_stream_t stream) {
    ThetaPlaneRegions pr;
    memset(&pr, 0, sizeof(pr));
    if (planes) {
        if (!planes->buf || ((uintptr_t)planes->buf & 15))
            return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_gather_planes: theta planes buffer NULL or misaligned");
        CflLayout lay;
        int rc = theta_plane_regions(shape, planes->buf, &pr);
        if (!rc) rc = cfl_layout(shape, &lay);
        if (rc) return rc;
        if (lay.total != n_adam)
            return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_gather_planes: n_adam=%lld is not the parameter count %lld of the shape",
                               (long long)n_adam, (long long)lay.total);
    }
    if (!theta || !sum_out || !stage || !flags || !lost) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_gather: NULL pointer");
    if (((uintptr_t)theta | (uintptr_t)sum_out | (uintptr_t)stage) & 15)
        return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_gather: theta / sum_out / stage must be 16-byte aligned");
    if (world < 1 || world > CFL_DP_MAX_WORLD) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_rs_gather: world %d out of range", world);
    int rc = dp_check_sizes("cfl_dp_rs_gather", n, n_adam, slice, world, rank);
    if (rc) return rc;
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > dp_max_blocks()) blocks = dp_max_blocks();
    {
        DpProf prof(stream);
        hipLaunchKernelGGL(cfl_dp_rs_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, sum_out, stage, flags,
                           world, rank, (long long)(n / 4), (long long)(n_adam / 4), (long long)(slice / 4), generation, (int *)lost,
                           dp_ticks(timeout_s), pr, scalars_copy);
    }
    if (hipGetLastError() != hipSuccess) return cfl_set_err(CFL_E_HIP, "cfl_dp_rs_gather launch failed");
    if (planes) planes->valid = 1;
    return CFL_OK;
}

extern "C" int cfl_dp_rs_gather_planes(const CflShape *shape, float *theta, float *sum_out, const float *stage,
                                       const uint32_t *flags, int32_t world, int32_t rank, int64_t n, int64_t n_adam,
                                       int64_t slice, uint32_t generation, int32_t *lost, double timeout_s,
                                       CflThetaPlanes *planes, cfl_stream_t stream) {
    return dp_gather_launch(shape, theta, sum_out, stage, flags, world, rank, n, n_adam, slice, generation, lost, timeout_s,
                            planes, nullptr, stream);
}

// ---- ABI 6: the exchange + update of one step from a CflDpExchange ------------------------------------------------------
static int dp_check_exchange(const CflDpExchange *ex, const char *who) {
    if (!ex) return cfl_set_err(CFL_E_SHAPE, "%s: exchange is NULL", who);
    if (ex->world < 1 || ex->world > CFL_DP_MAX_WORLD || ex->rank < 0 || ex->rank >= ex->world)
        return cfl_set_err(CFL_E_SHAPE, "%s: rank %d of %d", who, ex->rank, ex->world);
    if (!ex->tickets || !ex->lost) return cfl_set_err(CFL_E_SHAPE, "%s: tickets / lost is NULL", who);
    for (int p = 0; p < 2; ++p)
        if (!ex->slots[p] || !ex->stage[p] || !ex->flags_a[p] || !ex->flags_b[p])
            return cfl_set_err(CFL_E_SHAPE, "%s: local slots / stage / flags of parity %d are NULL", who, p);
    return dp_check_sizes(who, ex->n, ex->n_adam, ex->slice, ex->world, ex->rank);
}

extern "C" uint32_t cfl_dp_generation(uint64_t step) {   // never 0 (the flag words start at 0)
    const uint32_t g = (uint32_t)((step + 1) & 0xffffffffu);
    return g ? g : 1u;
}

extern "C" int cfl_dp_exchange_step(const CflShape *shape, CflDpExchange *ex, int32_t pushed, float *theta, float *m, float *v,
                                    float *gradbuf, float lr_t, float beta1, float beta2, float eps, CflThetaPlanes *planes,
                                    float *scalars_copy, cfl_stream_t stream) {
    int rc = dp_check_exchange(ex, "cfl_dp_exchange_step");
    if (rc) return rc;
    const int par = (int)(ex->step & 1);
    const uint32_t gen = cfl_dp_generation(ex->step);
    if (!pushed) {
        rc = cfl_dp_rs_push(gradbuf, ex->n, ex->slice, ex->peer_rows[par], ex->peer_flag_a[par], ex->world, gen, ex->tickets, stream);
        if (rc) return rc;
    }
    const char *split = getenv("CFL_DP_SPLIT_ADAM");
    if (!(split && atoi(split) > 0)) {
        // default: the sharded Adam and the all-gather in ONE launch (cfl_dp_adam_gather_kernel)
        ThetaPlaneRegions pr;
        memset(&pr, 0, sizeof(pr));
        if (planes) {
            if (!planes->buf || ((uintptr_t)planes->buf & 15)) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_exchange_step: theta planes buffer NULL or misaligned");
            rc = theta_plane_regions(shape, planes->buf, &pr);
            if (rc) return rc;
        }
        if (!theta || !m || !v || !gradbuf) return cfl_set_err(CFL_E_SHAPE, "cfl_dp_exchange_step: NULL pointer");
        if (((uintptr_t)theta | (uintptr_t)m | (uintptr_t)v | (uintptr_t)gradbuf) & 15)
            return cfl_set_err(CFL_E_SHAPE, "cfl_dp_exchange_step: theta / m / v / gradbuf must be 16-byte aligned");
        DpPeers p;
        rc = dp_fill_peers(&p, ex->peer_stage[par], ex->peer_flag_b[par], ex->world, "cfl_dp_exchange_step");
        if (rc) return rc;
        int blocks = (int)((ex->n / 4 + 255) / 256);
        if (blocks > dp_max_blocks()) blocks = dp_max_blocks();
        {
            DpProf prof(stream);
            hipLaunchKernelGGL(cfl_dp_adam_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, m, v, ex->slots[par],
                               ex->flags_a[par], ex->flags_b[par], ex->stage[par], ex->world, ex->rank, (long long)(ex->n / 4),
                               (long long)(ex->n_adam / 4), (long long)(ex->slice / 4), gradbuf, p, lr_t, beta1, beta2, eps, gen,
                               (int *)ex->lost, dp_ticks(ex->timeout_s), ex->tickets + 1, pr, scalars_copy);
        }
        if (hipGetLastError() != hipSuccess) return cfl_set_err(CFL_E_HIP, "cfl_dp_adam_gather launch failed");
        if (planes) planes->valid = 1;
        ex->step += 1;
        return CFL_OK;
    }
    rc = cfl_dp_rs_adam(theta, m, v, ex->slots[par], ex->flags_a[par], ex->world, ex->rank, ex->n, ex->n_adam, ex->slice, gradbuf,
                        ex->peer_stage[par], ex->peer_flag_b[par], lr_t, beta1, beta2, eps, gen, ex->lost, ex->timeout_s,
                        ex->tickets + 1, stream);
    if (rc) return rc;
    rc = dp_gather_launch(shape, theta, gradbuf, ex->stage[par], ex->flags_b[par], ex->world, ex->rank, ex->n, ex->n_adam, ex->slice,
                          gen, ex->lost, ex->timeout_s, planes, scalars_copy, stream);
    if (rc) return rc;
    ex->step += 1;
    return CFL_OK;
}

extern "C" int cfl_dp_rs_gather(float *theta, float *sum_out, const float *stage, const uint32_t *flags, int32_t world,
                                int32_t rank, int64_t n, int64_t n_adam, int64_t slice, uint32_t generation, int32_t *lost,
                                double timeout_s, cfl_stream_t stream) {
    return cfl_dp_rs_gather_planes(nullptr, theta, sum_out, stage, flags, world, rank, n, n_adam, slice, generation, lost,
                                   timeout_s, nullptr, stream);
}
