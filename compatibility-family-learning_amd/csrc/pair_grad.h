// pair_grad.h -- weight-gradient kernels of the pair step with the fused gradient + TF-Adam + planes tail, and the row-reduction blocks
// Part of the pair-step translation unit: included by cfl_hip.hip (and nothing else) behind the common device helpers; see the
// header comment of cfl_hip.hip for the launch structure and the fragment-major layouts, DESIGN.md section 4 for what runs when.
#pragma once

// ---------------------------------------------------------------------------
// grad: Wpart[p] (Wf layout) = sum_{r in range p} X[r][d] * dY[r][c]
//   workgroup = 4 waves, one 64-d tile and one row range; the waves split the range
//   in chunks of 64 rows.
//   A fragment  lane(i,kq), load (rg,j) : float4 X[p0+16rg+4kq+j][dbase+4i .. +3]
//                (each instruction: 4 rows x 256 contiguous bytes; 16 loads issued up front)
//   B fragment  one contiguous 1 KiB block of dYf per (nt, rg)
//   MFMA (j,t): A elem = xa[rg][j][t]  (M row i <-> d = dbase+4i+t, k = kq <-> row 16rg+4kq+j)
//               B elem = dy[nt][j]
//   z-slice 0 of the launch: row reductions (column sums of dYf etc.).
// ---------------------------------------------------------------------------
struct GradJob {
    int side;              // 0 = src rows, 1 = dst rows (GradArgs::rows)
    const float *dyf;      // dYf tile base: blocks [(nt)*RG + rg]
    float *wpart;          // Wf tile base inside slab 0; slabs are pstride apart
    long long pstride;     // floats between row-range slabs (npad * D)
    int nt;
};

// columns of the per-row loss-quantity tile written by mid and summed over rows by grad_red_block
enum {
    P_BCE_POS = 0, P_BCE_NEG, P_OK_POS, P_OK_NEG, P_D_POS, P_D_NEG, P_O_POS, P_O_NEG,
    P_DTHR, P_HINGE_NEG, P_SQRT_POS, P_SQRT_NEG, P_NROWQ = 12
};

// Row reductions that ride in the grad launch (z-slice 0).
//   kind 0: column sums of a fragment-major buffer: one job per 16-column tile
//   kind 1: gate head  dVm[l][k] = sum_r ya[r][l] * du[r][k]  (row-major buffers), one job per l
//   kind 2: column sums of a row-major buffer [Rpad][lda], columns 0..K-1, one job
struct RedRange {
    const float *A, *B;
    int kind, count, out_off, lda, ldb, K, kpad;
};
#define CFL_MAX_RED 8

// Fused tail of the weight-gradient launch (plan.fused: pcd with one encoder -- plain `Dist` heads and weight-normalised
// `CFL` heads --, any row split P <= 8; CFL_DEBUG_NOFUSE=1 in the environment restores the separate finalize launch):
// the launch itself
// turns the per-range partial gradients into the flat gradient and applies TF-Adam, so the step needs no finalize
// launch and no round trip of P gradient slabs through HBM.
//   * a (64-d tile, column job) is produced by P workgroups (row ranges).  The first P-1 row ranges
//     publish their partial tile into their slab -- `sc1` (write-through) stores, every storing wave drains with
//     s_waitcnt vmcnt(0), workgroup barrier, then ONE lane adds 1 to the tile's arrival counter (agent-scope
//     atomic) -- and leave; the workgroup of the last row range polls that counter with `sc1` loads (one lane),
//     barrier, reads the published tiles with `sc1` loads and runs the epilogue.  The memory side is the last-arriver hand-off of
//     MI355X_MICROARCH.md ("hand-offs measured with sc1 loads in place of the acquire", first row): no
//     agent-scope fence on either side.  The tiles are summed in the fixed order of the row ranges, so the result
//     is bit-reproducible, and bit-identical to the finalize kernel).
//   * the row-reduction blocks (z-slice 0) own whole columns, so they finish the bias / threshold entries and the
//     step's scalars themselves.
// Tickets and flags live in the workspace and are zeroed by the mid launch of the same step.
struct GradFuse {
    int on;
    int *ticket, *flag;          // [jobs * d tiles]
    const float *theta;
    float *grad;                 // flat gradient, layout of theta
    float *theta_out, *m, *v;    // fused TF-Adam (m == nullptr: gradient only)
    unsigned short *planes;      // kept bf16 planes of theta (CflThetaPlanes::buf) or nullptr: the tile finishers write the
                                 // planes of the weights they update (ushort index 3 * theta offset of the Wf block + ...)
    float lr_t, b1, b2, eps, in_mul, reg_const;
    long long w_off[CFL_MAX_JOBS];   // theta offset of the job's Wf tile base
    // row-reduction side: red range k (kind 0) feeds the bias array at red_b[k] (npad red_npad[k], n red_n[k])
    long long red_b[CFL_MAX_RED];
    int red_n[CFL_MAX_RED], red_npad[CFL_MAX_RED];
    // weight-normalised heads: red range k (column sums of dy * xv: c_j = sum_d V_dj (x^T dy)_dj) feeds the gain array
    // at red_g[k]; the W tiles need c_j too: the range's blocks publish it (sc1) and bump red_done
    int wn;
    long long red_g[CFL_MAX_RED];
    const float *red_n2[CFL_MAX_RED];
    int *red_done, red_expect;
    const float *wn_g[CFL_MAX_JOBS], *wn_n2[CFL_MAX_JOBS], *wn_cw[CFL_MAX_JOBS];   // at the job's first column
    int wn_n[CFL_MAX_JOBS];                                                        // valid columns from there
    // siamese (both sides project through ONE head): the tile of column job j of side 1 also receives the P row ranges
    // of job j - pair_jobs of side 0.  pair_jobs > 0: jobs [0, pair_jobs) only publish, job j >= pair_jobs finishes
    // slot j - pair_jobs after 2P - 1 arrivals, summing side 0's slabs first (the finalize kernel's order).
    // (scalars only: one more dynamically indexed array in this argument block and hipcc copies the whole block to
    // scratch -- 2.4 KB per lane, the weight-gradient launch 2.7x slower)
    int pair_jobs;
    int spin_limit;         // < 0: a hand-off is declared lost at once (the failure test, CFL_DEBUG_SPIN_LIMIT=-1); else unused
    unsigned long long spin_ticks;   // s_memrealtime ticks (100 MHz) before an in-launch hand-off is declared lost (CFL_HANDOFF_TIMEOUT_S)
    long long pair_delta;   // floats from side 0's slab array to side 1's (same column chunk, same row range)
    // monomer gate head V[L][kpad] (+ gains) of the SOURCE encoder: finished by the kind-1 / kind-2 reduction blocks
    long long mono_w, mono_g;   // theta offsets (-1: none)
    const float *mono_n2, *mono_gcopy, *mono_duc;   // weight-norm: squared norms, gain snapshot, [Rpad][kpad] rows of du * u
    int mono_L, mono_K, mono_kpad, mono_reg;
    // regions no side projects through (the unused heads of directed encoders): gradient = L2 term only, Adam applied
    // as the finalize kernel does; handled by element-wise blocks of kind 3
    int norph;
    long long orph_off[8], orph_cnt[8];
    int orph_reg[8];
    long long thr_off;
    // scalars
    const float *regpart;
    int nregblocks, B, use_threshold;
    float pos_weight, caffe_margin, lambda_m;
    float *scalars;
    float *scalars2;             // a second destination of the step's scalars (nullptr: none)
    const float *thr_copy;
    // DATA-PARALLEL one-shot exchange with the reduce-scatter FUSED into this launch (round 6; csrc/cfl_dp.hip): every finished
    // entry of [gradient | scalars] goes straight into the OWNER rank's slot array instead of the local flat buffer -- element
    // `off` belongs to rank off / dp_slice, and dp_rows[r] (a table in DEVICE memory: a dynamically indexed array inside this
    // argument block would send the whole block to scratch) is where THIS rank's row of rank r's slot array is mapped.  Every
    // workgroup of the launch ends with grad_dp_block_done(): its stores drained, one count on dp_ticket; the last one raises
    // this rank's arrival flag in every peer (dp_flags[r], generation dp_gen).  dp_slice == 0: off (every other caller).
    float *const *dp_rows;
    unsigned *const *dp_flags;
    long long dp_slice;
    int dp_world;
    unsigned dp_gen;
    unsigned *dp_ticket;
};

// where gradient entry `off` (and, past the parameters, the step's scalars) lives in its OWNER's slot array (DP kernels only)
__device__ __forceinline__ float *dp_grad_ptr(const GradFuse &f, long long off) {
    const unsigned s = (unsigned)off / (unsigned)f.dp_slice;       // (off < 2^31: make_plan rejects larger layouts)
    return f.dp_rows[s] + (off - (long long)s * f.dp_slice);
}
// Stores into the peers' slots are SYSTEM-SCOPE WRITE-THROUGH stores (sc0 sc1: the slots are fine-grained memory of this or another
// GPU, never held dirty in an L2), drained by every storing wave with s_waitcnt vmcnt(0) before the workgroup's arrival count --
// the hand-off form this library uses inside a launch (grad_fused_tail), one scope further out.  No __threadfence_system():
// on gfx950 that is a write-back of the whole L2 per calling wave; 257 workgroups doing it at the end of the launch cost the
// step tens of microseconds (measured on a one-rank group, round 6).
// (one asm statement per store, with the wait states a VALU write to the data registers of a > 64-bit store needs behind it: the
// compiler's hazard recogniser does not see inside inline asm)
__device__ __forceinline__ void dp_store16(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dp_store4(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// end of EVERY workgroup of a weight-gradient launch with the fused push
__device__ __forceinline__ void grad_dp_block_done(const GradFuse &f) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pushes have been acknowledged by the memory system
    __syncthreads();
    __shared__ unsigned dp_last;
    if (threadIdx.x == 0)
        dp_last = __hip_atomic_fetch_add(f.dp_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y * gridDim.z - 1u ? 1u : 0u;
    __syncthreads();
    if (dp_last) {
        // every workgroup of the launch drained its stores before it counted: raise this rank's arrival flag in every peer
        if ((int)threadIdx.x < f.dp_world) __hip_atomic_store(f.dp_flags[threadIdx.x], f.dp_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *f.dp_ticket = 0;   // the next launch on this stream starts from zero
    }
}

// Bounded wait of an in-launch hand-off: counter `ctr` reaches `expect`, or the wall clock runs out (-> false: the caller sets the
// sticky error word and poisons its output).  Bounded by TIME, not by polls (round 6): the partners are workgroups dispatched
// BEFORE the waiter, which never wait themselves, so on a GPU this process has to itself the wait is microseconds -- but on a
// SHARED GPU (several processes: the multi-rank tests, a neighbour job) the queue is preempted and restored piecemeal, a restored
// waiter can spin while its partner is still saved, and 2^22 polls (~100 ms) ran out in 6 of 8 four-rank runs
// (profiles/r06_handoff_timeout.txt).  The clock is read every 256 polls.
__device__ __forceinline__ int handoff_wait(const int *ctr, int expect, const GradFuse &f) {
    if (f.spin_limit < 0) return 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > f.spin_ticks) return 0;
    }
    return 1;
}

struct GradArgs {
    GradJob job[CFL_MAX_JOBS];
    RowSrc rows[2];
    int B, R, Rpad, D, P;
    int tps;  // > 0: 64-d tiles per projection slice, tiles are dealt to XCDs by slice (cfl_xcd_aligned)
    NormDev norm;
    RedRange red[CFL_MAX_RED];
    int nred, red_total;
    float *colsum;
    GradFuse fuse;
};

// d tile of this workgroup.  XCD = blockIdx.x mod 8 (gridDim.x is a multiple of 8 when tps > 0); tile dt
// belongs to projection slice dt / tps, which the projection launch ran on XCD (dt / tps) mod 8.
__device__ __forceinline__ int grad_dtile(int tps) {
    if (tps <= 0) return blockIdx.x;
    const int k = blockIdx.x & 7, j = blockIdx.x >> 3;
    return ((j / tps) * 8 + k) * tps + j % tps;
}

// A lane's four float4 of a tile (64 floats apart) as write-through stores / L1-bypassing loads -- the `sc1` forms of
// the hand-off table.  Each direction is ONE asm statement that ends with its own s_waitcnt: the compiler neither
// tracks the completion of memory instructions inside inline asm nor applies its hazard rules to them (a VALU write
// to the data registers of a > 64-bit store needs wait states after the store; a register filled by an asm load
// may be copied or consumed by compiler-scheduled code before a separate wait statement).  With separate statements
// both happened: the first two dwords of a published float4 were overwritten by the address arithmetic of the next
// store (found with forced row splits P = 4, 8 at small batches; tests/test_hip_parity.py).
__device__ __forceinline__ void store4_sc1_wait(float *p, f32x4 v0, f32x4 v1, f32x4 v2, f32x4 v3) {
    asm volatile(
        "global_store_dwordx4 %0, %1, off sc1\n\t"
        "global_store_dwordx4 %0, %2, off offset:256 sc1\n\t"
        "global_store_dwordx4 %0, %3, off offset:512 sc1\n\t"
        "global_store_dwordx4 %0, %4, off offset:768 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        :: "v"(p), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
}
__device__ __forceinline__ void load4_sc1_wait(const float *p, f32x4 (&o)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %4, off offset:256 sc1\n\t"
        "global_load_dwordx4 %2, %4, off offset:512 sc1\n\t"
        "global_load_dwordx4 %3, %4, off offset:768 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(p) : "memory");
}

__device__ __forceinline__ void load4x3_sc1_wait(const float *p0, const float *p1, const float *p2, f32x4 (&o)[12]) {
    asm volatile(
        "global_load_dwordx4 %0, %12, off sc1\n\t"
        "global_load_dwordx4 %1, %12, off offset:256 sc1\n\t"
        "global_load_dwordx4 %2, %12, off offset:512 sc1\n\t"
        "global_load_dwordx4 %3, %12, off offset:768 sc1\n\t"
        "global_load_dwordx4 %4, %13, off sc1\n\t"
        "global_load_dwordx4 %5, %13, off offset:256 sc1\n\t"
        "global_load_dwordx4 %6, %13, off offset:512 sc1\n\t"
        "global_load_dwordx4 %7, %13, off offset:768 sc1\n\t"
        "global_load_dwordx4 %8, %14, off sc1\n\t"
        "global_load_dwordx4 %9, %14, off offset:256 sc1\n\t"
        "global_load_dwordx4 %10, %14, off offset:512 sc1\n\t"
        "global_load_dwordx4 %11, %14, off offset:768 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
          "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
        : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

// half tiles (cfl_grad_x3_half_kernel): two float4 per lane, 64 floats apart
__device__ __forceinline__ void store2_sc1_wait(float *p, f32x4 v0, f32x4 v1) {
    asm volatile(
        "global_store_dwordx4 %0, %1, off sc1\n\t"
        "global_store_dwordx4 %0, %2, off offset:256 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        :: "v"(p), "v"(v0), "v"(v1) : "memory");
}
__device__ __forceinline__ void load2x3_sc1_wait(const float *p0, const float *p1, const float *p2, f32x4 (&o)[6]) {
    asm volatile(
        "global_load_dwordx4 %0, %6, off sc1\n\t"
        "global_load_dwordx4 %1, %6, off offset:256 sc1\n\t"
        "global_load_dwordx4 %2, %7, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off offset:256 sc1\n\t"
        "global_load_dwordx4 %4, %8, off sc1\n\t"
        "global_load_dwordx4 %5, %8, off offset:256 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
        : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

// TF-1.x Adam on one parameter (tensorflow/core/kernels/training_ops: the hyper-parameters are
// float32 scalars and (1 - beta) is formed in float32):  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// theta -= lr_t m / (sqrt(v) + eps).  Explicit fma's: every kernel that applies Adam (finalize, the fused tail of
// the weight-gradient launch, cfl_adam_kernel) rounds identically, whatever the compiler contracts around it.
__device__ __forceinline__ void adam1(float &th, float &mm, float &vv, float g, float lr_t, float b1, float b2,
                                      float eps) {
    mm = fmaf(b1, mm, (1.f - b1) * g);
    vv = fmaf(b2, vv, ((1.f - b2) * g) * g);
    th -= lr_t * mm / (sqrtf(vv) + eps);
}
__device__ __forceinline__ void adam4(f32x4 &th, f32x4 &mm, f32x4 &vv, const f32x4 g, float lr_t, float b1, float b2,
                                      float eps) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = th[e], m = mm[e], v = vv[e];
        adam1(t, m, v, g[e], lr_t, b1, b2, eps);
        th[e] = t; mm[e] = m; vv[e] = v;
    }
}

// gradient entry -> flat gradient (+ L2 term) -> optional TF-Adam, 4 consecutive parameters at `off`
template <bool DP = false>   // DP: the entry goes to its owner rank's slot (fused push of the one-shot exchange) instead of the flat gradient
__device__ __forceinline__ void fuse_apply(const GradFuse &f, long long off, f32x4 gr, f32x4 &th, f32x4 mm, f32x4 vv) {   // th: updated in place (the planes are split from it)
    if (f.reg_const != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gr[e] = fmaf(f.reg_const, th[e], gr[e]);
    }
    if (DP) dp_store16(dp_grad_ptr(f, off), gr);
    else *(f32x4 *)(f.grad + off) = gr;
    if (f.m) {
        adam4(th, mm, vv, gr, f.lr_t, f.b1, f.b2, f.eps);
        *(f32x4 *)(f.m + off) = mm;
        *(f32x4 *)(f.v + off) = vv;
        *(f32x4 *)(f.theta_out + off) = th;
    }
}

// Tail of a weight-gradient workgroup in fused mode.  `sum` = this workgroup's partial tile in the C/D mapping of
// the bodies below (valid in waves < NT; wave = nt); tile_off = float offset of the lane's first float4 inside the
// head's Wf array (the other three are 64 floats apart).
template <int NT>
__device__ __forceinline__ void grad_fused_tail(const GradArgs &a, int job, int P, int p, int wave,
                                                f32x4 (&sum)[4], size_t tile_off, float *slab0, long long pstride,
                                                int *lds_i) {
    const GradFuse &f = a.fuse;
    const bool paired = f.pair_jobs > 0;
    const bool side0 = paired && job < f.pair_jobs;
    const int slot = (paired && !side0 ? job - f.pair_jobs : job) * gridDim.x + blockIdx.x;
    const int expect = paired ? 2 * P - 1 : P - 1;
    // Roles are static: the workgroup of the LAST row range finishes the tile, the others publish.  (A ticket --
    // "whoever arrives last finishes" -- costs an atomic round trip on every workgroup's critical path, ~1 us, and buys
    // nothing: the finisher waits for the publishers' data either way.  No deadlock: a finisher only waits for
    // workgroups with a smaller linear id, which were dispatched before it and run to completion on their own.)
    if (p < P - 1 || side0) {
        // not the last of the (2) P row ranges: publish the partial tile into slab p and leave
        if (wave < NT) {
            f32x4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f32x4){sum[0][e], sum[1][e], sum[2][e], sum[3][e]};
            store4_sc1_wait(slab0 + (size_t)p * pstride + tile_off, v[0], v[1], v[2], v[3]);   // written through
        }
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(f.flag + slot, 1);                 // agent-scope arrival count
        return;
    }
    // the finisher: parameters first (they do not depend on the partners), then the published tiles
    const long long base = f.w_off[job] + (long long)tile_off;
    f32x4 th[4], mm[4], vv[4];
    if (wave < NT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            th[e] = *(const f32x4 *)(f.theta + base + e * 64);
            if (f.m) {
                mm[e] = *(const f32x4 *)(f.m + base + e * 64);
                vv[e] = *(const f32x4 *)(f.v + base + e * 64);
            }
        }
    }
    // weight-norm: per-column gain snapshot and squared norm (workspace, written by the projection launch)
    const int wcol = (wave < NT ? wave : 0) * 16 + (threadIdx.x & 15);
    float wg = 1.f, wn2 = 1.f;
    if (f.wn) { wg = f.wn_g[job][wcol]; wn2 = f.wn_n2[job][wcol]; }
    bool lost = false;   // a partner never arrived (bounded spin): poison instead of hanging or using stale tiles
    if (expect > 0 || f.wn) {
        if (threadIdx.x == 0) {
            // Bounded (handoff_wait: wall clock).  The waits are for workgroups dispatched BEFORE this one (smaller linear id),
            // which never wait themselves, so a time-out means the dispatch-order assumption or the visibility protocol failed.
            int ok = f.spin_limit < 0 ? 0 : 1;
            if (expect > 0 && ok) ok = handoff_wait(f.flag + slot, expect, f);
            if (f.wn && ok) ok = handoff_wait(f.red_done, f.red_expect, f);   // the c_j column sums of this launch's reduction blocks (dispatched first, short)
            lds_i[0] = ok;
            if (!ok) f.scalars[CFL_S_ERROR] = 1.f;   // sticky error word: the host raises at its next read-back
        }
        __syncthreads();
        lost = lds_i[0] == 0;
    }
    if (wave < NT) {
        // sum over the row ranges in the fixed order 0 .. P-1 (own registers at position p): the result does not
        // depend on which workgroup arrived last, and equals the finalize kernel's slab sum bit for bit
        f32x4 g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // published tiles in the finalize kernel's order: side 0's P row ranges first (siamese), then row ranges
        // 0 .. P-2 of this side; the finisher's own registers (row range P-1) come last.  Three tiles (12 loads) are
        // in flight per round trip -- one at a time, config 3 (three tiles) and config 4 (P = 4) paid three serial
        // misses to memory here
        const int npair = paired ? P : 0, nparts = npair + P - 1;
        const float *own = slab0 + tile_off;
        auto part_ptr = [&](int k) {
            k = k < nparts ? k : nparts - 1;
            return k < npair ? own - f.pair_delta + (size_t)k * pstride : own + (size_t)(k - npair) * pstride;
        };
        for (int k = 0; k < nparts; k += 3) {
            f32x4 part[12];
            load4x3_sc1_wait(part_ptr(k), part_ptr(k + 1), part_ptr(k + 2), part);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += part[e];
            if (k + 1 < nparts) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += part[4 + e];
            }
            if (k + 2 < nparts) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += part[8 + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += (f32x4){sum[0][e], sum[1][e], sum[2][e], sum[3][e]};
        if (lost) {   // loud, not silent: NaN gradient (and parameters) for this tile -> NaN loss at the next read-back
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (f32x4){NAN, NAN, NAN, NAN};
        }
        if (f.wn) {
            // dV = (g/n) in_mul X^T dy - (g c / n^3) V   (cfl/layers.py:80-90 differentiated; same operations in the
            // same order as the RK_W branch of the finalize kernel)
            // (siamese: the dual reduction range published c_j over both sides, side 0 first, at side 1's slot)
            const float cw = __hip_atomic_load(f.wn_cw[job] + wcol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool colok = wcol < f.wn_n[job];
            const float n = sqrtf(wn2);
            const float s1 = (colok && wn2 > 0.f) ? f.in_mul * wg / n : 0.f;
            const float s2 = (colok && wn2 > 0.f) ? wg * cw / (wn2 * n) : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 gr = g[e] * s1;
#pragma unroll
                for (int i = 0; i < 4; ++i) gr[i] = fmaf(-s2, th[e][i], gr[i]);
                fuse_apply(f, base + e * 64, gr, th[e], mm[e], vv[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) fuse_apply(f, base + e * 64, g[e] * f.in_mul, th[e], mm[e], vv[e]);
        }
        if (f.planes && f.m) {
            // kept bf16 planes of the updated weights (cfl_wplanes_kernel's layout, bit for bit): this lane holds d =
            // dbase + 16 kq + 4 e + e' of column i16, i.e. the two 8-value groups c = 0, 1 (e = 2c, 2c + 1) of 32-d quarter
            // tq = (Wf row group) / 2, fragment lane (2 (g & 1) + c) * 16 + i16
            const int lane = threadIdx.x & 63, i16 = lane & 15;
            const int G = a.D >> 4, Q = a.D >> 5;
            const int gg = (int)((tile_off >> 8) % (size_t)G), ntw = (int)((tile_off >> 8) / (size_t)G);
            unsigned short *pb = f.planes + 3 * f.w_off[job] + ((size_t)(ntw * Q + (gg >> 1)) * 3) * 512;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float vals[8] = {th[2 * c][0], th[2 * c][1], th[2 * c][2], th[2 * c][3],
                                 th[2 * c + 1][0], th[2 * c + 1][1], th[2 * c + 1][2], th[2 * c + 1][3]};
                bf16x8 fr[3];
                split_frag_rne(vals, fr);
                unsigned short *dst = pb + ((2 * (gg & 1) + c) * 16 + i16) * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *(bf16x8 *)(dst + pl * 512) = fr[pl];
            }
        }
    }
}

// the same reduction, handing the tile to the fused tail instead of storing a slab
#define CFL_GRAD_FUSED_EPILOGUE()                                                                     \
    if (a.fuse.on) {                                                                                  \
        f32x4 sum[4];                                                                                 \
        const int ntw = wave < NT ? wave : 0;                                                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                               \
            sum[t] = lds[((0 * NT + ntw) * 4 + t) * 64 + lane];                                       \
            _Pragma("unroll") for (int w = 1; w < 4; ++w) sum[t] += lds[((w * NT + ntw) * 4 + t) * 64 + lane]; \
        }                                                                                             \
        __syncthreads();                                                                              \
        grad_fused_tail<NT>(a, (int)blockIdx.z - 1, a.P, p, wave, sum,                                \
                            ((size_t)ntw * G + (dbase >> 4) + kq) * 256 + i16 * 4, jb.wpart, jb.pstride,  \
                            (int *)lds);                                                              \
        return;                                                                                       \
    }

template <int NT>
__device__ __forceinline__ void grad_body(const GradJob &jb, const GradArgs &a, f32x4 *lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform
    const int i16 = lane & 15, kq = lane >> 4;
    const int dbase = grad_dtile(a.tps) * 64;
    const int p = blockIdx.y;
    const int RG = a.Rpad >> 4, G = a.D >> 4;
    const int rows_wg = a.Rpad / a.P, rows_w = rows_wg >> 2;  // multiple of 64
    const int rbeg = p * rows_wg + wave * rows_w, rend = rbeg + rows_w;
    const int r64 = (a.R + 63) & ~63;
    const int rstop = rend < r64 ? rend : r64;  // rows >= R carry dY == 0: skip whole chunks

    f32x4 acc[4][NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *dyl = jb.dyf + lane * 4;
    for (int p0 = rbeg; p0 < rstop; p0 += 64) {
        // straight-line chunk of 64 rows: the (L2-resident) dY fragments are issued first,
        // then the 16 x loads in consumption order; vmcnt retires in issue order, so the
        // MFMAs of row group rg wait only for x loads 0 .. 4rg+3.
        f32x4 dy[4][NT], xa[4][4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                dy[rg][nt] = *(const f32x4 *)(dyl + ((size_t)nt * RG + (p0 >> 4) + rg) * 256);
        __builtin_amdgcn_sched_barrier(0);  // pin the issue order
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xa[rg][j] = *(const f32x4 *)(row_ptr(a.rows[jb.side], p0 + 16 * rg + 4 * kq + j, a.B, a.R, a.D) +
                                             dbase + 4 * i16);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[rg][j] = norm_apply(xa[rg][j], a.norm, dbase + 4 * i16);
#ifndef ABL_GRAD_NOMFMA
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[rg][j][t], dy[rg][nt][j],
                                                                         acc[t][nt], 0, 0, 0);
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(xa[rg][j]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(dy[rg][nt]));
#endif
        }
    }

    // cross-wave sum through LDS: lds[wave][nt][t][lane]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) lds[((wave * NT + nt) * 4 + t) * 64 + lane] = acc[t][nt];
    __syncthreads();
    CFL_GRAD_FUSED_EPILOGUE()
    if (wave < NT) {
        const int nt = wave;
        f32x4 sum[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sum[t] = lds[((0 * NT + nt) * 4 + t) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) sum[t] += lds[((w * NT + nt) * 4 + t) * 64 + lane];
        }
        // acc[t][nt][e]: M row 4*kq+e <-> d = dbase + 16*kq + 4*e + t ; N col = lane&15
        // Wf block (nt, g = dbase/16 + kq), position ((q = e)*16 + c16)*4 + (e' = t)
        float *dst = jb.wpart + (size_t)p * jb.pstride + ((size_t)nt * G + (dbase >> 4) + kq) * 256 +
                     i16 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 v = {sum[0][e], sum[1][e], sum[2][e], sum[3][e]};
            *(f32x4 *)(dst + e * 64) = v;
        }
    }
}



// the step's scalars (cfl/models/cfl.py:868-949) from the row sums `sc` of the per-row loss quantities
__device__ __forceinline__ void write_scalars(float *o, const float *sc, float regsum, int B, int use_threshold,
                                              float pos_weight, float caffe_margin, float lambda_m, float thr) {
    // no fused multiply-adds in here: the function is inlined into the finalize kernel and into the reduction block of the
    // weight-gradient launch, and a contraction across the call boundary (regsum = 0.5 * reg_const * rs is an expression at
    // one call site, a value from LDS at the other) made the two differ by an ulp in `total` (round 4, found by the
    // fused-vs-finalize test once unrelated edits moved the compiler's choice)
#pragma clang fp contract(off)
    const float invB = 1.f / (float)B;
    const float pw = pos_weight != 0.f ? pos_weight : 1.f;
    const float lpos = sc[P_BCE_POS] * invB, lneg = sc[P_BCE_NEG] * invB;
    const float thres = lpos * pw + lneg;
    float cd = 0.f;
    if (caffe_margin != 0.f)
        cd = 0.5f * (sc[P_D_POS] * invB * pw + sc[P_HINGE_NEG] * invB);
    else if (lambda_m != 0.f)
        cd = sc[P_D_POS] * invB * lambda_m * pw;
    float total = regsum + cd;
    if (use_threshold) total += thres;
    o[CFL_S_TOTAL] = total;
    o[CFL_S_REG] = regsum;
    o[CFL_S_THRES] = thres;
    o[CFL_S_LOSS_POS] = lpos;
    o[CFL_S_LOSS_NEG] = lneg;
    o[CFL_S_CD] = cd;
    o[CFL_S_ACCURACY] = 0.5f * (sc[P_OK_POS] * invB + sc[P_OK_NEG] * invB);
    o[CFL_S_MEAN_D_POS] = sc[P_D_POS] * invB;
    o[CFL_S_MEAN_D_NEG] = sc[P_D_NEG] * invB;
    o[CFL_S_MEAN_O_POS] = sc[P_O_POS] * invB;
    o[CFL_S_MEAN_O_NEG] = sc[P_O_NEG] * invB;
    o[CFL_S_THRESHOLD] = thr;
    o[CFL_S_DIST_ADAPT_POS] = sc[P_SQRT_POS] * invB;
    o[CFL_S_DIST_ADAPT_NEG] = sc[P_SQRT_NEG] * invB;
    o[14] = 0.f;
    // o[CFL_S_ERROR] is sticky: set by a kernel that gave up on a hand-off, never cleared by the library
}

// one parameter: flat gradient (+ L2 term) and optional TF-Adam (fused mode, bias / threshold entries)
template <bool DP = false>
__device__ __forceinline__ void fuse_apply1(const GradFuse &f, long long off, float gr, bool reg) {
    float th = f.theta[off];
    if (reg && f.reg_const != 0.f) gr = fmaf(f.reg_const, th, gr);
    if (DP) dp_store4(dp_grad_ptr(f, off), gr);
    else f.grad[off] = gr;
    if (f.m) {
        float mm = f.m[off], vv = f.v[off];
        adam1(th, mm, vv, gr, f.lr_t, f.b1, f.b2, f.eps);
        f.m[off] = mm;
        f.v[off] = vv;
        f.theta_out[off] = th;
    }
}

// column sums of tile `idx` of a fragment-major buffer (the whole workgroup): lane (kq, c16) adds its 4 rows; the
// result is valid in lanes 0 .. 15 of wave 0.  `buf2` (dual ranges: side 0's tile of a shared head) is summed the same
// way in the same pass and returned in *cs2.
// The reduction blocks are the critical path of the launch for weight-normalised heads (every tile finisher waits for
// their c_j) and wherever a block gets more than one job, so a job is ONE round of loads: 16 row groups per wave and
// buffer in flight at once (both buffers of a dual range together), one LDS exchange for both.  The order of the
// additions is the one the two-rounds-of-8 form had (row groups wave, wave + 4, ... ascending).
__device__ __forceinline__ float tile_colsum(const float *buf, int idx, int RG, int lane, int wave, float *lds,
                                             const float *buf2 = nullptr, float *cs2 = nullptr) {
    const f32x4 *pa = (const f32x4 *)(buf + (size_t)idx * RG * 256) + lane;
    const f32x4 *pb = (const f32x4 *)((buf2 ? buf2 : buf) + (size_t)idx * RG * 256) + lane;
    float acc = 0.f, acc2 = 0.f;
    for (int rg0 = wave; rg0 < RG; rg0 += 64) {
        f32x4 v[16], w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            v[u] = rg0 + 4 * u < RG ? pa[(size_t)(rg0 + 4 * u) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (buf2) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                w[u] = rg0 + 4 * u < RG ? pb[(size_t)(rg0 + 4 * u) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
        if (buf2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc2 += (w[u][0] + w[u][1]) + (w[u][2] + w[u][3]);
        }
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (buf2) {
        acc2 += __shfl_xor(acc2, 16);
        acc2 += __shfl_xor(acc2, 32);
    }
    __syncthreads();
    if (lane < 16) {
        lds[wave * 16 + lane] = acc;
        if (buf2) lds[64 + wave * 16 + lane] = acc2;
    }
    __syncthreads();
    float cs = 0.f;
    if (wave == 0 && lane < 16) {
        cs = (lds[lane] + lds[16 + lane]) + (lds[32 + lane] + lds[48 + lane]);
        if (buf2) *cs2 = (lds[64 + lane] + lds[80 + lane]) + (lds[96 + lane] + lds[112 + lane]);
    }
    return cs;
}

template <bool DP = false>
__device__ __forceinline__ void grad_red_block(const GradArgs &a, float *lds) {   // (forceinline: an out-of-line call takes the address of the argument block, which then lives in scratch)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblk = gridDim.x * gridDim.y;
    const int RG = a.Rpad >> 4;
    for (int job = blockIdx.y * gridDim.x + blockIdx.x; job < a.red_total; job += nblk) {
        int k = 0, idx = job;
        while (k < a.nred - 1 && idx >= a.red[k].count) { idx -= a.red[k].count; ++k; }
        const RedRange &rr = a.red[k];
        if (rr.kind == 0) {
            // column sums of tile `idx` of a fragment-major buffer: lane (kq, c16) adds its 4 rows
            // siamese, fused tail: this range also covers side 0's tile of the shared head -- each side summed exactly
            // as its own range would, then added side 0 first (the finalize kernel's order)
            const float *second = rr.B;   // (kind 0: B = the second buffer of a dual range, else null)
            float csum = 0.f;
            {
                float c0 = 0.f;
                const float c1 = tile_colsum(rr.A, idx, RG, lane, wave, lds, second, &c0);
                csum = second ? c0 + c1 : c1;
            }
            const bool publish = a.fuse.on && a.fuse.wn && a.fuse.red_g[k] >= 0;   // c_j sums the W tiles wait for
            if (wave == 0 && lane < 16) {
                if (publish)   // written through (agent scope): read by tile finishers of this launch, on any XCD
                    __hip_atomic_store(a.colsum + rr.out_off + idx * 16 + lane, csum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    a.colsum[rr.out_off + idx * 16 + lane] = csum;
            }
            if (publish && wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(a.fuse.red_done, 1);
            }
            if (a.fuse.on) {
                // this block owns the whole column: finish the entries that depend on it
                const GradFuse &f = a.fuse;
                if (k == 0) {
                    // row sums of the loss quantities: threshold gradient and the step's scalars
                    __syncthreads();
                    if (wave == 0 && lane < 16) lds[64 + lane] = csum;
                    __syncthreads();
                    if (wave == 0) {
                        if (lane == 0) {
                            const float th = f.theta[f.thr_off];
                            fuse_apply1<DP>(f, f.thr_off, th >= CFL_THR_FLOOR ? lds[64 + P_DTHR] : 0.f, false);
                        } else {
                            fuse_apply1<DP>(f, f.thr_off + lane, 0.f, false);   // rest of the 64-float threshold slot
                        }
                        float rs = 0.f;
                        for (int b = lane; b < f.nregblocks; b += 64) rs += f.regpart[b];
                        rs = wave_sum(rs);
                        if (lane == 0 && DP) {
                            // (the step's scalars live in their owner's slot row: staged in LDS, then system-scope stores)
                            float *stg = lds + 128;
                            write_scalars(stg, lds + 64, 0.5f * f.reg_const * rs, f.B, f.use_threshold,
                                          f.pos_weight, f.caffe_margin, f.lambda_m, f.thr_copy[0]);
                            for (int q = 0; q < CFL_S_ERROR; ++q) dp_store4(f.scalars + q, stg[q]);
                        } else if (lane == 0) {
                            write_scalars(f.scalars, lds + 64, 0.5f * f.reg_const * rs, f.B, f.use_threshold,
                                          f.pos_weight, f.caffe_margin, f.lambda_m, f.thr_copy[0]);
                            if (f.scalars2) {   // the caller's second copy (a pinned host ring slot: no copy command on the stream)
                                write_scalars(f.scalars2, lds + 64, 0.5f * f.reg_const * rs, f.B, f.use_threshold,
                                              f.pos_weight, f.caffe_margin, f.lambda_m, f.thr_copy[0]);
                                f.scalars2[CFL_S_ERROR] = f.scalars[CFL_S_ERROR];   // (sticky: an error of THIS launch shows in the next slot at the latest)
                            }
                        }
                    }
                } else if (f.red_b[k] >= 0 && wave == 0) {
                    const int c = idx * 16 + lane;
                    if (lane < 16) {
                        fuse_apply1<DP>(f, f.red_b[k] + c, c < f.red_n[k] ? csum : 0.f, true);
                    } else if (idx == 0) {
                        // pad of the bias array up to its 64-float slot: zero gradient
                        const int cp = f.red_npad[k] + lane - 16;
                        if (cp < ((f.red_npad[k] + 63) & ~63)) fuse_apply1<DP>(f, f.red_b[k] + cp, 0.f, true);
                    }
                } else if (f.wn && f.red_g[k] >= 0 && wave == 0) {
                    // gain entries: dg_j = c_j / n_j (the RK_GAIN branch of the finalize kernel; no L2 term)
                    const int c = idx * 16 + lane;
                    if (lane < 16) {
                        float gr = 0.f;
                        if (c < f.red_n[k]) {
                            const float n2 = f.red_n2[k][c];
                            gr = n2 > 0.f ? csum / sqrtf(n2) : 0.f;
                        }
                        fuse_apply1<DP>(f, f.red_g[k] + c, gr, false);
                    } else if (idx == 0) {
                        const int cp = f.red_npad[k] + lane - 16;
                        if (cp < ((f.red_npad[k] + 63) & ~63)) fuse_apply1<DP>(f, f.red_g[k] + cp, 0.f, false);
                    }
                }
            }
        } else if (rr.kind == 3) {
            // fused tail, directed encoders: 1024 floats of the regions nobody projects through.  The finalize kernel
            // gives them gradient 0 (+ the L2 term) and applies Adam; so does this
            const GradFuse &f = a.fuse;
            long long rel = (long long)idx * 1024 + threadIdx.x * 4;
            for (int o = 0; o < f.norph; ++o) {
                if (rel < f.orph_cnt[o]) {
                    const long long off = f.orph_off[o] + rel;
                    const f32x4 th = *(const f32x4 *)(f.theta + off);
                    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
                    if (f.orph_reg[o]) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) gr[e] = fmaf(f.reg_const, th[e], gr[e]);
                    }
                    if (DP) dp_store16(dp_grad_ptr(f, off), gr);
                    else *(f32x4 *)(f.grad + off) = gr;
                    if (f.m) {
                        f32x4 mm = *(const f32x4 *)(f.m + off), vv = *(const f32x4 *)(f.v + off), tn = th;
                        adam4(tn, mm, vv, gr, f.lr_t, f.b1, f.b2, f.eps);
                        *(f32x4 *)(f.m + off) = mm;
                        *(f32x4 *)(f.v + off) = vv;
                        *(f32x4 *)(f.theta_out + off) = tn;
                    }
                    break;
                }
                rel -= f.orph_cnt[o];
            }
        } else {
            // kind 1: gate head, l = idx, dVm[l][k] for all k ; kind 2: plain column sums
            const bool fin = a.fuse.on && a.fuse.mono_w >= 0;
            for (int kk = 0; kk < rr.K; ++kk) {
                float acc = 0.f;
                if (rr.kind == 1) {
                    for (int r = threadIdx.x; r < a.Rpad; r += 256)
                        acc = fmaf(rr.A[(size_t)r * rr.lda + idx], rr.B[(size_t)r * rr.ldb + kk], acc);
                } else {
                    for (int r = threadIdx.x; r < a.Rpad; r += 256) acc += rr.A[(size_t)r * rr.lda + kk];
                }
                acc = wave_sum(acc);
                __syncthreads();
                if (lane == 0) lds[wave] = acc;
                __syncthreads();
                const float tot = (lds[0] + lds[1]) + (lds[2] + lds[3]);
                if (threadIdx.x == 0) a.colsum[rr.out_off + (rr.kind == 1 ? idx * rr.kpad : 0) + kk] = tot;
                if (fin) {
                    // fused tail: this block owns row l = idx of the gate head (kind 1) / the gate gains (kind 2)
                    const GradFuse &f = a.fuse;
                    if (rr.kind == 1) {
                        float g1 = tot;
                        if (f.mono_duc) {
                            // weight-norm correction needs c_k = sum_r du_k u_k: the kind-2 sum, recomputed here in the
                            // same order (256 strided partial sums, wave sums, four waves) -- no cross-block wait
                            float c = 0.f;
                            for (int r = threadIdx.x; r < a.Rpad; r += 256) c += f.mono_duc[(size_t)r * rr.kpad + kk];
                            c = wave_sum(c);
                            __syncthreads();
                            if (lane == 0) lds[8 + wave] = c;
                            __syncthreads();
                            const float cw = (lds[8] + lds[9]) + (lds[10] + lds[11]);
                            const float n2 = f.mono_n2[kk], n = sqrtf(n2);
                            if (threadIdx.x == 0 && n2 > 0.f)
                                g1 = fmaf(-(f.mono_gcopy[kk] * cw / (n2 * n)), f.theta[f.mono_w + (long long)idx * rr.kpad + kk], g1);
                        }
                        if (threadIdx.x == 0) fuse_apply1<DP>(f, f.mono_w + (long long)idx * rr.kpad + kk, g1, f.mono_reg != 0);
                    } else if (threadIdx.x == 0) {
                        const float n2 = f.mono_n2[kk];
                        fuse_apply1<DP>(f, f.mono_g + kk, n2 > 0.f ? tot / sqrtf(n2) : 0.f, false);
                    }
                }
            }
            if (fin && wave == 0) {
                // the padding of the owned entries: columns K .. kpad of the row (kind 1), and -- last row / kind 2 -- the
                // rest of the region up to its 64-float boundary: zero gradient (+ L2 of a zero weight)
                const GradFuse &f = a.fuse;
                if (rr.kind == 1) {
                    for (int kk = rr.K + lane; kk < rr.kpad; kk += 64) fuse_apply1<DP>(f, f.mono_w + (long long)idx * rr.kpad + kk, 0.f, f.mono_reg != 0);
                    if (idx == f.mono_L - 1) {
                        const long long used = (long long)f.mono_L * rr.kpad, end = (used + 63) / 64 * 64;
                        for (long long o = used + lane; o < end; o += 64) fuse_apply1<DP>(f, f.mono_w + o, 0.f, f.mono_reg != 0);
                    }
                } else {
                    for (int kk = rr.K + lane; kk < ((rr.kpad + 63) & ~63); kk += 64) fuse_apply1<DP>(f, f.mono_g + kk, 0.f, false);
                }
            }
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void cfl_grad_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) { grad_red_block(a, (float *)smem); return; }
    const GradJob &jb = a.job[blockIdx.z - 1];
    switch (jb.nt) {
        case 1: grad_body<1>(jb, a, lds); break;
        case 2: grad_body<2>(jb, a, lds); break;
        case 3: grad_body<3>(jb, a, lds); break;
        default: grad_body<4>(jb, a, lds); break;
    }
}

// ---------------------------------------------------------------------------
// grad, bf16x3 variant: the same contraction on the bf16 matrix cores at fp32-level accuracy.
// Every fp32 operand v is split EXACTLY into three bf16 values v = h + m + l (8 + 8 + 8
// significand bits, by truncation: h = v & 0xffff0000, m = (v - h) & 0xffff0000, l = v - h - m;
// both subtractions are exact), and a product a*b is accumulated in fp32 from the six partial
// products whose weight is >= 2^-16 of it: ah*bh, ah*bm, am*bh, ah*bl, al*bh, am*bm.  The dropped
// terms (am*bl, al*bm, al*bl) are <= 2^-21 |a*b| in the worst case and 2^-24 |a*b| rms
// (tests/test_bf16x3_split.py), the size of an fp32 rounding; against the fp64 oracle the gradient error
// equals that of the fp32 kernel (tests/test_hip_parity.py).
// v_mfma_f32_16x16x32_bf16 runs 16x the fp32 MFMA rate, so six of them over K = 32 cost 96
// cycles against 256 for the eight v_mfma_f32_16x16x4_f32 they replace; the splits are VALU work
// that co-issues in the MFMA shadows.  Data layouts (row-major x, fragment-major dYf, Wf slabs)
// and the C/D mapping are those of the fp32 kernel; only the k <-> row assignment inside a
// 32-row group differs (k = 8*kq + jj <-> row 32*R2 + 8*kq + jj).
// ---------------------------------------------------------------------------
template <int NT, bool STAGED>
__device__ __forceinline__ void grad_body_x3(const GradJob &jb, const GradArgs &a, f32x4 *lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int dbase = grad_dtile(a.tps) * 64;
    const int p = blockIdx.y;
    const int RG = a.Rpad >> 4, G = a.D >> 4;
    const int rows_wg = a.Rpad / a.P, rows_w = rows_wg >> 2;
    const int rbeg = p * rows_wg + wave * rows_w, rend = rbeg + rows_w;
    const int r64 = (a.R + 63) & ~63;
    const int rstop = rend < r64 ? rend : r64;

    f32x4 acc[4][NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // dYf block (nt, rg) holds [kq'][c16][j] <-> row 16rg + 4kq' + j; this lane's k-group covers rows
    // 32*R2 + 8*kq + jj: rg = 2*R2 + (kq >> 1), kq' = 2*(kq & 1) + (jj >> 2), j = jj & 3.  The dY
    // fragments are L2 hits that land long before x does, so splitting them costs no wall time.
    const float *dyl = jb.dyf + ((size_t)(kq >> 1) * 256 + (2 * (kq & 1) * 16 + i16) * 4);
    // The addresses of the workgroup's whole row range are staged in LDS once (the region is reused by the
    // cross-wave sum below, behind a barrier): with an indexed source each address starts with an index load, and
    // those loads in front of every 64-row group's x loads -- a dependent global round trip per group -- cost
    // 2.6 us per launch at the headline shape (tools/idx_probe.py); this way one coalesced round trip is paid, at
    // the start.  Dense sources take the same route: 16 addresses per lane and group out of two ds_read_b128s
    // instead of 16 clamp / select / multiply chains in front of the loads.
    const RowSrc rs = jb.side ? a.rows[1] : a.rows[0];   // (a reference to a.rows[runtime index] would put `a` on the stack)
    // STAGED <=> rows_wg <= 8192 (64 KiB of LDS); otherwise row_ptr per group.
    // (staged as element offsets from x0, not as pointers: a pointer loaded from LDS has no known address space and
    // would turn the x loads into flat loads)
    const long long *lrow = (const long long *)lds;
    if (STAGED) {
        long long *w = (long long *)lds;
        for (int r = threadIdx.x; r < rows_wg; r += 256) w[r] = row_ptr(rs, p * rows_wg + r, a.B, a.R, a.D) - rs.x0;
        __syncthreads();
    }
    for (int p0 = rbeg; p0 < rstop; p0 += 64) {
        f32x4 dyr[2][NT][2], xr[2][8];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float *q = dyl + ((size_t)nt * RG + (p0 >> 4) + 2 * r2) * 256;
                dyr[r2][nt][0] = *(const f32x4 *)q;
                dyr[r2][nt][1] = *(const f32x4 *)(q + 64);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const float *xrow[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                xrow[jj] = STAGED ? rs.x0 + lrow[(p0 - p * rows_wg) + 32 * r2 + 8 * kq + jj]
                                  : row_ptr(rs, p0 + 32 * r2 + 8 * kq + jj, a.B, a.R, a.D);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                // non-temporal: the weight gradient is the step's LAST reader of x (measured: -0.3 us at B = 512, -2.3 us at
                // B = 2048, -12 % at B = 8192; the projection keeps the default policy so that this re-read hits the Infinity Cache)
                xr[r2][jj] = __builtin_nontemporal_load((const f32x4 *)(xrow[jj] + dbase + 4 * i16));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            bf16x8 bf[NT][3];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = dyr[r2][nt][jj >> 2][jj & 3];
                split_frag(v, bf[nt]);
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) xr[r2][jj] = norm_apply(xr[r2][jj], a.norm, dbase + 4 * i16);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = xr[r2][jj][t];
                bf16x8 af[3];
                split_frag_x(v, af);
                // small terms first; consecutive MFMAs hit different accumulators
#define CFL_X3(LA, LB)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[t][nt] =                                   \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[LA], bf[nt][LB], acc[t][nt], 0, 0, 0);
                CFL_X3(1, 1) CFL_X3(2, 0) CFL_X3(0, 2) CFL_X3(1, 0) CFL_X3(0, 1) CFL_X3(0, 0)
#undef CFL_X3
            }
        }
    }

    // cross-wave sum and slab store: identical to the fp32 body (same C/D mapping)
    if (STAGED) __syncthreads();   // every wave is done with the staged addresses
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) lds[((wave * NT + nt) * 4 + t) * 64 + lane] = acc[t][nt];
    __syncthreads();
    CFL_GRAD_FUSED_EPILOGUE()
    if (wave < NT) {
        const int nt = wave;
        f32x4 sum[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sum[t] = lds[((0 * NT + nt) * 4 + t) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) sum[t] += lds[((w * NT + nt) * 4 + t) * 64 + lane];
        }
        float *dst = jb.wpart + (size_t)p * jb.pstride + ((size_t)nt * G + (dbase >> 4) + kq) * 256 +
                     i16 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 v = {sum[0][e], sum[1][e], sum[2][e], sum[3][e]};
            *(f32x4 *)(dst + e * 64) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// grad, bf16x3, HALF tiles without a row split (round 3): a workgroup owns a 32-d tile and ALL rows (P = 1), so a
// gradient tile is complete inside ONE workgroup and the fused tail needs no hand-off at all (publish -> drain ->
// counter -> poll -> sc1 loads cost ~3 us of the 64-d / P = 2 launch at the headline shape, measured with CFL_DEBUG_P).
// Same number of workgroups (D/32 x jobs), same bytes of x and the same MFMA work per wave; lane i16 holds d =
// dbase + 2 i16 + t (t = 0, 1: two M blocks instead of four) and fetches 8 bytes per row, 4 rows x 128 bytes per
// instruction; the dY fragments are read by twice as many workgroups (L2 hits).
//   acc[t][nt][e]: M row 4 kq + e <-> d = dbase + 8 kq + 2 e + t ; N col = lane & 15
//   Wf block (nt, g = dbase/16 + (kq >> 1)), float4 h = e >> 1 at ((2 (kq & 1) + h) * 16 + c16) * 4: elements
//   (e & 1, t) = (0,0) (0,1) (1,0) (1,1)
// ---------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT, bool HO, int NW = 4, bool DP = false>   // NW: waves per workgroup (8: two waves per SIMD); HO: row split and / or siamese pairing (hand-off tail); false: the tile is complete in the workgroup; DP: fused push of the one-shot exchange
__device__ __forceinline__ void grad_body_x3_half(const GradJob &jb, const GradArgs &a, f32x4 *lds, int job, int dtile,
                                                  int p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int dbase = dtile * 32;
    const int RG = a.Rpad >> 4, G = a.D >> 4;
    // (HO == false keeps every trace of the hand-off out of the headline's kernel: the general tail, although it
    // takes the same branches there, measured +0.55 us per step)
    const GradFuse &f = a.fuse;
    const int P = HO ? a.P : 1;
    const int pair_jobs = HO ? f.pair_jobs : 0;
    const long long pair_delta = HO ? f.pair_delta : 0;
    const bool paired = pair_jobs > 0;
    const bool side0 = paired && job < pair_jobs;
    const int slot = (paired && !side0 ? job - pair_jobs : job) * (a.D >> 5) + dtile;
    const int expect = paired ? 2 * P - 1 : P - 1;
    const int rows_wg = HO ? a.Rpad / P : a.Rpad, rows_w = rows_wg / NW;   // multiple of 64
    const int rbeg = p * rows_wg + wave * rows_w, rend = rbeg + rows_w;
    const int r64 = (a.R + 63) & ~63;
    const int rstop = rend < r64 ? rend : r64;

    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *dyl = jb.dyf + ((size_t)(kq >> 1) * 256 + (2 * (kq & 1) * 16 + i16) * 4);
    const RowSrc rs = jb.side ? a.rows[1] : a.rows[0];
    const long long *lrow = (const long long *)lds;   // row addresses of the whole batch, staged once (grad_body_x3)
    auto loaddy = [&](int p0, f32x4 (*dyr)[NT][2]) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float *q = dyl + ((size_t)nt * RG + (p0 >> 4) + 2 * r2) * 256;
                dyr[r2][nt][0] = *(const f32x4 *)q;
                dyr[r2][nt][1] = *(const f32x4 *)(q + 64);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the first chunk's dL/dy does not depend on the row addresses: requested before they are staged, so that its
    // latency overlaps the index loads of the indexed entry points (and the staging barrier)
    f32x4 dyr[2][NT][2];
    if (rbeg < rstop) loaddy(rbeg, dyr);
    {
        // every wave stages the addresses of ITS rows only: LDS operations of one wave are ordered, no workgroup barrier
        long long *w = (long long *)lds;
        for (int r = rbeg + lane; r < rend; r += 64) w[r] = row_ptr(rs, r, a.B, a.R, a.D) - rs.x0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto loadx = [&](int p0, f32x2 (*dst)[8]) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const float *xrow[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) xrow[jj] = rs.x0 + lrow[p0 + 32 * r2 + 8 * kq + jj];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)   // non-temporal: last reader of x in the step
                dst[r2][jj] = __builtin_nontemporal_load((const f32x2 *)(xrow[jj] + dbase + 2 * i16));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int p0 = rbeg; p0 < rstop; p0 += 64) {
        f32x2 xr[2][8];
        if (p0 != rbeg) loaddy(p0, dyr);
        loadx(p0, xr);
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            bf16x8 bf[NT][3];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = dyr[r2][nt][jj >> 2][jj & 3];
                split_frag(v, bf[nt]);
            }
            if (a.norm.elementwise) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        xr[r2][jj][t] = dbase + 2 * i16 + t < a.norm.valid
                                            ? fminf(fmaxf(fmaf(xr[r2][jj][t], a.norm.mul, a.norm.add), a.norm.lo), a.norm.hi) : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = xr[r2][jj][t];
                bf16x8 af[3];
                split_frag_x(v, af);
#define CFL_X3(LA, LB)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[t][nt] =                                   \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[LA], bf[nt][LB], acc[t][nt], 0, 0, 0);
                CFL_X3(1, 1) CFL_X3(2, 0) CFL_X3(0, 2) CFL_X3(1, 0) CFL_X3(0, 1) CFL_X3(0, 0)
#undef CFL_X3
            }
        }
    }
    __syncthreads();   // every wave is done with the staged addresses
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 2; ++t) lds[((wave * NT + nt) * 2 + t) * 64 + lane] = acc[t][nt];
    __syncthreads();
    const int ntw = wave < NT ? wave : 0;
    f32x4 sum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        sum[t] = lds[((0 * NT + ntw) * 2 + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum[t] += lds[((w * NT + ntw) * 2 + t) * 64 + lane];
    }
    const size_t tile_off = ((size_t)ntw * G + (dbase >> 4) + (kq >> 1)) * 256 + (2 * (kq & 1) * 16 + i16) * 4;
    if (!a.fuse.on) {
        if (wave < NT) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *(f32x4 *)(jb.wpart + (size_t)p * jb.pstride + tile_off + h * 64) =
                    (f32x4){sum[0][2 * h], sum[1][2 * h], sum[0][2 * h + 1], sum[1][2 * h + 1]};
        }
        return;
    }
    // fused tail.  P == 1 and one side per head: the tile is complete here, no hand-off at all.  Otherwise the protocol of
    // grad_fused_tail on half tiles: the first P - 1 row ranges (siamese: and all of side 0) publish their partial tile
    // (write-through, drained, one arrival count per tile) and leave; the last row range (of side 1) finishes.
    if (HO && (p < P - 1 || side0)) {
        if (wave < NT)
            store2_sc1_wait(jb.wpart + (size_t)p * jb.pstride + tile_off, (f32x4){sum[0][0], sum[1][0], sum[0][1], sum[1][1]},
                            (f32x4){sum[0][2], sum[1][2], sum[0][3], sum[1][3]});
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(f.flag + slot, 1);
        return;
    }
    const long long base = f.w_off[job] + (long long)tile_off;
    f32x4 th[2], mm[2], vv[2];
    if (wave < NT) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            th[h] = *(const f32x4 *)(f.theta + base + h * 64);
            if (f.m) {
                mm[h] = *(const f32x4 *)(f.m + base + h * 64);
                vv[h] = *(const f32x4 *)(f.v + base + h * 64);
            }
        }
    }
    const int wcol = ntw * 16 + i16;
    float wg = 1.f, wn2 = 1.f;
    bool lost = false;
    if (f.wn) { wg = f.wn_g[job][wcol]; wn2 = f.wn_n2[job][wcol]; }
    if ((HO && expect > 0) || f.wn) {
        if (threadIdx.x == 0) {   // bounded waits, as in grad_fused_tail
            int ok = f.spin_limit < 0 ? 0 : 1;
            if (expect > 0 && ok) ok = handoff_wait(f.flag + slot, expect, f);
            if (f.wn && ok) ok = handoff_wait(f.red_done, f.red_expect, f);   // the c_j column sums of this launch's reduction blocks (dispatched first, short)
            ((int *)lds)[0] = ok;
            if (!ok) { if (DP) dp_store4(f.scalars + CFL_S_ERROR, 1.f); else f.scalars[CFL_S_ERROR] = 1.f; }   // sticky error word (see handoff_wait)
        }
        __syncthreads();
        lost = lost || ((int *)lds)[0] == 0;
    }
    if (wave < NT) {
        f32x4 g[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // published tiles in the finalize kernel's order (side 0's row ranges, then this side's 0 .. P-2), three per
        // round trip; the finisher's own registers last
        const int npair = paired ? P : 0, nparts = npair + P - 1;
        const float *own = jb.wpart + tile_off;
        auto part_ptr = [&](int k) {
            k = k < nparts ? k : nparts - 1;
            return k < npair ? own - pair_delta + (size_t)k * jb.pstride : own + (size_t)(k - npair) * jb.pstride;
        };
        for (int k = 0; HO && k < nparts; k += 3) {
            f32x4 part[6];
            load2x3_sc1_wait(part_ptr(k), part_ptr(k + 1), part_ptr(k + 2), part);
            g[0] += part[0]; g[1] += part[1];
            if (k + 1 < nparts) { g[0] += part[2]; g[1] += part[3]; }
            if (k + 2 < nparts) { g[0] += part[4]; g[1] += part[5]; }
        }
        if (HO && nparts > 0) {
            g[0] += (f32x4){sum[0][0], sum[1][0], sum[0][1], sum[1][1]};
            g[1] += (f32x4){sum[0][2], sum[1][2], sum[0][3], sum[1][3]};
        } else {
            g[0] = (f32x4){sum[0][0], sum[1][0], sum[0][1], sum[1][1]};
            g[1] = (f32x4){sum[0][2], sum[1][2], sum[0][3], sum[1][3]};
        }
        float s1 = f.in_mul, s2 = 0.f;
        if (f.wn) {
            const float cw = __hip_atomic_load(f.wn_cw[job] + wcol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool colok = wcol < f.wn_n[job];
            const float n = sqrtf(wn2);
            s1 = (colok && wn2 > 0.f) ? f.in_mul * wg / n : 0.f;
            s2 = (colok && wn2 > 0.f) ? wg * cw / (wn2 * n) : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 gh = g[h];
            if (lost) gh = (f32x4){NAN, NAN, NAN, NAN};
            f32x4 gr = gh * s1;
            if (f.wn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gr[i] = fmaf(-s2, th[h][i], gr[i]);
            }
            fuse_apply<DP>(f, base + h * 64, gr, th[h], mm[h], vv[h]);
        }
        if (f.planes && f.m) {
            // kept bf16 planes of the updated weights: this lane holds d = 32 dtile + 8 kq + (0 .. 7) of column i16 -- exactly
            // fragment lane `lane` of quarter tq = dtile in cfl_wplanes_kernel's layout: one 16-byte store per plane
            float vals[8] = {th[0][0], th[0][1], th[0][2], th[0][3], th[1][0], th[1][1], th[1][2], th[1][3]};
            bf16x8 fr[3];
            split_frag_rne(vals, fr);
            unsigned short *dst = f.planes + 3 * f.w_off[job] + ((size_t)(ntw * (a.D >> 5) + dtile) * 3) * 512 + lane * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *(bf16x8 *)(dst + pl * 512) = fr[pl];
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void cfl_grad_x3_half_kernel(GradArgs a_) {   // P == 1, one side per head; Rpad <= 8192 (staged row addresses)
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) { grad_red_block(a, (float *)smem); return; }
    const GradJob &jb = a.job[blockIdx.z - 1];
    const int job = (int)blockIdx.z - 1, dt = grad_dtile(a.tps);
    switch (jb.nt) {
        case 1: grad_body_x3_half<1, false>(jb, a, lds, job, dt, 0); break;
        case 2: grad_body_x3_half<2, false>(jb, a, lds, job, dt, 0); break;
        case 3: grad_body_x3_half<3, false>(jb, a, lds, job, dt, 0); break;
        default: grad_body_x3_half<4, false>(jb, a, lds, job, dt, 0); break;
    }
}

// the same tile with EIGHT waves (two per SIMD, half the rows each): the headline plan (CFL_DEBUG_GRAD_W8=-1: four waves)
extern "C" __global__ __launch_bounds__(512) void cfl_grad_x3_half_w8_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) {
        if (threadIdx.x >= 256) return;   // (the reduction blocks are written for four waves; a finished wave does not count at a barrier)
        grad_red_block(a, (float *)smem);
        return;
    }
    const GradJob &jb = a.job[blockIdx.z - 1];
    const int job = (int)blockIdx.z - 1, dt = grad_dtile(a.tps);
    switch (jb.nt) {
        case 1: grad_body_x3_half<1, false, 8>(jb, a, lds, job, dt, 0); break;
        case 2: grad_body_x3_half<2, false, 8>(jb, a, lds, job, dt, 0); break;
        case 3: grad_body_x3_half<3, false, 8>(jb, a, lds, job, dt, 0); break;
        default: grad_body_x3_half<4, false, 8>(jb, a, lds, job, dt, 0); break;
    }
}

// ... with a row split (grid y = P row ranges) and / or the siamese pairing: hand-off tail
extern "C" __global__ __launch_bounds__(256) void cfl_grad_x3_half_split_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) { grad_red_block(a, (float *)smem); return; }
    const GradJob &jb = a.job[blockIdx.z - 1];
    const int job = (int)blockIdx.z - 1, dt = grad_dtile(a.tps);
    switch (jb.nt) {
        case 1: grad_body_x3_half<1, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
        case 2: grad_body_x3_half<2, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
        case 3: grad_body_x3_half<3, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
        default: grad_body_x3_half<4, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
    }
}

// ---- the half-tile kernels with the FUSED PUSH of the data-parallel one-shot exchange (round 6; GradFuse::dp_*) ------------------
// Kernels of their own, so that the single-GPU kernels above stay what they were instruction for instruction (a run-time switch
// inside them measured +0.5 us per headline step, profiles/r06_nodp_ab_first.txt).  Same bodies, template parameter DP: every
// finished entry of [gradient | scalars] goes to its owner rank's slot, every workgroup ends with grad_dp_block_done().
extern "C" __global__ __launch_bounds__(256) void cfl_grad_x3_half_dp_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) { grad_red_block<true>(a, (float *)smem); grad_dp_block_done(a.fuse); return; }
    const GradJob &jb = a.job[blockIdx.z - 1];
    const int job = (int)blockIdx.z - 1, dt = grad_dtile(a.tps);
    switch (jb.nt) {
        case 1: grad_body_x3_half<1, false, 4, true>(jb, a, lds, job, dt, 0); break;
        case 2: grad_body_x3_half<2, false, 4, true>(jb, a, lds, job, dt, 0); break;
        case 3: grad_body_x3_half<3, false, 4, true>(jb, a, lds, job, dt, 0); break;
        default: grad_body_x3_half<4, false, 4, true>(jb, a, lds, job, dt, 0); break;
    }
    grad_dp_block_done(a.fuse);
}

extern "C" __global__ __launch_bounds__(512) void cfl_grad_x3_half_w8_dp_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) {
        if (threadIdx.x >= 256) return;   // (four-wave reduction blocks; a finished wave does not count at a barrier)
        grad_red_block<true>(a, (float *)smem);
        grad_dp_block_done(a.fuse);
        return;
    }
    const GradJob &jb = a.job[blockIdx.z - 1];
    const int job = (int)blockIdx.z - 1, dt = grad_dtile(a.tps);
    switch (jb.nt) {
        case 1: grad_body_x3_half<1, false, 8, true>(jb, a, lds, job, dt, 0); break;
        case 2: grad_body_x3_half<2, false, 8, true>(jb, a, lds, job, dt, 0); break;
        case 3: grad_body_x3_half<3, false, 8, true>(jb, a, lds, job, dt, 0); break;
        default: grad_body_x3_half<4, false, 8, true>(jb, a, lds, job, dt, 0); break;
    }
    grad_dp_block_done(a.fuse);
}

extern "C" __global__ __launch_bounds__(256) void cfl_grad_x3_half_split_dp_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) { grad_red_block<true>(a, (float *)smem); grad_dp_block_done(a.fuse); return; }
    const GradJob &jb = a.job[blockIdx.z - 1];
    const int job = (int)blockIdx.z - 1, dt = grad_dtile(a.tps);
    switch (jb.nt) {
        case 1: grad_body_x3_half<1, true, 4, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
        case 2: grad_body_x3_half<2, true, 4, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
        case 3: grad_body_x3_half<3, true, 4, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
        default: grad_body_x3_half<4, true, 4, true>(jb, a, lds, job, dt, (int)blockIdx.y); break;
    }
    grad_dp_block_done(a.fuse);
}

// Two kernels rather than one with both bodies: eight inlined instantiations make the compiler keep `a` on the
// stack (1.5 KiB of scratch per lane, occupancy 1).
template <bool STAGED>
__device__ __forceinline__ void grad_x3_kernel_body(const GradArgs &a, char *smem) {
    f32x4 *lds = (f32x4 *)smem;
    if (blockIdx.z == 0) { grad_red_block(a, (float *)smem); return; }
    const GradJob &jb = a.job[blockIdx.z - 1];
    switch (jb.nt) {
        case 1: grad_body_x3<1, STAGED>(jb, a, lds); break;
        case 2: grad_body_x3<2, STAGED>(jb, a, lds); break;
        case 3: grad_body_x3<3, STAGED>(jb, a, lds); break;
        default: grad_body_x3<4, STAGED>(jb, a, lds); break;
    }
}

extern "C" __global__ __launch_bounds__(256) void cfl_grad_x3_kernel(GradArgs a_) {   // Rpad / P <= 8192
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    grad_x3_kernel_body<true>(a, smem);
}

extern "C" __global__ __launch_bounds__(256) void cfl_grad_x3_longrange_kernel(GradArgs a_) {
    CFL_KERNARG_IN_PLACE(GradArgs, a, a_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    grad_x3_kernel_body<false>(a, smem);
}

