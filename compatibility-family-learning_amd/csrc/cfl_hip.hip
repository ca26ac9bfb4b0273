// cfl_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for the cfl pair-distance
// training / scoring hot path.  See include/cfl_hip.h for the boundary and
// DESIGN.md for the data layout and the roofline of every kernel.
//
// One training step = 3 launches on the caller's stream, for every model:
//   proj       projection partials  Y_s = X[:, slice_s] . W[slice_s, :]  (fp32 MFMA, k-ordered: chunk-at-a-time or
//              streaming form; bf16x3 with LDS-shared W planes from 4096 (scoring) / 3072 (training) rows per side, after a small plane-split
//              launch; weight-norm: the column norms ride in the launch as an extra slice)
//   mid        slice-sum + bias/scale/activation, distance, loss, dL/dY
//              (+ extra blocks: L2-regulariser partial sums)
//   grad       weight-gradient partials  dW_p = X[rows_p, :]^T . dY[rows_p, :]  (bf16x3; fp32 MFMA with CFL_EXACT_FP32=1),
//              row reductions for bias / gain / gate / loss scalars in z-slice 0, and the FUSED TAIL: partial tiles are
//              handed over inside the launch, summed in a fixed order, turned into the flat gradient and TF-Adam is
//              applied (GradFuse)
//   [finalize] only with CFL_DEBUG_NOFUSE=1: partial slabs -> flat gradient, scalars, optional fused TF-Adam
// cfl_adam_tf is a separate entry point so that a data-parallel caller can
// all-reduce the flat gradient between cfl_pair_step_fwd_bwd and the update.
//
// FRAGMENT-MAJOR LAYOUTS.  Every operand that the library owns is stored in the
// order the 16x16x4 fp32 MFMA consumes it, so that each wave instruction moves one
// contiguous 1 KiB block:
//   weights (theta, Adam slots, gradient, gradient slabs)  Wf[nt][g][q][c16][e]
//        = W[d = 16g + 4q + e][col = 16nt + c16]        (one 1 KiB block per (nt, g))
//   dL/dY (scratch)                                       dYf[nt][rg][kq][c16][j]
//        = dY[row = 16rg + 4kq + j][col = 16nt + c16]   (one 1 KiB block per (nt, rg))
// Lane l of a wave reads the float4 at block + 16*l bytes: l&15 is the MFMA N index
// (column), l>>4 the MFMA K index, and the four floats feed four consecutive MFMA
// K-steps.  The input batches x are caller-owned row-major [B][D]; the projection
// kernel reads them in full 256-byte row segments and transposes them into MFMA
// order through a wave-private XOR-swizzled LDS tile.
//
// Reference arithmetic restated (paths relative to the reference tree):
//   heads      cfl/models/dist.py:43-68, cfl/layers.py:80-90, cfl/models/base.py:43-105
//   distances  cfl/models/base.py:107-146 (== cfl/models/dist.py:70-89)
//   threshold  cfl/models/blocks.py:18-22
//   losses     cfl/models/cfl.py:868-949, cfl/models/dist.py:253-284
//   Adam       tf.train.AdamOptimizer (TF-1.x), cfl/models/cfl.py:1077-1085
#include <hip/hip_runtime.h>
#include <type_traits>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/cfl_hip.h"
#include "gemm_gather.h"
#include "theta_planes.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CFL_MAX_JOBS 16
#define CFL_MAX_REGIONS 20
#define CFL_THR_FLOOR 1e-6f
#define CFL_HANDOFF_SPIN_LIMIT (1 << 22)   // polls (with s_sleep) before an in-launch hand-off is declared lost (default; CFL_DEBUG_SPIN_LIMIT overrides, < 0: give up at once)
// A lost hand-off is LOUD: the kernel that gives up stores 1.0f into scalars[CFL_S_ERROR] -- a sticky word the library only
// ever sets (the caller zeroes it once) -- and poisons what it was about to write with NaN.  The host finds the word at
// its next read-back of the scalars (cfl_scalars_status(); PairEngine.read_scalars / DeferredScalars raise CflHipError).

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// shared with cfl_conv.hip
int cfl_set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess)                                                      \
            return set_err(CFL_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---------------------------------------------------------------------------
// optional per-kernel event timing (cfl_profile_enable / cfl_profile_read)
// ---------------------------------------------------------------------------
struct ProfRec { int kind; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

struct ProfScope {
    hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(hipStream_t s, int k) : st(s), kind(k), on(g_prof_on) {
        if (on) {
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(b, st);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof.push_back({kind, a, b});
        }
    }
};

extern "C" int cfl_profile_enable(int on) { g_prof_on = on != 0; return CFL_OK; }

extern "C" int cfl_profile_read(double *ms_sum, int64_t *launches) {
    if (!ms_sum || !launches) return CFL_E_SHAPE;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.kind >= 0 && r.kind < CFL_K_COUNT) {
            ms_sum[r.kind] += ms;
            launches[r.kind] += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return CFL_OK;
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

#ifdef CFL_STAMPS
// Diagnostic build only (tools/stamp_probe.py): per-wave s_memtime stamps of the proj / grad
// kernels, written to a buffer no other code reads.
__device__ unsigned long long cfl_stamps[16384 * 8];
#define STAMP(slot)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        unsigned long long _t = __builtin_readcyclecounter();                                \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if ((threadIdx.x & 63) == 0)                                                         \
            cfl_stamps[(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + \
                        (threadIdx.x >> 6)) * 8 + (slot)] = _t;                                \
    } while (0)
extern "C" int cfl_debug_read_stamps(unsigned long long *host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(cfl_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
extern "C" int cfl_debug_clear_stamps(void) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(cfl_stamps)) != hipSuccess) return -2;
    return hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 8) == hipSuccess ? 0 : -2;
}
#define RSTAMP(slot)                                                                         \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        unsigned long long _t = __builtin_readcyclecounter();                                \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if ((threadIdx.x & 63) == 0) cfl_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (slot)] = _t; \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define RSTAMP(slot) do {} while (0)
#endif

// The argument block is read IN PLACE from the kernel-argument segment (it is the launch's only explicit argument, at
// offset 0): clang gives a by-value aggregate parameter a private copy that is only optimised away while the number of
// accesses stays under an internal limit -- past it the whole block (2.2 KB per lane) lives in scratch and the
// weight-gradient launch takes 2.7x as long (seen twice: eight inlined bodies in one kernel, and again with the
// siamese pairing fields).
#define CFL_KERNARG_IN_PLACE(T, name, param)                                                          \
    (void)param;                                                                                       \
    const T &name = *(const T *)__builtin_amdgcn_kernarg_segment_ptr()

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
struct NormDev {
    float mul, add, lo, hi;
    int elementwise;  // 1: x_hat = clip(x*mul+add) at load; 0: mul folded into the epilogue
    int valid;        // feature columns >= valid are the zero padding up to a multiple of 64: they stay zero
};

// v = the four features of columns col .. col+3 of a row
// XOR swizzle of the wave-private 32-row x 128-byte transpose tile of cfl_proj_x3_kernel (eight 16-byte chunks per row), matched to
// the lane groups in which the LDS serves ds_read_b128 -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 (MI355X_MICROARCH.md): a
// group holds every fragment row once, rows {0-3, 12-15} with chunk X = 2 kq + c and rows {4-11} with X ^ 2.  Two rows share a
// 256-byte bank row, so the eight rows of one parity must land on eight different chunks: this table does that for both groups
// (the `row & 7` it replaces is 2-way on every slot).  Round 4, same box: scoring call 261 -> 252-254 us per 32768 pairs.  The
// chunk-at-a-time forms keep `row & 7`: measured, the new table changes nothing for the exact-fp32 kernels and costs the
// headline's cfl_proj_bx3_kernel step 0.5 us (three alternations) although it removes its conflicts too.  Stores (one row per 8
// lanes) are conflict-free either way.
__device__ __forceinline__ int xt_sw2(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) * 6); }

__device__ __forceinline__ f32x4 norm_apply(f32x4 v, const NormDev &n, int col) {
    if (n.elementwise) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = col + i < n.valid ? fminf(fmaxf(fmaf(v[i], n.mul, n.add), n.lo), n.hi) : 0.f;
    }
    return v;
}

// Where the input rows of one side (src or dst) of a launch come from: two dense row blocks (rows [0,B) in x0,
// rows [B,2B) in x1), or -- ix0 != nullptr -- rows of a resident feature table picked by two index streams
// (ix0[r * istride] for r < B, ix1[(r-B) * istride] above; x0 == x1 == table).  The indexed form is what lets the
// training loop feed the step straight from the HBM-resident features.b (no gather pass, no batch copy); a row is
// 4*D contiguous bytes either way, so every access pattern of the kernels is unchanged.
struct RowSrc {
    const float *x0, *x1;
    const int *ix0, *ix1;
    int istride;
    unsigned last_row;   // table rows - 1: indices are clamped (memory safety; valid indices are never changed)
};

__device__ __forceinline__ const float *row_ptr(const RowSrc &s, int r, int B, int R, int D) {
    // rows >= R are clamped to a valid row (their products are never stored / are multiplied by zero dY).
    int rc = r < R ? r : R - 1;
    if (s.ix0) {   // uniform
        const int *ip = rc < B ? s.ix0 + (size_t)rc * s.istride : s.ix1 + (size_t)(rc - B) * s.istride;
        unsigned t = (unsigned)*ip;
        t = t < s.last_row ? t : s.last_row;
        return s.x0 + (size_t)t * D;
    }
    return rc < B ? s.x0 + (size_t)rc * D : s.x1 + (size_t)(rc - B) * D;
}

// table row of batch row r of an indexed source (same clamps as row_ptr)
__device__ __forceinline__ unsigned row_index(const RowSrc &s, int r, int B, int R) {
    const int rc = r < R ? r : R - 1;
    const int *ip = rc < B ? s.ix0 + (size_t)rc * s.istride : s.ix1 + (size_t)(rc - B) * s.istride;
    const unsigned t = (unsigned)*ip;
    return t < s.last_row ? t : s.last_row;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// bf16x3 operand splitting (used by the grad x3 kernel below, which documents the numerics)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float v, float &h, float &m, float &l) {
    h = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
    const float r = v - h;
    m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l = r - m;
}
// {bf16(e1), bf16(e0)} by truncation: the high halves of the two floats
__device__ __forceinline__ unsigned pack_hi16(float e0, float e1) {
    return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
// Round-to-nearest variant of the split (round 4): h = bf16_rne(v), m = bf16_rne(v - h), l = v - h - m -- still exact
// (v - h and r - m are exact in fp32, l has at most 8 significant bits), but |m| <= 2^-9 |v| and |l| <= 2^-17 |v| with
// errors of either sign, so the partial products a kernel DROPS are 16x smaller than with truncation (am*bl + al*bm +
// al*bl <= 2^-23 |ab|, zero-mean instead of <= 2^-21 |ab|, one-signed) -- and it is cheaper: v_cvt_pk_bf16_f32 rounds and packs two
// values per instruction and the subtractions pair up in v_pk_add_f32 (9 VALU instructions per two values against 11).
// Used for the kept planes of theta (cfl_wplanes_kernel, the fused Adam tails) and the A operand of cfl_proj_bx3_kernel.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_rne(float v0, float v1, unsigned &h, unsigned &m, unsigned &l) {
    const f32x2 v = {v0, v1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const f32x2 r = v - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r - mf, bf16x2));
}
__device__ __forceinline__ void split_frag_rne(const float (&v)[8], bf16x8 (&f)[3]) {
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b, c;
        split_pair_rne(v[2 * i], v[2 * i + 1], a, b, c);
        ph[i] = a; pm[i] = b; pl[i] = c;
    }
    f[0] = __builtin_bit_cast(bf16x8, ph);
    f[1] = __builtin_bit_cast(bf16x8, pm);
    f[2] = __builtin_bit_cast(bf16x8, pl);
}

// three bf16x8 fragments (levels h, m, l) of the 8 floats v[0..7], by truncation.  (Round 4 measured the round-to-nearest
// form above in its place, -DCFL_SPLIT_RNE: fewer instructions -- 9 against 11 per two values -- but the weight-gradient
// launch got 0.4 .. 1.0 us SLOWER, profiles/r04_split_ab.txt: v_cvt_pk_bf16_f32 and v_pk_add_f32 do not issue at the rate
// of the and / sub / perm chain.  Truncation stays for the operands split inside the loops; the kept planes of theta,
// which are split once per update, are round-to-nearest -- one rounded operand is enough to make the dropped cross
// terms zero-mean.)
__device__ __forceinline__ void split_frag(const float (&v)[8], bf16x8 (&f)[3]) {
#ifdef CFL_SPLIT_RNE
    split_frag_rne(v, f);
    return;
#endif
    float h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split3(v[i], h[i], m[i], l[i]);
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ph[i] = pack_hi16(h[2 * i], h[2 * i + 1]);
        pm[i] = pack_hi16(m[2 * i], m[2 * i + 1]);
        pl[i] = pack_hi16(l[2 * i], l[2 * i + 1]);
    }
    f[0] = __builtin_bit_cast(bf16x8, ph);
    f[1] = __builtin_bit_cast(bf16x8, pm);
    f[2] = __builtin_bit_cast(bf16x8, pl);
}

// ---------------------------------------------------------------------------
// colnorm (weight-norm): n2[c] = sum_d V[d][c]^2   (cfl/layers.py:81) + gain snapshot.
// The blocks ride in the projection launch as an extra z-slice (the projection uses the raw V; `mid`, the next
// launch, is the first consumer of the norms), like the row reductions ride in the weight-gradient launch: one
// launch less per step of a weight-normalised model (a stand-alone colnorm launch measured 6.9 us, all dispatch).
// ---------------------------------------------------------------------------
struct ColnormArgs {
    const float *theta;
    float *n2;            // [ncols_total]
    float *gcopy;         // [ncols_total] snapshot of the gains (finalize may update theta in place)
    long long g_off[8];
    int nheads, ncols;
    long long w_off[8];
    int npad[8], n2_off[8], rowlen[8], strided[8];
    int D;
};

// One WAVE per column (round 4; a workgroup of 256 threads used to walk its columns one after the other, each with a
// dependent round trip to memory and two barriers: with 256 columns on 64 blocks -- config 3 -- the colnorm slice was
// the critical path of the projection launch).  All loads of a column are independent and issued together; the lane
// sums are combined with wave_sum in a fixed order; no LDS, no barrier.  Columns are dealt to the nb blocks of the slice
// first and to the four waves of a block second (column c -> block c mod nb, wave (c / nb) mod 4), so that few columns on
// many blocks (headline shape with weight norm: 96 on 256) still run one per block, all at once.
__device__ __forceinline__ void colnorm_columns(const ColnormArgs &a, int block, int nb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int col = block + wave * nb; col < a.ncols; col += 4 * nb) {
        int c = col, h = 0;
        while (h < a.nheads && c >= a.npad[h]) { c -= a.npad[h]; ++h; }
        if (h >= a.nheads) return;   // uniform per wave
        float acc = 0.f;
        if (!a.strided[h]) {
            // Wf layout: column c = 16nt + c16 lives at ((nt*G + g)*64 + q*16 + c16) float4s
            const int G = a.D >> 4, nt = c >> 4, c16 = c & 15;
            const f32x4 *w = (const f32x4 *)(a.theta + a.w_off[h]) + (size_t)nt * G * 64 + c16;
            for (int i0 = 0; i0 < G * 4; i0 += 64 * 16) {   // (D = 4096: all 16 loads of a lane in one round trip)
                f32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 64 + lane;
                    v[u] = i < G * 4 ? w[(size_t)(i >> 2) * 64 + (i & 3) * 16] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
            }
        } else {  // mono head V[L][kpad]: column c strided by kpad
            const float *w = a.theta + a.w_off[h];
            for (int l = lane; l < a.rowlen[h]; l += 64) {
                float v = w[l * a.npad[h] + c];
                acc = fmaf(v, v, acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            a.n2[a.n2_off[h] + c] = acc;
            a.gcopy[a.n2_off[h] + c] = a.g_off[h] >= 0 ? a.theta[a.g_off[h] + c] : 1.f;
        }
    }
}

// ---------------------------------------------------------------------------
// proj: Ypart[s][r][c] = sum_{d in slice s} X[r][d] * W[d][c]
//   workgroup = 4 waves, one 32-row tile; wave w owns chunks of 128 d (8 groups of
//   16) and computes the whole [32 x NT*16] tile for them with v_mfma_f32_16x16x4_f32.
//   * all 16 x-loads of a chunk (32 rows x 128 d, 16 KiB per wave) are issued up
//     front, each instruction covering 4 rows x 256 contiguous bytes;
//   * each 64-d half is transposed into MFMA A order through a wave-private LDS tile
//     [32 rows][16 x 16 B], chunk position XOR-swizzled by the row so that both the
//     ds_write_b128 and the ds_read_b128 are bank-conflict free;
//   * the W fragments are contiguous 1 KiB blocks of Wf, prefetched one group ahead.
//   The 4 partial tiles are summed through LDS in a fixed order.
// ---------------------------------------------------------------------------
struct ProjJob {
    int side;              // 0 = src rows, 1 = dst rows (ProjArgs::rows)
    const float *wf;       // Wf tile base: blocks [(nt)*G + g]
    float *ypart;          // chunk base (column offset applied) inside [S][Rpad][npad]
    long long sstride;     // floats between slices (Rpad * npad)
    int nt, npad;
};

// EXTRA SCORING ROWS of a training call (round 5; cfl_pair_train_val_steps_idx_planes): the reference's loop fetches the
// accuracy of a VALIDATION batch in the same sess.run as the training step (cfl/bin/train_dist.py:79-86).  The rows of that
// batch ride in the training step's own projection and row-math launches -- rows [row0, row0 + n) behind the (padded)
// training rows, read from their own resident table by their own index streams -- instead of a second projection +
// row-math launch pair per iteration.  They are forward-only: no dL/dY, no loss, no weight gradient; mid writes their
// scores straight to the caller's buffer.  n == 0 (tile0 = 0): no such rows, every existing path unchanged.
struct RowExtra {
    const float *table;        // resident feature table of the extra rows
    const int *ix[2][2];       // [side][group]: index streams (group 0 = rows [0, bx), group 1 = rows [bx, 2 bx))
    int istride;
    unsigned last_row;
    int row0, n, bx, tile0;    // first row / rows / rows per group / number of 32-row tiles of the extra rows (dispatched first)
};

struct ProjArgs {
    ProjJob job[CFL_MAX_JOBS];
    RowSrc rows[2];
    RowExtra xr;
    int B, R, Rpad, D, S;
    int xcd;  // 1: blockIdx.x enumerates the d slices (see cfl_xcd_aligned)
    NormDev norm;
    int njobs;            // z-slices [0, njobs) project; slice njobs (weight-norm only) computes the column norms
    ColnormArgs cn;
};

// row r of side `side` of a projection launch: a training / scoring row, or -- r >= xr.row0 -- an extra scoring row
__device__ __forceinline__ const float *proj_row_ptr(const ProjArgs &a, int side, int r) {
    if (a.xr.n > 0 && r >= a.xr.row0) {   // (uniform per 8-lane row group; rows past the end are clamped to the last one)
        int e = r - a.xr.row0;
        e = e < a.xr.n ? e : a.xr.n - 1;
        const int g = e >= a.xr.bx ? 1 : 0;
        const int *ip = a.xr.ix[side][g] + (size_t)(e - g * a.xr.bx) * a.xr.istride;
        unsigned t = (unsigned)*ip;
        t = t < a.xr.last_row ? t : a.xr.last_row;
        return a.xr.table + (size_t)t * a.D;
    }
    return row_ptr(a.rows[side], r, a.B, a.R, a.D);
}
// first row of 32-row tile `tile`.  The extra rows' tiles come FIRST in dispatch order (tiles [0, xr.tile0), rows from xr.row0 on),
// the training rows' tiles after them, so that what the projection touched last is what the weight gradient re-reads (measured
// either way: 44.5 us per iteration both -- the +1.3 us of the weight-gradient launch beside extra rows is not cache eviction)
__device__ __forceinline__ int proj_tile_row0(const ProjArgs &a, int tile) {
    return tile < a.xr.tile0 ? a.xr.row0 + tile * 32 : (tile - a.xr.tile0) * 32;
}

template <int NT>
__device__ __forceinline__ void proj_body(const ProjJob &jb, const ProjArgs &a, f32x4 *lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform
    const int r16 = lane & 15, q4 = lane >> 4;  // MFMA: row / k index
    const int rr8 = lane >> 3, ch8 = lane & 7;  // load: row within 8-row group / 16-B chunk
    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  With the d slices fastest an XCD
    // only ever touches 1/8 of the weights (one slice of every column tile) and one d band of x --
    // the same band the weight-gradient launch assigns to it, so part of x is still in that XCD's L2.
    const int row0 = proj_tile_row0(a, a.xcd ? blockIdx.y : blockIdx.x);
    const int s = a.xcd ? blockIdx.x : blockIdx.y;
    const int G = a.D >> 4;           // 16-d groups
    const int NC = (G + 7) >> 3;      // 128-d chunks
    const int nw = a.S * 4, wg = s * 4 + wave;
    const int cbeg = wg * NC / nw, cend = (wg + 1) * NC / nw;  // NC <= 2^16, nw <= 64

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xrow[i] = proj_row_ptr(a, jb.side, row0 + 8 * i + rr8) + 4 * ch8;
    const float *wfl = jb.wf + lane * 4;
    f32x4 *tile = lds + wave * 256;  // 32 rows x 8 chunks of 16 B = 4 KiB per wave
    STAMP(0);

    for (int c = cbeg; c < cend; ++c) {
        const int g0 = c * 8;
        const bool full = G - g0 >= 8;  // uniform; otherwise 4 groups (D % 64 == 0)
        // Loads are issued in consumption order (vmcnt retires in order): W fragments of
        // quarter 0, x of quarter 0, W of quarter 1, x of quarters 1..3; the W fragments of
        // quarters 2 and 3 are issued while quarters 0 and 1 are being multiplied.
        f32x4 bq[2][2][NT], araw[4][4];
        auto loadB = [&](int qq, f32x4 (*dst)[NT]) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    dst[gg][nt] = *(const f32x4 *)(wfl + ((size_t)nt * G + g0 + 2 * qq + gg) * 256);
        };
        auto loadA = [&](int qq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) araw[qq][i] = *(const f32x4 *)(xrow[i] + g0 * 16 + qq * 32);
        };
        // sched_barrier(0) pins the issue order (hipcc otherwise hoists the later quarters)
#if !defined(ABL_PROJ_NOB)
        loadB(0, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !defined(ABL_PROJ_NOA)
        loadA(0);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !defined(ABL_PROJ_NOB)
        loadB(1, bq[1]);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !defined(ABL_PROJ_NOA)
        loadA(1);
        __builtin_amdgcn_sched_barrier(0);
        if (full) {
            loadA(2);
            __builtin_amdgcn_sched_barrier(0);
            loadA(3);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            if (qq >= 2 && !full) break;
            // transpose this 32-d quarter into MFMA A order (wave-private LDS, XOR swizzle)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rr8;
                tile[row * 8 + (ch8 ^ (row & 7))] = norm_apply(araw[qq][i], a.norm, g0 * 16 + qq * 32 + 4 * ch8);
            }
            f32x4 af[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) af[mt][gg] = tile[(mt * 16 + r16) * 8 + ((4 * gg + q4) ^ (r16 & 7))];
#ifdef CFL_STAMPS
            if (qq == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); STAMP(1); }
#endif
#ifndef ABL_PROJ_NOMFMA
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                af[mt][gg][e], bq[qq & 1][gg][nt][e], acc[mt][nt], 0, 0, 0);
#else
            asm volatile("" ::"v"(af[0][0]), "v"(af[0][1]), "v"(af[1][0]), "v"(af[1][1]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bq[qq & 1][0][nt]), "v"(bq[qq & 1][1][nt]));
#endif
#if !defined(ABL_PROJ_NOB)
            if (qq < 2 && full) loadB(qq + 2, bq[qq & 1]);
#endif
            STAMP(2 + qq);
        }
    }

    // cross-wave sum: lds[wave][tile][lane]
    __syncthreads();
    STAMP(6);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds[(wave * 2 * NT + mt * NT + nt) * 64 + lane] = acc[mt][nt];
    __syncthreads();
    for (int t = wave; t < 2 * NT; t += 4) {
        const int mt = t / NT, nt = t % NT;
        f32x4 sum = lds[(0 * 2 * NT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += lds[(w * 2 * NT + t) * 64 + lane];
        // C layout: col = lane&15, rows 4*(lane>>4) .. +3  ->  Ypart[s][row][npad] (row-major:
        // the mid kernel then reads whole rows with 16-byte loads)
        float *dst = jb.ypart + (size_t)s * jb.sstride + (size_t)(row0 + mt * 16 + 4 * q4) * jb.npad +
                     nt * 16 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(size_t)e * jb.npad] = sum[e];   // (default policy: non-temporal stores here cost mid +0.9 us)
    }
    STAMP(7);
}

// ---------------------------------------------------------------------------
// proj, bf16x3 on the chunk-at-a-time skeleton (round 4): cfl_proj_bx3_kernel.
// At B <= 1024 the exact-fp32 projection above is bound by the fp32 matrix pipe wherever the heads are wide or the rows
// many (config 4: 15 us of v_mfma_f32_16x16x4_f32 at peak in a 21 us launch; headline: 6.8 us per SIMD of 14); the
// LDS-shared form (cfl_proj_x3_kernel) needs >= 512 work units of 128 rows to fill the chip and a d split that costs
// `mid` more than it saves below ~3000 rows per side.  This form keeps everything that shapes the launch -- 32-row tiles,
// one 128-d chunk per wave, S, the XCD-aligned order, the slab layout -- and swaps the arithmetic only:
//   * B operand = the KEPT bf16 planes of theta (CflThetaPlanes: written by the Adam tail of the previous step, round-to-
//     nearest split), fetched per wave as 1 KiB blocks like the fp32 fragments they replace (6 instead of 4 bytes per
//     weight from L2; no split of W anywhere in the step);
//   * A operand = the wave's 32 x 32 quarter of x, parked in the wave-private LDS tile as before, read back as whole
//     128-byte rows and split round-to-nearest in the VALU slots of the matrix pipe (16 values per lane and quarter);
//   * SIX partial products (ah bh, ah bm, am bh, ah bl, al bh, am bm) per 16x16x32 block on v_mfma_f32_16x16x32_bf16:
//     96 matrix-pipe cycles per block against 256 for the eight fp32 MFMAs; with round-to-nearest parts the dropped terms
//     are <= 2^-23 |ab| and zero-mean -- the size of one fp32 rounding (tests: error against the float64
//     oracle within 2x of the exact-fp32 form's).
// Selected by the plan when the caller keeps planes (the fused single-GPU training step); every other call keeps the
// exact-fp32 kernel.
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void proj_body_bx3(const ProjJob &jb, const ProjArgs &a, f32x4 *lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int rr8 = lane >> 3, ch8 = lane & 7;
    const int row0 = proj_tile_row0(a, a.xcd ? blockIdx.y : blockIdx.x);
    const int s = a.xcd ? blockIdx.x : blockIdx.y;
    const int G = a.D >> 4, Q = a.D >> 5;
    const int NC = (G + 7) >> 3;
    const int nw = a.S * 4, wg = s * 4 + wave;
    const int cbeg = wg * NC / nw, cend = (wg + 1) * NC / nw;

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xrow[i] = proj_row_ptr(a, jb.side, row0 + 8 * i + rr8) + 4 * ch8;
    const unsigned short *pll = (const unsigned short *)jb.wf + lane * 8;   // planes of this job's first column tile
    f32x4 *tile = lds + wave * 256;

    for (int c = cbeg; c < cend; ++c) {
        const int t0 = c * 4;                      // first 32-d quarter of the chunk
        const bool full = Q - t0 >= 4;             // otherwise 2 quarters (D % 64 == 0)
        bf16x8 bq[2][NT][3];
        f32x4 araw[4][4];
        auto loadB = [&](int qq, bf16x8 (*dst)[3]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    dst[nt][pl] = *(const bf16x8 *)(pll + ((size_t)(nt * Q + t0 + qq) * 3 + pl) * 512);
        };
        auto loadA = [&](int qq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) araw[qq][i] = *(const f32x4 *)(xrow[i] + (t0 + qq) * 32);
        };
        loadB(0, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
        loadA(0);
        __builtin_amdgcn_sched_barrier(0);
        loadB(1, bq[1]);
        __builtin_amdgcn_sched_barrier(0);
        loadA(1);
        __builtin_amdgcn_sched_barrier(0);
        if (full) {
            loadA(2);
            __builtin_amdgcn_sched_barrier(0);
            loadA(3);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            if (qq >= 2 && !full) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rr8;
                tile[row * 8 + (ch8 ^ (row & 7))] = norm_apply(araw[qq][i], a.norm, (t0 + qq) * 32 + 4 * ch8);
            }
            bf16x8 af[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = mt * 16 + i16;
                const f32x4 c0 = tile[row * 8 + ((2 * kq) ^ (row & 7))], c1 = tile[row * 8 + ((2 * kq + 1) ^ (row & 7))];
                float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#ifdef CFL_BX3_A_RNE
                split_frag_rne(v, af[mt]);
#else
                split_frag(v, af[mt]);
#endif
            }
            // six partial products, small terms first; consecutive MFMAs hit different accumulators
#define CFL_BX3(LA, LB)                                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][LA], bq[qq & 1][nt][LB], acc[mt][nt], 0, 0, 0);
            CFL_BX3(1, 1) CFL_BX3(2, 0) CFL_BX3(0, 2) CFL_BX3(1, 0) CFL_BX3(0, 1) CFL_BX3(0, 0)
#undef CFL_BX3
            if (qq < 2 && full) loadB(qq + 2, bq[qq & 1]);
        }
    }

    // cross-wave sum and slab store: identical to proj_body
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds[(wave * 2 * NT + mt * NT + nt) * 64 + lane] = acc[mt][nt];
    __syncthreads();
    for (int t = wave; t < 2 * NT; t += 4) {
        const int mt = t / NT, nt = t % NT;
        f32x4 sum = lds[(0 * 2 * NT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += lds[(w * 2 * NT + t) * 64 + lane];
        float *dst = jb.ypart + (size_t)s * jb.sstride + (size_t)(row0 + mt * 16 + 4 * kq) * jb.npad + nt * 16 + i16;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(size_t)e * jb.npad] = sum[e];
    }
}

// ---------------------------------------------------------------------------
// proj, streaming form: the same contraction for waves that own SEVERAL 128-d chunks (S <= 2: 2048 rows per
// side and more, and every dist_eval / dist_predict call).  proj_body above issues the 16 x loads of a chunk,
// waits for them and multiplies, chunk after chunk: with one chunk per wave (the training step at B = 512) that
// is all there is to overlap, with eight it leaves the matrix pipe idle for a memory latency per chunk (B = 8192:
// 162 us for 13 GF = 51 % of the fp32-MFMA roof).  Here the x registers of a quarter (32 d) are refilled with the
// same quarter of the NEXT chunk the moment they have been parked in LDS, i.e. before that quarter's MFMAs, so
// there are always 3-4 quarters (12-16 KiB per wave) of x in flight behind the one being multiplied; the W
// fragments of quarter t+2 are requested after the MFMAs of quarter t (two register sets, as before).
// The steady-state body is BRANCH-FREE (prefetch indices are clamped to the last quarter of the row instead of
// being guarded): with the loads inside uniform branches the compiler's waitcnt pass has to assume the path on
// which nothing was issued and drains the queue (s_waitcnt vmcnt(0)) in front of every quarter -- measured: no
// gain at all over proj_body.  Arithmetic, accumulation order and output are those of proj_body bit for bit
// (k-ordered fp32 FMA chains per wave, waves summed in wave order).
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void proj_stream_body(const ProjJob &jb, const ProjArgs &a, f32x4 *lds, int rowtile) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q4 = lane >> 4;
    const int rr8 = lane >> 3, ch8 = lane & 7;
    const int row0 = rowtile * 32;
    const int s = a.xcd ? blockIdx.x : blockIdx.y;
    const int G = a.D >> 4;
    const int NC = (G + 7) >> 3;
    const int nw = a.S * 4, wg = s * 4 + wave;
    const int cbeg = wg * NC / nw, cend = (wg + 1) * NC / nw;
    const int qlast = (G >> 1) - 1;   // last 32-d quarter of a row (D % 64 == 0)

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xrow[i] = row_ptr(a.rows[jb.side], row0 + 8 * i + rr8, a.B, a.R, a.D) + 4 * ch8;
    const float *wfl = jb.wf + lane * 4;
    f32x4 *tile = lds + wave * 256;

    f32x4 bq[2][2][NT] = {}, araw[4][4] = {}, af[2][2];
    auto loadB = [&](int tq, f32x4 (*dst)[NT]) {   // tq clamped: a prefetch past the row re-reads its last quarter
#ifndef ABL_PROJ_NOB
        tq = tq < qlast ? tq : qlast;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                dst[gg][nt] = *(const f32x4 *)(wfl + ((size_t)nt * G + 2 * tq + gg) * 256);
#endif
    };
    auto loadA = [&](int tq, f32x4 *dst) {
#ifndef ABL_PROJ_NOA
        tq = tq < qlast ? tq : qlast;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *(const f32x4 *)(xrow[i] + tq * 32);
#endif
    };
    auto park = [&](int tq, const f32x4 *src) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rr8;
            tile[row * 8 + (ch8 ^ (row & 7))] = norm_apply(src[i], a.norm, tq * 32 + 4 * ch8);
        }
    };
    auto frags = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) af[mt][gg] = tile[(mt * 16 + r16) * 8 + ((4 * gg + q4) ^ (r16 & 7))];
    };
    auto mfmas = [&](const f32x4 (*fb)[NT]) {
#ifdef ABL_PROJ_NOMFMA
        asm volatile("" ::"v"(af[0][0]), "v"(af[0][1]), "v"(af[1][0]), "v"(af[1][1]));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(fb[0][nt]), "v"(fb[1][nt]));
        return;
#endif
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][gg][e], fb[gg][nt][e], acc[mt][nt], 0, 0, 0);
    };

    if (cbeg < cend) {   // (a wave without a chunk only takes part in the sum below)
        // prologue: first chunk, in consumption order
        const int t0 = cbeg * 4;
        loadB(t0, bq[0]);        __builtin_amdgcn_sched_barrier(0);
        loadA(t0, araw[0]);      __builtin_amdgcn_sched_barrier(0);
        loadB(t0 + 1, bq[1]);    __builtin_amdgcn_sched_barrier(0);
        loadA(t0 + 1, araw[1]);  __builtin_amdgcn_sched_barrier(0);
        loadA(t0 + 2, araw[2]);  __builtin_amdgcn_sched_barrier(0);
        loadA(t0 + 3, araw[3]);  __builtin_amdgcn_sched_barrier(0);
        const int clast = cend - 1;
        for (int c = cbeg; c < clast; ++c) {   // every chunk but the last: 4 full quarters, next chunk prefetched
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int tq = 4 * c + qq;
                park(tq, araw[qq]);
                loadA(tq + 4, araw[qq]);
                __builtin_amdgcn_sched_barrier(0);
                frags();
                mfmas(bq[qq & 1]);
                __builtin_amdgcn_sched_barrier(0);
                loadB(tq + 2, bq[qq & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // last chunk: 4 quarters, or 2 when D % 128 == 64 and it is the row's last
        const int tl = 4 * clast;
        const bool full = qlast - tl >= 3;   // uniform
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            park(tl + qq, araw[qq]);
            frags();
            mfmas(bq[qq & 1]);
            __builtin_amdgcn_sched_barrier(0);
            loadB(tl + qq + 2, bq[qq & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (full) {
#pragma unroll
            for (int qq = 2; qq < 4; ++qq) {
                park(tl + qq, araw[qq]);
                frags();
                mfmas(bq[qq & 1]);
            }
        }
    }

    // cross-wave sum and store: identical to proj_body
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds[(wave * 2 * NT + mt * NT + nt) * 64 + lane] = acc[mt][nt];
    __syncthreads();
    for (int t = wave; t < 2 * NT; t += 4) {
        const int mt = t / NT, nt = t % NT;
        f32x4 sum = lds[(0 * 2 * NT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += lds[(w * 2 * NT + t) * 64 + lane];
        float *dst = jb.ypart + (size_t)s * jb.sstride + (size_t)(row0 + mt * 16 + 4 * q4) * jb.npad +
                     nt * 16 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(size_t)e * jb.npad] = sum[e];
    }
}

extern "C" __global__ __launch_bounds__(256) void cfl_proj_stream_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    const int rowtile = a.xcd ? blockIdx.y : blockIdx.x;
    const ProjJob &jb = a.job[blockIdx.z];
    switch (jb.nt) {
        case 0: {
            colnorm_columns(a.cn, (int)(blockIdx.y * gridDim.x + blockIdx.x), gridDim.x * gridDim.y);
            break;
        }
        case 1: proj_stream_body<1>(jb, a, lds, rowtile); break;
        case 2: proj_stream_body<2>(jb, a, lds, rowtile); break;
        case 3: proj_stream_body<3>(jb, a, lds, rowtile); break;
        default: proj_stream_body<4>(jb, a, lds, rowtile); break;
    }
}

// one 1 KiB LDS-DMA piece: lane l's 16 bytes at gsrc land at lds_dst + 16 l (lds_dst wave-uniform byte address)
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ---------------------------------------------------------------------------
// proj, bf16x3 form with W planes shared through LDS (large row counts; round 3).
// The exact-fp32 forms above sit under both of their roofs at once (41-61 us of fp32 matrix-core time and 43-64 us of x
// arrival per 8192-pair call, profiles/r03_proj_forms.md).  This form halves the matrix time without giving up fp32
// accuracy: every fp32 operand is split exactly into three bf16 values (split3) and a product is accumulated in fp32 from
// EIGHT of the nine partial products on v_mfma_f32_16x16x32_bf16 (only a_l * b_l, <= 2^-32 |ab|, is dropped -- the
// weight gradient drops three; the distances feed exp(), so the forward keeps two more): 8 x 16 cycles per 16x16x32
// product block against 8 x 32 cycles for the eight v_mfma_f32_16x16x4_f32 it replaces.
//   * W is split ONCE per call by cfl_wplanes_kernel into bf16 planes in the B-fragment order of the 16x16x32 MFMA
//     (one 1 KiB block per (column tile, 32-d quarter, plane)): no operand splitting of W inside the loop;
//   * a workgroup = 4 waves that own 32 rows each of a 128-row tile and walk the SAME d slice, so the W planes of a
//     step are fetched once per workgroup -- by LDS-DMA (global_load_lds_dwordx4), each wave issuing a share of the
//     pieces three steps ahead into a 4-slot ring (counted vmcnt + one raw s_barrier per step) -- instead of once per
//     wave from L2 (2 bytes of W per byte of x in the forms above, 0.75 here);
//   * x stays on the register path of the streaming form (ring of four quarters, refilled as soon as a quarter has
//     been parked in the wave-private LDS tile): 16 KiB per wave in flight, more than LDS could hold;
//   * the A fragments (16 rows x 32 d = whole 128-byte rows) are read back from the tile, split in the VALU slots the
//     MFMAs leave free, and multiplied.
// Work units (column job, 128-row tile, d slice) as in the ring form; two workgroups per CU.
// ---------------------------------------------------------------------------
#define PX3_SLOTS 4
#define PX3_AHEAD 3                                  // W planes are requested three steps ahead
#define PX3_SLOT_USHORTS (4 * 3 * 512)               // up to 4 column tiles x 3 planes x 1 KiB
#define PX3_LDS_BYTES (PX3_SLOTS * PX3_SLOT_USHORTS * 2 + 4 * 4096)

struct Px3Args {
    ProjJob job[CFL_MAX_JOBS];                       // wf = the job's PLANES base (ushort units, see cfl_wplanes_kernel)
    int order[CFL_MAX_JOBS];
    RowSrc rows[2];
    int B, R, D, S, njobs;
    int tiles, nunits, nwg, Kq;                      // Kq = 32-d quarters per slice
    NormDev norm;
    int ncn;                                         // weight-norm: workgroups [0, ncn) of the launch compute the column norms
    ColnormArgs cn;                                  // (dispatched first, short; `mid` is their first consumer)
};

// W (fragment-major fp32, Wf[nt][g][q][c16][e]) -> planes[((nt * Q + tq) * 3 + p) * 512 + lane * 8 + j]:
// bf16 level p of W[d = 32 tq + 8 (lane >> 4) + j][col = 16 nt + (lane & 15)]  (B operand of v_mfma_f32_16x16x32_bf16)
struct WPlanesArgs { const float *wf[2]; unsigned short *planes[2]; int ntiles[2]; int G; };
__global__ __launch_bounds__(256) void cfl_wplanes_kernel(WPlanesArgs w) {   // both sides' heads in one launch
    const int Q = w.G >> 1;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n0 = (long long)w.ntiles[0] * Q * 64;
    const int sd = i >= n0 ? 1 : 0;
    if (sd) i -= n0;
    if (i >= (long long)w.ntiles[sd] * Q * 64) return;
    const float *wf = sd ? w.wf[1] : w.wf[0];
    unsigned short *planes = sd ? w.planes[1] : w.planes[0];
    const int G = w.G;
    const int lane = (int)(i & 63);
    const int tq = (int)((i >> 6) % Q), nt = (int)((i >> 6) / Q);
    const int n = lane & 15, kq = lane >> 4;
    const int g = 2 * tq + (kq >> 1), q0 = 2 * (kq & 1);
    const float *src = wf + ((size_t)nt * G + g) * 256 + (q0 * 16 + n) * 4;
    const f32x4 v0 = *(const f32x4 *)src, v1 = *(const f32x4 *)(src + 64);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    bf16x8 f[3];
    split_frag_rne(v, f);
    unsigned short *dst = planes + ((size_t)(nt * Q + tq) * 3) * 512 + lane * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) *(bf16x8 *)(dst + p * 512) = f[p];
}

template <int NT, bool KEEP>   // KEEP: x loaded with the default cache policy (training: the weight gradient re-reads it from the Infinity Cache)
__device__ __forceinline__ void px3_unit(const Px3Args &a, const ProjJob &jb, int tile, int slice, char *smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int rr8 = lane >> 3, ch8 = lane & 7;
    const int Q = a.D >> 5;                       // quarters per row
    const int t0 = slice * a.Kq;                  // first quarter of the slice
    constexpr int NP = NT * 3;                    // W pieces (1 KiB) per step
    // pieces of a step dealt round-robin to the 4 waves: wave w issues pieces w, w + 4, ... < NP
    constexpr int PMAX = (NP + 3) / 4;
    const int mine = (NP - wave + 3) / 4;         // this wave's pieces per step (uniform per wave)
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char *)smem;
    f32x4 *xt = (f32x4 *)(smem + PX3_SLOTS * PX3_SLOT_USHORTS * 2) + wave * 256;    // wave-private transpose tile
    const unsigned short *wbase = (const unsigned short *)jb.wf;                    // planes of this job's tiles

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int row0 = tile * 128 + wave * 32;
    const float *xrow[4];
    {
        const RowSrc rs = jb.side ? a.rows[1] : a.rows[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) xrow[i] = row_ptr(rs, row0 + 8 * i + rr8, a.B, a.R, a.D) + t0 * 32 + 4 * ch8;
    }
    // piece k of step q: block (nt = k / 3, plane = k % 3) of quarter t0 + q; LDS slot layout = the same block order
    auto issueW = [&](int q) {
        const int qq = q < a.Kq ? q : a.Kq - 1;   // past the end: re-fetch the last step (keeps the counted waits exact)
        const unsigned sb = lds0 + (unsigned)((q % PX3_SLOTS) * PX3_SLOT_USHORTS) * 2;
#pragma unroll
        for (int j = 0; j < PMAX; ++j) {
            const int k = wave + 4 * j;
            if (k < NP) {
                const int nt = k / 3, pl = k - 3 * nt;
                glds16((const float *)(wbase + ((size_t)(nt * Q + t0 + qq) * 3 + pl) * 512 + lane * 8), sb + k * 1024);
            }
        }
    };
    f32x4 araw[4][4];
    auto loadA = [&](int q, f32x4 *dst) {
        const int qq = q < a.Kq ? q : a.Kq - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // nt: x is read once by this launch -- keep it from displacing the W planes in L2
            if (KEEP) dst[i] = *(const f32x4 *)(xrow[i] + qq * 32);
            else dst[i] = __builtin_nontemporal_load((const f32x4 *)(xrow[i] + qq * 32));   // (measured: -6 % scoring, -9 % at B = 8192)
        }
    };
    // prologue: W of steps 0 .. AHEAD-1, x of steps 0 .. 3 (consumption order); own W pieces of step 0 landed, barrier
#pragma unroll
    for (int q = 0; q < PX3_AHEAD; ++q) issueW(q);
#pragma unroll
    for (int q = 0; q < 4; ++q) { loadA(q, araw[q]); __builtin_amdgcn_sched_barrier(0); }
    // everything issued after this wave's W(0) pieces may stay in flight: W(1), W(2) and the 16 x loads
    {
        const int later = 2 * mine + 16;
        if (later == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (later == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (later == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // (mine == 0: nothing of its own to wait for)
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int q0 = 0; q0 < a.Kq; q0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u;                 // (Kq is a multiple of 4: whole 128-d chunks per slice)
            // W planes three steps ahead, then this quarter of x: park, refill the registers with the quarter 4 ahead
            issueW(q + PX3_AHEAD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rr8;
                xt[row * 8 + (ch8 ^ xt_sw2(row))] = norm_apply(araw[u][i], a.norm, (t0 + q) * 32 + 4 * ch8);
            }
            loadA(q + 4, araw[u]);
            __builtin_amdgcn_sched_barrier(0);
            // fragments: A = rows 16 mt + i16, d = 8 kq .. 8 kq + 7 (two 16-byte chunks of the parked row), split here;
            // B = planes from the shared slot (lane-linear 16 bytes per block)
            bf16x8 af[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = mt * 16 + i16;
                const f32x4 c0 = xt[row * 8 + ((2 * kq) ^ xt_sw2(row))], c1 = xt[row * 8 + ((2 * kq + 1) ^ xt_sw2(row))];
                float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                split_frag(v, af[mt]);
            }
            const bf16x8 *ws = (const bf16x8 *)(smem + (size_t)(q % PX3_SLOTS) * PX3_SLOT_USHORTS * 2) + lane;
            bf16x8 bf[NT][3];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = ws[(nt * 3 + pl) * 64];
            // six partial products (round 4; eight until the W planes were split round-to-nearest: with one rounded operand the
            // dropped cross terms x_m w_l + x_l w_m + x_l w_l are zero-mean and <= 2^-22 |x w|, as in cfl_proj_bx3_kernel),
            // small terms first; consecutive MFMAs hit different accumulators
#define PX3_MM(LA, LB)                                                                                        \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][LA], bf[nt][LB], acc[mt][nt], 0, 0, 0);
#ifdef CFL_PX3_EIGHT
            PX3_MM(2, 1) PX3_MM(1, 2)
#endif
            PX3_MM(1, 1) PX3_MM(2, 0) PX3_MM(0, 2) PX3_MM(1, 0) PX3_MM(0, 1) PX3_MM(0, 0)
#undef PX3_MM
            // own W pieces of step q + 1 have landed (issued at step q - 2: W(q+2), W(q+3) and 4 x quarters are younger)
            // (younger in the queue: x(q+2), W(q+2), x(q+3), W(q+3), x(q+4) = 12 loads + 2 * mine pieces)
            {
                const int later = 2 * mine + 12;
                if (later == 14) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
                else if (later == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                else if (later == 18) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetched tail pieces / quarters: nothing may land after the unit
    __builtin_amdgcn_s_barrier();
    // C layout: col = lane & 15, rows 4 (lane >> 4) .. + 3  ->  Ypart[slice][row][npad]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float *dst = jb.ypart + (size_t)slice * jb.sstride + (size_t)(row0 + mt * 16 + 4 * kq) * jb.npad + nt * 16 + i16;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)e * jb.npad] = acc[mt][nt][e];
        }
}

#define CFL_PROJ_X3_KERNEL(NAME, KEEP)                                                              \
    extern "C" __global__ __launch_bounds__(256, 2) void NAME(Px3Args a_) {                             \
        CFL_KERNARG_IN_PLACE(Px3Args, a, a_);                                                           \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                     \
        if ((int)blockIdx.x < a.ncn) {                                                                  \
            colnorm_columns(a.cn, (int)blockIdx.x, a.ncn);                                            \
            return;                                                                                     \
        }                                                                                               \
        const int w = blockIdx.x - a.ncn;                                                               \
        for (int i = 0;; ++i) {                                                                         \
            /* snake order over the heavy-to-light unit list (as the ring form) */                      \
            const int base = (i >> 1) * 2 * a.nwg;                                                      \
            const int uid = (i & 1) ? base + 2 * a.nwg - 1 - w : base + w;                              \
            if (uid >= a.nunits) break;                                                                 \
            const int per_job = a.tiles * a.S;                                                          \
            const int job = a.order[uid / per_job], rem = uid % per_job;                                \
            const int tile = rem / a.S, slice = rem % a.S;                                              \
            const ProjJob &jb = a.job[job];                                                             \
            switch (jb.nt) {                                                                            \
                case 1: px3_unit<1, KEEP>(a, jb, tile, slice, smem); break;                             \
                case 2: px3_unit<2, KEEP>(a, jb, tile, slice, smem); break;                             \
                case 3: px3_unit<3, KEEP>(a, jb, tile, slice, smem); break;                             \
                default: px3_unit<4, KEEP>(a, jb, tile, slice, smem); break;                            \
            }                                                                                           \
        }                                                                                               \
    }
CFL_PROJ_X3_KERNEL(cfl_proj_x3_kernel, false)        // scoring, and training batches larger than the Infinity Cache
CFL_PROJ_X3_KERNEL(cfl_proj_x3_keep_kernel, true)    // training: x stays cached for the weight gradient
#undef CFL_PROJ_X3_KERNEL

extern "C" __global__ __launch_bounds__(256) void cfl_proj_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    const ProjJob &jb = a.job[blockIdx.z];
    switch (jb.nt) {
        case 0: {   // the colnorm slice (marked by nt == 0: no kernel-argument load of its own in front of the dispatch)
            const int nb = gridDim.x * gridDim.y;
            colnorm_columns(a.cn, (int)(blockIdx.y * gridDim.x + blockIdx.x), nb);
            break;
        }
        case 1: proj_body<1>(jb, a, lds); break;
        case 2: proj_body<2>(jb, a, lds); break;
        case 3: proj_body<3>(jb, a, lds); break;
        default: proj_body<4>(jb, a, lds); break;
    }
}

extern "C" __global__ __launch_bounds__(256, 2) void cfl_proj_bx3_kernel(ProjArgs a) {   // ProjJob::wf = the job's kept planes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = (f32x4 *)smem;
    const ProjJob &jb = a.job[blockIdx.z];
    switch (jb.nt) {
        case 0: {   // the colnorm slice (reads the fp32 weights through a.cn)
            const int nb = gridDim.x * gridDim.y;
            colnorm_columns(a.cn, (int)(blockIdx.y * gridDim.x + blockIdx.x), nb);
            break;
        }
        case 1: proj_body_bx3<1>(jb, a, lds); break;
        case 2: proj_body_bx3<2>(jb, a, lds); break;
        case 3: proj_body_bx3<3>(jb, a, lds); break;
        default: proj_body_bx3<4>(jb, a, lds); break;
    }
}

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
    int spin_limit;         // polls before a hand-off is declared lost (CFL_HANDOFF_SPIN_LIMIT; < 0: at once -- the failure test)
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
};

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
__device__ __forceinline__ void fuse_apply(const GradFuse &f, long long off, f32x4 gr, f32x4 &th, f32x4 mm, f32x4 vv) {   // th: updated in place (the planes are split from it)
    if (f.reg_const != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gr[e] = fmaf(f.reg_const, th[e], gr[e]);
    }
    *(f32x4 *)(f.grad + off) = gr;
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
            // Bounded: ~2^22 polls with s_sleep is > 100 ms, four orders of magnitude beyond any hand-off of a healthy
            // launch.  The waits are for workgroups dispatched BEFORE this one (smaller linear id), which never wait
            // themselves, so a time-out means the dispatch-order assumption or the visibility protocol failed.
            int ok = f.spin_limit < 0 ? 0 : 1;
            if (expect > 0 && ok) {
                int spins = 0;
                while (__hip_atomic_load(f.flag + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > f.spin_limit) { ok = 0; break; }
                }
            }
            if (f.wn && ok) {   // the c_j column sums of this launch's reduction blocks (dispatched first, short)
                int spins = 0;
                while (__hip_atomic_load(f.red_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < f.red_expect) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > f.spin_limit) { ok = 0; break; }
                }
            }
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
__device__ __forceinline__ void fuse_apply1(const GradFuse &f, long long off, float gr, bool reg) {
    float th = f.theta[off];
    if (reg && f.reg_const != 0.f) gr = fmaf(f.reg_const, th, gr);
    f.grad[off] = gr;
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
                            fuse_apply1(f, f.thr_off, th >= CFL_THR_FLOOR ? lds[64 + P_DTHR] : 0.f, false);
                        } else {
                            fuse_apply1(f, f.thr_off + lane, 0.f, false);   // rest of the 64-float threshold slot
                        }
                        float rs = 0.f;
                        for (int b = lane; b < f.nregblocks; b += 64) rs += f.regpart[b];
                        rs = wave_sum(rs);
                        if (lane == 0) {
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
                        fuse_apply1(f, f.red_b[k] + c, c < f.red_n[k] ? csum : 0.f, true);
                    } else if (idx == 0) {
                        // pad of the bias array up to its 64-float slot: zero gradient
                        const int cp = f.red_npad[k] + lane - 16;
                        if (cp < ((f.red_npad[k] + 63) & ~63)) fuse_apply1(f, f.red_b[k] + cp, 0.f, true);
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
                        fuse_apply1(f, f.red_g[k] + c, gr, false);
                    } else if (idx == 0) {
                        const int cp = f.red_npad[k] + lane - 16;
                        if (cp < ((f.red_npad[k] + 63) & ~63)) fuse_apply1(f, f.red_g[k] + cp, 0.f, false);
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
                    *(f32x4 *)(f.grad + off) = gr;
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
                        if (threadIdx.x == 0) fuse_apply1(f, f.mono_w + (long long)idx * rr.kpad + kk, g1, f.mono_reg != 0);
                    } else if (threadIdx.x == 0) {
                        const float n2 = f.mono_n2[kk];
                        fuse_apply1(f, f.mono_g + kk, n2 > 0.f ? tot / sqrtf(n2) : 0.f, false);
                    }
                }
            }
            if (fin && wave == 0) {
                // the padding of the owned entries: columns K .. kpad of the row (kind 1), and -- last row / kind 2 -- the
                // rest of the region up to its 64-float boundary: zero gradient (+ L2 of a zero weight)
                const GradFuse &f = a.fuse;
                if (rr.kind == 1) {
                    for (int kk = rr.K + lane; kk < rr.kpad; kk += 64) fuse_apply1(f, f.mono_w + (long long)idx * rr.kpad + kk, 0.f, f.mono_reg != 0);
                    if (idx == f.mono_L - 1) {
                        const long long used = (long long)f.mono_L * rr.kpad, end = (used + 63) / 64 * 64;
                        for (long long o = used + lane; o < end; o += 64) fuse_apply1(f, f.mono_w + o, 0.f, f.mono_reg != 0);
                    }
                } else {
                    for (int kk = rr.K + lane; kk < ((rr.kpad + 63) & ~63); kk += 64) fuse_apply1(f, f.mono_g + kk, 0.f, false);
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
                split_frag(v, af);
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

template <int NT, bool HO, int NW = 4>   // NW: waves per workgroup (8: two waves per SIMD); HO: row split and / or siamese pairing (hand-off tail); false: the tile is complete in the workgroup
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
                split_frag(v, af);
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
            if (expect > 0 && ok) {
                int spins = 0;
                while (__hip_atomic_load(f.flag + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > f.spin_limit) { ok = 0; break; }
                }
            }
            if (f.wn && ok) {   // the c_j column sums of this launch's reduction blocks (dispatched first, short)
                int spins = 0;
                while (__hip_atomic_load(f.red_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < f.red_expect) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > f.spin_limit) { ok = 0; break; }
                }
            }
            ((int *)lds)[0] = ok;
            if (!ok) f.scalars[CFL_S_ERROR] = 1.f;   // sticky error word (see CFL_HANDOFF_SPIN_LIMIT)
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
            fuse_apply(f, base + h * 64, gr, th[h], mm[h], vv[h]);
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

// ---------------------------------------------------------------------------
// mid: per pair row: slice-sum, head epilogue, distance, loss, dL/dY.
//   workgroup = 64 threads = 4 rows x 16 column parts; lane (p = tid&15, j = tid>>4)
//   owns the latent coordinates l == p (mod 16) of row blockIdx.x*4 + j, for every
//   prototype k.
//   phase 1: the block's 4 rows of every slice slab are read with coalesced
//            16-byte loads (all slices in flight at once), summed in slice order
//            and parked in LDS as Y[row][col]; biases / weight-norm scales / gate
//            weights are staged in LDS in the same round of loads;
//   phase 2: per-lane math; sums over l are completed with xor-shuffles inside
//            the 16-lane row group.  Per-lane runtime-indexed state lives in LDS
//            as [slot][64].
//   dL/dy is written UNSCALED in fragment-major order (dYf) -- the weight-norm /
//   input scale is applied to the finished weight gradient by finalize -- so its
//   column sums are the bias gradients.  Everything else that needs a sum over
//   rows is written as one more fragment-major tile; the reductions ride in the
//   grad launch (grad_red_block).
// ---------------------------------------------------------------------------
#define MID_RB 4

struct MidSide {
    const float *ypart;   // [S][Rpad][npad]
    long long sstride;
    const float *b;       // biases or null
    const float *g;       // wn gains or null
    const float *n2;      // wn squared column norms or null
    float *dyf;           // fragment-major dL/dy (unscaled)
    float *cwf;           // fragment-major dL/dy * (x_hat.V) (weight-norm gain rows) or null
    int n, npad;
    int is_proto;         // 1: columns are k*L + l ; 0: columns are l
};

struct MidArgs {
    MidSide side[2];      // 0 = src, 1 = dst
    const float *mono_w, *mono_g, *mono_n2;  // monomer gate head V[L][kpad]
    float *mono_ya, *mono_du;                // row-major [Rpad][lpad], [Rpad][kpad]
    float *mono_duc;                         // row-major [Rpad][kpad] (weight-norm)
    int kpad, lpad;
    int S, L, K, Lq, dist_type, act, weight_norm;
    float in_mul;
    const float *thr;
    int B, R, Rpad;
    int train, use_threshold;
    float pos_weight, caffe_margin, lambda_m;
    float *scores, *dists;
    float *rowqf;         // fragment-major tile of the per-row loss quantities
    float *thr_copy;      // max(thr, 1e-6) of this step (read by finalize's scalar block)
    int *zero_i;          // hand-off tickets / flags of the fused weight-gradient launch: cleared here, every step
    int nzero;
    int nrb, ys;          // row blocks; LDS row stride of Y (floats)
    // extra scoring rows of a training call (RowExtra): rows [xrow0, xrow0 + xn) of the partial slabs, forward only, scores to
    // xscores[0 .. xn); nxb = their row blocks (wave-per-row kernels only), dispatched behind the training rows' blocks
    int xrow0, xn, nxb;
    float *xscores;
    // regulariser blocks
    const float *theta;
    float *regpart;
    int nreg_ranges;
    long long reg_off[CFL_MAX_REGIONS], reg_cnt[CFL_MAX_REGIONS];
    long long reg_total_groups;  // number of 64-float groups over all ranges
};

__device__ __forceinline__ float act_fn(float y, int act) {
    switch (act) {
        case CFL_ACT_SIGMOID: return 1.f / (1.f + expf(-y));
        case CFL_ACT_TANH: return tanhf(y);
        case CFL_ACT_RELU: return fmaxf(y, 0.f);
        default: return y;
    }
}
__device__ __forceinline__ float act_grad(float a, int act) {
    switch (act) {
        case CFL_ACT_SIGMOID: return a * (1.f - a);
        case CFL_ACT_TANH: return 1.f - a * a;
        case CFL_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}
// all-reduce over the 16 column parts of a row (one DPP row): pure VALU, no LDS crossbar.
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror form a butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float sum_p(float x) {
    x = dpp_add<0xB1>(x);   // lane ^ 1
    x = dpp_add<0x4E>(x);   // lane ^ 2
    x = dpp_add<0x141>(x);  // other quad of the 8-lane half
    x = dpp_add<0x140>(x);  // other half of the 16-lane row
    return x;
}
// sum over the 64 lanes, the same value in every lane: four DPP adds inside the 16-lane rows, then the four row sums
// through v_readlane (SGPRs).  The wave-per-row mid kernel IS its latency chain (tools/mid_stamp_probe.py); wave_sum's six
// ds_bpermute round trips through the LDS crossbar were ~450-750 cycles of it.
__device__ __forceinline__ float wave_sum_dpp(float x) {
    const int r = __float_as_int(sum_p(x));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(r, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(r, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(r, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(r, 48));
    return (r0 + r1) + (r2 + r3);
}
// 1-ulp hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32): the
// per-row math is one wave per block, so its instruction count is its latency.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float flog1pexp(float negabs) { return __logf(1.f + __expf(negabs)); }

__device__ void mid_reg_block(const MidArgs &a, int blk) {
    // 64 threads; block handles 64 groups of 64 floats of the regularised ranges
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int gi = 0; gi < 64; ++gi) {
        long long g = (long long)blk * 64 + gi;
        if (g >= a.reg_total_groups) break;
        long long rem = g;
        for (int k = 0; k < a.nreg_ranges; ++k) {
            long long ng = a.reg_cnt[k] >> 6;
            if (rem < ng) {
                float v = a.theta[a.reg_off[k] + rem * 64 + tid];
                acc = fmaf(v, v, acc);
                break;
            }
            rem -= ng;
        }
    }
    acc = wave_sum(acc);
    // (written through: in the mid-in-grad launch the consumer is a reduction block of the same launch)
    if (tid == 0) __hip_atomic_store(a.regpart + blk, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int S>
__device__ __forceinline__ f32x4 slab_sum(const float *src, long long sstride) {
    f32x4 t[S];
#pragma unroll
    for (int s = 0; s < S; ++s) t[s] = *(const f32x4 *)(src + (size_t)s * sstride);
    f32x4 acc = t[0];
#pragma unroll
    for (int s = 1; s < S; ++s) acc += t[s];
    return acc;
}

// float offset of (row r, column c) inside a fragment-major buffer with RG row groups
__device__ __forceinline__ size_t frag_off(int r, int c, int RG) {
    return ((size_t)(c >> 4) * RG + (r >> 4)) * 256 + (((r >> 2) & 3) * 16 + (c & 15)) * 4 + (r & 3);
}


// ---------------------------------------------------------------------------
// Register-resident version of the per-row math of the mid kernel for small
// (K <= KM, ceil(L/16) <= LQ): the same arithmetic as the generic LDS-array path
// below, with every per-lane array in VGPRs and fully unrolled loops (entries
// beyond the lane's own coordinates are zero-filled and contribute nothing), so
// that the kernel is a short straight-line ALU sequence instead of a chain of
// dependent LDS round trips.  Returns through the same row buffers.
// ---------------------------------------------------------------------------
template <int KM, int LQ>
__device__ __forceinline__ void mid_math_reg(const MidArgs &a, const float *Y, const float *SC,
                                             const float *BI, const float *MW, float thr_raw) {
    const int tid = threadIdx.x, p = tid & 15, j = tid >> 4;
    const int r = blockIdx.x * MID_RB + j;
    const bool valid = r < a.R;
    const int L = a.L, K = a.K, RG = a.Rpad >> 4;
    const int myL = p < L ? (L - p + 15) >> 4 : 0;
    const MidSide &ss = a.side[0], &sd = a.side[1];
    const int ks = ss.is_proto ? K : 1, kd = sd.is_proto ? K : 1;
    const int offd = ss.npad;

    float As[KM][LQ], Ad[KM][LQ], Xs[KM][LQ], Xd[KM][LQ], Rl[LQ];
    // ---- head epilogue ----
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int li = 0; li < LQ; ++li) {
            const int l = p + 16 * li;
            {
                const bool in = valid && li < myL && k < ks;
                const int c = in ? k * L + l : 0;
                const float xv = in ? Y[j * a.ys + c] * a.in_mul : 0.f;
                const float yy = xv * SC[c] + BI[c];
                Xs[k][li] = xv;
                if (a.dist_type == CFL_DIST_MONOMER && k == 0) Rl[li] = in ? yy : 0.f;
                As[k][li] = in ? act_fn(yy, a.act) : 0.f;
            }
            {
                const bool in = valid && li < myL && k < kd;
                const int c = in ? k * L + l : 0;
                const float xv = in ? Y[j * a.ys + offd + c] * a.in_mul : 0.f;
                const float yy = xv * SC[offd + c] + BI[offd + c];
                Xd[k][li] = xv;
                Ad[k][li] = in ? act_fn(yy, a.act) : 0.f;
            }
        }

    // ---- distance ----
    float d = 0.f, sK[KM], qK[KM], eK[KM], uK[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) sK[k] = qK[k] = eK[k] = uK[k] = 0.f;
    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    float e = 0.f;
#pragma unroll
                    for (int li = 0; li < LQ; ++li) { const float df = Ad[0][li] - As[k][li]; e = fmaf(df, df, e); }
                    e = -sum_p(e);
                    sK[k] = e;
                    mx = fmaxf(mx, e);
                }
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) { sK[k] = fexp(sK[k] - mx); den += sK[k]; }
            const float inv = frcp(den);
#pragma unroll
            for (int k = 0; k < KM; ++k) sK[k] = k < K ? sK[k] * inv : 0.f;
#pragma unroll
            for (int li = 0; li < LQ; ++li) {
                float m = 0.f;
#pragma unroll
                for (int k = 0; k < KM; ++k) m = fmaf(sK[k], As[k][li], m);
                const float rl = Ad[0][li] - m;
                Rl[li] = rl;
                d = fmaf(rl, rl, d);
#pragma unroll
                for (int k = 0; k < KM; ++k) qK[k] = fmaf(rl, As[k][li], qK[k]);
            }
            d = sum_p(d);
        } else {
#pragma unroll
            for (int li = 0; li < LQ; ++li) { const float df = Ad[0][li] - As[0][li]; d = fmaf(df, df, d); }
            d = sum_p(d);
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                float u = 0.f, e = 0.f;
#pragma unroll
                for (int li = 0; li < LQ; ++li) {
                    const int l = p + 16 * li;
                    const float w = li < myL ? MW[l * a.kpad + k] : 0.f;
                    u = fmaf(Rl[li], w, u);
                    const float df = As[0][li] - Ad[k][li];
                    e = fmaf(df, df, e);
                }
                u = sum_p(u);
                e = sum_p(e);
                uK[k] = u;
                if (a.weight_norm) u *= a.mono_g[k] / sqrtf(a.mono_n2[k]);
                sK[k] = u;
                eK[k] = e;
                mx = fmaxf(mx, u);
            }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) { sK[k] = fexp(sK[k] - mx); den += sK[k]; }
        const float inv = frcp(den);
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            sK[k] = k < K ? sK[k] * inv : 0.f;
            d = fmaf(sK[k], eK[k], d);
        }
    } else {
#pragma unroll
        for (int li = 0; li < LQ; ++li) { const float df = As[0][li] - Ad[0][li]; d = fmaf(df, df, d); }
        d = sum_p(d);
    }

    // ---- threshold, loss, dL/dd ----
    const float thr = fmaxf(thr_raw, CFL_THR_FLOOR);
    const float o = thr - d;
    if (!a.train) {
        if (valid && p == 0) {
            a.scores[r] = o;
            if (a.dists) a.dists[r] = d;
        }
        return;
    }
    const bool is_pos = r < a.B;
    const float invB = 1.f / (float)a.B;
    const float pw = a.pos_weight != 0.f ? a.pos_weight : 1.f;
    const float sp = flog1pexp(-fabsf(o));
    const float bce = fmaxf(o, 0.f) - (is_pos ? o : 0.f) + sp;
    const float eo = fexp(-fabsf(o));
    const float sig = (o >= 0.f ? 1.f : eo) * frcp(1.f + eo);
    const float dlo = is_pos ? (sig - 1.f) * pw * invB : sig * invB;
    float dd = 0.f;
    if (a.use_threshold) dd -= dlo;
    float hinge = 0.f;
    if (a.caffe_margin != 0.f) {
        if (is_pos) dd += 0.5f * pw * invB;
        else {
            hinge = fmaxf(0.f, a.caffe_margin - d);
            if (d < a.caffe_margin) dd -= 0.5f * invB;
        }
    } else if (a.lambda_m != 0.f) {
        if (is_pos) dd += pw * a.lambda_m * invB;
    }
    if (!valid) dd = 0.f;
    if (blockIdx.x == 0 && tid == 0) a.thr_copy[0] = thr;
    if (blockIdx.x == 0 && a.zero_i)
        for (int i = tid; i < a.nzero; i += blockDim.x) a.zero_i[i] = 0;
    {
        const bool pos = valid && is_pos, neg = valid && !is_pos;
        float qv = 0.f;
        switch (p) {
            case P_BCE_POS: qv = pos ? bce : 0.f; break;
            case P_BCE_NEG: qv = neg ? bce : 0.f; break;
            case P_OK_POS: qv = (pos && o > 0.f) ? 1.f : 0.f; break;
            case P_OK_NEG: qv = (neg && o <= 0.f) ? 1.f : 0.f; break;
            case P_D_POS: qv = pos ? d : 0.f; break;
            case P_D_NEG: qv = neg ? d : 0.f; break;
            case P_O_POS: qv = pos ? o : 0.f; break;
            case P_O_NEG: qv = neg ? o : 0.f; break;
            case P_DTHR: qv = valid ? dlo : 0.f; break;
            case P_HINGE_NEG: qv = neg ? hinge : 0.f; break;
            case P_SQRT_POS: qv = pos ? fsqrt(d + 1e-7f) : 0.f; break;
            case P_SQRT_NEG: qv = neg ? fsqrt(d + 1e-7f) : 0.f; break;
            default: break;
        }
        a.rowqf[frag_off(r, p, RG)] = qv;
    }

    // ---- backward ----
    auto emit = [&](const MidSide &sx, float A, float X, int k, int li, float dA, float extra_dy) {
        if (li >= myL) return;
        const int c = k * L + p + 16 * li;
        float dy = dA * act_grad(A, a.act) + extra_dy;
        if (!valid) dy = 0.f;
        const size_t o_ = frag_off(r, c, RG);
        sx.dyf[o_] = dy;
        if (sx.cwf) sx.cwf[o_] = dy * X;
    };
    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float qbar = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) { qK[k] = -2.f * sum_p(qK[k]); qbar = fmaf(sK[k], qK[k], qbar); }
#pragma unroll
            for (int k = 0; k < KM; ++k) qK[k] = sK[k] * (qK[k] - qbar);  // dl_k
#pragma unroll
            for (int li = 0; li < LQ; ++li) {
                const float v = Ad[0][li], rl = Rl[li];
                float dv = 2.f * rl;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        const float vmP = v - As[k][li];
                        dv = fmaf(-2.f * qK[k], vmP, dv);
                        const float dP = -2.f * sK[k] * rl + 2.f * qK[k] * vmP;
                        emit(ss, As[k][li], Xs[k][li], k, li, dP * dd, 0.f);
                    }
                emit(sd, Ad[0][li], Xd[0][li], 0, li, dv * dd, 0.f);
            }
        } else {
#pragma unroll
            for (int li = 0; li < LQ; ++li) {
                const float df = Ad[0][li] - As[0][li];
                emit(ss, As[0][li], Xs[0][li], 0, li, -2.f * df * dd, 0.f);
                emit(sd, Ad[0][li], Xd[0][li], 0, li, 2.f * df * dd, 0.f);
            }
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float du = sK[k] * (eK[k] - d) * dd;
                float scm = 1.f;
                if (a.weight_norm) scm = a.mono_g[k] / sqrtf(a.mono_n2[k]);
                if (p == 0) {
                    a.mono_du[(size_t)r * a.kpad + k] = du * scm;
                    if (a.weight_norm) a.mono_duc[(size_t)r * a.kpad + k] = valid ? du * uK[k] : 0.f;
                }
                qK[k] = du * scm;
            }
#pragma unroll
        for (int li = 0; li < LQ; ++li) {
            const int l = p + 16 * li;
            if (li < myL) a.mono_ya[(size_t)r * a.lpad + l] = valid ? Rl[li] : 0.f;
            float da = 0.f, ex = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    const float amP = As[0][li] - Ad[k][li];
                    da = fmaf(2.f * sK[k], amP, da);
                    emit(sd, Ad[k][li], Xd[k][li], k, li, -2.f * sK[k] * amP * dd, 0.f);
                    ex = fmaf(qK[k], li < myL ? MW[l * a.kpad + k] : 0.f, ex);
                }
            emit(ss, As[0][li], Xs[0][li], 0, li, da * dd, ex);
        }
    } else {
#pragma unroll
        for (int li = 0; li < LQ; ++li) {
            const float df = As[0][li] - Ad[0][li];
            emit(ss, As[0][li], Xs[0][li], 0, li, 2.f * df * dd, 0.f);
            emit(sd, Ad[0][li], Xd[0][li], 0, li, -2.f * df * dd, 0.f);
        }
    }
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        for (int c = sx.n + p; c < sx.npad; c += 16) {
            sx.dyf[frag_off(r, c, RG)] = 0.f;
            if (sx.cwf) sx.cwf[frag_off(r, c, RG)] = 0.f;
        }
    }
}

template <int KM, int LQ>
__global__ __launch_bounds__(64) void cfl_mid_kernel(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    if ((int)blockIdx.x >= a.nrb) {
        mid_reg_block(a, blockIdx.x - a.nrb);
        return;
    }
    const int tid = threadIdx.x, p = tid & 15, j = tid >> 4;
    const int row0 = blockIdx.x * MID_RB;
    const int r = row0 + j;
    const bool valid = r < a.R;
    const int L = a.L, K = a.K, Lq = a.Lq, RG = a.Rpad >> 4;
    const int myL = p < L ? (L - p + 15) >> 4 : 0;  // number of l = p + 16*li < L
    const MidSide &ss = a.side[0], &sd = a.side[1];
    const int ks = ss.is_proto ? K : 1, kd = sd.is_proto ? K : 1;

    // LDS carve: Y[4][ys], per-column scale SC[ys] and bias BI[ys], gate weights
    // MW[L*kpad], then per-lane arrays [slots][64]
    float *Y = lds;
    float *SC = Y + MID_RB * a.ys;
    float *BI = SC + a.ys;
    float *MW = BI + a.ys;
    float *As = MW + (a.dist_type == CFL_DIST_MONOMER ? ((L * a.kpad + 3) & ~3) : 0);  // src activations ks*Lq
    float *Ad = As + ks * Lq * 64;             // dst activations   kd*Lq
    float *Xs = Ad + kd * Lq * 64;             // src raw x_hat.V (weight-norm)  ks*Lq
    float *Xd = Xs + (a.weight_norm ? ks * Lq * 64 : 0);
    float *Rl = Xd + (a.weight_norm ? kd * Lq * 64 : 0);   // Lq : pcd residual / monomer pre-act
    float *Kv = Rl + Lq * 64;                  // 4 x K small vectors
    float *Ks = Kv, *Kq = Kv + K * 64, *Ke = Kv + 2 * K * 64, *Ku = Kv + 3 * K * 64;

    // ---- phase 1: slice sums + parameters -> LDS ----------------------------------
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        const int nq = sx.npad >> 2;
        const int coloff = side ? a.side[0].npad : 0;
        for (int idx = tid; idx < MID_RB * nq; idx += 64) {
            const int jj = idx / nq, c4 = idx - jj * nq;
            const float *src = sx.ypart + (size_t)(row0 + jj) * sx.npad + 4 * c4;
            f32x4 acc;
            switch (a.S) {
                case 1: acc = slab_sum<1>(src, sx.sstride); break;
                case 2: acc = slab_sum<2>(src, sx.sstride); break;
                case 4: acc = slab_sum<4>(src, sx.sstride); break;
                case 8: acc = slab_sum<8>(src, sx.sstride); break;
                default: acc = slab_sum<16>(src, sx.sstride); break;
            }
            *(f32x4 *)(Y + jj * a.ys + coloff + 4 * c4) = acc;
        }
        for (int c = tid; c < sx.n; c += 64) {
            SC[coloff + c] = a.weight_norm ? sx.g[c] * __builtin_amdgcn_rsqf(sx.n2[c]) : 1.f;
            BI[coloff + c] = sx.b ? sx.b[c] : 0.f;
        }
    }
    if (a.dist_type == CFL_DIST_MONOMER)
        for (int i = tid; i < L * a.kpad; i += 64) MW[i] = a.mono_w[i];
    const float thr_raw = *a.thr;
    __syncthreads();
    // small shapes: register-resident math (same arithmetic as the generic path below); one
    // kernel instantiation per shape class -- co-inlined variants made hipcc spill to scratch
    if constexpr (KM > 0) {
        mid_math_reg<KM, LQ>(a, Y, SC, BI, MW, thr_raw);
        return;
    }
#ifdef ABL_MID_P1ONLY
    if (a.train) { if (tid == 0) a.rowqf[blockIdx.x] = Y[0] + thr_raw; return; }
#endif

    // ---- phase 2.1: head epilogue ------------------------------------------------
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        float *A = side ? Ad : As, *X = side ? Xd : Xs;
        const int coloff = side ? a.side[0].npad : 0;
        const int kk = sx.is_proto ? K : 1;
        for (int k = 0; k < kk; ++k)
            for (int li = 0; li < myL; ++li) {
                const int c = k * L + p + 16 * li;
                float y = Y[j * a.ys + coloff + c];
                if (!valid) y = 0.f;  // rows >= R of the scratch slabs are never written
                const float xv = y * a.in_mul;
                const float yy = xv * SC[coloff + c] + BI[coloff + c];
                const int slot = (k * Lq + li) * 64 + tid;
                if (a.weight_norm) X[slot] = xv;
                if (a.dist_type == CFL_DIST_MONOMER && side == 0) Rl[li * 64 + tid] = yy;
                A[slot] = act_fn(yy, a.act);
            }
    }

    // ---- phase 2.2: distance -----------------------------------------------------
    float d = 0.f;
    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) {
                float e = 0.f;
                for (int li = 0; li < myL; ++li) {
                    float df = Ad[li * 64 + tid] - As[(k * Lq + li) * 64 + tid];
                    e = fmaf(df, df, e);
                }
                e = -sum_p(e);
                Ks[k * 64 + tid] = e;
                mx = fmaxf(mx, e);
            }
            float den = 0.f;
            for (int k = 0; k < K; ++k) {
                float ex = fexp(Ks[k * 64 + tid] - mx);
                Ks[k * 64 + tid] = ex;
                den += ex;
                Kq[k * 64 + tid] = 0.f;
            }
            const float inv = frcp(den);
            for (int k = 0; k < K; ++k) Ks[k * 64 + tid] *= inv;
            for (int li = 0; li < myL; ++li) {
                const float v = Ad[li * 64 + tid];
                float m = 0.f;
                for (int k = 0; k < K; ++k) m = fmaf(Ks[k * 64 + tid], As[(k * Lq + li) * 64 + tid], m);
                const float rl = v - m;
                Rl[li * 64 + tid] = rl;
                d = fmaf(rl, rl, d);
                for (int k = 0; k < K; ++k) Kq[k * 64 + tid] += rl * As[(k * Lq + li) * 64 + tid];
            }
            d = sum_p(d);
        } else {
            for (int li = 0; li < myL; ++li) {
                float df = Ad[li * 64 + tid] - As[li * 64 + tid];
                d = fmaf(df, df, d);
            }
            d = sum_p(d);
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
        // gate u_k = (ya . Vm[:,k]) * scale_k from the PRE-activation outputs (base.py:96)
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            float u = 0.f, e = 0.f;
            for (int li = 0; li < myL; ++li) {
                const int l = p + 16 * li;
                u = fmaf(Rl[li * 64 + tid], MW[l * a.kpad + k], u);
                float df = As[li * 64 + tid] - Ad[(k * Lq + li) * 64 + tid];
                e = fmaf(df, df, e);
            }
            u = sum_p(u);
            e = sum_p(e);
            Ku[k * 64 + tid] = u;  // raw ya.Vm (needed for the weight-norm gain grad)
            if (a.weight_norm) u *= a.mono_g[k] / sqrtf(a.mono_n2[k]);
            Ks[k * 64 + tid] = u;
            Ke[k * 64 + tid] = e;
            mx = fmaxf(mx, u);
        }
        float den = 0.f;
        for (int k = 0; k < K; ++k) {
            float ex = fexp(Ks[k * 64 + tid] - mx);
            Ks[k * 64 + tid] = ex;
            den += ex;
        }
        const float inv = frcp(den);
        for (int k = 0; k < K; ++k) {
            const float w = Ks[k * 64 + tid] * inv;
            Ks[k * 64 + tid] = w;
            d = fmaf(w, Ke[k * 64 + tid], d);
        }
    } else {
        for (int li = 0; li < myL; ++li) {
            float df = As[li * 64 + tid] - Ad[li * 64 + tid];
            d = fmaf(df, df, d);
        }
        d = sum_p(d);
    }

    // ---- phase 2.3: threshold, loss, dL/dd ----------------------------------------
    const float thr = fmaxf(thr_raw, CFL_THR_FLOOR);
    const float o = thr - d;
    if (!a.train) {
        if (valid && p == 0) {
            a.scores[r] = o;
            if (a.dists) a.dists[r] = d;
        }
        return;
    }
    const bool is_pos = r < a.B;
    const float invB = 1.f / (float)a.B;
    const float pw = a.pos_weight != 0.f ? a.pos_weight : 1.f;
    const float sp = flog1pexp(-fabsf(o));
    const float bce = fmaxf(o, 0.f) - (is_pos ? o : 0.f) + sp;
    const float eo = fexp(-fabsf(o));
    const float sig = (o >= 0.f ? 1.f : eo) * frcp(1.f + eo);
    const float dlo = is_pos ? (sig - 1.f) * pw * invB : sig * invB;  // dL_thr/do
    float dd = 0.f;
    if (a.use_threshold) dd -= dlo;
    float hinge = 0.f;
    if (a.caffe_margin != 0.f) {
        if (is_pos) dd += 0.5f * pw * invB;
        else {
            hinge = fmaxf(0.f, a.caffe_margin - d);
            if (d < a.caffe_margin) dd -= 0.5f * invB;
        }
    } else if (a.lambda_m != 0.f) {
        if (is_pos) dd += pw * a.lambda_m * invB;
    }
    if (!valid) dd = 0.f;

#ifdef ABL_MID_NOBWD
    if (a.train) { if (tid == 0) a.rowqf[blockIdx.x] = dd; return; }
#endif
    if (blockIdx.x == 0 && tid == 0) a.thr_copy[0] = thr;
    if (blockIdx.x == 0 && a.zero_i)
        for (int i = tid; i < a.nzero; i += blockDim.x) a.zero_i[i] = 0;
    // per-row loss quantities: lane p writes quantity #p of its row (one fragment tile,
    // summed over rows by grad_red_block)
    {
        const bool pos = valid && is_pos, neg = valid && !is_pos;
        float qv = 0.f;
        switch (p) {
            case P_BCE_POS: qv = pos ? bce : 0.f; break;
            case P_BCE_NEG: qv = neg ? bce : 0.f; break;
            case P_OK_POS: qv = (pos && o > 0.f) ? 1.f : 0.f; break;
            case P_OK_NEG: qv = (neg && o <= 0.f) ? 1.f : 0.f; break;
            case P_D_POS: qv = pos ? d : 0.f; break;
            case P_D_NEG: qv = neg ? d : 0.f; break;
            case P_O_POS: qv = pos ? o : 0.f; break;
            case P_O_NEG: qv = neg ? o : 0.f; break;
            case P_DTHR: qv = valid ? dlo : 0.f; break;
            case P_HINGE_NEG: qv = neg ? hinge : 0.f; break;
            case P_SQRT_POS: qv = pos ? fsqrt(d + 1e-7f) : 0.f; break;
            case P_SQRT_NEG: qv = neg ? fsqrt(d + 1e-7f) : 0.f; break;
            default: break;
        }
        a.rowqf[frag_off(r, p, RG)] = qv;
    }

    // ---- phase 2.4: backward to dL/dY (fragment-major, unscaled) -------------------
    auto emit = [&](const MidSide &sx, const float *A, const float *X, int k, int li, float dA,
                    float extra_dy) {
        const int c = k * L + p + 16 * li;
        const int slot = (k * Lq + li) * 64 + tid;
        float dy = dA * act_grad(A[slot], a.act) + extra_dy;
        if (!valid) dy = 0.f;
        const size_t o_ = frag_off(r, c, RG);
        sx.dyf[o_] = dy;
        if (sx.cwf) sx.cwf[o_] = valid ? dy * X[slot] : 0.f;
    };

    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float qbar = 0.f;
            for (int k = 0; k < K; ++k) {
                float qk = -2.f * sum_p(Kq[k * 64 + tid]);
                Kq[k * 64 + tid] = qk;
                qbar = fmaf(Ks[k * 64 + tid], qk, qbar);
            }
            for (int k = 0; k < K; ++k)  // dl_k = s_k (q_k - qbar)
                Kq[k * 64 + tid] = Ks[k * 64 + tid] * (Kq[k * 64 + tid] - qbar);
            for (int li = 0; li < myL; ++li) {
                const float v = Ad[li * 64 + tid], rl = Rl[li * 64 + tid];
                float dv = 2.f * rl;
                for (int k = 0; k < K; ++k) {
                    const float vmP = v - As[(k * Lq + li) * 64 + tid];
                    const float dl = Kq[k * 64 + tid];
                    dv = fmaf(-2.f * dl, vmP, dv);
                    const float dP = -2.f * Ks[k * 64 + tid] * rl + 2.f * dl * vmP;
                    emit(ss, As, Xs, k, li, dP * dd, 0.f);
                }
                emit(sd, Ad, Xd, 0, li, dv * dd, 0.f);
            }
        } else {
            for (int li = 0; li < myL; ++li) {
                const float df = Ad[li * 64 + tid] - As[li * 64 + tid];
                emit(ss, As, Xs, 0, li, -2.f * df * dd, 0.f);
                emit(sd, Ad, Xd, 0, li, 2.f * df * dd, 0.f);
            }
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
        // du_k = w_k (e_k - d) dd ; gate-head rows for dVm, dgm
        for (int k = 0; k < K; ++k) {
            float du = Ks[k * 64 + tid] * (Ke[k * 64 + tid] - d) * dd;
            float scm = 1.f;
            if (a.weight_norm) scm = a.mono_g[k] / sqrtf(a.mono_n2[k]);
            if (p == 0) {
                a.mono_du[(size_t)r * a.kpad + k] = du * scm;
                if (a.weight_norm) a.mono_duc[(size_t)r * a.kpad + k] = valid ? du * Ku[k * 64 + tid] : 0.f;
            }
            Kq[k * 64 + tid] = du * scm;  // grad wrt raw ya.Vm
        }
        for (int li = 0; li < myL; ++li) {
            const int l = p + 16 * li;
            const float av = As[li * 64 + tid], ya = Rl[li * 64 + tid];
            a.mono_ya[(size_t)r * a.lpad + l] = valid ? ya : 0.f;
            float da = 0.f, ex = 0.f;
            for (int k = 0; k < K; ++k) {
                const float amP = av - Ad[(k * Lq + li) * 64 + tid];
                const float w = Ks[k * 64 + tid];
                da = fmaf(2.f * w, amP, da);
                emit(sd, Ad, Xd, k, li, -2.f * w * amP * dd, 0.f);
                ex = fmaf(Kq[k * 64 + tid], MW[l * a.kpad + k], ex);
            }
            emit(ss, As, Xs, 0, li, da * dd, ex);
        }
    } else {
        for (int li = 0; li < myL; ++li) {
            const float df = As[li * 64 + tid] - Ad[li * 64 + tid];
            emit(ss, As, Xs, 0, li, 2.f * df * dd, 0.f);
            emit(sd, Ad, Xd, 0, li, -2.f * df * dd, 0.f);
        }
    }
    // zero the padding columns of dYf (read by the grad GEMM and the column sums)
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        for (int c = sx.n + p; c < sx.npad; c += 16) {
            sx.dyf[frag_off(r, c, RG)] = 0.f;
            if (sx.cwf) sx.cwf[frag_off(r, c, RG)] = 0.f;
        }
    }
}


// ---------------------------------------------------------------------------
// mid, wave-per-row form for PCD with at most 64 padded columns per side (the
// Monomer / Polyvore / dyadic pcd shapes): one wave owns one pair row, lane c owns
// source column c = k*L + l (and, for c < L, destination column c).  The per-wave
// instruction count -- which IS the latency of this one-wave-per-row kernel -- drops
// ~4x against the 16-lanes-per-row form: slice sums are one coalesced 256-byte load
// per slice, every per-column quantity is one VALU op, and the few cross-column sums
// go through a 1 KiB wave-private LDS scratch.  Same arithmetic, same outputs.
// ---------------------------------------------------------------------------
// `lead`: the one wave of the launch that also clears the hand-off flags of the weight-gradient launch and snapshots the threshold
template <int J>   // J = columns per lane: sides of up to 64 * J padded columns
__device__ __forceinline__ void mid_row_body(const MidArgs &a, int r, bool lead, float *W) {
    const int lane = threadIdx.x & 63;
    constexpr int CW = 64 * J;
    float *Pl = W, *Vl = W + CW, *Rl = W + 2 * CW, *T = W + 3 * CW, *S = W + 4 * CW, *Q = W + 5 * CW;
    RSTAMP(0);
    const bool extra = a.xn > 0 && r >= a.xrow0;
    const bool valid = extra ? r - a.xrow0 < a.xn : r < a.R;
    const int L = a.L, K = a.K, RG = a.Rpad >> 4;
    const MidSide &ss = a.side[0], &sd = a.side[1];
    const int ns = ss.n;
    int c[J], kk[J], ll[J];
    bool cs[J], cd[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        c[j] = lane + 64 * j;
        cs[j] = c[j] < ns;
        cd[j] = c[j] < L;
        kk[j] = cs[j] ? c[j] / L : 0;
        ll[j] = cs[j] ? c[j] - kk[j] * L : 0;
    }

    // ---- head parameters + slice sums: EVERY load of this phase is unconditional and issued in one batch ---------------
    // This one-wave-per-row kernel IS its latency chain.  Round 4 (tools/mid_stamp_probe.py + the ISA): written with
    // `cond ? ptr[i] : const` / `s < S ? slab[s] : 0` the loads sat inside uniform branches, and hipcc's waitcnt pass
    // drains the queue (s_waitcnt vmcnt(0)) at every such join -- the 16 + 16 slab loads went out one round trip after
    // the other (3.2 us of the 6.7 us wave lifetime at the headline shape: 7700 -> 4100 cycles with this form).  Now:
    // absent arrays point at a dummy word and their values are replaced by selects, and the slab count is a template
    // parameter of the loader (switch on S BEFORE anything is in flight), so the compiler sees straight-line loads.
    const float thr_raw = *a.thr;
    float scs[J], scd[J], bs[J], bd[J], ys[J], yd[J];
    {
        const bool wn = a.weight_norm != 0;
        const float *gsp = wn ? ss.g : a.thr, *nsp = wn ? ss.n2 : a.thr, *gdp = wn ? sd.g : a.thr, *ndp = wn ? sd.n2 : a.thr;
        const float *bsp = ss.b ? ss.b : a.thr, *bdp = sd.b ? sd.b : a.thr;
        const bool hbs = ss.b != nullptr, hbd = sd.b != nullptr;
        float gs[J], ns[J], gd[J], nd[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int ks = cs[j] ? c[j] : 0, kd = cd[j] ? c[j] : 0;   // clamped: unconditional loads
            gs[j] = gsp[wn ? ks : 0];
            ns[j] = nsp[wn ? ks : 0];
            gd[j] = gdp[wn ? kd : 0];
            nd[j] = ndp[wn ? kd : 0];
            bs[j] = bsp[hbs ? ks : 0];
            bd[j] = bdp[hbd ? kd : 0];
        }
        const float *srcs[J], *srcd[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int ccs = c[j] < ss.npad ? c[j] : 0, ccd = c[j] < sd.npad ? c[j] : 0;
            srcs[j] = ss.ypart + (size_t)r * ss.npad + ccs;
            srcd[j] = sd.ypart + (size_t)r * sd.npad + ccd;
        }
        // all slice loads of both sides are independent and in flight together; summed in slice order afterwards
        // (the same order of additions as before: s = 0, 1, ..., S - 1)
        auto slabs = [&](auto ns_c) {
            constexpr int NS = decltype(ns_c)::value;
            float ts[J][NS], td[J][NS];
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    ts[j][sl] = srcs[j][(size_t)sl * ss.sstride];
                    td[j][sl] = srcd[j][(size_t)sl * sd.sstride];
                }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                ys[j] = 0.f;
                yd[j] = 0.f;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) { ys[j] += ts[j][sl]; yd[j] += td[j][sl]; }
            }
        };
        switch (a.S) {   // (uniform; the d split is a power of two up to 16, anything else from CFL_DEBUG_S: generic tail)
            case 1: slabs(std::integral_constant<int, 1>()); break;
            case 2: slabs(std::integral_constant<int, 2>()); break;
            case 4: slabs(std::integral_constant<int, 4>()); break;
            case 8: slabs(std::integral_constant<int, 8>()); break;
            case 16: slabs(std::integral_constant<int, 16>()); break;
            default: {
                // any other split: clamped slice indices (re-reads of the last slice are masked out), still branch-free
                const int S1 = a.S - 1;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    float accs = 0.f, accd = 0.f, ts[16], td[16];
#pragma unroll
                    for (int sl = 0; sl < 16; ++sl) {
                        const int sc = sl < a.S ? sl : S1;
                        ts[sl] = srcs[j][(size_t)sc * ss.sstride];
                        td[sl] = srcd[j][(size_t)sc * sd.sstride];
                    }
#pragma unroll
                    for (int sl = 0; sl < 16; ++sl) { accs += sl < a.S ? ts[sl] : 0.f; accd += sl < a.S ? td[sl] : 0.f; }
                    ys[j] = accs;
                    yd[j] = accd;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            scs[j] = (wn && cs[j]) ? gs[j] * __builtin_amdgcn_rsqf(ns[j]) : 1.f;
            scd[j] = (wn && cd[j]) ? gd[j] * __builtin_amdgcn_rsqf(nd[j]) : 1.f;
            bs[j] = (hbs && cs[j]) ? bs[j] : 0.f;
            bd[j] = (hbd && cd[j]) ? bd[j] : 0.f;
        }
    }
    RSTAMP(1);   // slabs summed: the loads have landed
    float xvs[J], xvd[J], P[J], v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        xvs[j] = (valid && cs[j]) ? ys[j] * a.in_mul : 0.f;
        xvd[j] = (valid && cd[j]) ? yd[j] * a.in_mul : 0.f;
        P[j] = (valid && cs[j]) ? act_fn(xvs[j] * scs[j] + bs[j], a.act) : 0.f;
        v[j] = (valid && cd[j]) ? act_fn(xvd[j] * scd[j] + bd[j], a.act) : 0.f;
        Pl[c[j]] = P[j];
        Vl[c[j]] = v[j];
    }
    float diff[J];
#pragma unroll
    for (int j = 0; j < J; ++j) diff[j] = cs[j] ? Vl[ll[j]] - P[j] : 0.f;   // destination coordinate l of this column

    // ---- distance ----------------------------------------------------------------
    float d, sk[J], rl[J], dlk[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { sk[j] = 1.f; rl[j] = 0.f; dlk[j] = 0.f; }
    if (K > 1) {
#pragma unroll
        for (int j = 0; j < J; ++j) T[c[j]] = diff[j] * diff[j];
        // (measured, round 4: the segment sums with LDS reads batched eight at a time -- clamped indices, masked values, all
        // lanes -- and the K-loops on v_readlane instead of LDS broadcasts: this phase 4170 -> 4800 cycles at the headline
        // shape.  Three lanes reading a segment each is cheap; sixty-four reading strided segments is not.  Left as it was.)
        float e = 0.f;
        if (lane < K)
            for (int i = 0; i < L; ++i) e += T[lane * L + i];
        S[lane] = -e;                 // logits (lanes >= K: unused)
        float mx = -INFINITY;
        for (int i = 0; i < K; ++i) mx = fmaxf(mx, S[i]);
        float den = 0.f;
        for (int i = 0; i < K; ++i) den += fexp(S[i] - mx);
        const float inv = frcp(den);
        const float sme = lane < K ? fexp(-e - mx) * inv : 0.f;   // s_k for lanes < K
        Q[lane] = sme;
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            sk[j] = Q[kk[j]];         // softmax weight of this column's prototype
            float m = 0.f;
            if (cd[j])
                for (int i = 0; i < K; ++i) m = fmaf(Q[i], Pl[i * L + c[j]], m);
            rl[j] = cd[j] ? v[j] - m : 0.f;
            Rl[c[j]] = rl[j];
            dsum = fmaf(rl[j], rl[j], dsum);
        }
        d = wave_sum_dpp(dsum);
#pragma unroll
        for (int j = 0; j < J; ++j) T[c[j]] = cs[j] ? Rl[ll[j]] * P[j] : 0.f;
        float q = 0.f;
        if (lane < K)
            for (int i = 0; i < L; ++i) q += T[lane * L + i];
        q *= -2.f;
        S[lane] = q;                  // q_k (lanes < K)
        float qbar = 0.f;
        for (int i = 0; i < K; ++i) qbar = fmaf(Q[i], S[i], qbar);
        const float dl_me = lane < K ? sme * (q - qbar) : 0.f;
        T[lane] = dl_me;              // dl_k (lanes < K); T is free again, q has been reduced
#pragma unroll
        for (int j = 0; j < J; ++j) dlk[j] = T[kk[j]];
    } else {
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) dsum = fmaf(diff[j], diff[j], dsum);
        d = wave_sum_dpp(dsum);
    }

    // ---- threshold, loss, dL/dd ------------------------------------------------------
    RSTAMP(2);   // distance done
    const float thr = fmaxf(thr_raw, CFL_THR_FLOOR);
    const float o = thr - d;
    if (!a.train || extra) {
        if (valid && lane == 0) {
            if (extra) {
                a.xscores[r - a.xrow0] = o;
            } else {
                a.scores[r] = o;
                if (a.dists) a.dists[r] = d;
            }
        }
        return;
    }
    const bool is_pos = r < a.B;
    const float invB = 1.f / (float)a.B;
    const float pw = a.pos_weight != 0.f ? a.pos_weight : 1.f;
    const float eo = fexp(-fabsf(o));
    const float sp = __logf(1.f + eo);
    const float bce = fmaxf(o, 0.f) - (is_pos ? o : 0.f) + sp;
    const float sig = (o >= 0.f ? 1.f : eo) * frcp(1.f + eo);
    const float dlo = is_pos ? (sig - 1.f) * pw * invB : sig * invB;
    float dd = 0.f;
    if (a.use_threshold) dd -= dlo;
    float hinge = 0.f;
    if (a.caffe_margin != 0.f) {
        if (is_pos) dd += 0.5f * pw * invB;
        else {
            hinge = fmaxf(0.f, a.caffe_margin - d);
            if (d < a.caffe_margin) dd -= 0.5f * invB;
        }
    } else if (a.lambda_m != 0.f) {
        if (is_pos) dd += pw * a.lambda_m * invB;
    }
    if (!valid) dd = 0.f;
    auto put = [&](float *q, float val) { *q = val; };
    if (lead && lane == 0) put(a.thr_copy, thr);
    if (lead && a.zero_i)
        for (int i = lane; i < a.nzero; i += 64) a.zero_i[i] = 0;
    if (lane < 16) {
        const bool pos = valid && is_pos, neg = valid && !is_pos;
        float qv = 0.f;
        switch (lane) {
            case P_BCE_POS: qv = pos ? bce : 0.f; break;
            case P_BCE_NEG: qv = neg ? bce : 0.f; break;
            case P_OK_POS: qv = (pos && o > 0.f) ? 1.f : 0.f; break;
            case P_OK_NEG: qv = (neg && o <= 0.f) ? 1.f : 0.f; break;
            case P_D_POS: qv = pos ? d : 0.f; break;
            case P_D_NEG: qv = neg ? d : 0.f; break;
            case P_O_POS: qv = pos ? o : 0.f; break;
            case P_O_NEG: qv = neg ? o : 0.f; break;
            case P_DTHR: qv = valid ? dlo : 0.f; break;
            case P_HINGE_NEG: qv = neg ? hinge : 0.f; break;
            case P_SQRT_POS: qv = pos ? fsqrt(d + 1e-7f) : 0.f; break;
            case P_SQRT_NEG: qv = neg ? fsqrt(d + 1e-7f) : 0.f; break;
            default: break;
        }
        put(a.rowqf + frag_off(r, lane, RG), qv);
    }

    RSTAMP(3);   // loss + row quantities stored
    // ---- backward: J source columns and (columns < L) J destination columns per lane ----
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float dP, dv;
        if (K > 1) {
            const float rls = Rl[ll[j]];
            dP = -2.f * sk[j] * rls + 2.f * dlk[j] * diff[j];
            dv = 2.f * rl[j];
            if (cd[j])
                for (int i = 0; i < K; ++i) dv = fmaf(-2.f * T[i], v[j] - Pl[i * L + c[j]], dv);
        } else {
            dP = -2.f * diff[j];
            dv = 2.f * (v[j] - Pl[c[j]]);
        }
        if (c[j] < ss.npad) {
            const float dy = cs[j] ? dP * dd * act_grad(P[j], a.act) : 0.f;
            const size_t o_ = frag_off(r, c[j], RG);
            put(ss.dyf + o_, dy);
            if (ss.cwf) put(ss.cwf + o_, dy * xvs[j]);
        }
        if (c[j] < sd.npad) {
            const float dy = cd[j] ? dv * dd * act_grad(v[j], a.act) : 0.f;
            const size_t o_ = frag_off(r, c[j], RG);
            put(sd.dyf + o_, dy);
            if (sd.cwf) put(sd.cwf + o_, dy * xvd[j]);
        }
    }
    RSTAMP(4);   // stores issued
#ifdef CFL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSTAMP(5);   // stores acknowledged
#endif
}

template <int J>
__global__ __launch_bounds__(256) void cfl_mid_row_kernel(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)blockIdx.x >= a.nrb + a.nxb) {
        if (wave == 0) mid_reg_block(a, blockIdx.x - a.nrb - a.nxb);
        return;
    }
    if ((int)blockIdx.x >= a.nrb) {   // extra scoring rows (forward only)
        mid_row_body<J>(a, a.xrow0 + ((int)blockIdx.x - a.nrb) * 4 + wave, false, (float *)smem + wave * 6 * 64 * J);
        return;
    }
    // (Round 5 measured the other placement: row tiles dealt over the XCDs by proj -- all d slices and column jobs of a 32-row
    // tile on ONE XCD -- and the blocks here taking the rows whose slabs their own XCD's L2 still holds.  mid -0.2 us, proj
    // +0.6, grad +0.5: the slab loads are not what this launch waits for.  profiles/r05_mid_xcd_ab.txt; not kept)
    mid_row_body<J>(a, blockIdx.x * 4 + wave, blockIdx.x == 0 && wave == 0, (float *)smem + wave * 6 * 64 * J);
}


// ---------------------------------------------------------------------------
// finalize: weight-gradient slabs + row-reduced column sums -> flat gradient ;
//           last block -> scalars.  Purely element-wise: every reduction over rows
//           was done by grad_red_block, every reduction over theta by mid_reg_block.
// ---------------------------------------------------------------------------
enum { RK_ZERO = 0, RK_W, RK_BIAS, RK_GAIN, RK_THR, RK_MONO_W, RK_MONO_G };

struct Region {
    long long off, cnt;        // floats (64-aligned)
    int kind, reg;
    const float *slab[2];      // weight-gradient slabs [P] x Wf (one per contributing side)
    int cs_dy[2], cs_cw[2];    // colsum offsets of the bias / gain column sums (-1: none)
    int npad, n;               // padded / logical columns of the head
    const float *g, *n2;       // weight-norm (gain snapshot, squared norms)
};

struct FinArgs {
    // compact copy of the region bounds (floats): the region search reads these with three wide scalar loads
    int rbeg[CFL_MAX_REGIONS], rend[CFL_MAX_REGIONS];
    Region reg[CFL_MAX_REGIONS];
    int nregions;
    long long total;           // floats in theta
    const float *theta;
    float *grad;
    const float *colsum;
    int cs_rowq, cs_mono, cs_duc;
    int P, D, L, kpad, weight_norm;
    float in_mul, reg_const;
    int use_threshold;
    float pos_weight, caffe_margin, lambda_m;
    int B;
    const float *regpart;
    int nregblocks;
    float *scalars;
    int nblocks_main;
    // optional fused Adam (theta_out aliases theta)
    float *adam_m, *adam_v, *theta_out;
    float lr_t, b1, b2, eps;
    const float *thr_copy;     // max(thr,1e-6) as seen by the mid kernel of this step
};

// gradient of the 4 parameters at `base`, which lie in region `rg`
__device__ __forceinline__ f32x4 fin_region_grad(const FinArgs &a, const Region &rg, long long base, const f32x4 th) {
    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
    const long long rel = base - rg.off;
    switch (rg.kind) {
        case RK_W: {
            // Wf layout: block = rel/256 -> nt = block / G ; c16 = ((rel%256)/4) % 16
            const int G = a.D >> 4;
            const int c = (int)((rel >> 8) / G) * 16 + (int)((rel >> 2) & 15);
            const long long ps = (long long)rg.npad * a.D;
            f32x4 t[2][8];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
                    t[s][pp] = (rg.slab[s] && pp < a.P) ? *(const f32x4 *)(rg.slab[s] + rel + pp * ps)
                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) gr += t[s][pp];
            // the slabs hold X^T dy with unscaled dy: apply the input / weight-norm scale
            if (a.weight_norm) {
                if (c < rg.n) {
                    const float n2 = rg.n2[c], n = sqrtf(n2);
                    gr *= n2 > 0.f ? a.in_mul * rg.g[c] / n : 0.f;
                    float cw = 0.f;
                    for (int s = 0; s < 2; ++s)
                        if (rg.cs_cw[s] >= 0) cw += a.colsum[rg.cs_cw[s] + c];
                    // (explicit fma: the fused tail of the weight-gradient launch performs the same operations)
                    const float s2 = n2 > 0.f ? rg.g[c] * cw / (n2 * n) : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gr[e] = fmaf(-s2, th[e], gr[e]);
                } else {
                    gr *= 0.f;
                }
            } else {
                gr *= a.in_mul;
            }
            break;
        }
        case RK_BIAS: {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (int)rel + e;
                if (c < rg.n)
                    for (int s = 0; s < 2; ++s)
                        if (rg.cs_dy[s] >= 0) gr[e] += a.colsum[rg.cs_dy[s] + c];
            }
            break;
        }
        case RK_GAIN: {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (int)rel + e;
                if (c < rg.n) {
                    float cw = 0.f;
                    for (int s = 0; s < 2; ++s)
                        if (rg.cs_cw[s] >= 0) cw += a.colsum[rg.cs_cw[s] + c];
                    const float n2 = rg.n2[c];
                    gr[e] = n2 > 0.f ? cw / sqrtf(n2) : 0.f;
                }
            }
            break;
        }
        case RK_THR: {
            if (rel == 0) gr[0] = th[0] >= CFL_THR_FLOOR ? a.colsum[a.cs_rowq + P_DTHR] : 0.f;
            break;
        }
        case RK_MONO_W: {  // V[L][kpad]; cs_dy[0] >= 0 marks the encoder whose gate is used
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l = (int)((rel + e) / a.kpad), kk = (int)((rel + e) % a.kpad);
                if (l < a.L && kk < rg.n && rg.cs_dy[0] >= 0) {
                    float g1 = a.colsum[a.cs_mono + l * a.kpad + kk];
                    if (a.weight_norm) {
                        const float cw = a.colsum[a.cs_duc + kk];
                        const float n2 = rg.n2[kk], n = sqrtf(n2);
                        if (n2 > 0.f) g1 = fmaf(-(rg.g[kk] * cw / (n2 * n)), th[e], g1);   // (explicit: the fused tail performs the same operations)
                    }
                    gr[e] = g1;
                }
            }
            break;
        }
        case RK_MONO_G: {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = (int)rel + e;
                if (rg.cs_dy[0] >= 0 && kk < rg.n) {
                    const float n2 = rg.n2[kk];
                    gr[e] = n2 > 0.f ? a.colsum[a.cs_duc + kk] / sqrtf(n2) : 0.f;
                }
            }
            break;
        }
        default: break;
    }
    if (rg.reg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gr[e] = fmaf(a.reg_const, th[e], gr[e]);
    }
    return gr;
}

extern "C" __global__ __launch_bounds__(256) void cfl_finalize_kernel(FinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x == a.nblocks_main) {
        // ---- scalars (cfl/models/cfl.py:868-949) ----------------------------------
        __shared__ float regsum_s;
        if (wave == 0) {
            float s = 0.f;
            for (int b = lane; b < a.nregblocks; b += 64) s += a.regpart[b];
            s = wave_sum(s);
            if (lane == 0) regsum_s = 0.5f * a.reg_const * s;
        }
        __syncthreads();
        if (tid == 0)
            write_scalars(a.scalars, a.colsum + a.cs_rowq, regsum_s, a.B, a.use_threshold, a.pos_weight,
                          a.caffe_margin, a.lambda_m, a.thr_copy[0]);
        return;
    }
    // main blocks: one float4 (4 consecutive parameters, same region) per thread; all
    // loads of a thread are independent and issued together.
    const long long base = ((long long)blockIdx.x * 256 + tid) * 4;
    if (base >= a.total) return;
    // region of this float4: all descriptors' bounds are fetched at once (a search loop with an early
    // exit made every iteration a dependent kernel-argument load)
    int k = a.nregions;
    const int b32 = (int)base;   // theta has < 2^31 floats (make_plan)
#pragma unroll
    for (int i = CFL_MAX_REGIONS - 1; i >= 0; --i)
        k = ((b32 >= a.rbeg[i]) & (b32 < a.rend[i])) ? i : k;   // unused slots are empty ranges (0, 0)
    const f32x4 th = *(const f32x4 *)(a.theta + base);
    // the Adam slots are requested together with theta and the slabs (one memory round trip, not two:
    // behind the gradient store the compiler could not hoist them)
    f32x4 mm = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (a.adam_m) {
        mm = *(const f32x4 *)(a.adam_m + base);
        vv = *(const f32x4 *)(a.adam_v + base);
    }
    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
    if (k < a.nregions) {
        // Almost every wave lies inside one region (regions are 64-float aligned, a wave covers 256 floats): its
        // descriptor is then fetched with scalar loads.  Per-lane descriptors (vector loads from the kernel
        // arguments, one more dependent round trip before the slab loads can be issued) only at region boundaries.
        const int ku = __builtin_amdgcn_readfirstlane(k);
        if (__builtin_amdgcn_ballot_w64(k != ku) == 0)
            gr = fin_region_grad(a, a.reg[ku], base, th);
        else
            gr = fin_region_grad(a, a.reg[k], base, th);
    }
    *(f32x4 *)(a.grad + base) = gr;
    if (a.adam_m) {  // fused TF-Adam apply (single-GPU step)
        f32x4 tn = th;
        adam4(tn, mm, vv, gr, a.lr_t, a.b1, a.b2, a.eps);
        *(f32x4 *)(a.adam_m + base) = mm;
        *(f32x4 *)(a.adam_v + base) = vv;
        *(f32x4 *)(a.theta_out + base) = tn;
    }
}

// ---------------------------------------------------------------------------
// TF-1.x Adam, flat
// ---------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void cfl_adam_kernel(float *theta, float *m, float *v,
                                                                 const float *grad, long long n4,
                                                                 float lr_t, float b1, float b2,
                                                                 float eps, float gscale) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n4; i += stride) {
        f32x4 g = ((const f32x4 *)grad)[i] * gscale;
        f32x4 mm = ((f32x4 *)m)[i], vv = ((f32x4 *)v)[i], th = ((f32x4 *)theta)[i];
        adam4(th, mm, vv, g, lr_t, b1, b2, eps);
        ((f32x4 *)m)[i] = mm;
        ((f32x4 *)v)[i] = vv;
        ((f32x4 *)theta)[i] = th;
    }
}

// ... that also writes the kept bf16 planes of the weights it updates (the update of a data-parallel step: theta_planes.h)
extern "C" __global__ __launch_bounds__(256) void cfl_adam_planes_kernel(float *theta, float *m, float *v,
                                                                        const float *grad, long long n4,
                                                                        float lr_t, float b1, float b2,
                                                                        float eps, float gscale, ThetaPlaneRegions pr) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n4; i += stride) {
        f32x4 g = ((const f32x4 *)grad)[i] * gscale;
        f32x4 mm = ((f32x4 *)m)[i], vv = ((f32x4 *)v)[i], th = ((f32x4 *)theta)[i];
        adam4(th, mm, vv, g, lr_t, b1, b2, eps);
        ((f32x4 *)m)[i] = mm;
        ((f32x4 *)v)[i] = vv;
        ((f32x4 *)theta)[i] = th;
        theta_planes_store4(pr, i * 4, th);
    }
}

// ---------------------------------------------------------------------------
// row gather: out[i,:] = table[idx[i],:]   (one wave per row, 16 B per lane)
// ---------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void cfl_gather_kernel(const float *table,
                                                                   const long long *idx,
                                                                   long long n, long long D,
                                                                   float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long row = (long long)blockIdx.x * 4 + wave;
    const long long stride = (long long)gridDim.x * 4;
    const long long d4 = D >> 2;
    for (; row < n; row += stride) {
        const f32x4 *src = (const f32x4 *)(table + idx[row] * D);
        f32x4 *dst = (f32x4 *)(out + row * D);
        for (long long k = lane; k < d4; k += 64) dst[k] = src[k];
    }
}

// ===========================================================================
// host side
// ===========================================================================
static int check_shape(const CflShape *s) {
    if (!s) return set_err(CFL_E_SHAPE, "shape is NULL");
    if (s->D <= 0 || s->D % 64 != 0)
        return set_err(CFL_E_SHAPE, "D=%d must be a positive multiple of 64 (pad the inputs)", s->D);
    if (s->L <= 0 || s->K <= 0) return set_err(CFL_E_SHAPE, "L=%d K=%d must be positive", s->L, s->K);
    if (s->dist_type < 0 || s->dist_type > 2) return set_err(CFL_E_SHAPE, "bad dist_type %d", s->dist_type);
    if (s->act_type < 0 || s->act_type > 3) return set_err(CFL_E_SHAPE, "bad act_type %d", s->act_type);
    if (s->K > 64) return set_err(CFL_E_UNSUPPORTED, "num_components %d > 64", s->K);
    return CFL_OK;
}

extern "C" int cfl_version(void) { return CFL_ABI_VERSION; }

extern "C" int cfl_scalars_status(const float *host_scalars) {
    if (!host_scalars) return set_err(CFL_E_SHAPE, "scalars is NULL");
    if (host_scalars[CFL_S_ERROR] != 0.f)   // (NaN compares unequal to 0 too: a poisoned word is an error)
        return set_err(CFL_E_HANDOFF, "an in-launch hand-off was lost (a workgroup gave up waiting for its partners): the "
                                      "parameters are poisoned with NaN from that step on -- do not checkpoint them");
    return CFL_OK;
}
extern "C" const char *cfl_last_error(void) { return g_err; }

extern "C" int cfl_layout(const CflShape *s, CflLayout *out) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (!out) return set_err(CFL_E_SHAPE, "layout out is NULL");
    memset(out, 0, sizeof(*out));
    int64_t off = 0;
    auto head = [&](CflHead &h, int n, bool present, bool bias, bool gain) {
        h.n = n;
        h.npad = (int)round_up(n, 16);
        h.w = h.b = h.g = -1;
        if (!present) { h.n = h.npad = 0; return; }
        h.w = off; off += (int64_t)h.npad * s->D;
        if (bias) { h.b = off; off += round_up(h.npad, 64); }
        if (gain) { h.g = off; off += round_up(h.npad, 64); }
    };
    const int nenc = s->directed ? 2 : 1;
    const bool proto = s->dist_type != CFL_DIST_SIAMESE;
    for (int e = 0; e < nenc; ++e) {
        head(out->enc[e].outputs, s->L, true, s->has_bias, s->weight_norm);
        head(out->enc[e].proto, s->L * s->K, proto, s->has_bias, s->weight_norm);
        CflHead &m = out->enc[e].mono;
        m.w = m.b = m.g = -1; m.n = m.npad = 0;
        if (s->dist_type == CFL_DIST_MONOMER) {
            m.n = s->K; m.npad = (int)round_up(s->K, 16);
            m.w = off; off += round_up((int64_t)s->L * m.npad, 64);
            if (s->weight_norm) { m.g = off; off += round_up(m.npad, 64); }
        }
    }
    if (!s->directed) out->enc[1] = out->enc[0];
    out->thr = off; off += 64;
    out->total = off;
    return CFL_OK;
}

// ---- execution plan --------------------------------------------------------
// The kernels a plan launches: decided ONCE, in make_plan; run_pairs switches on these, cfl_plan_describe reports them.
enum ProjKernel { PK_EXACT = 0, PK_BX3, PK_STREAM, PK_X3, PK_X3_KEEP };
enum MidKernel { MK_ROW1 = 0, MK_ROW2, MK_ROW4, MK_REG_K4_LQ2, MK_REG_K8_LQ2, MK_REG_K4_LQ4, MK_GENERIC };
enum GradKernel { GK_NONE = 0, GK_HALF_W8, GK_HALF, GK_HALF_SPLIT, GK_X3, GK_X3_LONGRANGE, GK_EXACT };
static const char *const kProjKernelName[] = {"cfl_proj_kernel", "cfl_proj_bx3_kernel", "cfl_proj_stream_kernel",
                                              "cfl_proj_x3_kernel", "cfl_proj_x3_keep_kernel"};
static const char *const kMidKernelName[] = {"cfl_mid_row_kernel<1>", "cfl_mid_row_kernel<2>", "cfl_mid_row_kernel<4>",
                                             "cfl_mid_kernel<4, 2>", "cfl_mid_kernel<8, 2>", "cfl_mid_kernel<4, 4>",
                                             "cfl_mid_kernel<0, 0>"};
static const char *const kGradKernelName[] = {"", "cfl_grad_x3_half_w8_kernel", "cfl_grad_x3_half_kernel",
                                              "cfl_grad_x3_half_split_kernel", "cfl_grad_x3_kernel",
                                              "cfl_grad_x3_longrange_kernel", "cfl_grad_kernel"};

struct Plan {
    CflLayout lay;
    int proj_kernel, mid_kernel, grad_kernel;   // ProjKernel / MidKernel / GradKernel
    int Rx, Rxpad, Rtot;   // extra scoring rows of a training call (RowExtra), padded to 32, and Rpad + Rxpad (rows of the partial slabs)
    bool x_ok;             // ... and whether this plan's kernels can carry them (chunk-at-a-time projection, wave-per-row math)
    int R, Rpad, S, P, nrb, nregblocks;
    int kpad, lpad, Lq;
    bool has_cw, mono;
    // colsum vector offsets
    int cs_dy[2], cs_cw[2], cs_mono, cs_duc, cs_rowq, cs_total;
    // workspace offsets (floats)
    size_t ypart[2], dyf[2], cwf[2], wpart[2];
    bool x3;   // bf16x3 matrix-core path for the weight gradient
    bool xcd;  // XCD-aligned launch order of proj / grad (cfl_xcd_aligned)
    int proj_stream;  // 0: one wait per 128-d chunk (proj_body); 1: streaming form (proj_stream_body)
    bool proj_x3;     // bf16x3 forward with LDS-shared W planes (cfl_proj_x3_kernel); S is then its d split
    bool planes_kept; // the caller keeps the planes beside theta (CflThetaPlanes): no per-call split launch
    bool proj_bx3;    // ... and the chunk-at-a-time projection multiplies them on the bf16 matrix cores (cfl_proj_bx3_kernel)
    size_t wplanes[2];
    int x3_tiles, x3_units, x3_nwg;   // cfl_proj_x3_kernel: 128-row tiles per side, work units (job, tile, d slice), workgroups
    bool grad_half;   // 32-d tiles, no row split (cfl_grad_x3_half_kernel)
    bool grad_w8;     // ... with eight waves per workgroup (cfl_grad_x3_half_w8_kernel)
    bool fused;       // gradient + Adam finished inside the weight-gradient launch (GradFuse)
    size_t handoff;   // workspace offset of the hand-off tickets + flags (ints), nhandoff of each
    int nhandoff;
    int mid_generic, mid_norow;   // debug overrides of the mid kernel choice
    size_t mono_ya, mono_du, mono_duc, rowqf, thr_copy, colsum, regpart, n2, total_floats;
    size_t mid_lds;
    int ys;
};

static inline int pow2_floor(int x) { int p = 1; while (p * 2 <= x) p *= 2; return p; }
// XCD-aligned launch order of proj and grad: with S a multiple of 8 the d slices can be dealt one per XCD
// (S fastest in proj), and grad deals its 64-d tiles to the XCD that projected their slice.
static int debug_env(const char *name);
static inline bool cfl_xcd_aligned(int S, int dtiles) {
    return S % 8 == 0 && dtiles % S == 0 && debug_env("CFL_DEBUG_NOXCD") <= 0;
}

// tuning overrides for experiments (tools/kernel_probe.py): CFL_DEBUG_S / CFL_DEBUG_P
static int debug_env(const char *name) {
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
}

static void side_heads(const CflShape *s, const CflLayout &lay, const CflHead **src,
                       const CflHead **dst) {
    switch (s->dist_type) {
        case CFL_DIST_PCD: *src = &lay.enc[0].proto; *dst = &lay.enc[1].outputs; break;
        case CFL_DIST_MONOMER: *src = &lay.enc[0].outputs; *dst = &lay.enc[1].proto; break;
        default: *src = &lay.enc[0].outputs; *dst = &lay.enc[1].outputs; break;
    }
}

static int make_plan(const CflShape *s, int64_t rows, int groups, bool train, bool planes_kept, Plan *pl, int64_t xrows = 0) {
    int rc = cfl_layout(s, &pl->lay);
    if (rc) return rc;
    if (rows <= 0 || groups < 1 || groups > 2) return set_err(CFL_E_SHAPE, "rows=%lld groups=%d", (long long)rows, groups);
    if (rows * groups > (1ll << 30)) return set_err(CFL_E_SHAPE, "too many rows");
    const CflHead *hs, *hd;
    side_heads(s, pl->lay, &hs, &hd);
    if (hs->npad > 1024 || hd->npad > 1024) return set_err(CFL_E_UNSUPPORTED, "more than 1024 head columns");
    if (pl->lay.total >= (1ll << 31)) return set_err(CFL_E_UNSUPPORTED, "more than 2^31 parameters");
    pl->R = (int)(rows * groups);
    const int njobs = (hs->npad / 16 + 3) / 4 + (hd->npad / 16 + 3) / 4;
    if (njobs > CFL_MAX_JOBS) return set_err(CFL_E_UNSUPPORTED, "too many column chunks");
    // grad row split: aim at >= 256 workgroups (one per CU; measured: fewer, fatter
    // workgroups beat 512 because every extra row range costs a full-size gradient slab)
    int P = 1;
    if (train) {
        const int dtiles = s->D / 64;
        int want = (256 + dtiles * njobs - 1) / (dtiles * njobs);
        P = pow2_floor(want < 1 ? 1 : want);
        // ... and no more than two 64-row chunks per wave (measured: B = 1024 rows per side, P 2 -> 4: -4 us on grad)
        const int by_rows = pow2_floor(pl->R / 512 < 1 ? 1 : pl->R / 512);
        if (P < by_rows) P = by_rows;
        if (P > 8) P = 8;
        while (P > 1 && pl->R / P < 64) P /= 2;
        if (debug_env("CFL_DEBUG_P") > 0) P = debug_env("CFL_DEBUG_P");
    }
    // half tiles without a row split (cfl_grad_x3_half_kernel): when 32-d tiles alone fill the chip and the whole batch
    // is short enough for one workgroup per tile, a gradient tile is complete inside its workgroup and the fused tail
    // needs no hand-off.  Not for the siamese pairing (two sides per tile) and not with the fp32-MFMA contraction.
    pl->grad_half = false;
    if (train && debug_env("CFL_EXACT_FP32") <= 0 && debug_env("CFL_DEBUG_GRAD_HALF") >= 0) {
        const int ht = s->D / 32;
        const bool paired = s->dist_type == CFL_DIST_SIAMESE && !s->directed;
        // (measured: -2.6 us at B = 512 and with weight-norm, -5.9 us at B = 1024, -1.6 us at B = 2048; +5 us at B = 4096; +0.8 us
        // with 192 workgroups on 256 CUs, config 4 -- hence the bounds)
        // Beyond 2048 rows per side the rows are split in two (one published half tile per finisher): measured against
        // P = 1 -2.5 us at B = 1536 and -2.5 .. -3.9 us at B = 2048, +0.9 us at B = 1024; against the 64-d form -1 us at B = 3072
        if (!paired && ht * njobs >= 256 && ht * njobs <= 640 && pl->R <= 6144 && debug_env("CFL_DEBUG_P") <= 0) {
            pl->grad_half = true;
            P = pl->R > 2048 ? 2 : 1;
        }
        // fewer half tiles than CUs but rows enough to split in two (config 4: 64 x 3 = 192 tiles, 2048 rows per side): 384
        // workgroups, one published tile per finisher.  Round 4, same box: config 4 49.9 -> 47.9 us against the 64-d tiles
        // with four row ranges (three published tiles per finisher); without the split 50.3 (profiles/r04_split_ab.txt)
        if (!paired && ht * njobs >= 128 && ht * njobs < 256 && pl->R >= 2048 && pl->R <= 6144 && debug_env("CFL_DEBUG_P") <= 0) {
            pl->grad_half = true;
            P = 2;
        }
        // siamese: side 0's half tile is published, side 1's workgroup of the same tile finishes -- ONE published tile per
        // finisher and twice as many finishers as the 64-d / P = 2 form (config 3: three tiles per finisher)
        if (paired && ht * njobs >= 256 && ht * njobs <= 640 && pl->R <= 2048 && debug_env("CFL_DEBUG_P") <= 0) {
            pl->grad_half = true;
            P = 1;
        }
        // CFL_DEBUG_GRAD_HALF=1: half tiles with whatever row split was chosen above / forced by CFL_DEBUG_P (experiments)
        if (debug_env("CFL_DEBUG_GRAD_HALF") > 0 && round_up(pl->R, 256 * P) <= 8192) pl->grad_half = true;   // (staged row addresses: 64 KB of LDS)
    }
    // matrix-core arithmetic of the weight-gradient contraction: bf16x3 (fp32-equivalent, default) or,
    // with CFL_EXACT_FP32=1 in the environment, the k-ordered fp32 FMA chains of v_mfma_f32_16x16x4_f32
    // (the forward projection always uses the latter: measured, bf16x3 buys nothing there because each
    // W fragment is shared by only two row tiles, so the splits cost what the MFMAs save)
    pl->x3 = debug_env("CFL_EXACT_FP32") <= 0;
    // fused tail: every gradient entry must be complete inside the P workgroups of one (d tile, column job) -- pcd
    // (each side feeds its own head), one encoder; weight-normalised heads get their column coupling c_j from the
    // launch's own reduction blocks.  CFL_DEBUG_NOFUSE=1: the escape hatch (separate finalize launch)
    pl->fused = train && debug_env("CFL_DEBUG_NOFUSE") <= 0;
    pl->P = P;
    pl->Rpad = (int)round_up(pl->R, 256 * P);  // grad: 64-row chunks x 4 waves x P ranges
    // proj d split: one 128-d chunk per wave when that yields enough workgroups
    if (xrows < 0 || xrows > (1 << 24) || (xrows && !train)) return set_err(CFL_E_SHAPE, "extra scoring rows %lld", (long long)xrows);
    pl->Rx = (int)xrows;
    pl->Rxpad = (int)round_up(xrows, 32);
    pl->Rtot = pl->Rpad + pl->Rxpad;
    const int rtiles = (pl->R + 31) / 32 + pl->Rxpad / 32;
    const int nchunks = (s->D / 16 + 7) / 8;
    int S = (512 + rtiles * njobs - 1) / (rtiles * njobs);
    S = pow2_floor(S < 1 ? 1 : S);
    int maxS = (nchunks + 3) / 4;  // at least one chunk per wave
    if (S > maxS) S = pow2_floor(maxS);
    if (S > 16) S = 16;
    if (debug_env("CFL_DEBUG_S") > 0) S = debug_env("CFL_DEBUG_S");
    // bf16x3 forward with shared W planes: 128-row tiles, d split so that the 512 resident workgroups (two per CU) get
    // one or two units each; slices are whole 128-d chunks
    pl->proj_x3 = false;
    pl->planes_kept = planes_kept && train;
    {
        const int ov = debug_env("CFL_DEBUG_PROJ_X3");
        const int tiles = (pl->R + pl->Rxpad + 127) / 128;   // (extra scoring rows only steer the choice: the LDS-shared form cannot carry them)
        int rs = 1;
        const int want_units = debug_env("CFL_DEBUG_X3_UNITS") > 0 ? debug_env("CFL_DEBUG_X3_UNITS") : 384;
        // slices of at least 512 d -- 256 d (eight 32-d steps per unit) only to reach 256 units at all (configs 3 / 4)
        const int min_slice = debug_env("CFL_DEBUG_X3_MINSLICE") > 0 ? debug_env("CFL_DEBUG_X3_MINSLICE") : 512;
        while (rs < 16 && njobs * tiles * rs < want_units && (s->D / 128) % (2 * rs) == 0 && s->D / (2 * rs) >= min_slice) rs *= 2;
        while (rs < 16 && njobs * tiles * rs < 256 && (s->D / 128) % (2 * rs) == 0 && s->D / (2 * rs) >= 256) rs *= 2;
        const int units = njobs * tiles * rs;
        const bool ok = s->D % 128 == 0 && (s->D / 128) % rs == 0 && units >= 256 && pl->x3;
        // From 4096 rows per side for scoring calls (-7 % at 4096 pairs, +11 % at 2048), from 3072 (B >= 1536) in training --
        // there x is loaded with the default cache policy while both sides are within reach of the Infinity Cache, so that
        // the weight gradient's re-read hits it.  Bench medians, same box, step us with / without: B = 1536 69.2 / 76.3,
        // 2048 81.0 / 84.0, 3072 109.2 / 117.5, 4096 140.8 / 145.2 (and 143.2 with streamed x loads)
        // With planes kept beside theta by the fused training step (CflThetaPlanes) the per-call W split -- a launch of
        // ~4 us -- is gone, and the threshold of the training step drops to `kept_rows` (measured: see DESIGN section 4)
        const int kept_rows = debug_env("CFL_DEBUG_X3_KEPT_ROWS") > 0 ? debug_env("CFL_DEBUG_X3_KEPT_ROWS") : 3072;
        if (ok && ov >= 0 && (ov > 0 || pl->R >= (train ? (pl->planes_kept ? kept_rows : 3072) : 4096))) {
            pl->proj_x3 = true;
            pl->x3_tiles = tiles;
            pl->x3_units = units;
            pl->x3_nwg = units < 512 ? units : 512;
            S = rs;
        }
    }
    pl->S = S;
    pl->xcd = !pl->proj_x3 && cfl_xcd_aligned(S, s->D / 64);
    // streaming projection when a wave owns at least four 128-d chunks (measured: equal to the chunk-at-a-time form at
    // two, -11 % at eight; CFL_DEBUG_PROJ_STREAM: -1 never, 1 always)
    {
        const int per_wave = nchunks / (4 * S);
        const int ov = debug_env("CFL_DEBUG_PROJ_STREAM");
        pl->proj_stream = ov < 0 ? 0 : ov > 0 ? 1 : (per_wave >= 4 ? 1 : 0);
    }
    // bf16x3 arithmetic on the chunk-at-a-time skeleton (cfl_proj_bx3_kernel): whenever the caller keeps the planes of
    // theta current (the fused single-GPU training step) and the wave owns fewer than four chunks
    // (CFL_DEBUG_PROJ_BX3=1: also without kept planes -- per-call split into the workspace, the tests' reference run; -1: never)
    pl->proj_bx3 = (pl->planes_kept || debug_env("CFL_DEBUG_PROJ_BX3") > 0) && pl->x3 && !pl->proj_x3 &&
                   !pl->proj_stream && debug_env("CFL_DEBUG_PROJ_BX3") >= 0;
    pl->mid_generic = debug_env("CFL_DEBUG_MID_GENERIC") > 0;
    pl->mid_norow = debug_env("CFL_DEBUG_MID_NOROW") != 0;
    pl->nrb = pl->Rpad / MID_RB;
    pl->kpad = pl->lay.enc[0].mono.npad;
    pl->lpad = (int)round_up(s->L, 16);
    pl->Lq = (s->L + 15) / 16;
    pl->has_cw = train && s->weight_norm;
    pl->mono = s->dist_type == CFL_DIST_MONOMER;
    pl->nregblocks = (int)((pl->lay.total / 64 + 63) / 64);
    // colsum vector
    int cs = 0;
    auto cst = [&](int n) { int o = cs; cs += n; return o; };
    pl->cs_dy[0] = cst(hs->npad); pl->cs_dy[1] = cst(hd->npad);
    pl->cs_cw[0] = cst(hs->npad); pl->cs_cw[1] = cst(hd->npad);
    pl->cs_mono = cst(pl->mono ? s->L * pl->kpad : 0);
    pl->cs_duc = cst(pl->mono ? pl->kpad : 0);
    pl->cs_rowq = cst(16);
    pl->cs_total = cs;
    // workspace
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += round_up((int64_t)n, 64); return o; };
    const size_t rp = pl->Rpad;
    pl->ypart[0] = take((size_t)S * hs->npad * pl->Rtot);
    pl->ypart[1] = take((size_t)S * hd->npad * pl->Rtot);
    if (train) {
        pl->dyf[0] = take(hs->npad * rp);
        pl->dyf[1] = take(hd->npad * rp);
        pl->cwf[0] = take(pl->has_cw ? hs->npad * rp : 0);
        pl->cwf[1] = take(pl->has_cw ? hd->npad * rp : 0);
        pl->mono_ya = take(pl->mono ? pl->lpad * rp : 0);
        pl->mono_du = take(pl->mono ? pl->kpad * rp : 0);
        pl->mono_duc = take(pl->mono ? pl->kpad * rp : 0);
        pl->rowqf = take(16 * rp);
        pl->thr_copy = take(64);
        pl->colsum = take(cs);
        pl->wpart[0] = take((size_t)P * hs->npad * s->D);
        pl->wpart[1] = take((size_t)P * hd->npad * s->D);
        pl->regpart = take((size_t)pl->nregblocks);
        pl->nhandoff = njobs * (s->D / 32);   // (half tiles; the 64-d forms use the first half)
        pl->handoff = take(2 * (size_t)pl->nhandoff + 64);   // tickets, flags, + the reduction blocks' counter
    }
    const bool ws_planes = (pl->proj_x3 || pl->proj_bx3) && !pl->planes_kept;
    {
        // eight waves per workgroup (two per SIMD) for the unsplit half tiles: one wave per SIMD leaves the row loop's load
        // latency and its split arithmetic (8 VALU instructions per MFMA: SQ_INSTS_VALU) with nothing to overlap with.  Not for
        // the shared siamese heads: their fused form runs the split kernel's four-wave order, and the separate finalize launch
        // must keep adding the same partial sums (CFL_DEBUG_GRAD_W8=-1: four waves everywhere)
        pl->grad_w8 = train && pl->grad_half && pl->P == 1 && pl->x3 && pl->Rpad % 512 == 0 &&
                      !(s->dist_type == CFL_DIST_SIAMESE && !s->directed) && debug_env("CFL_DEBUG_GRAD_W8") >= 0;
    }
    pl->wplanes[0] = take(ws_planes ? (size_t)hs->npad * s->D * 3 / 2 : 0);   // bf16 planes: 6 bytes per weight
    pl->wplanes[1] = take(ws_planes ? (size_t)hd->npad * s->D * 3 / 2 : 0);
    pl->n2 = take(2 * 6 * 1024);  // squared column norms + gain snapshot of up to 6 heads
    pl->total_floats = off;
    const int ks = s->dist_type == CFL_DIST_PCD ? s->K : 1;
    const int kd = s->dist_type == CFL_DIST_MONOMER ? s->K : 1;
    const int slots = (ks + kd) * pl->Lq * (s->weight_norm ? 2 : 1) + pl->Lq + 4 * s->K;
    pl->ys = hs->npad + hd->npad + 4;
    const int mw = pl->mono ? ((s->L * pl->kpad + 3) & ~3) : 0;
    pl->mid_lds = ((size_t)(MID_RB + 2) * pl->ys + mw + (size_t)slots * 64) * sizeof(float);
    if (pl->mid_lds > 160 * 1024) return set_err(CFL_E_UNSUPPORTED, "L*K too large for the mid kernel");
    // ---- the kernels ---------------------------------------------------------------------------------------------------
    if (pl->proj_x3) {
        // training with both sides' rows within reach of the 256 MB Infinity Cache: x is loaded with the default policy, so
        // that the weight gradient's re-read hits it
        const double keep_bytes = debug_env("CFL_DEBUG_X3_KEEP_MB") > 0 ? debug_env("CFL_DEBUG_X3_KEEP_MB") * 1e6 : 300e6;
        const bool keep = train && 2.0 * pl->R * s->D * 4.0 <= keep_bytes && debug_env("CFL_DEBUG_PROJ_X3_KEEP") >= 0;
        pl->proj_kernel = keep ? PK_X3_KEEP : PK_X3;
    } else {
        pl->proj_kernel = pl->proj_bx3 ? PK_BX3 : pl->proj_stream ? PK_STREAM : PK_EXACT;
    }
    {
        // one wave per row: pcd (any K <= 64) and siamese with up to 256 padded columns per side.  Many rows of the small-K
        // shapes run the 4-rows-per-wave register form instead.  Measured: scoring 32768 pairs 13 vs 24 us; training 32768
        // rows (B = 8192) 14.4 vs 20.4 us -- but 9.7 vs 11.2 us the other way at B = 512 ... 2048 (the wave-per-row form is
        // the shorter latency chain, the register form the smaller instruction count)
        const int wide = hs->npad > hd->npad ? hs->npad : hd->npad;
        const bool small_reg = (s->K <= 8 && pl->Lq <= 2) || (s->K <= 4 && pl->Lq <= 4);
        const bool row_ok = (s->dist_type == CFL_DIST_PCD || s->dist_type == CFL_DIST_SIAMESE) && wide <= 256 &&
                            s->K <= 64 && !pl->mid_norow && !(small_reg && pl->R >= (train ? 16384 : 4096));
        const bool generic_only = pl->mid_generic != 0;
        pl->mid_kernel = generic_only ? MK_GENERIC
                         : row_ok ? (wide <= 64 ? MK_ROW1 : wide <= 128 ? MK_ROW2 : MK_ROW4)
                         : (s->K <= 4 && pl->Lq <= 2) ? MK_REG_K4_LQ2
                         : (s->K <= 8 && pl->Lq <= 2) ? MK_REG_K8_LQ2
                         : (s->K <= 4 && pl->Lq <= 4) ? MK_REG_K4_LQ4 : MK_GENERIC;
    }
    pl->x_ok = (pl->proj_kernel == PK_EXACT || pl->proj_kernel == PK_BX3) && pl->mid_kernel <= MK_ROW4;
    pl->grad_kernel = GK_NONE;
    if (train) {
        // (the siamese pairing of the FUSED tail always takes the hand-off kernel: two sides feed one head)
        const bool paired = pl->fused && s->dist_type == CFL_DIST_SIAMESE && !s->directed;
        pl->grad_kernel = (pl->grad_half && pl->P == 1 && !paired) ? (pl->grad_w8 ? GK_HALF_W8 : GK_HALF)
                          : pl->grad_half ? GK_HALF_SPLIT
                          : pl->x3 ? (pl->Rpad / pl->P <= 8192 ? GK_X3 : GK_X3_LONGRANGE) : GK_EXACT;
    }
    return CFL_OK;
}

// HOST-ONLY introspection (include/cfl_hip.h): what a call of this shape will launch -- from the same plan run_pairs executes
extern "C" int cfl_plan_describe(const CflShape *s, int64_t rows, int32_t groups, int32_t train, int32_t planes_kept,
                                 CflPlanInfo *out) {
    if (!out) return set_err(CFL_E_SHAPE, "cfl_plan_describe: out is NULL");
    if (train && groups != 2) return set_err(CFL_E_SHAPE, "cfl_plan_describe: a training call has 2 pair groups");
    Plan pl;
    const bool kept = planes_kept && train && debug_env("CFL_DEBUG_NOFUSE") <= 0;
    int rc = make_plan(s, rows, groups, train != 0, kept, &pl);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    snprintf(out->proj, sizeof(out->proj), "%s", kProjKernelName[pl.proj_kernel]);
    snprintf(out->mid, sizeof(out->mid), "%s", kMidKernelName[pl.mid_kernel]);
    snprintf(out->grad, sizeof(out->grad), "%s", kGradKernelName[pl.grad_kernel]);
    snprintf(out->tail, sizeof(out->tail), "%s", (train && !pl.fused) ? "cfl_finalize_kernel" : "");
    const bool per_call_split = (pl.proj_x3 || pl.proj_bx3) && !pl.planes_kept;
    out->launches = 2 + (train ? 1 : 0) + ((train && !pl.fused) ? 1 : 0) + (per_call_split ? 1 : 0);
    out->per_call_plane_split = per_call_split;
    out->S = pl.S; out->P = pl.P; out->rows_padded = pl.Rpad;
    out->proj_tile_rows = pl.proj_x3 ? 128 : 32;
    const CflHead *hs, *hd;
    side_heads(s, pl.lay, &hs, &hd);
    const int njobs = (hs->npad / 16 + 3) / 4 + (hd->npad / 16 + 3) / 4;
    out->column_jobs = njobs;
    out->proj_workgroups = pl.proj_x3 ? pl.x3_nwg : ((pl.R + 31) / 32) * pl.S * njobs;
    out->grad_tile_d = train ? (pl.grad_half ? 32 : 64) : 0;
    out->grad_workgroups = train ? (s->D / out->grad_tile_d) * pl.P * njobs : 0;
    out->grad_waves = train ? (pl.grad_kernel == GK_HALF_W8 ? 8 : 4) : 0;
    out->fused_tail = train && pl.fused;
    out->reads_planes = pl.proj_x3 || pl.proj_bx3;
    out->xcd_aligned = pl.xcd;
    return CFL_OK;
}

extern "C" size_t cfl_workspace_bytes(const CflShape *s, int64_t rows, int32_t groups) {
    // one workspace serves every call shape of (rows, groups): training with and without kept planes, and the scoring
    // call of the same size (cfl_pair_scores_idx4 runs the non-training plan with groups == 2)
    size_t need = 0;
    for (int train = 0; train < 2; ++train)
        for (int kept = 0; kept < 2; ++kept) {
            if (train && groups != 2) continue;
            if (kept && !train) continue;
            Plan pl;
            if (make_plan(s, rows, groups, train != 0, kept != 0, &pl)) return 0;
            if (pl.total_floats > need) need = pl.total_floats;
            // ... and the training step that carries a validation batch of the same size as extra scoring rows
            if (train && make_plan(s, rows, groups, true, kept != 0, &pl, 2 * rows) == CFL_OK && pl.total_floats > need)
                need = pl.total_floats;
        }
    return need * sizeof(float);
}

extern "C" size_t cfl_theta_planes_bytes(const CflShape *s) {
    CflLayout lay;
    if (cfl_layout(s, &lay)) return 0;
    return (size_t)lay.total * 6;   // three bf16 per theta float (only the weight matrices' ranges are ever written)
}

static NormDev make_norm(const CflNorm *n, float *in_mul) {
    NormDev d;
    d.mul = n ? n->mul : 1.f;
    d.add = n ? n->add : 0.f;
    d.lo = (n && n->has_lo) ? n->lo : -INFINITY;
    d.hi = (n && n->has_hi) ? n->hi : INFINITY;
    d.elementwise = n && (n->add != 0.f || n->has_lo || n->has_hi);
    d.valid = (n && n->valid_cols > 0) ? n->valid_cols : 0x7fffffff;
    *in_mul = d.elementwise ? 1.f : d.mul;
    return d;
}

// weight-norm squared column norms for every head of both encoders: arguments of the colnorm slice of the projection launch
static int fill_colnorm(const CflShape *s, const Plan &pl, const float *theta, float *ws,
                        int n2_off[2][3], ColnormArgs *out) {
    ColnormArgs &ca = *out;
    memset(&ca, 0, sizeof(ca));
    ca.theta = theta;
    ca.n2 = ws + pl.n2;
    ca.gcopy = ws + pl.n2 + 6 * 1024;
    ca.D = s->D;
    int nh = 0, off = 0, ncols = 0;
    const int nenc = s->directed ? 2 : 1;
    for (int e = 0; e < 2; ++e) {
        const CflHead *hh[3] = {&pl.lay.enc[e].outputs, &pl.lay.enc[e].proto, &pl.lay.enc[e].mono};
        for (int k = 0; k < 3; ++k) {
            if (e >= nenc) { n2_off[e][k] = n2_off[0][k]; continue; }
            n2_off[e][k] = -1;
            if (hh[k]->w < 0) continue;
            ca.w_off[nh] = hh[k]->w;
            ca.g_off[nh] = hh[k]->g;
            ca.npad[nh] = hh[k]->npad;
            ca.n2_off[nh] = off;
            ca.strided[nh] = k == 2;
            ca.rowlen[nh] = k == 2 ? s->L : s->D;
            n2_off[e][k] = off;
            off += hh[k]->npad;
            ncols += hh[k]->npad;
            ++nh;
        }
    }
    ca.nheads = nh;
    ca.ncols = (s->weight_norm && ncols > 0) ? ncols : 0;   // 0: no colnorm slice in the projection launch
    return CFL_OK;
}

struct SideRt {
    const CflHead *head;
    int enc;     // encoder index of the head
    int which;   // 0 outputs, 1 proto
};

struct AdamFuse { float *theta, *m, *v; float lr_t, b1, b2, eps; };

// Indexed row source of a call (cfl_pair_*_idx): rows of `table` picked by 2 * groups index streams.
struct IndexSrc { const float *table; int64_t table_rows; const int32_t *const *idx; int64_t stride; };
// Extra scoring rows of a training call (RowExtra): bx pairs per group, idx = {src g0, dst g0, src g1, dst g1}; their 2 bx scores
// go to `scores` and a second copy of the step's scalars to `scalars_copy` (both may be host-mapped memory)
struct ExtraSrc { const float *table; int64_t table_rows; const int32_t *idx[4]; int64_t stride; int64_t bx; float *scores, *scalars_copy; };

// The plan of a (shape, rows, groups, train) combination never changes within a process (the tuning overrides
// are read from the environment once per combination): a training loop re-plans nothing per step.
static std::atomic<int> g_env_generation{0};
extern "C" int cfl_reload_env(void) { return ++g_env_generation; }

static int cached_plan(const CflShape *s, int64_t rows, int groups, bool train, bool kept, Plan *out, int64_t xrows = 0) {
    struct Entry { CflShape s; int64_t rows, xrows; int groups; bool train, kept; Plan pl; };
    static thread_local std::vector<Entry> cache;
    static thread_local int seen_generation = 0;
    if (seen_generation != g_env_generation.load()) {
        cache.clear();
        seen_generation = g_env_generation.load();
    }
    if (!s) return set_err(CFL_E_SHAPE, "shape is NULL");
    for (const Entry &e : cache)
        if (e.rows == rows && e.xrows == xrows && e.groups == groups && e.train == train && e.kept == kept && memcmp(&e.s, s, sizeof(CflShape)) == 0) {
            *out = e.pl;
            return CFL_OK;
        }
    int rc = make_plan(s, rows, groups, train, kept, out, xrows);
    if (rc) return rc;
    if (cache.size() >= 64) cache.erase(cache.begin());
    cache.push_back({*s, rows, xrows, groups, train, kept, *out});
    return CFL_OK;
}

static int run_pairs(const CflShape *s, const CflNorm *norm, const CflLossCfg *loss,
                     const float *const *x, int groups, int64_t rows, const float *theta,
                     float *grad, float *scalars, float *scores, float *dists, void *workspace,
                     size_t workspace_bytes, hipStream_t st, const AdamFuse *adam = nullptr,
                     const IndexSrc *isrc = nullptr, CflThetaPlanes *kept = nullptr, const ExtraSrc *xs = nullptr) {
    const bool train = grad != nullptr;
    Plan pl;
    if (kept && (!kept->buf || ((uintptr_t)kept->buf & 15))) return set_err(CFL_E_SHAPE, "theta planes buffer NULL or misaligned");
    // a training call with a kept plane buffer projects from it (splitting theta into it first when it is stale).  The fused
    // Adam tail then WRITES the planes of the updated weights (`adam`); a call that leaves theta alone (the forward /
    // backward of a data-parallel step: the update is cfl_adam_tf_planes, after the exchange) only reads them
    const bool keeping = kept && train && debug_env("CFL_DEBUG_NOFUSE") <= 0;
    int rc = cached_plan(s, rows, groups, train, keeping, &pl, xs ? 2 * xs->bx : 0);
    if (rc) return rc;
    if (xs) {
        if (!train || !pl.x_ok || !pl.fused)
            return set_err(CFL_E_UNSUPPORTED, "extra scoring rows need the chunk-at-a-time projection, the wave-per-row math and the fused tail");
        if (!xs->table || ((uintptr_t)xs->table & 15) || xs->table_rows <= 0 || xs->table_rows >= (1ll << 31) || xs->bx <= 0 ||
            xs->stride <= 0 || xs->stride > (1 << 20) || !xs->scores)
            return set_err(CFL_E_SHAPE, "extra scoring rows: table / stride / row count / scores");
        for (int i = 0; i < 4; ++i)
            if (!xs->idx[i] || ((uintptr_t)xs->idx[i] & 3)) return set_err(CFL_E_SHAPE, "extra scoring rows: index stream %d NULL or misaligned", i);
    }
    if (!theta || !workspace) return set_err(CFL_E_SHAPE, "NULL theta/workspace");
    if (workspace_bytes < pl.total_floats * sizeof(float))
        return set_err(CFL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes,
                       pl.total_floats * sizeof(float));
    if (((uintptr_t)workspace & 15) || ((uintptr_t)theta & 15))
        return set_err(CFL_E_SHAPE, "theta / workspace must be 16-byte aligned");
    if (isrc) {
        if (!isrc->table || ((uintptr_t)isrc->table & 15) || isrc->table_rows <= 0 || isrc->table_rows >= (1ll << 31))
            return set_err(CFL_E_SHAPE, "feature table NULL / misaligned / row count %lld out of range",
                           (long long)isrc->table_rows);
        if (!isrc->idx || isrc->stride <= 0 || isrc->stride > (1 << 20))
            return set_err(CFL_E_SHAPE, "index streams NULL or stride %lld out of range", (long long)isrc->stride);
        for (int i = 0; i < 2 * groups; ++i)
            if (!isrc->idx[i] || ((uintptr_t)isrc->idx[i] & 3))
                return set_err(CFL_E_SHAPE, "index stream %d NULL or misaligned", i);
    } else {
        for (int i = 0; i < 2 * groups; ++i)
            if (!x[i] || ((uintptr_t)x[i] & 15)) return set_err(CFL_E_SHAPE, "input %d NULL or misaligned", i);
    }
    float *ws = (float *)workspace;
    float in_mul;
    NormDev nd = make_norm(norm, &in_mul);
    const size_t rp = pl.Rpad;
    const int G = s->D / 16, RG = pl.Rpad / 16;

    SideRt side[2];
    switch (s->dist_type) {
        case CFL_DIST_PCD: side[0] = {&pl.lay.enc[0].proto, 0, 1}; side[1] = {&pl.lay.enc[1].outputs, 1, 0}; break;
        case CFL_DIST_MONOMER: side[0] = {&pl.lay.enc[0].outputs, 0, 0}; side[1] = {&pl.lay.enc[1].proto, 1, 1}; break;
        default: side[0] = {&pl.lay.enc[0].outputs, 0, 0}; side[1] = {&pl.lay.enc[1].outputs, 1, 0}; break;
    }
    // x layout: train: pos_src,pos_dst,neg_src,neg_dst ; score: src,dst
    RowSrc rsrc[2];
    for (int sd = 0; sd < 2; ++sd) {
        RowSrc &r = rsrc[sd];
        memset(&r, 0, sizeof(r));
        if (isrc) {
            r.x0 = r.x1 = isrc->table;
            r.ix0 = isrc->idx[sd];
            r.ix1 = groups == 2 ? isrc->idx[2 + sd] : isrc->idx[sd];
            r.istride = (int)isrc->stride;
            r.last_row = (unsigned)(isrc->table_rows - 1);
        } else {
            r.x0 = x[sd];
            r.x1 = groups == 2 ? x[2 + sd] : x[sd];
        }
    }

    int n2_off[2][3] = {{-1, -1, -1}, {-1, -1, -1}};
    ColnormArgs cna;
    fill_colnorm(s, pl, theta, ws, n2_off, &cna);
    const float *n2base = ws + pl.n2;
    const float *gbase = ws + pl.n2 + 6 * 1024;

    // ---- proj ---------------------------------------------------------------
    ProjArgs pa;
    {
        memset(&pa, 0, sizeof(pa));
        int nj = 0;
        for (int sd = 0; sd < 2; ++sd) {
            const CflHead *h = side[sd].head;
            const int tiles = h->npad / 16;
            for (int c0 = 0; c0 < tiles; c0 += 4) {
                ProjJob &j = pa.job[nj++];
                j.side = sd;
                j.wf = theta + h->w + (size_t)c0 * G * 256;
                j.ypart = ws + pl.ypart[sd] + (size_t)c0 * 16;
                j.sstride = (long long)h->npad * pl.Rtot;
                j.nt = tiles - c0 < 4 ? tiles - c0 : 4;
                j.npad = h->npad;
            }
        }
        pa.rows[0] = rsrc[0]; pa.rows[1] = rsrc[1];
        pa.B = (int)rows; pa.R = pl.R; pa.Rpad = pl.Rpad; pa.D = s->D; pa.S = pl.S; pa.norm = nd;
        pa.xcd = pl.xcd;
        pa.njobs = nj; pa.cn = cna;
        pa.xr.tile0 = 0;              // no extra scoring rows
        if (xs) {
            pa.xr.table = xs->table;
            for (int sd = 0; sd < 2; ++sd)
                for (int g = 0; g < 2; ++g) pa.xr.ix[sd][g] = xs->idx[2 * g + sd];
            pa.xr.istride = (int)xs->stride;
            pa.xr.last_row = (unsigned)(xs->table_rows - 1);
            pa.xr.row0 = pl.Rpad; pa.xr.n = pl.Rx; pa.xr.bx = (int)xs->bx; pa.xr.tile0 = pl.Rxpad / 32;
        }
        const bool cn_slice = cna.ncols > 0;
        if (cn_slice && nj >= CFL_MAX_JOBS) return set_err(CFL_E_UNSUPPORTED, "too many column chunks");
        if (cn_slice) pa.job[nj].nt = 0;      // marks the colnorm slice
        const int nz = nj + (cn_slice ? 1 : 0);
        // W planes of the bf16x3 forms: kept beside theta by the caller (current: nothing to do; stale: split into the kept
        // buffer), or split per call into the workspace.  job_planes[j] = planes of job j's first column tile
        const unsigned short *job_planes[CFL_MAX_JOBS] = {};
        if (pl.proj_x3 || pl.proj_bx3) {
            const int Q = s->D / 32;
            WPlanesArgs wa;
            memset(&wa, 0, sizeof(wa));
            wa.G = G;
            long long items = 0;
            int jn = 0;
            const bool shared_head = side[0].head->w == side[1].head->w;   // siamese: one head, one set of planes
            for (int sd = 0; sd < 2; ++sd) {
                const CflHead *h = side[sd].head;
                unsigned short *planes = keeping ? (unsigned short *)kept->buf + 3 * h->w
                                                 : (unsigned short *)(ws + pl.wplanes[shared_head ? 0 : sd]);
                if (!(shared_head && sd == 1)) {
                    wa.wf[sd] = theta + h->w; wa.planes[sd] = planes; wa.ntiles[sd] = h->npad / 16;
                    items += (long long)(h->npad / 16) * Q * 64;
                }
                for (int c0 = 0; c0 < h->npad / 16; c0 += 4, ++jn) job_planes[jn] = planes + (size_t)c0 * Q * 3 * 512;
            }
            if (!(keeping && kept->valid)) {
                ProfScope psw(st, CFL_K_COLNORM);   // (profile slot reused: the per-call split of W into bf16 planes)
                hipLaunchKernelGGL(cfl_wplanes_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, wa);
            }
        }
        if (pl.proj_x3) {
            Px3Args xa;
            memset(&xa, 0, sizeof(xa));
            const int Q = s->D / 32;
            for (int jn = 0; jn < nj; ++jn) {
                xa.job[jn] = pa.job[jn];
                xa.job[jn].wf = (const float *)job_planes[jn];
            }
            int k = 0;
            for (int nt = 4; nt >= 1; --nt)
                for (int i = 0; i < nj; ++i)
                    if (pa.job[i].nt == nt) xa.order[k++] = i;
            xa.rows[0] = rsrc[0]; xa.rows[1] = rsrc[1];
            xa.B = (int)rows; xa.R = pl.R; xa.D = s->D; xa.S = pl.S; xa.njobs = nj;
            xa.tiles = pl.x3_tiles; xa.nunits = pl.x3_units; xa.nwg = pl.x3_nwg; xa.Kq = Q / pl.S;
            xa.norm = nd;
            // weight-norm: the column norms ride in this launch as its first workgroups (as they ride in cfl_proj_kernel as
            // a z-slice): no launch of their own
            xa.cn = cna;
            xa.ncn = cn_slice ? (cna.ncols < 64 ? cna.ncols : 64) : 0;
            ProfScope ps(st, CFL_K_PROJ);
            if (pl.proj_kernel == PK_X3_KEEP) hipLaunchKernelGGL(cfl_proj_x3_keep_kernel, dim3(pl.x3_nwg + xa.ncn), dim3(256), PX3_LDS_BYTES, st, xa);
            else hipLaunchKernelGGL(cfl_proj_x3_kernel, dim3(pl.x3_nwg + xa.ncn), dim3(256), PX3_LDS_BYTES, st, xa);
        } else {
        const int rtiles = (pl.R + 31) / 32 + pl.Rxpad / 32;
        dim3 grid(rtiles, pl.S, nz);
        if (pa.xcd) grid = dim3(pl.S, rtiles, nz);
        ProfScope ps(st, CFL_K_PROJ);
        // The colnorm slice goes FIRST in dispatch order (z = 0): its blocks are short, and as the last z-slice they only
        // started once projection workgroups had retired -- the launch ended a colnorm round trip later than it had to
        ProjArgs pz = pa;
        if (pl.proj_bx3)
            for (int jn = 0; jn < nj; ++jn) pz.job[jn].wf = (const float *)job_planes[jn];
        if (cn_slice) {
            for (int jn = nj; jn > 0; --jn) pz.job[jn] = pz.job[jn - 1];
            memset(&pz.job[0], 0, sizeof(pz.job[0]));   // nt == 0 marks the colnorm slice
        }
        // 32 KiB: cross-wave sum (the 4 (8) KiB/wave transpose tiles alias it)
        switch (pl.proj_kernel) {
            case PK_BX3: hipLaunchKernelGGL(cfl_proj_bx3_kernel, grid, dim3(256), 4 * 8 * 64 * sizeof(f32x4), st, pz); break;
            case PK_STREAM: hipLaunchKernelGGL(cfl_proj_stream_kernel, grid, dim3(256), 4 * 8 * 64 * sizeof(f32x4), st, pz); break;
            default: hipLaunchKernelGGL(cfl_proj_kernel, grid, dim3(256), 4 * 8 * 64 * sizeof(f32x4), st, pz); break;
        }
        }
    }

    // ---- mid ----------------------------------------------------------------
    MidArgs ma;
    memset(&ma, 0, sizeof(ma));
    for (int sd = 0; sd < 2; ++sd) {
        const CflHead *h = side[sd].head;
        MidSide &m = ma.side[sd];
        m.ypart = ws + pl.ypart[sd];
        m.sstride = (long long)h->npad * pl.Rtot;
        m.b = h->b >= 0 ? theta + h->b : nullptr;
        m.g = h->g >= 0 ? theta + h->g : nullptr;
        m.n2 = s->weight_norm ? n2base + n2_off[side[sd].enc][side[sd].which] : nullptr;
        m.dyf = train ? ws + pl.dyf[sd] : nullptr;
        m.cwf = pl.has_cw ? ws + pl.cwf[sd] : nullptr;
        m.n = h->n; m.npad = h->npad;
        m.is_proto = side[sd].which == 1;
    }
    const CflHead &mono = pl.lay.enc[0].mono;
    if (pl.mono) {
        ma.mono_w = theta + mono.w;
        ma.mono_g = mono.g >= 0 ? theta + mono.g : nullptr;
        ma.mono_n2 = s->weight_norm ? n2base + n2_off[0][2] : nullptr;
        if (train) {
            ma.mono_ya = ws + pl.mono_ya; ma.mono_du = ws + pl.mono_du; ma.mono_duc = ws + pl.mono_duc;
        }
    }
    ma.kpad = pl.kpad; ma.lpad = pl.lpad;
    ma.S = pl.S; ma.L = s->L; ma.K = s->K; ma.Lq = pl.Lq; ma.dist_type = s->dist_type;
    ma.act = s->act_type; ma.weight_norm = s->weight_norm; ma.in_mul = in_mul;
    ma.thr = theta + pl.lay.thr;
    ma.B = (int)rows; ma.R = pl.R; ma.Rpad = pl.Rpad;
    ma.train = train;
    if (train) {
        ma.use_threshold = loss->use_threshold;
        ma.pos_weight = loss->pos_weight; ma.caffe_margin = loss->caffe_margin; ma.lambda_m = loss->lambda_m;
        ma.rowqf = ws + pl.rowqf;
        ma.thr_copy = ws + pl.thr_copy;
        ma.regpart = ws + pl.regpart;
        ma.theta = theta;
        if (pl.fused) {   // (P > 1, weight-norm c_j hand-off, siamese pairing: the flags are cheap to clear always)
            ma.zero_i = (int *)(ws + pl.handoff);
            ma.nzero = 2 * pl.nhandoff + 1;
        }
    }
    ma.scores = scores; ma.dists = dists;
    ma.nrb = train ? pl.nrb : (pl.R + MID_RB - 1) / MID_RB;
    ma.ys = pl.ys;
    if (xs) { ma.xrow0 = pl.Rpad; ma.xn = pl.Rx; ma.nxb = pl.Rxpad / 4; ma.xscores = xs->scores; }


    // regions (shared by the mid regulariser blocks and finalize)
    FinArgs fa;
    memset(&fa, 0, sizeof(fa));
    int nreg_blocks = 0;
    if (train) {
        int nr = 0;
        const int nenc = s->directed ? 2 : 1;
        auto add = [&](int kind, int64_t off, int64_t cnt, int reg, int npad, int n) -> Region & {
            Region &r = fa.reg[nr++];
            r.off = off; r.cnt = round_up(cnt, 64); r.kind = kind; r.reg = reg; r.npad = npad; r.n = n;
            r.slab[0] = r.slab[1] = nullptr;
            r.cs_dy[0] = r.cs_dy[1] = r.cs_cw[0] = r.cs_cw[1] = -1;
            r.g = r.n2 = nullptr;
            return r;
        };
        const int regon = loss->reg_const > 0.f;
        for (int e = 0; e < nenc; ++e) {
            const CflHead *hh[2] = {&pl.lay.enc[e].outputs, &pl.lay.enc[e].proto};
            for (int k = 0; k < 2; ++k) {
                const CflHead *h = hh[k];
                if (h->w < 0) continue;
                Region &rw = add(RK_W, h->w, (int64_t)h->npad * s->D, regon, h->npad, h->n);
                Region *rb = h->b >= 0 ? &add(RK_BIAS, h->b, h->npad, regon, h->npad, h->n) : nullptr;
                Region *rgn = h->g >= 0 ? &add(RK_GAIN, h->g, h->npad, 0, h->npad, h->n) : nullptr;
                int ns = 0;
                for (int sd = 0; sd < 2; ++sd) {
                    // a side contributes when it projects through this head
                    const bool same_enc = s->directed ? side[sd].enc == e : true;
                    if (side[sd].head->w == h->w && same_enc && side[sd].which == k) {
                        rw.slab[ns] = ws + pl.wpart[sd];
                        rw.cs_cw[ns] = pl.cs_cw[sd];
                        if (rb) rb->cs_dy[ns] = pl.cs_dy[sd];
                        if (rgn) rgn->cs_cw[ns] = pl.cs_cw[sd];
                        ++ns;
                    }
                }
                if (s->weight_norm) {
                    const float *n2p = n2base + n2_off[e][k];
                    rw.g = gbase + n2_off[e][k]; rw.n2 = n2p;
                    if (rgn) { rgn->g = rw.g; rgn->n2 = n2p; }
                }
            }
            const CflHead &m = pl.lay.enc[e].mono;
            if (m.w >= 0) {
                Region &rm = add(RK_MONO_W, m.w, (int64_t)s->L * m.npad, regon, m.npad, m.n);
                if (e == 0) rm.cs_dy[0] = 0;   // the gate head of the SRC encoder is the one used
                if (s->weight_norm) { rm.g = gbase + n2_off[e][2]; rm.n2 = n2base + n2_off[e][2]; }
                if (m.g >= 0) {
                    Region &rmg = add(RK_MONO_G, m.g, m.npad, 0, m.npad, m.n);
                    if (e == 0) rmg.cs_dy[0] = 0;
                    rmg.g = gbase + n2_off[e][2]; rmg.n2 = n2base + n2_off[e][2];
                }
            }
        }
        add(RK_THR, pl.lay.thr, 64, 0, 0, 1);
        fa.nregions = nr;
        // regulariser ranges for the mid kernel's extra blocks
        if (regon) {
            long long groups_total = 0;
            int k = 0;
            for (int i = 0; i < nr; ++i)
                if (fa.reg[i].reg) {
                    ma.reg_off[k] = fa.reg[i].off; ma.reg_cnt[k] = fa.reg[i].cnt;
                    groups_total += fa.reg[i].cnt >> 6; ++k;
                }
            ma.nreg_ranges = k;
            ma.reg_total_groups = groups_total;
            nreg_blocks = (int)((groups_total + 63) / 64);
            if (nreg_blocks > pl.nregblocks) return set_err(CFL_E_WORKSPACE, "regpart too small");
        }
    }
    {
        ProfScope ps(st, CFL_K_MID);
        const dim3 mgrid(ma.nrb + nreg_blocks), mblk(64);   // (register forms: no extra scoring rows, plan.x_ok)
        const dim3 rgrid(ma.nrb + ma.nxb + nreg_blocks);
        switch (pl.mid_kernel) {
            case MK_ROW1: hipLaunchKernelGGL((cfl_mid_row_kernel<1>), rgrid, dim3(256), 4 * 6 * 64 * sizeof(float), st, ma); break;
            case MK_ROW2: hipLaunchKernelGGL((cfl_mid_row_kernel<2>), rgrid, dim3(256), 4 * 6 * 128 * sizeof(float), st, ma); break;
            case MK_ROW4: hipLaunchKernelGGL((cfl_mid_row_kernel<4>), rgrid, dim3(256), 4 * 6 * 256 * sizeof(float), st, ma); break;
            case MK_REG_K4_LQ2: hipLaunchKernelGGL((cfl_mid_kernel<4, 2>), mgrid, mblk, pl.mid_lds, st, ma); break;
            case MK_REG_K8_LQ2: hipLaunchKernelGGL((cfl_mid_kernel<8, 2>), mgrid, mblk, pl.mid_lds, st, ma); break;
            case MK_REG_K4_LQ4: hipLaunchKernelGGL((cfl_mid_kernel<4, 4>), mgrid, mblk, pl.mid_lds, st, ma); break;
            default: hipLaunchKernelGGL((cfl_mid_kernel<0, 0>), mgrid, mblk, pl.mid_lds, st, ma); break;
        }
    }
    if (!train) {
        HIP_TRY(hipGetLastError());
        return CFL_OK;
    }

    // ---- grad (+ row reductions in z-slice 0) ---------------------------------
    {
        GradArgs ga;
        memset(&ga, 0, sizeof(ga));
        int nj = 0;
        for (int sd = 0; sd < 2; ++sd) {
            const CflHead *h = side[sd].head;
            const int tiles = h->npad / 16;
            for (int c0 = 0; c0 < tiles; c0 += 4) {
                GradJob &j = ga.job[nj++];
                j.side = sd;
                j.dyf = ws + pl.dyf[sd] + (size_t)c0 * RG * 256;
                j.wpart = ws + pl.wpart[sd] + (size_t)c0 * G * 256;
                j.pstride = (long long)h->npad * s->D;
                j.nt = tiles - c0 < 4 ? tiles - c0 : 4;
            }
        }
        ga.rows[0] = rsrc[0]; ga.rows[1] = rsrc[1];
        ga.B = (int)rows; ga.R = pl.R; ga.Rpad = pl.Rpad; ga.D = s->D; ga.P = pl.P; ga.norm = nd;
        int nr = 0, tot = 0;
        auto red = [&](int kind, const float *A, const float *B, int count, int out) -> RedRange & {
            RedRange &r = ga.red[nr++];
            memset(&r, 0, sizeof(r));
            r.A = A; r.B = B; r.kind = kind; r.count = count; r.out_off = out;
            tot += count;
            return r;
        };
        // siamese + fused tail: both sides feed ONE head, so side 1's ranges are dual (they also sum side 0's tile of
        // the same columns) and side 0 gets none
        const bool paired = pl.fused && s->dist_type == CFL_DIST_SIAMESE && !s->directed;
        int red_dy[2] = {-1, -1}, red_cw[2] = {-1, -1};   // range index of a side's dY / dy * xv column sums
        red(0, ws + pl.rowqf, nullptr, 1, pl.cs_rowq);
        for (int sd = paired ? 1 : 0; sd < 2; ++sd) {
            red_dy[sd] = nr;
            red(0, ws + pl.dyf[sd], nullptr, side[sd].head->npad / 16, pl.cs_dy[sd]);
            if (pl.has_cw) {
                red_cw[sd] = nr;
                red(0, ws + pl.cwf[sd], nullptr, side[sd].head->npad / 16, pl.cs_cw[sd]);
            }
        }
        if (pl.mono) {
            RedRange &r = red(1, ws + pl.mono_ya, ws + pl.mono_du, s->L, pl.cs_mono);
            r.lda = pl.lpad; r.ldb = pl.kpad; r.K = s->K; r.kpad = pl.kpad;
            if (s->weight_norm) {
                RedRange &r2 = red(2, ws + pl.mono_duc, nullptr, 1, pl.cs_duc);
                r2.lda = pl.kpad; r2.K = s->K; r2.kpad = pl.kpad;
            }
        }
        ga.nred = nr; ga.red_total = tot; ga.colsum = ws + pl.colsum;
        if (pl.fused) {
            GradFuse &f = ga.fuse;
            f.on = 1;
            f.ticket = (int *)(ws + pl.handoff);
            f.flag = f.ticket + pl.nhandoff;
            f.theta = theta; f.grad = grad;
            if (adam) {
                f.theta_out = adam->theta; f.m = adam->m; f.v = adam->v;
                f.lr_t = adam->lr_t; f.b1 = adam->b1; f.b2 = adam->b2; f.eps = adam->eps;
            }
            f.in_mul = in_mul; f.reg_const = loss->reg_const;
            f.spin_limit = debug_env("CFL_DEBUG_SPIN_LIMIT") != 0 ? debug_env("CFL_DEBUG_SPIN_LIMIT") : CFL_HANDOFF_SPIN_LIMIT;
            // kept planes: written by the tile finishers only when the next step's projection will read them
            f.planes = (keeping && adam && (pl.proj_x3 || pl.proj_bx3)) ? (unsigned short *)kept->buf : nullptr;
            int jn = 0;
            for (int sd = 0; sd < 2; ++sd) {
                const CflHead *h = side[sd].head;
                for (int c0 = 0; c0 < h->npad / 16; c0 += 4) f.w_off[jn++] = h->w + (long long)c0 * G * 256;
            }
            for (int k = 0; k < CFL_MAX_RED; ++k) { f.red_b[k] = -1; f.red_g[k] = -1; }
            // red ranges: 0 = row quantities, then per side: dY (-> bias) and, with weight norm, dy * xv (-> gain, c_j)
            f.wn = s->weight_norm ? 1 : 0;
            f.red_done = f.flag + pl.nhandoff;
            f.red_expect = 0;
            jn = 0;
            if (paired) {
                f.pair_jobs = (side[0].head->npad / 16 + 3) / 4;
                f.pair_delta = (long long)pl.wpart[1] - (long long)pl.wpart[0];
            }
            for (int sd = 0; sd < 2; ++sd) {
                const CflHead *h = side[sd].head;
                const int kd = red_dy[sd], kc = red_cw[sd];
                if (kd >= 0) {
                    f.red_b[kd] = h->b;
                    f.red_n[kd] = h->n;
                    f.red_npad[kd] = h->npad;
                    if (paired) ga.red[kd].B = ws + pl.dyf[0];
                }
                const float *n2p = f.wn ? n2base + n2_off[side[sd].enc][side[sd].which] : nullptr;
                const float *gp = f.wn ? gbase + n2_off[side[sd].enc][side[sd].which] : nullptr;
                if (f.wn && kc >= 0) {
                    f.red_g[kc] = h->g;
                    f.red_n[kc] = h->n;
                    f.red_npad[kc] = h->npad;
                    f.red_n2[kc] = n2p;
                    f.red_expect += h->npad / 16;
                    if (paired) ga.red[kc].B = ws + pl.cwf[0];
                }
                for (int c0 = 0; c0 < h->npad / 16; c0 += 4, ++jn) {
                    if (f.wn) {
                        f.wn_g[jn] = gp + c0 * 16;
                        f.wn_n2[jn] = n2p + c0 * 16;
                        f.wn_cw[jn] = ws + pl.colsum + pl.cs_cw[sd] + c0 * 16;
                        f.wn_n[jn] = h->n - c0 * 16;
                    }
                }
            }
            f.thr_off = pl.lay.thr;
            // monomer: the gate head of the source encoder is finished by its own reduction blocks
            f.mono_w = f.mono_g = -1;
            if (pl.mono) {
                const CflHead &mh = pl.lay.enc[0].mono;
                f.mono_w = mh.w; f.mono_g = mh.g;
                f.mono_L = s->L; f.mono_K = s->K; f.mono_kpad = pl.kpad; f.mono_reg = loss->reg_const > 0.f;
                if (s->weight_norm) {
                    f.mono_n2 = n2base + n2_off[0][2];
                    f.mono_gcopy = gbase + n2_off[0][2];
                    f.mono_duc = ws + pl.mono_duc;
                }
            }
            // directed encoders: the heads no side projects through (and the target encoder's gate head)
            f.norph = 0;
            long long orph_total = 0;
            if (s->directed) {
                auto orphan = [&](long long off, long long cnt, int reg) {
                    if (off < 0 || cnt <= 0 || f.norph >= 8) return;
                    f.orph_off[f.norph] = off; f.orph_cnt[f.norph] = round_up(cnt, 64); f.orph_reg[f.norph] = reg;
                    orph_total += round_up(cnt, 64);
                    ++f.norph;
                };
                const int regon2 = loss->reg_const > 0.f;
                for (int e = 0; e < 2; ++e) {
                    const CflHead *hh[2] = {&pl.lay.enc[e].outputs, &pl.lay.enc[e].proto};
                    for (int k = 0; k < 2; ++k) {
                        const CflHead *h = hh[k];
                        if (h->w < 0) continue;
                        bool used = false;
                        for (int sd = 0; sd < 2; ++sd) used |= side[sd].enc == e && side[sd].which == k;
                        if (used) continue;
                        // W, bias and gain arrays are contiguous in theta (cfl_layout): one region each
                        orphan(h->w, (long long)h->npad * s->D, regon2);
                        orphan(h->b, h->b >= 0 ? round_up(h->npad, 64) : 0, regon2);
                        orphan(h->g, h->g >= 0 ? round_up(h->npad, 64) : 0, 0);
                    }
                }
                if (pl.mono) {
                    const CflHead &m1 = pl.lay.enc[1].mono;
                    orphan(m1.w, round_up((long long)s->L * m1.npad, 64), regon2);
                    orphan(m1.g, m1.g >= 0 ? round_up(m1.npad, 64) : 0, 0);
                }
                if (orph_total > 0) {
                    RedRange &ro = red(3, nullptr, nullptr, (int)((orph_total + 1023) / 1024), 0);
                    (void)ro;
                    ga.nred = nr; ga.red_total = tot;
                }
            }
            f.regpart = ws + pl.regpart; f.nregblocks = nreg_blocks;
            f.B = (int)rows; f.use_threshold = loss->use_threshold;
            f.pos_weight = loss->pos_weight; f.caffe_margin = loss->caffe_margin; f.lambda_m = loss->lambda_m;
            f.scalars = scalars; f.thr_copy = ws + pl.thr_copy;
            f.scalars2 = xs ? xs->scalars_copy : nullptr;
        }
        ga.tps = pl.xcd ? (s->D / (pl.grad_half ? 32 : 64)) / pl.S : 0;
        dim3 grid(s->D / 64, pl.P, nj + 1);
        ProfScope ps(st, CFL_K_GRAD);
        const size_t glds = 4 * 4 * 4 * 64 * sizeof(f32x4);
        const dim3 hgrid(s->D / 32, pl.P, nj + 1);
        switch (pl.grad_kernel) {
            case GK_HALF_W8: hipLaunchKernelGGL(cfl_grad_x3_half_w8_kernel, hgrid, dim3(512), glds, st, ga); break;
            case GK_HALF: hipLaunchKernelGGL(cfl_grad_x3_half_kernel, hgrid, dim3(256), glds, st, ga); break;
            case GK_HALF_SPLIT: hipLaunchKernelGGL(cfl_grad_x3_half_split_kernel, hgrid, dim3(256), glds, st, ga); break;
            case GK_X3: hipLaunchKernelGGL(cfl_grad_x3_kernel, grid, dim3(256), glds, st, ga); break;
            case GK_X3_LONGRANGE: hipLaunchKernelGGL(cfl_grad_x3_longrange_kernel, grid, dim3(256), glds, st, ga); break;
            default: hipLaunchKernelGGL(cfl_grad_kernel, grid, dim3(256), glds, st, ga); break;
        }
    }

    if (kept && adam) kept->valid = (keeping && pl.fused && (pl.proj_x3 || pl.proj_bx3)) ? 1 : 0;   // theta has changed; were the planes written?
    else if (keeping && (pl.proj_x3 || pl.proj_bx3)) kept->valid = 1;   // theta unchanged: the buffer holds its planes now (split above if it was stale)
    if (pl.fused) {
        HIP_TRY(hipGetLastError());
        return CFL_OK;
    }
    // ---- finalize -----------------------------------------------------------
    fa.total = pl.lay.total; fa.theta = theta; fa.grad = grad;
    fa.colsum = ws + pl.colsum; fa.cs_rowq = pl.cs_rowq; fa.cs_mono = pl.cs_mono; fa.cs_duc = pl.cs_duc;
    fa.P = pl.P; fa.D = s->D;
    fa.L = s->L; fa.kpad = pl.kpad > 0 ? pl.kpad : 1; fa.weight_norm = s->weight_norm;
    fa.in_mul = in_mul;
    fa.reg_const = loss->reg_const; fa.use_threshold = loss->use_threshold;
    fa.pos_weight = loss->pos_weight; fa.caffe_margin = loss->caffe_margin; fa.lambda_m = loss->lambda_m;
    fa.B = (int)rows; fa.regpart = ws + pl.regpart; fa.nregblocks = nreg_blocks;
    fa.scalars = scalars;
    fa.nblocks_main = (int)((pl.lay.total / 4 + 255) / 256);
    fa.thr_copy = ws + pl.thr_copy;
    if (adam) {
        fa.adam_m = adam->m; fa.adam_v = adam->v; fa.theta_out = adam->theta;
        fa.lr_t = adam->lr_t; fa.b1 = adam->b1; fa.b2 = adam->b2; fa.eps = adam->eps;
    }
    for (int i = 0; i < fa.nregions; ++i) {
        fa.rbeg[i] = (int)fa.reg[i].off;
        fa.rend[i] = (int)(fa.reg[i].off + fa.reg[i].cnt);
    }
    {
        ProfScope ps(st, CFL_K_FINALIZE);
        hipLaunchKernelGGL(cfl_finalize_kernel, dim3(fa.nblocks_main + 1), dim3(256), 0, st, fa);
    }
    HIP_TRY(hipGetLastError());
    return CFL_OK;
}

extern "C" int cfl_pair_scores(const CflShape *shape, const CflNorm *norm, const float *xs,
                               const float *xt, int64_t n, const float *theta, float *scores,
                               float *dists, void *workspace, size_t workspace_bytes,
                               cfl_stream_t stream) {
    if (!scores) return set_err(CFL_E_SHAPE, "scores is NULL");
    const float *x[2] = {xs, xt};
    return run_pairs(shape, norm, nullptr, x, 1, n, theta, nullptr, nullptr, scores, dists,
                     workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int cfl_pair_step_fwd_bwd(const CflShape *shape, const CflNorm *norm,
                                     const CflLossCfg *loss, const float *const x4[4], int64_t B,
                                     const float *theta, float *grad, float *scalars,
                                     void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    if (!loss || !x4 || !grad || !scalars) return set_err(CFL_E_SHAPE, "NULL loss/x4/grad/scalars");
    if (loss->caffe_margin != 0.f && loss->lambda_m != 0.f)
        return set_err(CFL_E_SHAPE, "caffe_margin and lambda_m are exclusive (cfl/utils.py:72-73)");
    return run_pairs(shape, norm, loss, x4, 2, B, theta, grad, scalars, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

extern "C" int cfl_pair_step_fwd_bwd_planes(const CflShape *shape, const CflNorm *norm,
                                            const CflLossCfg *loss, const float *const x4[4], int64_t B,
                                            const float *theta, float *grad, float *scalars, CflThetaPlanes *planes,
                                            void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    if (!loss || !x4 || !grad || !scalars) return set_err(CFL_E_SHAPE, "NULL loss/x4/grad/scalars");
    if (loss->caffe_margin != 0.f && loss->lambda_m != 0.f)
        return set_err(CFL_E_SHAPE, "caffe_margin and lambda_m are exclusive (cfl/utils.py:72-73)");
    return run_pairs(shape, norm, loss, x4, 2, B, theta, grad, scalars, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, nullptr, planes);
}

extern "C" int cfl_pair_train_step_planes(const CflShape *shape, const CflNorm *norm,
                                          const CflLossCfg *loss, const float *const x4[4], int64_t B,
                                          float *theta, float *m, float *v, float *grad, float *scalars,
                                          float lr_t, float beta1, float beta2, float eps, CflThetaPlanes *planes,
                                          void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    if (!loss || !x4 || !grad || !scalars || !m || !v) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (loss->caffe_margin != 0.f && loss->lambda_m != 0.f)
        return set_err(CFL_E_SHAPE, "caffe_margin and lambda_m are exclusive (cfl/utils.py:72-73)");
    AdamFuse af = {theta, m, v, lr_t, beta1, beta2, eps};
    return run_pairs(shape, norm, loss, x4, 2, B, theta, grad, scalars, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream, &af, nullptr, planes);
}

extern "C" int cfl_pair_train_step(const CflShape *shape, const CflNorm *norm,
                                   const CflLossCfg *loss, const float *const x4[4], int64_t B,
                                   float *theta, float *m, float *v, float *grad, float *scalars,
                                   float lr_t, float beta1, float beta2, float eps,
                                   void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    return cfl_pair_train_step_planes(shape, norm, loss, x4, B, theta, m, v, grad, scalars, lr_t, beta1, beta2, eps,
                                      nullptr, workspace, workspace_bytes, stream);
}

static int check_train_args(const CflLossCfg *loss, const void *grad, const void *scalars) {
    if (!loss || !grad || !scalars) return set_err(CFL_E_SHAPE, "NULL loss/grad/scalars");
    if (loss->caffe_margin != 0.f && loss->lambda_m != 0.f)
        return set_err(CFL_E_SHAPE, "caffe_margin and lambda_m are exclusive (cfl/utils.py:72-73)");
    return CFL_OK;
}

extern "C" int cfl_pair_scores_idx(const CflShape *shape, const CflNorm *norm, const float *table,
                                   int64_t table_rows, const int32_t *const idx2[2], int64_t idx_stride, int64_t n,
                                   const float *theta, float *scores, float *dists, void *workspace,
                                   size_t workspace_bytes, cfl_stream_t stream) {
    if (!scores) return set_err(CFL_E_SHAPE, "scores is NULL");
    IndexSrc is = {table, table_rows, idx2, idx_stride};
    return run_pairs(shape, norm, nullptr, nullptr, 1, n, theta, nullptr, nullptr, scores, dists, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, &is);
}

extern "C" int cfl_pair_scores_idx4(const CflShape *shape, const CflNorm *norm, const float *table,
                                    int64_t table_rows, const int32_t *const idx4[4], int64_t idx_stride, int64_t n,
                                    const float *theta, float *scores, float *dists, void *workspace,
                                    size_t workspace_bytes, cfl_stream_t stream) {
    if (!scores) return set_err(CFL_E_SHAPE, "scores is NULL");
    IndexSrc is = {table, table_rows, idx4, idx_stride};
    return run_pairs(shape, norm, nullptr, nullptr, 2, n, theta, nullptr, nullptr, scores, dists, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, &is);
}

extern "C" int cfl_pair_step_fwd_bwd_idx(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                         const float *table, int64_t table_rows, const int32_t *const idx4[4],
                                         int64_t idx_stride, int64_t B, const float *theta, float *grad,
                                         float *scalars, void *workspace, size_t workspace_bytes,
                                         cfl_stream_t stream) {
    int rc = check_train_args(loss, grad, scalars);
    if (rc) return rc;
    IndexSrc is = {table, table_rows, idx4, idx_stride};
    return run_pairs(shape, norm, loss, nullptr, 2, B, theta, grad, scalars, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, &is);
}

extern "C" int cfl_pair_step_fwd_bwd_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                                const float *table, int64_t table_rows, const int32_t *const idx4[4],
                                                int64_t idx_stride, int64_t B, const float *theta, float *grad,
                                                float *scalars, CflThetaPlanes *planes, void *workspace,
                                                size_t workspace_bytes, cfl_stream_t stream) {
    int rc = check_train_args(loss, grad, scalars);
    if (rc) return rc;
    IndexSrc is = {table, table_rows, idx4, idx_stride};
    return run_pairs(shape, norm, loss, nullptr, 2, B, theta, grad, scalars, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, &is, planes);
}

extern "C" int cfl_pair_train_step_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                              const float *table, int64_t table_rows, const int32_t *const idx4[4],
                                              int64_t idx_stride, int64_t B, float *theta, float *m, float *v,
                                              float *grad, float *scalars, float lr_t, float beta1, float beta2,
                                              float eps, CflThetaPlanes *planes, void *workspace, size_t workspace_bytes,
                                              cfl_stream_t stream) {
    int rc = check_train_args(loss, grad, scalars);
    if (rc) return rc;
    if (!m || !v) return set_err(CFL_E_SHAPE, "NULL Adam slots");
    AdamFuse af = {theta, m, v, lr_t, beta1, beta2, eps};
    IndexSrc is = {table, table_rows, idx4, idx_stride};
    return run_pairs(shape, norm, loss, nullptr, 2, B, theta, grad, scalars, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream, &af, &is, planes);
}

extern "C" int cfl_pair_train_step_idx(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                       const float *table, int64_t table_rows, const int32_t *const idx4[4],
                                       int64_t idx_stride, int64_t B, float *theta, float *m, float *v,
                                       float *grad, float *scalars, float lr_t, float beta1, float beta2,
                                       float eps, void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    return cfl_pair_train_step_idx_planes(shape, norm, loss, table, table_rows, idx4, idx_stride, B, theta, m, v, grad,
                                          scalars, lr_t, beta1, beta2, eps, nullptr, workspace, workspace_bytes, stream);
}

// A train of consecutive training steps on windows of the (shuffled) pair lists: step i trains rows
// [head + i*B + lo, head + i*B + lo + rows) of both lists.  This is the inner loop of cfl/bin/train_dist.py:77-87
// (`for i in t: sess.run([s_optim, ...])`) between two read-backs of the display scalars: the host enqueues the
// launches of all steps back to back, so the loop is bound by the GPU, not by per-step interpreter work.  TF-Adam's
// float32 power accumulators (beta1_power / beta2_power) are advanced here; `scalars` / `grad` hold
// the values of the LAST step.  switched[i] != 0 swaps source and target of step i (data_switch,
// cfl/input_data.py:575-577); NULL = never.
extern "C" int cfl_pair_train_steps_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                               const float *table, int64_t table_rows, const int32_t *pos_pairs, int64_t n_pos,
                                               const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
                                               int64_t batch_rows, int64_t shard_lo, int64_t rows,
                                               const uint8_t *switched, int64_t nsteps, float *theta, float *m, float *v,
                                               float *grad, float *scalars, float lr, float beta1, float beta2, float eps,
                                               float *beta1_power, float *beta2_power, CflThetaPlanes *planes,
                                               void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    int rc = check_train_args(loss, grad, scalars);
    if (rc) return rc;
    if (!m || !v || !pos_pairs || !neg_pairs || !beta1_power || !beta2_power)
        return set_err(CFL_E_SHAPE, "NULL pointer");
    if (nsteps <= 0 || batch_rows <= 0 || rows <= 0 || shard_lo < 0 || shard_lo + rows > batch_rows ||
        pos_head < 0 || neg_head < 0)
        return set_err(CFL_E_SHAPE, "bad step window");
    if (nsteps > (1ll << 40) / batch_rows || pos_head + nsteps * batch_rows > n_pos || neg_head + nsteps * batch_rows > n_neg)
        return set_err(CFL_E_SHAPE, "step window runs past the pair lists: heads %lld / %lld + %lld steps x %lld rows, lists %lld / %lld",
                       (long long)pos_head, (long long)neg_head, (long long)nsteps, (long long)batch_rows,
                       (long long)n_pos, (long long)n_neg);
    float b1p = *beta1_power, b2p = *beta2_power;
    for (int64_t i = 0; i < nsteps; ++i) {
        const int32_t *ps = pos_pairs + 2 * (pos_head + i * batch_rows + shard_lo);
        const int32_t *ng = neg_pairs + 2 * (neg_head + i * batch_rows + shard_lo);
        const int c0 = (switched && switched[i]) ? 1 : 0;
        const int32_t *idx4[4] = {ps + c0, ps + (1 - c0), ng + c0, ng + (1 - c0)};
        const float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
        AdamFuse af = {theta, m, v, lr_t, beta1, beta2, eps};
        IndexSrc is = {table, table_rows, idx4, 2};
        rc = run_pairs(shape, norm, loss, nullptr, 2, rows, theta, grad, scalars, nullptr, nullptr, workspace,
                       workspace_bytes, (hipStream_t)stream, &af, &is, planes);
        if (rc) return rc;
        b1p *= beta1;
        b2p *= beta2;
    }
    *beta1_power = b1p;
    *beta2_power = b2p;
    return CFL_OK;
}

// ... with the reference's validation fetch inside the steps (include/cfl_hip.h): step i with val_mask[i] != 0 also scores the next
// validation batch (window `val_head + k * val_batch_rows` of the validation pair lists, k = validation steps so far) as extra
// scoring rows of its own projection / row-math launches and leaves [scalars | scores] in ring_slots[k].
extern "C" int cfl_train_val_fusable(const CflShape *shape, int64_t rows, int64_t val_rows) {
    Plan pl;
    if (make_plan(shape, rows, 2, true, true, &pl, 2 * val_rows)) return 0;
    return (pl.x_ok && pl.fused) ? 1 : 0;
}

extern "C" int cfl_pair_train_val_steps_idx_planes(
    const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *table, int64_t table_rows,
    const int32_t *pos_pairs, int64_t n_pos, const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
    int64_t batch_rows, int64_t shard_lo, int64_t rows, const uint8_t *switched, int64_t nsteps,
    const float *val_table, int64_t val_table_rows, const int32_t *val_pos_pairs, int64_t n_val_pos,
    const int32_t *val_neg_pairs, int64_t n_val_neg, int64_t val_pos_head, int64_t val_neg_head, int64_t val_batch_rows,
    const uint8_t *val_switched, const uint8_t *val_mask, float *const *ring_slots,
    float *theta, float *m, float *v, float *grad, float *scalars, float lr, float beta1, float beta2, float eps,
    float *beta1_power, float *beta2_power, CflThetaPlanes *planes, void *workspace, size_t workspace_bytes,
    cfl_stream_t stream) {
    int rc = check_train_args(loss, grad, scalars);
    if (rc) return rc;
    if (!m || !v || !pos_pairs || !neg_pairs || !beta1_power || !beta2_power || !val_mask || !ring_slots || !val_pos_pairs ||
        !val_neg_pairs)
        return set_err(CFL_E_SHAPE, "NULL pointer");
    if (nsteps <= 0 || batch_rows <= 0 || rows <= 0 || shard_lo < 0 || shard_lo + rows > batch_rows || pos_head < 0 ||
        neg_head < 0 || val_batch_rows <= 0 || val_pos_head < 0 || val_neg_head < 0)
        return set_err(CFL_E_SHAPE, "bad step window");
    int64_t nval = 0;
    for (int64_t i = 0; i < nsteps; ++i) nval += val_mask[i] ? 1 : 0;
    if (nsteps > (1ll << 40) / batch_rows || pos_head + nsteps * batch_rows > n_pos || neg_head + nsteps * batch_rows > n_neg ||
        val_pos_head + nval * val_batch_rows > n_val_pos || val_neg_head + nval * val_batch_rows > n_val_neg)
        return set_err(CFL_E_SHAPE, "step window runs past the pair lists");
    for (int64_t k = 0; k < nval; ++k)
        if (!ring_slots[k]) return set_err(CFL_E_SHAPE, "ring slot %lld is NULL", (long long)k);
    float b1p = *beta1_power, b2p = *beta2_power;
    int64_t k = 0;
    for (int64_t i = 0; i < nsteps; ++i) {
        const int32_t *ps = pos_pairs + 2 * (pos_head + i * batch_rows + shard_lo);
        const int32_t *ng = neg_pairs + 2 * (neg_head + i * batch_rows + shard_lo);
        const int c0 = (switched && switched[i]) ? 1 : 0;
        const int32_t *idx4[4] = {ps + c0, ps + (1 - c0), ng + c0, ng + (1 - c0)};
        const float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
        AdamFuse af = {theta, m, v, lr_t, beta1, beta2, eps};
        IndexSrc is = {table, table_rows, idx4, 2};
        ExtraSrc xs;
        if (val_mask[i]) {
            const int32_t *vp = val_pos_pairs + 2 * (val_pos_head + k * val_batch_rows);
            const int32_t *vn = val_neg_pairs + 2 * (val_neg_head + k * val_batch_rows);
            const int v0 = (val_switched && val_switched[k]) ? 1 : 0;
            xs = {val_table, val_table_rows, {vp + v0, vp + (1 - v0), vn + v0, vn + (1 - v0)}, 2, val_batch_rows,
                  ring_slots[k] + CFL_S_COUNT, ring_slots[k]};
            ++k;
        }
        rc = run_pairs(shape, norm, loss, nullptr, 2, rows, theta, grad, scalars, nullptr, nullptr, workspace,
                       workspace_bytes, (hipStream_t)stream, &af, &is, planes, val_mask[i] ? &xs : nullptr);
        if (rc) return rc;
        b1p *= beta1;
        b2p *= beta2;
    }
    *beta1_power = b1p;
    *beta2_power = b2p;
    return CFL_OK;
}

extern "C" int cfl_pair_train_steps_idx(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                        const float *table, int64_t table_rows, const int32_t *pos_pairs, int64_t n_pos,
                                        const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
                                        int64_t batch_rows, int64_t shard_lo, int64_t rows,
                                        const uint8_t *switched, int64_t nsteps, float *theta, float *m, float *v,
                                        float *grad, float *scalars, float lr, float beta1, float beta2, float eps,
                                        float *beta1_power, float *beta2_power, void *workspace,
                                        size_t workspace_bytes, cfl_stream_t stream) {
    return cfl_pair_train_steps_idx_planes(shape, norm, loss, table, table_rows, pos_pairs, n_pos, neg_pairs, n_neg, pos_head,
                                           neg_head, batch_rows, shard_lo, rows, switched, nsteps, theta, m, v, grad, scalars,
                                           lr, beta1, beta2, eps, beta1_power, beta2_power, nullptr, workspace,
                                           workspace_bytes, stream);
}

// ---- input gradient of the two heads (needed when the pair rows are not leaves: ConvPCD) --
struct DyfScaled {   // A(m = row, k = col) = dy[row][col] * sc[col]   (dYf is fragment-major)
    const float *dyf, *g, *n2; int RG, n; float in_mul;
    __device__ float operator()(int m, int k) const {
        if (k >= n) return 0.f;
        const size_t o = ((size_t)(k >> 4) * RG + (m >> 4)) * 256 + (((m >> 2) & 3) * 16 + (k & 15)) * 4 + (m & 3);
        float sc = in_mul;
        if (g) sc *= g[k] * rsqrtf(n2[k]);
        return dyf[o] * sc;
    }
};
struct WfKN {        // B(k = col, n = d) = W[d][col]   (Wf is fragment-major)
    const float *wf; int G;
    __device__ float operator()(int k, int n) const {
        return wf[((size_t)(k >> 4) * G + (n >> 4)) * 256 + (((n >> 2) & 3) * 16 + (k & 15)) * 4 + (n & 3)];
    }
};
struct StoreRowMajor {
    float *out; int ld;
    __device__ void operator()(int m, int n, float v, int) const { out[(size_t)m * ld + n] = v; }
};

extern "C" int cfl_pair_input_grad(const CflShape *s, const CflNorm *norm, int64_t B, const float *theta,
                                   const void *workspace, size_t workspace_bytes, float *dx_src,
                                   float *dx_dst, cfl_stream_t stream) {
    Plan pl;
    int rc = make_plan(s, B, 2, true, false, &pl);
    if (rc) return rc;
    if (!theta || !workspace || !dx_src || !dx_dst) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (workspace_bytes < pl.total_floats * sizeof(float)) return set_err(CFL_E_WORKSPACE, "workspace too small");
    float in_mul;
    NormDev nd = make_norm(norm, &in_mul);
    if (nd.elementwise) return set_err(CFL_E_UNSUPPORTED, "input gradient with an element-wise normaliser");
    const float *ws = (const float *)workspace;
    const CflHead *hs, *hd;
    side_heads(s, pl.lay, &hs, &hd);
    // gain snapshot / squared norms were left in the workspace by the step (same order as
    // launch_colnorm: encoder 0 outputs, proto, mono, then encoder 1)
    int n2_off[2][3], off = 0;
    const int nenc = s->directed ? 2 : 1;
    for (int e = 0; e < 2; ++e) {
        const CflHead *hh[3] = {&pl.lay.enc[e].outputs, &pl.lay.enc[e].proto, &pl.lay.enc[e].mono};
        for (int k = 0; k < 3; ++k) {
            if (e >= nenc) { n2_off[e][k] = n2_off[0][k]; continue; }
            n2_off[e][k] = -1;
            if (hh[k]->w < 0) continue;
            n2_off[e][k] = off;
            off += hh[k]->npad;
        }
    }
    const int which[2] = {s->dist_type == CFL_DIST_PCD ? 1 : 0, s->dist_type == CFL_DIST_MONOMER ? 1 : 0};
    const CflHead *heads[2] = {hs, hd};
    float *outs[2] = {dx_src, dx_dst};
    const int G = s->D / 16, RG = pl.Rpad / 16;
    for (int sd = 0; sd < 2; ++sd) {
        const float *gp = nullptr, *n2p = nullptr;
        if (s->weight_norm) {
            const int o = n2_off[sd == 0 ? 0 : 1][which[sd]];
            n2p = ws + pl.n2 + o;
            gp = ws + pl.n2 + 6 * 1024 + o;
        }
        gemm_gather(pl.R, s->D, heads[sd]->npad, gg_klen(heads[sd]->npad, 1),
                    DyfScaled{ws + pl.dyf[sd], gp, n2p, RG, heads[sd]->n, in_mul},
                    WfKN{theta + heads[sd]->w, G}, StoreRowMajor{outs[sd], s->D}, (hipStream_t)stream);
    }
    HIP_TRY(hipGetLastError());
    return CFL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Host-only integer work: the per-epoch reshuffle of a pair list, `pairs[rng.permutation(n)]`
// (cfl/input_data.py:543-551) in the legacy numpy.random.RandomState stream, bit for bit.
//   MT19937 (Matsumoto & Nishimura) exactly as numpy's legacy bit generator drives it: 32-bit tempered outputs,
//   `permutation(n)` = Fisher-Yates from the top (for i = n-1 .. 1: j = interval(i); swap(i, j)) over arange(n),
//   interval(max) = masked rejection on 32-bit draws (64-bit draws above 2^32 - 1).
// It runs without the interpreter lock (ctypes releases it), so a worker thread can prepare the next epoch's
// order while the main thread keeps the GPU queue full; pinned to numpy by tests/test_input_data.py.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct Mt19937 {
    uint32_t *key;
    int pos;
    uint32_t out[624];   // tempered outputs of the current block (filled by temper_block)
    bool have_out = false;
    void temper_block() {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[i] = y;
        }
        have_out = true;
    }
    void regenerate() {
        const int N = 624, M = 397;
        const uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
        int i;
        uint32_t y;
        for (i = 0; i < N - M; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        for (; i < N - 1; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        y = (key[N - 1] & UPPER) | (key[0] & LOWER);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        pos = 0;
        have_out = false;
    }
    // tempering is done a block at a time (a plain vectorisable loop), so that the rejection loop of interval()
    // is a load, a mask and a compare per draw
    uint32_t next32() {
        if (pos == 624) regenerate();
        if (!have_out) temper_block();
        return out[pos++];
    }
    uint64_t next64() { const uint64_t hi = next32(); return (hi << 32) | next32(); }
    uint64_t interval(uint64_t max) {
        if (max == 0) return 0;
        uint64_t mask = max, value;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffull) {
            while ((value = (next32() & mask)) > max) {}
        } else {
            while ((value = (next64() & mask)) > max) {}
        }
        return value;
    }
};
}  // namespace

extern "C" int cfl_mt19937_reshuffle(uint32_t *key, int32_t *pos, int64_t n, const int64_t *rows_in,
                                     int64_t cols, int64_t *rows_out, int64_t *perm_out, int32_t *rows_out32) {
    if (!key || !pos || n < 0 || *pos < 0 || *pos > 624) return set_err(CFL_E_SHAPE, "bad MT19937 state");
    if ((rows_in != nullptr) != (rows_out != nullptr) || (rows_in && cols <= 0))
        return set_err(CFL_E_SHAPE, "rows_in / rows_out / cols");
    Mt19937 mt;
    mt.key = key;
    mt.pos = (int)*pos;
    // Phase 1: the swap partners j_i, i = n-1 .. 1 -- the only part that is serial in the generator.
    // Phase 2: the swaps, with the partner's line requested a few iterations ahead (the permutation of a long list
    // does not fit the core's L2, and a swap chain that waits for every miss is what made this ~30 ns per element).
    std::vector<uint32_t> js32;
    std::vector<int64_t> js64, tmp;
    const bool small = n <= 0x7fffffffll;
    if (small) js32.resize((size_t)(n > 0 ? n : 1)); else js64.resize((size_t)n);
    for (int64_t i = n - 1; i >= 1; --i) {
        const uint64_t j = mt.interval((uint64_t)i);
        if (small) js32[(size_t)i] = (uint32_t)j; else js64[(size_t)i] = (int64_t)j;
    }
    *pos = mt.pos;
    int64_t *perm = perm_out;
    if (!perm) { tmp.resize((size_t)(n > 0 ? n : 1)); perm = tmp.data(); }
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    const int64_t AHEAD = 16;
    for (int64_t i = n - 1; i >= 1; --i) {
        if (i > AHEAD) __builtin_prefetch(perm + (small ? (int64_t)js32[(size_t)(i - AHEAD)] : js64[(size_t)(i - AHEAD)]), 1);
        const int64_t j = small ? (int64_t)js32[(size_t)i] : js64[(size_t)i];
        const int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    if (rows_in) {
        for (int64_t i = 0; i < n; ++i) {
            if (i + AHEAD < n) __builtin_prefetch(rows_in + perm[i + AHEAD] * cols, 0);
            const int64_t *src = rows_in + perm[i] * cols;
            int64_t *dst = rows_out + i * cols;
            for (int64_t c = 0; c < cols; ++c) dst[c] = src[c];
            if (rows_out32)
                for (int64_t c = 0; c < cols; ++c) rows_out32[i * cols + c] = (int32_t)src[c];
        }
    }
    return CFL_OK;
}

// HOST-ONLY: CRC-32C (Castagnoli) of n bytes, continuing from `crc` (0 to start): the checksum of TensorFlow's tensor-bundle
// checkpoint files (cfl/tf_bundle.py writes / verifies them; cfl/utils.py:465-497 of the reference reads them with tf.train.Saver)
extern "C" uint32_t cfl_crc32c(const void *data, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
            table[i] = c;
        }
    });
    const unsigned char *p = (const unsigned char *)data;
    uint32_t c = ~crc;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}

extern "C" int cfl_adam_tf(float *theta, float *m, float *v, const float *grad, int64_t n,
                           float lr_t, float beta1, float beta2, float eps, float grad_scale,
                           cfl_stream_t stream) {
    if (!theta || !m || !v || !grad) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (n <= 0 || n % 4) return set_err(CFL_E_SHAPE, "n=%lld must be a positive multiple of 4", (long long)n);
    const long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    {
        ProfScope ps((hipStream_t)stream, CFL_K_ADAM);
        hipLaunchKernelGGL(cfl_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, m,
                           v, grad, n4, lr_t, beta1, beta2, eps, grad_scale);
    }
    HIP_TRY(hipGetLastError());
    return CFL_OK;
}

extern "C" int cfl_adam_tf_planes(const CflShape *shape, float *theta, float *m, float *v, const float *grad,
                                  float lr_t, float beta1, float beta2, float eps, float grad_scale,
                                  CflThetaPlanes *planes, cfl_stream_t stream) {
    if (!theta || !m || !v || !grad) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (planes && (!planes->buf || ((uintptr_t)planes->buf & 15))) return set_err(CFL_E_SHAPE, "theta planes buffer NULL or misaligned");
    ThetaPlaneRegions pr;
    int rc = theta_plane_regions(shape, planes ? planes->buf : nullptr, &pr);
    if (rc) return rc;
    CflLayout lay;
    rc = cfl_layout(shape, &lay);
    if (rc) return rc;
    const long long n4 = lay.total / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    {
        ProfScope ps((hipStream_t)stream, CFL_K_ADAM);
        hipLaunchKernelGGL(cfl_adam_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, m,
                           v, grad, n4, lr_t, beta1, beta2, eps, grad_scale, pr);
    }
    HIP_TRY(hipGetLastError());
    if (planes) planes->valid = 1;
    return CFL_OK;
}

extern "C" int cfl_gather_rows(const float *table, const int64_t *idx, int64_t n, int64_t D,
                               float *out, cfl_stream_t stream) {
    if (!table || !idx || !out) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (n <= 0 || D <= 0 || D % 4) return set_err(CFL_E_SHAPE, "n=%lld D=%lld", (long long)n, (long long)D);
    int blocks = (int)((n + 3) / 4);
    if (blocks > 4096) blocks = 4096;
    {
        ProfScope ps((hipStream_t)stream, CFL_K_GATHER);
        hipLaunchKernelGGL(cfl_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table,
                           (const long long *)idx, (long long)n, (long long)D, out);
    }
    HIP_TRY(hipGetLastError());
    return CFL_OK;
}
