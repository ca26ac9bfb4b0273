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
#define CFL_HANDOFF_TIMEOUT_S 60.0   // wall-clock bound of an in-launch hand-off (CFL_HANDOFF_TIMEOUT_S overrides; CFL_DEBUG_SPIN_LIMIT < 0: give up at once)
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

// (the kernels of cfl_dp.hip are timed through these: one record list for the whole library)
void *cfl_prof_scope_begin(void *stream, int kind) { return g_prof_on ? new ProfScope((hipStream_t)stream, kind) : nullptr; }
void cfl_prof_scope_end(void *scope) { delete (ProfScope *)scope; }

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

// split of an X operand (rows of the feature table / the batch).  ABL_X_NOSPLIT (ablation build, tools/r06_xsplit_ab.sh): the three
// fragments are the truncated high halves only (one v_perm per pair, no subtract / mask chains) -- WRONG numbers, the instruction
// count of a kernel whose x arrives already split (round 6: what resident bf16 planes of the feature table could save at most)
__device__ __forceinline__ void split_frag_x(const float (&v)[8], bf16x8 (&f)[3]) {
#ifdef ABL_X_NOSPLIT
    u32x4 ph;
#pragma unroll
    for (int i = 0; i < 4; ++i) ph[i] = pack_hi16(v[2 * i], v[2 * i + 1]);
    f[0] = f[1] = f[2] = __builtin_bit_cast(bf16x8, ph);
#else
    split_frag(v, f);
#endif
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

#include "pair_proj.h"
#include "pair_grad.h"
#include "pair_mid.h"
#include "pair_finalize.h"

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
    bool pad512 = false;
    if (train && debug_env("CFL_EXACT_FP32") <= 0 && debug_env("CFL_DEBUG_GRAD_HALF") >= 0) {
        const int ht = s->D / 32;
        const bool paired = s->dist_type == CFL_DIST_SIAMESE && !s->directed;
        // (measured: -2.6 us at B = 512 and with weight-norm, -5.9 us at B = 1024, -1.6 us at B = 2048; +5 us at B = 4096; +0.8 us
        // with 192 workgroups on 256 CUs, config 4 -- hence the bounds)
        // Beyond 2048 rows per side the rows are split in two (one published half tile per finisher): measured against
        // P = 1 -2.5 us at B = 1536 and -2.5 .. -3.9 us at B = 2048, +0.9 us at B = 1024; against the 64-d form -1 us at B = 3072
        // Round 5: with eight waves per workgroup (cfl_grad_x3_half_w8_kernel, two per SIMD) the UNSPLIT tile wins far beyond 2048
        // rows per side -- same box, bench medians, step us with P = 1 / with the rows split in two (64-d tiles from 7168): B = 1280
        // 59.9 / 63.9, 1536 61.8 / 65.1, 2048 73.9 / 77.3, 2560 92.4 / 95.1, 3072 101.9 / 103.6, 3584 116.9 / 132.6; B = 4096
        // 144.9 / 142.7 (profiles/r05_b2048_ab.txt): one workgroup per tile up to 7680 rows (the staged row addresses: 60 KB of
        // LDS), rows padded to whole 512-row groups so that the eight-wave kernel applies
        if (!paired && ht * njobs >= 256 && ht * njobs <= 640 && pl->R <= 7680 && debug_env("CFL_DEBUG_P") <= 0) {
            const int P64 = P;                                           // (row split of the 64-d form, from above)
            pl->grad_half = true;
            P = 1;
            pad512 = pl->R > 2048 && debug_env("CFL_DEBUG_GRAD_W8") >= 0;
            if (!pad512 && pl->R > 6144) { pl->grad_half = false; P = P64; }   // (four waves only: the old bounds)
            else if (!pad512 && pl->R > 2048) P = 2;
        }
        // fewer half tiles than CUs but rows enough to split in two (config 4: 64 x 3 = 192 tiles, 2048 rows per side): 384
        // workgroups, one published tile per finisher.  Round 4, same box: config 4 49.9 -> 47.9 us against the 64-d tiles
        // with four row ranges (three published tiles per finisher); without the split 50.3 (profiles/r04_split_ab.txt)
        // (round 5: unsplit with eight waves where the rows come in whole 512-row groups -- 192 workgroups x 8 waves: config 4 48.2 ->
        // 47.4 us, profiles/r05_c4_p1_ab.txt)
        if (!paired && ht * njobs >= 128 && ht * njobs < 256 && pl->R >= 2048 && pl->R <= 6144 && debug_env("CFL_DEBUG_P") <= 0) {
            pl->grad_half = true;
            P = (round_up(pl->R, 256) % 512 == 0 && debug_env("CFL_DEBUG_GRAD_W8") >= 0) ? 1 : 2;
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
    pl->Rpad = (int)round_up(pl->R, pad512 ? 512 : 256 * P);  // grad: 64-row chunks x 4 (8) waves x P ranges
    // proj d split: one 128-d chunk per wave when that yields enough workgroups
    if (xrows < 0 || xrows > (1 << 24) || (xrows && !train)) return set_err(CFL_E_SHAPE, "extra scoring rows %lld", (long long)xrows);
    pl->Rx = (int)xrows;
    pl->Rxpad = (int)round_up(xrows, 32);
    pl->Rtot = pl->Rpad + pl->Rxpad;
    const int rtiles = (pl->R + 31) / 32 + pl->Rxpad / 32;
    const int nchunks = (s->D / 16 + 7) / 8;
    int S = (512 + rtiles * njobs - 1) / (rtiles * njobs);
    {
        // power of two below the wanted split -- or, training, EIGHT (one slice per XCD: the aligned launch order of proj and grad)
        // when 6 or 7 are wanted (round 5: B = 640 wants 7: 41.4 us per step with S = 8 against 43.2 with 4; B = 768 (6) 43.7 /
        // 44.4; B = 1024 (4) stays -- profiles/r05_s_ab.txt.  Not a general round-to-nearest: config 4 wants 3, and S = 4
        // instead of 2 costs it 1.3 us, profiles/r05_c4_s_ab.txt)
        const int want_s = S < 1 ? 1 : S;
        S = pow2_floor(want_s);
        if (train && S == 4 && want_s >= 6) S = 8;
    }
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
        // (round 5, the eight-wave weight gradient behind both: B = 768 43.8 / 44.1 us per step chunk-at-a-time / LDS-shared, B = 1024
        // 49.4 / 50.1, B = 1280 60.2 / 57.1 -- profiles/r05_kept_rows_ab.txt: the LDS-shared form from 2560 rows per call)
        const int kept_rows = debug_env("CFL_DEBUG_X3_KEPT_ROWS") > 0 ? debug_env("CFL_DEBUG_X3_KEPT_ROWS") : 2560;
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

// ... of a training call of `rows` rows per group that carries a validation batch of `val_rows` pairs per group as extra scoring
// rows (under data parallelism a rank trains its shard but scores the WHOLE validation batch: val_rows > rows)
extern "C" size_t cfl_workspace_bytes_val(const CflShape *s, int64_t rows, int64_t val_rows) {
    size_t need = cfl_workspace_bytes(s, rows, 2);
    if (!need) return 0;
    need /= sizeof(float);
    for (int kept = 0; kept < 2; ++kept) {
        Plan pl;
        if (make_plan(s, rows, 2, true, kept != 0, &pl, 2 * val_rows)) return 0;
        if (pl.total_floats > need) need = pl.total_floats;
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

// Fused push of the data-parallel one-shot exchange (GradFuse::dp_*): the weight-gradient launch stores [gradient | scalars]
// into the owners' slot arrays.  rows_tab / flags_tab: DEVICE tables [world]; scalars_remote: where the step's 16 scalars live
// in their owner's slot row (resolved on the host: their offset is static).  Honoured by plans with the fused tail only
// (*pushed tells the caller); without it the caller runs cfl_dp_rs_push on the flat buffer.
struct DpPush {
    float *const *rows_tab;
    unsigned *const *flags_tab;
    long long slice;
    int world;
    unsigned gen;
    unsigned *ticket;
    float *scalars_remote;
};

static int run_pairs(const CflShape *s, const CflNorm *norm, const CflLossCfg *loss,
                     const float *const *x, int groups, int64_t rows, const float *theta,
                     float *grad, float *scalars, float *scores, float *dists, void *workspace,
                     size_t workspace_bytes, hipStream_t st, const AdamFuse *adam = nullptr,
                     const IndexSrc *isrc = nullptr, CflThetaPlanes *kept = nullptr, const ExtraSrc *xs = nullptr,
                     const DpPush *dpp = nullptr, bool *pushed = nullptr) {
    if (pushed) *pushed = false;
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
            f.spin_limit = debug_env("CFL_DEBUG_SPIN_LIMIT") < 0 ? -1 : 0;
            {
                static const double handoff_s = [] {
                    const char *e = getenv("CFL_HANDOFF_TIMEOUT_S");
                    const double v = e ? atof(e) : 0.0;
                    return v > 0.0 ? v : CFL_HANDOFF_TIMEOUT_S;
                }();
                f.spin_ticks = (unsigned long long)(handoff_s * 1e8);   // s_memrealtime counts at 100 MHz
            }
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
            const bool dp_kernel = pl.grad_kernel == GK_HALF_W8 || pl.grad_kernel == GK_HALF || pl.grad_kernel == GK_HALF_SPLIT;
            if (dpp && !adam && dp_kernel) {
                // data parallel, one-shot exchange: the finished entries go straight to their owner ranks (GradFuse::dp_*).  The
                // half-tile kernels have a fused-push form; the other plans emit the flat gradient and the caller pushes it
                // with cfl_dp_rs_push (*pushed stays false)
                f.dp_rows = dpp->rows_tab; f.dp_flags = dpp->flags_tab; f.dp_slice = dpp->slice; f.dp_world = dpp->world;
                f.dp_gen = dpp->gen; f.dp_ticket = dpp->ticket;
                f.scalars = dpp->scalars_remote;
                if (pushed) *pushed = true;
            }
        }
        ga.tps = pl.xcd ? (s->D / (pl.grad_half ? 32 : 64)) / pl.S : 0;
        dim3 grid(s->D / 64, pl.P, nj + 1);
        ProfScope ps(st, CFL_K_GRAD);
        const size_t glds = 4 * 4 * 4 * 64 * sizeof(f32x4);
        const dim3 hgrid(s->D / 32, pl.P, nj + 1);
        const bool push = ga.fuse.dp_slice != 0;      // (set above for the half-tile kernels only)
        switch (pl.grad_kernel) {
            case GK_HALF_W8:
                if (push) hipLaunchKernelGGL(cfl_grad_x3_half_w8_dp_kernel, hgrid, dim3(512), glds, st, ga);
                else hipLaunchKernelGGL(cfl_grad_x3_half_w8_kernel, hgrid, dim3(512), glds, st, ga);
                break;
            case GK_HALF:
                if (push) hipLaunchKernelGGL(cfl_grad_x3_half_dp_kernel, hgrid, dim3(256), glds, st, ga);
                else hipLaunchKernelGGL(cfl_grad_x3_half_kernel, hgrid, dim3(256), glds, st, ga);
                break;
            case GK_HALF_SPLIT:
                if (push) hipLaunchKernelGGL(cfl_grad_x3_half_split_dp_kernel, hgrid, dim3(256), glds, st, ga);
                else hipLaunchKernelGGL(cfl_grad_x3_half_split_kernel, hgrid, dim3(256), glds, st, ga);
                break;
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

static int adam_tf_planes_impl(const CflShape *shape, float *theta, float *m, float *v, const float *grad,
                              float lr_t, float beta1, float beta2, float eps, float grad_scale,
                              CflThetaPlanes *planes, const float *scal_src, float *scal_dst, cfl_stream_t stream) {
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
                           v, grad, n4, lr_t, beta1, beta2, eps, grad_scale, pr, scal_src, scal_dst);
    }
    HIP_TRY(hipGetLastError());
    if (planes) planes->valid = 1;
    return CFL_OK;
}

extern "C" int cfl_adam_tf_planes(const CflShape *shape, float *theta, float *m, float *v, const float *grad,
                                  float lr_t, float beta1, float beta2, float eps, float grad_scale,
                                  CflThetaPlanes *planes, cfl_stream_t stream) {
    return adam_tf_planes_impl(shape, theta, m, v, grad, lr_t, beta1, beta2, eps, grad_scale, planes, nullptr, nullptr, stream);
}

// ===========================================================================================================================
// ABI 6: the whole data-parallel step behind one call (include/cfl_hip.h)
// ===========================================================================================================================
#define CFL_GRADBUF_PAD 64    // gradbuf = [cfl_layout.total floats of gradient | 16 scalars | 48 pad]

static int dp_check(const CflShape *shape, const CflLossCfg *loss, const float *theta, const float *m, const float *v,
                    const float *gradbuf, const CflDpExchange *ex, const CflAllReduce *ar, CflLayout *lay) {
    if (!loss || !theta || !m || !v || !gradbuf) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (loss->caffe_margin != 0.f && loss->lambda_m != 0.f)
        return set_err(CFL_E_SHAPE, "caffe_margin and lambda_m are exclusive (cfl/utils.py:72-73)");
    if (ex && ar) return set_err(CFL_E_SHAPE, "one exchange at most: CflDpExchange or CflAllReduce");
    int rc = cfl_layout(shape, lay);
    if (rc) return rc;
    if (ex && (ex->n_adam != lay->total || ex->n != lay->total + CFL_GRADBUF_PAD))
        return set_err(CFL_E_SHAPE, "CflDpExchange: n_adam=%lld n=%lld do not match the shape (%lld parameters + %d)",
                       (long long)ex->n_adam, (long long)ex->n, (long long)lay->total, CFL_GRADBUF_PAD);
    if (ar && (!ar->fn || !ar->comm || ar->world < 1)) return set_err(CFL_E_SHAPE, "CflAllReduce: NULL entry point / communicator");
    return CFL_OK;
}

// forward / backward of this rank's rows (+ the fused push when the exchange has device tables and the plan a fused tail),
// the exchange, the update.  xs: extra scoring rows (their scalars copy is written by the exchange's LAST kernel: global sums)
static int dp_one_step(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *const *x4,
                       const IndexSrc *isrc, int64_t rows, const CflLayout &lay, float *theta, float *m, float *v, float *gradbuf,
                       float lr_t, float beta1, float beta2, float eps, CflThetaPlanes *planes, CflDpExchange *ex,
                       const CflAllReduce *ar, void *workspace, size_t workspace_bytes, const ExtraSrc *xs, hipStream_t st) {
    float *scalars = gradbuf + lay.total;
    DpPush dpp;
    const DpPush *use = nullptr;
    if (ex && ex->dev_tables && debug_env("CFL_DP_PUSH_SEPARATE") <= 0) {
        const int par = (int)(ex->step & 1);
        void *const *tab = (void *const *)ex->dev_tables;
        dpp.rows_tab = (float *const *)(tab + (size_t)(par * 2 + 0) * CFL_DP_MAX_WORLD);
        dpp.flags_tab = (unsigned *const *)(tab + (size_t)(par * 2 + 1) * CFL_DP_MAX_WORLD);
        dpp.slice = ex->slice; dpp.world = ex->world;
        const uint32_t g = (uint32_t)((ex->step + 1) & 0xffffffffu);
        dpp.gen = g ? g : 1u;
        dpp.ticket = ex->tickets;
        const int64_t owner = lay.total / ex->slice;          // (n > total: the scalars always have an owner)
        dpp.scalars_remote = ex->peer_rows[par][owner] + (lay.total - owner * ex->slice);
        use = &dpp;
    }
    ExtraSrc xl;
    float *scalars_copy = nullptr;
    if (xs) { xl = *xs; scalars_copy = xl.scalars_copy; xl.scalars_copy = nullptr; }
    bool pushed = false;
    int rc = run_pairs(shape, norm, loss, x4, 2, rows, theta, gradbuf, scalars, nullptr, nullptr, workspace, workspace_bytes, st,
                       nullptr, isrc, planes, xs ? &xl : nullptr, use, &pushed);
    if (rc) return rc;
    if (ex) return cfl_dp_exchange_step(shape, ex, pushed ? 1 : 0, theta, m, v, gradbuf, lr_t, beta1, beta2, eps, planes,
                                        scalars_copy, (cfl_stream_t)st);
    float scale = 1.f;
    if (ar) {
        const int nrc = ar->fn(gradbuf, gradbuf, (size_t)(lay.total + CFL_GRADBUF_PAD), ar->dtype, ar->op, ar->comm, (void *)st);
        if (nrc) return set_err(CFL_E_HIP, "the caller's all-reduce returned %d", nrc);
        scale = 1.f / (float)ar->world;
    }
    return adam_tf_planes_impl(shape, theta, m, v, gradbuf, lr_t, beta1, beta2, eps, scale, planes,
                               scalars_copy ? scalars : nullptr, scalars_copy, (cfl_stream_t)st);
}

extern "C" int cfl_dp_push_fusable(const CflShape *shape, int64_t rows) {
    Plan pl;
    if (make_plan(shape, rows, 2, true, true, &pl)) return 0;
    return (pl.fused && (pl.grad_kernel == GK_HALF_W8 || pl.grad_kernel == GK_HALF || pl.grad_kernel == GK_HALF_SPLIT)) ? 1 : 0;
}

extern "C" int cfl_pair_dp_step_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *const x4[4],
                                       int64_t B, float *theta, float *m, float *v, float *gradbuf, float lr_t, float beta1,
                                       float beta2, float eps, CflThetaPlanes *planes, CflDpExchange *ex, const CflAllReduce *ar,
                                       void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    CflLayout lay;
    int rc = dp_check(shape, loss, theta, m, v, gradbuf, ex, ar, &lay);
    if (rc) return rc;
    if (!x4) return set_err(CFL_E_SHAPE, "x4 is NULL");
    return dp_one_step(shape, norm, loss, x4, nullptr, B, lay, theta, m, v, gradbuf, lr_t, beta1, beta2, eps, planes, ex, ar,
                       workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

extern "C" int cfl_pair_dp_step_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *table,
                                           int64_t table_rows, const int32_t *const idx4[4], int64_t idx_stride, int64_t B,
                                           float *theta, float *m, float *v, float *gradbuf, float lr_t, float beta1, float beta2,
                                           float eps, CflThetaPlanes *planes, CflDpExchange *ex, const CflAllReduce *ar,
                                           void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    CflLayout lay;
    int rc = dp_check(shape, loss, theta, m, v, gradbuf, ex, ar, &lay);
    if (rc) return rc;
    IndexSrc is = {table, table_rows, idx4, idx_stride};
    return dp_one_step(shape, norm, loss, nullptr, &is, B, lay, theta, m, v, gradbuf, lr_t, beta1, beta2, eps, planes, ex, ar,
                       workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

extern "C" int cfl_pair_dp_steps_idx_planes(
    const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *table, int64_t table_rows,
    const int32_t *pos_pairs, int64_t n_pos, const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
    int64_t batch_rows, int64_t shard_lo, int64_t rows, const uint8_t *switched, int64_t nsteps,
    const float *val_table, int64_t val_table_rows, const int32_t *val_pos_pairs, int64_t n_val_pos,
    const int32_t *val_neg_pairs, int64_t n_val_neg, int64_t val_pos_head, int64_t val_neg_head, int64_t val_batch_rows,
    const uint8_t *val_switched, const uint8_t *val_mask, float *const *ring_slots,
    float *theta, float *m, float *v, float *gradbuf, float lr, float beta1, float beta2, float eps,
    float *beta1_power, float *beta2_power, CflThetaPlanes *planes, CflDpExchange *ex, const CflAllReduce *ar,
    void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    CflLayout lay;
    int rc = dp_check(shape, loss, theta, m, v, gradbuf, ex, ar, &lay);
    if (rc) return rc;
    if (!pos_pairs || !neg_pairs || !beta1_power || !beta2_power) return set_err(CFL_E_SHAPE, "NULL pointer");
    if (nsteps <= 0 || batch_rows <= 0 || rows <= 0 || shard_lo < 0 || shard_lo + rows > batch_rows || pos_head < 0 || neg_head < 0)
        return set_err(CFL_E_SHAPE, "bad step window");
    int64_t nval = 0;
    if (val_mask) {
        if (!ring_slots || !val_pos_pairs || !val_neg_pairs || !val_table || val_batch_rows <= 0 || val_pos_head < 0 || val_neg_head < 0)
            return set_err(CFL_E_SHAPE, "validation fetch: NULL pointer / bad window");
        for (int64_t i = 0; i < nsteps; ++i) nval += val_mask[i] ? 1 : 0;
        for (int64_t k = 0; k < nval; ++k)
            if (!ring_slots[k]) return set_err(CFL_E_SHAPE, "ring slot %lld is NULL", (long long)k);
    }
    if (nsteps > (1ll << 40) / batch_rows || pos_head + nsteps * batch_rows > n_pos || neg_head + nsteps * batch_rows > n_neg ||
        (nval && (val_pos_head + nval * val_batch_rows > n_val_pos || val_neg_head + nval * val_batch_rows > n_val_neg)))
        return set_err(CFL_E_SHAPE, "step window runs past the pair lists");
    float b1p = *beta1_power, b2p = *beta2_power;
    int64_t k = 0;
    for (int64_t i = 0; i < nsteps; ++i) {
        const int32_t *ps = pos_pairs + 2 * (pos_head + i * batch_rows + shard_lo);
        const int32_t *ng = neg_pairs + 2 * (neg_head + i * batch_rows + shard_lo);
        const int c0 = (switched && switched[i]) ? 1 : 0;
        const int32_t *idx4[4] = {ps + c0, ps + (1 - c0), ng + c0, ng + (1 - c0)};
        const float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
        IndexSrc is = {table, table_rows, idx4, 2};
        ExtraSrc xs;
        const bool val = val_mask && val_mask[i];
        if (val) {
            const int32_t *vp = val_pos_pairs + 2 * (val_pos_head + k * val_batch_rows);
            const int32_t *vn = val_neg_pairs + 2 * (val_neg_head + k * val_batch_rows);
            const int v0 = (val_switched && val_switched[k]) ? 1 : 0;
            xs = {val_table, val_table_rows, {vp + v0, vp + (1 - v0), vn + v0, vn + (1 - v0)}, 2, val_batch_rows,
                  ring_slots[k] + CFL_S_COUNT, ring_slots[k]};
            ++k;
        }
        rc = dp_one_step(shape, norm, loss, nullptr, &is, rows, lay, theta, m, v, gradbuf, lr_t, beta1, beta2, eps, planes, ex, ar,
                         workspace, workspace_bytes, val ? &xs : nullptr, (hipStream_t)stream);
        if (rc) return rc;
        b1p *= beta1;
        b2p *= beta2;
    }
    *beta1_power = b1p;
    *beta2_power = b2p;
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
