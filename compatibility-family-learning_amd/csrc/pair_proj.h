// pair_proj.h -- projection kernels of the pair step (chunk-at-a-time exact fp32 / bf16x3 on kept planes, streaming, bf16x3 with LDS-shared W planes)
// Part of the pair-step translation unit: included by cfl_hip.hip (and nothing else) behind the common device helpers; see the
// header comment of cfl_hip.hip for the launch structure and the fragment-major layouts, DESIGN.md section 4 for what runs when.
#pragma once

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
                split_frag_x(v, af[mt]);
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
                split_frag_x(v, af[mt]);
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

