// theta_planes.h -- the kept bf16 planes of theta (include/cfl_hip.h: CflThetaPlanes), as written by a kernel that updates
// theta element-wise over the FLAT array: the stand-alone TF-Adam of the data-parallel step (cfl_adam_tf_planes) and the
// all-gather of the one-shot exchange (cfl_dp_rs_gather_planes).  The fused weight-gradient tails of cfl_hip.hip write
// the same planes from their own tile registers (one 16-byte store per plane and lane); here a thread owns the four
// consecutive floats Wf[nt][g][q][c16][0..3] = W[d = 16 g + 4 q + e][col = 16 nt + c16] and stores the matching four
// ushorts of each plane (8 bytes; the 16 threads of a (q, c16) row pair fill whole 512-byte runs per plane).
// Bit-identical to cfl_wplanes_kernel / split_frag_rne: the same pairwise round-to-nearest split of (e0, e1), (e2, e3).
#ifndef CFL_THETA_PLANES_H
#define CFL_THETA_PLANES_H

#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/cfl_hip.h"

typedef __bf16 tp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tp_f32x2 __attribute__((ext_vector_type(2)));
typedef float tp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int tp_u32x2 __attribute__((ext_vector_type(2)));

#define CFL_PLANE_REGIONS 4   // weight matrices of theta: outputs / prototype heads of up to two encoders

struct ThetaPlaneRegions {
    long long off[CFL_PLANE_REGIONS], end[CFL_PLANE_REGIONS];   // [off, end) floats of theta: one fragment-major Wf array each
    unsigned short *planes;                                     // CflThetaPlanes.buf (NULL: nothing is written)
    int n, G;                                                   // regions used; G = D / 16
};

// h = bf16_rne(v), m = bf16_rne(v - h), l = bf16_rne(v - h - m) of two values, packed {e1, e0} per level
__device__ __forceinline__ void tp_split_pair_rne(float v0, float v1, unsigned &h, unsigned &m, unsigned &l) {
    const tp_f32x2 v = {v0, v1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, tp_bf16x2));
    const tp_f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const tp_f32x2 r = v - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, tp_bf16x2));
    const tp_f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r - mf, tp_bf16x2));
}

// th = the UPDATED theta[base .. base + 4), base a multiple of 4.  No-op outside the weight matrices.
__device__ __forceinline__ void theta_planes_store4(const ThetaPlaneRegions &pr, long long base, const tp_f32x4 th) {
    if (!pr.planes) return;
    for (int r = 0; r < pr.n; ++r) {
        if (base < pr.off[r] || base >= pr.end[r]) continue;
        const long long rel = base - pr.off[r];
        const long long blk = rel >> 8;                  // 1 KiB block (nt, g)
        const int u = (int)(rel >> 2) & 63, q = u >> 4, c16 = u & 15;
        const int nt = (int)(blk / pr.G), g = (int)(blk % pr.G);
        const int lane = (2 * (g & 1) + (q >> 1)) * 16 + c16, j0 = 4 * (q & 1);
        const int Q = pr.G >> 1;
        unsigned h0, m0, l0, h1, m1, l1;
        tp_split_pair_rne(th[0], th[1], h0, m0, l0);
        tp_split_pair_rne(th[2], th[3], h1, m1, l1);
        unsigned short *dst = pr.planes + 3 * pr.off[r] + ((size_t)(nt * Q + (g >> 1)) * 3) * 512 + lane * 8 + j0;
        *(tp_u32x2 *)dst = (tp_u32x2){h0, h1};
        *(tp_u32x2 *)(dst + 512) = (tp_u32x2){m0, m1};
        *(tp_u32x2 *)(dst + 1024) = (tp_u32x2){l0, l1};
        return;
    }
}

// host: the weight matrices of `s` (heads with a Wf array; the monomer gate head has none) and the caller's plane buffer
static inline int theta_plane_regions(const CflShape *s, void *buf, ThetaPlaneRegions *pr) {
    CflLayout lay;
    const int rc = cfl_layout(s, &lay);
    if (rc) return rc;
    memset(pr, 0, sizeof(*pr));
    pr->planes = (unsigned short *)buf;
    pr->G = s->D / 16;
    for (int e = 0; e < (s->directed ? 2 : 1); ++e) {
        const CflHead *hh[2] = {&lay.enc[e].outputs, &lay.enc[e].proto};
        for (int k = 0; k < 2; ++k)
            if (hh[k]->w >= 0 && pr->n < CFL_PLANE_REGIONS) {
                pr->off[pr->n] = hh[k]->w;
                pr->end[pr->n] = hh[k]->w + (long long)hh[k]->npad * s->D;
                ++pr->n;
            }
    }
    return CFL_OK;
}

#endif
