// cfl_eval.hip -- ROC AUC and sign accuracy of a split's pair scores on the GPU
// (cfl/utils.py:227-274 dist_eval: roc_auc_score(y_true, y_score) and accuracy by the sign of the score).
//
// AUC = ( sum over positives of [#negatives with a smaller score + 0.5 * #negatives with an equal score] )
//       / (n_pos * n_neg)            (the Mann-Whitney statistic; ties get half credit, as sklearn's
//                                     trapezoidal ROC integration does)
// Steps: pack (score, label) into one sortable 64-bit key; bitonic sort in HBM (strides below 2048 run inside
// one LDS-resident kernel per merge step); exclusive scan of the negative indicator; one binary search pair
// per positive; fixed-order reductions in double.  All integer / comparison work: HBM-streaming kernels.
#include <hip/hip_runtime.h>

#include "../../include/cfl_hip.h"

extern int cfl_set_err(int code, const char *fmt, ...);

namespace {

typedef unsigned long long u64;
constexpr int TPB = 256;
constexpr int LOCAL_N = 2048;   // keys sorted / merged inside one workgroup's LDS

__device__ __forceinline__ unsigned sortable(float f) {
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;                        // -0.0 == +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone float -> unsigned
}

// key = sortable(score) << 1 | label ; padding keys are all ones (sort to the end)
__global__ __launch_bounds__(TPB) void auc_pack_kernel(const float *pos, long long np, const float *neg,
                                                       long long nn, long long N, u64 *keys,
                                                       unsigned *okcount) {
    __shared__ unsigned red[TPB];
    unsigned ok = 0;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < N; i += (long long)gridDim.x * TPB) {
        u64 k = ~0ull;
        if (i < np) {
            k = ((u64)sortable(pos[i]) << 1) | 1ull;
            ok += pos[i] > 0.f ? 1u : 0u;
        } else if (i < np + nn) {
            k = ((u64)sortable(neg[i - np]) << 1);
            ok += neg[i - np] <= 0.f ? 1u : 0u;
        }
        keys[i] = k;
    }
    red[threadIdx.x] = ok;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) okcount[blockIdx.x] = red[0];
}

// bitonic network, global stage: compare-exchange partners j apart inside blocks of size k
__global__ __launch_bounds__(TPB) void bitonic_global_kernel(u64 *keys, long long N, long long k, long long j) {
    for (long long t = blockIdx.x * (long long)TPB + threadIdx.x; t < N / 2; t += (long long)gridDim.x * TPB) {
        const long long i = 2 * t - (t & (j - 1));     // index with bit j clear
        const long long p = i + j;
        const bool up = (i & k) == 0;
        const u64 a = keys[i], b = keys[p];
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
    }
}

// all stages with j < LOCAL_N of the merge step k (or the whole sort of a LOCAL_N chunk when k <= LOCAL_N)
__global__ __launch_bounds__(TPB) void bitonic_local_kernel(u64 *keys, long long k_lo, long long k_hi) {
    __shared__ u64 s[LOCAL_N];
    const long long base = (long long)blockIdx.x * LOCAL_N;
    for (int i = threadIdx.x; i < LOCAL_N; i += TPB) s[i] = keys[base + i];
    __syncthreads();
    for (long long k = k_lo; k <= k_hi; k <<= 1) {
        for (long long j = (k < LOCAL_N ? k : LOCAL_N) >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < LOCAL_N / 2; t += TPB) {
                const int i = 2 * t - (t & ((int)j - 1));
                const int p = i + (int)j;
                const bool up = ((base + i) & k) == 0;
                const u64 a = s[i], b = s[p];
                if ((a > b) == up) { s[i] = b; s[p] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < LOCAL_N; i += TPB) keys[base + i] = s[i];
}

// exclusive scan of the negative indicator, three phases (chunks of LOCAL_N)
__global__ __launch_bounds__(TPB) void scan_block_kernel(const u64 *keys, unsigned *cnt, unsigned *blocksum) {
    __shared__ unsigned s[TPB];
    const long long base = (long long)blockIdx.x * LOCAL_N;
    constexpr int PER = LOCAL_N / TPB;
    unsigned v[PER], run = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const u64 k = keys[base + threadIdx.x * PER + e];
        v[e] = run;
        run += (k != ~0ull && (k & 1ull) == 0) ? 1u : 0u;
    }
    s[threadIdx.x] = run;
    __syncthreads();
    for (int o = 1; o < TPB; o <<= 1) {          // Hillis-Steele inclusive scan of the per-thread totals
        const unsigned add = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    const unsigned before = threadIdx.x ? s[threadIdx.x - 1] : 0u;
#pragma unroll
    for (int e = 0; e < PER; ++e) cnt[base + threadIdx.x * PER + e] = before + v[e];
    if (threadIdx.x == TPB - 1) blocksum[blockIdx.x] = s[TPB - 1];
}
__global__ void scan_sums_kernel(unsigned *blocksum, int nblocks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < nblocks; ++i) { const unsigned t = blocksum[i]; blocksum[i] = run; run += t; }
    }
}
__global__ __launch_bounds__(TPB) void scan_add_kernel(unsigned *cnt, const unsigned *blocksum) {
    const long long base = (long long)blockIdx.x * LOCAL_N;
    const unsigned add = blocksum[blockIdx.x];
    for (int i = threadIdx.x; i < LOCAL_N; i += TPB) cnt[base + i] += add;
}

// per positive: negatives below + half the negatives tied with it; block partial sums in double
__global__ __launch_bounds__(TPB) void auc_rank_kernel(const u64 *keys, const unsigned *cnt, long long N,
                                                       long long total, long long nn, double *partial) {
    __shared__ double red[TPB];
    double acc = 0.0;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const u64 k = keys[i];
        if ((k & 1ull) == 0) continue;                  // negatives contribute nothing
        const u64 sc = k >> 1;
        long long lo = 0, hi = total;                   // first index with score >= sc
        while (lo < hi) { const long long m = (lo + hi) >> 1; if ((keys[m] >> 1) < sc) lo = m + 1; else hi = m; }
        const long long first = lo;
        hi = total;                                     // first index with score > sc
        while (lo < hi) { const long long m = (lo + hi) >> 1; if ((keys[m] >> 1) <= sc) lo = m + 1; else hi = m; }
        const long long last = lo;
        const unsigned below = cnt[first];
        const unsigned upto = last < N ? cnt[last] : (unsigned)nn;
        acc += (double)below + 0.5 * (double)(upto - below);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void auc_final_kernel(const double *partial, int nparts, const unsigned *okcount, int nok,
                                 long long np, long long nn, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double u = 0.0;
        for (int i = 0; i < nparts; ++i) u += partial[i];
        unsigned long long ok = 0;
        for (int i = 0; i < nok; ++i) ok += okcount[i];
        out[0] = (np > 0 && nn > 0) ? u / ((double)np * (double)nn) : 0.0;
        out[1] = (double)ok / (double)(np + nn);
    }
}

constexpr int RANK_BLOCKS = 1024, PACK_BLOCKS = 256;

inline long long pow2_at_least(long long n) {
    long long p = LOCAL_N;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace

extern "C" size_t cfl_auc_workspace_bytes(int64_t n_pos, int64_t n_neg) {
    if (n_pos < 0 || n_neg < 0 || n_pos + n_neg <= 0 || n_pos + n_neg > (1ll << 30)) return 0;
    const long long N = pow2_at_least(n_pos + n_neg);
    return (size_t)N * (sizeof(u64) + sizeof(unsigned)) + (size_t)(N / LOCAL_N) * sizeof(unsigned) +
           RANK_BLOCKS * sizeof(double) + PACK_BLOCKS * sizeof(unsigned) + 256;
}

extern "C" int cfl_auc(const float *scores_pos, int64_t n_pos, const float *scores_neg, int64_t n_neg,
                       double *out, void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    if (!scores_pos || !scores_neg || !out || !workspace || n_pos <= 0 || n_neg <= 0)
        return cfl_set_err(CFL_E_SHAPE, "cfl_auc: bad argument");
    const size_t need = cfl_auc_workspace_bytes(n_pos, n_neg);
    if (need == 0) return cfl_set_err(CFL_E_SHAPE, "cfl_auc: too many scores");
    if (workspace_bytes < need) return cfl_set_err(CFL_E_WORKSPACE, "cfl_auc: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long long total = n_pos + n_neg, N = pow2_at_least(total);
    const int nchunks = (int)(N / LOCAL_N);
    char *w = (char *)workspace;
    u64 *keys = (u64 *)w;                       w += (size_t)N * sizeof(u64);
    double *partial = (double *)w;              w += RANK_BLOCKS * sizeof(double);
    unsigned *cnt = (unsigned *)w;              w += (size_t)N * sizeof(unsigned);
    unsigned *blocksum = (unsigned *)w;         w += (size_t)nchunks * sizeof(unsigned);
    unsigned *okcount = (unsigned *)w;

    hipLaunchKernelGGL(auc_pack_kernel, dim3(PACK_BLOCKS), dim3(TPB), 0, st, scores_pos, (long long)n_pos,
                       scores_neg, (long long)n_neg, N, keys, okcount);
    // sort every LOCAL_N chunk completely, then merge: global stages for j >= LOCAL_N, the rest in LDS
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(nchunks), dim3(TPB), 0, st, keys, 2ll, (long long)LOCAL_N);
    const int gblocks = (int)((N / 2 + TPB - 1) / TPB > 4096 ? 4096 : (N / 2 + TPB - 1) / TPB);
    for (long long k = 2 * LOCAL_N; k <= N; k <<= 1) {
        for (long long j = k >> 1; j >= LOCAL_N; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_kernel, dim3(gblocks), dim3(TPB), 0, st, keys, N, k, j);
        hipLaunchKernelGGL(bitonic_local_kernel, dim3(nchunks), dim3(TPB), 0, st, keys, k, k);
    }
    hipLaunchKernelGGL(scan_block_kernel, dim3(nchunks), dim3(TPB), 0, st, keys, cnt, blocksum);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(64), 0, st, blocksum, nchunks);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nchunks), dim3(TPB), 0, st, cnt, blocksum);
    hipLaunchKernelGGL(auc_rank_kernel, dim3(RANK_BLOCKS), dim3(TPB), 0, st, keys, cnt, N, total, (long long)n_neg,
                       partial);
    hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(64), 0, st, partial, RANK_BLOCKS, okcount, PACK_BLOCKS,
                       (long long)n_pos, (long long)n_neg, out);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "cfl_auc launch failed");
}
