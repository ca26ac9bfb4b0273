// Semantics check of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) as used by gemm_gather.h:
// image img[k][m]; lane (q = l>>4, w = l&15) supplies &img[8q + (w>>2)][4*(w&3)] and must receive
// { img[8q+0][w], img[8q+1][w], img[8q+2][w], img[8q+3][w] }.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out) {
    __shared__ short img[32][64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) ((short *)img)[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, q = l >> 4, w = l & 15;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)&img[8 * q + (w >> 2)][4 * (w & 3)]);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int want = (8 * (l >> 4) + j) * 64 + (l & 15);
            if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h[l * 4 + j], want); ++bad; }
        }
    printf("tr16_b64: %d mismatches\n", bad);
    return bad != 0;
}
