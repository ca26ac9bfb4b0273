// Microbenchmark: what does the SIZE of a by-value kernel-argument struct cost per launch, and what do reads from
// it cost at the start of a launch?  (experiment helper; result recorded in DESIGN.md)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Args { int v[N]; };
template <int N> __global__ void k_first(Args<N> a, int *out) { if (a.v[0] == 12345) out[0] = 1; }
template <int N> __global__ void k_last(Args<N> a, int *out) { if (a.v[N - 1] == 12345) out[0] = 1; }
template <int N> __global__ void k_all(Args<N> a, int *out) {   // one value from every 64-byte line, independent
    int s = 0;
#pragma unroll
    for (int i = 0; i < N; i += 16) s += a.v[i];
    if (s == 12345) out[0] = 1;
}
template <int N> __global__ void k_dev(const Args<N> *a, int *out) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < N; i += 16) s += a->v[i];
    if (s == 12345) out[0] = 1;
}
int main() {
    int *out; hipMalloc(&out, 64);
    void *dev; hipMalloc(&dev, 8192); hipMemset(dev, 0, 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
#define TIME(expr)                                                                             \
    [&] { for (int i = 0; i < 50; ++i) { expr; } hipDeviceSynchronize(); hipEventRecord(e0);  \
          for (int i = 0; i < iters; ++i) { expr; } hipEventRecord(e1); hipEventSynchronize(e1); \
          float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3 / iters; }()
#define ROW(N)                                                                                                   \
    { Args<N> h = {};                                                                                            \
      const double f = TIME(hipLaunchKernelGGL(k_first<N>, dim3(512), dim3(256), 0, 0, h, out));                 \
      const double l = TIME(hipLaunchKernelGGL(k_last<N>, dim3(512), dim3(256), 0, 0, h, out));                  \
      const double a = TIME(hipLaunchKernelGGL(k_all<N>, dim3(512), dim3(256), 0, 0, h, out));                   \
      const double d = TIME(hipLaunchKernelGGL(k_dev<N>, dim3(512), dim3(256), 0, 0, (const Args<N> *)dev, out)); \
      printf("args %5d B : first dword %5.2f  last dword %5.2f  every line %5.2f  | same struct in device memory %5.2f us/launch\n", \
             (int)sizeof(h), f, l, a, d); }
    ROW(16) ROW(32) ROW(64) ROW(128) ROW(256) ROW(512) ROW(768) ROW(1000)
    return 0;
}
