// How long does a kernel wait for the part of its argument block that is NOT preloaded into SGPRs?  (round 6)
// Every launch gets a fresh kernarg slot in device memory, written by the host: the first s_load of it misses the scalar cache and the
// L2.  The kernel stamps s_memtime, loads one dword at byte offset 200 of its (256-byte) argument struct, waits, stamps again.
// Launched back to back in one stream (as the library's launches are), with and without a dependent predecessor that takes ~5 us.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 tools/experiments/kernarg_latency.hip -o build/kernarg_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
struct Big { int lead[14]; int pad[36]; int field; int pad2[13]; unsigned long long* out; };
__global__ void probe(Big a) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  int v;
  asm volatile("s_load_dword %0, %1, 0xc8\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) { a.out[0] = t1 - t0; a.out[1] = (unsigned long long)v; }
}
__global__ void busy(float* p, int n) {
  float x = p[threadIdx.x];
  for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
  p[threadIdx.x] = x;
}
int main() {
  const int N = 200;
  unsigned long long* out; hipMalloc(&out, N * 16);
  float* p; hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<unsigned long long> h(N * 2);
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < N; ++i) {
        if (mode) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, 0, p, 600);
        Big a{}; a.field = i; a.out = out + 2 * i;
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, a);
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, N * 16, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> d;
    for (int i = 0; i < N; ++i) d.push_back(h[2 * i]);
    std::sort(d.begin(), d.end());
    printf("%s: s_load of a non-preloaded kernarg dword, s_memtime ticks (100 MHz? see below): median %llu  p10 %llu  p90 %llu\n",
           mode ? "behind a ~5 us kernel" : "back to back probes  ", d[N / 2], d[N / 10], d[N * 9 / 10]);
  }
  // tick length: two stamps around a known delay
  return 0;
}
