// How long does a kernel wait for its first kernel argument?  (development aid)
// hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=14] tools/experiments/kernarg_probe.hip -o build/kernarg_probe[_pre]
// Each workgroup's wave 0 stamps s_memtime at entry and again once an argument value has reached a register; the probe is
// launched behind a kernel that touches 64 MB (so the scalar cache / L2 hold nothing of the kernarg segment, as in the
// network, where every launch has its own freshly written kernarg block).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Tail { long long pad[20]; int last; };
__global__ void probe(unsigned long long* out, int v, Tail t) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  int x;
  asm volatile("s_mov_b32 %0, %1" : "=s"(x) : "s"(v));          // first use of a scalar argument
  const unsigned long long t1 = __builtin_readcyclecounter();
  int y;
  asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(t.last));     // an argument beyond the preloaded range
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t1 - t0;
    out[blockIdx.x * 4 + 1] = t2 - t0;
    out[blockIdx.x * 4 + 2] = (unsigned long long)(x + y);
  }
}
__global__ void trash(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f;
}
int main() {
  const int WG = 256;
  unsigned long long* d; CK(hipMalloc(&d, WG * 4 * 8));
  float* big; const size_t n = 16u << 20; CK(hipMalloc(&big, n * 4)); CK(hipMemset(big, 0, n * 4));
  Tail t{}; t.last = 3;
  std::vector<double> a, b;
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL(trash, dim3(1024), dim3(256), 0, 0, big, n);
    hipLaunchKernelGGL(probe, dim3(WG), dim3(256), 0, 0, d, rep, t);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(WG * 4);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    for (int w = 0; w < WG; ++w) { a.push_back((double)h[w * 4]); b.push_back((double)h[w * 4 + 1]); }
  }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("cycles (s_memtime, 100 MHz ticks x?) entry -> first scalar argument usable: median %.0f p10 %.0f p90 %.0f\n", a[a.size() / 2], a[a.size() / 10], a[a.size() * 9 / 10]);
  printf("                                     entry -> argument at byte 168 usable:    median %.0f p10 %.0f p90 %.0f\n", b[b.size() / 2], b[b.size() / 10], b[b.size() * 9 / 10]);
  return 0;
}
