// How fast does a wave run straight-line code it has never fetched?  (development aid)
// hipcc --offload-arch=gfx950 -O3 tools/experiments/icache_probe.hip -o build/icache_probe
// One wave per workgroup runs a block of N independent 8-byte VALU instructions twice: the first pass fetches the code from
// beyond the instruction cache (cold: the cache is invalidated at every dispatch), the second finds it there.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KB>
__global__ void probe(unsigned long long* out) {
  unsigned long long t[3];
  unsigned v = threadIdx.x;
  for (int pass = 0; pass < 2; ++pass) {
    t[pass] = __builtin_readcyclecounter();
    // KB * 128 instructions of 8 bytes (v_add_u32 with a 32-bit literal) = KB KiB of code
#ifdef SCALAR_CODE
    // scalar ALU instructions with a 32-bit literal (8 bytes, one per cycle when the code is there): 4 independent chains
    asm volatile(".rept %1\n s_add_u32 s20, s20, 0x12345\n s_add_u32 s21, s21, 0x12345\n s_add_u32 s22, s22, 0x12345\n s_add_u32 s23, s23, 0x12345\n .endr"
                 : "+v"(v) : "n"(KB * 32) : "s20", "s21", "s22", "s23", "scc");
#else
    asm volatile(".rept %1\n v_add_u32 %0, 0x12345, %0\n .endr" : "+v"(v) : "n"(KB * 128));
#endif
  }
  t[2] = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t[1] - t[0];
    out[blockIdx.x * 4 + 1] = t[2] - t[1];
    out[blockIdx.x * 4 + 2] = v;
  }
}
template <int KB>
int run(unsigned long long* d, int wgs, int threads) {
  std::vector<double> a, b;
  for (int rep = 0; rep < 10; ++rep) {
    hipLaunchKernelGGL(probe<KB>, dim3(wgs), dim3(threads), 0, 0, d);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(wgs * 4);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    for (int w = 0; w < wgs; ++w) { a.push_back((double)h[w * 4]); b.push_back((double)h[w * 4 + 1]); }
  }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("%3d KiB of straight-line code, %4d workgroups x %4d threads: first pass %7.0f cycles (p90 %7.0f), second pass %7.0f  -> cold fetch costs %5.0f cycles = %.2f cycles per byte\n",
         KB, wgs, threads, a[a.size() / 2], a[a.size() * 9 / 10], b[b.size() / 2], a[a.size() / 2] - b[b.size() / 2], (a[a.size() / 2] - b[b.size() / 2]) / (KB * 1024.0));
  return 0;
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 4096 * 4 * 8));
  for (int threads : {64, 512}) {
    run<1>(d, 256, threads); run<4>(d, 256, threads); run<16>(d, 256, threads); run<48>(d, 256, threads);
  }
  run<16>(d, 1024, 64);
  return 0;
}
