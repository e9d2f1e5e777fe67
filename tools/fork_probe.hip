// Does a stream stall after it forks work to a second stream?  (round 5: in kernel traces of the overlapped forward tail the
// caller's queue ran nothing for 0.3-1 ms after the fork while the side queue ran alone.)  Stand-alone, no tracer: every kernel
// stamps wall_clock64() (100 MHz, device-wide) at its start and end.
//   main:  A x nA ... | fork | B x nB            side:  wait(fork) | C x nC | join
// Variants: event flags (default / hipEventDisableSystemFence), side stream priority, the order in which the host enqueues
// B and C, and a fork through a device flag instead of a HIP event.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/fork_probe tools/fork_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void spin(unsigned long long* stamps, int slot, int ticks) {      // `ticks` of the 100 MHz clock per workgroup
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = wall_clock64(); }
}
__global__ void set_flag(unsigned* flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void wait_flag(unsigned* flag, unsigned v, int limit) {
  int n = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v && ++n < limit) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
  const int nA = 10, nB = 20, nC = 40, reps = 30;
  const int wgsB = 64, wgsC = 128, lds = 0;
  CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  unsigned long long* stamps;
  unsigned* flag;
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 4096));
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  std::vector<unsigned long long> h(2 * 4096);
  struct Variant { const char* name; unsigned evflags; int prio; bool side_first; bool flag_fork; };
  const Variant vs[] = {
      {"event default flags, low-priority side, side enqueued first", hipEventDisableTiming, 1, true, false},
      {"event no system fence, low-priority side, side first", hipEventDisableTiming | hipEventDisableSystemFence, 1, true, false},
      {"event default, plain non-blocking side, side first", hipEventDisableTiming, 0, true, false},
      {"event default, low-priority side, MAIN's work enqueued first", hipEventDisableTiming, 1, false, false},
      {"event default, plain side, MAIN first", hipEventDisableTiming, 0, false, false},
      {"device-flag fork, low-priority side, side first", hipEventDisableTiming, 1, true, true},
      {"device-flag fork, plain side, MAIN first", hipEventDisableTiming, 0, false, true},
  };
  for (const Variant& v : vs) {
    hipStream_t mainS, side;
    CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (v.prio) CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
    else CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, v.evflags));
    CK(hipEventCreateWithFlags(&join, v.evflags));
    std::vector<double> stall, spanB, spanC;
    unsigned gen = 0;
    for (int r = 0; r < reps; ++r) {
      int slot = 0;
      for (int i = 0; i < nA; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), lds, mainS, stamps, slot++, 1000);      // 10 us each, fills the chip
      const int firstB = nA, firstC = nA + nB;
      ++gen;
      if (v.flag_fork) hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, mainS, flag, gen);
      else CK(hipEventRecord(fork, mainS));
      auto enqueue_side = [&] {
        if (v.flag_fork) hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, side, flag, gen, 1 << 24);
        else CK(hipStreamWaitEvent(side, fork, 0));
        for (int i = 0; i < nC; ++i) hipLaunchKernelGGL(spin, dim3(wgsC), dim3(512), 131072, side, stamps, firstC + i, 2000);  // 20 us, 128 KB LDS: one per CU
        CK(hipEventRecord(join, side));
      };
      auto enqueue_main = [&] {
        for (int i = 0; i < nB; ++i) hipLaunchKernelGGL(spin, dim3(wgsB), dim3(256), lds, mainS, stamps, firstB + i, 800);     // 8 us each
      };
      if (v.side_first) { enqueue_side(); enqueue_main(); } else { enqueue_main(); enqueue_side(); }
      CK(hipStreamWaitEvent(mainS, join, 0));
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), lds, mainS, stamps, nA + nB + nC, 100);
      CK(hipStreamSynchronize(mainS));
      CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * (nA + nB + nC + 1), hipMemcpyDeviceToHost));
      if (r < 3) continue;
      const double endA = (double)h[2 * (nA - 1) + 1];
      stall.push_back(((double)h[2 * firstB] - endA) / 100.0);
      spanB.push_back(((double)h[2 * (firstB + nB - 1) + 1] - endA) / 100.0);
      spanC.push_back(((double)h[2 * (firstC + nC - 1) + 1] - endA) / 100.0);
    }
    auto med = [](std::vector<double> x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    printf("%-64s | first B starts %7.1f us after the last A ends; B chain done at %7.1f (alone: %d us); C chain done at %7.1f (alone: %d us)\n", v.name,
           med(stall), med(spanB), nB * 8, med(spanC), nC * 20);
    CK(hipStreamDestroy(mainS)); CK(hipStreamDestroy(side)); CK(hipEventDestroy(fork)); CK(hipEventDestroy(join));
  }
  return 0;
}
