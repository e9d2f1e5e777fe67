// Launch-floor probe: what one DEPENDENT kernel boundary costs on this stack, measured, so that DESIGN's
// per-layer ceilings rest on one number.  Back-to-back launches of trivial kernels on ONE stream, timed by HIP events
// around the whole batch (= GPU timeline per launch when the host keeps ahead) and by the host clock around the enqueue
// loop (= host cost per launch); grids of 256 and 1024 workgroups; with no argument, a 200-byte by-value struct (the size
// of IgemmArgs) and the same struct whose fields are read; eager and replayed from a HIP graph.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { long long v[25]; };   // 200 bytes

__global__ void k_empty() {}
__global__ void k_args(Big b) { if (b.v[3] == 0x7fffffffffffffffLL) __builtin_trap(); }
__global__ void k_write(float* p, Big b) { if (threadIdx.x == 0) p[blockIdx.x * 16] = (float)b.v[blockIdx.x % 25]; }
__global__ void k_rw(const float* q, float* p, Big b) {            // reads what the previous launch wrote: a true dependency
  if (threadIdx.x == 0) p[blockIdx.x * 16] = q[blockIdx.x * 16] + (float)b.v[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename F>
static int run(const char* name, int n, hipStream_t s, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 200; ++i) launch(i);
  CK(hipStreamSynchronize(s));
  double best_gpu = 1e9, best_host = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a, s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < n; ++i) launch(i);
    auto t1 = std::chrono::high_resolution_clock::now();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double gpu = ms * 1e3 / n, host = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    if (gpu < best_gpu) best_gpu = gpu;
    if (host < best_host) best_host = host;
  }
  printf("%-64s  timeline %5.2f us/launch   host enqueue %5.2f us/launch\n", name, best_gpu, best_host);
  return 0;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  float *p, *q;
  CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&q, 1 << 20));
  CK(hipMemset(p, 0, 1 << 20)); CK(hipMemset(q, 0, 1 << 20));
  Big big{};
  const int n = 4000;
  printf("back-to-back launches on one stream, best of 5 batches of %d (eager)\n", n);
  for (int wgs : {256, 1024}) {
    for (int thr : {64, 256, 512}) {
      char nm[128];
      snprintf(nm, sizeof nm, "empty kernel, %4d x %3d threads, no arguments", wgs, thr);
      if (run(nm, n, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(thr), 0, s); })) return 1;
      snprintf(nm, sizeof nm, "empty kernel, %4d x %3d threads, 200-byte argument", wgs, thr);
      if (run(nm, n, s, [&](int) { hipLaunchKernelGGL(k_args, dim3(wgs), dim3(thr), 0, s, big); })) return 1;
    }
    char nm[128];
    snprintf(nm, sizeof nm, "one 4-byte store per workgroup, %4d x 256, 200-byte argument", wgs);
    if (run(nm, n, s, [&](int) { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, s, p, big); })) return 1;
    snprintf(nm, sizeof nm, "reads the previous launch's store, %4d x 256 (true dependency)", wgs);
    if (run(nm, n, s, [&](int i) { hipLaunchKernelGGL(k_rw, dim3(wgs), dim3(256), 0, s, (i & 1) ? p : q, (i & 1) ? q : p, big); })) return 1;
    snprintf(nm, sizeof nm, "same with 64 KB of dynamic LDS requested, %4d x 256", wgs);
    CK(hipFuncSetAttribute((const void*)k_rw, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    if (run(nm, n, s, [&](int i) { hipLaunchKernelGGL(k_rw, dim3(wgs), dim3(256), 65536, s, (i & 1) ? p : q, (i & 1) ? q : p, big); })) return 1;
  }
  // the same dependent chain replayed from a graph (400 nodes)
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_rw, dim3(256), dim3(256), 0, s, (i & 1) ? p : q, (i & 1) ? q : p, big);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a, s));
      for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
      float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
      if (ms * 1e3 / 4000 < best) best = ms * 1e3 / 4000;
    }
    printf("%-64s  timeline %5.2f us/launch\n", "graph replay of the 400-launch dependent chain (256 x 256)", best);
  }
  return 0;
}
