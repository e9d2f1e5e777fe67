// Standalone timing harness for the LDS-DMA implicit-GEMM kernel (development aid, not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DABL_NOLOAD|-DABL_NOMFMA] tools/bench_igemm.hip -o /tmp/bench_igemm
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../ishapediting_amd/csrc/igemm2.hip"
#include "../ishapediting_amd/csrc/igemm4.hip"
#include "../ishapediting_amd/csrc/igemm_skinny.hip"
hipEvent_t g_igemm_prof_start = nullptr, g_igemm_prof_stop = nullptr;
#ifdef IG_STAMPS
__device__ unsigned long long* g_ig_stamps;
#include <algorithm>
#endif
void ishap_set_error(const std::string& m) { fprintf(stderr, "ERR %s\n", m.c_str()); }

__global__ void empty_kernel(int* p) { if (p) *p = 0; }
// floor of an epilogue: every workgroup writes `per_wg` bytes as 16-byte stores, nothing else
__global__ void write_kernel(half8* out, int per_wg) {
  half8 v = {1, 2, 3, 4, 5, 6, 7, 8};
  half8* o = out + (size_t)blockIdx.x * (per_wg / 16);
  for (int c = threadIdx.x; c < per_wg / 16; c += blockDim.x) o[c] = v;
}

int main(int argc, char** argv) {
  int H = argc > 1 ? atoi(argv[1]) : 128, Cin = argc > 2 ? atoi(argv[2]) : 256, Cout = argc > 3 ? atoi(argv[3]) : 256;
  int big = argc > 4 ? atoi(argv[4]) : 1, ksplit = argc > 5 ? atoi(argv[5]) : 1, gen = argc > 6 ? atoi(argv[6]) : 2;
  int ksize = argc > 7 ? atoi(argv[7]) : 3;
  int stats = argc > 8 ? atoi(argv[8]) : 0;      // 1: accumulate the per-channel GroupNorm statistics in the epilogue
  int nbuf = argc > 9 ? atoi(argv[9]) : 1;       // weight copies cycled through (> 256 MB in total = HBM-cold weights, as in the network)
  int k2 = argc > 10 ? atoi(argv[10]) : 0;       // channels of a folded 1x1 second source (3x3 launches): K = 9 Cin + k2
  int M = H * H, K = ksize * ksize * Cin + k2;
#ifdef IG_STAMPS
  const int nwg_st = 8192;                         // before ANY launch: the stamped kernels write through this pointer
  unsigned long long* sb; hipMalloc(&sb, (size_t)nwg_st * 16 * 8); hipMemset(sb, 0, (size_t)nwg_st * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_ig_stamps), &sb, sizeof(sb));
  if (M / 64 * ((Cout + 63) / 64) * ksplit > nwg_st) { printf("too many workgroups for the stamp buffer\n"); return 1; }
#endif
  half_t *X, *W, *O; float* ws;
  const size_t wel = (size_t)((Cout + 127) / 128 * 128) * K;
  hipMalloc(&X, (size_t)M * Cin * 2); hipMalloc(&W, wel * 2 * nbuf); hipMalloc(&O, (size_t)M * Cout * 2);
  hipMalloc(&ws, (size_t)ksplit * M * Cout * 4);
  std::vector<half_t> hx((size_t)M * Cin), hw((size_t)((Cout + 127) / 128 * 128) * K);
  for (auto& v : hx) v = (half_t)((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hw) v = (half_t)((rand() % 2001 - 1000) / 20000.f);
  hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  for (int b = 0; b < nbuf; ++b) hipMemcpy(W + b * wel, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  IgemmArgs a;
  a.X = X; a.Wt = W; a.out = O; a.M = M; a.N = Cout; a.K = K; a.conv3 = ksize == 3; a.Cin = Cin; a.ldx = Cin; a.ldw = K; a.ldo = Cout;
  a.H = H; a.W = H; a.ksplit = ksplit; a.ws = ws;
  if (k2) {
    half_t* X2; hipMalloc(&X2, (size_t)M * k2 * 2);
    std::vector<half_t> hx2((size_t)M * k2);
    for (auto& v : hx2) v = (half_t)((rand() % 2001 - 1000) / 1000.f);
    hipMemcpy(X2, hx2.data(), hx2.size() * 2, hipMemcpyHostToDevice);
    a.X2 = X2; a.ldx2 = k2; a.K2 = k2;
  }
  long long* st = nullptr;
  if (stats) { hipMalloc(&st, (size_t)Cout * 2 * 8 * 64); hipMemset(st, 0, (size_t)Cout * 2 * 8 * 64); a.stat_out = st; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int mt = ksplit;                          // gen 4 (skinny kernel): argument 5 is MT (pixels per workgroup / 16)
  if (gen == 4) { a.ksplit = 1; }
  int turn = 0;
  auto run = [&]() {
    a.Wt = W + (size_t)(turn++ % nbuf) * wel;
    if (gen == 4) igemm_skinny_launch(a, mt, 0);
    else if (gen == 6 && igemm4_applicable(a, big)) igemm4_launch_main(a, big, 0);
    else igemm2_launch_main(a, big, 0);
  };
  if (gen == 6 && !igemm4_applicable(a, big)) { printf("gen6: shape not applicable\n"); return 1; }
  if (gen == 4 || gen == 6) {         // check against the tiled kernel
    half_t* O1; hipMalloc(&O1, (size_t)M * Cout * 2);
    IgemmArgs b = a; b.out = O1; b.stat_out = nullptr; b.ksplit = 1;
    igemm2_launch_main(b, 0, 0);
    run();
    std::vector<half_t> o((size_t)M * Cout), o1((size_t)M * Cout);
    hipMemcpy(o.data(), O, o.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), O1, o1.size() * 2, hipMemcpyDeviceToHost);
    if (gen == 6 && ksplit > 1) {   // partial tiles: add the slices up on the host
      std::vector<float> sl((size_t)ksplit * M * Cout);
      hipMemcpy(sl.data(), ws, sl.size() * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < o.size(); ++i) { float v = 0; for (int z = 0; z < ksplit; ++z) v += sl[(size_t)z * M * Cout + i]; o[i] = (half_t)v; }
    }
    double md = 0, mx = 0;
    for (size_t i = 0; i < o.size(); ++i) { md = fmax(md, fabs((double)o[i] - (double)o1[i])); mx = fmax(mx, fabs((double)o1[i])); }
    printf("one-launch vs tiled: max |diff| %.4g of max |out| %.4g\n", md, mx);
  }
  for (int i = 0; i < 5; ++i) run();
  hipDeviceSynchronize();
  const int it = 50;
  if (getenv("BI_ALT")) {
    // instruction-cache probe: the same launch back to back vs alternated with a launch of a different kernel (the
    // other tile size on a small problem of its own): pair time minus the two single times = what the code switch costs
    IgemmArgs b = a;
    const int Hb = 32, Mb = Hb * Hb;
    half_t *Xb, *Ob; hipMalloc(&Xb, (size_t)Mb * Cin * 2); hipMalloc(&Ob, (size_t)Mb * Cout * 2);
    hipMemset(Xb, 0, (size_t)Mb * Cin * 2);
    b.X = Xb; b.out = Ob; b.M = Mb; b.H = Hb; b.W = Hb; b.stat_out = nullptr; b.ksplit = 1;
    auto time_it = [&](int mode) {
      for (int i = 0; i < 5; ++i) { if (mode != 1) run(); if (mode != 0) igemm2_launch_main(b, !big, 0); }
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int i = 0; i < it; ++i) { if (mode != 1) run(); if (mode != 0) igemm2_launch_main(b, !big, 0); }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      return t * 1e3 / it;
    };
    const double ta = time_it(0), tb = time_it(1), tab = time_it(2);
    printf("alone A %.2f us, alone B %.2f us, alternated A+B %.2f us: switch cost %.2f us per pair\n", ta, tb, tab, tab - ta - tb);
  }
  if (getenv("BI_EMPTY")) {                       // floor of a dependent launch in this harness
    const int lds = atoi(getenv("BI_EMPTY"));
    hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEventRecord(e0, 0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), lds, 0, (int*)nullptr);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms0; hipEventElapsedTime(&ms0, e0, e1);
    printf("empty kernel, 256 x 512 threads, %d B LDS: %.2f us per launch\n", lds, ms0 * 1e3 / it);
    for (int per = 8192; per <= 65536; per *= 2) {
      half8* buf; hipMalloc(&buf, (size_t)256 * per);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(write_kernel, dim3(256), dim3(512), 0, 0, buf, per);
      hipEventRecord(e0, 0);
      for (int i = 0; i < it; ++i) hipLaunchKernelGGL(write_kernel, dim3(256), dim3(512), 0, 0, buf, per);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms0, e0, e1);
      printf("write kernel, 256 WGs x %d B: %.2f us per launch\n", per, ms0 * 1e3 / it);
      hipFree(buf);
    }
  }
#ifdef IG_STAMPS
  {
    // in-kernel timeline: per workgroup, s_memtime at fixed points of one wave; median over workgroups of the differences
    const int nwg = nwg_st;
    hipDeviceSynchronize();
    hipMemset(sb, 0, (size_t)nwg * 16 * 8);
    for (int i = 0; i < 3; ++i) run();
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * 16);
    hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[11] = {"entry", "setup done", "step 0 landed (fill)", "K loop done", "team merge done", "epilogue entry",
                             "acc parked in LDS, operands issued", "rows finished + stored", "statistics pass / butterfly", "barrier", "final sums + atomics issued"};
    std::vector<double> d[11];
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < nwg; ++w) {
      const unsigned long long* r = &h[(size_t)w * 16];
      if (!r[0]) continue;
      tmin = std::min(tmin, r[0]);
      for (int k = 1; k < 11; ++k) if (r[k] && r[k] >= r[k - 1]) d[k].push_back((double)(r[k] - r[k - 1]));
      for (int k = 0; k < 11; ++k) tmax = std::max(tmax, r[k]);
    }
    printf("in-kernel timeline (shader cycles, median over workgroups; 100 cycles ~ 0.042-0.05 us):\n");
    double tot = 0;
    for (int k = 1; k < 11; ++k) {
      if (d[k].empty()) continue;
      std::sort(d[k].begin(), d[k].end());
      const double m = d[k][d[k].size() / 2];
      tot += m;
      printf("  %-28s +%7.0f   (p10 %7.0f  p90 %7.0f)\n", names[k], m, d[k][d[k].size() / 10], d[k][d[k].size() * 9 / 10]);
    }
    printf("  sum of medians %.0f cycles; first entry -> last stamp of any workgroup %llu cycles\n", tot, tmax - tmin);
  }
#endif
  hipEventRecord(e0, 0);
  for (int i = 0; i < it; ++i) run();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / it, tf = 2.0 * M * Cout * K / (us * 1e-6) / 1e12;
  printf("gen%d H=%d Cin=%d Cout=%d big=%d ksplit=%d stats=%d : %.2f us  %.1f TFLOP/s\n", gen, H, Cin, Cout, big, ksplit, stats, us, tf);
  return 0;
}
