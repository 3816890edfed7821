// Sandbox for the GEMM kernel: per-phase shader-clock breakdown (wave 0 of every workgroup) for one problem.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -I include -DVPTR_GEMM_TIMING
//         tools/gemm_probe.hip vptr_amd/csrc/api.hip -o gpurun_out/gemm_probe ;  gemm_probe M N K amode bmode split_k
#include "../vptr_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 10240, N = argc > 2 ? atoi(argv[2]) : 528, K = argc > 3 ? atoi(argv[3]) : 528;
  int am = argc > 4 ? atoi(argv[4]) : 0, bm = argc > 5 ? atoi(argv[5]) : 0, sk = argc > 6 ? atoi(argv[6]) : 1;
  g_gemm_variant = argc > 7 ? atoi(argv[7]) : 1;
  float *A, *B, *D;
  size_t na = (size_t)M * K, nb = (size_t)N * K, nd = (size_t)M * N;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&D, nd * 4);
  std::vector<float> h(na > nb ? na : nb);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
  hipMemset(D, 0, nd * 4);
  long long* tb; const int maxblk = 1 << 16;
  hipMalloc(&tb, maxblk * 8 * sizeof(long long)); hipMemset(tb, 0, maxblk * 8 * sizeof(long long));
  hipMemcpyToSymbol(HIP_SYMBOL(vptr_gemm_timing_buf), &tb, sizeof(tb));
  vptr_gemm_desc d = {};
  d.A = A; d.B = B; d.D = D; d.M = M; d.N = N; d.K = K; d.a_mode = am; d.b_mode = bm; d.precision = 3; d.split_k = sk; d.atomic = sk > 1;
  d.lda = am == 0 ? K : M; d.ldb = bm == 0 ? K : N; d.ldd = N; d.alpha = 1.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (vptr_gemm(&d, nullptr)) { printf("error: %s\n", vptr_last_error()); return 1; }
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) vptr_gemm(&d, nullptr); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d  M %d N %d K %d a%d b%d split %d: %.1f us  %.1f TF/s\n", g_gemm_variant, M, N, K, am, bm, sk, ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12);
  {  // sampled correctness check against fp64 host dot products
    hipMemset(D, 0, nd * 4);
    vptr_gemm(&d, nullptr);
    hipDeviceSynchronize();
    std::vector<float> hd(nd);
    hipMemcpy(hd.data(), D, nd * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int sidx = 0; sidx < 256; ++sidx) {
      const int m = (int)((sidx * 2654435761ull + 12345) % M), n = (int)((sidx * 40503ull + 977) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) {
        const double a = am == 0 ? h[(size_t)m * K + k] : h[(size_t)k * M + m];
        const double b = bm == 0 ? h[(size_t)n * K + k] : h[(size_t)k * N + n];
        ref += a * b;
      }
      const double e = fabs(hd[(size_t)m * N + n] - ref) / (fabs(ref) + 0.3 * sqrt((double)K) * 0.29 * 0.29);  // |a|,|b| ~ U(-.5,.5)
      if (e > worst) worst = e;
    }
    printf("sampled max error vs fp64: %.2e %s\n", worst, worst < 1e-4 ? "(ok)" : "(MISMATCH)");
  }
  std::vector<long long> t(maxblk * 8);
  hipMemcpy(t.data(), tb, t.size() * 8, hipMemcpyDeviceToHost);
  int nblk = 0; while (nblk < maxblk && t[nblk * 8 + 7]) ++nblk;
  double s[6] = {0}; long long tmin = t[6], tmax = t[7]; double dur = 0;
  for (int b = 0; b < nblk; ++b) { for (int q = 0; q < 6; ++q) s[q] += t[b * 8 + q]; dur += t[b * 8 + 7] - t[b * 8 + 6];
    if (t[b * 8 + 6] < tmin) tmin = t[b * 8 + 6]; if (t[b * 8 + 7] > tmax) tmax = t[b * 8 + 7]; }
  const char* nm[6] = {"prologue(load issue)", "prologue cvt+store", "barriers", "load issue", "MFMA + cvt/store", "epilogue"};
  printf("blocks %d  avg block cycles %.0f  (clock64 ticks; span first-start..last-end %lld)\n", nblk, dur / nblk, tmax - tmin);
  for (int q = 0; q < 6; ++q) printf("  %-30s %9.0f ticks/block  %5.1f %%\n", nm[q], s[q] / nblk, 100.0 * s[q] / dur);
  return 0;
}
