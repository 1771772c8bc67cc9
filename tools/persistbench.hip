// persistbench.hip — the persistent recurrence (lstm_persist.hip) against the per-step kernels (lstm.hip) on the same
// random layer: max-abs differences of every buffer both write, and time per step of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/persistbench.hip neuralasr_amd/csrc/lstm.hip neuralasr_amd/csrc/lstm_persist.hip neuralasr_amd/csrc/optim.hip -o tools/sb_persist
//   tools/sb_persist [H=500] [B=16] [T=500] [D=2] [ragged=1]
#include "../neuralasr_amd/csrc/kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace nasr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static unsigned rs = 12345u;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xffff) / 65536.f - 0.5f; }

static double maxdiff(const float* a, const float* b, size_t n, double* maxabs) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double m = 0, ma = 0;
  for (size_t i = 0; i < n; ++i) {
    const double d = std::fabs((double)ha[i] - hb[i]);
    if (!(d <= m)) m = d;           // NaN-propagating
    if (std::fabs(ha[i]) > ma) ma = std::fabs(ha[i]);
  }
  *maxabs = ma;
  return m;
}

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 500, B = argc > 2 ? atoi(argv[2]) : 16, T = argc > 3 ? atoi(argv[3]) : 500;
  const int D = argc > 4 ? atoi(argv[4]) : 2, ragged = argc > 5 ? atoi(argv[5]) : 1;
  const int f16 = argc > 6 ? atoi(argv[6]) : 1;      // forward recurrence on fp16 planes (the engine's default) or fp32
  const int Hp = (H + 63) / 64 * 64, Bp = (B + 15) / 16 * 16, N4 = 4 * Hp;
  const size_t R = (size_t)T * Bp;
  const LstmDims dm{T, B, Bp, H, Hp, D};
  printf("H %d (Hp %d) B %d (Bp %d) T %d D %d ragged %d  persist_supported %d\n", H, Hp, B, Bp, T, D, ragged, (int)persist_supported(Hp));
  if (!persist_supported(Hp)) return 0;
  CK(persist_prepare());

  // canonical U per direction: zero rows/cols for padded units
  std::vector<float> hU((size_t)D * Hp * N4, 0.f);
  const float lim = std::sqrt(6.f / (2 * H + 4 * H));
  for (int d = 0; d < D; ++d)
    for (int k = 0; k < H; ++k)
      for (int j = 0; j < H; ++j)
        for (int g = 0; g < 4; ++g) hU[((size_t)d * Hp + k) * N4 + 4 * j + g] = 2.f * lim * frand();
  std::vector<float> hG(R * D * N4, 0.f), hDo(R * D * Hp, 0.f);
  std::vector<int> hseq(Bp, 0);
  for (int b = 0; b < B; ++b) hseq[b] = ragged ? T - (int)((rs = rs * 1664525u + 1013904223u, rs >> 16) % (T / 2 + 1)) : T;
  if (ragged && B > 1) hseq[1] = T;
  for (size_t r = 0; r < R; ++r)
    for (int d = 0; d < D; ++d)
      for (int j = 0; j < H; ++j) {
        for (int g = 0; g < 4; ++g) hG[(r * D + d) * N4 + 4 * j + g] = 3.f * frand();
        hDo[(r * D + d) * Hp + j] = frand() * 0.1f;
      }
  float *U, *Uf, *Ub, *Upf, *Upb, *gates0, *gatesA, *gatesB, *cA, *cB, *outA, *outB, *dout, *dgA, *dgB, *hst, *par, *dcs, *xch;
  int* seq; PersistCtl* ctl;
  const size_t nU = (size_t)D * Hp * N4;
  CK(hipMalloc(&U, nU * 4)); CK(hipMalloc(&Uf, nU * 4)); CK(hipMalloc(&Ub, nU * 4));
  const size_t imf = persist_image_floats(Hp, false), imb = persist_image_floats(Hp, true);
  CK(hipMalloc(&Upf, D * imf * 4)); CK(hipMalloc(&Upb, D * imb * 4));
  for (float** p : {&gates0, &gatesA, &gatesB, &dgA, &dgB}) CK(hipMalloc(p, R * D * N4 * 4));
  for (float** p : {&cA, &cB, &outA, &outB, &dout}) CK(hipMalloc(p, R * D * Hp * 4));
  const size_t hs = (size_t)D * Bp * Hp, ps = (size_t)D * (Hp / 32) * Bp * Hp;
  CK(hipMalloc(&hst, 2 * hs * 4)); CK(hipMalloc(&par, 2 * ps * 4)); CK(hipMalloc(&dcs, 2 * hs * 4));
  CK(hipMalloc(&xch, persist_xch_floats(Hp) * 4)); CK(hipMalloc(&ctl, sizeof(PersistCtl))); CK(hipMalloc(&seq, Bp * 4));
  CK(hipMemcpy(U, hU.data(), nU * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(gates0, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dout, hDo.data(), hDo.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(seq, hseq.data(), Bp * 4, hipMemcpyHostToDevice));
  // poison everything the kernels must fully define
  for (float* p : {cA, cB, outA, outB}) CK(hipMemset(p, 0xff, R * D * Hp * 4));
  for (float* p : {dgA, dgB}) CK(hipMemset(p, 0xff, R * D * N4 * 4));
  CK(hipMemset(xch, 0xff, persist_xch_floats(Hp) * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int d = 0; d < D; ++d) {
    launch_repack_u(U + (size_t)d * Hp * N4, Uf + (size_t)d * Hp * N4, Ub + (size_t)d * Hp * N4, Hp, st);
  }
  float *cs = nullptr, *cinv = nullptr;
  if (f16) {   // column scales on the host: largest magnitude of every column of U into [2^14, 2^15)
    std::vector<float> hs((size_t)D * N4), hi((size_t)D * N4);
    for (int d = 0; d < D; ++d)
      for (int c = 0; c < N4; ++c) {
        float m = 0.f;
        for (int k = 0; k < Hp; ++k) m = fmaxf(m, fabsf(hU[((size_t)d * Hp + k) * N4 + c]));
        int e = 15;
        if (m > 0.f) frexpf(m, &e);
        hs[(size_t)d * N4 + c] = ldexpf(1.f, 15 - e);
        hi[(size_t)d * N4 + c] = ldexpf(1.f, e - 15);
      }
    CK(hipMalloc(&cs, hs.size() * 4)); CK(hipMalloc(&cinv, hi.size() * 4));
    CK(hipMemcpy(cs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cinv, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  }
  {
    int64_t offs[2] = {0, (int64_t)Hp * N4};
    launch_repack_persist(U, offs, D, Upf, Upb, Hp, cs, st);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  PersistCtl hc;

  for (int rep = 0; rep < 3; ++rep) {
    // ---- forward, per-step
    CK(hipMemcpyAsync(gatesA, gates0, R * D * N4 * 4, hipMemcpyDeviceToDevice, st));
    CK(hipMemsetAsync(hst, 0, hs * 4, st));
    CK(hipEventRecord(e0, st));
    for (int s = 0; s < T; ++s)
      launch_lstm_fwd_step(dm, s, Uf, hst + (s & 1) * hs, hst + ((s + 1) & 1) * hs, gatesA, cA, outA, seq, 1.0f, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const float fstep = ms;
    // ---- forward, persistent
    CK(hipMemcpyAsync(gatesB, gates0, R * D * N4 * 4, hipMemcpyDeviceToDevice, st));
    CK(hipEventRecord(e0, st));
    launch_lstm_persist_fwd(dm, Upf, cinv, gatesB, cB, outB, seq, xch, ctl, nullptr, nullptr, 1.0f, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipMemcpy(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost));
    printf("forward : per-step %.3f us/step   persistent %.3f us/step (%.3f ms)   error word %u  xcc", fstep * 1000 / T, ms * 1000 / T, ms, hc.error);
    for (int i = 0; i < 8; ++i) printf(" %u", hc.xcc_count[i]);
    printf("\n");
    if (hc.pad[0] | hc.pad[1]) {
      printf("   wave 0 cycles/step: loop-top %.0f  round-barrier %.0f  h load+check %.0f  mfma %.0f  barrier %.0f  cell %.0f  publish %.0f\n", hc.pad[0] / (double)T,
             hc.pad[1] / (double)T, hc.pad[2] / (double)T, hc.pad[3] / (double)T, hc.pad[4] / (double)T, hc.pad[5] / (double)T, hc.pad[6] / (double)T);
      printf("   barrier arrival after wave 0 (cycles): w1 %.0f  w2 %.0f  w3 %.0f  mem %.0f   extra hand-off attempts of wave 0 per step %.3f\n", (int)hc.pad[7] / (double)T, (int)hc.pad[8] / (double)T,
             (int)hc.pad[9] / (double)T, (int)hc.pad[10] / (double)T, hc.pad[11] / (double)T);
    }
#if defined(NASR_PSTAMP) && NASR_PSTAMP
    {   // every workgroup's wave 0: the slowest chain sets the step; a phase that waits for others (poll) is shortest there
      const char* nm[7] = {"loop-top", "round-barrier", "h-load+check", "mfma", "barrier", "cell", "publish"};
      for (int ph = 0; ph < 7; ++ph) {
        std::vector<double> v;
        for (int i = 0; i < 256; ++i) v.push_back(hc.stamps[i][ph] / (double)T);
        std::sort(v.begin(), v.end());
        printf("   fwd %-15s over 256 CUs: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f\n", nm[ph], v[0], v[25], v[128], v[230], v[255]);
      }
      int lo = 0;
      for (int i = 1; i < 256; ++i) if (hc.stamps[i][1] < hc.stamps[lo][1]) lo = i;
      printf("   fwd CU with the shortest poll (xcc %d member %d):", lo / 32, lo % 32);
      for (int ph = 0; ph < 7; ++ph) printf(" %s %.0f", nm[ph], hc.stamps[lo][ph] / (double)T);
      printf("\n");
    }
#endif
    if (hc.error) return 1;
    // ---- backward, per-step (on the per-step forward's buffers)
    CK(hipMemsetAsync(par, 0, ps * 4, st)); CK(hipMemsetAsync(dcs, 0, hs * 4, st));
    CK(hipEventRecord(e0, st));
    for (int s = T - 1; s >= 0; --s) {
      const int k = T - 1 - s;
      launch_lstm_bwd_step(dm, s, Ub, par + (k & 1) * ps, par + ((k + 1) & 1) * ps, gatesA, dgA, cA, dout, dcs + (k & 1) * hs,
                           dcs + ((k + 1) & 1) * hs, seq, st);
    }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const float bstep = ms;
    CK(hipEventRecord(e0, st));
    launch_lstm_persist_bwd(dm, Upb, gatesB, dgB, cB, dout, seq, xch, ctl, nullptr, nullptr, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipMemcpy(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost));
    printf("backward: per-step %.3f us/step   persistent %.3f us/step (%.3f ms)   error word %u\n", bstep * 1000 / T, ms * 1000 / T, ms, hc.error);
    if (hc.pad[0] | hc.pad[1]) {
      printf("   wave 0 cycles/step: tail+prefetch %.0f  round-barrier %.0f  partial loads+check+sum %.0f  cell bwd %.0f  barrier %.0f  mfma+stores %.0f  publish %.0f   extra hand-off attempts per step %.3f\n",
             hc.pad[0] / (double)T, hc.pad[1] / (double)T, hc.pad[2] / (double)T, hc.pad[3] / (double)T, hc.pad[4] / (double)T, hc.pad[5] / (double)T,
             hc.pad[6] / (double)T, hc.pad[11] / (double)T);
    }
#if defined(NASR_PSTAMP) && NASR_PSTAMP
    {
      const char* nm[7] = {"tail", "round-barrier", "loads+check+sum", "cell", "barrier", "mfma+st", "publish"};
      for (int ph = 0; ph < 7; ++ph) {
        std::vector<double> v;
        for (int i = 0; i < 256; ++i) v.push_back(hc.stamps[i][ph] / (double)T);
        std::sort(v.begin(), v.end());
        printf("   bwd %-15s over 256 CUs: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f\n", nm[ph], v[0], v[25], v[128], v[230], v[255]);
      }
      int lo = 0;
      for (int i = 1; i < 256; ++i) if (hc.stamps[i][1] < hc.stamps[lo][1]) lo = i;
      printf("   bwd CU with the shortest poll (xcc %d member %d):", lo / 32, lo % 32);
      for (int ph = 0; ph < 7; ++ph) printf(" %s %.0f", nm[ph], hc.stamps[lo][ph] / (double)T);
      printf("\n");
    }
#endif
    if (hc.error) return 1;
  }
  double ma;
  double d1 = maxdiff(outA, outB, R * D * Hp, &ma); printf("out   max|diff| %.3e (max|ref| %.3e)\n", d1, ma);
  d1 = maxdiff(gatesA, gatesB, R * D * N4, &ma);    printf("gates max|diff| %.3e (max|ref| %.3e)\n", d1, ma);
  d1 = maxdiff(dgA, dgB, R * D * N4, &ma);          printf("dG    max|diff| %.3e (max|ref| %.3e)\n", d1, ma);
  // c is only defined at valid frames: compare through out/dG (both depend on it); report it where both are finite
  {
    std::vector<float> a(R * D * Hp), b(R * D * Hp);
    CK(hipMemcpy(a.data(), cA, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), cB, b.size() * 4, hipMemcpyDeviceToHost));
    double m = 0; size_t nn = 0, mism = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      const bool fa = std::isfinite(a[i]), fb = std::isfinite(b[i]);
      if (fa != fb) ++mism;
      if (fa && fb) { ++nn; m = std::fmax(m, std::fabs((double)a[i] - b[i])); }
    }
    printf("c     max|diff| %.3e over %zu defined cells, %zu defined in only one\n", m, nn, mism);
  }
  return 0;
}
