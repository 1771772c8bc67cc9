// ctcbench.hip — the CTC kernels alone: the engineered lattice (ctc.hip (2b)) against the plain one of (2) on the same logits -
// nll, the gradient wrt the logits (largest difference and where) and the time of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ctcbench.hip -o tools/sb_ctc
//   tools/sb_ctc [B=16] [T=500] [C=29] [Lmin=40] [Lmax=80] [sharp=1.0] [ragged=0]
#include "../neuralasr_amd/csrc/ctc.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace nasr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <typename T> static T* dev(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0xff, n * sizeof(T))); return p; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, T = argc > 2 ? atoi(argv[2]) : 500, C = argc > 3 ? atoi(argv[3]) : 29;
  const int Lmin = argc > 4 ? atoi(argv[4]) : 40, Lmax = argc > 5 ? atoi(argv[5]) : 80;
  const float sharp = argc > 6 ? atof(argv[6]) : 1.f;
  const int ragged = argc > 7 ? atoi(argv[7]) : 0;
  const int Bp = (B + 15) / 16 * 16, Cp = (C + 31) / 32 * 32;
  srand(7);
  std::vector<int> seq(Bp, 0), ll(B), lab((size_t)B * Lmax, 0);
  for (int b = 0; b < B; ++b) {
    seq[b] = ragged ? T / 2 + rand() % (T / 2 + 1) : T;
    if (b == B - 1) seq[b] = T;
    ll[b] = Lmin + rand() % (Lmax - Lmin + 1);
    if (ll[b] > seq[b] / 2) ll[b] = seq[b] / 2 > 0 ? seq[b] / 2 : 0;
    for (int i = 0; i < ll[b]; ++i) lab[(size_t)b * Lmax + i] = rand() % (C - 1);
  }
  std::vector<float> lg((size_t)T * Bp * Cp, 0.f);
  for (auto& v : lg) v = sharp * (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  // class-sorted label positions (what the host side of the library builds per upload)
  std::vector<int> cstart((size_t)B * (C + 1), 0), cpos((size_t)B * Lmax, 0);
  for (int b = 0; b < B; ++b) {
    int* cs = &cstart[(size_t)b * (C + 1)];
    for (int i = 0; i < ll[b]; ++i) cs[lab[(size_t)b * Lmax + i] + 1]++;
    for (int c = 0; c < C; ++c) cs[c + 1] += cs[c];
    std::vector<int> fill(cs, cs + C);
    for (int i = 0; i < ll[b]; ++i) cpos[(size_t)b * Lmax + fill[lab[(size_t)b * Lmax + i]]++] = i;
  }
  CtcDims d{};
  d.Tp = T; d.B = B; d.Bp = Bp; d.C = C; d.Cp = Cp; d.Lmax = Lmax;
  const int KS = std::max(1, (2 * Lmax + 1 + 63) / 64);
  d.KS = KS <= 1 ? 2 : KS <= 8 ? KS : (KS <= 12 ? 12 : 16);
  d.Tws = T + 8;
  const size_t nl = (size_t)T * Bp * Cp, nws = (size_t)B * d.Tws * d.KS * 64;
  float *logits = dev<float>(nl), *logz = dev<float>((size_t)T * Bp), *alpha = dev<float>(nws), *beta = dev<float>(nws), *nll = dev<float>(Bp);
  double *aoff = dev<double>((size_t)B * d.Tws), *boff = dev<double>((size_t)B * d.Tws), *logp = dev<double>(Bp);
  int *dseq = dev<int>(Bp), *dll = dev<int>(B), *dlab = dev<int>(lab.size()), *dcs = dev<int>(cstart.size()), *dcp = dev<int>(cpos.size());
  float* lprobs = dev<float>(nl);
  double* goff = dev<double>((size_t)B * 2 * (d.Tws / 4 + 3));
  CK(hipMemcpy(dseq, seq.data(), Bp * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dll, ll.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlab, lab.data(), lab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcs, cstart.data(), cstart.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcp, cpos.data(), cpos.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> grad[2], nllh[2];
  for (int mode = 0; mode < 2; ++mode) {      // 0: the plain lattice, 1: the engineered one
    CtcDims dd = d;
    if (mode == 1) { dd.lprobs = lprobs; dd.goff = goff; }
    float best = 1e9f;
    std::vector<float> prev(nl), cur(nl);
    int unequal = 0;
    for (int it = 0; it < 6; ++it) {
      CK(hipMemcpyAsync(logits, lg.data(), nl * 4, hipMemcpyHostToDevice, st));
      launch_ctc_logz(dd, logits, dseq, logz, st);
      CK(hipEventRecord(e0, st));
      launch_ctc_alpha_beta(dd, logits, logz, dlab, dll, dseq, alpha, beta, aoff, boff, nll, logp, st);
      CK(hipEventRecord(e1, st));
      launch_ctc_grad(dd, logits, logz, dll, dseq, dcs, dcp, alpha, beta, aoff, boff, logp, 1.f, st);
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      CK(hipMemcpy(cur.data(), logits, nl * 4, hipMemcpyDeviceToHost));
      if (it > 0 && memcmp(cur.data(), prev.data(), nl * 4) != 0) ++unequal;
      prev.swap(cur);
    }
    if (unequal) printf("  !! %d of 5 repeats gave other gradient bits\n", unequal);
    grad[mode].resize(nl); nllh[mode].resize(Bp);
    CK(hipMemcpy(grad[mode].data(), logits, nl * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nllh[mode].data(), nll, Bp * 4, hipMemcpyDeviceToHost));
    printf("%s: alpha/beta kernel %.1f us (B %d T %d C %d KS %d)\n", mode ? "engineered" : "plain     ", best * 1e3, B, T, C, d.KS);
  }
  // fp64 reference on the host: log-domain alpha / beta, TF's conventions (ctc.hip header)
  std::vector<double> gref(nl, 0.0), nllref(B, 0.0);
  for (int b = 0; b < B; ++b) {
    const int Tb = seq[b], L = ll[b], S = 2 * L + 1;
    auto ext = [&](int s) { return (s & 1) ? lab[(size_t)b * Lmax + (s >> 1)] : C - 1; };
    auto lse = [](double x, double y) { if (x < y) std::swap(x, y); return y < -1e29 ? x : x + log1p(exp(y - x)); };
    std::vector<double> lp((size_t)Tb * C), al((size_t)Tb * S, -1e30), be((size_t)Tb * S, -1e30);
    for (int t = 0; t < Tb; ++t) {
      const float* x = &lg[((size_t)t * Bp + b) * Cp];
      double m = -1e300; for (int c = 0; c < C; ++c) m = std::max(m, (double)x[c]);
      double z = 0; for (int c = 0; c < C; ++c) z += exp(x[c] - m);
      z = m + log(z);
      for (int c = 0; c < C; ++c) lp[(size_t)t * C + c] = x[c] - z;
    }
    al[0] = lp[ext(0)]; if (S > 1) al[1] = lp[ext(1)];
    for (int t = 1; t < Tb; ++t)
      for (int s = 0; s < S; ++s) {
        double v = al[(size_t)(t - 1) * S + s];
        if (s >= 1) v = lse(v, al[(size_t)(t - 1) * S + s - 1]);
        if (s >= 2 && (s & 1) && ext(s) != ext(s - 2)) v = lse(v, al[(size_t)(t - 1) * S + s - 2]);
        al[(size_t)t * S + s] = v < -1e29 ? -1e30 : v + lp[(size_t)t * C + ext(s)];
      }
    be[(size_t)(Tb - 1) * S + S - 1] = 0; if (S > 1) be[(size_t)(Tb - 1) * S + S - 2] = 0;
    for (int t = Tb - 2; t >= 0; --t)
      for (int s = 0; s < S; ++s) {
        auto term = [&](int u) { const double q = be[(size_t)(t + 1) * S + u]; return q < -1e29 ? -1e30 : q + lp[(size_t)(t + 1) * C + ext(u)]; };
        double v = term(s);
        if (s + 1 < S) v = lse(v, term(s + 1));
        if (s + 2 < S && (s & 1) && ext(s + 2) != ext(s)) v = lse(v, term(s + 2));
        be[(size_t)t * S + s] = v;
      }
    double logp = al[(size_t)(Tb - 1) * S + S - 1];
    if (S > 1) logp = lse(logp, al[(size_t)(Tb - 1) * S + S - 2]);
    nllref[b] = -logp;
    for (int t = 0; t < Tb; ++t) {
      std::vector<double> post(C, 0.0);
      for (int s = 0; s < S; ++s) post[ext(s)] += exp(al[(size_t)t * S + s] + be[(size_t)t * S + s] - logp);
      for (int c = 0; c < C; ++c) gref[((size_t)t * Bp + b) * Cp + c] = exp(lp[(size_t)t * C + c]) - post[c];
    }
  }
  for (int mode = 0; mode < 2; ++mode) {
    double num = 0, den = 0, nw = 0;
    for (int b = 0; b < B; ++b) {
      nw = std::max(nw, fabs(nllh[mode][b] - nllref[b]) / std::max(1.0, fabs(nllref[b])));
      for (int t = 0; t < seq[b]; ++t)
        for (int c = 0; c < C; ++c) {
          const size_t i = ((size_t)t * Bp + b) * Cp + c;
          const double df = grad[mode][i] - gref[i];
          num += df * df; den += gref[i] * gref[i];
        }
    }
    printf("%s vs fp64: gradient rel %.2e, worst nll rel %.2e\n", mode ? "engineered" : "plain     ", sqrt(num / (den > 0 ? den : 1)), nw);
  }
  for (int b = 0; b < B; ++b) {
    double num = 0, den = 0, worst = 0; int wt = -1, wc = -1;
    for (int t = 0; t < seq[b]; ++t)
      for (int c = 0; c < C; ++c) {
        const size_t i = ((size_t)t * Bp + b) * Cp + c;
        const double df = grad[1][i] - grad[0][i];
        num += df * df; den += (double)grad[0][i] * grad[0][i];
        if (fabs(df) > worst) { worst = fabs(df); wt = t; wc = c; }
      }
    printf("  utt %2d  T %3d L %3d  nll %.6f / %.6f  grad rel %.2e  worst |d| %.2e at t %d class %d\n", b, seq[b], ll[b], nllh[1][b],
           nllh[0][b], sqrt(num / (den > 0 ? den : 1)), worst, wt, wc);
  }
  return 0;
}
