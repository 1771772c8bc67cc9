// ctcbench.hip — the CTC kernels alone: the lattice on probabilities (ctc.hip: ctc_ab_lin) against the log-domain recursion on
// the same logits - nll, the gradient wrt the logits (first frame that differs) and the time of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ctcbench.hip -o tools/sb_ctc
//   tools/sb_ctc [B=16] [T=500] [C=29] [Lmin=40] [Lmax=80] [sharp=1.0] [ragged=0]
#include "../neuralasr_amd/csrc/ctc.hip"
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
  float* probs = dev<float>(nl);
  int* kexp = dev<int>((size_t)B * 2 * (d.Tws / 4 + 3) * 64);
  int* fmt = dev<int>(Bp);
  CK(hipMemcpy(dseq, seq.data(), Bp * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dll, ll.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlab, lab.data(), lab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcs, cstart.data(), cstart.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcp, cpos.data(), cpos.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> grad[2], nllh[2];
  std::vector<int> fmth(Bp, -1);
  for (int mode = 0; mode < 2; ++mode) {      // 0: log domain only, 1: probabilities (+ log-domain redo where flagged)
    CtcDims dd = d;
    if (mode == 1) { dd.probs = probs; dd.kexp = kexp; dd.fmt = fmt; }
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      CK(hipMemcpyAsync(logits, lg.data(), nl * 4, hipMemcpyHostToDevice, st));
      launch_ctc_logz(dd, logits, dseq, logz, st);
      CK(hipEventRecord(e0, st));
      launch_ctc_alpha_beta(dd, logits, logz, dlab, dll, dseq, alpha, beta, aoff, boff, nll, logp, st);
      CK(hipEventRecord(e1, st));
      launch_ctc_grad(dd, logits, logz, dll, dseq, dcs, dcp, alpha, beta, aoff, boff, logp, 1.f, st);
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    grad[mode].resize(nl); nllh[mode].resize(Bp);
    CK(hipMemcpy(grad[mode].data(), logits, nl * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nllh[mode].data(), nll, Bp * 4, hipMemcpyDeviceToHost));
    if (mode == 1) CK(hipMemcpy(fmth.data(), fmt, Bp * 4, hipMemcpyDeviceToHost));
    printf("%s: alpha/beta kernel %.1f us (B %d T %d C %d KS %d)\n", mode ? "probabilities" : "log domain   ", best * 1e3, B, T, C, d.KS);
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
    printf("  utt %2d  T %3d L %3d  form %d  nll %.6f / %.6f  grad rel %.2e  worst |d| %.2e at t %d class %d\n", b, seq[b], ll[b], fmth[b],
           nllh[1][b], nllh[0][b], sqrt(num / (den > 0 ? den : 1)), worst, wt, wc);
  }
  return 0;
}
