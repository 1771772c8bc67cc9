// stepbench.hip — development microbenchmark: times T graph-captured launches of the per-timestep
// recurrence kernels with ablations (-DNASR_ABL=mask) to find where a step's time goes.
// -DSB_H=2048 -DSB_B=32 -DSB_T=100: DeepSpeech's wide layer.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNASR_ABL=<mask> -I neuralasr_amd/csrc tools/stepbench.hip -o /tmp/sb
#include "../neuralasr_amd/csrc/lstm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace nasr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
#ifndef SB_H
#define SB_H 500
#endif
#ifndef SB_B
#define SB_B 16
#endif
#ifndef SB_T
#define SB_T 500
#endif
  const int T = SB_T, B = SB_B, Bp = (SB_B + 15) / 16 * 16, H = SB_H, Hp = (SB_H + 63) / 64 * 64, D = 2, N4 = 4 * Hp;
  const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0 fwd, 1 bwd
  const size_t R = (size_t)T * Bp;
  float *Uf, *Ub, *hst, *part, *dcst, *gates, *dgbuf, *cbuf, *out, *dout; int* seq;
  CK(hipMalloc(&Uf, (size_t)D * Hp * N4 * 4)); CK(hipMalloc(&Ub, (size_t)D * Hp * N4 * 4));
  CK(hipMalloc(&hst, (size_t)2 * D * Bp * Hp * 4)); CK(hipMalloc(&part, (size_t)2 * D * (Hp / 32) * Bp * Hp * 4)); CK(hipMalloc(&dgbuf, R * D * N4 * 4));
  CK(hipMalloc(&dcst, (size_t)2 * D * Bp * Hp * 4));
  CK(hipMalloc(&gates, R * D * N4 * 4)); CK(hipMalloc(&cbuf, R * D * Hp * 4)); CK(hipMalloc(&out, R * D * Hp * 4));
  CK(hipMalloc(&dout, R * D * Hp * 4)); CK(hipMalloc(&seq, Bp * 4));
  std::vector<float> hu((size_t)D * Hp * N4);
  for (auto& v : hu) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
  CK(hipMemcpy(Uf, hu.data(), hu.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Ub, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hg(R * D * N4);
  for (auto& v : hg) v = (rand() / (float)RAND_MAX - 0.5f);
  CK(hipMemcpy(gates, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(cbuf, 0, R * D * Hp * 4)); CK(hipMemset(out, 0, R * D * Hp * 4)); CK(hipMemset(dout, 0, R * D * Hp * 4));
  CK(hipMemset(hst, 0, (size_t)2 * D * Bp * Hp * 4)); CK(hipMemset(part, 0, (size_t)2 * D * (Hp / 32) * Bp * Hp * 4)); CK(hipMemset(dcst, 0, (size_t)2 * D * Bp * Hp * 4));
  std::vector<int> hs(Bp, T);
  CK(hipMemcpy(seq, hs.data(), Bp * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  LstmDims dm{T, B, Bp, H, Hp, D};
  const size_t hsz = (size_t)D * Bp * Hp, psz = (size_t)D * (Hp / 32) * Bp * Hp;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < T; ++s) {
    if (mode == 0) launch_lstm_fwd_step(dm, s, Uf, hst + (s & 1) * hsz, hst + ((s + 1) & 1) * hsz, gates, cbuf, out, seq, 1.f, st);
    else launch_lstm_bwd_step(dm, T - 1 - s, Ub, part + (s & 1) * psz, part + ((s + 1) & 1) * psz, gates, dgbuf, cbuf, dout, dcst + (s & 1) * hsz, dcst + ((s + 1) & 1) * hsz, seq, st);
  }
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ex, st));
  CK(hipStreamSynchronize(st));
  float best = 1e9f, sum = 0;
  const int N = 10;
  for (int i = 0; i < N; ++i) {
    CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
  }
  printf("mode=%d abl=%d  us/step: best %.3f avg %.3f\n", mode, NASR_ABL, best * 1000 / T, sum / N * 1000 / T);
  return 0;
}
