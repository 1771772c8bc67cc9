// widebench.hip — development check + timing of the wide persistent forward recurrence (lstm_wide.hip) against the
// per-timestep kernels (lstm.hip) at DeepSpeech's width: Hp 2048, both directions, random U / gate pre-activations.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I neuralasr_amd/csrc tools/widebench.hip neuralasr_amd/csrc/lstm.hip \
//        neuralasr_amd/csrc/lstm_wide.hip -o tools/sb_wide
// Run:   tools/sb_wide [T=64] [B=32] [check=1]
#include "kernels.h"
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace nasr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static double maxdiff(const std::vector<float>& a, const std::vector<float>& b, double* ref) {
  double m = 0, r = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    m = std::fmax(m, std::fabs((double)a[i] - b[i]));
    r = std::fmax(r, std::fabs((double)b[i]));
  }
  *ref = r;
  return m;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 64, B = argc > 2 ? atoi(argv[2]) : 32, check = argc > 3 ? atoi(argv[3]) : 1;
  const int Bp = (B + 15) / 16 * 16, H = 2048, Hp = 2048, D = 2, N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  const size_t R = (size_t)T * Bp;
  if (!wide_supported(Hp, Bp)) { printf("unsupported shape\n"); return 1; }
  CK(wide_prepare());
  srand(3);
  std::vector<float> hU((size_t)D * Hp * N4), hcs((size_t)D * N4), hci((size_t)D * N4);
  for (auto& v : hU) v = (rand() / (float)RAND_MAX - 0.5f) * 0.04f;
  for (int d = 0; d < D; ++d)
    for (int c = 0; c < N4; ++c) {
      float m = 0.f;
      for (int k = 0; k < Hp; ++k) m = std::fmax(m, std::fabs(hU[((size_t)d * Hp + k) * N4 + c]));
      int e;
      std::frexp(m, &e);                       // m = f * 2^e, f in [0.5, 1)
      hcs[(size_t)d * N4 + c] = std::ldexp(1.f, 15 - e);   // m * cs in [2^14, 2^15)
      hci[(size_t)d * N4 + c] = std::ldexp(1.f, e - 15);
    }
  std::vector<float> hg(R * DN);
  for (auto& v : hg) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  std::vector<int> hs(Bp, T);
  for (int b = 0; b < Bp; ++b) hs[b] = b < B ? T - (b * 7) % (T / 2 + 1) : 1;
  float *U, *Uf, *Ub, *cs, *ci, *gates0, *gatesA, *gatesB, *cA, *cB, *oA, *oB, *hst, *part;
  void *Uw, *hx;
  int* seq;
  WideCtl* ctl;
  CK(hipMalloc(&U, hU.size() * 4)); CK(hipMalloc(&Uf, hU.size() * 4)); CK(hipMalloc(&Ub, hU.size() * 4));
  CK(hipMalloc(&cs, hcs.size() * 4)); CK(hipMalloc(&ci, hci.size() * 4));
  CK(hipMalloc(&gates0, hg.size() * 4)); CK(hipMalloc(&gatesA, hg.size() * 4)); CK(hipMalloc(&gatesB, hg.size() * 4));
  CK(hipMalloc(&cA, R * DH * 4)); CK(hipMalloc(&cB, R * DH * 4)); CK(hipMalloc(&oA, R * DH * 4)); CK(hipMalloc(&oB, R * DH * 4));
  CK(hipMalloc(&hst, (size_t)2 * D * Bp * Hp * 4));
  CK(hipMalloc(&Uw, D * wide_image_bytes(Hp))); CK(hipMalloc(&hx, wide_hx_bytes(Bp))); CK(hipMalloc(&part, wide_part_bytes(Bp)));
  CK(hipMalloc(&seq, Bp * 4)); CK(hipMalloc(&ctl, sizeof(WideCtl)));
  CK(hipMemcpy(U, hU.data(), hU.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(cs, hcs.data(), hcs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ci, hci.data(), hci.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(gates0, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(seq, hs.data(), Bp * 4, hipMemcpyHostToDevice));
  CK(hipMemset(hx, 0, wide_hx_bytes(Bp))); CK(hipMemset(part, 0, wide_part_bytes(Bp)));
  hipStream_t st; CK(hipStreamCreate(&st));
  LstmDims dm{T, B, Bp, H, Hp, D};
  for (int d = 0; d < D; ++d) {
    launch_repack_u(U + (size_t)d * Hp * N4, Uf + (size_t)d * Hp * N4, Ub + (size_t)d * Hp * N4, Hp, st);
    launch_repack_wide(U + (size_t)d * Hp * N4, cs + (size_t)d * N4, (char*)Uw + d * wide_image_bytes(Hp), Hp, st);
  }
  CK(hipStreamSynchronize(st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t hsz = (size_t)D * Bp * Hp;
  float msA = 0, msB = 0;
  // ---- reference: per-timestep kernels
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemcpyAsync(gatesA, gates0, hg.size() * 4, hipMemcpyDeviceToDevice, st));
    CK(hipMemsetAsync(cA, 0, R * DH * 4, st)); CK(hipMemsetAsync(oA, 0xff, R * DH * 4, st));
    CK(hipMemsetAsync(hst, 0, 2 * hsz * 4, st));
    CK(hipEventRecord(a, st));
    for (int s = 0; s < T; ++s)
      launch_lstm_fwd_step(dm, s, Uf, hst + (s & 1) * hsz, hst + ((s + 1) & 1) * hsz, gatesA, cA, oA, seq, 1.f, st);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&msA, a, b));
  }
  // ---- wide persistent kernel, one launch per direction
  unsigned herr = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpyAsync(gatesB, gates0, hg.size() * 4, hipMemcpyDeviceToDevice, st));
    CK(hipMemsetAsync(cB, 0, R * DH * 4, st)); CK(hipMemsetAsync(oB, 0xff, R * DH * 4, st));
    CK(hipEventRecord(a, st));
    for (int d = 0; d < D; ++d) {
      launch_lstm_wide_fwd(dm, d, (char*)Uw + d * wide_image_bytes(Hp), ci + (size_t)d * N4, gatesB, cB, oB, seq, hx, part, ctl,
                           nullptr, nullptr, 1.f, st);
      if (rep == 0) {
        CK(hipStreamSynchronize(st));
        WideCtl hc;
        CK(hipMemcpy(&hc, ctl, sizeof(unsigned) * 16, hipMemcpyDeviceToHost));
        herr |= hc.error;
        printf("direction %d: error word %u, members per XCD:", d, hc.error);
        for (int i = 0; i < 8; ++i) printf(" %u", hc.xcc_count[i]);
        printf("\n");
      }
    }
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&msB, a, b));
  }
  if (getenv("WIDE_STAMPS")) {
    std::vector<unsigned> hc(sizeof(WideCtl) / 4);
    CK(hipMemcpy(hc.data(), ctl, sizeof(WideCtl), hipMemcpyDeviceToHost));
    const unsigned* st_ = hc.data() + (offsetof(WideCtl, stamps) / 4);
    const char* names[10] = {"loop-top", "h-poll", "h-load+lds", "barrier1", "mfma", "part-store+ack", "part-poll", "part-load+lds", "barrier2", "cell..barrier3"};
    printf("cycles per step (100 MHz s_memtime ticks x 24 ~ core cycles at 2.4 GHz):\n");
    for (int i = 0; i < 10; ++i) {
      printf("  %-16s", names[i]);
      for (int wv = 0; wv < 16; ++wv) printf(" %6.1f", st_[wv * 10 + i] / (double)T);
      printf("\n");
    }
  }
  printf("T %d B %d: per-step kernels %.3f ms (%.2f us/step), wide persistent %.3f ms (%.2f us per direction-step)\n", T, B, msA,
         msA * 1e3 / T, msB, msB * 1e3 / (2 * T));
  if (check) {
    std::vector<float> x(R * DH), y(R * DH);
    double ref;
    CK(hipMemcpy(x.data(), oA, x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), oB, y.size() * 4, hipMemcpyDeviceToHost));
    double m = maxdiff(y, x, &ref);
    printf("out   max |diff| %.3e (max |ref| %.3f)\n", m, ref);
    CK(hipMemcpy(x.data(), cA, x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), cB, y.size() * 4, hipMemcpyDeviceToHost));
    double m2 = maxdiff(y, x, &ref);
    printf("c     max |diff| %.3e (max |ref| %.3f)\n", m2, ref);
    std::vector<float> ga(hg.size()), gb(hg.size());
    CK(hipMemcpy(ga.data(), gatesA, ga.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), gatesB, gb.size() * 4, hipMemcpyDeviceToHost));
    double m3 = maxdiff(gb, ga, &ref);
    printf("gates max |diff| %.3e (max |ref| %.3f)\n", m3, ref);
    if (getenv("WIDE_DEBUG")) {
      long bad = 0, by_ci[8] = {0}, by_row[64] = {0}, by_x[8] = {0}, by_d[2] = {0}, by_t[8] = {0};
      for (size_t i = 0; i < ga.size(); ++i)
        if (std::fabs(ga[i] - gb[i]) > 1e-4) {
          const int col = (int)(i % DN), rr = (int)(i / DN), d_ = col / N4, u_ = (col % N4) / 4, b_ = rr % Bp, t_ = rr / Bp;
          if (bad < 12) printf("  mismatch t %d b %d d %d unit %d gate %d: ref %.5f got %.5f\n", t_, b_, d_, u_, col & 3, ga[i], gb[i]);
          ++bad; ++by_ci[u_ & 7]; ++by_row[b_]; ++by_x[u_ / 256]; ++by_d[d_]; ++by_t[t_ < 8 ? t_ : 7];
        }
      printf("  %ld mismatching gate values; by unit&7:", bad);
      for (int i = 0; i < 8; ++i) printf(" %ld", by_ci[i]);
      printf("; by slice:");
      for (int i = 0; i < 8; ++i) printf(" %ld", by_x[i]);
      printf("; by direction: %ld %ld; by frame:", by_d[0], by_d[1]);
      for (int i = 0; i < 8; ++i) printf(" %ld", by_t[i]);
      printf("; by row:");
      for (int i = 0; i < Bp; ++i) printf(" %ld", by_row[i]);
      printf("\n");
    }
    if (getenv("WIDE_DEBUG") && T >= 2) {
      // partial sums of step 1, direction 1 (the last launch), destination (0,0): units 0..7, from source slices 1..7
      const int MT = Bp / 16;
      std::vector<float> hp_((size_t)2 * 256 * 8 * MT * 2 * 256), ho(R * DH);
      CK(hipMemcpy(hp_.data(), part, hp_.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ho.data(), oA, ho.size() * 4, hipMemcpyDeviceToHost));
      const int d = 1;
      long nbad = 0, ntot = 0;
      for (int sx = 1; sx < 8; ++sx)
        for (int b = 0; b < Bp; ++b)
          for (int i = 0; i < 8; ++i)
            for (int g = 0; g < 4; ++g) {
              const int len = hs[b], fr = len - 1;             // step 0 of the bw direction is frame len-1
              double e = 0;
              for (int k = 256 * sx; k < 256 * sx + 256; ++k)
                e += (double)ho[((size_t)fr * Bp + b) * DH + d * Hp + k] * hU[((size_t)d * Hp + k) * N4 + 4 * i + g];
              const int m_ = b >> 4, h2 = i >> 2, lane = 16 * ((b & 15) >> 2) + 4 * (i & 3) + g, r = b & 3;
              const float got = hp_[((((size_t)1 * 256 + 0) * 8 + sx) * MT * 2 + (m_ * 2 + h2)) * 256 + lane * 4 + r];
              ++ntot;
              if (std::fabs(e - got) > 1e-5) {
                if (nbad < 16) printf("src %d row %d unit %d gate %d: expected %.6f got %.6f\n", sx, b, i, g, e, got);
                ++nbad;
              }
            }
      printf("  partial sums in the inbox of workgroup (0,0): %ld of %ld differ from the host sum\n", nbad, ntot);
    }
    printf(herr == 0 && m < 1e-4 && m2 < 1e-4 && m3 < 1e-4 ? "OK\n" : "MISMATCH\n");
  }
  // ================= BPTT: per-step kernels vs the wide persistent kernel, on the forward results above
  {
    std::vector<float> hrs((size_t)D * Hp), hri((size_t)D * Hp), hdo(R * DH);
    for (int d = 0; d < D; ++d)
      for (int k = 0; k < Hp; ++k) {
        float m = 0.f;
        for (int c = 0; c < N4; ++c) m = std::fmax(m, std::fabs(hU[((size_t)d * Hp + k) * N4 + c]));
        int e;
        std::frexp(m, &e);
        hrs[(size_t)d * Hp + k] = std::ldexp(1.f, 15 - e);
        hri[(size_t)d * Hp + k] = std::ldexp(1.f, e - 15);
      }
    for (auto& v : hdo) v = (rand() / (float)RAND_MAX - 0.5f) * 2e-3f;
    float *rs, *ri, *dout, *dgA, *dgB, *par, *dcs, *srow;
    void *Uwb, *px;
    const int np = lstm_bwd_partials(Hp);
    const size_t ps = (size_t)D * np * Bp * Hp;
    CK(hipMalloc(&rs, hrs.size() * 4)); CK(hipMalloc(&ri, hri.size() * 4)); CK(hipMalloc(&dout, hdo.size() * 4));
    CK(hipMalloc(&dgA, hg.size() * 4)); CK(hipMalloc(&dgB, hg.size() * 4)); CK(hipMalloc(&par, 2 * ps * 4));
    CK(hipMalloc(&dcs, 2 * hsz * 4)); CK(hipMalloc(&srow, (size_t)D * Bp * 4));
    CK(hipMalloc(&Uwb, D * wide_image_bytes(Hp))); CK(hipMalloc(&px, wide_px_bytes(Bp)));
    CK(hipMemcpy(rs, hrs.data(), hrs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ri, hri.data(), hri.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dout, hdo.data(), hdo.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(px, 0, wide_px_bytes(Bp)));
    for (int d = 0; d < D; ++d)
      launch_repack_wide_bwd(U + (size_t)d * Hp * N4, rs + (size_t)d * Hp, (char*)Uwb + d * wide_image_bytes(Hp), Hp, st);
    CK(hipStreamSynchronize(st));
    float msC = 0, msD = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(dgA, 0xff, hg.size() * 4, st));
      CK(hipMemsetAsync(par, 0, 2 * ps * 4, st)); CK(hipMemsetAsync(dcs, 0, 2 * hsz * 4, st));
      CK(hipEventRecord(a, st));
      for (int s = T - 1; s >= 0; --s) {
        const int k = T - 1 - s;
        launch_lstm_bwd_step(dm, s, Ub, par + (k & 1) * ps, par + ((k + 1) & 1) * ps, gatesA, dgA, cA, dout, dcs + (k & 1) * hsz,
                             dcs + ((k + 1) & 1) * hsz, seq, st);
      }
      CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&msC, a, b));
    }
    unsigned herr2 = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(dgB, 0xff, hg.size() * 4, st));
      CK(hipEventRecord(a, st));
      launch_wide_row_scales(dm, dout, seq, srow, st);
      for (int d = 0; d < D; ++d) {
        launch_lstm_wide_bwd(dm, d, (char*)Uwb + d * wide_image_bytes(Hp), ri + (size_t)d * Hp, srow, gatesA, dgB, cA, dout, seq, part,
                             px, ctl, nullptr, nullptr, st);
        if (rep == 0) {
          CK(hipStreamSynchronize(st));
          WideCtl hc;
          CK(hipMemcpy(&hc, ctl, sizeof(unsigned) * 16, hipMemcpyDeviceToHost));
          herr2 |= hc.error;
          printf("BPTT direction %d: error word %u\n", d, hc.error);
        }
      }
      CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&msD, a, b));
    }
    if (getenv("WIDE_STAMPS")) {
      std::vector<unsigned> hc(sizeof(WideCtl) / 4);
      CK(hipMemcpy(hc.data(), ctl, sizeof(WideCtl), hipMemcpyDeviceToHost));
      const unsigned* st_ = hc.data() + (offsetof(WideCtl, stamps) / 4);
      const char* names[10] = {"loop-top", "dG-poll+load", "dG-lds", "barrier1", "mfma+stores", "ack+flag", "px-poll", "px-load+lds", "barrier2", "cell..publish"};
      printf("BPTT cycles per step:\n");
      for (int i = 0; i < 10; ++i) {
        printf("  %-16s", names[i]);
        for (int wv = 0; wv < 16; ++wv) printf(" %6.1f", st_[wv * 10 + i] / (double)T);
        printf("\n");
      }
    }
    printf("BPTT T %d B %d: per-step kernels %.3f ms (%.2f us/step), wide persistent %.3f ms (%.2f us per direction-step)\n", T, B,
           msC, msC * 1e3 / T, msD, msD * 1e3 / (2 * T));
    if (check) {
      std::vector<float> ga(hg.size()), gb(hg.size());
      CK(hipMemcpy(ga.data(), dgA, ga.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), dgB, gb.size() * 4, hipMemcpyDeviceToHost));
      double ref, m = 0, nrm = 0, dn = 0;
      long nbad = 0;
      for (size_t i = 0; i < ga.size(); ++i) {
        const double df = (double)ga[i] - gb[i];
        if (!(std::fabs(df) <= 1e30)) { if (nbad < 5) printf("  non-finite / missing at %zu: ref %g got %g\n", i, ga[i], gb[i]); ++nbad; continue; }
        m = std::fmax(m, std::fabs(df)); nrm += (double)ga[i] * ga[i]; dn += df * df;
      }
      (void)ref;
      printf("dG   max |diff| %.3e, rel L2 %.3e, %ld bad values\n", m, std::sqrt(dn / (nrm + 1e-300)), nbad);
      printf(herr2 == 0 && nbad == 0 && std::sqrt(dn / (nrm + 1e-300)) < 1e-5 ? "BPTT OK\n" : "BPTT MISMATCH\n");
    }
  }
  return 0;
}
