// ctc.hip — tf.nn.ctc_loss (networks/tfnetwork.py:58-59) forward-backward and its gradient, and the
// greedy decoder named at networks/tfnetwork.py:62-63, as gfx950 kernels.
//
// Semantics: unnormalised time-major logits [T',B,C], blank = C-1, softmax over C
// for t < seq_len[b]; extended label l' (blanks interleaved), S = 2L+1;
//   alpha(u,t) = log y(l'_u,t) + LSE(alpha(u,t-1), alpha(u-1,t-1), [alpha(u-2,t-1) if l'_u != blank, != l'_{u-2}])
//   beta (TF convention: excludes the emission at t), log p = LSE_u(alpha+beta),
//   d nll / d logit(t,k) = y(t,k) - exp(LSE_{u: l'_u = k}(alpha+beta) - log p), zero for t >= seq_len.
//
// Kernel split: (1) logZ per (t,b) row, one wave per row; (2) alpha and beta recursions, one workgroup per
// utterance, wave 0 = alpha, wave 1 = beta, the lattice column lives in registers (KS consecutive states per
// lane), neighbours come over wave shuffles, so a timestep has no barrier and no LDS; emissions are
// prefetched four frames ahead; (3) gradient, one wave per (t,b) row, class posteriors summed in a fixed order.
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace nasr {

constexpr float NEG = -1e30f;

// wave-uniform base + 32-bit BYTE offset: the zero-extended offset is the addressing mode's own (saddr + voffset), no
// 64-bit arithmetic per access
__device__ __forceinline__ float ldf(const float* base, unsigned byteoff) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byteoff);
}
__device__ __forceinline__ void stf(float* base, unsigned byteoff, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byteoff) = v;
}

// log(e^a + e^b + e^c).  The lattice spends 144 instructions per frame on three of these: the bare v_exp_f32 / v_log_f32
// are used, without the denormal scaling __expf / __logf wrap around them - the arguments of exp are <= 0 (results
// below 2^-126 may flush to zero: they vanish against the 1.0 of the maximum's own term) and the argument of log is in
// [1, 3].
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  const float L2E = 1.44269504088896341f, LN2 = 0.69314718055994531f;
  const float s = __builtin_amdgcn_exp2f((a - m) * L2E) + __builtin_amdgcn_exp2f((b - m) * L2E) +
                  __builtin_amdgcn_exp2f((c - m) * L2E);
  return m + __builtin_amdgcn_logf(s) * LN2;
}

// lane l <- lane l-1 / l+1 as ONE DPP move (wave_shr:1 / wave_shl:1; the first / last lane gets `edge`) instead of a
// ds_bpermute round trip through the LDS crossbar: the two neighbour exchanges of a lattice step are on its dependent chain
template <bool DPP>
__device__ __forceinline__ float lane_up1(float v, float edge) {
  if constexpr (DPP) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
  } else {
    const float r = __shfl_up(v, 1);
    return (threadIdx.x & 63) == 0 ? edge : r;
  }
}
template <bool DPP>
__device__ __forceinline__ float lane_down1(float v, float edge) {
  if constexpr (DPP) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
  } else {
    const float r = __shfl_down(v, 1);
    return (threadIdx.x & 63) == 63 ? edge : r;
  }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// the same all-lanes maximum without the six ds_bpermute round trips (720 cycles on the lattice's dependent chain every
// 4 frames): four DPP steps make every row of 16 lanes uniform, four v_readlane + scalar max join the rows
__device__ __forceinline__ float wave_max_dpp(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
  const int vi = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------ (1) log partition per row
__global__ __launch_bounds__(256) void ctc_logz_kernel(const float* __restrict__ logits, const int* __restrict__ seq_len,
                                                       float* __restrict__ logz, int Tp, int B, int Bp, int C, int Cp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= Tp * Bp) return;
  const int t = row / Bp, b = row % Bp;
  if (b >= B || t >= seq_len[b]) return;
  const float* x = logits + (size_t)row * Cp;
  float m = NEG;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = wave_sum(s);
  if (lane == 0) logz[row] = m + __logf(s);
}

void launch_ctc_logz(const CtcDims& d, const float* logits, const int* seq_len, float* logz, hipStream_t st) {
  const int rows = d.Tp * d.Bp;
  hipLaunchKernelGGL(ctc_logz_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, seq_len, logz, d.Tp, d.B, d.Bp,
                     d.C, d.Cp);
}

// ------------------------------------------------------------------ (2) alpha / beta
// workspace layout: alpha[b][t][i][lane] with state s = lane*KS + i, Tws = T + 8 rows per utterance.
// The stored columns are RESCALED: alpha~(t,.) = alpha(t,.) - aoff[t], the offset (fp64, cumulative) being
// bumped by the column maximum every 4 frames.  Raw fp32 log-domain values reach |1500| at T = 500, where one
// ulp is 1.2e-4 and alpha+beta-logp (the posterior exponent) loses 3 digits; rescaled columns stay O(10).
// The time loop runs in branch-free groups of 4 frames (loads clamped, updates selected) so the emission
// gathers of the NEXT group are in flight behind counted waits while this group computes.
#ifndef NASR_CTC_GROUP
#define NASR_CTC_GROUP 4   // 8 (twice the prefetch distance, rescale every 8 frames): same time - the lattice is not waiting for its emissions
#endif
template <int KS, bool DPP>
__global__ __launch_bounds__(128) void ctc_alpha_beta_kernel(
    const float* __restrict__ logits, const float* __restrict__ logz, const int* __restrict__ labels,
    const int* __restrict__ label_len, const int* __restrict__ seq_len, float* __restrict__ alpha,
    float* __restrict__ beta, double* __restrict__ aoff, double* __restrict__ boff, float* __restrict__ nll,
    double* __restrict__ logp_out, int Bp, int Cp, int C, int Lmax, int Tws) {
  constexpr int G = NASR_CTC_GROUP;          // frames per branch-free group = prefetch distance of the emissions
  __shared__ float fin[64 * KS];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int L = label_len[b], Tb = seq_len[b], S = 2 * L + 1;
  const int blank = C - 1;
  const int* lab = labels + (size_t)b * Lmax;

  int ext[KS];
  bool act[KS], skip[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int s = lane * KS + i;
    act[i] = s < S;
    ext[i] = (act[i] && (s & 1)) ? lab[s >> 1] : blank;
  }
  // 32-bit element offsets from wave-uniform bases (the launcher checks T' * Bp * Cp < 2^32): one v_mad_u32 per gather
  // or store instead of 64-bit address arithmetic - the lattice is bound by its instruction count
  const unsigned rstride = (unsigned)Bp * (unsigned)Cp;  // logits row stride between frames
  const float* lg = logits + (size_t)b * Cp;
  const float* lz = logz + b;
  auto emit = [&](int t, float (&e)[KS]) {               // t must be a valid frame (callers clamp)
    const float z = ldf(lz, (unsigned)t * (unsigned)Bp * 4u);
    float v[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) v[i] = ldf(lg, ((unsigned)t * rstride + (unsigned)ext[i]) * 4u);   // unconditional: ext is always a class id
#pragma unroll
    for (int i = 0; i < KS; ++i) e[i] = act[i] ? v[i] - z : NEG;
  };
  // The same in two halves, for the group-ahead prefetch: the loads are issued before a group of 4 frames and turned
  // into emissions only after it.  (With `emit` hipcc computed v - z right behind the loads, i.e. waited for all 16
  // gathers of the next group at the top of every group: ~500 of the ~900 cycles a frame took.)
  auto emit_raw = [&](int t, float (&v)[KS], float& z) {
    z = ldf(lz, (unsigned)t * (unsigned)Bp * 4u);
#pragma unroll
    for (int i = 0; i < KS; ++i) v[i] = ldf(lg, ((unsigned)t * rstride + (unsigned)ext[i]) * 4u);
  };
  auto emit_finish = [&](float (&v)[KS], float z, float (&e)[KS]) {
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      float x = v[i];
      asm volatile("" : "+v"(x));          // the load result enters hipcc's view here, not earlier
      e[i] = act[i] ? x - z : NEG;
    }
  };
  float* ws = (w == 0 ? alpha : beta) + (size_t)b * Tws * KS * 64;
  double* off = (w == 0 ? aoff : boff) + (size_t)b * Tws;
  auto store = [&](int t, const float (&a)[KS], double o) {
#pragma unroll
    for (int i = 0; i < KS; ++i) stf(ws, (((unsigned)t * KS + i) * 64u + (unsigned)lane) * 4u, a[i]);
    if (lane == 0) off[t] = o;
  };
  auto renorm = [&](float (&a)[KS], double& o) {
    float m = a[0];
#pragma unroll
    for (int i = 1; i < KS; ++i) m = fmaxf(m, a[i]);
    m = DPP ? wave_max_dpp(m) : wave_max(m);
#pragma unroll
    for (int i = 0; i < KS; ++i) a[i] = a[i] > 0.5f * NEG ? a[i] - m : NEG;
    o += (double)m;
  };

  if (w == 0) {
    // ---------------- alpha, forward in time
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      const int e2 = (s >= 2 && (s & 1)) ? lab[(s >> 1) - 1] : blank;   // l'_{s-2}
      skip[i] = act[i] && s >= 2 && ext[i] != blank && ext[i] != e2;
    }
    float a[KS], e[G][KS];
    double A = 0.0;
    emit(0, e[0]);
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      a[i] = (s < 2 && act[i]) ? e[0][i] : NEG;
    }
    store(0, a, A);
#pragma unroll
    for (int k = 0; k < G; ++k) emit(min(1 + k, Tb - 1), e[k]);
    for (int t0 = 1; t0 < Tb; t0 += G) {
      float vn[G][KS], zn[G];
#pragma unroll
      for (int k = 0; k < G; ++k) emit_raw(min(t0 + G + k, Tb - 1), vn[k], zn[k]);
      renorm(a, A);
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int t = t0 + k;
        const bool live = t < Tb;
        float p1 = lane_up1<DPP>(a[KS - 1], NEG);
        float p2 = (KS >= 2) ? lane_up1<DPP>(a[KS >= 2 ? KS - 2 : 0], NEG) : __shfl_up(a[0], 2);
        if (KS == 1 && lane <= 1) p2 = NEG;
        float na[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          const float x1 = (i >= 1) ? a[i >= 1 ? i - 1 : 0] : p1;
          const float x2 = (i >= 2) ? a[i >= 2 ? i - 2 : 0] : (i == 1 ? p1 : p2);
          na[i] = e[k][i] + lse3(a[i], x1, skip[i] ? x2 : NEG);
        }
#pragma unroll
        for (int i = 0; i < KS; ++i) a[i] = live ? na[i] : a[i];   // a state past S has e = NEG: its na is ~NEG by itself
        store(t, a, A);              // rows Tb .. Tb+G-2 of the workspace take dead copies (Tws = T+8)
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < G; ++k) emit_finish(vn[k], zn[k], e[k]);
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) fin[lane * KS + i] = a[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane == 0) {
      const float x = fin[S - 1];
      const float y = S > 1 ? fin[S - 2] : NEG;
      const double lp = A + (double)lse3(x, y, NEG);
      logp_out[b] = lp;
      nll[b] = (float)(-lp);
    }
  } else {
    // ---------------- beta, backward in time (excludes the emission at t)
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      const int e2 = (s + 2 < S && (s & 1)) ? lab[(s >> 1) + 1] : blank;   // l'_{s+2}
      skip[i] = (s + 2 < S) && e2 != blank && e2 != ext[i];
    }
    float bt[KS], e[G][KS];
    double Bo = 0.0;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      bt[i] = (act[i] && (s == S - 1 || s == S - 2)) ? 0.f : NEG;
    }
    store(Tb - 1, bt, Bo);
    // e[k] holds the emission of frame (t+1) for the k-th step of a group
#pragma unroll
    for (int k = 0; k < G; ++k) emit(max(Tb - 1 - k, 0), e[k]);
    for (int t0 = Tb - 2; t0 >= 0; t0 -= G) {
      float vn[G][KS], zn[G];
#pragma unroll
      for (int k = 0; k < G; ++k) emit_raw(max(t0 - G - k + 1, 0), vn[k], zn[k]);
      renorm(bt, Bo);
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int t = t0 - k;
        const bool live = t >= 0;
        float bb[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) bb[i] = bt[i] + e[k][i];      // states past S: NEG + NEG, rescaled back to NEG every group
        float n1 = lane_down1<DPP>(bb[0], NEG);
        float n2 = (KS >= 2) ? lane_down1<DPP>(bb[KS >= 2 ? 1 : 0], NEG) : __shfl_down(bb[0], 2);
        if (KS == 1 && lane >= 62) n2 = NEG;
        float nb[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          const float x1 = (i + 1 < KS) ? bb[i + 1 < KS ? i + 1 : 0] : n1;
          const float x2 = (i + 2 < KS) ? bb[i + 2 < KS ? i + 2 : 0] : (i + 1 < KS ? n1 : n2);
          nb[i] = lse3(bb[i], x1, skip[i] ? x2 : NEG);
        }
#pragma unroll
        for (int i = 0; i < KS; ++i) bt[i] = live ? nb[i] : bt[i];
        if (live) store(t, bt, Bo);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < G; ++k) emit_finish(vn[k], zn[k], e[k]);
    }
  }
}

void launch_ctc_alpha_beta(const CtcDims& d, const float* logits, const float* logz, const int* labels,
                           const int* label_len, const int* seq_len, float* alpha, float* beta, double* aoff,
                           double* boff, float* nll, double* logp, hipStream_t st) {
  static const bool no_dpp = getenv("NASR_CTC_DPP") && getenv("NASR_CTC_DPP")[0] == '0';
#define NASR_AB(K)                                                                                                       \
  if (no_dpp)                                                                                                            \
    hipLaunchKernelGGL((ctc_alpha_beta_kernel<K, false>), dim3(d.B), dim3(128), 0, st, logits, logz, labels, label_len, \
                       seq_len, alpha, beta, aoff, boff, nll, logp, d.Bp, d.Cp, d.C, d.Lmax, d.Tws);                    \
  else                                                                                                                   \
    hipLaunchKernelGGL((ctc_alpha_beta_kernel<K, true>), dim3(d.B), dim3(128), 0, st, logits, logz, labels, label_len,  \
                       seq_len, alpha, beta, aoff, boff, nll, logp, d.Bp, d.Cp, d.C, d.Lmax, d.Tws)
  switch (d.KS) {
    case 1: NASR_AB(1); break;
    case 2: NASR_AB(2); break;
    case 3: NASR_AB(3); break;
    case 4: NASR_AB(4); break;
    case 5: NASR_AB(5); break;
    case 6: NASR_AB(6); break;
    case 7: NASR_AB(7); break;
    case 8: NASR_AB(8); break;
    case 9: case 10: case 11: case 12: NASR_AB(12); break;
    default: NASR_AB(16); break;
  }
#undef NASR_AB
}

// ------------------------------------------------------------------ (3) gradient, in place over the logits
// One wave per (t,b) row.  The posterior of class k is the sum of exp(alpha+beta-logp) over the lattice states that carry
// k, in a FIXED order (no atomics: two runs give the same bits): every lane turns its own states into weights in LDS;
// the blank (all even states) is a per-lane sum in state order + the wave's xor tree; label k is summed by lane k over
// the positions of k in the utterance's label in ascending order - cstart [B][C+1] / cpos [B][Lmax] is the label sorted
// by class (a counting sort the host does once per upload, nasr_api.hip).
__global__ __launch_bounds__(256) void ctc_grad_kernel(float* __restrict__ logits, const float* __restrict__ logz,
                                                       const int* __restrict__ label_len,
                                                       const int* __restrict__ seq_len,
                                                       const int* __restrict__ cstart, const int* __restrict__ cpos,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const double* __restrict__ aoff, const double* __restrict__ boff,
                                                       const double* __restrict__ logp, float scale, int Tp, int B,
                                                       int Bp, int C, int Cp, int Lmax, int KS, int Tws) {
  extern __shared__ __attribute__((aligned(16))) float wl_all[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wv;
  if (row >= Tp * Bp) return;
  const int t = row / Bp, b = row % Bp;
  float* x = logits + (size_t)row * Cp;
  if (b >= B || t >= seq_len[b]) {
    for (int c = lane; c < Cp; c += 64) x[c] = 0.f;
    return;
  }
  float* wl = wl_all + (size_t)wv * KS * 64;      // weight of state s at wl[s]
  const int L = label_len[b], S = 2 * L + 1;
  const float coff = (float)(aoff[(size_t)b * Tws + t] + boff[(size_t)b * Tws + t] - logp[b]);
  const float* al = alpha + ((size_t)b * Tws + t) * KS * 64;
  const float* be = beta + ((size_t)b * Tws + t) * KS * 64;
  float blank = 0.f;
  for (int i = 0; i < KS; ++i) {
    const int s = lane * KS + i;
    const float wgt = s < S ? __expf(al[i * 64 + lane] + be[i * 64 + lane] + coff) : 0.f;
    wl[s] = wgt;
    if (!(s & 1)) blank += wgt;
  }
  blank = wave_sum(blank);                         // fixed xor tree
  __builtin_amdgcn_s_waitcnt(0xc07f);              // this wave's LDS writes (no other wave reads them)
  const float z = logz[row];
  const int* cs = cstart + (size_t)b * (C + 1);
  const int* cp = cpos + (size_t)b * Lmax;
  for (int c = lane; c < Cp; c += 64) {
    float g = 0.f;
    if (c < C) {
      float post = blank;
      if (c < C - 1) {
        post = 0.f;
        const int j1 = cs[c + 1];
        for (int j = cs[c]; j < j1; ++j) post += wl[2 * cp[j] + 1];
      }
      g = (__expf(x[c] - z) - post) * scale;
    }
    x[c] = g;
  }
}

void launch_ctc_grad(const CtcDims& d, float* logits, const float* logz, const int* label_len, const int* seq_len,
                     const int* cstart, const int* cpos, const float* alpha, const float* beta, const double* aoff,
                     const double* boff, const double* logp, float scale, hipStream_t st) {
  const int rows = d.Tp * d.Bp;
  const int rpb = 4;
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((rows + rpb - 1) / rpb), dim3(64 * rpb), (size_t)rpb * d.KS * 64 * 4, st, logits,
                     logz, label_len, seq_len, cstart, cpos, alpha, beta, aoff, boff, logp, scale, d.Tp, d.B, d.Bp, d.C, d.Cp,
                     d.Lmax, d.KS, d.Tws);
}

// mean of n floats (n small: the batch), fixed order
__global__ void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];
    *out = s / (float)n;
  }
}
void launch_mean(const float* v, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, st, v, n, out);
}

// ------------------------------------------------------------------ greedy decode (A.6)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, const int* __restrict__ seq_len,
                                                     int* __restrict__ am, int Tp, int B, int Bp, int C, int Cp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= Tp * Bp) return;
  const int t = row / Bp, b = row % Bp;
  if (b >= B || t >= seq_len[b]) return;
  const float* x = logits + (size_t)row * Cp;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > best) { best = v; bi = c; }   // strict >: lowest index wins inside a lane
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) am[row] = bi;
}

__global__ void collapse_kernel(const int* __restrict__ am, const int* __restrict__ seq_len, int* __restrict__ ids,
                                int* __restrict__ lens, int Tp, int B, int Bp, int blank) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0, prev = -1;
  const int Tb = seq_len[b];
  for (int t = 0; t < Tb; ++t) {
    const int k = am[t * Bp + b];
    if (k != prev && k != blank) ids[(size_t)b * Tp + n++] = k;
    prev = k;
  }
  lens[b] = n;
}

void launch_greedy(const CtcDims& d, const float* logits, const int* seq_len, int* argmax_ws, int* ids, int* lens,
                   hipStream_t st) {
  const int rows = d.Tp * d.Bp;
  hipLaunchKernelGGL(argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, seq_len, argmax_ws, d.Tp, d.B, d.Bp,
                     d.C, d.Cp);
  hipLaunchKernelGGL(collapse_kernel, dim3((d.B + 63) / 64), dim3(64), 0, st, argmax_ws, seq_len, ids, lens, d.Tp, d.B,
                     d.Bp, d.C - 1);
}

}  // namespace nasr
