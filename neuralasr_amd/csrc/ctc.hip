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

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace nasr {

constexpr float NEG = -1e30f;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

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

// the engineered lattice (2b) wants its workspaces, 32-float emission rows with a spare column to park idle states on, two
// states per lane at least and DPP (NASR_CTC_FAST=0 / NASR_CTC_DPP=0: the plain one of (2))
static bool ctc_fast_ok(const CtcDims& d) {
  static const bool off = (getenv("NASR_CTC_FAST") && getenv("NASR_CTC_FAST")[0] == '0') ||
                          (getenv("NASR_CTC_DPP") && getenv("NASR_CTC_DPP")[0] == '0');
  return !off && d.lprobs && d.goff && d.Cp == 32 && d.C <= 31 && d.KS >= 2;
}

// ------------------------------------------------------------------ (1) log partition per row
// (lprobs, or NULL: the row's emissions in the form the engineered lattice (2b) reads them too, log2 y(t,k) =
// (x - logZ) log2(e), NEG in the columns from C on)
__global__ __launch_bounds__(256) void ctc_logz_kernel(const float* __restrict__ logits, const int* __restrict__ seq_len,
                                                       float* __restrict__ logz, float* __restrict__ lprobs, int Tp, int B, int Bp,
                                                       int C, int Cp) {
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
  const float z = m + __logf(s);
  if (lane == 0) logz[row] = z;
  if (lprobs)
    for (int c = lane; c < Cp; c += 64) lprobs[(size_t)row * Cp + c] = c < C ? (x[c] - z) * 1.44269504088896341f : NEG;
}

void launch_ctc_logz(const CtcDims& d, const float* logits, const int* seq_len, float* logz, hipStream_t st) {
  const int rows = d.Tp * d.Bp;
  hipLaunchKernelGGL(ctc_logz_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, seq_len, logz, ctc_fast_ok(d) ? d.lprobs : nullptr,
                     d.Tp, d.B, d.Bp, d.C, d.Cp);
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
__device__ __forceinline__ void ctc_ab_log(
    const float* __restrict__ logits, const float* __restrict__ logz, const int* __restrict__ labels,
    const int* __restrict__ label_len, const int* __restrict__ seq_len, float* __restrict__ alpha,
    float* __restrict__ beta, double* __restrict__ aoff, double* __restrict__ boff, float* __restrict__ nll,
    double* __restrict__ logp_out, int Bp, int Cp, int C, int Lmax, int Tws, float* fin) {
  constexpr int G = NASR_CTC_GROUP;          // frames per branch-free group = prefetch distance of the emissions
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

// ------------------------------------------------------------------ (2b) the same recursions, engineered: the default
// What (2) above spends a frame's ~600 cycles on, at KS = 3: 12 transcendentals (each issues at a quarter of the plain rate)
// and ~100 plain instructions, of which the recursion itself needs about a third.  Here:
//  * base-2 logarithms and the SORTED three-term sum: log2(2^a + 2^b + 2^c) = mx + log2(1 + 2^(md-mx) + 2^(mn-mx)) with
//    v_max3 / v_med3 / v_min3 - the largest term is exactly 1, so two v_exp and one v_log per state instead of three and one;
//  * the emissions log2 y(t,k) as dense 32-float rows (ctc_logz_kernel writes them, NEG in the columns from C on: idle states
//    park there), brought into LDS by a THIRD wave with LDS-DMA, 32 frames of both walks per 8 instructions, a chunk ahead;
//    the walks gather their KS values per frame from that ring with ds_read_b32 a group ahead - no per-frame global gathers,
//    no "x - logZ" on the walk's instruction stream;
//  * one store per frame ([t][lane][KS], wave-uniform base + 32-bit offset) and the column offset once per GROUP of 4 frames
//    (it only changes there) instead of KS stores + a predicated fp64 store per frame with 64-bit address arithmetic;
//  * frames past the end only in the last group's code; the next group's emissions land in the registers the group after
//    will read (two groups per loop turn, roles swapped) instead of being copied.
// Numerics are those of (2): rescaled columns (alpha~ = alpha - off, off bumped by the column maximum every 4 frames, fp64
// cumulative), TF's conventions, NEG = -1e30 for "no path".  (A version on PROBABILITIES - two multiply-adds and a multiply per
// state, binary exponents per lane - was built first: 70 us against this one's, and exact on fresh nets; but the first
// label emitted at a frame where training has made it improbable moves a lane's level by 2^20 and more per FRAME, beyond any
// rescaling that is not itself per frame - from step ~10 of a training run every utterance had to be redone in the log
// domain.  Logarithms it is.)
constexpr int FAST_G = 4, FAST_CH = 32, FAST_RING = 64;

typedef float fast_f2 __attribute__((ext_vector_type(2)));
typedef float fast_f3 __attribute__((ext_vector_type(3)));
typedef float fast_f4 __attribute__((ext_vector_type(4)));
// The store of a frame's column as ONE instruction per <= 4 states, at wave-uniform base + 32-bit byte offset (the addressing
// mode's own sum: one VALU add per frame, no 64-bit math).  (s_nop: a store of more than 64 bits reads its data registers a
// cycle after it issues; hipcc keeps the next VALU write of them away from its own stores, not from these.)
template <int KS>
__device__ __forceinline__ void fast_store(float* base, unsigned voff, const float (&a)[KS]) {
#pragma unroll
  for (int j = 0; j + 4 <= KS; j += 4) {
    const fast_f4 v = {a[j], a[j + 1], a[j + 2], a[j + 3]};
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "n"(j * 4) : "memory");
  }
  constexpr int j = KS & ~3;
  if constexpr (KS % 4 == 3) {
    const fast_f3 v = {a[j], a[j + 1], a[j + 2]};
    asm volatile("global_store_dwordx3 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "n"(j * 4) : "memory");
  } else if constexpr (KS % 4 == 2) {
    const fast_f2 v = {a[j], a[j + 1]};
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" : : "v"(voff), "v"(v), "s"(base), "n"(j * 4) : "memory");
  } else if constexpr (KS % 4 == 1) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(voff), "v"(a[j]), "s"(base), "n"(j * 4) : "memory");
  }
}
// lane l <- lane l-1 / l+1 with `edge` at the wave's first / last lane, as in lane_up1 / lane_down1
__device__ __forceinline__ float fast_up(float v) { return lane_up1<true>(v, NEG); }
__device__ __forceinline__ float fast_down(float v) { return lane_down1<true>(v, NEG); }

// log2(2^a + 2^b + 2^c)
__device__ __forceinline__ float lse3_2(float a, float b, float c) {
  const float mx = __builtin_fmaxf(a, __builtin_fmaxf(b, c));                   // (v_max3_f32)
  const float md = __builtin_amdgcn_fmed3f(a, b, c);
  const float mn = __builtin_fminf(a, __builtin_fminf(b, c));                   // (v_min3_f32)
  const float s = 1.f + __builtin_amdgcn_exp2f(md - mx) + __builtin_amdgcn_exp2f(mn - mx);
  return mx + __builtin_amdgcn_logf(s);
}

template <int KS, bool fwd>
__device__ __forceinline__ void ctc_fast_walk(const int* __restrict__ labels, const int* __restrict__ label_len,
                                              const int* __restrict__ seq_len, float* __restrict__ alpha, float* __restrict__ beta,
                                              double* __restrict__ goff, float* __restrict__ nll, double* __restrict__ logp_out,
                                              const float* __restrict__ lprobs, int Bp, int C, int Lmax, int Tws, int KG, float* ring,
                                              float* fin, int* sync) {
  static_assert(KS >= 2, "two states per lane at least: the skip transition then never reaches past the neighbour lane");
  constexpr int G = FAST_G;
  constexpr double LN2 = 0.693147180559945309417;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = fwd ? 0 : 1;
  const int L = label_len[b], Tb = seq_len[b], S = 2 * L + 1;
  const int blank = C - 1;
  const int* lab = labels + (size_t)b * Lmax;
  int ext[KS];
  bool skip[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int s = lane * KS + i;
    const bool act = s < S;
    ext[i] = act ? ((s & 1) ? lab[s >> 1] : blank) : 31;      // column 31 of an emission row is NEG
    if (fwd) {
      const int e2 = (s >= 2 && (s & 1)) ? lab[(s >> 1) - 1] : blank;   // l'_{s-2}
      skip[i] = act && s >= 2 && ext[i] != blank && ext[i] != e2;
    } else {
      const int e2 = (s + 2 < S && (s & 1)) ? lab[(s >> 1) + 1] : blank;   // l'_{s+2}
      skip[i] = (s + 2 < S) && e2 != blank && e2 != ext[i];
    }
  }
  // positions p = 0 .. Tb-2 of this wave's walk: the emission frame of position p is p+1 (alpha, producing frame p+1) or
  // Tb-1-p (beta, producing frame Tb-2-p)
  const int npos = Tb - 1;
  float* rg = ring + w * (FAST_RING * 32);
  // (plain LDS loads between compiler barriers: the rows change under this wave - the loader's LDS-DMA is nothing hipcc can
  //  see - so nothing may be kept from an earlier gather, and the gathers must stay behind the wait for the loader's word.
  //  Volatile accesses would do that too, but hipcc waits for EVERY outstanding store around each of them.)
  auto read_group = [&](int g, float (&e)[G][KS]) {        // emissions of positions 4g .. 4g+3
    asm volatile("" ::: "memory");
    const float* r0 = rg + ((G * g) & (FAST_RING - 1)) * 32;
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int k = 0; k < G; ++k) e[k][i] = r0[k * 32 + ext[i]];
    asm volatile("" ::: "memory");
  };
  // the words the three waves talk through, read and written without the waits hipcc wraps around a volatile access:
  // sync[0] = chunks the loader wave has landed in the rings (both directions), sync[1 + w] = chunks this walk is done with
  const unsigned sync_lds = (unsigned)(size_t)((lds_ptr_t)sync);
  auto wait_ready = [&](int n) {
    for (unsigned spin = 0; spin < (1u << 24); ++spin) {
      int v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(sync_lds) : "memory");
      if (v >= n) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  float* ws = (fwd ? alpha : beta) + (size_t)b * Tws * KS * 64;
  double* go = goff + ((size_t)b * 2 + w) * KG;             // column offset of group g (frames 4g-3 .. 4g; group 0 = the start column)
  constexpr unsigned ROWB = 64u * KS * 4u;                  // bytes of one frame's column
  const unsigned lane_off = (unsigned)lane * (KS * 4u);
  auto store = [&](int t, const float (&a)[KS]) { fast_store<KS>(ws, lane_off + (unsigned)t * ROWB, a); };

  float a[KS];
  double A = 0.0;
  if (fwd) {
    const float* y0 = lprobs + (size_t)b * 32;              // frame 0
#pragma unroll
    for (int i = 0; i < KS; ++i) a[i] = (lane * KS + i < 2) ? y0[ext[i]] : NEG;
    store(0, a);
  } else {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      a[i] = (s < S && (s == S - 1 || s == S - 2)) ? 0.f : NEG;
    }
    store(Tb - 1, a);
  }
  if (lane == 0) go[0] = 0.0;
  // one group of 4 frames: gathers the NEXT group's emissions into en, rescales the column, then the frames (masked: the last
  // group of a walk, where frames past the end leave the state alone)
  auto group = [&](int g, float (&e)[G][KS], float (&en)[G][KS], auto maskedc) {
    constexpr bool MASKED = decltype(maskedc)::value;
    const int p0 = G * g;
    read_group(g + 1, en);
    {
      float m = a[0];
#pragma unroll
      for (int i = 1; i < KS; ++i) m = fmaxf(m, a[i]);
      m = wave_max_dpp(m);
#pragma unroll
      for (int i = 0; i < KS; ++i) a[i] = a[i] > 0.5f * NEG ? a[i] - m : NEG;
      A += (double)m;
      if (lane == 0) go[g + 1] = A;
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
      float na[KS];
      if (fwd) {
        const float p1 = fast_up(a[KS - 1]), p2 = fast_up(a[KS - 2]);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          const float x1 = (i >= 1) ? a[i >= 1 ? i - 1 : 0] : p1;
          const float x2 = (i >= 2) ? a[i >= 2 ? i - 2 : 0] : (i == 1 ? p1 : p2);
          na[i] = e[k][i] + lse3_2(a[i], x1, skip[i] ? x2 : NEG);
        }
      } else {
        float bb[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) bb[i] = a[i] + e[k][i];      // states past S: NEG + NEG, rescaled back to NEG every group
        const float n1 = fast_down(bb[0]), n2 = fast_down(bb[1]);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          const float x1 = (i + 1 < KS) ? bb[i + 1 < KS ? i + 1 : 0] : n1;
          const float x2 = (i + 2 < KS) ? bb[i + 2 < KS ? i + 2 : 0] : (i + 1 < KS ? n1 : n2);
          na[i] = lse3_2(bb[i], x1, skip[i] ? x2 : NEG);
        }
      }
      if constexpr (MASKED) {
        const bool live = p0 + k < npos;
#pragma unroll
        for (int i = 0; i < KS; ++i) a[i] = live ? na[i] : a[i];
      } else {
#pragma unroll
        for (int i = 0; i < KS; ++i) a[i] = na[i];
      }
      // (dead frames of the last group store dead copies: alpha in the rows Tb .. Tb+2 - the workspace has T+8 - beta in row 0,
      //  whose value they are)
      store(fwd ? p0 + k + 1 : max(Tb - 2 - p0 - k, 0), a);
    }
  };
  if (npos > 0) {
    const int nch = (npos + FAST_CH - 1) / FAST_CH;
    wait_ready(1);
    float e[G][KS], en[G][KS];
    read_group(0, e);
    for (int c = 0; c < nch; ++c) {
      // two groups per turn: the emission registers swap roles instead of being copied
      for (int gi = 0; gi < FAST_CH / G; gi += 2) {
        const int g = c * (FAST_CH / G) + gi;
        if (G * g >= npos) break;
        if (G * g + G <= npos) group(g, e, en, std::false_type{});
        else group(g, e, en, std::true_type{});
        if (G * (g + 1) >= npos) break;
        if (gi + 2 == FAST_CH / G && c + 1 < nch) wait_ready(c + 2);     // the next chunk's first group: normally long there
        if (G * (g + 1) + G <= npos) group(g + 1, en, e, std::false_type{});
        else group(g + 1, en, e, std::true_type{});
      }
      // every gather from this chunk's rows has returned (their values were used above): the rows may be reused
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1 offset:%2" : : "v"(sync_lds), "v"(c + 1), "n"(4 * (1 + w)) : "memory");
    }
  }
  if (fwd) {
#pragma unroll
    for (int i = 0; i < KS; ++i) fin[lane * KS + i] = a[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane == 0) {
      const float x = fin[S - 1];
      const float y = S > 1 ? fin[S - 2] : NEG;
      const double lp = (A + (double)lse3_2(x, y, NEG)) * LN2;
      logp_out[b] = lp;
      nll[b] = (float)(-lp);
    }
  }
}

// wave 2: the emissions' way into LDS.  Chunk n (32 positions of both walks) goes to the ring half chunk n-2 was in, once both
// walks are done with that one; its LDS-DMA is this wave's only vector-memory traffic, so one s_waitcnt vmcnt(0) says it
// has landed (a walk's own stores would sit in the same counter, and loads and stores do not retire in order).
__device__ __forceinline__ void ctc_fast_loader(const float* __restrict__ lprobs, const int* __restrict__ seq_len, int Bp, float* ring,
                                                volatile int* sync) {
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const int Tb = seq_len[b], npos = Tb - 1;
  const int nch = (npos + FAST_CH - 1) / FAST_CH;
  const char* yb = reinterpret_cast<const char*>(lprobs + (size_t)b * 32);
  const unsigned rstride = (unsigned)Bp * 32u * 4u;         // bytes between the rows of two frames
  for (int n = 0; n < nch; ++n) {
    if (n >= 2)
      for (unsigned spin = 0; (sync[1] < n - 1 || sync[2] < n - 1) && spin < (1u << 24); ++spin) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = min(FAST_CH * n + 8 * q + (lane >> 3), npos - 1);          // (clamped to the last position)
        const int f = w == 0 ? p + 1 : Tb - 1 - p;
        const char* g = yb + (size_t)((unsigned)f * rstride) + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(ring + w * (FAST_RING * 32) + (((FAST_CH * n) & (FAST_RING - 1)) + 8 * q) * 32),
                                         16, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) sync[0] = n + 1;
  }
}

// one workgroup per utterance: wave 0 = alpha, wave 1 = beta, (FAST) wave 2 = the emissions' loader
template <int KS, bool DPP, bool FAST>
__global__ __launch_bounds__(FAST ? 192 : 128) void ctc_alpha_beta_kernel(
    const float* __restrict__ logits, const float* __restrict__ logz, const float* __restrict__ lprobs,
    const int* __restrict__ labels, const int* __restrict__ label_len, const int* __restrict__ seq_len,
    float* __restrict__ alpha, float* __restrict__ beta, double* __restrict__ aoff, double* __restrict__ boff,
    double* __restrict__ goff, float* __restrict__ nll, double* __restrict__ logp_out, int Bp, int Cp, int C, int Lmax, int Tws,
    int KG) {
  __shared__ float fin[64 * KS];
  if constexpr (FAST) {
    __shared__ __attribute__((aligned(1024))) float ring[2 * FAST_RING * 32];
    __shared__ int sync[3];
    if (threadIdx.x < 3) sync[threadIdx.x] = 0;
    __syncthreads();
    constexpr int K2 = KS >= 2 ? KS : 2;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w == 0)
      ctc_fast_walk<K2, true>(labels, label_len, seq_len, alpha, beta, goff, nll, logp_out, lprobs, Bp, C, Lmax, Tws, KG, ring, fin, sync);
    else if (w == 1)
      ctc_fast_walk<K2, false>(labels, label_len, seq_len, alpha, beta, goff, nll, logp_out, lprobs, Bp, C, Lmax, Tws, KG, ring, fin, sync);
    else
      ctc_fast_loader(lprobs, seq_len, Bp, ring, sync);
  } else {
    ctc_ab_log<KS, DPP>(logits, logz, labels, label_len, seq_len, alpha, beta, aoff, boff, nll, logp_out, Bp, Cp, C, Lmax, Tws, fin);
  }
}

void launch_ctc_alpha_beta(const CtcDims& d, const float* logits, const float* logz, const int* labels,
                           const int* label_len, const int* seq_len, float* alpha, float* beta, double* aoff,
                           double* boff, float* nll, double* logp, hipStream_t st) {
  static const bool no_dpp = getenv("NASR_CTC_DPP") && getenv("NASR_CTC_DPP")[0] == '0';
  const bool fast = ctc_fast_ok(d);
  const int KG = d.Tws / FAST_G + 3;
#define NASR_AB2(K, D, F)                                                                                                \
  hipLaunchKernelGGL((ctc_alpha_beta_kernel<K, D, F>), dim3(d.B), dim3(F ? 192 : 128), 0, st, logits, logz, d.lprobs, labels, \
                     label_len, seq_len, alpha, beta, aoff, boff, d.goff, nll, logp, d.Bp, d.Cp, d.C, d.Lmax, d.Tws, KG)
#define NASR_AB(K)                                                                                                       \
  if (fast) NASR_AB2(K, true, true);                                                                                     \
  else if (no_dpp) NASR_AB2(K, false, false);                                                                            \
  else NASR_AB2(K, true, false)
  switch (d.KS) {
    case 1: NASR_AB2(1, true, false); break;
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
#undef NASR_AB2
#undef NASR_AB2
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
                                                       const double* __restrict__ logp, const double* __restrict__ goff,
                                                       float scale, int Tp, int B, int Bp, int C, int Cp, int Lmax, int KS,
                                                       int Tws, int KG) {
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
  const float* al = alpha + ((size_t)b * Tws + t) * KS * 64;
  const float* be = beta + ((size_t)b * Tws + t) * KS * 64;
  float blank = 0.f;
  if (goff) {
    // the workspace of the engineered lattice: base-2 logarithms, [lane][KS] per frame, one column offset per group of 4
    // frames and walk
    const int Tb = seq_len[b];
    const int ga = t == 0 ? 0 : 1 + (t - 1) / FAST_G, gb = t == Tb - 1 ? 0 : 1 + (Tb - 2 - t) / FAST_G;
    const float coff2 = (float)(goff[(size_t)b * 2 * KG + ga] + goff[((size_t)b * 2 + 1) * KG + gb] - logp[b] * 1.44269504088896341);
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      const float wgt = s < S ? exp2f(al[lane * KS + i] + be[lane * KS + i] + coff2) : 0.f;
      wl[s] = wgt;
      if (!(s & 1)) blank += wgt;
    }
  } else {
    const float coff = (float)(aoff[(size_t)b * Tws + t] + boff[(size_t)b * Tws + t] - logp[b]);
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      const float wgt = s < S ? __expf(al[i * 64 + lane] + be[i * 64 + lane] + coff) : 0.f;
      wl[s] = wgt;
      if (!(s & 1)) blank += wgt;
    }
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
                     logz, label_len, seq_len, cstart, cpos, alpha, beta, aoff, boff, logp, ctc_fast_ok(d) ? d.goff : nullptr, scale,
                     d.Tp, d.B, d.Bp, d.C, d.Cp, d.Lmax, d.KS, d.Tws, d.Tws / FAST_G + 3);
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

// merge repeats, drop blanks: one wave per utterance, 64 frames per turn (a frame is kept when it differs from the frame before
// and is no blank; its place is the count of kept frames before it: ballot + popcount) - T/64 turns instead of T dependent
// loads on one thread (90 us of every training step with the greedy LER at T = 500)
__global__ __launch_bounds__(64) void collapse_kernel(const int* __restrict__ am, const int* __restrict__ seq_len,
                                                      int* __restrict__ ids, int* __restrict__ lens, int Tp, int B, int Bp,
                                                      int blank) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int Tb = seq_len[b];
  int n = 0, carry = -1;                      // kept so far; the class of the last frame of the turn before
  for (int t0 = 0; t0 < Tb; t0 += 64) {
    const int t = t0 + lane;
    const int k = t < Tb ? am[t * Bp + b] : blank;
    int prev = __shfl_up(k, 1);
    if (lane == 0) prev = carry;
    const bool keep = t < Tb && k != prev && k != blank;
    const unsigned long long m = __ballot(keep);
    if (keep) ids[(size_t)b * Tp + n + __popcll(m & ((1ull << lane) - 1ull))] = k;
    n += __popcll(m);
    carry = __shfl(k, 63);
  }
  if (lane == 0) lens[b] = n;
}

void launch_greedy(const CtcDims& d, const float* logits, const int* seq_len, int* argmax_ws, int* ids, int* lens,
                   hipStream_t st) {
  const int rows = d.Tp * d.Bp;
  hipLaunchKernelGGL(argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, seq_len, argmax_ws, d.Tp, d.B, d.Bp,
                     d.C, d.Cp);
  hipLaunchKernelGGL(collapse_kernel, dim3(d.B), dim3(64), 0, st, argmax_ws, seq_len, ids, lens, d.Tp, d.B, d.Bp, d.C - 1);
}

}  // namespace nasr
