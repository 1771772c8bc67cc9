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

// the lattice on probabilities (ctc_ab_lin) wants its workspaces, 32-float logit rows with a zero column to park idle states on
static bool ctc_lin_ok(const CtcDims& d) {
  static const bool no_lin = getenv("NASR_CTC_LIN") && getenv("NASR_CTC_LIN")[0] == '0';     // log-domain recursions only
  return !no_lin && d.probs && d.kexp && d.fmt && d.Cp == 32 && d.C <= 31 && d.KS >= 2;
}

// ------------------------------------------------------------------ (1) log partition per row
// (probs, or NULL: the row's softmax y(t,k) = exp(x - logZ) too, zero in the columns from C on - the emissions of the
// lattice kernel that works on probabilities)
__global__ __launch_bounds__(256) void ctc_logz_kernel(const float* __restrict__ logits, const int* __restrict__ seq_len,
                                                       float* __restrict__ logz, float* __restrict__ probs, int Tp, int B, int Bp,
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
  if (probs)
    for (int c = lane; c < Cp; c += 64) probs[(size_t)row * Cp + c] = c < C ? __expf(x[c] - z) : 0.f;
}

void launch_ctc_logz(const CtcDims& d, const float* logits, const int* seq_len, float* logz, hipStream_t st) {
  const int rows = d.Tp * d.Bp;
  hipLaunchKernelGGL(ctc_logz_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, seq_len, logz, ctc_lin_ok(d) ? d.probs : nullptr,
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

// ------------------------------------------------------------------ (2b) the same recursions on PROBABILITIES
// alpha(t,u) = y(l'_u,t) (alpha(t-1,u) + alpha(t-1,u-1) [+ alpha(t-1,u-2)]): two fused multiply-adds and a multiply per state
// and frame behind the two DPP moves, against three exp and a log in the log domain - the lattice is a chain of T dependent
// frames on ONE wave, bound by its own instruction count (144 instructions a frame there).
//
// Range.  A column of alpha spans far more than fp32's exponents (at T = 500 the forward mass runs hundreds of binary orders
// ahead of the band the posterior lives in), so every LANE carries its own binary exponent: alpha = a * 2^kme for the KS
// states of a lane.  A neighbour's values arrive in the neighbour's scale and are brought into the lane's own by the factor
// f = 2^(kme_n - kme), which rides in the multiply-add that sums them: no instruction more.  Every 4 frames a lane looks at its
// own largest value and chooses the power of two kp that the emissions of the first frame of the NEXT group will carry
// (kp_next = floor(log2 max) - kp: the level after the scale still pending, so the level is back at ~1 a group later; taken
// from max alone it would oscillate with period 6 groups).  A lane the recursion has not reached yet (all zero) copies its
// neighbour's exponents, so that what first arrives there is representable.  All of this is integer work off the chain.
//
// Emissions.  The softmax rows y(t,.) (ctc_logz_kernel: 32 floats a frame, a zero column for the idle states) come in by
// LDS-DMA, 32 frames per 4 instructions, a chunk ahead; the lattice gathers its KS values per frame from that ring with
// ds_read_b32, one group ahead.  No register staging, no per-frame global gathers to wait for.
//
// Workspace: a values [t][lane][KS] (one store per frame) and the lane exponents per GROUP of frames, kexp[g][lane].
// An utterance whose numbers leave the range this scheme covers (|exponent step| > 100 between two rescalings or between two
// neighbouring lanes: emission probabilities under ~1e-4 for 8 frames in a row, or no valid path at all) is FLAGGED and redone by
// the log-domain recursion above in the same launch; fmt[b] tells ctc_grad which form the workspace of utterance b holds.
constexpr int LIN_G = 4, LIN_CH = 32, LIN_RING = 64;

// The stores of the lattice as ONE instruction per <= 4 states.  (s_nop: a store of more than 64 bits reads its data
// registers a cycle after it issues; hipcc keeps the next VALU write of them away from its own stores, not from these.)
typedef float lin_f2 __attribute__((ext_vector_type(2)));
typedef float lin_f3 __attribute__((ext_vector_type(3)));
typedef float lin_f4 __attribute__((ext_vector_type(4)));
template <int KS>
__device__ __forceinline__ void lin_store(float* dst, const float (&a)[KS]) {
#pragma unroll
  for (int j = 0; j + 4 <= KS; j += 4) {
    const lin_f4 v = {a[j], a[j + 1], a[j + 2], a[j + 3]};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst + j), "v"(v) : "memory");
  }
  constexpr int j = KS & ~3;
  if constexpr (KS % 4 == 3) {
    const lin_f3 v = {a[j], a[j + 1], a[j + 2]};
    asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" : : "v"(dst + j), "v"(v) : "memory");
  } else if constexpr (KS % 4 == 2) {
    const lin_f2 v = {a[j], a[j + 1]};
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst + j), "v"(v) : "memory");
  } else if constexpr (KS % 4 == 1) {
    asm volatile("global_store_dword %0, %1, off" : : "v"(dst + j), "v"(a[j]) : "memory");
  }
}
template <int KS>
constexpr int lin_nst() { return (KS + 3) / 4; }            // store instructions per frame

__device__ __forceinline__ int dpp_shr_i(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_shl_i(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float pow2i(int d) { return d <= -127 ? 0.f : __int_as_float((d + 127) << 23); }

template <int KS, bool fwd>
__device__ __forceinline__ void ctc_lin_walk(const float* __restrict__ probs, const int* __restrict__ labels,
                                           const int* __restrict__ label_len, const int* __restrict__ seq_len,
                                           float* __restrict__ alpha, float* __restrict__ beta, int* __restrict__ kexp,
                                           float* __restrict__ nll, double* __restrict__ logp_out, int Bp, int C, int Lmax, int Tws,
                                           int KG, float* ring, float* fin, int* fink, int* bad, volatile int* sync) {
  static_assert(KS >= 2, "two states per lane at least: the skip transition then never reaches past the neighbour lane");
  constexpr int G = LIN_G;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = fwd ? 0 : 1;
  const int L = label_len[b], Tb = seq_len[b], S = 2 * L + 1;
  const int blank = C - 1;
  const int* lab = labels + (size_t)b * Lmax;
  int ext[KS];
  float sk[KS];                             // 1: the transition from two states away exists
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int s = lane * KS + i;
    const bool act = s < S;
    ext[i] = act ? ((s & 1) ? lab[s >> 1] : blank) : 31;      // column 31 of a softmax row is zero
    bool k2;
    if (fwd) {
      const int e2 = (s >= 2 && (s & 1)) ? lab[(s >> 1) - 1] : blank;   // l'_{s-2}
      k2 = act && s >= 2 && ext[i] != blank && ext[i] != e2;
    } else {
      const int e2 = (s + 2 < S && (s & 1)) ? lab[(s >> 1) + 1] : blank;   // l'_{s+2}
      k2 = (s + 2 < S) && e2 != blank && e2 != ext[i];
    }
    sk[i] = k2 ? 1.f : 0.f;
  }
  // positions p = 0 .. Tb-2 of this wave's walk: the emission frame of position p is p+1 (alpha, producing frame p+1) or
  // Tb-1-p (beta, producing frame Tb-2-p)
  const int npos = Tb - 1;
  auto frame_of = [&](int p) { return fwd ? p + 1 : Tb - 1 - p; };
  float* rg = ring + w * (LIN_RING * 32);
  const float* yb = probs + (size_t)b * 32;
  const unsigned rstride = (unsigned)Bp * 32u * 4u;         // bytes between the rows of two frames
  // (plain LDS loads between compiler barriers: the rows change under this wave - the loader's LDS-DMA is nothing hipcc can
  //  see - so nothing may be kept from an earlier gather, and the gathers must stay behind the wait for the loader's word.
  //  Volatile accesses would do that too, but hipcc waits for EVERY outstanding store around each of them.)
  auto read_group = [&](int g, float (&e)[G][KS]) {        // emissions of positions 4g .. 4g+3
    asm volatile("" ::: "memory");
    const float* r0 = rg + ((G * g) & (LIN_RING - 1)) * 32;
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int k = 0; k < G; ++k) e[k][i] = r0[k * 32 + ext[i]];
    asm volatile("" ::: "memory");
  };
  auto read_done = [&](float (&)[G][KS]) {};
  // the words the three waves talk through, read and written without the waits hipcc wraps around a volatile access
  const unsigned sync_lds = (unsigned)(size_t)((lds_ptr_t)sync);
  auto sync_ld = [&](int i) {
    int v;
    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(sync_lds), "n"(0) : "memory");
    (void)i;
    return v;
  };
  float* ws = (fwd ? alpha : beta) + (size_t)b * Tws * KS * 64;
  int* kx = kexp + ((size_t)b * 2 + w) * KG * 64;
  auto store = [&](int t, const float (&a)[KS]) { lin_store<KS>(ws + ((size_t)t * 64 + lane) * KS, a); };
  // sync[0]: chunks the loader wave has landed in the rings (both directions); sync[1 + w]: chunks this walk is done with
  auto wait_ready = [&](int n) {
    for (unsigned spin = 0; sync_ld(0) < n && spin < (1u << 24); ++spin) __builtin_amdgcn_s_sleep(1);
  };

  float a[KS];
  int kme = 0, kp = 0;
  if (fwd) {
    const float* y0 = yb;                                   // frame 0
#pragma unroll
    for (int i = 0; i < KS; ++i) a[i] = (lane * KS + i < 2) ? y0[ext[i]] : 0.f;
    store(0, a);
  } else {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      a[i] = (s < S && (s == S - 1 || s == S - 2)) ? 1.f : 0.f;
    }
    store(Tb - 1, a);
  }
  kx[lane] = 0;                                             // group 0: the column above, exponent 0
  if (npos > 0) {
    const int nch = (npos + LIN_CH - 1) / LIN_CH;
    wait_ready(1);
    float e[G][KS], en[G][KS];
    read_group(0, e);
    read_done(e);
    for (int c = 0; c < nch; ++c) {
      for (int gi = 0; gi < LIN_CH / G; ++gi) {
        const int g = c * (LIN_CH / G) + gi, p0 = G * g;
        if (p0 >= npos) break;
        if (gi == LIN_CH / G - 1 && c + 1 < nch) wait_ready(c + 2);     // the next chunk's first group: normally long there
        read_group(g + 1, en);
        // ---- exponents (integer work, off the chain)
        float m = a[0];
#pragma unroll
        for (int i = 1; i < KS; ++i) m = fmaxf(m, a[i]);
        const int em = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
        const bool nz = m > 0.f;
        int kpn = nz ? em - kp : 0;
        // lanes the recursion has not reached (all zero; they lie beyond its front: alpha's non-zero lanes are a prefix of the
        // wave, beta's end at the last active lane) take the FRONT lane's exponents, so that what arrives there first is
        // representable however fast the front moves (up to 8 states = 2.7 lanes a group)
        const unsigned long long nzmask = __ballot(nz);
        if (nzmask) {
          const int lf = fwd ? 63 - __builtin_clzll(nzmask) : __builtin_ctzll(nzmask);
          const int kme_f = __builtin_amdgcn_readlane(kme, lf), kp_f = __builtin_amdgcn_readlane(kp, lf);
          const int kpn_f = __builtin_amdgcn_readlane(kpn, lf);
          if (!nz) { kme = kme_f; kp = kp_f; kpn = kpn_f; }
        }
        const int nzi = nz ? 1 : 0;
        int kme_n, kp_n, nz_n;
        if (fwd) { kme_n = dpp_shr_i(kme, kme); kp_n = dpp_shr_i(kp, kp); nz_n = dpp_shr_i(nzi, 0); }
        else { kme_n = dpp_shl_i(kme, kme); kp_n = dpp_shl_i(kp, kp); nz_n = dpp_shl_i(nzi, 0); }
        const int d1 = kme_n - kme, d2 = (kme_n + kp_n) - (kme + kp);
        if ((nz && (em < -100 || em > 100 || kp < -120 || kp > 120)) || (nz_n && (d1 > 100 || d2 > 100))) *bad = 1;
        const float f1 = pow2i(min(d1, 100)), f2 = pow2i(min(d2, 100));
        const float sc = pow2i(-kp);
#pragma unroll
        for (int i = 0; i < KS; ++i) e[0][i] *= sc;            // the first frame of the group carries 2^-kp
        kme += kp;
        kp = kpn;
        asm volatile("global_store_dword %0, %1, off" : : "v"(kx + (g + 1) * 64 + lane), "v"(kme) : "memory");
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const bool live = p0 + k < npos;
          float na[KS];
          if (fwd) {
            const float f = k == 0 ? f1 : f2;
            const float p1 = lane_up1<true>(a[KS - 1], 0.f), p2 = lane_up1<true>(a[KS - 2], 0.f);
#pragma unroll
            for (int i = 0; i < KS; ++i) {
              float x;
              if (i == 0) x = fmaf(f * sk[0], p2, fmaf(f, p1, a[0]));
              else if (i == 1) x = fmaf(f * sk[1], p1, a[1] + a[0]);
              else x = fmaf(sk[i], a[i - 2 >= 0 ? i - 2 : 0], a[i] + a[i - 1]);
              na[i] = e[k][i] * x;
            }
          } else {
            float bb[KS];
#pragma unroll
            for (int i = 0; i < KS; ++i) bb[i] = a[i] * e[k][i];
            const float n1 = lane_down1<true>(bb[0], 0.f), n2 = lane_down1<true>(bb[1], 0.f);
#pragma unroll
            for (int i = 0; i < KS; ++i) {
              if (i == KS - 1) na[i] = fmaf(f2 * sk[i], n2, fmaf(f2, n1, bb[i]));
              else if (i == KS - 2) na[i] = fmaf(f2 * sk[i], n1, bb[i] + bb[KS - 1]);
              else na[i] = fmaf(sk[i], bb[i + 2 < KS ? i + 2 : 0], bb[i] + bb[i + 1]);
            }
          }
#pragma unroll
          for (int i = 0; i < KS; ++i) a[i] = live ? na[i] : a[i];
          // (dead frames of the last group store dead copies: rows up to Tb+2 / down to -3 ... the workspace has Tws = T+8
          //  rows and beta's walk is clamped to row 0: the store count per group stays what the counted wait above assumes)
          store(fwd ? p0 + k + 1 : max(Tb - 2 - p0 - k, 0), a);
        }
        read_done(en);
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
          for (int i = 0; i < KS; ++i) e[k][i] = en[k][i];
      }
      // every gather from this chunk's rows has returned (their values were used above): the rows may be reused
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1 offset:%2" : : "v"(sync_lds), "v"(c + 1), "n"(4 * (1 + w)) : "memory");
    }
  }
  if (fwd) {
#pragma unroll
    for (int i = 0; i < KS; ++i) fin[lane * KS + i] = a[i];
    fink[lane] = kme;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane == 0) {
      const int s1 = S - 1, s2 = S - 2;
      const int k1 = fink[s1 / KS], k2 = s2 >= 0 ? fink[s2 / KS] : k1;
      const int km = max(k1, k2);
      const double x = ldexp((double)fin[s1], k1 - km) + (s2 >= 0 ? ldexp((double)fin[s2], k2 - km) : 0.0);
      if (!(x > 0.0) || !(x < 1e300)) *bad = 1;      // no path, or nothing finite left of it: the log-domain pass says which
      const double lp = (double)km * 0.693147180559945309417 + log(x);
      logp_out[b] = lp;
      nll[b] = (float)(-lp);
    }
  }
}

// wave 2: the emissions' way into LDS.  Chunk n (32 positions of both walks) goes to the ring half chunk n-2 was in, once both
// walks are done with that one; its LDS-DMA is this wave's only vector-memory traffic, so one s_waitcnt vmcnt(0) says it
// has landed (a walk's own stores would sit in the same counter, and loads and stores do not retire in order).
__device__ __forceinline__ void ctc_lin_loader(const float* __restrict__ probs, const int* __restrict__ seq_len, int Bp, float* ring,
                                               volatile int* sync) {
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const int Tb = seq_len[b], npos = Tb - 1;
  const int nch = (npos + LIN_CH - 1) / LIN_CH;
  const char* yb = reinterpret_cast<const char*>(probs + (size_t)b * 32);
  const unsigned rstride = (unsigned)Bp * 32u * 4u;         // bytes between the rows of two frames
  for (int n = 0; n < nch; ++n) {
    if (n >= 2)
      for (unsigned spin = 0; (sync[1] < n - 1 || sync[2] < n - 1) && spin < (1u << 24); ++spin) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = min(LIN_CH * n + 8 * q + (lane >> 3), npos - 1);          // (clamped to the last position)
        const int f = w == 0 ? p + 1 : Tb - 1 - p;
        const char* g = yb + (size_t)((unsigned)f * rstride) + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(ring + w * (LIN_RING * 32) + (((LIN_CH * n) & (LIN_RING - 1)) + 8 * q) * 32),
                                         16, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) sync[0] = n + 1;
  }
}

template <int KS>
__device__ __forceinline__ void ctc_ab_lin(const float* __restrict__ probs, const int* __restrict__ labels,
                                           const int* __restrict__ label_len, const int* __restrict__ seq_len,
                                           float* __restrict__ alpha, float* __restrict__ beta, int* __restrict__ kexp,
                                           float* __restrict__ nll, double* __restrict__ logp_out, int Bp, int C, int Lmax, int Tws,
                                           int KG, float* ring, float* fin, int* fink, int* bad, volatile int* sync) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w == 0)
    ctc_lin_walk<KS, true>(probs, labels, label_len, seq_len, alpha, beta, kexp, nll, logp_out, Bp, C, Lmax, Tws, KG, ring, fin, fink, bad,
                           sync);
  else if (w == 1)
    ctc_lin_walk<KS, false>(probs, labels, label_len, seq_len, alpha, beta, kexp, nll, logp_out, Bp, C, Lmax, Tws, KG, ring, fin, fink,
                            bad, sync);
  else
    ctc_lin_loader(probs, seq_len, Bp, ring, sync);
}

// one workgroup per utterance, wave 0 = alpha, wave 1 = beta: on probabilities (LIN), redone in the log domain where flagged
template <int KS, bool DPP, bool LIN>
__global__ __launch_bounds__(LIN ? 192 : 128) void ctc_alpha_beta_kernel(
    const float* __restrict__ logits, const float* __restrict__ logz, const float* __restrict__ probs,
    const int* __restrict__ labels, const int* __restrict__ label_len, const int* __restrict__ seq_len,
    float* __restrict__ alpha, float* __restrict__ beta, double* __restrict__ aoff, double* __restrict__ boff,
    int* __restrict__ kexp, float* __restrict__ nll, double* __restrict__ logp_out, int* __restrict__ fmt, int Bp, int Cp, int C,
    int Lmax, int Tws, int KG) {
  __shared__ float fin[64 * KS];
  if constexpr (LIN) {
    __shared__ __attribute__((aligned(1024))) float ring[2 * LIN_RING * 32];
    __shared__ int fink[64];
    __shared__ int bad;
    __shared__ int sync[3];
    if (threadIdx.x < 3) sync[threadIdx.x] = 0;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    ctc_ab_lin<(KS >= 2 ? KS : 2)>(probs, labels, label_len, seq_len, alpha, beta, kexp, nll, logp_out, Bp, C, Lmax, Tws, KG, ring,
                                   fin, fink, &bad, sync);
    __syncthreads();
    if (!bad) {
      if (threadIdx.x == 0) fmt[blockIdx.x] = 0;
      return;
    }
    if (threadIdx.x >= 128) return;      // the loader wave has no part in the log-domain pass
  }
  ctc_ab_log<KS, DPP>(logits, logz, labels, label_len, seq_len, alpha, beta, aoff, boff, nll, logp_out, Bp, Cp, C, Lmax, Tws, fin);
  if (fmt && threadIdx.x == 0) fmt[blockIdx.x] = 1;
}

void launch_ctc_alpha_beta(const CtcDims& d, const float* logits, const float* logz, const int* labels,
                           const int* label_len, const int* seq_len, float* alpha, float* beta, double* aoff,
                           double* boff, float* nll, double* logp, hipStream_t st) {
  static const bool no_dpp = getenv("NASR_CTC_DPP") && getenv("NASR_CTC_DPP")[0] == '0';
  const bool lin = ctc_lin_ok(d) && !no_dpp;
  const int KG = d.Tws / LIN_G + 3;
#define NASR_AB2(K, D, LN)                                                                                               \
  hipLaunchKernelGGL((ctc_alpha_beta_kernel<K, D, LN>), dim3(d.B), dim3(LN ? 192 : 128), 0, st, logits, logz, d.probs, labels, \
                     label_len, seq_len, alpha, beta, aoff, boff, d.kexp, nll, logp, d.fmt, d.Bp, d.Cp, d.C, d.Lmax, d.Tws, KG)
#define NASR_AB(K)                                                                                                       \
  if (lin) NASR_AB2(K, true, true);                                                                                      \
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
                                                       const double* __restrict__ logp, const int* __restrict__ kexp,
                                                       const int* __restrict__ fmt, float scale, int Tp, int B, int Bp, int C,
                                                       int Cp, int Lmax, int KS, int Tws, int KG) {
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
  if (fmt && fmt[b] == 0) {
    // probabilities a * 2^k, [lane][KS] per frame, one exponent per lane and group of 4 frames (ctc_ab_lin); the product as a
    // sum of logarithms: alpha beta may be far outside fp32 where the posterior is not
    const int Tb = seq_len[b];
    const int ga = t == 0 ? 0 : 1 + (t - 1) / LIN_G, gb = t == Tb - 1 ? 0 : 1 + (Tb - 2 - t) / LIN_G;
    const int ka = kexp[((size_t)b * 2 * KG + ga) * 64 + lane], kb = kexp[(((size_t)b * 2 + 1) * KG + gb) * 64 + lane];
    const double lp2 = -logp[b] * 1.44269504088896341;        // posterior = a_alpha a_beta 2^(ka + kb + lp2)
    const double lpi = floor(lp2);
    const int E0 = ka + kb + (int)lpi;
    const float cm = exp2f((float)(lp2 - lpi));                // in [1, 2)
    for (int i = 0; i < KS; ++i) {
      const int s = lane * KS + i;
      const float av = al[lane * KS + i], bv = be[lane * KS + i];
      float wgt = 0.f;
      if (s < S && av > 0.f && bv > 0.f) {
        int ea, eb;
        const float ma = frexpf(av, &ea), mb = frexpf(bv, &eb);     // mantissas in [0.5, 1): their product cannot leave fp32
        wgt = ldexpf(ma * mb * cm, min(max(E0 + ea + eb, -200), 100));
      }
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
                     logz, label_len, seq_len, cstart, cpos, alpha, beta, aoff, boff, logp, d.kexp, ctc_lin_ok(d) ? d.fmt : nullptr, scale,
                     d.Tp, d.B, d.Bp, d.C, d.Cp, d.Lmax, d.KS, d.Tws, d.Tws / LIN_G + 3);
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
