// lstm_persist.hip — the LSTM recurrence and its BPTT as ONE persistent launch per layer pass.
//
// Same semantics as lstm.hip (networks/bilstm_ctc_net.py:17-28,
// networks/lstm_ctc_net.py:17-23) and the same buffers (gates / c / out / dG, frame-indexed, zero past seq_len).
//
// Why: the per-timestep launches of lstm.hip cost 3.7 (forward) / 4.3 us (BPTT) each, 1.55 us of it the launch
// boundary and most of the rest re-streaming the recurrent matrix (8 MB per step for 2 x 500) through the fabric.
// Measured with tools/xcdbench.hip on MI355X: 32 workgroups on ONE XCD can all-gather a 8 KB state through that
// XCD's L2 in ~0.6 us (store -> L2 ack 245 cycles, flag poll ~500, payload ~530), with plain stores and `sc1`
// (L1-bypassing) loads, 0 stale reads in 2 M checked words; the chip-wide forms (sc1 write-through stores,
// cross-XCD groups) cost 1.7 us.  So:
//
//   * an XCD is a GROUP: (direction d, a slice of <= 4 utterances of the batch).  2 directions x 4 slices (or
//     1 x 8 for a unidirectional net) use all 8 XCDs, and the groups never talk to each other.
//   * every CU of the group (its 32 workgroups, one per CU) owns Hp/32 hidden units for ALL T steps and keeps its
//     slice of the recurrent matrix (Hp x 4Hp/32 floats = 128 KB at Hp = 512) in REGISTERS: 128 VGPRs per lane in
//     each of its 4 waves.  The matrix is read from HBM once per layer pass instead of once per timestep.
//   * the 4 utterances are the 4 rows of v_mfma_f32_4x4x1_16b_f32 (exact fp32): its 16 blocks are 16 hidden
//     units x (i,j,f,o), the A operand (h) is broadcast from one block with cbsz/abid, so one 16-byte load per
//     lane feeds 64 MFMAs.
//   * forward step: each MFMA wave waits for the 8 producers of its K quarter (one flag word each), loads their h
//     (2 x 16 B per lane), 128 MFMAs, 4-wave LDS reduction, wave 0 does the 64 cell updates (c stays in a
//     register for the whole sequence), publishes h and the flag.  A fifth wave owns the per-frame HBM traffic
//     (gates in, activations / c / out out) and talks to the cell wave through LDS.
//   * BPTT step: split-K the other way round (as lstm.hip): a CU turns dh of its own 16 units into dG (dc stays
//     in a register), multiplies by ITS columns of U for all Hp outputs and hands 32 partial rows to the 32
//     consumers; the consumer sums 32 partials.  Deterministic, no atomics.
//
// Placement: which workgroups share an XCD is read from HW_REG_XCC_ID at run time and a ticket per XCD gives the
// member index; nothing is assumed about dispatch order.  The hand-off relies on one hardware fact only: CUs that
// report the same XCC id share one L2, stores write through L1 to it and `sc1` loads are served from it.  Every
// spin is bounded; a timeout or an unexpected placement (not 32 workgroups on each of 8 XCDs) raises
// PersistCtl::error, the launch drains, and the host falls back to the per-step kernels of lstm.hip for good.
#include "kernels.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

#ifndef NASR_PSTAMP
#define NASR_PSTAMP 0   // 1: wave 0 of one workgroup accumulates s_memtime deltas per phase into PersistCtl::pad (tools/persistbench)
#endif

#ifndef NASR_FWD_EPOCH
#define NASR_FWD_EPOCH 1   // forward hand-off: h words carry an epoch bit, consumers poll the payload itself (0: flag, then payload)
#endif
#ifndef NASR_BWD_EPOCH
#define NASR_BWD_EPOCH 1   // BPTT hand-off: every partial sum carries an epoch bit in its last mantissa bit, consumers poll the sums
#endif
#ifndef NASR_BWD_NACC
#define NASR_BWD_NACC 4    // accumulator chains per output group of the BPTT product (2 or 4)
#endif
#ifndef NASR_EP_DELAY
#define NASR_EP_DELAY 0    // s_sleep units between "this CU has published" and the first load of the others' h
#endif

namespace nasr {

namespace {

struct Stamps {
  unsigned long long last;
  unsigned acc[12];
  bool on;
  __device__ __forceinline__ void start(bool enable) {
    on = enable;
    for (int i = 0; i < 12; ++i) acc[i] = 0;
    last = NASR_PSTAMP ? __builtin_amdgcn_s_memtime() : 0ull;
  }
  __device__ __forceinline__ void mark(int i) {
#if NASR_PSTAMP
    if (on) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      acc[i] += (unsigned)(t - last);
      last = t;
    }
#endif
  }
  __device__ __forceinline__ void flush(PersistCtl* ctl, unsigned slot, bool first) {
#if NASR_PSTAMP
    if (on && (threadIdx.x & 63) == 0) {
      for (int i = 0; i < 12; ++i) ctl->stamps[slot & 255][i] = acc[i];
      if (first)
        for (int i = 0; i < 12; ++i) ctl->pad[i] = acc[i];
    }
#endif
  }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef volatile __attribute__((address_space(3))) unsigned lds_vu32;   // a volatile access through a GENERIC pointer to LDS
                                                                        // compiles to flat_store sc0 sc1 + vmcnt(0)

// The bare v_exp_f32 (2^x), without the denormal scaling __expf can wrap around it: an exponential that overflows to inf
// or flushes to 0 gives the saturated value of the sigmoid / tanh either way, and in the normal range the two are the
// same instruction on the same input (results bitwise equal; measured time equal too - the cell wave's chain is latency,
// not issue).
__device__ __forceinline__ float pexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float psig(float x) { return __builtin_amdgcn_rcpf(1.f + pexp(-x)); }
__device__ __forceinline__ float ptanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + pexp(2.f * x)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr unsigned SPIN_BUDGET = 1u << 21;   // polls before a wave gives up (~0.5 s)

// wait until every active lane's word is >= want (monotonic step counters; wrap-safe compare)
__device__ __forceinline__ bool poll_ge(gu32* p, bool active, unsigned want) {
  for (unsigned n = 0; n < SPIN_BUDGET; ++n) {
    const unsigned v = active ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
    if (__all((int)(v - want) >= 0)) return true;
  }
  return false;
}

// (xcc id, ticket within the XCD) of this workgroup; false when the placement is not 32-per-XCD-of-8
__device__ __forceinline__ void raise_error(PersistCtl* ctl, unsigned* sticky, float* fault, unsigned code) {
  atomicOr(&ctl->error, code);
  if (fault) *fault = 1.f;   // sits behind the gradients: all-reduced with them, makes Adam a no-op on every rank
  if (sticky) __hip_atomic_store(sticky, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible, never cleared by a launch
}

__device__ __forceinline__ bool join_group(PersistCtl* ctl, unsigned* sticky, float* fault, unsigned* info, unsigned& xcc,
                                           unsigned& member) {
  if (threadIdx.x == 0) {
    const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
    info[0] = x;
    info[1] = x < 8 ? atomicAdd(&ctl->xcc_count[x], 1u) : 0xffffu;
    info[2] = 0;
    info[3] = 0;
    info[4] = 0;
    info[5] = 0;
  }
  __syncthreads();
  xcc = info[0];
  member = info[1];
  if (xcc >= 8 || member >= 32) {
    if (threadIdx.x == 0) raise_error(ctl, sticky, fault, 2u);
    return false;
  }
  return true;
}

}  // namespace

// ------------------------------------------------------------------ operand images
// NU = Hp/32 units per CU, KW = 8*NU.  U canonical [Hp][N4], column 4*j+g.
//   Upf [32 m][4 w][KW idx][64 lane]: idx = 4*bb + r contracts unit k = w*KW + idx; lane = 4*b' + g is the
//        output column of unit NU*m + b', gate g (zero for b' >= NU).
//   Upb [32 m][4 w][NOG*4NU idx][64 lane]: idx = og*4NU + c contracts gate column 4*NU*m + c; lane = 4*b' + jc is
//        the output unit k = w*KW + og*64 + lane (zero when og*64 + lane >= KW).
// blockIdx.y = (layer, direction) k: its matrix starts at P + off.o[k], its images at Upf0 + k*imf / Upb0 + k*imb
struct RepackOffs { long long o[PERSIST_MAX_MATS]; };
// cs0 != NULL: the forward image holds TWO fp16 planes of U scaled per column (cs0 + k*N4: power-of-two scales that bring
// each column's largest magnitude into [2^14, 2^15), gemm_tph.hip) for v_mfma_f32_4x4x4_16B_f16, same bytes:
//   Upf as 8-byte units [32 m][4 w][KW/4 bb][2 planes][64 lane]: the 4 halfs are units k = w*KW + 4*bb + 0..3 of the
//   lane's column; plane 0 = fp16(U*s), plane 1 = fp16(U*s - plane 0).
__global__ __launch_bounds__(256) void repack_persist_kernel(const float* __restrict__ P, RepackOffs off,
                                                             float* __restrict__ Upf0, float* __restrict__ Upb0,
                                                             long long imf, long long imb, int Hp,
                                                             const float* __restrict__ cs0) {
  const float* __restrict__ U = P + off.o[blockIdx.y];
  float* __restrict__ Upf = Upf0 + (size_t)blockIdx.y * imf;
  float* __restrict__ Upb = Upb0 + (size_t)blockIdx.y * imb;
  const int NU = Hp / 32, KW = 8 * NU, N4 = 4 * Hp;
  const int NOG = (KW + 63) / 64, KB = NOG * 4 * NU;
  const int64_t nf = (int64_t)32 * 4 * KW * 64, nb = (int64_t)32 * 4 * KB * 64;
  const int64_t nfe = cs0 ? nf / 4 : nf;      // fp16 image: one work item per (chunk of 4 units, lane)
  (void)nb; (void)Upb; (void)KB;
  for (int64_t e0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e0 < nfe; e0 += (int64_t)gridDim.x * blockDim.x) {
    {
      const int lane = e0 & 63;
      int64_t x = e0 >> 6;
      const int KWe = cs0 ? KW / 4 : KW;
      const int idx = (int)(x % KWe) * (cs0 ? 4 : 1); x /= KWe;
      const int w = (int)(x & 3), m = (int)(x >> 2);
      const int bp = lane >> 2, g = lane & 3;
      if (!cs0) {
        Upf[e0] = bp < NU ? U[(size_t)(w * KW + idx) * N4 + 4 * (NU * m + bp) + g] : 0.f;
      } else {                           // both planes of units idx .. idx+3
        const int col = 4 * (NU * m + bp) + g;
        const float sc = bp < NU ? cs0[(size_t)blockIdx.y * N4 + col] : 1.f;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 p1, p2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = bp < NU ? U[(size_t)(w * KW + idx + r) * N4 + col] * sc : 0.f;
          const _Float16 h1 = (_Float16)v;
          p1[r] = h1;
          p2[r] = (_Float16)(v - (float)h1);
        }
        h4* dst = reinterpret_cast<h4*>(Upf) + ((((size_t)m * 4 + w) * (KW / 4) + idx / 4) * 2) * 64 + lane;
        dst[0] = p1;
        dst[64] = p2;
      }
    }
  }
}

// The backward image is a transpose of U's tiles (lane = contraction row, index = gate column): one block per (member m,
// wave w, output group og) reads its 64 rows x 4*NU columns row-wise (coalesced) into LDS and writes them column-wise.
// (Read column-wise straight from HBM, as the first version did, the pass moved 2.8x its bytes: 69 us per optimiser step.)
__global__ __launch_bounds__(256) void repack_persist_bwd_kernel(const float* __restrict__ P, RepackOffs off,
                                                                 float* __restrict__ Upb0, long long imb, int Hp) {
  __shared__ float tile[64][65];
  const float* __restrict__ U = P + off.o[blockIdx.y];
  float* __restrict__ Upb = Upb0 + (size_t)blockIdx.y * imb;
  const int NU = Hp / 32, KW = 8 * NU, N4 = 4 * Hp, NC = 4 * NU;
  const int NOG = (KW + 63) / 64, KB = NOG * NC;
  const int og = blockIdx.x % NOG, w = (blockIdx.x / NOG) & 3, m = blockIdx.x / (NOG * 4);
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  for (int r = ry; r < 64; r += 4) {
    const int kl = og * 64 + r;
    tile[r][cx] = (kl < KW && cx < NC) ? U[(size_t)(w * KW + kl) * N4 + NC * m + cx] : 0.f;
  }
  __syncthreads();
  float* dst = Upb + (((size_t)m * 4 + w) * KB + (size_t)og * NC) * 64;
  for (int c = ry; c < NC; c += 4) dst[(size_t)c * 64 + cx] = tile[cx][c];
}

size_t persist_image_floats(int Hp, bool bwd) {
  const int NU = Hp / 32, KW = 8 * NU, NOG = (KW + 63) / 64;
  return (size_t)32 * 4 * 64 * (bwd ? NOG * 4 * NU : KW);
}

void launch_repack_persist(const float* P, const int64_t* offs, int n, float* Upf, float* Upb, int Hp, const float* col_scale,
                           hipStream_t st) {
  const long long imf = (long long)persist_image_floats(Hp, false), imb = (long long)persist_image_floats(Hp, true);
  for (int k0 = 0; k0 < n; k0 += PERSIST_MAX_MATS) {
    const int m = n - k0 < PERSIST_MAX_MATS ? n - k0 : PERSIST_MAX_MATS;
    RepackOffs off{};
    for (int k = 0; k < m; ++k) off.o[k] = offs[k0 + k];
    hipLaunchKernelGGL(repack_persist_kernel, dim3(512, m), dim3(256), 0, st, P, off, Upf + (size_t)k0 * imf,
                       Upb + (size_t)k0 * imb, imf, imb, Hp, col_scale ? col_scale + (size_t)k0 * 4 * Hp : nullptr);
    const int NU = Hp / 32, NOG = (8 * NU + 63) / 64;
    hipLaunchKernelGGL(repack_persist_bwd_kernel, dim3(32 * 4 * NOG, m), dim3(256), 0, st, P, off, Upb + (size_t)k0 * imb, imb, Hp);
  }
}

// ------------------------------------------------------------------ forward
struct PersistGeom {
  int T, Bp, Hp, D;
  int ub;        // utterance rows per group and round (<= 4)
  int rounds;    // passes over the batch (weights stay in registers)
  int inject;    // test hook (NASR_PERSIST_FAULT=s): member 0 of every group treats the poll of step s as timed out
  float* fault;  // device word raised by an aborted launch (or NULL)
};

constexpr int PERSIST_LDS_BYTES = 96 * 1024;   // > half of the CU's 160 KB: one workgroup per CU
constexpr int PERSIST_LDS_LEAN = 24 * 1024;    // what the kernels use (LDS map below: 20.1 KB): leaves room for a 3-wave GEMM
                                               // workgroup on the same CU (gemm_tph.hip <4,1,3>)
static bool g_bwd_lean = false;                // persist_set_bwd_lean

// LDS map (floats): red [2][4][4][64] | adg [2][256] | side [2][64][8] | xgb [2][64][4] | pfb [2][64][8] | info[8]
constexpr int LDS_RED = 0, LDS_ADG = 2 * 4 * 4 * 64, LDS_SIDE = LDS_ADG + 2 * 256, LDS_XGB = LDS_SIDE + 2 * 64 * 8,
              LDS_PFB = LDS_XGB + 2 * 64 * 4, LDS_INFO = LDS_PFB + 2 * 64 * 8;   // info: 16 words

// Wave roles inside a workgroup (320 threads): waves 0-3 run the MFMA part, each with a quarter of the CU's slice of
// the recurrent matrix in registers; wave 0 (the CELL wave) additionally does the 64 cell updates and publishes the
// state; wave 4 (the MEMORY wave) owns every HBM access that is not part of the hand-off: it loads the next step's
// per-frame operands ahead of time and passes them to the cell wave through LDS, and it stores the per-frame results
// (activations, c, out / dG) from LDS one step later.  It does this while the other waves are in their MFMA phase,
// when the memory system is otherwise idle.  Measured (forward, us per step): loads and stores in the cell wave 1.69,
// in an MFMA wave right after the publish 1.64 (they compete with the hand-off), none at all 1.46.

// F16: the recurrent product runs on v_mfma_f32_4x4x4_16B_f16 - U as two fp16 planes under per-column scales (register
// footprint unchanged), h (|h| < 1) split into two fp16 parts of h * 2^14 once per step BY ITS PRODUCER (they travel
// packed in the 4 bytes the fp32 value took), three MFMAs per chunk of 4 units
// (h1 U1 + h1 U2 + h2 U1) instead of four fp32 ones: 96 instead of 128 MFMAs of the same duration per wave and step,
// the same accuracy class as the fp16-plane GEMMs.  cinv [D][N4]: 1 / column scale.
template <int NU, bool F16>
__global__ __launch_bounds__(320, 1) void lstm_persist_fwd_kernel(
    const float* __restrict__ Upf,   // [D] images
    float* gates, float* cbuf, float* out, const int* __restrict__ seq_len,
    float* hx,                       // [8 groups][2 parity][Hp*4]
    PersistCtl* ctl, unsigned* sticky, PersistGeom gm, float fb, const float* __restrict__ cinv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KW = 8 * NU;                 // units (= MFMAs) per wave
  constexpr int NCH = 2 * NU;                // 16-byte chunks (4 units x 1 utterance) per wave and utterance
  constexpr int NJ = (NCH + 15) / 16;        // 16-byte loads per lane
  constexpr bool EP = F16 && NASR_FWD_EPOCH;  // hand-off validated by the payload's own epoch bits
  float* red = lds + LDS_RED;
  float* side = lds + LDS_SIDE;
  float* xgb = lds + LDS_XGB;
  unsigned* info = reinterpret_cast<unsigned*>(lds + LDS_INFO);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  unsigned xcc, member;
  if (!join_group(ctl, sticky, gm.fault, info, xcc, member)) return;
  const int T = gm.T, Bp = gm.Bp, Hp = gm.Hp, D = gm.D;
  const int NGD = 8 / D, d = (int)xcc / NGD, grp = (int)xcc % NGD;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  // the memory wave shares a SIMD with one of the MFMA waves: it only gets the issue slots that wave leaves free
  if (w < 4) __builtin_amdgcn_s_setprio(3);

  // this wave's slice of the recurrent matrix, resident for the whole launch
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  float wreg[F16 ? 1 : KW];
  h4 w1[F16 ? KW / 4 : 1], w2[F16 ? KW / 4 : 1];
  float oscale = 1.f;                               // F16: 2^-14 / column scale of this lane's output column
  if (w < 4) {
    if constexpr (F16) {
      const h4* wp = reinterpret_cast<const h4*>(Upf) + ((((size_t)d * 32 + member) * 4 + w) * (KW / 4) * 2) * 64 + lane;
#pragma unroll
      for (int i = 0; i < KW / 4; ++i) { w1[i] = wp[(size_t)(2 * i) * 64]; w2[i] = wp[(size_t)(2 * i + 1) * 64]; }
      const int bp = lane >> 2, g = lane & 3;
      oscale = (bp < NU ? cinv[(size_t)d * 4 * gm.Hp + 4 * (NU * (int)member + bp) + g] : 1.f) * (1.f / 16384.f);
    } else {
      const float* wp = Upf + ((((size_t)d * 32 + member) * 4 + w) * KW) * 64 + lane;
#pragma unroll
      for (int i = 0; i < KW; ++i) wreg[i] = wp[(size_t)i * 64];
    }
  }
  float* ghx = hx + (size_t)xcc * 2 * Hp * 4;
  gu32* gflag = (gu32*)(ctl->flags + xcc * 128);

  // cell decomposition, the same in every wave: lane = 16*q + u -> utterance slot q, unit NU*member + u
  const int q = lane >> 4, u = lane & 15;
  const bool lane_ok = u < NU;
  const int j = NU * (int)member + u;
  const int jcl = j < Hp ? j : Hp - 1;
  const int hidx = (j >> 2) * 16 + q * 4 + (j & 3);
  bool aborted = false;
  Stamps stp;
  stp.start(w == 0);
  // epoch of the use of exchange buffer (s & 1) that step s of round rd is: uses alternate 1, 0, 1, ... from a cleared buffer
  auto epoch_of = [&](int rd, int s) -> unsigned { return (unsigned)(rd * ((T + 1 - (s & 1)) >> 1) + (s >> 1) + 1) & 1u; };

  for (int rd = 0; rd < gm.rounds; ++rd) {
    const int b0 = (rd * NGD + grp) * gm.ub;
    if (b0 >= Bp) continue;                       // uniform over the group
    const int b = b0 + q;
    const bool rowok = lane_ok && q < gm.ub && b < Bp;
    const int bcl = b < Bp ? b : Bp - 1;
    const int len = rowok ? seq_len[b] : 0;
    const unsigned tagbase = (unsigned)(rd * T);
    float c = 0.f;
    // memory wave: gate pre-activations of frame(s), loaded a step ahead.  UNCONDITIONAL loads (lanes without a valid
    // frame read a clamped address, the value is never used): a load under a divergent branch makes hipcc copy its
    // result into loop-carried registers right away, i.e. wait for it on the spot.
    // element offsets fit 32 bits (T*Bp*D*N4 < 2^32 is checked by the launcher): cheap address arithmetic
    const unsigned xoff = (unsigned)bcl * (unsigned)DN + (unsigned)(d * N4 + 4 * jcl), xstep = (unsigned)Bp * (unsigned)DN;
    auto frame_of = [&](int s) { return (rowok && s < len) ? (d ? len - 1 - s : s) : 0; };
    auto load_xg = [&](int s) { return *reinterpret_cast<const f32x4*>(gates + (xoff + (unsigned)frame_of(s) * xstep)); };
    f32x4 xg_pf = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (w == 4) xg_pf = load_xg(0);
    // results of step s-1 (activations, c, h) leave through the memory wave one step later
    auto store_side = [&](int s) {   // memory wave, after the barrier that follows the cell update of step s
      if (rowok) {
        if (s < len) {
          const float* sp = side + ((s & 1) * 64 + lane) * 8;
          const f32x4 act = *reinterpret_cast<const f32x4*>(sp);
          const float2 ch = *reinterpret_cast<const float2*>(sp + 4);
          const unsigned r = (unsigned)((d ? (len - 1 - s) : s) * Bp + b);
          *reinterpret_cast<f32x4*>(gates + (r * (unsigned)DN + (unsigned)(d * N4 + 4 * j))) = act;
          const unsigned oc = r * (unsigned)DH + (unsigned)(d * Hp + j);
          cbuf[oc] = ch.x;
          out[oc] = ch.y;
        } else {
          out[(unsigned)(s * Bp + b) * (unsigned)DH + (unsigned)(d * Hp + j)] = 0.f;   // frame s is past seq_len in both directions
        }
      }
    };
    for (int s = 0; s < T; ++s) {
      const int par = s & 1;
      stp.mark(0);
      bool ok = true;
      if (w < 4) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 P[NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i) P[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (EP) {
          // 1+2. ONE round trip: every published word carries the epoch of its buffer's current use in bit 30 (a bit the
          // scaled low part of h never sets, see the cell update), so the payload loads themselves say whether the 8
          // producers of this wave's K quarter have published h_{s-1}.  A stale word (a producer that is late) is loaded
          // again by its own lane only.  The previous use of the buffer (h_{s-3}) has the other epoch; nobody can be a
          // whole use ahead, because publishing h_{s+1} takes everybody's h_s.
          // Step 0 of a later round reuses the buffers of the round before: there (only) the flags are the barrier.
          if (s == 0 && rd > 0) ok = poll_ge(gflag + w * 8 + (lane & 7), lane < 8, tagbase);
          if (s == gm.inject && member == 0) ok = false;
          stp.mark(1);
          if (s > 0 && ok) {
            if (w > 0) {   // this CU's cell wave has published h_{s-1}: the others have, or are about to
              const unsigned want = tagbase + (unsigned)s;
              for (unsigned n = 0; n < (1u << 26) && (int)(*(lds_vu32*)(info + 5) - want) < 0; ++n) __builtin_amdgcn_s_sleep(1);
            }
            if (NASR_EP_DELAY) __builtin_amdgcn_s_sleep(NASR_EP_DELAY);
            const float* src = ghx + (size_t)((s - 1) & 1) * Hp * 4 + ((size_t)w * KW * 4 + (size_t)lane * 4);
            const unsigned eexp = epoch_of(rd, s - 1) << 30;
            const bool has0 = NCH >= 16 || lane < 4 * NCH, has1 = NJ == 2 && lane < 4 * (NCH - 16);
            bool need = true;
            ok = false;
            for (unsigned n = 0; n < SPIN_BUDGET; ++n) {
              if (need) {
                if (has0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(P[0]) : "v"(src) : "memory");
                if constexpr (NJ == 2)
                  if (has1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=&v"(P[NJ - 1]) : "v"(src) : "memory");
              }
              if constexpr (NJ == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(P[0]), "+v"(P[NJ - 1]) : : "memory");
              else asm volatile("s_waitcnt vmcnt(0)" : "+v"(P[0]) : : "memory");
              unsigned bad = 0;
              if (has0)
                bad |= (__float_as_uint(P[0][0]) ^ eexp) | (__float_as_uint(P[0][1]) ^ eexp) | (__float_as_uint(P[0][2]) ^ eexp) |
                       (__float_as_uint(P[0][3]) ^ eexp);
              if constexpr (NJ == 2)
                if (has1)
                  bad |= (__float_as_uint(P[NJ - 1][0]) ^ eexp) | (__float_as_uint(P[NJ - 1][1]) ^ eexp) |
                         (__float_as_uint(P[NJ - 1][2]) ^ eexp) | (__float_as_uint(P[NJ - 1][3]) ^ eexp);
              need = (bad & 0x40000000u) != 0;
              if (!__any(need)) { ok = true; break; }
#if NASR_PSTAMP
              if (stp.on) stp.acc[11] += 1;      // extra attempts of wave 0
#endif
            }
          }
        } else {
        // 1. the 8 producers of this wave's K quarter have published h_{s-1} (and finished with h_{s-2})
        ok = poll_ge(gflag + w * 8 + (lane & 7), lane < 8, tagbase + (unsigned)s) && !(s == gm.inject && member == 0);
        stp.mark(1);
        if (s > 0 && ok) {
          // 2. h_{s-1} of this K quarter: chunk = (unit/4)*4 + utterance, 16 B = 4 consecutive units
          const float* src = ghx + (size_t)((s - 1) & 1) * Hp * 4 + ((size_t)w * KW * 4 + (size_t)lane * 4);
          if constexpr (NCH == 32) {
            asm volatile(
                "global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                : "=&v"(P[0]), "=&v"(P[1])
                : "v"(src)
                : "memory");
          } else if constexpr (NJ == 2) {   // the second load covers chunks 16 .. NCH-1 only
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(P[0]) : "v"(src) : "memory");
            if (lane < 4 * (NCH - 16)) asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=&v"(P[1]) : "v"(src) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(P[0]), "+v"(P[1]) : : "memory");
          } else {
            if (lane < 4 * NCH)
              asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(P[0]) : "v"(src) : "memory");
          }
        }
        }
        if (w == 0 && lane == 0) *(lds_vu32*)(info + 4) = tagbase + (unsigned)s + 1u;   // MFMA phase starts (ds_write)
        stp.mark(2);
        if (s > 0 && ok) {
          // 3. acc[.][utt] (16 units x 4 gates) += h[utt][k] * U[k][cols]; A broadcast from block bb%16
          if constexpr (F16) {
            // the producer published each h as its two fp16 parts in one word (low half a1, high half a2): gather the
            // four a1 / a2 of this lane's chunk with two byte permutes each
            h4 a1[NJ], a2[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
              typedef unsigned u2 __attribute__((ext_vector_type(2)));
              const unsigned d0 = __float_as_uint(P[i][0]), d1 = __float_as_uint(P[i][1]);
              const unsigned d2 = __float_as_uint(P[i][2]), d3 = __float_as_uint(P[i][3]);
              const u2 lo = {__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u)};
              const u2 hi = {__builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u)};
              a1[i] = __builtin_bit_cast(h4, lo);
              a2[i] = __builtin_bit_cast(h4, EP ? (u2){hi[0] & 0xBFFFBFFFu, hi[1] & 0xBFFFBFFFu} : hi);
            }
            static_for<0, NCH>([&](auto bbc) {
              constexpr int bb = decltype(bbc)::value;
              acc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(a2[bb / 16], w1[bb], acc[0], 4, bb % 16, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1[bb / 16], w2[bb], acc[1], 4, bb % 16, 0);
              acc[2] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1[bb / 16], w1[bb], acc[2], 4, bb % 16, 0);
            });
          } else {
            static_for<0, NCH>([&](auto bbc) {
              constexpr int bb = decltype(bbc)::value;
              static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                acc[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[bb / 16][r], wreg[bb * 4 + r], acc[r], 4, bb % 16, 0);
              });
            });
          }
        }
        if constexpr (EP) acc[0] *= 8.f;          // the low parts travel as h2 / 8 (exact): see the cell update
        f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        if constexpr (F16) sum *= oscale;
        float* rw = red + ((par * 4 + w) * 4) * 64 + lane;
        rw[0] = sum[0]; rw[64] = sum[1]; rw[128] = sum[2]; rw[192] = sum[3];
        if (!ok) info[2 + par] = 1;
      } else {
        // memory wave: hand the gates of frame(s) to the cell wave, then - once the MFMA waves have their operands and
        // the hand-off traffic is over - load the gates of frame(s+1) and store the results of step s-1.  (Measured
        // alternatives, forward us/step with stamps: this 1.645; right after the cell update, i.e. beside the
        // hand-off, 1.74; in an MFMA wave 1.64; in the cell wave 1.69.)
        *reinterpret_cast<f32x4*>(xgb + (par * 64 + lane) * 4) = xg_pf;
        const unsigned want = tagbase + (unsigned)s + 1u;
        for (unsigned n = 0; n < (1u << 26) && (int)(*(lds_vu32*)(info + 4) - want) < 0; ++n)
          __builtin_amdgcn_s_sleep(1);
        if (s + 1 < T) xg_pf = load_xg(s + 1);     // the load first: its wait must not sit behind the stores' acknowledgements
        if (s > 0) store_side(s - 1);
      }
      stp.mark(3);
#if NASR_PSTAMP
      if (lane == 0) info[8 + w] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
      stp.mark(4);
#if NASR_PSTAMP
      if (stp.on) for (int i = 1; i < 5; ++i) stp.acc[6 + i] += info[8 + i] - info[8];   // arrival of wave i after wave 0 (mod 2^32)
#endif
      // (the abort word is tested at the END of the iteration: tested here, its LDS round trip sat in front of the cell
      //  update's own reads on the cell wave's chain; an aborted step publishes garbage, which nobody uses)
      const unsigned abort_word = info[2 + par];
      if (w == 0) {
        // 4. cell update: lane = (utterance q, unit u)
        const bool valid = rowok && s < len;
        float h = 0.f;
        if (lane_ok) {
          const float* rr = red + (par * 4 * 4 + q) * 64 + 4 * u;
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(rr);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(rr + 256);
          const f32x4 g2 = *reinterpret_cast<const f32x4*>(rr + 512);
          const f32x4 g3 = *reinterpret_cast<const f32x4*>(rr + 768);
          const f32x4 xg = *reinterpret_cast<const f32x4*>(xgb + (par * 64 + lane) * 4);
          const f32x4 pre = xg + ((g0 + g1) + (g2 + g3));
          f32x4 act;
          act.x = psig(pre.x);
          act.y = ptanh(pre.y);
          act.z = psig(pre.z + fb);
          act.w = psig(pre.w);
          if (valid) {
            c = c * act.z + act.x * act.y;
            h = ptanh(c) * act.w;
          }
          if constexpr (F16) {
            // split h (|h| < 1) into its two fp16 parts of h * 2^14 HERE, once, instead of in every consumer wave of the
            // group (32 CUs x 4 waves redid these five operations per value on their MFMA chain): same arithmetic, same bits
            const float v = h * 16384.f;
            const _Float16 h1 = (_Float16)v;
            // EP: the low part as (v - h1) / 8: |v - h1| <= 4, so its fp16 exponent field stays below 16 and bit 14 of the
            // half - bit 30 of the word - is free for the epoch of this use of the buffer (the consumers multiply the
            // h2 U1 product by 8; both scalings are exact)
            const _Float16 h2 = (_Float16)(EP ? (v - (float)h1) * 0.125f : (v - (float)h1));
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, h1) |
                                ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16) | (EP ? epoch_of(rd, s) << 30 : 0u);
            ghx[(size_t)par * Hp * 4 + hidx] = __uint_as_float(pk);
          } else {
            ghx[(size_t)par * Hp * 4 + hidx] = h;          // plain store: lands in this XCD's L2
          }
          float* sp = side + (par * 64 + lane) * 8;
          *reinterpret_cast<f32x4*>(sp) = act;
          *reinterpret_cast<float2*>(sp + 4) = make_float2(c, h);
        }
        stp.mark(5);
        if constexpr (EP) {
          // no acknowledgement to wait for: the words validate themselves.  The other waves of this CU start loading.
          if (lane == 0) *(lds_vu32*)(info + 5) = tagbase + (unsigned)s + 1u;
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and is acknowledged before the flag goes out
        }
        stp.mark(6);
        if (lane == 0) *(ctl->flags + xcc * 128 + member) = tagbase + (unsigned)s + 1u;   // (EP: the round barrier only)
      }
      if (abort_word) { aborted = true; break; }
    }
    if (aborted) break;
    __syncthreads();                  // the last cell update's results are in LDS
    if (w == 4) store_side(T - 1);
  }
  stp.flush(ctl, xcc * 32 + member, xcc == 0 && member == 0);
  if (aborted && tid == 0) raise_error(ctl, sticky, gm.fault, 1u);
}

// ------------------------------------------------------------------ BPTT
template <int NU>
__global__ __launch_bounds__(320, 1) void lstm_persist_bwd_kernel(
    const float* __restrict__ Upb, const float* __restrict__ gates, float* dgbuf, const float* __restrict__ cbuf,
    const float* __restrict__ dout, const int* __restrict__ seq_len,
    float* px,                      // [8 groups][2 parity][32 consumers][32 producers][16 units][4 utterances]
    PersistCtl* ctl, unsigned* sticky, PersistGeom gm,
    // (or NULL) largest |dG| of every frame row over this CU's gate columns [D*32 parts][T*Bp], and of every gate column
    // over this group's utterances [8/D parts][D*4Hp]: the partial maxima behind the operand scales of the GEMMs that
    // read dG (gemm_tph.hip), taken by the memory wave from the values it stores anyway - no pass over dG afterwards
    float* __restrict__ rowpart, float* __restrict__ colpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KW = 8 * NU;                  // output units per wave
  constexpr int NOG = (KW + 63) / 64;         // 64-unit output groups per wave
  constexpr int NC = 4 * NU;                  // gate columns this CU contracts
  constexpr int NV = (NC + 15) / 16;          // A registers
  float* adg = lds + LDS_ADG;
  float* pfb = lds + LDS_PFB;
  unsigned* info = reinterpret_cast<unsigned*>(lds + LDS_INFO);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  unsigned xcc, member;
  if (!join_group(ctl, sticky, gm.fault, info, xcc, member)) return;
  const int T = gm.T, Bp = gm.Bp, Hp = gm.Hp, D = gm.D;
  const int NGD = 8 / D, d = (int)xcc / NGD, grp = (int)xcc % NGD;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  // the memory wave shares a SIMD with one of the MFMA waves: it only gets the issue slots that wave leaves free
  if (w < 4) __builtin_amdgcn_s_setprio(3);

  float wreg[NOG * NC];
  if (w < 4) {
    const float* wp = Upb + ((((size_t)d * 32 + member) * 4 + w) * (NOG * NC)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NOG * NC; ++i) wreg[i] = wp[(size_t)i * 64];
  }
  float* gpx = px + (size_t)xcc * 2 * 32 * 32 * 64;
  gu32* gflag = (gu32*)(ctl->flags + xcc * 128);

  const int q = lane & 3, u = lane >> 2;        // cell decomposition: lane = 4*u + q, the order of an exchange row
  const bool lane_ok = u < NU;
  const int j = NU * (int)member + u;
  const int jcl = j < Hp ? j : Hp - 1;
  bool aborted = false;
  Stamps stp;
  stp.start(w == 0);
  constexpr bool EP = NASR_BWD_EPOCH != 0;
  f32x4 cmax = (f32x4){0.f, 0.f, 0.f, 0.f};   // memory wave: running column maxima of this lane's (unit, utterance slot)
  // epoch of the use of exchange buffer (k & 1) that step k of round rd is (uses alternate 1, 0, 1, ... from a cleared buffer)
  // (the rounds a group runs come first - b0 grows with rd - so rd counts its uses.  The buffer is CLEARED before every
  //  launch: the bit pattern the sums carry is then a function of the launch's shape alone, and two runs of the same
  //  step give the same bits)
  auto epoch_of = [&](int rd, int k) -> unsigned { return (unsigned)(rd * ((T + 1 - (k & 1)) >> 1) + (k >> 1) + 1) & 1u; };

  for (int rd = 0; rd < gm.rounds; ++rd) {
    const int b0 = (rd * NGD + grp) * gm.ub;
    if (b0 >= Bp) continue;
    const int b = b0 + q;
    const bool rowok = lane_ok && q < gm.ub && b < Bp;
    const int bcl = b < Bp ? b : Bp - 1;
    const int len = rowok ? seq_len[b] : 0;
    const unsigned tagbase = (unsigned)(rd * T);
    float dc = 0.f;
    // memory wave: the operands of step s (activations, c, c of the frame the forward pass visited before, dOut),
    // loaded two steps ahead, unconditionally (clamped address on lanes without a valid frame, value unused there)
    f32x4 pa = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 pb = (f32x4){0.f, 0.f, 0.f, 0.f};   // cc, cpv, dOut, -
    auto prefetch = [&](int s) {
      const bool v = rowok && s < len;
      const int tb = v ? (d ? (len - 1 - s) : s) : 0;
      const unsigned r = (unsigned)(tb * Bp + bcl);
      const unsigned rp = (v && s > 0) ? (d ? r + (unsigned)Bp : r - (unsigned)Bp) : r;
      const unsigned cj = (unsigned)(d * Hp + jcl);
      pa = *reinterpret_cast<const f32x4*>(gates + (r * (unsigned)DN + (unsigned)(d * N4 + 4 * jcl)));
      pb.x = cbuf[r * (unsigned)DH + cj];
      pb.y = cbuf[rp * (unsigned)DH + cj];
      pb.z = dout[r * (unsigned)DH + cj];
    };
    auto hand_over = [&](int k) {   // operands of step k -> LDS, read by the cell wave after the next barrier
      float* dst = pfb + ((k & 1) * 64 + lane) * 8;
      *reinterpret_cast<f32x4*>(dst) = pa;
      *reinterpret_cast<f32x4*>(dst + 4) = pb;
    };
    auto store_dg = [&](int k) {    // memory wave: frame-indexed dG of step k (zero at masked frames) from the A image
      float m = 0.f;
      unsigned row = 0;
      if (rowok) {
        const int s = T - 1 - k;
        const float* ad = adg + (k & 1) * 256 + 16 * u + q;
        const f32x4 dg = (f32x4){ad[0], ad[4], ad[8], ad[12]};
        row = (unsigned)(((s < len) ? (d ? (len - 1 - s) : s) : s) * Bp + b);
        *reinterpret_cast<f32x4*>(dgbuf + (row * (unsigned)DN + (unsigned)(d * N4 + 4 * j))) = dg;
        const f32x4 ab = (f32x4){fabsf(dg.x), fabsf(dg.y), fabsf(dg.z), fabsf(dg.w)};
        cmax = (f32x4){fmaxf(cmax.x, ab.x), fmaxf(cmax.y, ab.y), fmaxf(cmax.z, ab.z), fmaxf(cmax.w, ab.w)};
        m = fmaxf(fmaxf(ab.x, ab.y), fmaxf(ab.z, ab.w));
      }
      if (rowpart) {   // max over the CU's units of this utterance slot: lanes 4u + q, u = 0..15 (order-independent: exact)
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x124, 0xf, 0xf, false)));   // row_ror:4
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x128, 0xf, 0xf, false)));   // row_ror:8
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < 4 && rowok) rowpart[(size_t)(d * 32 + (int)member) * ((size_t)T * Bp) + row] = m;
      }
    };
    if (w == 4) {
      prefetch(T - 1);
      hand_over(0);
      if (T > 1) prefetch(T - 2);
    }
    __syncthreads();
    for (int k = 0; k < T; ++k) {
      const int s = T - 1 - k, par = k & 1;
      stp.mark(0);
      if (w == 0) {
        // 1. every wave of every producer has published its partial sums of step k-1
        bool ok = false;
        if (!EP || (k == 0 && rd > 0)) {   // (EP: the flags are only the barrier between rounds, which reuse the buffers)
          const unsigned want = tagbase + (unsigned)k;
          for (unsigned n = 0; n < SPIN_BUDGET && !ok; ++n) {
            const unsigned v0 = __hip_atomic_load(gflag + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned v1 = __hip_atomic_load(gflag + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = __all((int)(v0 - want) >= 0 && (int)(v1 - want) >= 0);
          }
        } else {
          ok = true;
        }
        if (k == gm.inject && member == 0) ok = false;
        stp.mark(1);
        const bool valid = rowok && s < len;
        const float* pf = pfb + (par * 64 + lane) * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(pf);
        const f32x4 o = *reinterpret_cast<const f32x4*>(pf + 4);
        float dhs = o.z;
        if (k > 0 && ok) {
          const float* src = gpx + ((size_t)(((k - 1) & 1) * 32 + member) * 32) * 64 + lane;
          float pv[32];
#pragma unroll
          for (int p = 0; p < 32; ++p) pv[p] = 0.f;
#define NASR_LD8(g8)                                                                                               \
  asm volatile(                                                                                                    \
      "global_load_dword %0, %8, off sc1\n\tglobal_load_dword %1, %8, off offset:256 sc1\n\t"                       \
      "global_load_dword %2, %8, off offset:512 sc1\n\tglobal_load_dword %3, %8, off offset:768 sc1\n\t"            \
      "global_load_dword %4, %8, off offset:1024 sc1\n\tglobal_load_dword %5, %8, off offset:1280 sc1\n\t"          \
      "global_load_dword %6, %8, off offset:1536 sc1\n\tglobal_load_dword %7, %8, off offset:1792 sc1"              \
      : "=&v"(pv[g8 + 0]), "=&v"(pv[g8 + 1]), "=&v"(pv[g8 + 2]), "=&v"(pv[g8 + 3]), "=&v"(pv[g8 + 4]),             \
        "=&v"(pv[g8 + 5]), "=&v"(pv[g8 + 6]), "=&v"(pv[g8 + 7])                                                    \
      : "v"(src + (size_t)(g8) * 64)                                                                               \
      : "memory")
          // EP: the sums themselves say whether they are those of step k-1 - bit 0 of every word is the epoch of this use
          // of the buffer; a lane that finds a stale word loads its 32 again (one round trip when everybody is on time,
          // and no acknowledgement wait or flag on the producers' side)
          const unsigned eexp = epoch_of(rd, k - 1);
          bool need = lane_ok;
          bool got = false;
          for (unsigned n = 0; n < (EP ? SPIN_BUDGET : 1u); ++n) {
            if (need) { NASR_LD8(0); NASR_LD8(8); NASR_LD8(16); NASR_LD8(24); }
            // one wait for all 32 loads; naming every destination keeps hipcc from touching them before it
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]),
                           "+v"(pv[7]), "+v"(pv[8]), "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]),
                           "+v"(pv[14]), "+v"(pv[15])
                         :
                         : "memory");
            asm volatile(""
                         : "+v"(pv[16]), "+v"(pv[17]), "+v"(pv[18]), "+v"(pv[19]), "+v"(pv[20]), "+v"(pv[21]), "+v"(pv[22]),
                           "+v"(pv[23]), "+v"(pv[24]), "+v"(pv[25]), "+v"(pv[26]), "+v"(pv[27]), "+v"(pv[28]), "+v"(pv[29]),
                           "+v"(pv[30]), "+v"(pv[31])
                         :
                         : "memory");
            if constexpr (EP) {
              unsigned bad;                   // bit 0: some word of this lane is not of epoch eexp (wave-uniform branch)
              if (eexp) {
                unsigned a_ = 1u;
#pragma unroll
                for (int p = 0; p < 32; ++p) a_ &= __float_as_uint(pv[p]);
                bad = ~a_;
              } else {
                unsigned o_ = 0u;
#pragma unroll
                for (int p = 0; p < 32; ++p) o_ |= __float_as_uint(pv[p]);
                bad = o_;
              }
              need = lane_ok && (bad & 1u) != 0;
              if (!__any(need)) { got = true; break; }
#if NASR_PSTAMP
              if (stp.on) stp.acc[11] += 1;      // extra attempts
#endif
            } else {
              got = true;
            }
          }
#undef NASR_LD8
          if (!got) ok = false;
          if (lane_ok) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int p = 0; p < 32; p += 4) { s0 += pv[p]; s1 += pv[p + 1]; s2 += pv[p + 2]; s3 += pv[p + 3]; }
            dhs += (s0 + s1) + (s2 + s3);
          }
        }
        stp.mark(2);
        // 2. gate derivatives of this CU's cells
        f32x4 dg = (f32x4){0.f, 0.f, 0.f, 0.f};
        float dcn = 0.f;
        if (valid) {
          const float tc = ptanh(o.x);
          const float dct = dc + dhs * a.w * (1.f - tc * tc);
          dg.x = dct * a.y * a.x * (1.f - a.x);
          dg.y = dct * a.x * (1.f - a.y * a.y);
          dg.z = dct * (s > 0 ? o.y : 0.f) * a.z * (1.f - a.z);
          dg.w = dhs * tc * a.w * (1.f - a.w);
          dcn = dct * a.z;
        }
        dc = dcn;
        if (lane_ok) {   // A image: adg[par][c = 4u+g][utterance q]
          float* ad = adg + par * 256 + 16 * u + q;
          ad[0] = dg.x; ad[4] = dg.y; ad[8] = dg.z; ad[12] = dg.w;
        }
        if (!ok) info[2 + par] = 1;
        stp.mark(3);
      } else if (w == 4) {
        if (k + 1 < T) hand_over(k + 1);     // loaded during the previous step's MFMA phase
      }
      __syncthreads();
      stp.mark(4);
      const unsigned abort_word = info[2 + par];   // tested at the end of the iteration (see the forward kernel)
      if (w == 4) {
        // memory wave, while the others are in their MFMA phase: dG of this step out, operands of step k+2 in
        if (k + 2 < T) prefetch(T - 3 - k);    // loads first: their wait must not sit behind the store's acknowledgement
        store_dg(k);
        if (abort_word) { aborted = true; break; }
        continue;
      }
      // 3. partial[utt][k'] = sum_c dG[utt][c] * U[k'][c] for this wave's output units, waves 0-3
      float av[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) av[v] = (64 * v + lane < 4 * NC) ? adg[par * 256 + 64 * v + lane] : 0.f;
      // NACC independent accumulator chains per output group: with NOG * NACC = 8 chains in flight the wave always has an
      // MFMA ready to issue (it runs at priority 3: beside another kernel's MFMA waves it keeps the pipe for its phase)
      constexpr int NACC = NASR_BWD_NACC;
      f32x4 acc[NOG][NACC];
#pragma unroll
      for (int og = 0; og < NOG; ++og)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[og][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
      static_for<0, NC>([&](auto cc_) {
        constexpr int c = decltype(cc_)::value;
        static_for<0, NOG>([&](auto ogc) {
          constexpr int og = decltype(ogc)::value;
          acc[og][c % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c / 16], wreg[og * NC + c], acc[og][c % NACC], 4, c % 16, 0);
        });
      });
      // 4. hand the partial rows to their consumers: unit k' -> consumer k'/NU, row element 4*(k'%NU) + utt
      const unsigned eb = epoch_of(rd, k);
#pragma unroll
      for (int og = 0; og < NOG; ++og) {
        f32x4 sum = acc[og][0] + acc[og][1];
        if constexpr (NACC == 4) sum = sum + (acc[og][2] + acc[og][3]);
        if constexpr (EP) {   // the last mantissa bit of every sum = the epoch of this use of the buffer
#pragma unroll
          for (int i = 0; i < 4; ++i) sum[i] = __uint_as_float((__float_as_uint(sum[i]) & ~1u) | eb);
        }
        const int kl = og * 64 + lane;
        if (kl < KW) {
          const int kk = w * KW + kl;
          float* dst = gpx + ((size_t)((par * 32 + kk / NU) * 32) + member) * 64 + 4 * (kk % NU);
          *reinterpret_cast<f32x4*>(dst) = sum;
        }
      }
      stp.mark(5);
      if constexpr (!EP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stp.mark(6);
      if (lane == 0) *(ctl->flags + xcc * 128 + member * 4 + w) = tagbase + (unsigned)k + 1u;
      if (abort_word) { aborted = true; break; }
    }
    if (aborted) break;
    __syncthreads();                  // pfb / adg are reused by the next round
  }
  if (w == 4 && colpart) {   // column maxima of this group's utterances: over the 4 utterance slots (lanes 4u + q), then out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = cmax[i];
      v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
      v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
      cmax[i] = v;
    }
    if (q == 0 && lane_ok) *reinterpret_cast<f32x4*>(colpart + ((size_t)grp * DN + (size_t)(d * N4 + 4 * j))) = cmax;
  }
  stp.flush(ctl, xcc * 32 + member, xcc == 0 && member == 0);
  if (aborted && tid == 0) raise_error(ctl, sticky, gm.fault, 1u);
}

// ------------------------------------------------------------------ launchers
bool persist_supported(int Hp) {
  const int NU = Hp / 32;
  return Hp % 64 == 0 && NU >= 2 && NU <= 16;   // every padded hidden size up to 512
}

static PersistGeom make_geom(const LstmDims& dm, bool bwd) {
  PersistGeom g;
  g.T = dm.T; g.Bp = dm.Bp; g.Hp = dm.Hp; g.D = dm.D;
  const int NGD = 8 / dm.D;
  g.ub = (dm.Bp + NGD - 1) / NGD;
  if (g.ub > 4) g.ub = 4;
  g.rounds = (dm.Bp + NGD * g.ub - 1) / (NGD * g.ub);
  const char* e = test_hook("NASR_PERSIST_FAULT");
  g.inject = (e && *e) ? atoi(e) : -1;
  const char* ek = test_hook("NASR_PERSIST_FAULT_KERNEL");     // "fwd" / "bwd": inject into that kernel only
  if (ek && *ek && ((ek[0] == 'b') != bwd)) g.inject = -1;
  g.fault = nullptr;
  return g;
}

// lean = true: the BPTT launches declare only the LDS they use, so that another kernel's workgroup can share their CUs.
// Only for Hp = 512, where the kernel's 5 x 220 VGPRs alone keep it at one workgroup per CU.
void persist_set_bwd_lean(bool lean) { g_bwd_lean = lean; }

size_t persist_dgmax_floats(int T, int Bp, int Hp, int D) { return (size_t)D * 32 * T * Bp + (size_t)(8 / D) * D * 4 * Hp; }

size_t persist_px_bytes() { return (size_t)8 * 2 * 32 * 32 * 64 * sizeof(float); }   // the BPTT kernel's exchange buffer

size_t persist_hx_bytes(int Hp) { return (size_t)8 * 2 * Hp * 4 * sizeof(float); }   // the forward kernel's part of xch

size_t persist_xch_floats(int Hp) {
  const size_t f = (size_t)8 * 2 * Hp * 4, b = (size_t)8 * 2 * 32 * 32 * 64;
  return f > b ? f : b;
}

hipError_t persist_prepare() {
  hipError_t e = hipSuccess;
#define NASR_PATTR(NUV)                                                                                             \
  if (e == hipSuccess)                                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_persist_fwd_kernel<NUV, false>),                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS_BYTES);                         \
  if (e == hipSuccess)                                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_persist_fwd_kernel<NUV, true>),                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS_BYTES);                         \
  if (e == hipSuccess)                                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_persist_bwd_kernel<NUV>),                           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS_BYTES);
  NASR_PATTR(2) NASR_PATTR(4) NASR_PATTR(6) NASR_PATTR(8) NASR_PATTR(10) NASR_PATTR(12) NASR_PATTR(14) NASR_PATTR(16)
#undef NASR_PATTR
  return e;
}

void launch_lstm_persist_fwd(const LstmDims& dm, const float* Upf, const float* cinv, float* gates, float* cbuf,
                             float* out, const int* seq_len, float* xch, PersistCtl* ctl, unsigned* sticky, float* fault,
                             float forget_bias, hipStream_t st, bool ctl_zeroed) {
  PersistGeom gm = make_geom(dm, false);
  gm.fault = fault;
  if (!ctl_zeroed) (void)hipMemsetAsync(ctl, 0, sizeof(PersistCtl), st);
  // epoch-validated hand-off: the exchange buffers start from epoch 0 (ctl_zeroed: the caller cleared them with *ctl)
  if (NASR_FWD_EPOCH && cinv && !ctl_zeroed) (void)hipMemsetAsync(xch, 0, persist_hx_bytes(dm.Hp), st);
  dim3 grid(256), block(320);
#define NASR_PF(NUV)                                                                                                  \
  if (cinv)                                                                                                           \
    hipLaunchKernelGGL((lstm_persist_fwd_kernel<NUV, true>), grid, block, PERSIST_LDS_BYTES, st, Upf, gates, cbuf, out,   \
                       seq_len, xch, ctl, sticky, gm, forget_bias, cinv);                                               \
  else                                                                                                                \
    hipLaunchKernelGGL((lstm_persist_fwd_kernel<NUV, false>), grid, block, PERSIST_LDS_BYTES, st, Upf, gates, cbuf, out,  \
                       seq_len, xch, ctl, sticky, gm, forget_bias, cinv)
  switch (dm.Hp / 32) {
    case 2: NASR_PF(2); break;
    case 4: NASR_PF(4); break;
    case 6: NASR_PF(6); break;
    case 8: NASR_PF(8); break;
    case 10: NASR_PF(10); break;
    case 12: NASR_PF(12); break;
    case 14: NASR_PF(14); break;
    default: NASR_PF(16); break;
  }
#undef NASR_PF
}

void launch_lstm_persist_bwd(const LstmDims& dm, const float* Upb, const float* gates, float* dgbuf, const float* cbuf,
                             const float* dout, const int* seq_len, float* xch, PersistCtl* ctl, unsigned* sticky,
                             float* fault, hipStream_t st, bool ctl_zeroed, float* rowpart, float* colpart) {
  PersistGeom gm = make_geom(dm, true);
  gm.fault = fault;
  if (!ctl_zeroed) (void)hipMemsetAsync(ctl, 0, sizeof(PersistCtl), st);
  // epoch-validated hand-off: the exchange buffer starts from epoch 0 (ctl_zeroed: the caller cleared it with *ctl)
  if (NASR_BWD_EPOCH && !ctl_zeroed) (void)hipMemsetAsync(xch, 0, persist_px_bytes(), st);
  dim3 grid(256), block(320);
  // (lean only at Hp = 512, where the kernel's 5 x 220 VGPRs alone keep it at one workgroup per CU)
  const int bwd_lds = (g_bwd_lean && dm.Hp == 512) ? PERSIST_LDS_LEAN : PERSIST_LDS_BYTES;
#define NASR_PB(NUV)                                                                                                   \
  hipLaunchKernelGGL((lstm_persist_bwd_kernel<NUV>), grid, block, bwd_lds, st, Upb, gates, dgbuf, cbuf, dout, \
                     seq_len, xch, ctl, sticky, gm, rowpart, colpart)
  switch (dm.Hp / 32) {
    case 2: NASR_PB(2); break;
    case 4: NASR_PB(4); break;
    case 6: NASR_PB(6); break;
    case 8: NASR_PB(8); break;
    case 10: NASR_PB(10); break;
    case 12: NASR_PB(12); break;
    case 14: NASR_PB(14); break;
    default: NASR_PB(16); break;
  }
#undef NASR_PB
}

}  // namespace nasr
