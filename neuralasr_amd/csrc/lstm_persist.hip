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
//   * forward step: each wave waits for the 8 producers of its K quarter (one flag word each), loads their h
//     (2 x 16 B per lane), 128 MFMAs, 4-wave LDS reduction, wave 0 does the 64 cell updates (c stays in a
//     register for the whole sequence), publishes h and the flag.
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

#include <type_traits>
#include <utility>

namespace nasr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ float psig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ptanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

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
__device__ __forceinline__ void raise_error(PersistCtl* ctl, unsigned* sticky, unsigned code) {
  atomicOr(&ctl->error, code);
  if (sticky) __hip_atomic_store(sticky, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible, never cleared by a launch
}

__device__ __forceinline__ bool join_group(PersistCtl* ctl, unsigned* sticky, unsigned* info, unsigned& xcc, unsigned& member) {
  if (threadIdx.x == 0) {
    const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
    info[0] = x;
    info[1] = x < 8 ? atomicAdd(&ctl->xcc_count[x], 1u) : 0xffffu;
    info[2] = 0;
    info[3] = 0;
  }
  __syncthreads();
  xcc = info[0];
  member = info[1];
  if (xcc >= 8 || member >= 32) {
    if (threadIdx.x == 0) raise_error(ctl, sticky, 2u);
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
__global__ __launch_bounds__(256) void repack_persist_kernel(const float* __restrict__ U, float* __restrict__ Upf,
                                                             float* __restrict__ Upb, int Hp) {
  const int NU = Hp / 32, KW = 8 * NU, N4 = 4 * Hp;
  const int NOG = (KW + 63) / 64, KB = NOG * 4 * NU;
  const int64_t nf = (int64_t)32 * 4 * KW * 64, nb = (int64_t)32 * 4 * KB * 64;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nf + nb; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < nf) {
      const int lane = e & 63;
      int64_t x = e >> 6;
      const int idx = (int)(x % KW); x /= KW;
      const int w = (int)(x & 3), m = (int)(x >> 2);
      const int bp = lane >> 2, g = lane & 3;
      Upf[e] = bp < NU ? U[(size_t)(w * KW + idx) * N4 + 4 * (NU * m + bp) + g] : 0.f;
    } else {
      const int64_t eb = e - nf;
      const int lane = eb & 63;
      int64_t x = eb >> 6;
      const int idx = (int)(x % KB); x /= KB;
      const int w = (int)(x & 3), m = (int)(x >> 2);
      const int og = idx / (4 * NU), c = idx % (4 * NU);
      const int kl = og * 64 + lane;
      Upb[eb] = kl < KW ? U[(size_t)(w * KW + kl) * N4 + 4 * NU * m + c] : 0.f;
    }
  }
}

size_t persist_image_floats(int Hp, bool bwd) {
  const int NU = Hp / 32, KW = 8 * NU, NOG = (KW + 63) / 64;
  return (size_t)32 * 4 * 64 * (bwd ? NOG * 4 * NU : KW);
}

void launch_repack_persist(const float* U, float* Upf, float* Upb, int Hp, hipStream_t st) {
  hipLaunchKernelGGL(repack_persist_kernel, dim3(1024), dim3(256), 0, st, U, Upf, Upb, Hp);
}

// ------------------------------------------------------------------ forward
struct PersistGeom {
  int T, Bp, Hp, D;
  int ub;        // utterance rows per group and round (<= 4)
  int rounds;    // passes over the batch (weights stay in registers)
};

constexpr int PERSIST_LDS_BYTES = 96 * 1024;   // > half of the CU's 160 KB: one workgroup per CU

// LDS map (floats): red [2][4][4][64] | adg [2][256] | info[8]
constexpr int LDS_RED = 0, LDS_ADG = 2 * 4 * 4 * 64, LDS_INFO = LDS_ADG + 2 * 256;

template <int NU>
__global__ __launch_bounds__(256, 1) void lstm_persist_fwd_kernel(
    const float* __restrict__ Upf,   // [D] images
    float* gates, float* cbuf, float* out, const int* __restrict__ seq_len,
    float* hx,                       // [8 groups][2 parity][Hp*4]
    PersistCtl* ctl, unsigned* sticky, PersistGeom gm, float fb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KW = 8 * NU;                 // units (= MFMAs) per wave
  constexpr int NCH = 2 * NU;                // 16-byte chunks (4 units x 1 utterance) per wave and utterance
  constexpr int NJ = (NCH + 15) / 16;        // 16-byte loads per lane
  float* red = lds + LDS_RED;
  unsigned* info = reinterpret_cast<unsigned*>(lds + LDS_INFO);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  unsigned xcc, member;
  if (!join_group(ctl, sticky, info, xcc, member)) return;
  const int T = gm.T, Bp = gm.Bp, Hp = gm.Hp, D = gm.D;
  const int NGD = 8 / D, d = (int)xcc / NGD, grp = (int)xcc % NGD;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;

  // this wave's slice of the recurrent matrix, resident for the whole launch
  float wreg[KW];
  {
    const float* wp = Upf + ((((size_t)d * 32 + member) * 4 + w) * KW) * 64 + lane;
#pragma unroll
    for (int i = 0; i < KW; ++i) wreg[i] = wp[(size_t)i * 64];
  }
  float* ghx = hx + (size_t)xcc * 2 * Hp * 4;
  gu32* gflag = (gu32*)(ctl->flags + xcc * 128);

  // cell lanes (wave 0): lane = 16*q + u -> utterance slot q, unit NU*member + u
  const int q = lane >> 4, u = lane & 15;
  const bool cell_lane = w == 0 && u < NU;
  const int j = NU * (int)member + u;
  const int hidx = (j >> 2) * 16 + q * 4 + (j & 3);
  bool aborted = false;

  for (int rd = 0; rd < gm.rounds; ++rd) {
    const int b0 = (rd * NGD + grp) * gm.ub;
    if (b0 >= Bp) continue;                       // uniform over the group
    const int b = b0 + q;
    const bool rowok = cell_lane && q < gm.ub && b < Bp;
    const int len = rowok ? seq_len[b] : 0;
    const unsigned tagbase = (unsigned)(rd * T);
    float c = 0.f;
    float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rowok && 0 < len) {
      const int tb = d ? len - 1 : 0;
      xg = *reinterpret_cast<const float4*>(gates + ((size_t)tb * Bp + b) * DN + d * N4 + 4 * j);
    }
    for (int s = 0; s < T; ++s) {
      const int par = s & 1;
      // 1. the 8 producers of this wave's K quarter have published h_{s-1} (and finished with h_{s-2})
      const bool ok = poll_ge(gflag + w * 8 + (lane & 7), lane < 8, tagbase + (unsigned)s);
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (s > 0 && ok) {
        // 2. h_{s-1} of this K quarter: chunk = (unit/4)*4 + utterance, 16 B = 4 consecutive units
        f32x4 P[NJ];
        const float* src = ghx + (size_t)((s - 1) & 1) * Hp * 4 + ((size_t)w * KW * 4 + (size_t)lane * 4);
        if constexpr (NJ == 2) {
          asm volatile(
              "global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
              : "=&v"(P[0]), "=&v"(P[1])
              : "v"(src)
              : "memory");
        } else {
          P[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (lane < 4 * NCH)
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(P[0]) : "v"(src) : "memory");
        }
        // 3. acc[.][utt] (16 units x 4 gates) += h[utt][k] * U[k][cols]; A broadcast from block bb%16
        static_for<0, NCH>([&](auto bbc) {
          constexpr int bb = decltype(bbc)::value;
          static_for<0, 4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            acc[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[bb / 16][r], wreg[bb * 4 + r], acc[r], 4, bb % 16, 0);
          });
        });
      }
      {
        const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        float* rw = red + ((par * 4 + w) * 4) * 64 + lane;
        rw[0] = sum[0]; rw[64] = sum[1]; rw[128] = sum[2]; rw[192] = sum[3];
      }
      if (!ok) info[2 + par] = 1;
      __syncthreads();
      if (info[2 + par]) { aborted = true; break; }
      // 4. cell update: wave 0, lane = (utterance q, unit u)
      if (w == 0) {
        const bool valid = rowok && s < len;
        float h = 0.f;
        float4 act = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cell_lane) {
          const float* rr = red + (par * 4 * 4 + q) * 64 + 4 * u;
          const float4 g0 = *reinterpret_cast<const float4*>(rr);
          const float4 g1 = *reinterpret_cast<const float4*>(rr + 256);
          const float4 g2 = *reinterpret_cast<const float4*>(rr + 512);
          const float4 g3 = *reinterpret_cast<const float4*>(rr + 768);
          act.x = psig(xg.x + ((g0.x + g1.x) + (g2.x + g3.x)));
          act.y = ptanh(xg.y + ((g0.y + g1.y) + (g2.y + g3.y)));
          act.z = psig(xg.z + ((g0.z + g1.z) + (g2.z + g3.z)) + fb);
          act.w = psig(xg.w + ((g0.w + g1.w) + (g2.w + g3.w)));
          if (valid) {
            c = c * act.z + act.x * act.y;
            h = ptanh(c) * act.w;
          }
          ghx[(size_t)par * Hp * 4 + hidx] = h;          // plain store: lands in this XCD's L2
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and is acknowledged before the flag goes out
        if (lane == 0) *(ctl->flags + xcc * 128 + member) = tagbase + (unsigned)s + 1u;
        if (rowok) {
          if (valid) {
            const int tb = d ? (len - 1 - s) : s;
            const size_t r = (size_t)tb * Bp + b;
            *reinterpret_cast<float4*>(gates + r * DN + d * N4 + 4 * j) = act;
            cbuf[r * DH + d * Hp + j] = c;
            out[r * DH + d * Hp + j] = h;
          } else {
            out[((size_t)s * Bp + b) * DH + d * Hp + j] = 0.f;   // frame s is past seq_len in both directions
          }
          if (s + 1 < len) {
            const int tb = d ? (len - 2 - s) : s + 1;
            xg = *reinterpret_cast<const float4*>(gates + ((size_t)tb * Bp + b) * DN + d * N4 + 4 * j);
          }
        }
      }
    }
    if (aborted) break;
  }
  if (aborted && tid == 0) raise_error(ctl, sticky, 1u);
}

// ------------------------------------------------------------------ BPTT
template <int NU>
__global__ __launch_bounds__(256, 1) void lstm_persist_bwd_kernel(
    const float* __restrict__ Upb, const float* __restrict__ gates, float* dgbuf, const float* __restrict__ cbuf,
    const float* __restrict__ dout, const int* __restrict__ seq_len,
    float* px,                      // [8 groups][2 parity][32 consumers][32 producers][64]
    PersistCtl* ctl, unsigned* sticky, PersistGeom gm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KW = 8 * NU;                  // output units per wave
  constexpr int NOG = (KW + 63) / 64;         // 64-unit output groups per wave
  constexpr int NC = 4 * NU;                  // gate columns this CU contracts
  constexpr int NV = (NC + 15) / 16;          // A registers
  float* adg = lds + LDS_ADG;
  unsigned* info = reinterpret_cast<unsigned*>(lds + LDS_INFO);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  unsigned xcc, member;
  if (!join_group(ctl, sticky, info, xcc, member)) return;
  const int T = gm.T, Bp = gm.Bp, Hp = gm.Hp, D = gm.D;
  const int NGD = 8 / D, d = (int)xcc / NGD, grp = (int)xcc % NGD;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;

  float wreg[NOG * NC];
  {
    const float* wp = Upb + ((((size_t)d * 32 + member) * 4 + w) * (NOG * NC)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NOG * NC; ++i) wreg[i] = wp[(size_t)i * 64];
  }
  float* gpx = px + (size_t)xcc * 2 * 32 * 32 * 64;
  gu32* gflag = (gu32*)(ctl->flags + xcc * 128);

  const int q = lane >> 4, u = lane & 15;
  const bool cell_lane = w == 0 && u < NU;
  const int j = NU * (int)member + u;
  bool aborted = false;

  for (int rd = 0; rd < gm.rounds; ++rd) {
    const int b0 = (rd * NGD + grp) * gm.ub;
    if (b0 >= Bp) continue;
    const int b = b0 + q;
    const bool rowok = cell_lane && q < gm.ub && b < Bp;
    const int len = rowok ? seq_len[b] : 0;
    const unsigned tagbase = (unsigned)(rd * T);
    float dc = 0.f;
    // operands of the step about to run (prefetched a step ahead): activations, c, c of the previous frame, dOut
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float cc = 0.f, cpv = 0.f, dha = 0.f;
    auto prefetch = [&](int s) {
      if (rowok && s < len) {
        const int tb = d ? (len - 1 - s) : s;
        const size_t r = (size_t)tb * Bp + b;
        a = *reinterpret_cast<const float4*>(gates + r * DN + d * N4 + 4 * j);
        cc = cbuf[r * DH + d * Hp + j];
        cpv = s > 0 ? cbuf[(d ? r + Bp : r - Bp) * DH + d * Hp + j] : 0.f;
        dha = dout[r * DH + d * Hp + j];
      }
    };
    prefetch(T - 1);
    for (int k = 0; k < T; ++k) {
      const int s = T - 1 - k, par = k & 1;
      bool ok = true;
      if (w == 0) {
        // 1. every wave of every producer has published its partial sums of step k-1
        ok = poll_ge(gflag + lane, true, tagbase + (unsigned)k) && poll_ge(gflag + 64 + lane, true, tagbase + (unsigned)k);
        const bool valid = rowok && s < len;
        float dhs = dha;
        if (k > 0 && ok && cell_lane) {
          const float* src = gpx + ((size_t)(((k - 1) & 1) * 32 + member) * 32) * 64 + lane;
          float pv[32];
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
          NASR_LD8(0); NASR_LD8(8); NASR_LD8(16); NASR_LD8(24);
#undef NASR_LD8
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
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int p = 0; p < 32; p += 4) { s0 += pv[p]; s1 += pv[p + 1]; s2 += pv[p + 2]; s3 += pv[p + 3]; }
          dhs += (s0 + s1) + (s2 + s3);
        }
        // 2. gate derivatives of this CU's cells
        float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
        float dcn = 0.f;
        if (valid) {
          const float tc = ptanh(cc);
          const float dct = dc + dhs * a.w * (1.f - tc * tc);
          dg.x = dct * a.y * a.x * (1.f - a.x);
          dg.y = dct * a.x * (1.f - a.y * a.y);
          dg.z = dct * cpv * a.z * (1.f - a.z);
          dg.w = dhs * tc * a.w * (1.f - a.w);
          dcn = dct * a.z;
        }
        dc = dcn;
        if (cell_lane) {   // A image: adg[par][c = 4u+g][utterance q]
          float* ad = adg + par * 256 + 16 * u + q;
          ad[0] = dg.x; ad[4] = dg.y; ad[8] = dg.z; ad[12] = dg.w;
        }
        if (rowok) {       // frame-indexed dG for the weight-gradient GEMMs (zero at masked frames)
          const size_t row = valid ? (size_t)(d ? (len - 1 - s) : s) * Bp + b : (size_t)s * Bp + b;
          *reinterpret_cast<float4*>(dgbuf + row * DN + d * N4 + 4 * j) = dg;
        }
        a = make_float4(0.f, 0.f, 0.f, 0.f); cc = 0.f; cpv = 0.f; dha = 0.f;
        if (s > 0) prefetch(s - 1);
        if (!ok) info[2 + par] = 1;
      }
      __syncthreads();
      if (info[2 + par]) { aborted = true; break; }
      // 3. partial[utt][k'] = sum_c dG[utt][c] * U[k'][c] for this wave's output units, all 4 waves
      float av[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) av[v] = (64 * v + lane < 4 * NC) ? adg[par * 256 + 64 * v + lane] : 0.f;
      f32x4 acc[NOG][2];
#pragma unroll
      for (int og = 0; og < NOG; ++og) { acc[og][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[og][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      static_for<0, NC>([&](auto cc_) {
        constexpr int c = decltype(cc_)::value;
        static_for<0, NOG>([&](auto ogc) {
          constexpr int og = decltype(ogc)::value;
          acc[og][c & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c / 16], wreg[og * NC + c], acc[og][c & 1], 4, c % 16, 0);
        });
      });
      // 4. hand the partial rows to their consumers: unit k' -> consumer k'/NU, cell lane 16*utt + k'%NU
#pragma unroll
      for (int og = 0; og < NOG; ++og) {
        const f32x4 sum = acc[og][0] + acc[og][1];
        const int kl = og * 64 + lane;
        if (kl < KW) {
          const int kk = w * KW + kl;
          float* dst = gpx + ((size_t)((par * 32 + kk / NU) * 32) + member) * 64 + (kk % NU);
          dst[0] = sum[0]; dst[16] = sum[1]; dst[32] = sum[2]; dst[48] = sum[3];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) *(ctl->flags + xcc * 128 + member * 4 + w) = tagbase + (unsigned)k + 1u;
    }
    if (aborted) break;
  }
  if (aborted && tid == 0) raise_error(ctl, sticky, 1u);
}

// ------------------------------------------------------------------ launchers
bool persist_supported(int Hp) {
  const int NU = Hp / 32;
  return Hp % 32 == 0 && (NU == 2 || NU == 4 || NU == 8 || NU == 16);
}

static PersistGeom make_geom(const LstmDims& dm) {
  PersistGeom g;
  g.T = dm.T; g.Bp = dm.Bp; g.Hp = dm.Hp; g.D = dm.D;
  const int NGD = 8 / dm.D;
  g.ub = (dm.Bp + NGD - 1) / NGD;
  if (g.ub > 4) g.ub = 4;
  g.rounds = (dm.Bp + NGD * g.ub - 1) / (NGD * g.ub);
  return g;
}

size_t persist_xch_floats(int Hp) {
  const size_t f = (size_t)8 * 2 * Hp * 4, b = (size_t)8 * 2 * 32 * 32 * 64;
  return f > b ? f : b;
}

hipError_t persist_prepare() {
  hipError_t e = hipSuccess;
#define NASR_PATTR(NUV)                                                                                             \
  if (e == hipSuccess)                                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_persist_fwd_kernel<NUV>),                           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS_BYTES);                         \
  if (e == hipSuccess)                                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_persist_bwd_kernel<NUV>),                           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS_BYTES);
  NASR_PATTR(2) NASR_PATTR(4) NASR_PATTR(8) NASR_PATTR(16)
#undef NASR_PATTR
  return e;
}

void launch_lstm_persist_fwd(const LstmDims& dm, const float* Upf, float* gates, float* cbuf, float* out,
                             const int* seq_len, float* xch, PersistCtl* ctl, unsigned* sticky, float forget_bias,
                             hipStream_t st) {
  const PersistGeom gm = make_geom(dm);
  (void)hipMemsetAsync(ctl, 0, sizeof(PersistCtl), st);
  dim3 grid(256), block(256);
#define NASR_PF(NUV)                                                                                                  \
  hipLaunchKernelGGL((lstm_persist_fwd_kernel<NUV>), grid, block, PERSIST_LDS_BYTES, st, Upf, gates, cbuf, out, seq_len, \
                     xch, ctl, sticky, gm, forget_bias)
  switch (dm.Hp / 32) {
    case 2: NASR_PF(2); break;
    case 4: NASR_PF(4); break;
    case 8: NASR_PF(8); break;
    default: NASR_PF(16); break;
  }
#undef NASR_PF
}

void launch_lstm_persist_bwd(const LstmDims& dm, const float* Upb, const float* gates, float* dgbuf, const float* cbuf,
                             const float* dout, const int* seq_len, float* xch, PersistCtl* ctl, unsigned* sticky,
                             hipStream_t st) {
  const PersistGeom gm = make_geom(dm);
  (void)hipMemsetAsync(ctl, 0, sizeof(PersistCtl), st);
  dim3 grid(256), block(256);
#define NASR_PB(NUV)                                                                                                   \
  hipLaunchKernelGGL((lstm_persist_bwd_kernel<NUV>), grid, block, PERSIST_LDS_BYTES, st, Upb, gates, dgbuf, cbuf, dout, \
                     seq_len, xch, ctl, sticky, gm)
  switch (dm.Hp / 32) {
    case 2: NASR_PB(2); break;
    case 4: NASR_PB(4); break;
    case 8: NASR_PB(8); break;
    default: NASR_PB(16); break;
  }
#undef NASR_PB
}

}  // namespace nasr
