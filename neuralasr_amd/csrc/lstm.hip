// lstm.hip — the per-timestep LSTM recurrence and its BPTT as gfx950 kernels.
//
// Semantics (call sites networks/bilstm_ctc_net.py:17-28,
// networks/lstm_ctc_net.py:17-23): g = [x,h]·kernel + bias, gates i,j,f,o,
// c' = c·σ(f+forget_bias) + σ(i)·tanh(j), h' = tanh(c')·σ(o); zero output and carried state past
// seq_len; the bw direction consumes frame seq_len-1-s at step s.  The x·kernel_x + bias half is
// hoisted out of the loop into one MFMA GEMM (gemm.hip); what remains per step is the
// [Bp,Hp]x[Hp,4Hp] recurrent product fused with the cell update.
//
// One launch = one timestep of BOTH directions (blockIdx.y).  Measured on MI355X (tools/stepbench.hip):
// a dependent launch costs 1.55 us whatever its shape, and a CU pulls only ~35-70 GB/s from the Infinity
// Cache / its XCD's L2 (PMC: every launch re-fetches the matrices through the fabric, the L2s do not keep
// them across a kernel boundary), so a step is priced by BYTES PER CU, not by chip bandwidth.  Both kernels
// are therefore cut so that every CU streams ~32 KB of recurrent weights plus the smallest possible share
// of the step's state:
//   forward : block = 4 hidden units x 4 gates (one 16-column MFMA tile), Hp/4 * D blocks (256 for
//             2x512), its 4 waves split K = Hp and reduce through LDS; state = h (32 KB).
//   backward: block = (64-unit output tile, 32-unit K slice); it rebuilds its slice of dG from the
//             previous launch's partial sums, multiplies by its 32 KB tile of U^T and hands 16 x 64
//             partial sums to the next launch (split-K across launches: deterministic, no atomics).
// Operands are stored in HBM in MFMA-fragment order ("swizzled"), four k-steps per 16-byte load, so
// every wave load is one contiguous 1 KiB.
//
// MFMA 16x16x4 f32 fragment maps: A[row = l&15][k = l>>4], B[k = l>>4][col = l&15],
// C/D col = l&15, row = 4*(l>>4) + reg.
#include "kernels.h"

#ifndef NASR_NT
#define NASR_NT 0    // 1: non-temporal loads for the recurrent weights (tools/stepbench.hip experiment)
#endif
#ifndef NASR_NTST
#define NASR_NTST 0  // 1: non-temporal stores for the state handed to the next launch (experiment)
#endif
#ifndef NASR_ABL
#define NASR_ABL 0   // tools/stepbench.hip ablation mask: 1 no U loads, 2 no h loads, 4 no MFMA, 8 no cell I/O
#endif

namespace nasr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ldw(const float4* p) {
#if NASR_NT
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_state(float* p, float v) {
#if NASR_NTST
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
// bare v_exp_f32, as in lstm_persist.hip (saturation makes the denormal scaling of __expf pointless here)
__device__ __forceinline__ float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + exp_(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + exp_(2.f * x)); }

// Element (row b16, unit k) of M-tile mt inside the swizzled h-state image (contraction length Kd).
// Lane l of 16-byte group q holds, in component i, the value the MFMA k-step 4q+i wants from lane l:
// unit k = 16q + 4*(l>>4) + i, row l&15.  The 4 units x 16 rows one block produces are therefore one
// contiguous 256-byte run (coalesced state write).
__device__ __forceinline__ int sw_index(int Kd, int mt, int b16, int k) {
  return ((((mt * (Kd >> 4) + (k >> 4)) * 64) + ((k >> 2) & 3) * 16 + b16) << 2) + (k & 3);
}

// ------------------------------------------------------------------ feature transpose + pad
// feats [B][T][F] batch-major (dataset.py:75-82) -> X0 [(t*Bp+b)][Fp], zero padded
__global__ __launch_bounds__(256) void pack_feats_kernel(const float* __restrict__ f, float* __restrict__ x, int B,
                                                         int Bp, int T, int F, int Fp) {
  const int r = blockIdx.x;  // t*Bp + b
  const int t = r / Bp, b = r % Bp;
  float* dst = x + (size_t)r * Fp;
  if (b >= B) {
    for (int i = threadIdx.x; i < Fp; i += blockDim.x) dst[i] = 0.f;
    return;
  }
  const float* src = f + ((size_t)b * T + t) * F;
  for (int i = threadIdx.x; i < Fp; i += blockDim.x) dst[i] = i < F ? src[i] : 0.f;
}

void launch_pack_feats(const float* feats_bm, float* X0, int B, int Bp, int T, int F, int Fp, hipStream_t st) {
  hipLaunchKernelGGL(pack_feats_kernel, dim3(T * Bp), dim3(256), 0, st, feats_bm, X0, B, Bp, T, F, Fp);
}

// include_context on the device (utils.py:8-21): the caller ships only the centre frame [B][T][numcep]; every
// time-major row gets its (2*ctx+1)-frame window, out-of-range frames of an utterance filled with that
// utterance's pad value (0 before the utterance-level normalisation of utils.py:29, (0-mean)/std after it),
// frames past seq_len zero (dataset.py:75-77).  21x fewer bytes over PCIe for numcontext = 10.
__global__ __launch_bounds__(256) void expand_context_kernel(const float* __restrict__ centre,
                                                             const float* __restrict__ pad, const int* __restrict__ seq_len,
                                                             float* __restrict__ x, int B, int Bp, int T, int ctx,
                                                             int ncep, int Fp) {
  const int r = blockIdx.x;  // t*Bp + b
  const int t = r / Bp, b = r % Bp;
  float* dst = x + (size_t)r * Fp;
  const int len = b < B ? seq_len[b] : 0;
  const int F = (2 * ctx + 1) * ncep;
  for (int i = threadIdx.x; i < Fp; i += blockDim.x) {
    float v = 0.f;
    if (i < F && t < len) {
      const int wdw = i / ncep, c = i - wdw * ncep;
      const int ts = t + wdw - ctx;
      v = (ts >= 0 && ts < len) ? centre[((size_t)b * T + ts) * ncep + c] : pad[b];
    }
    dst[i] = v;
  }
}

void launch_expand_context(const float* centre, const float* pad, const int* seq_len, float* X0, int B, int Bp, int T,
                           int ctx, int ncep, int Fp, hipStream_t st) {
  hipLaunchKernelGGL(expand_context_kernel, dim3(T * Bp), dim3(256), 0, st, centre, pad, seq_len, X0, B, Bp, T, ctx,
                     ncep, Fp);
}

// ------------------------------------------------------------------ recurrent weight repack
// U [Hp][N4] (row k = h unit, col n = 4*j+g) ->
//   Uf [N4/16 tiles][Hp/16][64][4] : Uf[tile][q][l][i] = U[16q + 4*(l>>4) + i][16*tile + (l&15)]
//        forward B operand; k-step 4q+i contracts units {16q + 4*sub + i}, matching sw_index().
//   Ub [Hp/64 jt][Hp/32 ks][4 w][4 nt][2 q2][64][4] :
//        Ub[..][l][i] = U[64jt + 16nt + (l&15)][4*(32ks + 4*(4q2+i) + (l>>4)) + w]
//        backward B operand (= U^T): block (jt,ks), wave w = gate w, k-step 4q2+i contracts the
//        slice-local units {4*(4q2+i) + sub}.
__global__ __launch_bounds__(256) void repack_u_kernel(const float* __restrict__ U, float* __restrict__ Uf,
                                                       float* __restrict__ Ub, int Hp) {
  const int N4 = 4 * Hp;
  const int64_t total = (int64_t)Hp * N4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = e & 3, l = (e >> 2) & 63;
    {  // forward image
      const int64_t tq = e >> 8;
      const int q = (int)(tq % (Hp >> 4)), tile = (int)(tq / (Hp >> 4));
      Uf[e] = U[(size_t)(16 * q + 4 * (l >> 4) + i) * N4 + 16 * tile + (l & 15)];
    }
    {  // backward image
      int64_t x = e >> 8;
      const int q2 = (int)(x & 1); x >>= 1;
      const int nt = (int)(x & 3); x >>= 2;
      const int w = (int)(x & 3); x >>= 2;
      const int ks = (int)(x % (Hp >> 5)), jt = (int)(x / (Hp >> 5));
      const int ju = 4 * (4 * q2 + i) + (l >> 4);
      Ub[e] = U[(size_t)(64 * jt + 16 * nt + (l & 15)) * N4 + 4 * (32 * ks + ju) + w];
    }
  }
}

void launch_repack_u(const float* U, float* Uf, float* Ub, int Hp, hipStream_t st) {
  hipLaunchKernelGGL(repack_u_kernel, dim3(1024), dim3(256), 0, st, U, Uf, Ub, Hp);
}

// ------------------------------------------------------------------ forward step
// NQ = Hp/64 (16-byte operand groups per wave) as a template constant keeps the load/MFMA stream
// straight-line, so hipcc emits counted vmcnt waits and the MFMA chain starts when the first pair lands;
// NQ = 0 is the generic (runtime) form.
template <int MT, int NQ>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(
    const float* __restrict__ Uf,    // [D][Hp/4][Hp/16][64][4]
    const float* __restrict__ hin,   // [D][MT][Hp/16][64][4]
    float* __restrict__ hout, float* __restrict__ gates, float* __restrict__ cbuf, float* __restrict__ out,
    const int* __restrict__ seq_len, int s, int T, int Bp, int Hp, int D, float fb) {
  __shared__ __attribute__((aligned(16))) float red[4][MT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, d = blockIdx.y;
  if ((NASR_ABL & 64) && s != 123456) return;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  const int nq = NQ > 0 ? NQ : (Hp >> 6);   // float4 groups per wave (K quarter)
  const int q0 = w * nq;
  constexpr int CH = NQ == 0 ? 1 : (NQ < 8 ? NQ : 8);

  // seq_len first: vmcnt retires in order, so the wait on it must not sit behind the operand stream
  const int cmt = tid >> 6, cl = tid & 63;
  const int b16 = cl & 15, u = cl >> 4;
  const int b = cmt * 16 + b16;
  const int len_ld = (NASR_ABL & 16) ? T : seq_len[b < Bp ? b : 0];

  // ---- recurrent product: acc[mt] (16 x 16) over this wave's K quarter.  Loads are issued in
  // consumption order (h, U, h, U, ...) ahead of everything else.
  const float4* ub = reinterpret_cast<const float4*>(Uf) + ((size_t)(d * (Hp >> 2) + tile) * (Hp >> 4) + q0) * 64 + lane;
  const float4* ha = reinterpret_cast<const float4*>(hin) + ((size_t)d * MT * (Hp >> 4) + q0) * 64 + lane;
  float4 bu[CH];
  float4 av[MT][CH];
#pragma unroll
  for (int x = 0; x < CH; ++x) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
      av[m][x] = (NASR_ABL & 2) ? make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f) : ha[((size_t)m * (Hp >> 4) + x) * 64];
    bu[x] = (NASR_ABL & 1) ? make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f) : ldw(ub + (size_t)x * 64);
  }

  // ---- cell threads: issue the loads the cell update needs
  const bool cell = tid < 64 * MT;
  const int j = tile * 4 + u;
  int len = 0;
  bool valid = false;
  int r = 0;
  float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
  float cprev = 0.f;
  if (cell) {
    len = len_ld;
    valid = s < len;
    if (valid) {
      const int tb = d ? (len - 1 - s) : s;
      r = tb * Bp + b;
      if (!(NASR_ABL & 8)) {
        xg = *reinterpret_cast<const float4*>(gates + (size_t)r * DN + d * N4 + 4 * j);
        if (s > 0) cprev = cbuf[(size_t)(d ? r + Bp : r - Bp) * DH + d * Hp + j];
      }
    }
  }

  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int qc = 0; qc < nq; qc += CH) {
    if (qc > 0) {
#pragma unroll
      for (int x = 0; x < CH; ++x) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (!(NASR_ABL & 2)) av[m][x] = ha[((size_t)m * (Hp >> 4) + qc + x) * 64];
        if (!(NASR_ABL & 1)) bu[x] = ldw(ub + (size_t)(qc + x) * 64);
      }
    }
#pragma unroll
    for (int x = 0; x < CH; ++x) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (NASR_ABL & 4) {
          acc[m][0][0] += av[m][x].x * bu[x].x + av[m][x].y * bu[x].y + av[m][x].z * bu[x].z + av[m][x].w * bu[x].w;
          continue;
        }
        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].x, bu[x].x, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].y, bu[x].y, acc[m][1], 0, 0, 0);
        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].z, bu[x].z, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].w, bu[x].w, acc[m][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x4 sacc = acc[m][0] + acc[m][1];
    *reinterpret_cast<f32x4*>(&red[w][m][lane][0]) = sacc;
  }
  __syncthreads();

  // ---- fused cell update: thread = (batch row b, hidden unit j)
  if (cell) {
    float* hdst = hout + (size_t)d * MT * Hp * 16 + sw_index(Hp, cmt, b16, j);
    if (valid) {
      const int ls = 16 * (b16 >> 2) + 4 * u, rg = b16 & 3;
      float g[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
        g[gi] = red[0][cmt][ls + gi][rg] + red[1][cmt][ls + gi][rg] + red[2][cmt][ls + gi][rg] + red[3][cmt][ls + gi][rg];
      const float si = sigmoidf_(xg.x + g[0]);
      const float tj = tanhf_(xg.y + g[1]);
      const float sf = sigmoidf_(xg.z + g[2] + fb);
      const float so = sigmoidf_(xg.w + g[3]);
      const float c = cprev * sf + si * tj;
      const float h = tanhf_(c) * so;
      if (!(NASR_ABL & 8)) {
        *reinterpret_cast<float4*>(gates + (size_t)r * DN + d * N4 + 4 * j) = make_float4(si, tj, sf, so);
        cbuf[(size_t)r * DH + d * Hp + j] = c;
        out[(size_t)r * DH + d * Hp + j] = h;
      }
      if (!(NASR_ABL & 32) || h == 123.456f) st_state(hdst, h);
    } else {
      // frame s of row b is past seq_len for both directions: zero output (A.2); state is dead
      if (s < T) out[((size_t)s * Bp + b) * DH + d * Hp + j] = 0.f;
      *hdst = 0.f;
    }
  }
}

void launch_lstm_fwd_step(const LstmDims& dm, int s, const float* Uf, const float* hin, float* hout, float* gates,
                          float* cbuf, float* out, const int* seq_len, float forget_bias, hipStream_t st) {
  dim3 grid(dm.Hp / 4, dm.D), block(256);
  const int MT = dm.Bp / 16, nq = dm.Hp / 64;
#define NASR_FWD(MTV, NQV)                                                                                     \
  hipLaunchKernelGGL((lstm_fwd_step_kernel<MTV, NQV>), grid, block, 0, st, Uf, hin, hout, gates, cbuf, out, \
                     seq_len, s, dm.T, dm.Bp, dm.Hp, dm.D, forget_bias)
#define NASR_FWD_NQ(MTV)                                  \
  switch (nq) {                                           \
    case 1: NASR_FWD(MTV, 1); break;                      \
    case 2: NASR_FWD(MTV, 2); break;                      \
    case 4: NASR_FWD(MTV, 4); break;                      \
    case 8: NASR_FWD(MTV, 8); break;                      \
    case 16: NASR_FWD(MTV, 16); break;                    \
    default: NASR_FWD(MTV, 0); break;                     \
  }
  switch (MT) {
    case 1: NASR_FWD_NQ(1); break;
    case 2: NASR_FWD_NQ(2); break;
    case 3: NASR_FWD_NQ(3); break;
    default: NASR_FWD_NQ(4); break;
  }
#undef NASR_FWD_NQ
#undef NASR_FWD
}

// ------------------------------------------------------------------ BPTT step
// grid.x = (Hp/64 output tiles jt) x (Hp/32 K slices ks), blockIdx.x = jt*KSPLIT + ks so the blocks that
// rebuild the same dG slice share an XCD (ids equal mod 8).  Per block and step:
//   P: for its 32 units x Bp rows: dh = sum_k partial_in[k] + dOut (gradient from above), gate
//      derivatives from the saved activations -> dG slice [Bp x 128] into LDS (k order g*32+ju);
//      the jt == 0 block also stores dG frame-indexed (for the weight-gradient GEMMs) and dc.
//   G: partial_out[ks][b][64jt..] = dG_slice x U^T tile on v_mfma_f32_16x16x4 (wave w = gate w),
//      4-wave LDS reduction, plain stores.  The next launch sums the KSPLIT partials in fixed order.
// Masked frames get dG = 0 so the GEMMs need no mask.
// Wide layers (Hp > 512, e.g. DeepSpeech's 2048): every output tile's block would re-sum the same Hp/32 partials
// (Hp^3 traffic: 434 us per step at Hp = 2048).  There the step is two launches: lstm_bwd_cell_kernel sums the partials
// and does the cell arithmetic ONCE per cell (writing dG frame-indexed), and this kernel runs with PRE = true: its P
// stage only copies its dG slice from that buffer, and each block walks KSL = 4 consecutive K slices with the
// accumulators kept in registers, so 4x fewer partial sums are handed on.
template <int MT, int KSP, bool PRE = false, int KSL = 1>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(
    const float* __restrict__ Ub,     // [D][Hp/64][Hp/32][4][4][2][64][4]
    const float* __restrict__ pin,    // [D][KSPLIT][Bp][Hp] partial sums of dh_rec from the previous launch
    float* __restrict__ pout,
    const float* __restrict__ gates,  // [R][D*N4] activations si,tj,sf,so
    float* __restrict__ dgbuf,        // [R][D*N4] dG, frame indexed
    const float* __restrict__ cbuf, const float* __restrict__ dout, const float* __restrict__ dcin,
    float* __restrict__ dcout, const int* __restrict__ seq_len, int s, int T, int Bp, int Hp, int D) {
  // wide form (PRE): the reduction buffer reuses the dG slice's LDS (32 KB instead of 49 KB at MT = 2), so that the four
  // blocks a CU gets at Hp = 2048 are resident together instead of three and a straggler
  constexpr int AS_FLOATS = MT * 128 * 17, RED_FLOATS = 4 * MT * 4 * 64 * 4;
  __shared__ __attribute__((aligned(16))) float lds_[PRE ? (AS_FLOATS > RED_FLOATS ? AS_FLOATS : RED_FLOATS) : AS_FLOATS + RED_FLOATS];
  float (*As)[128][17] = reinterpret_cast<float (*)[128][17]>(lds_);
  float (*red)[MT][4][64][4] = reinterpret_cast<float (*)[MT][4][64][4]>(PRE ? lds_ : lds_ + AS_FLOATS);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int KSPLIT = KSP > 0 ? KSP : (Hp >> 5);     // K slices of 32 units in the operand image
  const int NKG = KSPLIT / KSL;                      // K-slice groups = partial sums per output
  const int jt = blockIdx.x / NKG, ksg = blockIdx.x % NKG, d = blockIdx.y;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kl = 0; kl < KSL; ++kl) {
  const int ks = ksg * KSL + kl;
  if (KSL > 1 && kl > 0) __syncthreads();            // the previous slice's fragment reads of As are done
  // ---- this wave's 8 B-operand fragments (32 KB per block), in flight during the P stage
  const float4* ub = reinterpret_cast<const float4*>(Ub) +
                     ((((size_t)(d * (Hp >> 6) + jt) * KSPLIT + ks) * 4 + w) * 8) * 64 + lane;
  float4 bu[4][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) bu[nt][q2] = ldw(ub + (size_t)(nt * 2 + q2) * 64);

  // ---- P stage: 2*MT cells per thread, processed in pairs: every load of a pair is issued before any
  // of its arithmetic so the two latency chains overlap.
  const float* pbase = pin + (size_t)d * KSPLIT * Bp * Hp;
  constexpr int KV = KSP > 0 ? KSP : 1;
  if constexpr (PRE) {
#pragma unroll
    for (int cp = 0; cp < 2 * MT; ++cp) {
      const int c = tid + 256 * cp;
      const int ju = c & 31, b = c >> 5;
      const int mt = b >> 4, b16 = b & 15;
      const int len = seq_len[b];
      const int tb = s < len ? (d ? (len - 1 - s) : s) : s;
      const float4 dg = *reinterpret_cast<const float4*>(dgbuf + ((size_t)tb * Bp + b) * DN + d * N4 + 4 * (32 * ks + ju));
      As[mt][0 * 32 + ju][b16] = dg.x;
      As[mt][1 * 32 + ju][b16] = dg.y;
      As[mt][2 * 32 + ju][b16] = dg.z;
      As[mt][3 * 32 + ju][b16] = dg.w;
    }
  } else {
#pragma unroll
  for (int cp2 = 0; cp2 < MT; ++cp2) {
    int lenv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) lenv[e] = seq_len[(tid + 256 * (2 * cp2 + e)) >> 5];
    float4 a[2];
    float cc[2], cpv[2], dh[2], dci[2], pv[2][KV];
    int rr[2];
    bool val[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid + 256 * (2 * cp2 + e);
      const int ju = c & 31, b = c >> 5;
      const int j = 32 * ks + ju;
      val[e] = s < lenv[e];
      const int tb = val[e] ? (d ? (lenv[e] - 1 - s) : s) : 0;
      const int r = tb * Bp + b;
      rr[e] = r;
      // unconditional loads (a masked cell reads frame 0 of its row and is zeroed below): no divergent
      // branch, so both cells' loads are in flight together
      const int rp = s > 0 ? (d ? r + Bp : r - Bp) : r;
      a[e] = *reinterpret_cast<const float4*>(gates + (size_t)r * DN + d * N4 + 4 * j);  // si,tj,sf,so
      cc[e] = cbuf[(size_t)r * DH + d * Hp + j];
      cpv[e] = cbuf[(size_t)(val[e] ? rp : r) * DH + d * Hp + j];
      if (s == 0) cpv[e] = 0.f;
      dh[e] = dout[(size_t)r * DH + d * Hp + j];
      dci[e] = dcin[((size_t)d * Bp + b) * Hp + j];
      const float* pp = pbase + (size_t)b * Hp + j;
      if (KSP > 0) {
#pragma unroll
        for (int k = 0; k < KV; ++k) pv[e][k] = pp[(size_t)k * Bp * Hp];
      } else {
        float acc0 = 0.f;
        for (int k = 0; k < KSPLIT; ++k) acc0 += pp[(size_t)k * Bp * Hp];
        pv[e][0] = acc0;
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid + 256 * (2 * cp2 + e);
      const int ju = c & 31, b = c >> 5;
      const int mt = b >> 4, b16 = b & 15;
      const int j = 32 * ks + ju;
      float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
      float dcn = 0.f;
      if (val[e]) {
        float dhs = dh[e];
#pragma unroll
        for (int k = 0; k < KV; ++k) dhs += pv[e][k];
        const float tc = tanhf_(cc[e]);
        const float dct = dci[e] + dhs * a[e].w * (1.f - tc * tc);
        dg.x = dct * a[e].y * a[e].x * (1.f - a[e].x);
        dg.y = dct * a[e].x * (1.f - a[e].y * a[e].y);
        dg.z = dct * cpv[e] * a[e].z * (1.f - a[e].z);
        dg.w = dhs * tc * a[e].w * (1.f - a[e].w);
        dcn = dct * a[e].z;
      }
      if (jt == 0) {   // the frame of a masked step is frame s itself (past seq_len in both directions): dG = 0 there
        const size_t row = val[e] ? (size_t)rr[e] : (size_t)s * Bp + b;
        *reinterpret_cast<float4*>(dgbuf + row * DN + d * N4 + 4 * j) = dg;
        dcout[((size_t)d * Bp + b) * Hp + j] = dcn;
      }
      As[mt][0 * 32 + ju][b16] = dg.x;
      As[mt][1 * 32 + ju][b16] = dg.y;
      As[mt][2 * 32 + ju][b16] = dg.z;
      As[mt][3 * 32 + ju][b16] = dg.w;
    }
  }
  }   // !PRE
  __syncthreads();

  // ---- G stage: wave w contracts k_local in [32w, 32w+32) (gate w) for the 4 N-tiles
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kl = 32 * w + 4 * (4 * q2 + i) + (lane >> 4);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float a = As[m][kl][lane & 15];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float bv = i == 0 ? bu[nt][q2].x : i == 1 ? bu[nt][q2].y : i == 2 ? bu[nt][q2].z : bu[nt][q2].w;
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[m][nt], 0, 0, 0);
        }
      }
    }
  }
  }   // K-slice loop
  if (PRE) __syncthreads();                            // every wave's fragment reads of As are done: red overwrites it
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(&red[w][m][nt][lane][0]) = acc[m][nt];
  __syncthreads();

  // ---- wave w finalises N-tile w: C/D map col = lane&15 (unit), row = 4*(lane>>4)+reg (batch row)
  float* po = pout + ((size_t)d * NKG + ksg) * Bp * Hp;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][m][w][lane][0]) +
                    *reinterpret_cast<const f32x4*>(&red[1][m][w][lane][0]) +
                    *reinterpret_cast<const f32x4*>(&red[2][m][w][lane][0]) +
                    *reinterpret_cast<const f32x4*>(&red[3][m][w][lane][0]);
    const int jo = 64 * jt + 16 * w + (lane & 15);
    const int b0 = 16 * m + 4 * (lane >> 4);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) st_state(&po[(size_t)(b0 + rg) * Hp + jo], v[rg]);
  }
}

// one thread per cell (b, j) of direction d: dh = sum of the NP partial sums + dOut, gate derivatives, dc; writes the
// frame-indexed dG row (zero at masked frames) and the carried dc
__global__ __launch_bounds__(256) void lstm_bwd_cell_kernel(const float* __restrict__ pin, int NP,
                                                            const float* __restrict__ gates, float* __restrict__ dgbuf,
                                                            const float* __restrict__ cbuf, const float* __restrict__ dout,
                                                            const float* __restrict__ dcin, float* __restrict__ dcout,
                                                            const int* __restrict__ seq_len, int s, int Bp, int Hp, int D) {
  const int c = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
  if (c >= Bp * Hp) return;
  const int j = c % Hp, b = c / Hp;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  const int len = seq_len[b];
  const bool val = s < len;
  const int tb = val ? (d ? (len - 1 - s) : s) : s;
  const size_t r = (size_t)tb * Bp + b;
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
  float dcn = 0.f;
  if (val) {
    const float* pp = pin + (size_t)d * NP * Bp * Hp + (size_t)b * Hp + j;
    float dhs = dout[r * DH + d * Hp + j];
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < NP; k += 2) { s0 += pp[(size_t)k * Bp * Hp]; s1 += pp[(size_t)(k + 1) * Bp * Hp]; }
    if (k < NP) s0 += pp[(size_t)k * Bp * Hp];
    dhs += s0 + s1;
    const float4 a = *reinterpret_cast<const float4*>(gates + r * DN + d * N4 + 4 * j);
    const float cc = cbuf[r * DH + d * Hp + j];
    const float cpv = s > 0 ? cbuf[(d ? r + Bp : r - Bp) * DH + d * Hp + j] : 0.f;
    const float tc = tanhf_(cc);
    const float dct = dcin[((size_t)d * Bp + b) * Hp + j] + dhs * a.w * (1.f - tc * tc);
    dg.x = dct * a.y * a.x * (1.f - a.x);
    dg.y = dct * a.x * (1.f - a.y * a.y);
    dg.z = dct * cpv * a.z * (1.f - a.z);
    dg.w = dhs * tc * a.w * (1.f - a.w);
    dcn = dct * a.z;
  }
  *reinterpret_cast<float4*>(dgbuf + r * DN + d * N4 + 4 * j) = dg;
  dcout[((size_t)d * Bp + b) * Hp + j] = dcn;
}

// partial sums a BPTT step hands on: Hp/32 of them per output, Hp/128 for the wide-layer form
#ifndef NASR_BWD_KSL
#define NASR_BWD_KSL 4   // K slices a block of the wide form walks (tools/stepbench.hip A/B: 2, 4, 8)
#endif
static bool bwd_wide_form(int Hp) { return Hp > 512 && Hp % (32 * NASR_BWD_KSL) == 0; }
int lstm_bwd_partials(int Hp) { return bwd_wide_form(Hp) ? Hp / (32 * NASR_BWD_KSL) : Hp / 32; }

void launch_lstm_bwd_step(const LstmDims& dm, int s, const float* Ub, const float* pin, float* pout,
                          const float* gates, float* dgbuf, const float* cbuf, const float* dout, const float* dcin,
                          float* dcout, const int* seq_len, hipStream_t st) {
  const int MT = dm.Bp / 16, ksp = dm.Hp / 32;
  if (bwd_wide_form(dm.Hp)) {     // wide layer: cell arithmetic once, then the product (see the kernel's header)
    const int np = dm.Hp / (32 * NASR_BWD_KSL);
    hipLaunchKernelGGL(lstm_bwd_cell_kernel, dim3((dm.Bp * dm.Hp + 255) / 256, dm.D), dim3(256), 0, st, pin, np, gates, dgbuf,
                       cbuf, dout, dcin, dcout, seq_len, s, dm.Bp, dm.Hp, dm.D);
    dim3 gridw((dm.Hp / 64) * np, dm.D);
#define NASR_BWDW(MTV)                                                                                              \
  hipLaunchKernelGGL((lstm_bwd_step_kernel<MTV, 0, true, NASR_BWD_KSL>), gridw, dim3(256), 0, st, Ub, pin, pout, gates, dgbuf, cbuf, \
                     dout, dcin, dcout, seq_len, s, dm.T, dm.Bp, dm.Hp, dm.D)
    switch (MT) {
      case 1: NASR_BWDW(1); break;
      case 2: NASR_BWDW(2); break;
      case 3: NASR_BWDW(3); break;
      default: NASR_BWDW(4); break;
    }
#undef NASR_BWDW
    return;
  }
  dim3 grid((dm.Hp / 64) * (dm.Hp / 32), dm.D), block(256);
#define NASR_BWD(MTV, KV)                                                                                        \
  hipLaunchKernelGGL((lstm_bwd_step_kernel<MTV, KV>), grid, block, 0, st, Ub, pin, pout, gates, dgbuf, cbuf, \
                     dout, dcin, dcout, seq_len, s, dm.T, dm.Bp, dm.Hp, dm.D)
#define NASR_BWD_K(MTV)                                   \
  switch (ksp) {                                          \
    case 2: NASR_BWD(MTV, 2); break;                      \
    case 4: NASR_BWD(MTV, 4); break;                      \
    case 8: NASR_BWD(MTV, 8); break;                      \
    case 16: NASR_BWD(MTV, 16); break;                    \
    default: NASR_BWD(MTV, 0); break;                     \
  }
  switch (MT) {
    case 1: NASR_BWD_K(1); break;
    case 2: NASR_BWD_K(2); break;
    case 3: NASR_BWD_K(3); break;
    default: NASR_BWD_K(4); break;
  }
#undef NASR_BWD_K
#undef NASR_BWD
}

}  // namespace nasr
