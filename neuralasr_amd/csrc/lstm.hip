// lstm.hip — the per-timestep LSTM recurrence and its BPTT as gfx950 kernels.
//
// Semantics (call sites networks/bilstm_ctc_net.py:17-28,
// networks/lstm_ctc_net.py:17-23): g = [x,h]·kernel + bias, gates i,j,f,o,
// c' = c·σ(f+forget_bias) + σ(i)·tanh(j), h' = tanh(c')·σ(o); zero output and carried state past
// seq_len; the bw direction consumes frame seq_len-1-s at step s.  The x·kernel_x + bias half is
// hoisted out of the loop into one MFMA GEMM (gemm.hip); what remains per step is the
// [Bp,Hp]x[Hp,4Hp] recurrent product fused with the cell update.
//
// One launch = one timestep of BOTH directions (blockIdx.y).  A block owns 4 hidden units x 4 gates
// = one 16-column MFMA tile of the gate-interleaved recurrent matrix, so Hp/4 * D blocks (256 for
// 2x512) fill the chip; its 4 waves split K = Hp and reduce through LDS.  Operands are stored in HBM
// in MFMA-fragment order ("swizzled"): lane l of k-step ks reads element 64*ks + l, four k-steps per
// 16-byte load, so every wave load is one contiguous 1 KiB.  The recurrent weights (32 KB per block)
// are re-read every step from the XCD's L2 (block -> XCD mapping is fixed across launches).
//
// MFMA 16x16x4 f32 fragment maps: A[row = l&15][k = l>>4], B[k = l>>4][col = l&15],
// C/D col = l&15, row = 4*(l>>4) + reg.
#include "kernels.h"

namespace nasr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f / (1.f + __expf(2.f * x)); }

// element (b16, k) of M-tile mt inside a swizzled A-operand image with contraction length Kd
__device__ __forceinline__ int sw_index(int Kd, int mt, int b16, int k) {
  return ((((mt * (Kd >> 4) + (k >> 4)) * 64) + (k & 3) * 16 + b16) << 2) + ((k >> 2) & 3);
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

// ------------------------------------------------------------------ recurrent weight repack
// U [Hp][N4] (row k = h unit, col n = 4*j+g) ->
//   Uf [N4/16 tiles][Hp/16][64][4] : Uf[tile][q][l][i] = U[16q+4i+(l>>4)][16*tile+(l&15)]   (fwd B operand)
//   Ub [Hp/16 tiles][N4/16][64][4] : Ub[jt][q][l][i]   = U[16*jt+(l&15)][16q+4i+(l>>4)]     (bwd B operand = U^T)
__global__ __launch_bounds__(256) void repack_u_kernel(const float* __restrict__ U, float* __restrict__ Uf,
                                                       float* __restrict__ Ub, int Hp) {
  const int N4 = 4 * Hp;
  const int64_t total = (int64_t)Hp * N4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    {  // forward image
      const int i = e & 3, l = (e >> 2) & 63;
      const int64_t tq = e >> 8;
      const int q = (int)(tq % (Hp >> 4)), tile = (int)(tq / (Hp >> 4));
      Uf[e] = U[(size_t)(16 * q + 4 * i + (l >> 4)) * N4 + 16 * tile + (l & 15)];
    }
    {  // backward image
      const int i = e & 3, l = (e >> 2) & 63;
      const int64_t tq = e >> 8;
      const int q = (int)(tq % (N4 >> 4)), jt = (int)(tq / (N4 >> 4));
      Ub[e] = U[(size_t)(16 * jt + (l & 15)) * N4 + 16 * q + 4 * i + (l >> 4)];
    }
  }
}

void launch_repack_u(const float* U, float* Uf, float* Ub, int Hp, hipStream_t st) {
  hipLaunchKernelGGL(repack_u_kernel, dim3(1024), dim3(256), 0, st, U, Uf, Ub, Hp);
}

// ------------------------------------------------------------------ forward step
template <int MT>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(
    const float* __restrict__ Uf,    // [D][Hp/4][Hp/16][64][4]
    const float* __restrict__ hin,   // [D][MT][Hp/16][64][4]
    float* __restrict__ hout, float* __restrict__ gates, float* __restrict__ cbuf, float* __restrict__ out,
    const int* __restrict__ seq_len, int s, int T, int Bp, int Hp, int D, float fb) {
  __shared__ __attribute__((aligned(16))) float red[4][MT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, d = blockIdx.y;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  const int nq = Hp >> 6;          // float4 groups per wave (K quarter)
  const int q0 = w * nq;

  // ---- cell threads: issue the loads the cell update needs before the GEMM
  const bool cell = tid < 64 * MT;
  const int cmt = tid >> 6, cl = tid & 63;
  const int b16 = cl & 15, u = cl >> 4;
  const int b = cmt * 16 + b16;
  const int j = tile * 4 + u;
  int len = 0;
  bool valid = false;
  int r = 0;
  float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
  float cprev = 0.f;
  if (cell) {
    len = seq_len[b];
    valid = s < len;
    if (valid) {
      const int tb = d ? (len - 1 - s) : s;
      r = tb * Bp + b;
      xg = *reinterpret_cast<const float4*>(gates + (size_t)r * DN + d * N4 + 4 * j);
      if (s > 0) cprev = cbuf[(size_t)(d ? r + Bp : r - Bp) * DH + d * Hp + j];
    }
  }

  // ---- recurrent product: acc[mt] (16 x 16) over this wave's K quarter
  const float4* ub = reinterpret_cast<const float4*>(Uf) + ((size_t)(d * (Hp >> 2) + tile) * (Hp >> 4) + q0) * 64 + lane;
  const float4* ha = reinterpret_cast<const float4*>(hin) + ((size_t)d * MT * (Hp >> 4) + q0) * 64 + lane;
  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int qc = 0; qc < nq; qc += 8) {
    float4 bu[8];
    float4 av[MT][8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
      if (qc + x < nq) bu[x] = ub[(size_t)(qc + x) * 64];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (qc + x < nq) av[m][x] = ha[((size_t)m * (Hp >> 4) + qc + x) * 64];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      if (qc + x < nq) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].x, bu[x].x, acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].y, bu[x].y, acc[m][1], 0, 0, 0);
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].z, bu[x].z, acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].w, bu[x].w, acc[m][1], 0, 0, 0);
        }
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
      *reinterpret_cast<float4*>(gates + (size_t)r * DN + d * N4 + 4 * j) = make_float4(si, tj, sf, so);
      cbuf[(size_t)r * DH + d * Hp + j] = c;
      out[(size_t)r * DH + d * Hp + j] = h;
      *hdst = h;
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
  const int MT = dm.Bp / 16;
#define NASR_FWD(MTV)                                                                                            \
  hipLaunchKernelGGL((lstm_fwd_step_kernel<MTV>), grid, block, 0, st, Uf, hin, hout, gates, cbuf, out, seq_len, \
                     s, dm.T, dm.Bp, dm.Hp, dm.D, forget_bias)
  switch (MT) {
    case 1: NASR_FWD(1); break;
    case 2: NASR_FWD(2); break;
    case 3: NASR_FWD(3); break;
    default: NASR_FWD(4); break;
  }
#undef NASR_FWD
}

// ------------------------------------------------------------------ BPTT step
// Block = 16 hidden units of one direction.  Phase 1: dh_rec[b][j] = sum_n dG_{next}[b][n] U[j][n]
// (K = N4 split over the 4 waves).  Phase 2 (thread = (b, j)): add the gradient from above,
// gate derivatives from the saved activations, write dG for this frame in place over the
// activations (frame-indexed, consumed by the weight-gradient GEMMs) and in swizzled A-operand
// order for the next step.  Masked frames get dG = 0 so the GEMMs need no mask.
template <int MT>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(
    const float* __restrict__ Ub,    // [D][Hp/16][N4/16][64][4]
    const float* __restrict__ dgin,  // [D][MT][N4/16][64][4]
    float* __restrict__ dgout, float* __restrict__ gates, const float* __restrict__ cbuf,
    const float* __restrict__ dout, float* __restrict__ dcstate, const int* __restrict__ seq_len, int s, int T,
    int Bp, int Hp, int D) {
  __shared__ __attribute__((aligned(16))) float red[4][MT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int jt = blockIdx.x, d = blockIdx.y;
  const int N4 = 4 * Hp, DH = D * Hp, DN = D * N4;
  const int nq = N4 >> 6;
  const int q0 = w * nq;

  const float4* ub = reinterpret_cast<const float4*>(Ub) + ((size_t)(d * (Hp >> 4) + jt) * (N4 >> 4) + q0) * 64 + lane;
  const float4* ga = reinterpret_cast<const float4*>(dgin) + ((size_t)d * MT * (N4 >> 4) + q0) * 64 + lane;
  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int qc = 0; qc < nq; qc += 8) {
    float4 bu[8];
    float4 av[MT][8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
      if (qc + x < nq) bu[x] = ub[(size_t)(qc + x) * 64];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (qc + x < nq) av[m][x] = ga[((size_t)m * (N4 >> 4) + qc + x) * 64];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      if (qc + x < nq) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].x, bu[x].x, acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].y, bu[x].y, acc[m][1], 0, 0, 0);
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].z, bu[x].z, acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][x].w, bu[x].w, acc[m][1], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x4 sacc = acc[m][0] + acc[m][1];
    *reinterpret_cast<f32x4*>(&red[w][m][lane][0]) = sacc;
  }
  __syncthreads();

  const int b16 = tid & 15, ju = tid >> 4;
  const int j = jt * 16 + ju;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int b = m * 16 + b16;
    const int len = seq_len[b];
    float* dgs = dgout + (size_t)d * MT * N4 * 16;
    float* dcs = dcstate + ((size_t)d * Bp + b) * Hp + j;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < len) {
      const int tb = d ? (len - 1 - s) : s;
      const int r = tb * Bp + b;
      const int ls = 16 * (b16 >> 2) + ju, rg = b16 & 3;
      const float dh = red[0][m][ls][rg] + red[1][m][ls][rg] + red[2][m][ls][rg] + red[3][m][ls][rg] +
                       dout[(size_t)r * DH + d * Hp + j];
      const float4 a = *reinterpret_cast<const float4*>(gates + (size_t)r * DN + d * N4 + 4 * j);  // si,tj,sf,so
      const float c = cbuf[(size_t)r * DH + d * Hp + j];
      const float cp = s > 0 ? cbuf[(size_t)(d ? r + Bp : r - Bp) * DH + d * Hp + j] : 0.f;
      const float tc = tanhf_(c);
      const float dct = *dcs + dh * a.w * (1.f - tc * tc);
      dg.x = dct * a.y * a.x * (1.f - a.x);
      dg.y = dct * a.x * (1.f - a.y * a.y);
      dg.z = dct * cp * a.z * (1.f - a.z);
      dg.w = dh * tc * a.w * (1.f - a.w);
      *dcs = dct * a.z;
      *reinterpret_cast<float4*>(gates + (size_t)r * DN + d * N4 + 4 * j) = dg;
    } else {
      *dcs = 0.f;
      if (s < T) *reinterpret_cast<float4*>(gates + ((size_t)s * Bp + b) * DN + d * N4 + 4 * j) = dg;
    }
    dgs[sw_index(N4, m, b16, 4 * j + 0)] = dg.x;
    dgs[sw_index(N4, m, b16, 4 * j + 1)] = dg.y;
    dgs[sw_index(N4, m, b16, 4 * j + 2)] = dg.z;
    dgs[sw_index(N4, m, b16, 4 * j + 3)] = dg.w;
  }
}

void launch_lstm_bwd_step(const LstmDims& dm, int s, const float* Ub, const float* dgin, float* dgout, float* gates,
                          const float* cbuf, const float* dout, float* dcstate, const int* seq_len, hipStream_t st) {
  dim3 grid(dm.Hp / 16, dm.D), block(256);
  const int MT = dm.Bp / 16;
#define NASR_BWD(MTV)                                                                                         \
  hipLaunchKernelGGL((lstm_bwd_step_kernel<MTV>), grid, block, 0, st, Ub, dgin, dgout, gates, cbuf, dout, \
                     dcstate, seq_len, s, dm.T, dm.Bp, dm.Hp, dm.D)
  switch (MT) {
    case 1: NASR_BWD(1); break;
    case 2: NASR_BWD(2); break;
    case 3: NASR_BWD(3); break;
    default: NASR_BWD(4); break;
  }
#undef NASR_BWD
}

}  // namespace nasr
