// gemm_bf16.hip — fp32-accurate NT GEMM on the bf16 matrix cores:  C[M,N] = A[M,K] * B[N,K]^T (+bias[n]).
//
// Each fp32 operand element is split ON THE FLY (while it is staged into LDS) into three bf16 parts
// x = x1 + x2 + x3 (8+8+8 significant bits = the whole fp32 mantissa), and a product is formed from the six
// bf16 MFMAs x1y1 + x1y2 + x2y1 + x1y3 + x2y2 + x3y1 accumulated in fp32 (bf16 x bf16 is exact in fp32; the
// dropped terms are < 2^-24 relative).  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32-input MFMA,
// so six of them cost 6/16 of v_mfma_f32_32x32x2_f32 for the same tile: the GEMM becomes load-bound instead of
// MFMA-bound.  Both operands must have K contiguous ("NT"): the callers keep transposed copies where the
// contraction index is the row index (weights: re-packed after every Adam step; activations / dG: one
// transpose kernel per layer), see nasr_api.hip.
//
// 128x128x32 block tile, 4 waves of 64x64 (2x2 MFMA 32x32 tiles), XOR-swizzled LDS images [part][row][32 k] bf16
// (conflict-free for both the ds_write_b64 staging stores and the ds_read_b128 fragment reads), register-staged
// prefetch of the next k-tile while the current one is multiplied.
//
// STATUS (round 1): validated against the f32-MFMA kernel (relative L2 5e-7 .. 2e-6 on the step's shapes,
// tools/gemmbench.hip) and measured at 123-146 TF-equivalent vs 96-105 TF for gemm.hip (1.3-1.4x).  It is the
// default for the input-to-hidden, input-gradient and weight-gradient GEMMs (NASR_GEMM=f32 selects gemm.hip):
// 3x500 step 19.0 -> 17.1 ms.  The projection GEMMs keep gemm.hip (row gather/scatter maps).  The on-the-fly
// split (~180 VALU ops + 48 KB of ds_write_b64 per k-tile per block) is what keeps it from the 2.7x the MFMA
// count alone would give: ablations (NASR_NT_ABL, xproj 8000x4096x1024, 0.49 ms) put the six MFMAs at 40 %, the
// LDS stores at 40 % and the split arithmetic at 13 % of the time, nearly additive.  Tried and measured slower:
// issuing the split inside the MFMA stream (spills, 0.50 ms) and a producer/consumer warp-specialised 8-wave
// form with double-buffered LDS (160 KB => one block per CU, 0.64 ms); an XCD-aware tile order (no change: the
// operand panels are L2/Infinity-Cache resident either way).
#include "kernels.h"

#ifndef NASR_NT_ABL
#define NASR_NT_ABL 0   // tools/gemmbench ablations: 1 cheap split (no subtraction), 2 one MFMA instead of six, 4 no LDS stores
#endif

namespace nasr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TBM = 128, TBN = 128, TBK = 32, TROW = 32;   // 64-byte LDS rows, XOR-swizzled (see lds_col)

struct GemmNTParams {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  int a_kshift;          // A is read at k + a_kshift (zero outside [0,K)): the h_{t-1} shift of the dU GEMMs
  const float* bias;
  int split_k, kchunk;
  float* slabs;
};

// LDS image [part][row][32 k] bf16 with 64-byte rows.  The four 16-byte k-groups of a row are XOR-swizzled with
// (row >> 2) & 3: a ds_read_b128 lane group (16 rows, same k-group) then lands on 16 different 16-byte slots, and a
// ds_write_b64 lane group (two consecutive rows) covers all 32 banks exactly once — both conflict-free.  (The
// unswizzled 80-byte-row image measured 30 B/clk/CU of LDS store throughput, a 2-way write conflict.)
__device__ __forceinline__ int lds_col(int row, int k) {   // k multiple of 4
  return (((k >> 3) ^ ((row >> 2) & 3)) << 3) | (k & 7);
}

__device__ __forceinline__ void split3(const float4 v, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h1 = (__bf16)x[i];
    if (NASR_NT_ABL & 1) { p1[i] = h1; p2[i] = h1; p3[i] = h1; continue; }
    const float r1 = x[i] - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    p1[i] = h1; p2[i] = h2; p3[i] = (__bf16)r2;
  }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x6_kernel(GemmNTParams p) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][TBM][TROW];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][TBN][TROW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.y * TBM, n0 = blockIdx.x * TBN;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + TBK - 1) / TBK;
  const float* __restrict__ Ag = p.A;
  const float* __restrict__ Bg = p.B;

  // 4 load slots per operand: idx = tid + 256 j -> row = idx >> 3 (0..127), kq = idx & 7 (4 consecutive k)
  bool aok[4], bok[4];
  size_t ao[4], bo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j, row = idx >> 3, kq = idx & 7;
    aok[j] = m0 + row < p.M;
    bok[j] = n0 + row < p.N;
    ao[j] = aok[j] ? (size_t)(m0 + row) * p.lda + 4 * kq : 0;
    bo[j] = bok[j] ? (size_t)(n0 + row) * p.ldb + 4 * kq : 0;
  }
  const int kq_ = tid & 7;
  float4 ra[4], rb[4];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define NASR_NT_GLOAD(K0)                                                                     \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
    const int kb_ = (K0) + 4 * kq_;                                                           \
    const int ka_ = kb_ + p.a_kshift;                                                         \
    const bool oa_ = aok[j] && kb_ < kend && ka_ >= 0 && ka_ < p.K;                           \
    const bool ob_ = bok[j] && kb_ < kend;                                                    \
    const float4 va_ = *reinterpret_cast<const float4*>(Ag + (oa_ ? ao[j] + (K0) + p.a_kshift : 0)); \
    const float4 vb_ = *reinterpret_cast<const float4*>(Bg + (ob_ ? bo[j] + (K0) : 0));      \
    ra[j] = oa_ ? va_ : z4;                                                                   \
    rb[j] = ob_ ? vb_ : z4;                                                                   \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;

  if (nk > 0) NASR_NT_GLOAD(kbeg);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j, row = idx >> 3, kq = idx & 7;
      bf16x4 p1, p2, p3;
      split3(ra[j], p1, p2, p3);
      if ((NASR_NT_ABL & 4) && kt > 0) continue;
      const int lc = lds_col(row, 4 * kq);
      *reinterpret_cast<bf16x4*>(&As[0][row][lc]) = p1;
      *reinterpret_cast<bf16x4*>(&As[1][row][lc]) = p2;
      *reinterpret_cast<bf16x4*>(&As[2][row][lc]) = p3;
      split3(rb[j], p1, p2, p3);
      *reinterpret_cast<bf16x4*>(&Bs[0][row][lc]) = p1;
      *reinterpret_cast<bf16x4*>(&Bs[1][row][lc]) = p2;
      *reinterpret_cast<bf16x4*>(&Bs[2][row][lc]) = p3;
    }
    __syncthreads();
    if (kt + 1 < nk) NASR_NT_GLOAD(kbeg + (kt + 1) * TBK);   // in flight while this tile is multiplied
#pragma unroll
    for (int s = 0; s < TBK / 16; ++s) {
      // fragment: lane (r = lane&31, h = lane>>5) holds k = 16 s + 8 h .. +7 of row r
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          a[i][q] = *reinterpret_cast<const bf16x8*>(&As[q][wm + 32 * i + li][lds_col(wm + 32 * i + li, 16 * s + 8 * lh)]);
          b[i][q] = *reinterpret_cast<const bf16x8*>(&Bs[q][wn + 32 * i + li][lds_col(wn + 32 * i + li, 16 * s + 8 * lh)]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];       // smallest terms first
          if (!(NASR_NT_ABL & 2)) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }
#undef NASR_NT_GLOAD

  // epilogue: C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn + 32 * ni + li;
      if (col >= p.N) continue;
      const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        if (p.split_k > 1) p.slabs[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[mi][ni][r];
        else p.C[(size_t)row * p.ldc + col] = acc[mi][ni][r] + bv;
      }
    }
}

__global__ __launch_bounds__(256) void gemm_nt_reduce_kernel(GemmNTParams p) {
  const int64_t n4 = (int64_t)p.M * p.N / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    const int row = (int)(e / p.N), col = (int)(e % p.N);
    float4 s = *reinterpret_cast<const float4*>(p.slabs + e);
    for (int k = 1; k < p.split_k; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(p.slabs + (size_t)k * p.M * p.N + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (p.bias) { s.x += p.bias[col]; s.y += p.bias[col + 1]; s.z += p.bias[col + 2]; s.w += p.bias[col + 3]; }
    *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = s;
  }
}

void launch_gemm_nt(const GemmNTDesc& g, hipStream_t st) {
  GemmNTParams p;
  p.A = g.A; p.B = g.B; p.C = g.C; p.M = g.M; p.N = g.N; p.K = g.K;
  p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.a_kshift = g.a_kshift; p.bias = g.bias;
  p.split_k = g.split_k < 1 ? 1 : g.split_k;
  p.slabs = g.slabs;
  const int kt = (g.K + TBK - 1) / TBK;
  const int per = (kt + p.split_k - 1) / p.split_k;
  p.kchunk = per * TBK;
  p.split_k = (kt + per - 1) / per;
  dim3 grid((g.N + TBN - 1) / TBN, (g.M + TBM - 1) / TBM, p.split_k);
  hipLaunchKernelGGL(gemm_nt_bf16x6_kernel, grid, dim3(256), 0, st, p);
  if (p.split_k > 1) {
    const int64_t n4 = (int64_t)g.M * g.N / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_nt_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
}

// out[c][r] = in[r][c]  (in: R x C with leading dimension ld_in; out: C x R with leading dimension ld_out)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C, int ld_in, int ld_out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) out[(size_t)c * ld_out + r] = tile[tx][ty + 8 * i];
  }
}

void launch_transpose(const float* in, float* out, int R, int C, int ld_in, int ld_out, hipStream_t st) {
  dim3 grid((C + 31) / 32, (R + 31) / 32), block(256);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, in, out, R, C, ld_in, ld_out);
}

}  // namespace nasr
