// gemm.hip — fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 = a k-ordered
// fmaf chain), used for every bulk contraction of the training step:
//   input-to-hidden  gates = X * Wx + bias            (replaces the per-step concat+matmul inside
//                                                      BasicLSTMCell, hoisted over all T)
//   projection       logits = gather(out) * W + b     (networks/bilstm_ctc_net.py:33-45, incl. the D3 row map)
//   weight grads     dWx = X^T dG, dU = shift(H)^T dG, dW = gather(out)^T dlogits   (TN, split-K slabs)
//   input grads      dOut = dG * Wx^T,  dOut = scatter(dlogits * W^T)               (NT)
// 128x128x16 block tile, 4 waves of 64x64 (2x2 MFMA 32x32 tiles), LDS [k][m] / [k][n] images so a
// fragment read is 32 consecutive dwords per half-wave (conflict-free ds_read_b32), register-staged
// double buffering with one barrier per k-tile.  Split-K writes per-split slabs that a second kernel
// sums in fixed order (deterministic; no float atomics).
#include "kernels.h"

namespace nasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NASR_GEMM_BK
#define NASR_GEMM_BK 16
#endif
constexpr int BM = 128, BN = 128, BK = NASR_GEMM_BK, LDT = 132;

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  const int* a_map;
  int a_shift, a_rows;
  const int* c_map;
  const float* bias;
  int split_k, kchunk;
  float* slabs;
};

__device__ __forceinline__ int phys_row(const int* map, int shift, int rows, int logical) {
  int r = map ? map[logical] : logical + shift;
  return (r >= 0 && r < rows) ? r : -1;
}

__device__ __forceinline__ float4 ld4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// Every global load is unconditional (an invalid slot reads the matrix base and is zeroed by a select), so the
// k-loop has no divergent branch and the staged tile lives in registers, not scratch.
template <bool ACOL, bool BCOL>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;      // a K tail (K % BK != 0, K % 4 == 0) is zero-filled
  const float* __restrict__ Ag = p.A;
  const float* __restrict__ Bg = p.B;

  // ---- NS 16-byte load slots per operand per thread (slot j covers idx = tid + 256*j)
  //   "row" slots (A when !ACOL, B when BCOL): r = idx / KQ (0..127), kq = idx % KQ     (4 consecutive k)
  //   "col" slots (A when ACOL, B when !BCOL): kk = idx >> 5 (0..BK-1), q = idx & 31    (4 consecutive m/n)
  constexpr int NS = BK / 8, KQ = BK / 4;
  bool aok[NS], bok[NS];
  size_t ao[NS], bo[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int idx = tid + 256 * j;
    const int r = idx / KQ, kq = idx % KQ, q = idx & 31;
    if (!ACOL) {
      const int pa = (m0 + r < p.M) ? phys_row(p.a_map, p.a_shift, p.a_rows, m0 + r) : -1;
      aok[j] = pa >= 0;
      ao[j] = aok[j] ? (size_t)pa * p.lda + 4 * kq : 0;
    } else {
      aok[j] = (m0 + 4 * q < p.M);
      ao[j] = aok[j] ? (size_t)(m0 + 4 * q) : 0;
    }
    if (BCOL) {
      bok[j] = n0 + r < p.N;
      bo[j] = bok[j] ? (size_t)(n0 + r) * p.ldb + 4 * kq : 0;
    } else {
      bok[j] = (n0 + 4 * q < p.N);
      bo[j] = bok[j] ? (size_t)(n0 + 4 * q) : 0;
    }
  }

  float4 ra[NS], rb[NS];
#define NASR_GLOAD(K0)                                                                              \
  _Pragma("unroll") for (int j = 0; j < NS; ++j) {                                                  \
    const int kk_ = (tid + 256 * j) >> 5;                                                           \
    const bool kr_ = (K0) + 4 * ((tid + 256 * j) % KQ) < kend, kc_ = (K0) + kk_ < kend;             \
    if (!ACOL) {                                                                                    \
      const bool o_ = aok[j] && kr_;                                                                \
      ra[j] = sel4(o_, ld4(Ag + (o_ ? ao[j] + (K0) : 0)));                                          \
    } else {                                                                                        \
      const int pr_ = kc_ ? phys_row(p.a_map, p.a_shift, p.a_rows, (K0) + kk_) : -1;                \
      const bool o_ = aok[j] && pr_ >= 0;                                                           \
      ra[j] = sel4(o_, ld4(Ag + (o_ ? (size_t)pr_ * p.lda + ao[j] : 0)));                           \
    }                                                                                               \
    if (BCOL) {                                                                                     \
      const bool o_ = bok[j] && kr_;                                                                \
      rb[j] = sel4(o_, ld4(Bg + (o_ ? bo[j] + (K0) : 0)));                                          \
    } else {                                                                                        \
      const bool o_ = bok[j] && kc_;                                                                \
      rb[j] = sel4(o_, ld4(Bg + (o_ ? (size_t)((K0) + kk_) * p.ldb + bo[j] : 0)));                  \
    }                                                                                               \
  }
#define NASR_LSTORE(BUF)                                                                            \
  _Pragma("unroll") for (int j = 0; j < NS; ++j) {                                                  \
    const int idx_ = tid + 256 * j;                                                                 \
    const int r_ = idx_ / KQ, kq_ = idx_ % KQ, kk_ = idx_ >> 5, q_ = idx_ & 31;                     \
    if (!ACOL) {                                                                                    \
      As[BUF][4 * kq_ + 0][r_] = ra[j].x; As[BUF][4 * kq_ + 1][r_] = ra[j].y;                       \
      As[BUF][4 * kq_ + 2][r_] = ra[j].z; As[BUF][4 * kq_ + 3][r_] = ra[j].w;                       \
    } else {                                                                                        \
      *reinterpret_cast<float4*>(&As[BUF][kk_][4 * q_]) = ra[j];                                    \
    }                                                                                               \
    if (BCOL) {                                                                                     \
      Bs[BUF][4 * kq_ + 0][r_] = rb[j].x; Bs[BUF][4 * kq_ + 1][r_] = rb[j].y;                       \
      Bs[BUF][4 * kq_ + 2][r_] = rb[j].z; Bs[BUF][4 * kq_ + 3][r_] = rb[j].w;                       \
    } else {                                                                                        \
      *reinterpret_cast<float4*>(&Bs[BUF][kk_][4 * q_]) = rb[j];                                    \
    }                                                                                               \
  }

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int li = lane & 31, lk = lane >> 5;

  if (nk > 0) {
    NASR_GLOAD(kbeg);
    NASR_LSTORE(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) NASR_GLOAD(kbeg + (kt + 1) * BK);
    // fragments of k-step ks+1 are read while the MFMAs of k-step ks issue
    float a0 = As[buf][lk][wm + li], a1 = As[buf][lk][wm + 32 + li];
    float b0 = Bs[buf][lk][wn + li], b1 = Bs[buf][lk][wn + 32 + li];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (ks + 1 < BK / 2) {
        const int k = 2 * (ks + 1) + lk;
        na0 = As[buf][k][wm + li]; na1 = As[buf][k][wm + 32 + li];
        nb0 = Bs[buf][k][wn + li]; nb1 = Bs[buf][k][wn + 32 + li];
      }
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (more) NASR_LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef NASR_GLOAD
#undef NASR_LSTORE

  // ---- epilogue: C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int mi = t >> 1, ni = t & 1;
    const f32x16 acc = t == 0 ? acc00 : t == 1 ? acc01 : t == 2 ? acc10 : acc11;
    const int col = n0 + wn + 32 * ni + li;
    if (col >= p.N) continue;
    const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row >= p.M) continue;
      if (p.split_k > 1) {
        p.slabs[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[r];
      } else {
        const int pr = p.c_map ? p.c_map[row] : row;
        if (pr >= 0) p.C[(size_t)pr * p.ldc + col] = acc[r] + bv;
      }
    }
  }
}

// fixed-order sum of the split-K slabs, then bias / row map
__global__ __launch_bounds__(256) void gemm_reduce_kernel(GemmParams p) {
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
    const int pr = p.c_map ? p.c_map[row] : row;
    if (pr >= 0) *reinterpret_cast<float4*>(p.C + (size_t)pr * p.ldc + col) = s;
  }
}

int gemm_pick_split(int M, int N, int K) {
  // 2 blocks per CU are resident (LDS 34 KB, 106 VGPRs): aim at >= 512 blocks, keep >= 128 k per split
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int s = (512 + tiles - 1) / tiles;
  const int max_s = K / 128;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

void launch_gemm(const GemmDesc& g, hipStream_t st) {
  GemmParams p;
  p.A = g.A; p.B = g.B; p.C = g.C;
  p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
  p.a_map = g.a_map; p.a_shift = g.a_shift; p.a_rows = g.a_rows; p.c_map = g.c_map; p.bias = g.bias;
  p.split_k = g.split_k < 1 ? 1 : g.split_k;
  p.slabs = g.slabs;
  int kt = (g.K + BK - 1) / BK;
  int per = (kt + p.split_k - 1) / p.split_k;
  p.kchunk = per * BK;
  p.split_k = (kt + per - 1) / per;          // drop empty trailing splits
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, p.split_k), block(256);
  if (!g.a_col && !g.b_col) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, st, p);
  else if (g.a_col && !g.b_col) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, st, p);
  else if (!g.a_col && g.b_col) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, st, p);
  if (p.split_k > 1) {
    int64_t n4 = (int64_t)g.M * g.N / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
}

}  // namespace nasr
