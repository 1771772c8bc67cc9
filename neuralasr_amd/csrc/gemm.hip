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

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;

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

template <bool ACOL, bool BCOL>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg) / BK;

  // ---- per-thread global load slots: 2 float4 of A and 2 of B per k-tile
  const float* aptr[2];
  int a_i0[2], a_i1[2];   // LDS coordinates
  const float* bptr[2];
  int b_i0[2], b_i1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 256 * j;
    if (!ACOL) {  // rows m, 4 consecutive k
      const int mm = idx >> 2, kq = idx & 3;
      a_i0[j] = mm; a_i1[j] = kq;
      const int m = m0 + mm;
      int pr = (m < p.M) ? phys_row(p.a_map, p.a_shift, p.a_rows, m) : -1;
      aptr[j] = pr >= 0 ? p.A + (size_t)pr * p.lda + 4 * kq : nullptr;
    } else {      // rows k, 4 consecutive m
      const int kk = idx >> 5, mq = idx & 31;
      a_i0[j] = kk; a_i1[j] = mq;
      aptr[j] = (m0 + 4 * mq < p.M) ? p.A + m0 + 4 * mq : nullptr;
    }
    if (BCOL) {   // rows n, 4 consecutive k
      const int nn = idx >> 2, kq = idx & 3;
      b_i0[j] = nn; b_i1[j] = kq;
      const int n = n0 + nn;
      bptr[j] = (n < p.N) ? p.B + (size_t)n * p.ldb + 4 * kq : nullptr;
    } else {      // rows k, 4 consecutive n
      const int kk = idx >> 5, nq = idx & 31;
      b_i0[j] = kk; b_i1[j] = nq;
      bptr[j] = (n0 + 4 * nq < p.N) ? p.B + n0 + 4 * nq : nullptr;
    }
  }

  float4 ra[2], rb[2];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!ACOL) {
        ra[j] = aptr[j] ? *reinterpret_cast<const float4*>(aptr[j] + k0) : z4;
      } else {
        int pr = phys_row(p.a_map, p.a_shift, p.a_rows, k0 + a_i0[j]);
        ra[j] = (aptr[j] && pr >= 0) ? *reinterpret_cast<const float4*>(aptr[j] + (size_t)pr * p.lda) : z4;
      }
      if (BCOL) {
        rb[j] = bptr[j] ? *reinterpret_cast<const float4*>(bptr[j] + k0) : z4;
      } else {
        rb[j] = bptr[j] ? *reinterpret_cast<const float4*>(bptr[j] + (size_t)(k0 + b_i0[j]) * p.ldb) : z4;
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!ACOL) {
        const int mm = a_i0[j], kq = a_i1[j];
        As[buf][4 * kq + 0][mm] = ra[j].x; As[buf][4 * kq + 1][mm] = ra[j].y;
        As[buf][4 * kq + 2][mm] = ra[j].z; As[buf][4 * kq + 3][mm] = ra[j].w;
      } else {
        *reinterpret_cast<float4*>(&As[buf][a_i0[j]][4 * a_i1[j]]) = ra[j];
      }
      if (BCOL) {
        const int nn = b_i0[j], kq = b_i1[j];
        Bs[buf][4 * kq + 0][nn] = rb[j].x; Bs[buf][4 * kq + 1][nn] = rb[j].y;
        Bs[buf][4 * kq + 2][nn] = rb[j].z; Bs[buf][4 * kq + 3][nn] = rb[j].w;
      } else {
        *reinterpret_cast<float4*>(&Bs[buf][b_i0[j]][4 * b_i1[j]]) = rb[j];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int li = lane & 31, lk = lane >> 5;

  if (nk > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int k = 2 * ks + lk;
      const float a0 = As[buf][k][wm + li], a1 = As[buf][k][wm + 32 + li];
      const float b0 = Bs[buf][k][wn + li], b1 = Bs[buf][k][wn + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn + 32 * ni + li;
      if (col >= p.N) continue;
      const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= p.M) continue;
        if (p.split_k > 1) {
          p.slabs[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[mi][ni][r];
        } else {
          const int pr = p.c_map ? p.c_map[row] : row;
          if (pr >= 0) p.C[(size_t)pr * p.ldc + col] = acc[mi][ni][r] + bv;
        }
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
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int s = (384 + tiles - 1) / tiles;         // aim at >= ~1.5 blocks per CU
  const int max_s = K / (BK * 8);            // keep >= 8 k-tiles per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (tiles >= 192) s = 1;
  return s;
}

void launch_gemm(const GemmDesc& g, hipStream_t st) {
  GemmParams p;
  p.A = g.A; p.B = g.B; p.C = g.C;
  p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
  p.a_map = g.a_map; p.a_shift = g.a_shift; p.a_rows = g.a_rows; p.c_map = g.c_map; p.bias = g.bias;
  p.split_k = g.split_k < 1 ? 1 : g.split_k;
  p.slabs = g.slabs;
  int kt = g.K / BK;
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
