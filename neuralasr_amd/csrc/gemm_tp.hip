// gemm_tp.hip — fp32-accurate GEMM on the bf16 matrix cores from PRE-SPLIT, PRE-TILED operands:
//     C[M,N] = A[M,K] * B[N,K]^T (+bias[n])
//
// gemm_bf16.hip splits every fp32 operand element into three bf16 parts while it stages it into LDS, in EVERY block
// that touches it (an 8000 x 4096 x 1024 GEMM re-splits A 32 times and B 63 times); measured there: the six MFMAs
// are 40 % of the time, the LDS staging stores 40 %, the split arithmetic 13 %.  Here the split happens ONCE per
// operand, in a streaming pass (tp_split_kernel) that also writes the parts in the exact order the GEMM wants them
// in LDS ("tiled planes", TP):
//
//     TP[rb = row/32][kb = k/16][part p = 0..2][1 KiB tile]
//     tile: 32 rows x 16 k of bf16, row r's k-half h (8 values = 16 B) at byte ((2r + (h ^ ((r >> 3) & 1))) * 16)
//
// so that (1) one `global_load_lds_dwordx4` wave-instruction moves one whole, contiguous 1 KiB tile HBM/L2 -> LDS with
// no registers, no ds_write and no VALU (LDS-DMA: destination = wave-uniform base + lane * 16), and (2) the
// ds_read_b128 fragment reads of v_mfma_f32_32x32x16_bf16 (lane = row, k-half) are bank-conflict-free (the XOR puts
// rows r and r + 8, which share a bank row, on different 16-byte slots).  x = x1 + x2 + x3 holds the whole fp32
// mantissa and a product is x1y1 + x1y2 + x2y1 + x1y3 + x2y2 + x3y1 accumulated in fp32, as in gemm_bf16.hip.
//
// Kernel: 256 x 256 output tile, BK = 16 (one MFMA k-step = 96 MFMAs per wave pair... 48 per wave), 8 waves as
// 2 (M) x 4 (N), each 128 x 64; two 48 KiB LDS buffers; per k-step every wave issues its 6 of the 48 tile DMAs for
// step t+1, multiplies step t, then `s_waitcnt vmcnt(0)` + one barrier.  One block per CU (96 KiB LDS).
#include "kernels.h"

namespace nasr {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int TPB = 1024;                       // bytes of one tile
constexpr int BUF_BYTES = 48 * TPB;             // A: 8 row blocks x 3 parts, B: 8 x 3
#ifndef NASR_TP_NBUF
#define NASR_TP_NBUF 2      // LDS buffers: 2 = one k-step of DMA in flight; 3 = two (counted vmcnt + raw s_barrier):
                            // measured 0-1 % SLOWER on every shape of tools/gemmbench.hip - DMA latency is not the limit
#endif
constexpr int GEMM_TP_LDS = NASR_TP_NBUF * BUF_BYTES;

__device__ __attribute__((aligned(1024))) unsigned char g_tp_zero[TPB];   // what out-of-range tiles read

__device__ __forceinline__ int tp_slot(int r, int h) { return ((r << 1) | (h ^ ((r >> 3) & 1))) << 4; }

struct GemmTPParams {
  const unsigned char* A;
  const unsigned char* B;
  float* C;
  int M, N;              // logical sizes (multiples of 4)
  int nkbA, nkbB;        // k-blocks stored per row block in A / B
  int kbs;               // k-blocks of the contraction
  int ldc;
  int a_kb_shift;        // A is read at k-block kb + a_kb_shift (zero outside its range)
  const float* bias;
  int split_k, kb_chunk;
  float* slabs;
};

}  // namespace

// ------------------------------------------------------------------ fp32 -> tiled planes
// element (row, k) = transposed ? src[k * ld + row] : src[row * ld + k]; zero outside [0,rows) x [0,K).
// grid (ceil(K/64), ceil(rows/32)), 256 threads: thread (r = t >> 3, c = t & 7) owns 8 consecutive k of one row.
__global__ __launch_bounds__(256) void tp_split_kernel(const float* __restrict__ src, unsigned char* __restrict__ tp, int rows,
                                                       int K, int ld, int transposed, int nkb) {
  __shared__ float tile[64][33];
  const int t = threadIdx.x, r = t >> 3, c = t & 7;
  const int rb = blockIdx.y, k0 = blockIdx.x * 64;
  const int row = rb * 32 + r;
  float x[8];
  if (transposed) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = (t >> 5) + 8 * i, col = t & 31;
      tile[kk][col] = (k0 + kk < K && rb * 32 + col < rows) ? src[(size_t)(k0 + kk) * ld + rb * 32 + col] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = tile[8 * c + e][r];
  } else {
    const int k = k0 + 8 * c;
    if (row < rows && k + 8 <= K) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)row * ld + k);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)row * ld + k + 4);
      x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (row < rows && k + e < K) ? src[(size_t)row * ld + k + e] : 0.f;
    }
  }
  const int kb = (k0 >> 4) + (c >> 1);
  if (kb >= nkb) return;
  bf16x8 p1, p2, p3;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h1 = (__bf16)x[e];
    const float r1 = x[e] - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    p1[e] = h1; p2[e] = h2; p3[e] = (__bf16)r2;
  }
  unsigned char* dst = tp + ((size_t)rb * nkb + kb) * 3 * TPB + tp_slot(r, c & 1);
  *reinterpret_cast<bf16x8*>(dst) = p1;
  *reinterpret_cast<bf16x8*>(dst + TPB) = p2;
  *reinterpret_cast<bf16x8*>(dst + 2 * TPB) = p3;
}

size_t tp_bytes(int rows, int K) { return (size_t)((rows + 31) / 32) * ((K + 15) / 16) * 3 * TPB; }

void launch_tp_split(const float* src, unsigned char* tp, int rows, int K, int ld, bool transposed, hipStream_t st) {
  const int nkb = (K + 15) / 16;
  dim3 grid((K + 63) / 64, (rows + 31) / 32);
  hipLaunchKernelGGL(tp_split_kernel, grid, dim3(256), 0, st, src, tp, rows, K, ld, transposed ? 1 : 0, nkb);
}

// ------------------------------------------------------------------ the GEMM
__global__ __launch_bounds__(512, 2) void gemm_tp_kernel(GemmTPParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 256;
  const int kb0 = blockIdx.z * p.kb_chunk;
  const int kb1 = min(p.kbs, kb0 + p.kb_chunk);
  const int wm = w >> 2, wn = w & 3;

  // this wave's 6 of the 48 tiles of a k-step: tile ti = (operand, row block, part)
  const unsigned char* tbase[6];   // address of the tile at k-block 0, or NULL when the row block is outside the matrix
  int tstride[6], tshift[6], tnkb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int ti = w * 6 + i;
    const bool isB = ti >= 24;
    const int t2 = isB ? ti - 24 : ti;
    const int rbl = t2 / 3, part = t2 - 3 * rbl;
    const int rb = ((isB ? n0 : m0) >> 5) + rbl;
    const int nkb = isB ? p.nkbB : p.nkbA;
    const bool ok = rb * 32 < (isB ? p.N : p.M);
    tbase[i] = ok ? (isB ? p.B : p.A) + ((size_t)rb * nkb * 3 + part) * TPB : nullptr;
    tstride[i] = 3 * TPB;
    tshift[i] = isB ? 0 : p.a_kb_shift;
    tnkb[i] = nkb;
  }
  auto issue = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int kk = kb + tshift[i];
      const unsigned char* g = (tbase[i] && kk >= 0 && kk < tnkb[i]) ? tbase[i] + (size_t)kk * tstride[i] : g_tp_zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + lane * 16), (lds_ptr_t)(lds + buf * BUF_BYTES + (w * 6 + i) * TPB), 16, 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int foff = tp_slot(lane & 31, lane >> 5);
#if NASR_TP_NBUF == 2
  if (kb0 < kb1) issue(kb0, 0);
  for (int kb = kb0; kb < kb1; ++kb) {
    const int buf = (kb - kb0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's tiles of step kb have landed ...
    __syncthreads();                                    // ... and everybody's; buffer buf^1 is no longer being read
    if (kb + 1 < kb1) issue(kb + 1, buf ^ 1);
#else
  // experiment (see NASR_TP_NBUF): three buffers, two k-steps of DMA in flight.  Counted vmcnt (6 DMAs per wave and
  // step) and a raw s_barrier: a __syncthreads() would drain the DMA queue (vmcnt(0)).
  if (kb0 < kb1) issue(kb0, 0);
  if (kb0 + 1 < kb1) issue(kb0 + 1, 1);
  int buf = 0;
  for (int kb = kb0; kb < kb1; ++kb) {
    if (kb + 1 < kb1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but the newest step's tiles have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // everybody's; the buffer read in step kb-1 is free again
    asm volatile("" ::: "memory");
    if (kb + 2 < kb1) issue(kb + 2, buf >= 1 ? buf - 1 : 2);
#endif
    const unsigned char* ab = lds + buf * BUF_BYTES + foff;
    bf16x8 a[4][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(ab + (((wm * 4 + i) * 3 + q) * TPB));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(ab + ((24 + (wn * 2 + j) * 3 + q) * TPB));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];       // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
#if NASR_TP_NBUF != 2
    buf = buf == 2 ? 0 : buf + 1;
#endif
  }

  // epilogue: C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + 32 * ni + li;
      if (col >= p.N) continue;
      const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        if (p.split_k > 1) p.slabs[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[mi][ni][r];
        else p.C[(size_t)row * p.ldc + col] = acc[mi][ni][r] + bv;
      }
    }
}

hipError_t gemm_tp_prepare() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             GEMM_TP_LDS);
}

// split so that the grid has >= ~256 blocks (one per CU) while every slice keeps >= 16 k-steps
int gemm_tp_pick_split(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  const int kbs = (K + 15) / 16;
  int s = 1;
  while (tiles * s < 256 && kbs / (2 * s) >= 16) s *= 2;
  return s;
}

void launch_gemm_tp(const GemmTPDesc& g, hipStream_t st) {
  GemmTPParams p;
  p.A = g.A; p.B = g.B; p.C = g.C; p.M = g.M; p.N = g.N;
  p.nkbA = g.nkbA; p.nkbB = g.nkbB; p.kbs = (g.K + 15) / 16; p.ldc = g.ldc;
  p.a_kb_shift = g.a_kshift / 16;
  p.bias = g.bias;
  int split = g.split_k < 1 ? 1 : g.split_k;
  const int per = (p.kbs + split - 1) / split;
  p.kb_chunk = per;
  p.split_k = (p.kbs + per - 1) / per;
  p.slabs = g.slabs;
  dim3 grid((g.N + 255) / 256, (g.M + 255) / 256, p.split_k);
  hipLaunchKernelGGL(gemm_tp_kernel, grid, dim3(512), GEMM_TP_LDS, st, p);
  if (p.split_k > 1) launch_reduce_slabs(g.slabs, p.split_k, (int64_t)g.M * g.N, g.C, st);   // needs ldc == N, no bias
}

}  // namespace nasr
