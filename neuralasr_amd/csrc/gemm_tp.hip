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
#define NASR_TP_NBUF 2      // LDS buffers: 2 = fragments read after the barrier of their own k-step; 3 = fragments of step k+1
                            // read under the MFMAs of step k (DMA two steps ahead): measured equal or 1-4 % SLOWER on every
                            // shape of tools/gemmbench.hip - neither DMA nor LDS-read latency is what limits this kernel
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
  // two products in one launch (blockIdx.z = slice * nbatch + batch): batch 1 reads A / B this many bytes further on,
  // writes C c_bstride floats further on and uses its own k-block shift
  int nbatch;
  long long a_bstride, b_bstride, c_bstride;
  int a_kb_shift1;
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

// One pass over src [rows][K] (row stride ld) that writes BOTH plane sets: tpN = planes of src (row = src row, k = src
// column) and tpT = planes of its transpose (row = src column, k = src row) - dG feeds the input-gradient GEMM in the
// first form and the weight-gradient GEMMs in the second.  A block stages a 64 x 64 fp32 piece in LDS; grid
// (ceil(K/64), ceil(rows/64)), 256 threads.
// tpN may be NULL (transposed planes only).  colpart (or NULL): [gridDim.y][K] partial column sums of src, 64 rows
// each, summed afterwards by launch_colsum_parts - the bias gradient rides on the pass that reads dG anyway.
__global__ __launch_bounds__(256) void tp_split2_kernel(const float* __restrict__ src, unsigned char* __restrict__ tpN,
                                                        unsigned char* __restrict__ tpT, int rows, int K, int ld,
                                                        float* __restrict__ colpart) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int nkbN = (K + 15) / 16, nrbN = (rows + 31) / 32;
  const int nkbT = (rows + 15) / 16, nrbT = (K + 31) / 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {                   // 16 rows per pass, one float4 per thread
    const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows) {
      const float* s = src + (size_t)(r0 + r) * ld + c0 + c;
      if (c0 + c + 4 <= K) v = *reinterpret_cast<const float4*>(s);
      else {
        if (c0 + c < K) v.x = s[0];
        if (c0 + c + 1 < K) v.y = s[1];
        if (c0 + c + 2 < K) v.z = s[2];
      }
    }
    tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  }
  __syncthreads();
  __shared__ float csum[4][64];
  if (colpart) {
    const int j = t & 63, q = t >> 6;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += tile[16 * q + r][j];
    csum[q][j] = sum;
    __syncthreads();
    if (q == 0 && c0 + j < K) colpart[(size_t)blockIdx.y * K + c0 + j] = (csum[0][j] + csum[1][j]) + (csum[2][j] + csum[3][j]);
  }
  auto emit = [&](const float (&x)[8], unsigned char* dst) {
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 h1 = (__bf16)x[e];
      const float q1 = x[e] - (float)h1;
      const __bf16 h2 = (__bf16)q1;
      const float q2 = q1 - (float)h2;
      p1[e] = h1; p2[e] = h2; p3[e] = (__bf16)q2;
    }
    *reinterpret_cast<bf16x8*>(dst) = p1;
    *reinterpret_cast<bf16x8*>(dst + TPB) = p2;
    *reinterpret_cast<bf16x8*>(dst + 2 * TPB) = p3;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g = t + 256 * i;                    // granule 0..511
    float x[8];
    if (tpN) {  // planes of src: granule = (row g >> 3, columns 8 (g & 7) ..)
      const int r = g >> 3, c = g & 7;
      const int row = r0 + r, kb = (c0 >> 4) + (c >> 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[r][8 * c + e];
      if ((row >> 5) < nrbN && kb < nkbN)
        emit(x, tpN + ((size_t)(row >> 5) * nkbN + kb) * 3 * TPB + tp_slot(row & 31, c & 1));
    }
    {  // planes of the transpose: granule = (src column g & 63, src rows 8 (g >> 6) ..)
      const int j = g & 63, c = g >> 6;
      const int row = c0 + j, kb = (r0 >> 4) + (c >> 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[8 * c + e][j];
      if ((row >> 5) < nrbT && kb < nkbT)
        emit(x, tpT + ((size_t)(row >> 5) * nkbT + kb) * 3 * TPB + tp_slot(row & 31, c & 1));
    }
  }
}

int tp_split2_parts(int rows) { return (rows + 63) / 64; }

void launch_tp_split2(const float* src, unsigned char* tpN, unsigned char* tpT, int rows, int K, int ld, float* colpart,
                      hipStream_t st) {
  dim3 grid((K + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(tp_split2_kernel, grid, dim3(256), 0, st, src, tpN, tpT, rows, K, ld, colpart);
}

size_t tp_bytes(int rows, int K) { return (size_t)((rows + 31) / 32) * ((K + 15) / 16) * 3 * TPB; }

void launch_tp_split(const float* src, unsigned char* tp, int rows, int K, int ld, bool transposed, hipStream_t st) {
  const int nkb = (K + 15) / 16;
  dim3 grid((K + 63) / 64, (rows + 31) / 32);
  hipLaunchKernelGGL(tp_split_kernel, grid, dim3(256), 0, st, src, tp, rows, K, ld, transposed ? 1 : 0, nkb);
}

// ------------------------------------------------------------------ the GEMM
// TMW = 32-row tiles per wave in M: 4 -> the 256 x 256 block tile, 3 -> 192 x 256 for M such as 576 = 3 x 192 (the
// layer-0 weight gradient: with 256-row tiles a third of its blocks would work on 64 live rows)
template <int TMW>
__global__ __launch_bounds__(512, 2) void gemm_tp_kernel(GemmTPParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int TM = 64 * TMW;               // block rows
  constexpr int NA3 = 3 * (TM / 32);         // A tiles per k-step (row blocks x parts)
  constexpr int NT = NA3 + 24;               // + B tiles
  constexpr int BUFB = NT * TPB;             // bytes of one LDS buffer
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * 256;
  const int bz = p.nbatch > 1 ? (int)(blockIdx.z % p.nbatch) : 0, zs = p.nbatch > 1 ? (int)(blockIdx.z / p.nbatch) : (int)blockIdx.z;
  const int kb0 = zs * p.kb_chunk;
  const int kb1 = min(p.kbs, kb0 + p.kb_chunk);
  const int wm = w >> 2, wn = w & 3;

  // this wave's 6 of the NT tiles of a k-step: tile ti = (operand, row block, part).  Everything here is wave-uniform
  // (SGPRs); a tile outside its matrix, or a k-block outside the operand, reads the zero tile.
  uint64_t zero = (uint64_t)g_tp_zero;
  asm volatile("" : "+s"(zero));               // keep the address in SGPRs (otherwise re-fetched from the GOT per tile)
  uint64_t tbase[6];                           // address of the tile at k-block 0, or 0 when the row block is outside the matrix
  int tshift[6], tnkb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int ti = min(w * 6 + i, NT - 1);      // (tiles past NT - 1 are never issued)
    const bool isB = ti >= NA3;
    const int t2 = isB ? ti - NA3 : ti;
    const int rbl = t2 / 3, part = t2 - 3 * rbl;
    const int rb = ((isB ? n0 : m0) >> 5) + rbl;
    const int nkb = isB ? p.nkbB : p.nkbA;
    const bool ok = rb * 32 < (isB ? p.N : p.M);
    tbase[i] = ok ? (uint64_t)(isB ? p.B : p.A) + (uint64_t)(bz ? (isB ? p.b_bstride : p.a_bstride) : 0) +
                        ((size_t)rb * nkb * 3 + part) * TPB : 0;
    tshift[i] = isB ? 0 : (bz ? p.a_kb_shift1 : p.a_kb_shift);
    tnkb[i] = ok ? nkb : 0;                     // 0: every k-block reads as zero
  }
  auto issue = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (w * 6 + i >= NT) continue;             // wave-uniform
      const int kk = kb + tshift[i];
      const uint64_t g = ((unsigned)kk < (unsigned)tnkb[i]) ? tbase[i] + (uint64_t)kk * (3 * TPB) : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + lane * 16), (lds_ptr_t)(lds + buf * BUFB + (w * 6 + i) * TPB), 16, 0, 0);
    }
  };

  f32x16 acc[TMW][2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int foff = tp_slot(lane & 31, lane >> 5);
  auto chain6 = [](f32x16 c, const bf16x8 (&x)[3], const bf16x8 (&y)[3]) {   // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], c, 0, 0, 0);
    return c;
  };
  bf16x8 a[TMW][3], b[2][3];
  auto load_a = [&](int buf, int i) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(lds + buf * BUFB + foff + ((wm * TMW + i) * 3 + q) * TPB);
  };
  auto load_b = [&](int buf, int j) {
#pragma unroll
    for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(lds + buf * BUFB + foff + (NA3 + (wn * 2 + j) * 3 + q) * TPB);
  };
#if NASR_TP_NBUF == 2
  if (kb0 < kb1) issue(kb0, 0);
  for (int kb = kb0; kb < kb1; ++kb) {
    const int buf = (kb - kb0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's tiles of step kb have landed ...
    __syncthreads();                                    // ... and everybody's; buffer buf^1 is no longer being read
    if (kb + 1 < kb1) issue(kb + 1, buf ^ 1);
#pragma unroll
    for (int i = 0; i < TMW; ++i) load_a(buf, i);
    load_b(buf, 0); load_b(buf, 1);
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = chain6(acc[i][j], a[i], b[j]);
  }
#else
  // three LDS buffers: the DMA runs two k-steps ahead and the fragments of step k+1 are read from LDS WHILE the MFMAs of
  // step k run, into the registers of fragments that have just had their last use (b[0] after the first column pass,
  // a[i] after its second, b[1] at the end) - no extra registers, and the matrix pipe does not idle after the barrier.
  int buf = 0;                                          // holds step kb
  if (kb0 < kb1) {
    issue(kb0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb0 + 1 < kb1) issue(kb0 + 1, 1);
#pragma unroll
    for (int i = 0; i < TMW; ++i) load_a(0, i);
    load_b(0, 0); load_b(0, 1);
  }
  for (int kb = kb0; kb < kb1; ++kb) {
    const int nb = buf == 2 ? 0 : buf + 1;              // holds step kb + 1 once its DMA has landed
    const int fb = nb == 2 ? 0 : nb + 1;                // held step kb - 1: every wave read it during step kb - 2 ... kb - 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my tiles of step kb + 1 (issued one iteration ago) have landed
    __builtin_amdgcn_s_barrier();                       // ... and everybody's
    asm volatile("" ::: "memory");
    if (kb + 2 < kb1) issue(kb + 2, fb);
#pragma unroll
    for (int i = 0; i < TMW; ++i) acc[i][0] = chain6(acc[i][0], a[i], b[0]);
    load_b(nb, 0);
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      acc[i][1] = chain6(acc[i][1], a[i], b[1]);
      load_a(nb, i);
    }
    load_b(nb, 1);
    // pin that interleaving (the scheduler would otherwise cluster the 18 LDS reads behind the 48 MFMAs)
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * TMW, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    buf = nb;
  }
#endif

  // epilogue: C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + 32 * ni + li;
      if (col >= p.N) continue;
      const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (32 * TMW) + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        if (p.split_k > 1) p.slabs[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[mi][ni][r];   // [slice][batch][M][N]
        else p.C[(size_t)bz * p.c_bstride + (size_t)row * p.ldc + col] = acc[mi][ni][r] + bv;
      }
    }
}

hipError_t gemm_tp_prepare() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tp_kernel<4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_TP_LDS);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tp_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            GEMM_TP_LDS);
  return e;
}

// 192-row block tiles where 256-row tiles would leave more than a tenth of their rows empty
int gemm_tp_tile_rows(int M) {
  const int w256 = (M + 255) / 256 * 256 - M, w192 = (M + 191) / 192 * 192 - M;
  return (10 * w256 > M && w192 < w256) ? 192 : 256;
}

// K split by a cost model in units of one k-step of one block (measured 2.6 us for the 256-row tile whether one or two
// blocks share a CU: they share its MFMA pipes): blocks run in rounds of 256 (one per CU), every slice keeps >= 16
// k-steps, and each slab costs a write + a read of M x N floats at ~4 TB/s.
int gemm_tp_pick_split(int M, int N, int K, int nbatch) {
  const int tm = gemm_tp_tile_rows(M);
  const int tiles = ((M + tm - 1) / tm) * ((N + 255) / 256) * (nbatch > 1 ? 2 : 1);
  const int kbs = (K + 15) / 16;
  const double kstep = 2.6e-6 * tm / 256.0;
  const double slab = (double)(nbatch > 1 ? 2 : 1) * M * N * 8.0 / 4e12 / kstep;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 64; ++s) {
    if (s > 1 && kbs / s < 16) break;
    const int per = (kbs + s - 1) / s;
    const int rounds = (tiles * s + 255) / 256;
    const double cost = (double)rounds * per + (s > 1 ? s * slab : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
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
  const int tm = g.tile_rows ? g.tile_rows : gemm_tp_tile_rows(g.M);
  p.nbatch = g.nbatch > 1 ? 2 : 1;
  p.a_bstride = (long long)g.a_bstride; p.b_bstride = (long long)g.b_bstride; p.c_bstride = (long long)g.c_bstride;
  p.a_kb_shift1 = g.a_kshift1 / 16;
  dim3 grid((g.N + 255) / 256, (g.M + tm - 1) / tm, p.split_k * p.nbatch);
  if (tm == 192) hipLaunchKernelGGL(gemm_tp_kernel<3>, grid, dim3(512), GEMM_TP_LDS, st, p);
  else hipLaunchKernelGGL(gemm_tp_kernel<4>, grid, dim3(512), GEMM_TP_LDS, st, p);
  // needs ldc == N, no bias; two batches: c_bstride == M * N (their results are adjacent, one reduction covers both)
  if (p.split_k > 1) launch_reduce_slabs(g.slabs, p.split_k, (int64_t)p.nbatch * g.M * g.N, g.C, st);
}

}  // namespace nasr
