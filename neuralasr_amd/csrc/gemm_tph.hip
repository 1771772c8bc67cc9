// gemm_tph.hip — the bulk GEMM of the step: fp32 products from operands kept as TWO fp16 parts per element ("tiled
// planes") and THREE MFMA products:
//     C[M,N] = A[M,K] * B[N,K]^T (+bias[n])
//
// x*s = h1 + h2 with h1 = fp16(x*s), h2 = fp16(x*s - h1): the two 11-bit significands and the sign of h2 hold x*s to
// 2^-23 relative (measured bound, tools/gemmbench.hip) - fp32's own rounding class - PROVIDED neither part leaves
// fp16's exponent range.  That is what the scale s
// is for: a power of two per row of the operand (constant along the contraction, so it factors out of the sum and the
// epilogue multiplies by 1/(s_row * s_col), exactly) chosen so that the row's largest magnitude lands in [2^14, 2^15).
// Elements down to 2^-18 of their row's maximum keep all those bits; below that the error is absolute, 2^-39 of the row
// maximum (fp16 subnormal spacing 2^-24 against a maximum of 2^15) - far under what fp32 accumulation of the same dot
// product loses.  x*y = h1*h1' + h1*h2' + h2*h1' + O(2^-22 xy), accumulated in fp32 by v_mfma_f32_32x32x16_f16: the same
// accuracy class as the six bf16 products (bf16's 8-bit exponent needs no scale, its 8-bit significand three parts), at
// half the matrix-core work and two thirds of the operand bytes.
//
// Layout: TPH[rb = row/32][kb = k/16][part 0..1][1 KiB tile = 32 rows x 16 k], tiles XOR-swizzled (tph_slot) so that the
// fragment reads are conflict-free ds_read_b128.  Kernel: 256
// (or 192) x 256 output tile, TWO k-blocks per barrier (the 3-product chain is half as long: 48 MFMAs per wave and
// barrier as before), two 64 KiB LDS buffers, 8 LDS-DMA instructions per wave and step.
#include "kernels.h"

#include <cstdlib>

namespace nasr {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int HTB = 1024;
constexpr int TPH_LDS = 2 * 64 * HTB;

__device__ __attribute__((aligned(1024))) unsigned char g_tph_zero[HTB];

__device__ __forceinline__ int tph_slot(int r, int h) { return ((r << 1) | (h ^ ((r >> 3) & 1))) << 4; }

struct GemmTPHParams {
  const unsigned char* A;
  const unsigned char* B;
  float* C;
  int M, N;
  int nkbA, nkbB;
  int kbs;
  int ldc;
  int a_kb_shift;
  const float* bias;
  const float* a_inv;      // [M] 1 / scale of A's rows
  const float* b_inv;      // [N] 1 / scale of B's rows
  int split_k, kb_chunk;
  float* slabs;
  int nbatch;
  long long a_bstride, b_bstride, c_bstride, ainv_bstride, binv_bstride;
  int a_kb_shift1;
  // XCD-aware tile order (swz != 0, 1-D grid): workgroup L runs on XCD L % 8 (observed dispatch order; speed only, any
  // placement gives the same result), so the 8 XCDs form a pr x pc x pz process grid over (row tiles, column tiles,
  // K slices x batch) and the i-th workgroup of an XCD takes tile (r fastest, then c, then z) of its own sub-grid:
  // the workgroups that run together on one XCD share their operand tiles through that XCD's L2.
  int swz, gx, gy, gz, pr, pc, pz, sr, sc, sz;    // sr x sc x sz: sub-grid of one XCD
  const int* c_map;        // (or NULL) output row m is written to row c_map[m] of C (-1: not at all): compacted rows scattered back
};

__device__ __forceinline__ void emit_h2(const float (&x)[8], float s, unsigned char* dst) {
  f16x8 p1, p2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = x[e] * s;                  // exact: s is a power of two
    const _Float16 h1 = (_Float16)v;
    p1[e] = h1;
    p2[e] = (_Float16)(v - (float)h1);
  }
  *reinterpret_cast<f16x8*>(dst) = p1;
  *reinterpret_cast<f16x8*>(dst + HTB) = p2;
}

}  // namespace

// ------------------------------------------------------------------ scales
// max |src| per row and per column of src [rows][K]: 64 x 64 pieces, partial maxima [gridDim.x][rows] / [gridDim.y][K],
// then one kernel turns maxima into (scale, 1/scale) = (2^(15-e), 2^(e-15)) with max < 2^e (frexp), 1 for an all-zero line.
__global__ __launch_bounds__(256) void absmax_part_kernel(const float* __restrict__ src, int rows, int K, int ld,
                                                          float* __restrict__ rowpart, float* __restrict__ colpart) {
  __shared__ float rm[64][5], cm[4][64];
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int cq = t & 15, rq = t >> 4;          // thread: columns 4cq..4cq+3, rows rq, rq+16, rq+32, rq+48
  float cmax[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rq + 16 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows) {
      const float* s = src + (size_t)(r0 + r) * ld + c0 + 4 * cq;
      if (c0 + 4 * cq + 4 <= K) v = *reinterpret_cast<const float4*>(s);
      else {
        if (c0 + 4 * cq < K) v.x = s[0];
        if (c0 + 4 * cq + 1 < K) v.y = s[1];
        if (c0 + 4 * cq + 2 < K) v.z = s[2];
      }
    }
    v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w);
    cmax[0] = fmaxf(cmax[0], v.x); cmax[1] = fmaxf(cmax[1], v.y); cmax[2] = fmaxf(cmax[2], v.z); cmax[3] = fmaxf(cmax[3], v.w);
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));   // row r, this thread's 4 columns: reduce over the 16 cq lanes
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
    if (cq == 0) rm[r][0] = m;
  }
  // columns: reduce over the 4 row groups of a wave (lanes 16 apart) then over the 4 waves through LDS
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float m = cmax[j];
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    if ((t & 63) < 16) cm[t >> 6][4 * cq + j] = m;
  }
  __syncthreads();
  if (t < 64) {
    if (rowpart && r0 + t < rows) rowpart[(size_t)blockIdx.x * rows + r0 + t] = rm[t][0];
    if (colpart && c0 + t < K) colpart[(size_t)blockIdx.y * K + c0 + t] = fmaxf(fmaxf(cm[0][t], cm[1][t]), fmaxf(cm[2][t], cm[3][t]));
  }
}

// 64 lines per block, 4 threads per line (partial rows q, q+4, ...) combined through LDS
__device__ __forceinline__ void scale_final_body(const float* __restrict__ part, int nparts, int n, float* __restrict__ scale,
                                                 float* __restrict__ inv, int block) {
  __shared__ float sm[4][64];
  const int cx = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = block * 64 + cx;
  float m = 0.f;
  if (i < n)
    for (int k = q; k < nparts; k += 4) m = fmaxf(m, part[(size_t)k * n + i]);
  sm[q][cx] = m;
  __syncthreads();
  if (q == 0 && i < n) {
    m = fmaxf(fmaxf(sm[0][cx], sm[1][cx]), fmaxf(sm[2][cx], sm[3][cx]));
    int e = 0;
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e); else e = 15;     // m = f * 2^e, f in [0.5, 1)
    e = max(-100, min(100, e));
    scale[i] = ldexpf(1.f, 15 - e);
    inv[i] = ldexpf(1.f, e - 15);
  }
}
__global__ __launch_bounds__(256) void scale_final_kernel(const float* __restrict__ part, int nparts, int n,
                                                          float* __restrict__ scale, float* __restrict__ inv) {
  scale_final_body(part, nparts, n, scale, inv, blockIdx.x);
}

size_t tph_scale_ws_floats(int rows, int K) { return (size_t)((K + 63) / 64) * rows + (size_t)((rows + 63) / 64) * K; }

// row_scale / row_inv [rows], col_scale / col_inv [K] (either pair may be NULL); ws: tph_scale_ws_floats(rows, K)
void launch_tph_scales_batch(const TphScaleJob* jobs, int n, float* ws, hipStream_t st);
void launch_tph_scales(const float* src, int rows, int K, int ld, float* row_scale, float* row_inv, float* col_scale,
                       float* col_inv, float* ws, hipStream_t st) {
  if (col_scale) {        // one pass + ONE finishing launch for rows and columns (the batch kernels with a single job)
    const TphScaleJob j{src, rows, K, ld, row_scale, row_inv, col_scale, col_inv};
    launch_tph_scales_batch(&j, 1, ws, st);
    return;
  }
  dim3 grid((K + 63) / 64, (rows + 63) / 64);
  float* rp = row_scale ? ws : nullptr;
  hipLaunchKernelGGL(absmax_part_kernel, grid, dim3(256), 0, st, src, rows, K, ld, rp, (float*)nullptr);
  if (row_scale) hipLaunchKernelGGL(scale_final_kernel, dim3((rows + 63) / 64), dim3(256), 0, st, rp, (int)grid.x, rows, row_scale, row_inv);
}

// ---- the same for several (small) matrices in two launches: the weights, once per optimiser step
struct ScaleJobs {
  const float* src[TPH_MAX_JOBS];
  int rows[TPH_MAX_JOBS], K[TPH_MAX_JOBS], ld[TPH_MAX_JOBS];
  float* rowpart[TPH_MAX_JOBS];   // NULL: no row scales wanted
  float* colpart[TPH_MAX_JOBS];
  float *rs[TPH_MAX_JOBS], *ri[TPH_MAX_JOBS], *cs[TPH_MAX_JOBS], *ci[TPH_MAX_JOBS];
  int gx[TPH_MAX_JOBS], gy[TPH_MAX_JOBS];
};
__global__ __launch_bounds__(256) void absmax_part_batch_kernel(ScaleJobs j) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x >= j.gx[z] || (int)blockIdx.y >= j.gy[z]) return;
  // same body as absmax_part_kernel on job z
  __shared__ float rm[64], cm[4][64];
  const float* __restrict__ src = j.src[z];
  const int rows = j.rows[z], K = j.K[z], ld = j.ld[z];
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int cq = t & 15, rq = t >> 4;
  float cmax[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rq + 16 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows) {
      const float* s = src + (size_t)(r0 + r) * ld + c0 + 4 * cq;
      if (c0 + 4 * cq + 4 <= K) v = *reinterpret_cast<const float4*>(s);
      else {
        if (c0 + 4 * cq < K) v.x = s[0];
        if (c0 + 4 * cq + 1 < K) v.y = s[1];
        if (c0 + 4 * cq + 2 < K) v.z = s[2];
      }
    }
    v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w);
    cmax[0] = fmaxf(cmax[0], v.x); cmax[1] = fmaxf(cmax[1], v.y); cmax[2] = fmaxf(cmax[2], v.z); cmax[3] = fmaxf(cmax[3], v.w);
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
    if (cq == 0) rm[r] = m;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float m = cmax[q];
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    if ((t & 63) < 16) cm[t >> 6][4 * cq + q] = m;
  }
  __syncthreads();
  if (t < 64) {
    if (j.rowpart[z] && r0 + t < rows) j.rowpart[z][(size_t)blockIdx.x * rows + r0 + t] = rm[t];
    if (c0 + t < K) j.colpart[z][(size_t)blockIdx.y * K + c0 + t] = fmaxf(fmaxf(cm[0][t], cm[1][t]), fmaxf(cm[2][t], cm[3][t]));
  }
}
__global__ __launch_bounds__(256) void scale_final_batch_kernel(ScaleJobs j) {
  const int z = blockIdx.z, which = blockIdx.y;           // which: 0 = rows, 1 = columns
  const float* part = which ? j.colpart[z] : j.rowpart[z];
  const int n = which ? j.K[z] : j.rows[z], nparts = which ? j.gy[z] : j.gx[z];
  if (!part || (int)blockIdx.x * 64 >= n) return;         // block-uniform
  scale_final_body(part, nparts, n, which ? j.cs[z] : j.rs[z], which ? j.ci[z] : j.ri[z], blockIdx.x);
}

size_t tph_scale_batch_ws_floats(const TphScaleJob* jobs, int n) {
  size_t f = 0;
  for (int i = 0; i < n; ++i) f += tph_scale_ws_floats(jobs[i].rows, jobs[i].K);
  return f;
}

void launch_tph_scales_batch(const TphScaleJob* jobs, int n, float* ws, hipStream_t st) {
  for (int j0 = 0; j0 < n; j0 += TPH_MAX_JOBS) {
    const int m = n - j0 < TPH_MAX_JOBS ? n - j0 : TPH_MAX_JOBS;
    ScaleJobs sj{};
    int gxm = 1, gym = 1, nmax = 1;
    for (int i = 0; i < m; ++i) {
      const TphScaleJob& q = jobs[j0 + i];
      sj.src[i] = q.src; sj.rows[i] = q.rows; sj.K[i] = q.K; sj.ld[i] = q.ld;
      sj.gx[i] = (q.K + 63) / 64; sj.gy[i] = (q.rows + 63) / 64;
      sj.rowpart[i] = q.row_scale ? ws : nullptr;
      sj.colpart[i] = ws + (size_t)sj.gx[i] * q.rows;
      ws += tph_scale_ws_floats(q.rows, q.K);
      sj.rs[i] = q.row_scale; sj.ri[i] = q.row_inv; sj.cs[i] = q.col_scale; sj.ci[i] = q.col_inv;
      gxm = gxm > sj.gx[i] ? gxm : sj.gx[i]; gym = gym > sj.gy[i] ? gym : sj.gy[i];
      nmax = nmax > q.rows ? nmax : q.rows; nmax = nmax > q.K ? nmax : q.K;
    }
    hipLaunchKernelGGL(absmax_part_batch_kernel, dim3(gxm, gym, m), dim3(256), 0, st, sj);
    hipLaunchKernelGGL(scale_final_batch_kernel, dim3((nmax + 63) / 64, 2, m), dim3(256), 0, st, sj);
  }
}

void launch_tph_scales_from_parts(const float* rowpart, int nrp, int rows, float* row_scale, float* row_inv, const float* colpart,
                                  int ncp, int K, float* col_scale, float* col_inv, hipStream_t st) {
  ScaleJobs sj{};
  sj.rows[0] = rows; sj.K[0] = K;
  sj.rowpart[0] = row_scale ? const_cast<float*>(rowpart) : nullptr; sj.colpart[0] = col_scale ? const_cast<float*>(colpart) : nullptr;
  sj.gx[0] = nrp; sj.gy[0] = ncp;            // scale_final_batch_kernel: parts per row line / per column line
  sj.rs[0] = row_scale; sj.ri[0] = row_inv; sj.cs[0] = col_scale; sj.ci[0] = col_inv;
  const int nmax = rows > K ? rows : K;
  hipLaunchKernelGGL(scale_final_batch_kernel, dim3((nmax + 63) / 64, 2, 1), dim3(256), 0, st, sj);
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill(float* p, float v, int n, hipStream_t st) {
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, v, n);
}

// ------------------------------------------------------------------ fp32 -> two fp16 planes
// One pass over src [rows][K]: tpN = planes of src scaled per src row (row_scale, or the constant rs when NULL), tpT =
// planes of its transpose scaled per src column (col_scale / cs); either may be NULL.  colpart as in gemm_tp.hip's fused
// pass: [gridDim.y][K] partial column sums of the UNSCALED src.
// rowmap (or NULL): logical row i of this pass is physical row rowmap[i] of src (-1: a zero row) - the COMPACTED rows of a ragged
// batch (nasr_batch.hip: only the frames t < seq_len[b], time-major); rowmap2 replaces it for the column blocks from col2 on
// (the two directions of a layer's output shifted by one frame in opposite directions: the recurrent weight gradient).
template <bool MAP>
__global__ __launch_bounds__(256) void tph_split2_kernel(const float* __restrict__ src, unsigned char* __restrict__ tpN,
                                                         unsigned char* __restrict__ tpT, int rows, int K, int ld,
                                                         const float* __restrict__ row_scale, float rs,
                                                         const float* __restrict__ col_scale, float cs,
                                                         float* __restrict__ colpart, const int* __restrict__ rowmap,
                                                         const int* __restrict__ rowmap2, int col2) {
  __shared__ float tile[64][65];
  __shared__ float csum[4][64];
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int nkbN = (K + 15) / 16, nrbN = (rows + 31) / 32;
  const int nkbT = (rows + 15) / 16, nrbT = (K + 31) / 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows) {
      int pr = r0 + r;
      if constexpr (MAP) pr = ((rowmap2 && c0 >= col2) ? rowmap2 : rowmap)[r0 + r];
      if (pr >= 0) {
        const float* s = src + (size_t)pr * ld + c0 + c;
        if (c0 + c + 4 <= K) v = *reinterpret_cast<const float4*>(s);
        else {
          if (c0 + c < K) v.x = s[0];
          if (c0 + c + 1 < K) v.y = s[1];
          if (c0 + c + 2 < K) v.z = s[2];
        }
      }
    }
    tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  }
  __syncthreads();
  if (colpart) {
    const int j = t & 63, q = t >> 6;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += tile[16 * q + r][j];
    csum[q][j] = sum;
    __syncthreads();
    if (q == 0 && c0 + j < K) colpart[(size_t)blockIdx.y * K + c0 + j] = (csum[0][j] + csum[1][j]) + (csum[2][j] + csum[3][j]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g = t + 256 * i;
    float x[8];
    if (tpN) {
      const int r = g >> 3, c = g & 7;
      const int row = r0 + r, kb = (c0 >> 4) + (c >> 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[r][8 * c + e];
      if ((row >> 5) < nrbN && kb < nkbN) {
        const float s = row_scale ? (row < rows ? row_scale[row] : 1.f) : rs;
        emit_h2(x, s, tpN + ((size_t)(row >> 5) * nkbN + kb) * 2 * HTB + tph_slot(row & 31, c & 1));
      }
    }
    if (tpT) {
      const int j = g & 63, c = g >> 6;
      const int row = c0 + j, kb = (r0 >> 4) + (c >> 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[8 * c + e][j];
      if ((row >> 5) < nrbT && kb < nkbT) {
        const float s = col_scale ? (row < K ? col_scale[row] : 1.f) : cs;
        emit_h2(x, s, tpT + ((size_t)(row >> 5) * nkbT + kb) * 2 * HTB + tph_slot(row & 31, c & 1));
      }
    }
  }
}

int tp_split2_parts(int rows) { return (rows + 63) / 64; }

// 192-row block tiles where 256-row tiles would leave more than a tenth of their rows empty
int gemm_tp_tile_rows(int M) {
  const int w256 = (M + 255) / 256 * 256 - M, w192 = (M + 191) / 192 * 192 - M;
  return (10 * w256 > M && w192 < w256) ? 192 : 256;
}

size_t tph_bytes(int rows, int K) { return (size_t)((rows + 31) / 32) * ((K + 15) / 16) * 2 * HTB; }

void launch_tph_split2(const float* src, unsigned char* tpN, unsigned char* tpT, int rows, int K, int ld,
                       const float* row_scale, float rs, const float* col_scale, float cs, float* colpart, hipStream_t st,
                       const int* rowmap, const int* rowmap2, int col2) {
  dim3 grid((K + 63) / 64, (rows + 63) / 64);
  if (rowmap)
    hipLaunchKernelGGL(tph_split2_kernel<true>, grid, dim3(256), 0, st, src, tpN, tpT, rows, K, ld, row_scale, rs, col_scale, cs,
                       colpart, rowmap, rowmap2, col2);
  else
    hipLaunchKernelGGL(tph_split2_kernel<false>, grid, dim3(256), 0, st, src, tpN, tpT, rows, K, ld, row_scale, rs, col_scale, cs,
                       colpart, rowmap, rowmap2, col2);
}

// dst[i] = map[i] >= 0 ? src[map[i]] : fill  (row scales of a compacted operand)
__global__ __launch_bounds__(256) void gather_rows_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                          const int* __restrict__ map, int n, float fill) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = map[i] >= 0 ? src[map[i]] : fill;
}
void launch_gather_rows(float* dst, const float* src, const int* map, int n, float fill, hipStream_t st) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, map, n, fill);
}

// ------------------------------------------------------------------ the GEMM
// WM x WN waves, each with a (32 TMW) x 64 piece of the (32 TMW WM) x (64 WN) block tile.  <4,2,4> / <3,2,4>: the 256 (192) x
// 256 tiles of the step, 8 waves, one block per CU.  <4,1,3>: a 128 x 192 tile on THREE waves and 80 KB of LDS - what fits on
// a CU beside a workgroup of the persistent recurrence (lstm_persist.hip: 5 waves, two of them on one SIMD), for weight
// gradients that run under the BPTT launch of the layer below (nasr_api.hip, weight_grads on the side stream).
// DBG (diagnostics of the co-residency experiment, NASR_SIDE_DBG, side launches only): 1 = no operand DMA after the first
// step, 2 = no MFMAs - results are garbage, timing only.
// CMAP: the result rows are scattered by p.c_map (compacted rows of a ragged batch) - its own instantiations: the test per
// stored element cost the plain epilogue 3 us a launch.
template <int TMW, int WM, int WN, int DBG = 0, bool CMAP = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN + 3) / 4) void gemm_tph_kernel(GemmTPHParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int TM = 32 * TMW * WM, TN = 64 * WN;
  constexpr int NWV = WM * WN;               // waves
  constexpr int NA = TM / 32;                // A row blocks
  constexpr int NRB = NA + 2 * WN;           // + B row blocks
  constexpr int NT = NRB * 2 * 2;            // tiles per step: 2 k-blocks x row blocks x 2 parts
  constexpr int NW = (NT + NWV - 1) / NWV;   // DMA instructions per wave and step
  constexpr int BUFB = NT * HTB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx = blockIdx.x, by = blockIdx.y, bzz = blockIdx.z;
  if (p.swz) {
    const int L = blockIdx.x, x = L & 7, i = L >> 3;
    const int xr = x % p.pr, xc = (x / p.pr) % p.pc, xz = x / (p.pr * p.pc);
    const int r = i % p.sr, c = (i / p.sr) % p.sc, z = i / (p.sr * p.sc);
    by = xr * p.sr + r; bx = xc * p.sc + c; bzz = xz * p.sz + z;
    if (by >= p.gy || bx >= p.gx || bzz >= p.gz) return;      // padding of an uneven sub-grid (block-uniform)
  }
  const int m0 = by * TM, n0 = bx * TN;
  const int bz = p.nbatch > 1 ? (int)(bzz % p.nbatch) : 0, zs = p.nbatch > 1 ? (int)(bzz / p.nbatch) : bzz;
  const int kb0 = zs * p.kb_chunk;                       // even
  const int kb1 = min(p.kbs, kb0 + p.kb_chunk);
  const int wm = w / WN, wn = w % WN;

  uint64_t zero = (uint64_t)g_tph_zero;
  asm volatile("" : "+s"(zero));
  // tile ti = ((kk * NRB + row block) * 2 + part); everything wave-uniform
  uint64_t tbase[NW];
  int tshift[NW], tnkb[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int ti = min(w * NW + i, NT - 1);
    const int kk = ti / (NRB * 2), rem = ti - kk * (NRB * 2);
    const int rbi = rem >> 1, part = rem & 1;
    const bool isB = rbi >= NA;
    const int rb = ((isB ? n0 : m0) >> 5) + (isB ? rbi - NA : rbi);
    const int nkb = isB ? p.nkbB : p.nkbA;
    const bool ok = rb * 32 < (isB ? p.N : p.M);
    tbase[i] = ok ? (uint64_t)(isB ? p.B : p.A) + (uint64_t)(bz ? (isB ? p.b_bstride : p.a_bstride) : 0) +
                        ((size_t)rb * nkb * 2 + part) * HTB : 0;
    tshift[i] = kk + (isB ? 0 : (bz ? p.a_kb_shift1 : p.a_kb_shift));
    tnkb[i] = ok ? nkb : 0;
  }
  auto issue = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      if (w * NW + i >= NT) continue;
      const int kk = kb + tshift[i];
      const uint64_t g = ((unsigned)kk < (unsigned)tnkb[i]) ? tbase[i] + (uint64_t)kk * (2 * HTB) : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + lane * 16), (lds_ptr_t)(lds + buf * BUFB + (w * NW + i) * HTB), 16, 0, 0);
    }
  };

  f32x16 acc[TMW][2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int foff = tph_slot(lane & 31, lane >> 5);
  auto chain3 = [](f32x16 c, const f16x8 (&x)[2], const f16x8 (&y)[2]) {   // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], y[0], c, 0, 0, 0);
    return c;
  };

  if (kb0 < kb1) issue(kb0, 0);
  for (int kb = kb0; kb < kb1; kb += 2) {
    const int buf = ((kb - kb0) >> 1) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + 2 < kb1 && !(DBG & 1)) issue(kb + 2, buf ^ 1);   // (issued after the first fragment reads instead: 2 % slower)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 a[TMW][2], b[2][2];
      const unsigned char* base = lds + buf * BUFB + kk * (NRB * 2 * HTB) + foff;
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) a[i][q] = *reinterpret_cast<const f16x8*>(base + ((wm * TMW + i) * 2 + q) * HTB);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[j][q] = *reinterpret_cast<const f16x8*>(base + ((NA + wn * 2 + j) * 2 + q) * HTB);
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (DBG & 2) acc[i][j][0] += (float)a[i][0][0] * (float)b[j][1][0];
          else acc[i][j] = chain3(acc[i][j], a[i], b[j]);
        }
    }
  }

  const int li = lane & 31, lh = lane >> 5;
  const float* ainv = p.a_inv + (bz ? p.ainv_bstride : 0);
  const float* binv = p.b_inv + (bz ? p.binv_bstride : 0);
#pragma unroll
  for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + 32 * ni + li;
      if (col >= p.N) continue;
      const float bv = (p.bias && p.split_k == 1) ? p.bias[col] : 0.f;
      const float sb = binv[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (32 * TMW) + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        const float v = acc[mi][ni][r] * (ainv[row] * sb);        // powers of two: exact
        if (p.split_k > 1) {
          p.slabs[((size_t)bzz * p.M + row) * p.N + col] = v;
        } else {
          if constexpr (CMAP) {
            const int orow = p.c_map[row];
            if (orow >= 0) p.C[(size_t)bz * p.c_bstride + (size_t)orow * p.ldc + col] = v + bv;
          } else {
            p.C[(size_t)bz * p.c_bstride + (size_t)row * p.ldc + col] = v + bv;
          }
        }
      }
    }
}

constexpr int TPH_LDS_SIDE = 2 * 40 * HTB;     // <4,1,3>: (4 + 6) row blocks x 2 k-blocks x 2 parts, two buffers

hipError_t gemm_tph_prepare() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tph_kernel<4, 2, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, TPH_LDS);
  for (const void* f : {reinterpret_cast<const void*>(&gemm_tph_kernel<3, 2, 4>), reinterpret_cast<const void*>(&gemm_tph_kernel<4, 2, 4, 0, true>),
                        reinterpret_cast<const void*>(&gemm_tph_kernel<3, 2, 4, 0, true>)})
    if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, TPH_LDS);
  for (const void* f : {reinterpret_cast<const void*>(&gemm_tph_kernel<4, 1, 3>), reinterpret_cast<const void*>(&gemm_tph_kernel<4, 1, 3, 1>),
                        reinterpret_cast<const void*>(&gemm_tph_kernel<4, 1, 3, 2>), reinterpret_cast<const void*>(&gemm_tph_kernel<4, 1, 3, 3>)})
    if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, TPH_LDS_SIDE);
  return e;
}

// K split by a cost model in units of one k-step of one block: blocks run in rounds of 256 (one per CU), every slice keeps
// >= 32 k-blocks, and each slab costs a write + a read of M x N floats at ~4 TB/s; slices are even numbers of k-blocks
int gemm_tph_pick_split(int M, int N, int K, int nbatch, bool side) {
  const int tm = side ? 128 : gemm_tp_tile_rows(M), tn = side ? 192 : 256;
  const int tiles = ((M + tm - 1) / tm) * ((N + tn - 1) / tn) * (nbatch > 1 ? 2 : 1);
  const int kbs = (K + 15) / 16;
  const double kstep = side ? 0.9e-6 : 1.3e-6 * tm / 256.0;     // one step = two k-blocks of one block
  const double slab = (double)(nbatch > 1 ? 2 : 1) * M * N * 8.0 / 4e12 / kstep;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 64; ++s) {
    if (s > 1 && kbs / s < 32) break;
    const int per = ((kbs + s - 1) / s + 1) & ~1;
    const int rounds = (tiles * s + 255) / 256;
    const double cost = (double)rounds * per + (s > 1 ? s * slab : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
}

void launch_gemm_tph(const GemmTPHDesc& g, hipStream_t st) {
  GemmTPHParams p;
  p.A = g.A; p.B = g.B; p.C = g.C; p.M = g.M; p.N = g.N;
  p.nkbA = g.nkbA; p.nkbB = g.nkbB; p.kbs = (g.K + 15) / 16; p.ldc = g.ldc;
  p.a_kb_shift = g.a_kshift / 16;
  p.bias = g.bias; p.a_inv = g.a_inv; p.b_inv = g.b_inv;
  int split = g.split_k < 1 ? 1 : g.split_k;
  const int per = ((p.kbs + split - 1) / split + 1) & ~1;       // even: a step is two k-blocks
  p.kb_chunk = per;
  p.split_k = (p.kbs + per - 1) / per;
  p.slabs = g.slabs;
  const int tm = g.side ? 128 : g.tile_rows ? g.tile_rows : gemm_tp_tile_rows(g.M), tn = g.side ? 192 : 256;
  p.nbatch = g.nbatch > 1 ? 2 : 1;
  p.a_bstride = (long long)g.a_bstride; p.b_bstride = (long long)g.b_bstride; p.c_bstride = (long long)g.c_bstride;
  p.ainv_bstride = (long long)g.ainv_bstride; p.binv_bstride = (long long)g.binv_bstride;
  p.a_kb_shift1 = g.a_kshift1 / 16;
  dim3 grid((g.N + tn - 1) / tn, (g.M + tm - 1) / tm, p.split_k * p.nbatch);
  p.c_map = g.c_map;
  p.swz = 0;
  p.gx = (int)grid.x; p.gy = (int)grid.y; p.gz = (int)grid.z;
  p.pr = p.pc = p.pz = 1; p.sr = p.gy; p.sc = p.gx; p.sz = p.gz;
  {
    static const int mode = [] { const char* e = getenv("NASR_GEMM_SWZ"); return e ? atoi(e) : 1; }();
    // process grid with the least fabric traffic pc * |A| + pr * |B| (an XCD fetches the A tiles of its rows once for all
    // of its columns and vice versa; K slices replicate nothing) among those that pad the grid by at most 1/8
    double best = 1e300;
    const long long total = (long long)p.gx * p.gy * p.gz;
    for (int pz = 1; pz <= 8 && mode; pz *= 2)
      for (int pr = 1; pr * pz <= 8; pr *= 2) {
        const int pc = 8 / (pz * pr);
        const int sr = (p.gy + pr - 1) / pr, sc = (p.gx + pc - 1) / pc, sz = (p.gz + pz - 1) / pz;
        const long long padded = 8LL * sr * sc * sz;
        if (padded * 8 > total * 9) continue;
        const double cost = (double)pc * g.M + (double)pr * g.N + 1e-3 * (padded - total);
        if (cost < best) { best = cost; p.swz = 1; p.pr = pr; p.pc = pc; p.pz = pz; p.sr = sr; p.sc = sc; p.sz = sz; }
      }
    if (p.swz) grid = dim3(8 * p.sr * p.sc * p.sz, 1, 1);
  }
  static const int side_dbg = [] { const char* e = getenv("NASR_SIDE_DBG"); return e ? atoi(e) : 0; }();
  // rows scattered in the epilogue: main-stream products only (gemm_xproj, gemm_dx); a K-split one scatters in its reduction
  const bool scatter = g.c_map && p.split_k == 1 && !g.side;
  if (g.side && side_dbg == 1) hipLaunchKernelGGL((gemm_tph_kernel<4, 1, 3, 1>), grid, dim3(192), TPH_LDS_SIDE, st, p);
  else if (g.side && side_dbg == 2) hipLaunchKernelGGL((gemm_tph_kernel<4, 1, 3, 2>), grid, dim3(192), TPH_LDS_SIDE, st, p);
  else if (g.side && side_dbg == 3) hipLaunchKernelGGL((gemm_tph_kernel<4, 1, 3, 3>), grid, dim3(192), TPH_LDS_SIDE, st, p);
  else if (g.side) hipLaunchKernelGGL((gemm_tph_kernel<4, 1, 3>), grid, dim3(192), TPH_LDS_SIDE, st, p);
  else if (scatter && tm == 192) hipLaunchKernelGGL((gemm_tph_kernel<3, 2, 4, 0, true>), grid, dim3(512), TPH_LDS, st, p);
  else if (scatter) hipLaunchKernelGGL((gemm_tph_kernel<4, 2, 4, 0, true>), grid, dim3(512), TPH_LDS, st, p);
  else if (tm == 192) hipLaunchKernelGGL((gemm_tph_kernel<3, 2, 4>), grid, dim3(512), TPH_LDS, st, p);
  else hipLaunchKernelGGL((gemm_tph_kernel<4, 2, 4>), grid, dim3(512), TPH_LDS, st, p);
  if (p.split_k > 1) {
    if (g.c_map) launch_reduce_slabs_rows(g.slabs, p.split_k, g.M, g.N, g.ldc, g.c_map, g.C, st);
    else launch_reduce_slabs(g.slabs, p.split_k, (int64_t)p.nbatch * g.M * g.N, g.C, st);
  }
}

}  // namespace nasr
