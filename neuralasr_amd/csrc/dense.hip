// dense.hip — epilogues of the DeepSpeech dense stages (networks/deepspeech.py:43-68,106-113):
//     y = tf.nn.dropout(tf.minimum(tf.nn.relu(x W + b), relu_clip), keep_prob = 1 - p)
// The affine part runs on the GEMM kernels; these two elementwise kernels turn its output into y in place, and the
// gradient wrt y into the gradient wrt (x W + b) in place.  TensorFlow's random stream cannot be reproduced, so the
// keep-mask is DEFINED by a counter-based hash that the oracle computes identically (oracle/nasr_oracle.py
// dropout_mask): element (t, b, j) of stage `stage` on forward pass number `counter` is kept iff
//     lowbias32(((t*B + b)*W + j) ^ key) >> 8 >= floor(p * 2^24),   key = seed + 0x9E3779B9*(stage+1) + 0x85EBCA6B*counter.
// The backward pass needs neither the mask nor the pre-activation: y > 0 implies kept and relu active, and
// y < clip/(1-p) implies not clipped.
#include "kernels.h"

#include <cmath>

namespace nasr {

namespace {
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
}  // namespace

// z [R][ld] (rows r = t*Bp + b), columns [0,W) are transformed; rows with b >= B (padding) are zeroed
__global__ __launch_bounds__(256) void dense_act_kernel(float* __restrict__ z, int R, int Bp, int B, int W, int ld,
                                                        float clip, float p, float inv_keep, uint32_t key, uint32_t thr) {
  const int64_t n = (int64_t)R * W;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / W), j = (int)(e - (int64_t)r * W);
    const int t = r / Bp, b = r - t * Bp;
    float* q = z + (size_t)r * ld + j;
    if (b >= B) { *q = 0.f; continue; }
    float a = fminf(fmaxf(*q, 0.f), clip);
    if (p > 0.f) {
      const uint32_t idx = (uint32_t)(((uint32_t)t * (uint32_t)B + (uint32_t)b) * (uint32_t)W + (uint32_t)j);
      const bool keep = (lowbias32(idx ^ key) >> 8) >= thr;
      a = keep ? a * inv_keep : 0.f;
    }
    *q = a;
  }
}

__global__ __launch_bounds__(256) void dense_act_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, int64_t n,
                                                            float ymax, float inv_keep) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = y[e];
    dy[e] = (v > 0.f && v < ymax) ? dy[e] * inv_keep : 0.f;
  }
}

void launch_dense_act(float* z, int R, int Bp, int B, int W, int ld, float clip, float p, uint32_t seed, uint32_t counter,
                      int stage, hipStream_t st) {
  const uint32_t key = seed + 0x9E3779B9u * (uint32_t)(stage + 1) + 0x85EBCA6Bu * counter;
  const uint32_t thr = (uint32_t)floor((double)p * 16777216.0);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int64_t n = (int64_t)R * W;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dense_act_kernel, dim3(blocks), dim3(256), 0, st, z, R, Bp, B, W, ld, clip, p, inv_keep, key, thr);
}

// dy, y contiguous [R][ld] (padded columns hold y = 0, so their gradient becomes 0)
void launch_dense_act_bwd(float* dy, const float* y, int64_t n, float clip, float p, hipStream_t st) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dense_act_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, y, n, clip * inv_keep, inv_keep);
}

}  // namespace nasr
