// gemmbench.hip — times the GEMM shapes of the training step (B 16, T 500, H 500, F 546).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNASR_GEMM_BK=..] tools/gemmbench.hip -o tools/sb_gemm
#include "../neuralasr_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace nasr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static float* dev_rand(size_t n) {
  std::vector<float> h(n);
  for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}
int main() {
  const int R = 8000;
  float* X = dev_rand((size_t)R * 1024);
  float* G = dev_rand((size_t)R * 4096);
  float* W = dev_rand((size_t)1024 * 4096);
  float* O = dev_rand((size_t)R * 4096);
  float* bias = dev_rand(4096);
  float* slabs; CK(hipMalloc(&slabs, (size_t)8 * 1024 * 4096 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  struct Case { const char* name; GemmDesc g; };
  std::vector<Case> cases;
  { GemmDesc g{}; g.A = X; g.B = W; g.C = O; g.M = R; g.N = 4096; g.K = 576; g.lda = 576; g.ldb = 4096; g.ldc = 4096; g.a_rows = R; g.bias = bias; g.split_k = 1; cases.push_back({"xproj  NN 8000x4096x576 ", g}); }
  { GemmDesc g{}; g.A = X; g.B = W; g.C = O; g.M = R; g.N = 4096; g.K = 1024; g.lda = 1024; g.ldb = 4096; g.ldc = 4096; g.a_rows = R; g.bias = bias; g.split_k = 1; cases.push_back({"xproj  NN 8000x4096x1024", g}); }
  { GemmDesc g{}; g.A = X; g.B = G; g.C = O; g.M = 576; g.N = 4096; g.K = R; g.lda = 576; g.ldb = 4096; g.ldc = 4096; g.a_col = true; g.a_rows = R; g.split_k = gemm_pick_split(576, 4096, R); g.slabs = slabs; cases.push_back({"dWx    TN 576x4096x8000 ", g}); }
  { GemmDesc g{}; g.A = X; g.B = G; g.C = O; g.M = 1024; g.N = 4096; g.K = R; g.lda = 1024; g.ldb = 4096; g.ldc = 4096; g.a_col = true; g.a_rows = R; g.split_k = gemm_pick_split(1024, 4096, R); g.slabs = slabs; cases.push_back({"dWx    TN 1024x4096x8000", g}); }
  { GemmDesc g{}; g.A = X; g.B = G; g.C = O; g.M = 512; g.N = 2048; g.K = R; g.lda = 1024; g.ldb = 4096; g.ldc = 2048; g.a_col = true; g.a_shift = -16; g.a_rows = R; g.split_k = gemm_pick_split(512, 2048, R); g.slabs = slabs; cases.push_back({"dU     TN 512x2048x8000 ", g}); }
  { GemmDesc g{}; g.A = G; g.B = W; g.C = O; g.M = R; g.N = 1024; g.K = 4096; g.lda = 4096; g.ldb = 4096; g.ldc = 1024; g.b_col = true; g.a_rows = R; g.split_k = 1; cases.push_back({"dX     NT 8000x1024x4096", g}); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& c : cases) {
    for (int i = 0; i < 3; ++i) launch_gemm(c.g, st);
    CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int i = 0; i < 10; ++i) {
      CK(hipEventRecord(a, st)); launch_gemm(c.g, st); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    double fl = 2.0 * c.g.M * c.g.N * c.g.K;
    printf("%s split %d : %.3f ms  %.1f TF\n", c.name, c.g.split_k, best, fl / best / 1e9);
  }
  return 0;
}
