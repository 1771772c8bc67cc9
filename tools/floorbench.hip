// floorbench.hip — launch-floor probe: T dependent launches of an (almost) empty kernel under a hipGraph
// for several grid/block shapes.  hipcc --offload-arch=gfx950 -O3 tools/floorbench.hip -o tools/fb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_empty(float* p, int s) { if (s == 123456) p[threadIdx.x] = 1.f; }
__global__ void k_store(float* p, int s) { if (threadIdx.x < 64) p[(blockIdx.x * 64 + threadIdx.x) * 4 + (s & 3)] = (float)s; }
__global__ void k_store_co(float* p, int s) { if (threadIdx.x < 64) p[blockIdx.x * 64 + threadIdx.x] = (float)s; }
__global__ void k_ldst(const float* q, float* p, int s) {
  float4 v = reinterpret_cast<const float4*>(q)[(blockIdx.x & 127) * 0 + threadIdx.x + 256 * (s & 1)];
  if (threadIdx.x < 64) p[blockIdx.x * 64 + threadIdx.x] = v.x + v.y + v.z + v.w;
}
int main() {
  float *p, *q; CK(hipMalloc(&p, 1 << 24)); CK(hipMalloc(&q, 1 << 24)); CK(hipMemset(q, 0, 1 << 24));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int T = 500;
  int shapes[][2] = {{256, 256}, {256, 64}, {128, 256}, {128, 512}, {64, 256}, {512, 128}, {256, 512}, {32, 1024}};
  for (int kind = 0; kind < 4; ++kind)
    for (auto& sh : shapes) {
      hipGraph_t g; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < T; ++s) {
        if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), 0, st, p, s);
        else if (kind == 1) hipLaunchKernelGGL(k_store, dim3(sh[0]), dim3(sh[1]), 0, st, p, s);
        else if (kind == 2) hipLaunchKernelGGL(k_store_co, dim3(sh[0]), dim3(sh[1]), 0, st, p, s);
        else hipLaunchKernelGGL(k_ldst, dim3(sh[0]), dim3(sh[1]), 0, st, s & 1 ? p : q, s & 1 ? q : p, s);
      }
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ex, st));
      CK(hipStreamSynchronize(st));
      float best = 1e9f;
      for (int i = 0; i < 5; ++i) {
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
      }
      printf("kind %d (%s) grid %4d x %4d : %.3f us/launch\n", kind,
             kind == 0 ? "empty" : kind == 1 ? "strided 4B store" : kind == 2 ? "coalesced store" : "load prev + store", sh[0], sh[1], best * 1000 / T);
      CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
  return 0;
}
