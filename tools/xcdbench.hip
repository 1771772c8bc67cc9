// xcdbench.hip — price of a per-timestep all-gather inside ONE persistent launch, 32 workgroups per group
// (one group per XCD), data-tagged 8-byte granules (cdna_hip_programming.md Guideline 16, R2).
//   mode 0: sc1 stores + sc1 loads, groups = workgroups that report the same HW_REG_XCC_ID
//   mode 1: plain stores + sc1 loads, same grouping (same-XCD only: the XCD's L2 is the meeting point)
//   mode 2: sc1 stores + sc1 loads, groups deliberately spread over all 8 XCDs
// Every granule read is checked against its expected value (stale or torn reads are counted).
// hipcc --offload-arch=gfx950 -O3 tools/xcdbench.hip -o tools/sb_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef float f4 __attribute__((ext_vector_type(4)));

struct Ctl { unsigned xcc_count[8]; unsigned timeout; unsigned errors; unsigned spins; unsigned pad[5]; unsigned census[256]; };

__device__ __forceinline__ unsigned gval(unsigned s, unsigned row, unsigned l) {
  return ((s * 2654435761u) ^ (row * 40503u + l * 9176u + 12345u)) & 0x3fffffffu;
}

#define MF(v, a) acc[(a) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(A[v]), w[(v) * 16 + (a)], acc[(a) & 3], 4, a, 0);
#define MF16(v) MF(v,0) MF(v,1) MF(v,2) MF(v,3) MF(v,4) MF(v,5) MF(v,6) MF(v,7) MF(v,8) MF(v,9) MF(v,10) MF(v,11) MF(v,12) MF(v,13) MF(v,14) MF(v,15)

template <int MODE, int NMF>
__global__ __launch_bounds__(256, 1) void xb(unsigned long long* gran, Ctl* ctl, const float* wsrc, int steps, float* sink) {
  __shared__ unsigned sh_info[4];
  __shared__ float big[24 * 1024];                 // 96 KB: one workgroup per CU
  __shared__ float red[4][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;     // HW_REG_XCC_ID[3:0]
    const unsigned slot = atomicAdd(&ctl->xcc_count[xcc & 7], 1u);
    sh_info[0] = xcc; sh_info[1] = slot;
    ctl->census[blockIdx.x] = (xcc << 8) | slot;
  }
  big[tid] = 0.f;
  __syncthreads();
  const unsigned xcc = sh_info[0], slot = sh_info[1];
  unsigned group, member;
  if (MODE == 2) { group = slot & 7; member = xcc * 4 + (slot >> 3); }
  else           { group = xcc; member = slot; }
  if (slot >= 32 || xcc >= 8) { if (tid == 0) atomicOr(&ctl->timeout, 2u); return; }
  gu64* base = (gu64*)(gran + (size_t)group * 2 * 32 * 64);
  float w[NMF > 0 ? NMF : 1];
#pragma unroll
  for (int i = 0; i < NMF; ++i) w[i] = wsrc[(blockIdx.x * 4 + wave) * 128 * 64 + i * 64 + lane];
  f4 acc[4];
  unsigned nerr = 0, nspin = 0;
  bool dead = false;
  for (int s = 1; s <= steps && !dead; ++s) {
    unsigned A[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    if (s > 1) {
      gu64* src = base + (size_t)((s - 1) & 1) * 32 * 64 + (wave * 8) * 64 + lane;
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned long long x = __hip_atomic_load(src + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          A[k] = (unsigned)x; ok &= (unsigned)(x >> 32) == (unsigned)(s - 1);
        }
        if (__all(ok)) break;
        ++nspin;
        if (spins > (1u << 18)) { dead = true; break; }
      }
      if (!dead) {
#pragma unroll
        for (int k = 0; k < 8; ++k) nerr += A[k] != gval(s - 1, wave * 8 + k, lane);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) A[k] = 0;
    }
    if (NMF >= 128) { MF16(0) MF16(1) MF16(2) MF16(3) MF16(4) MF16(5) MF16(6) MF16(7) }
    else if (NMF >= 64) { MF16(0) MF16(1) MF16(2) MF16(3) }
    const f4 r = acc[0] + acc[1] + acc[2] + acc[3];
    red[wave][lane * 4 + 0] = r.x; red[wave][lane * 4 + 1] = r.y; red[wave][lane * 4 + 2] = r.z; red[wave][lane * 4 + 3] = r.w;
    dead = __syncthreads_or(dead);
    if (dead) break;
    if (wave == 0) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) t += red[q][lane] + red[q][lane + 64] + red[q][lane + 128] + red[q][lane + 192];
      unsigned v = gval(s, member, lane);
      if (t == 123456.789f) v ^= 1;              // keeps the MFMA chain live
      const unsigned long long g = ((unsigned long long)(unsigned)s << 32) | v;
      gu64* dst = base + (size_t)(s & 1) * 32 * 64 + member * 64 + lane;
      if (MODE == 1) *dst = g;                   // plain store: stays in this XCD's L2
      else __hip_atomic_store(dst, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (dead && tid == 0) atomicOr(&ctl->timeout, 1u);
  if (nerr) atomicAdd(&ctl->errors, nerr);
  if (lane == 0 && nspin) atomicAdd(&ctl->spins, nspin);
  if (big[tid] == 7.f) sink[tid] = acc[0].x;
}


// modes 3/4/5: untagged 4-byte payload rows + one monotonic flag word per producer.
//   3: plain stores, every wave polls the 8 flags of the producers whose rows it loads (no workgroup sync before the loads)
//   4: plain stores, wave 0 polls all 32 flags, then a workgroup barrier, then every wave loads
//   5: as 3 with sc1 (write-through) stores
#define MG(j, r, a) acc[(a) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(P[j][r]), w[((j) * 4 + (r)) * 16 + (a)], acc[(a) & 3], 4, a, 0);
#define MG16(j, r) MG(j,r,0) MG(j,r,1) MG(j,r,2) MG(j,r,3) MG(j,r,4) MG(j,r,5) MG(j,r,6) MG(j,r,7) MG(j,r,8) MG(j,r,9) MG(j,r,10) MG(j,r,11) MG(j,r,12) MG(j,r,13) MG(j,r,14) MG(j,r,15)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u4 gu128;
template <int MODE, int NMF>
__global__ __launch_bounds__(256, 1) void xf(unsigned* pay, unsigned* flags, Ctl* ctl, const float* wsrc, int steps, float* sink) {
  __shared__ unsigned sh_info[4];
  __shared__ float big[24 * 1024];
  __shared__ float red[4][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    const unsigned slot = atomicAdd(&ctl->xcc_count[xcc & 7], 1u);
    sh_info[0] = xcc; sh_info[1] = slot;
  }
  big[tid] = 0.f;
  __syncthreads();
  const unsigned group = sh_info[0], member = sh_info[1];
  if (member >= 32 || group >= 8) { if (tid == 0) atomicOr(&ctl->timeout, 2u); return; }
  unsigned* gpay = pay + (size_t)group * 2 * 32 * 64;          // [parity][row][64]
  gu32* gflag = (gu32*)(flags + group * 64);                    // 32 flags (+ pad)
  float w[NMF > 0 ? NMF : 1];
#pragma unroll
  for (int i = 0; i < NMF; ++i) w[i] = wsrc[(blockIdx.x * 4 + wave) * 128 * 64 + i * 64 + lane];
  f4 acc[4];
  unsigned nerr = 0, nspin = 0;
  bool dead = false;
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sAB = 0, sBC = 0, sCD = 0, sDE = 0, sEA = 0;
  for (int s = 1; s <= steps && !dead; ++s) {
    u4 P[2];
    tC = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    if (s > 1) {
      if (MODE == 4) {
        if (wave == 0) {
          for (unsigned spins = 0;; ++spins) {
            const unsigned f = lane < 32 ? __hip_atomic_load(gflag + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)(s - 1);
            if (__all((int)(f - (unsigned)(s - 1)) >= 0)) break;
            ++nspin;
            if (spins > (1u << 20)) { dead = true; break; }
          }
        }
        dead = __syncthreads_or(dead);
      } else {
        for (unsigned spins = 0;; ++spins) {
          const unsigned f = lane < 8 ? __hip_atomic_load(gflag + wave * 8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)(s - 1);
          if (__all((int)(f - (unsigned)(s - 1)) >= 0)) break;
          ++nspin;
          if (spins > (1u << 20)) { dead = true; break; }
        }
      }
      tD = __builtin_amdgcn_s_memtime();
      if (!dead) {
        gu128* src = (gu128*)(gpay + (size_t)((s - 1) & 1) * 32 * 64 + wave * 8 * 64) + lane;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(P[0]), "=&v"(P[1]) : "v"(src) : "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) nerr += P[j][r] != gval(s - 1, wave * 8 + j * 4 + (lane >> 4), (lane & 15) * 4 + r);
      }
    } else { P[0] = u4{0, 0, 0, 0}; P[1] = u4{0, 0, 0, 0}; }
    tE = __builtin_amdgcn_s_memtime();
    if (s > 2) { sBC += tC - tB; sCD += tD - tC; sDE += tE - tD; }
    if (NMF >= 128) { MG16(0,0) MG16(0,1) MG16(0,2) MG16(0,3) MG16(1,0) MG16(1,1) MG16(1,2) MG16(1,3) }
    const f4 r = acc[0] + acc[1] + acc[2] + acc[3];
    red[wave][lane * 4 + 0] = r.x; red[wave][lane * 4 + 1] = r.y; red[wave][lane * 4 + 2] = r.z; red[wave][lane * 4 + 3] = r.w;
    dead = __syncthreads_or(dead);
    if (dead) break;
    if (wave == 0) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) t += red[q][lane] + red[q][lane + 64] + red[q][lane + 128] + red[q][lane + 192];
      unsigned v = gval(s, member, lane);
      if (t == 123456.789f) v ^= 1;
      gu32* dst = (gu32*)(gpay + (size_t)(s & 1) * 32 * 64 + member * 64 + lane);
      tA = __builtin_amdgcn_s_memtime();
      if (s > 2) sEA += tA - tE;
      if (MODE == 5) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *dst = v;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tB = __builtin_amdgcn_s_memtime();
      if (s > 2) sAB += tB - tA;
      if (lane == 0) {
        if (MODE == 5) __hip_atomic_store(gflag + member, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *(gflag + member) = (unsigned)s;
      }
    }
    __syncthreads();
  }
  if (dead && tid == 0) atomicOr(&ctl->timeout, 1u);
  if (nerr) atomicAdd(&ctl->errors, nerr);
  if (lane == 0 && nspin) atomicAdd(&ctl->spins, nspin);
  if (big[tid] == 7.f) sink[tid] = acc[0].x;
  if (tid == 0 && group == 0 && member == 0) {
    unsigned long long* o = (unsigned long long*)ctl->census;
    o[0] = sAB; o[1] = sBC; o[2] = sCD; o[3] = sDE; o[4] = sEA;
  }
}

template <int MODE, int NMF>
static void runf(const char* name, unsigned long long* gran, Ctl* ctl, float* wsrc, float* sink, int steps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f; Ctl h;
  unsigned* pay = (unsigned*)gran; unsigned* flags = pay + 8 * 2 * 32 * 64;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(gran, 0, 8 * 2 * 32 * 64 * 8));
    CK(hipEventRecord(a)); hipLaunchKernelGGL((xf<MODE, NMF>), dim3(256), dim3(256), 0, 0, pay, flags, ctl, wsrc, steps, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best) best = ms;
    CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    if (h.timeout || h.errors) break;
  }
  printf("%-34s mfma/wave %3d : %.3f us/step  timeout %u errors %u spins/step/wave %.2f\n", name, NMF, best * 1000 / steps,
         h.timeout, h.errors, (double)h.spins / steps / 1024);
  const unsigned long long* o = (const unsigned long long*)h.census;
  printf("    wave0 of one workgroup, ticks (100 MHz?) per step: store->ack %.1f  ack->loop top %.1f  poll %.1f  payload load %.1f  compute+reduce %.1f\n",
         (double)o[0] / (steps - 2), (double)o[1] / (steps - 2), (double)o[2] / (steps - 2), (double)o[3] / (steps - 2), (double)o[4] / (steps - 2));
}

template <int MODE, int NMF>
static void run(const char* name, unsigned long long* gran, Ctl* ctl, float* wsrc, float* sink, int steps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f; Ctl h;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(gran, 0, 8 * 2 * 32 * 64 * 8));
    CK(hipEventRecord(a)); hipLaunchKernelGGL((xb<MODE, NMF>), dim3(256), dim3(256), 0, 0, gran, ctl, wsrc, steps, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best) best = ms;
    CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    if (h.timeout || h.errors) break;
  }
  printf("%-34s mfma/wave %3d : %.3f us/step  timeout %u errors %u spins/step/wave %.2f  xcc counts", name, NMF, best * 1000 / steps,
         h.timeout, h.errors, (double)h.spins / steps / 1024);
  for (int i = 0; i < 8; ++i) printf(" %u", h.xcc_count[i]);
  printf("\n");
}

int main() {
  unsigned long long* gran; Ctl* ctl; float *wsrc, *sink;
  CK(hipMalloc(&gran, 8 * 2 * 32 * 64 * 8)); CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&wsrc, 256 * 4 * 128 * 64 * 4)); CK(hipMalloc(&sink, 4096));
  float* hw = (float*)malloc(256 * 4 * 128 * 64 * 4);
  for (int i = 0; i < 256 * 4 * 128 * 64; ++i) hw[i] = (float)((i * 7919) % 1000) * 1e-3f - 0.5f;
  CK(hipMemcpy(wsrc, hw, 256 * 4 * 128 * 64 * 4, hipMemcpyHostToDevice));
  const int steps = 2000;
  run<0, 0>("sc1 store/sc1 load, same XCD", gran, ctl, wsrc, sink, steps);
  run<1, 0>("plain store/sc1 load, same XCD", gran, ctl, wsrc, sink, steps);
  run<2, 0>("sc1 store/sc1 load, cross XCD", gran, ctl, wsrc, sink, steps);
  run<0, 128>("sc1 store/sc1 load, same XCD", gran, ctl, wsrc, sink, steps);
  run<1, 128>("plain store/sc1 load, same XCD", gran, ctl, wsrc, sink, steps);
  run<2, 128>("sc1 store/sc1 load, cross XCD", gran, ctl, wsrc, sink, steps);
  run<0, 64>("sc1 store/sc1 load, same XCD", gran, ctl, wsrc, sink, steps);
  run<1, 64>("plain store/sc1 load, same XCD", gran, ctl, wsrc, sink, steps);
  runf<3, 0>("flags+payload plain, wave polls 8", gran, ctl, wsrc, sink, steps);
  runf<4, 0>("flags+payload plain, wave0 polls 32", gran, ctl, wsrc, sink, steps);
  runf<5, 0>("flags+payload sc1, wave polls 8", gran, ctl, wsrc, sink, steps);
  runf<3, 128>("flags+payload plain, wave polls 8", gran, ctl, wsrc, sink, steps);
  runf<4, 128>("flags+payload plain, wave0 polls 32", gran, ctl, wsrc, sink, steps);
  runf<5, 128>("flags+payload sc1, wave polls 8", gran, ctl, wsrc, sink, steps);
  return 0;
}
