// Load phase of the batched AR step's skinny GEMMs (gemm_skinny.hip, M = 64): profiles/r03_ktrace_b64_timeline.csv shows QKV /
// FFN1 spending 5.0 of their 6.1 us between the first wave's start and the last wave's first MFMA -- a workgroup requests 32 KB
// of W (its own, from HBM) and the whole 128 KB of X (shared by all 256 workgroups, L2) in one burst; the M-split out-proj
// (32 KB + 32 KB) needs 1.9 us.  Per-CU L2 bandwidth (~135 GB/s by the guide's aggregate) would make that 1.2 + latency.
// This program times the same burst in isolation, as one link of a 60-kernel dependent hipGraph chain whose X is written by the
// previous link (the real situation), and varies what could matter:
//   w_only / x_only / both     which operand costs what
//   half_x                     64 KB of X (M = 32)
//   rot                        every CU walks X in its own rotated order (L2-channel hot spot: all CUs ask for the same line at once)
//   sc1 / nt                   X loads that bypass the vector L1 / non-temporal
//   x_first                    X requested before W
//   copies8                    8 replicas of X, CUs spread over them (fewer requesters per line)
//   lds_dma                    X by global_load_lds (no VGPR return path)
//   static_x                   X never rewritten (resident in every XCD's L2 after the first launch)
//   waves8                     512-thread workgroups, half the loads per lane
// Reported per variant: chain period (us per kernel) and, from s_memrealtime stamps, start -> all loads landed per wave
// (mean / max over the waves of the last launch).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_xload tools/ubench_xload.hip && tools/bin/ubench_xload
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

enum { M_PLAIN = 0, M_ROT = 1, M_SC1 = 2, M_NT = 3, M_XFIRST = 4, M_COPIES = 5, M_LDS = 6 };

__device__ inline u64 rt() {
  u64 t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// every load is inline asm: hipcc sinks ordinary loads of __restrict__ const data past the stamp and into the predicated store
// (first version of this file: "landed" 0.08 us and an eighth of the traffic).  The explicit vmcnt(0) below is their wait.
__device__ inline u32x4 ld_plain(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline u32x4 ld_sc1(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline u32x4 ld_nt(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// NW waves; per lane NWL 16-byte loads of W and NXL of X.  X = NW * NXL KB, wave w owns KB [w * NXL, (w + 1) * NXL).
template <int NW, int NWL, int NXL, int MODE>
__global__ __launch_bounds__(NW * 64) void k_burst(const u32x4* __restrict__ w, const u32x4* __restrict__ x, u32x4* __restrict__ xout,
                                                   unsigned* __restrict__ stamps) {
  extern __shared__ u32x4 lds[];  // MODE M_LDS: NW * NXL KB (dynamic)
  const u64 t0 = rt();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wg = blockIdx.x;
  const int cu = wg >> 3;  // index within the XCD (block -> XCD = wg % 8)
  u32x4 wr[NWL > 0 ? NWL : 1], xr[(NXL > 0 && MODE != M_LDS) ? NXL : 1];
  const u32x4* wb = w + ((size_t)(wg * NW + wv) * (NWL > 0 ? NWL : 1)) * 64 + lane;
  const u32x4* xb = x + (MODE == M_COPIES ? (size_t)(cu & 7) * NW * NXL * 64 : 0) + (size_t)wv * NXL * 64 + lane;
  auto load_w = [&]() {
#pragma unroll
    for (int i = 0; i < NWL; ++i) wr[i] = ld_nt(wb + i * 64);
  };
  auto load_x = [&]() {
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int j = MODE == M_ROT ? ((i + cu) % NXL) : i;
      if constexpr (MODE == M_LDS)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + j * 64),
                                         (__attribute__((address_space(3))) void*)(lds + (wv * NXL + i) * 64), 16, 0, 0);
      else if constexpr (MODE == M_SC1) xr[i] = ld_sc1(xb + j * 64);
      else if constexpr (MODE == M_NT) xr[i] = ld_nt(xb + j * 64);
      else xr[i] = ld_plain(xb + j * 64);
    }
  };
  if constexpr (MODE == M_XFIRST) { load_x(); load_w(); } else { load_w(); load_x(); }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const u64 t1 = rt();
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < NWL; ++i) acc ^= wr[i].x ^ wr[i].y ^ wr[i].z ^ wr[i].w;
  if constexpr (MODE != M_LDS) {
#pragma unroll
    for (int i = 0; i < NXL; ++i) acc ^= xr[i].x ^ xr[i].y ^ xr[i].z ^ xr[i].w;
  } else {
#pragma unroll
    for (int i = 0; i < NXL; ++i) { const u32x4 v = lds[(wv * NXL + i) * 64 + lane]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  // the next link's X: this workgroup's share of it (NW * NXL KB over gridDim.x workgroups), a function of what was read
  const int share = (NW * NXL * 64) / (int)gridDim.x;  // u32x4 per workgroup
  if (NXL > 0 && (int)threadIdx.x < share) {
    const unsigned v = 0x3c003c00u | (acc & 0x00010001u);
    xout[(size_t)wg * share + threadIdx.x] = u32x4{v, v, v, v};
    if constexpr (MODE == M_COPIES)
      for (int c = 1; c < 8; ++c) xout[(size_t)c * NW * NXL * 64 + (size_t)wg * share + threadIdx.x] = u32x4{v, v, v, v};
  } else if (NXL == 0 && threadIdx.x == 0 && acc == 0x12345u) xout[wg] = u32x4{acc, acc, acc, acc};
  if (lane == 0) stamps[wg * NW + wv] = (unsigned)(t1 - t0);
}

template <typename K>
static void run(const char* name, K kern, int nw, size_t shmem, const u32x4* W, size_t region, size_t slab, u32x4* xa, u32x4* xb, bool static_x,
                unsigned* stamps, hipStream_t st) {
  if (shmem > 0) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int NKER = 60, REPS = 30, blocks = 256;
  const size_t nslab = region / slab;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NKER; ++i) {
    const u32x4* xin = static_x ? xa : ((i & 1) ? xb : xa);
    u32x4* xo = static_x ? xb : ((i & 1) ? xa : xb);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nw * 64), shmem, st, W + (i % nslab) * (slab / 16), xin, xo, stamps);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned> h((size_t)blocks * nw);
  CK(hipMemcpy(h.data(), stamps, h.size() * 4, hipMemcpyDeviceToHost));
  double mean = 0; unsigned mx = 0;
  for (unsigned v : h) { mean += v; mx = std::max(mx, v); }
  mean /= h.size();
  printf(" \"%s\": {\"us_per_kernel\": %.3f, \"landed_mean_us\": %.2f, \"landed_max_us\": %.2f},\n", name, ms * 1e3 / (REPS * NKER), mean * 0.01, mx * 0.01);
  fflush(stdout);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t region = 512ull << 20, slab = 8ull << 20;
  u32x4 *W, *xa, *xb;
  unsigned* stamps;
  CK(hipMalloc(&W, region + slab));
  CK(hipMemset(W, 0x3c, region + slab));
  CK(hipMalloc(&xa, 8 * (128 << 10))); CK(hipMalloc(&xb, 8 * (128 << 10)));
  CK(hipMemset(xa, 0x3c, 8 * (128 << 10))); CK(hipMemset(xb, 0x3c, 8 * (128 << 10)));
  CK(hipMalloc(&stamps, 256 * 16 * 4));
  printf("{\n");
#define RUN(name, NW, NWL, NXL, MODE, STATIC) run(name, k_burst<NW, NWL, NXL, MODE>, NW, (MODE == M_LDS ? (size_t)NW * NXL * 1024 : 0), W, region, slab, xa, xb, STATIC, stamps, st)
  RUN("w_only_32KB", 4, 8, 0, M_PLAIN, false);
  RUN("x_only_128KB", 4, 0, 32, M_PLAIN, false);
  RUN("both_32KB_128KB", 4, 8, 32, M_PLAIN, false);
  RUN("both_half_x_64KB", 4, 8, 16, M_PLAIN, false);
  RUN("both_quarter_x_32KB", 4, 8, 8, M_PLAIN, false);
  RUN("both_rot", 4, 8, 32, M_ROT, false);
  RUN("both_sc1", 4, 8, 32, M_SC1, false);
  RUN("both_nt", 4, 8, 32, M_NT, false);
  RUN("both_x_first", 4, 8, 32, M_XFIRST, false);
  RUN("both_copies8", 4, 8, 32, M_COPIES, false);
  RUN("both_lds_dma", 4, 8, 32, M_LDS, false);
  RUN("both_static_x", 4, 8, 32, M_PLAIN, true);
  RUN("both_static_x_rot", 4, 8, 32, M_ROT, true);
  RUN("both_waves8", 8, 4, 16, M_PLAIN, false);
  RUN("both_waves8_rot", 8, 4, 16, M_ROT, false);
  RUN("both_waves16", 16, 2, 8, M_PLAIN, false);
  printf(" \"note\": \"256 workgroups, 60-kernel dependent hipGraph chain; W from HBM (8 MB per launch, 512 MB walk), X written by the previous link\"\n}\n");
  return 0;
}
