// Micro-benchmark behind the batch-1 AR-step design (DESIGN.md "what bounds a decode step"):
// a decode step is a chain of ~60 dependent kernels, each streaming 2-8 MB of weights once.
//   1. chain of empty kernels                    -> the launch-boundary floor per kernel
//   2. chain of load->store kernels              -> + one dependent memory round trip
//   3. chain of weight-streaming GEMV-like kernels over a rotating 320 MB region (HBM-cold)
//   4. the same chain where kernel k also prefetches kernel k+1's slab (same block -> slab map,
//      so the lines land in the L2 of the XCD that will read them; or a shifted map: MALL only)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o gpurun_out/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }

__global__ void k_loadstore(const float* in, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] + 1.0f;
}

// Each wave reads NV 16-byte vectors per lane from its slab (wave w: contiguous NV KiB), reduces,
// lane 0 stores.  x comes from the previous kernel's output (dependency).  Optionally every lane
// also requests PV vectors of the NEXT kernel's slab (prefetch; the values are kept alive only).
template <int NV, int PV, bool NT>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ w, const u32x4* __restrict__ wnext, int shift,
                                                const float* __restrict__ xin, float* __restrict__ xout) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float xv = xin[(wave * 64 + lane) & 1023];
  u32x4 v[NV];
  const u32x4* base = w + (size_t)wave * NV * 64 + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = NT ? __builtin_nontemporal_load(base + i * 64) : base[i * 64];
  u32x4 pv[PV > 0 ? PV : 1];
  if constexpr (PV > 0) {
    const int nb = gridDim.x;
    const int pb = (blockIdx.x + shift) % nb;  // shift = 0: same block -> slab map as the consumer
    const u32x4* pbase = wnext + ((size_t)(pb * 4 + (threadIdx.x >> 6))) * PV * 64 + lane;
#pragma unroll
    for (int i = 0; i < PV; ++i) pv[i] = pbase[i * 64];
  }
  __builtin_amdgcn_sched_barrier(0);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    acc += __uint_as_float(v[i].x << 16) * xv + __uint_as_float(v[i].y << 16) * xv + __uint_as_float(v[i].z << 16) * xv +
           __uint_as_float(v[i].w << 16) * xv;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) xout[wave & 1023] = acc * 1e-30f;
  if constexpr (PV > 0) {
#pragma unroll
    for (int i = 0; i < PV; ++i) asm volatile("" ::"v"(pv[i]));
  }
}

// same memory behaviour as k_stream<NV,0,true> but ~CODE x 8 bytes more straight-line code (cold I-cache cost?)
template <int NV, int CODE>
__global__ __launch_bounds__(256) void k_stream_big(const u32x4* __restrict__ w, const u32x4* __restrict__ wnext, int shift,
                                                    const float* __restrict__ xin, float* __restrict__ xout) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float xv = xin[(wave * 64 + lane) & 1023];
  u32x4 v[NV];
  const u32x4* base = w + (size_t)wave * NV * 64 + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(base + i * 64);
  __builtin_amdgcn_sched_barrier(0);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float f = __uint_as_float(v[i].x << 16) * xv + __uint_as_float(v[i].y << 16) + __uint_as_float(v[i].z << 16) + __uint_as_float(v[i].w << 16);
#pragma unroll
    for (int j = 0; j < CODE / NV; ++j) a[j & 7] = fmaf(f, 1.0f + 0.001f * (float)(i * 131 + j * 7 + 1), a[j & 7]);
  }
  float acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) xout[wave & 1023] = acc * 1e-30f;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
};

template <typename F>
static double time_graph(hipStream_t st, int nk, int reps, F enqueue) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nk; ++i) enqueue(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  Timer t;
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(t.a, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(t.b, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, t.a, t.b));
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return ms * 1e3 / (reps * nk);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t region = 512ull << 20;  // > MALL (256 MiB): a rotating walk is HBM-cold
  u32x4* W;
  CK(hipMalloc(&W, region + (16 << 20)));
  CK(hipMemset(W, 0x3c, region + (16 << 20)));
  float *xa, *xb;
  CK(hipMalloc(&xa, 1 << 20));
  CK(hipMalloc(&xb, 1 << 20));
  CK(hipMemset(xa, 0, 1 << 20));
  CK(hipMemset(xb, 0, 1 << 20));
  const int NK = 60, REPS = 50;

  printf("empty 256x256            : %.2f us/kernel\n", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, xa); }));
  printf("empty 1024x256           : %.2f us/kernel\n", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, xa); }));
  printf("empty 64x256             : %.2f us/kernel\n", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st, xa); }));
  printf("load->store 4x256        : %.2f us/kernel\n", time_graph(st, NK, REPS, [&](int i) { hipLaunchKernelGGL(k_loadstore, dim3(4), dim3(256), 0, st, (i & 1) ? xb : xa, (i & 1) ? xa : xb); }));
  printf("load->store 256x256      : %.2f us/kernel\n", time_graph(st, NK, REPS, [&](int i) { hipLaunchKernelGGL(k_loadstore, dim3(256), dim3(256), 0, st, (i & 1) ? xb : xa, (i & 1) ? xa : xb); }));

  // streaming chains: slab bytes = blocks * 4 waves * NV KiB
  auto run_stream = [&](const char* name, auto kern, int blocks, int nv, int shift, bool prefetch) {
    const size_t slab = (size_t)blocks * 4 * nv * 1024;
    const size_t nslab = region / slab;
    size_t cursor = 0;
    // the captured graph replays the same slabs every rep: nk * slab must exceed the MALL for a cold walk
    const int nk = (int)std::max<size_t>(NK, (448ull << 20) / slab);
    const double us = time_graph(st, nk, std::max(4, REPS * NK / nk), [&](int i) {
      const u32x4* w = W + (cursor % nslab) * (slab / 16);
      const u32x4* wn = W + ((cursor + 1) % nslab) * (slab / 16);
      ++cursor;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, w, prefetch ? wn : w, shift, (i & 1) ? xb : xa, (i & 1) ? xa : xb);
    });
    printf("%-34s: %.2f us/kernel  slab %.1f MB  (%d slabs per replay = %.0f MB)  %.0f GB/s\n", name, us, slab / 1048576.0, nk,
           nk * slab / 1048576.0, slab / us / 1e3);
  };
  run_stream("stream 2MB  nv2 256blk nt", k_stream<2, 0, true>, 256, 2, 0, false);
  run_stream("stream 6MB  nv6 256blk nt", k_stream<6, 0, true>, 256, 6, 0, false);
  run_stream("stream 8MB  nv8 256blk nt", k_stream<8, 0, true>, 256, 8, 0, false);
  run_stream("stream 8MB  nv8 256blk plain", k_stream<8, 0, false>, 256, 8, 0, false);
  run_stream("stream 8MB  nv4 512blk nt", k_stream<4, 0, true>, 512, 4, 0, false);
  run_stream("stream 16MB nv8 512blk nt", k_stream<8, 0, true>, 512, 8, 0, false);
  run_stream("stream 8MB nt + 512 extra FMAs", k_stream_big<8, 512>, 256, 8, 0, false);
  run_stream("stream 8MB nt + 1536 extra FMAs", k_stream_big<8, 1536>, 256, 8, 0, false);
  run_stream("stream 2MB nt + 512 extra FMAs", k_stream_big<2, 512>, 256, 2, 0, false);
  run_stream("stream 2MB nt + 1536 extra FMAs", k_stream_big<2, 1536>, 256, 2, 0, false);
  run_stream("stream 2MB + prefetch same-map", k_stream<2, 2, true>, 256, 2, 0, true);
  run_stream("stream 6MB + prefetch same-map", k_stream<6, 6, true>, 256, 6, 0, true);
  run_stream("stream 8MB + prefetch same-map", k_stream<8, 8, true>, 256, 8, 0, true);
  run_stream("stream 8MB + prefetch same plain", k_stream<8, 8, false>, 256, 8, 0, true);
  run_stream("stream 8MB + prefetch shift-1", k_stream<8, 8, true>, 256, 8, 1, true);
  run_stream("stream 8MB + prefetch shift-3", k_stream<8, 8, true>, 256, 8, 3, true);
  run_stream("stream 2MB + prefetch shift-1", k_stream<2, 2, true>, 256, 2, 1, true);
  return 0;
}
