// Bisect of the dependent-kernel boundary of the batch-1 AR step (VERDICT r2 "next" 3): the in-kernel timeline prices a
// boundary between the step's GEMV kernels at 1.6-2.1 us where /opt/skills/guides/MI355X_MICROARCH.md quotes ~1.2 us for the
// same kind of chain.  Every row below is a hipGraph of 60 dependent launches, replayed 50 times; us per kernel.
//   A  trivial kernels: grid size (256 / 384 / 512 / 1024 workgroups), block size
//   B  kernel arguments: one pointer vs a 256-byte struct by value (SkinnyArgs is ~230 bytes) vs a pointer to a device-resident
//      argument block; and, measured INSIDE the kernel, the time from wave start to the arrival of the first kernarg dword
//      (s_memrealtime before / after a volatile load of the kernarg segment) -- the kernels' own ktrace stamp sits AFTER that
//      load, so this latency is booked as "gap" in profiles/r02_ktrace_b1_*
//   C  what the previous kernel leaves behind: no store / 4 B per workgroup / 4 KB per workgroup / 64 KB per workgroup (dirty L2
//      lines to write back at the boundary)
//   D  real streaming kernels (8 MB of weights, HBM-cold walk) with 256 / 384 workgroups, pointer args vs struct args
//   E  static LDS / scratch-free variants: 1 KB static LDS in the kernel (wave launch waits for the LDS allocation)
// Run under different runtime settings by the caller (tools/gpu_r3_boundary.sh): HIP_FORCE_DEV_KERNARG=0/1,
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0/1.  The kernarg-preload build (-mllvm -amdgpu-kernarg-preload-count=16) is a second binary.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_boundary.hip -o gpurun_out/ubench_boundary
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct BigArgs {  // 256 bytes
  const float* in;
  float* out;
  long long pad[29];
  int n;
  int last;
};
static_assert(sizeof(BigArgs) == 256, "BigArgs");

__global__ void k_triv(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_triv_big(BigArgs a) { if (a.last == 12345) a.out[0] = 1.f; }
__global__ void k_triv_ind(const BigArgs* a) { if (a->last == 12345) a->out[0] = 1.f; }
__global__ void k_triv_lds(float* p) {
  __shared__ float s[256];
  s[threadIdx.x] = 1.f;
  __syncthreads();
  if (p == nullptr) p[0] = s[(threadIdx.x + 1) & 255];
}
template <int BYTES>
__global__ void k_store(float* p) {  // every workgroup leaves BYTES dirty bytes
  if constexpr (BYTES >= 1024) {
    for (int i = threadIdx.x; i < BYTES / 4; i += blockDim.x) p[(size_t)blockIdx.x * (BYTES / 4) + i] = 1.f;
  } else if (BYTES > 0) {
    if (threadIdx.x == 0) p[blockIdx.x] = 1.f;
  } else if (p == nullptr) p[0] = 1.f;
}

__device__ inline unsigned long long rt() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// wave start -> first kernarg dword -> first dependent global dword, per workgroup, in 10 ns ticks
__global__ void k_probe(unsigned long long* out, const float* src) {
  const unsigned long long t0 = rt();
  const volatile unsigned long long* kp = (const volatile unsigned long long*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned long long* o = (unsigned long long*)kp[0];
  const unsigned long long t1 = rt();
  const float* s = (const float*)kp[1];
  const float v = __builtin_nontemporal_load(s + blockIdx.x * 64 + threadIdx.x % 64);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = rt();
  if (threadIdx.x == 0) {
    o[blockIdx.x * 4 + 0] = t0;
    o[blockIdx.x * 4 + 1] = t1 - t0;
    o[blockIdx.x * 4 + 2] = t2 - t1;
    o[blockIdx.x * 4 + 3] = (unsigned long long)(v != 12345.f);
  }
  (void)out; (void)src;
}

// streaming GEMV stand-in: wave reads NV 16-byte vectors per lane (non-temporal), reduces, lane 0 stores
template <int NV, bool STRUCT>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ w, const float* __restrict__ xin, float* __restrict__ xout, BigArgs big) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if constexpr (STRUCT) { xin = big.in; xout = big.out; }
  const float xv = xin[(wave * 64 + lane) & 1023];
  u32x4 v[NV];
  const u32x4* base = w + (size_t)wave * NV * 64 + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(base + i * 64);
  __builtin_amdgcn_sched_barrier(0);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    acc += __uint_as_float(v[i].x << 16) * xv + __uint_as_float(v[i].y << 16) * xv + __uint_as_float(v[i].z << 16) * xv + __uint_as_float(v[i].w << 16) * xv;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) xout[wave & 1023] = acc * 1e-30f;
}

template <typename F>
static double time_graph(hipStream_t st, int nk, int reps, F enqueue) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nk; ++i) enqueue(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms * 1e3 / (reps * nk);
}

int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "default";
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t region = 512ull << 20;
  u32x4* W;
  CK(hipMalloc(&W, region + (16 << 20)));
  CK(hipMemset(W, 0x3c, region + (16 << 20)));
  float *xa, *xb, *dirty;
  CK(hipMalloc(&xa, 1 << 20)); CK(hipMalloc(&xb, 1 << 20)); CK(hipMalloc(&dirty, 64 << 20));
  CK(hipMemset(xa, 0, 1 << 20)); CK(hipMemset(xb, 0, 1 << 20));
  BigArgs big{};
  big.in = xa; big.out = xb; big.n = 1; big.last = 7;
  BigArgs* big_dev;
  CK(hipMalloc(&big_dev, sizeof(BigArgs)));
  CK(hipMemcpy(big_dev, &big, sizeof(BigArgs), hipMemcpyHostToDevice));
  const int NK = 60, REPS = 50;
  printf("{\"tag\": \"%s\"", tag);
  auto row = [&](const char* name, double us) { printf(",\n \"%s\": %.3f", name, us); fflush(stdout); };

  // A: trivial, grid size / block size
  for (int blocks : {64, 256, 384, 512, 1024}) {
    char nm[64]; snprintf(nm, sizeof nm, "A_triv_ptr_%dx256", blocks);
    row(nm, time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_triv, dim3(blocks), dim3(256), 0, st, xa); }));
  }
  row("A_triv_ptr_256x64", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_triv, dim3(256), dim3(64), 0, st, xa); }));
  row("A_triv_ptr_256x1024", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_triv, dim3(256), dim3(1024), 0, st, xa); }));
  // B: argument passing
  row("B_triv_struct256_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_triv_big, dim3(256), dim3(256), 0, st, big); }));
  row("B_triv_indirect_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_triv_ind, dim3(256), dim3(256), 0, st, (const BigArgs*)big_dev); }));
  row("E_triv_lds1k_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_triv_lds, dim3(256), dim3(256), 0, st, xa); }));
  // C: dirty bytes left by every kernel
  row("C_store0_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_store<0>, dim3(256), dim3(256), 0, st, dirty); }));
  row("C_store4B_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_store<4>, dim3(256), dim3(256), 0, st, dirty); }));
  row("C_store4KB_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_store<4096>, dim3(256), dim3(256), 0, st, dirty); }));
  row("C_store64KB_256x256", time_graph(st, NK, REPS, [&](int) { hipLaunchKernelGGL(k_store<65536>, dim3(256), dim3(256), 0, st, dirty); }));

  // B2: kernarg arrival measured inside the kernel (chain of probes, one graph)
  {
    unsigned long long* pr;
    const int blocks = 256;
    CK(hipMalloc(&pr, (size_t)NK * blocks * 4 * 8));
    CK(hipMemset(pr, 0, (size_t)NK * blocks * 4 * 8));
    const double us = time_graph(st, NK, REPS, [&](int i) { hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, st, pr + (size_t)i * blocks * 4, (const float*)W); });
    row("B2_probe_chain_us_per_kernel", us);
    std::vector<unsigned long long> h((size_t)NK * blocks * 4);
    CK(hipMemcpy(h.data(), pr, h.size() * 8, hipMemcpyDeviceToHost));
    double karg = 0, kmax = 0, first = 0, glob = 0, gapsum = 0;
    for (int i = 0; i < NK; ++i) {
      unsigned long long t_first = ~0ull, t_last = 0;
      for (int b = 0; b < blocks; ++b) {
        const unsigned long long* q = &h[((size_t)i * blocks + b) * 4];
        karg += (double)q[1]; kmax = std::max(kmax, (double)q[1]); glob += (double)q[2];
        t_first = std::min(t_first, q[0]); t_last = std::max(t_last, q[0] + q[1] + q[2]);
      }
      // kernarg latency of the FIRST wave of each kernel (the one on the chain's critical path)
      for (int b = 0; b < blocks; ++b) {
        const unsigned long long* q = &h[((size_t)i * blocks + b) * 4];
        if (q[0] == t_first) { first += (double)q[1]; break; }
      }
      if (i > 0) {
        unsigned long long prev_last = 0;
        for (int b = 0; b < blocks; ++b) {
          const unsigned long long* q = &h[((size_t)(i - 1) * blocks + b) * 4];
          prev_last = std::max(prev_last, q[0] + q[1] + q[2]);
        }
        gapsum += (double)t_first - (double)prev_last;
      }
      (void)t_last;
    }
    row("B2_wave_start_to_kernarg_us_mean", karg / (NK * blocks) * 0.01);
    row("B2_wave_start_to_kernarg_us_max", kmax * 0.01);
    row("B2_wave_start_to_kernarg_us_first_wave", first / NK * 0.01);
    row("B2_kernarg_to_global_dword_us_mean", glob / (NK * blocks) * 0.01);
    row("B2_last_end_to_next_first_wave_start_us", gapsum / (NK - 1) * 0.01);
  }

  // D: streaming chains (HBM-cold walk over 512 MB)
  auto run_stream = [&](const char* name, auto kern, int blocks, int nv) {
    const size_t slab = (size_t)blocks * 4 * nv * 1024;
    const size_t nslab = region / slab;
    size_t cursor = 0;
    const int nk = (int)std::max<size_t>(NK, (448ull << 20) / slab);
    const double us = time_graph(st, nk, std::max(4, REPS * NK / nk), [&](int i) {
      const u32x4* w = W + (cursor % nslab) * (slab / 16);
      ++cursor;
      BigArgs b2 = big;
      b2.in = (i & 1) ? xb : xa; b2.out = (i & 1) ? xa : xb;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, w, (const float*)b2.in, b2.out, b2);
    });
    row(name, us);
  };
  run_stream("D_stream8MB_256blk_ptrargs", k_stream<8, false>, 256, 8);
  run_stream("D_stream8MB_256blk_structargs", k_stream<8, true>, 256, 8);
  run_stream("D_stream6MB_256blk_ptrargs", k_stream<6, false>, 256, 6);
  run_stream("D_stream6MB_384blk_ptrargs", k_stream<4, false>, 384, 4);
  run_stream("D_stream2MB_256blk_ptrargs", k_stream<2, false>, 256, 2);
  run_stream("D_stream2MB_256blk_structargs", k_stream<2, true>, 256, 2);
  printf("}\n");
  return 0;
}
