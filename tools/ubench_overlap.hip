// Feasibility of OVERLAPPED dependent launches for the batch-1 AR step (round 3).
//
// Today the step is a chain of 50 dependent launches; each pays the launch boundary (1.2 us) + ramp + one HBM round trip for
// its weights before it can do anything (tools/ubench_boundary.hip: 3.3 us per 8 MB kernel, 2.3 us per 2 MB kernel).  The
// round-3 edge measurement (tools/ubench_edges.hip) says a 1024-float vector reaches every CU 1.5 us after it was published.
// So: launch kernel N+1 BEFORE kernel N has finished -- two streams, kernels alternating between them, captured as two parallel
// chains of one hipGraph -- let it request its weights at once, and make the true dependency explicit in the data: kernel N
// publishes its output as 8-byte {epoch, value} granules, kernel N+1 polls them (after its weight burst has landed, so that its
// memory queue is quiet).  Stream order still serialises N-1 -> N+1, so at most two kernels are alive and every workgroup is
// resident: no deadlock; every spin is bounded anyway.
//
// This program measures the period of such a chain against the plain dependent chain, for 8 MB / 2 MB weight kernels.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_overlap tools/ubench_overlap.hip && tools/bin/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int NG = 1024;          // granules per vector (the residual row)
constexpr unsigned SPIN_LIMIT = 20000u;

// plain chain kernel: x from the previous kernel's fp32 output
template <int NV>
__global__ __launch_bounds__(256) void k_plain(const u32x4* __restrict__ w, const float* __restrict__ xin, float* __restrict__ xout) {
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float xv = xin[(wave * 64 + lane) & (NG - 1)];
  u32x4 v[NV];
  const u32x4* base = w + (size_t)wave * NV * 64 + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(base + i * 64);
  __builtin_amdgcn_sched_barrier(0);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) acc += __uint_as_float(v[i].x << 16) * xv + __uint_as_float(v[i].y << 16) * xv;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) xout[wave & (NG - 1)] = acc * 1e-30f;
}

// overlapped chain kernel: weights first, then poll the predecessor's granules, then compute, then publish
template <int NV>
__global__ __launch_bounds__(256) void k_ovl(const u32x4* __restrict__ w, const u64* __restrict__ gin, u64* __restrict__ gout,
                                             const unsigned* __restrict__ replay, int node, int nodes, unsigned* __restrict__ fail) {
  __shared__ float sx[NG];
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wave = blockIdx.x * 4 + wv;
  u32x4 v[NV];
  const u32x4* base = w + (size_t)wave * NV * 64 + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load(base + i * 64);
  const unsigned rep = replay[0];
  const unsigned epoch_in = rep * (unsigned)nodes + (unsigned)node;       // the predecessor's tag (node 0: tag of the last node of the previous replay)
  const unsigned epoch_out = epoch_in + 1u;
  __builtin_amdgcn_sched_barrier(0);
  // wait for the weight burst first: polling with loads in flight queues behind them anyway (vmcnt is in order)
  if (wv == 0) {
    bool ok = false;
    u64 gx[16];
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
#pragma unroll
      for (int k = 0; k < 16; ++k) gx[k] = __hip_atomic_load(gin + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool mine = true;
#pragma unroll
      for (int k = 0; k < 16; ++k) mine &= (unsigned)(gx[k] >> 32) == epoch_in;
      if (__all(mine)) { ok = true; break; }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) sx[k * 64 + lane] = __uint_as_float((unsigned)gx[k]);
    if (lane == 0) { s_ok = ok; if (!ok) atomicAdd(fail, 1u); }
  }
  __syncthreads();
  const float xv = sx[(wave * 64 + lane) & (NG - 1)];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) acc += __uint_as_float(v[i].x << 16) * xv + __uint_as_float(v[i].y << 16) * xv;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  // publish: this workgroup's 4 values (one per wave) as granules
  if (lane == 0) {
    const float out = acc * 1e-30f + 1.0f;
    __hip_atomic_store(gout + (wave & (NG - 1)), ((u64)epoch_out << 32) | (u64)__float_as_uint(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  (void)s_ok;
}

__global__ void k_bump(unsigned* replay) { if (threadIdx.x == 0 && blockIdx.x == 0) replay[0] += 1u; }

template <typename F>
static double time_graph_exec(hipStream_t st, hipGraphExec_t ge, int nk, int reps) {
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
  return ms * 1e3 / (reps * nk);
}

template <int NV>
static void run_case(const char* name, hipStream_t sa, hipStream_t sb, const u32x4* W, size_t region) {
  const int NKER = 60, REPS = 40, blocks = 256;
  const size_t slab = (size_t)blocks * 4 * NV * 1024, nslab = region / slab;
  float *xa, *xb;
  CK(hipMalloc(&xa, NG * 4)); CK(hipMalloc(&xb, NG * 4));
  CK(hipMemset(xa, 0, NG * 4)); CK(hipMemset(xb, 0, NG * 4));
  // ---- plain dependent chain (one stream)
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NKER; ++i)
    hipLaunchKernelGGL((k_plain<NV>), dim3(blocks), dim3(256), 0, sa, W + (i % nslab) * (slab / 16), (i & 1) ? xb : xa, (i & 1) ? xa : xb);
  CK(hipStreamEndCapture(sa, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  const double plain = time_graph_exec<int>(sa, ge, NKER, REPS);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  // ---- overlapped: two streams, alternating kernels, granule dependencies
  u64* gran;  // [NKER + 1][NG] one granule vector per node (node i reads vector i, writes vector i + 1; vector 0 = vector NKER of the previous replay)
  unsigned *replay, *fail;
  CK(hipMalloc(&gran, sizeof(u64) * NG * NKER));
  CK(hipMalloc(&replay, 4)); CK(hipMalloc(&fail, 4));
  CK(hipMemset(replay, 0, 4)); CK(hipMemset(fail, 0, 4));
  // seed: node 0 of replay 0 expects tag 0 on vector 0's slot, which is vector NKER-1's storage after a wrap: zeros carry tag 0
  CK(hipMemset(gran, 0, sizeof(u64) * NG * NKER));
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  for (int two = 1; two >= 0; --two) {  // two = 1: two streams; 0: the same kernels on ONE stream (what the granule protocol alone costs)
    CK(hipMemset(replay, 0, 4));
    CK(hipMemset(gran, 0, sizeof(u64) * NG * NKER));
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    if (two) { CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0)); }
    for (int i = 0; i < NKER; ++i) {
      hipStream_t s = (two && (i & 1)) ? sb : sa;
      const u64* gin = gran + (size_t)((i + NKER - 1) % NKER) * NG;  // predecessor's vector (node 0: the last node's, previous replay)
      u64* gout = gran + (size_t)i * NG;
      hipLaunchKernelGGL((k_ovl<NV>), dim3(blocks), dim3(256), 0, s, W + (i % nslab) * (slab / 16), gin, gout, replay, i, NKER, fail);
    }
    if (two) { CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0)); }
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, sa, replay);
    CK(hipStreamEndCapture(sa, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double t = time_graph_exec<int>(sa, ge, NKER, REPS);
    unsigned nf = 0;
    CK(hipMemcpy(&nf, fail, 4, hipMemcpyDeviceToHost));
    printf(" \"%s_%s\": {\"us_per_kernel\": %.3f, \"plain_chain_us_per_kernel\": %.3f, \"spin_timeouts\": %u},\n", name, two ? "two_streams" : "one_stream", t, plain, nf);
    fflush(stdout);
    CK(hipMemset(fail, 0, 4));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  CK(hipFree(gran)); CK(hipFree(replay)); CK(hipFree(fail)); CK(hipFree(xa)); CK(hipFree(xb));
}

int main() {
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const size_t region = 512ull << 20;
  u32x4* W;
  CK(hipMalloc(&W, region + (16 << 20)));
  CK(hipMemset(W, 0x3c, region + (16 << 20)));
  printf("{\n");
  run_case<8>("stream8MB", sa, sb, W, region);
  run_case<2>("stream2MB", sa, sb, W, region);
  printf(" \"note\": \"period per kernel of a 60-kernel dependent chain in a hipGraph, 256 workgroups x 256 threads\"\n}\n");
  return 0;
}
