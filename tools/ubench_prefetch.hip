// Microbenchmark: can a SIDE-STREAM prefetcher keep the batch-1 decode chain's weights in L2 ahead of it?
//
// tools/ubench_l2keep.hip shows that lines a kernel READS (default cache policy) are still in that XCD's L2 for the next kernel
// (8 MB body 3.1 -> 1.3 us).  The batch-1 AR step is 62 dependent launches that leave HBM idle ~80 % of the time (boundaries,
// latency floors).  Here: a stand-in for that chain (per layer five readers of 6.3 / 1.6 / 2.1 / 8.4 / 8.4 MB with the real grids:
// block b reads one contiguous slice and runs on XCD b % 8; 12 layers; every kernel consumes the previous kernel's output word;
// captured in a hipGraph, 8 steps per graph) runs on stream 0, while on stream 1 a small persistent kernel walks the same
// weights in consumption order with default-policy loads, each of its workgroups fetching the slices of the consumer blocks
// of ITS XCD, paced by a progress word the chain kernels publish (at most `lookahead` kernels ahead, skipping what is late).
// Every spin is bounded.  Output: us per step without / with the prefetcher.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_prefetch tools/ubench_prefetch.hip && tools/bin/ubench_prefetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Op {
  size_t off;   // byte offset of the op's weights
  int blocks;   // consumer grid
  int slice;    // bytes per consumer block (multiple of 4096)
};
constexpr int MAX_OPS = 64;
struct Chain {
  Op op[MAX_OPS];
  int n;
};

template <bool NT>
__global__ __launch_bounds__(256) void reader(const unsigned char* __restrict__ w, Op op, const unsigned* __restrict__ prev, unsigned* __restrict__ out,
                                              unsigned* __restrict__ progress, unsigned tick) {
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(progress, tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned dep = prev[threadIdx.x & 63];  // the previous kernel's output (a fresh line from another XCD, like x)
  const u32x4* p = reinterpret_cast<const u32x4*>(w + op.off + (size_t)blockIdx.x * op.slice);
  const int nv = op.slice / 16;
  u32x4 acc = {dep, 0u, 0u, 0u};
  for (int i = threadIdx.x; i < nv; i += 256) {
    const u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    acc ^= v;
  }
  unsigned x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  for (int o = 32; o; o >>= 1) x ^= __shfl_xor(x, o, 64);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 63] = x;  // a few words, like the 1024-float activations
}

// grid = 8 * per_xcd workgroups; workgroup g serves XCD g % 8 (round-robin dispatch), sub-slot g / 8
__global__ __launch_bounds__(256) void prefetcher(const unsigned char* __restrict__ w, Chain ch, int steps, int lookahead, int per_xcd,
                                                  const unsigned* __restrict__ progress, unsigned* __restrict__ sink, unsigned* __restrict__ stats) {
  const int xcd = blockIdx.x & 7, sub = blockIdx.x >> 3;
  u32x4 acc = {0u, 0u, 0u, 0u};
  unsigned skipped = 0, done = 0;
  for (int s = 0; s < steps; ++s)
    for (int k = 0; k < ch.n; ++k) {
      const unsigned tick = (unsigned)(s * ch.n + k) + 1;  // the consumer publishes `tick` when it starts
      // wait until the chain is within `lookahead` kernels of this op (bounded), skip the op if the chain already passed it
      unsigned cur = 0;
      for (int spin = 0; spin < 2000000; ++spin) {
        cur = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur + (unsigned)lookahead >= tick) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (cur + (unsigned)lookahead < tick) {  // timed out: the chain is not running -- leave
        if (threadIdx.x == 0) atomicAdd(&stats[2], 1u);
        return;
      }
      if (cur >= tick) {
        ++skipped;
        continue;
      }
      const Op op = ch.op[k];
      // consumer blocks of this XCD: b = xcd, xcd + 8, ...; this workgroup takes every per_xcd-th of them (J blocks), as ONE index
      // space of J * nv vectors with 16 loads per thread in flight (64 KB per workgroup)
      const int nv = op.slice / 16;
      const int nb_xcd = (op.blocks - xcd + 7) / 8;                   // consumer blocks on this XCD
      const int J = nb_xcd > sub ? (nb_xcd - sub + per_xcd - 1) / per_xcd : 0;
      const int M = J * nv;
      const unsigned char* base = w + op.off;
      for (int t0 = threadIdx.x; t0 < M; t0 += 256 * 16) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int idx = t0 + u * 256;
          const int j = idx / nv, i = idx - j * nv;
          const int b = xcd + 8 * (sub + per_xcd * j);
          v[u] = idx < M ? *reinterpret_cast<const u32x4*>(base + (size_t)b * op.slice + (size_t)i * 16) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u];
      }
      ++done;
    }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], done);
    atomicAdd(&stats[1], skipped);
  }
}

int main() {
  // the chain: 12 layers x {qkv 384 x 16 KB, kv 256 x 6 KB, out-proj 256 x 8 KB, ffn1 256 x 32 KB, ffn2 256 x 32 KB} + logits 258 x 8 KB
  Chain ch;
  ch.n = 0;
  size_t off = 0;
  auto add = [&](int blocks, int slice) {
    ch.op[ch.n++] = Op{off, blocks, slice};
    off += (size_t)blocks * slice;
  };
  for (int l = 0; l < 12; ++l) {
    add(384, 16384);
    add(256, 6144);
    add(256, 8192);
    add(256, 32768);
    add(256, 32768);
  }
  add(258, 8192);
  const size_t total = off;
  unsigned char* W;
  unsigned *outA, *outB, *progress, *sink, *stats;
  CK(hipMalloc(&W, total));
  CK(hipMemset(W, 1, total));
  CK(hipMalloc(&outA, 256));
  CK(hipMalloc(&outB, 256));
  CK(hipMemset(outA, 0, 256));
  CK(hipMemset(outB, 0, 256));
  CK(hipMalloc(&progress, 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&stats, 16));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int SPG = 8, GRAPHS = 6;  // steps per graph, graph launches per measurement
  printf("{\"weights_MB\": %.1f, \"kernels_per_step\": %d", total / 1048576.0, ch.n);
  for (int nt = 0; nt < 2; ++nt) {
    // one graph per (graph index): ticks are absolute, so capture GRAPHS graphs of SPG steps each
    std::vector<hipGraphExec_t> execs;
    for (int gi = 0; gi < GRAPHS; ++gi) {
      hipGraph_t g;
      CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < SPG; ++s)
        for (int k = 0; k < ch.n; ++k) {
          const unsigned tick = (unsigned)((gi * SPG + s) * ch.n + k) + 1;
          unsigned* prev = (k & 1) ? outA : outB;
          unsigned* out = (k & 1) ? outB : outA;
          if (nt) hipLaunchKernelGGL((reader<true>), dim3(ch.op[k].blocks), dim3(256), 0, s0, W, ch.op[k], prev, out, progress, tick);
          else hipLaunchKernelGGL((reader<false>), dim3(ch.op[k].blocks), dim3(256), 0, s0, W, ch.op[k], prev, out, progress, tick);
        }
      CK(hipStreamEndCapture(s0, &g));
      hipGraphExec_t e;
      CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
      execs.push_back(e);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Cfg {
      int per_xcd, lookahead;
    };
    const Cfg cfgs[] = {{0, 0}, {8, 4}, {16, 1}, {16, 2}, {16, 3}, {32, 1}, {32, 2}, {32, 3}};
    for (const Cfg& c : cfgs) {
      float best = 1e9f;
      unsigned hst[4] = {0, 0, 0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(progress, 0, 4));
        CK(hipMemset(stats, 0, 16));
        CK(hipDeviceSynchronize());
        if (c.per_xcd)
          hipLaunchKernelGGL(prefetcher, dim3(8 * c.per_xcd), dim3(256), 0, s1, W, ch, SPG * GRAPHS, c.lookahead, c.per_xcd, progress, sink, stats);
        CK(hipEventRecord(e0, s0));
        for (int gi = 0; gi < GRAPHS; ++gi) CK(hipGraphLaunch(execs[gi], s0));
        CK(hipEventRecord(e1, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
          best = ms;
          CK(hipMemcpy(hst, stats, 16, hipMemcpyDeviceToHost));
        }
      }
      printf(",\n \"%s_pf%d_la%d\": {\"us_per_step\": %.1f, \"ops_prefetched_per_wg\": %.1f, \"ops_skipped_per_wg\": %.1f, \"timeouts\": %u}", nt ? "nt" : "default",
             c.per_xcd, c.lookahead, best * 1000.f / (SPG * GRAPHS), c.per_xcd ? hst[0] / (8.0 * c.per_xcd) : 0.0,
             c.per_xcd ? hst[1] / (8.0 * c.per_xcd) : 0.0, hst[2]);
      fflush(stdout);
    }
    for (auto e : execs) CK(hipGraphExecDestroy(e));
  }
  printf("\n}\n");
  return 0;
}
