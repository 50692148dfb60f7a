// Microbenchmark: what would the all-to-all hand-offs of a PERSISTENT batch-1 decode step cost on this chip?
//
// The batch-1 AR step is a chain of 62 dependent kernels: the in-kernel timeline (tools/ktrace_step.py)
// attributes ~1.5 us of every link to the launch boundary and ~2-3.5 us to the body (one HBM round trip under load).  A
// persistent kernel replaces each boundary by an in-launch all-gather of the op's output vector (every CU needs all of
// x / attn / h before its next GEMV slice).  This program measures exactly that edge, with the guide's R2 recipe
// (/opt/skills/guides/cdna_hip_programming.md Guideline 16: 8-byte {epoch, value} granules, sc1 stores, one wave per
// workgroup sweeping with relaxed agent loads), one workgroup per CU, and optionally a weight stream running beside it:
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_edges tools/ubench_edges.hip && tools/bin/ubench_edges
//
// Output: us per edge for vectors of 1024 fp32 (x, attention output) and 4096 bf16 (FFN hidden), idle and with every CU
// streaming `--stream-kb` KB of HBM per edge (non-temporal loads, as the decode GEMVs do).  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 256;            // threads per workgroup (4 waves): wave 0 gathers, waves 1-3 stream
constexpr u64 SPIN_LIMIT = 4000000ull;  // polls before giving up (seconds): a hang would cost a GPU strike

struct Params {
  u64* gran;        // [2][gmax] granules, double-buffered by epoch parity
  int gmax;         // granules per buffer
  int per_wg;       // granules each workgroup publishes per edge (gmax = per_wg * gridDim.x)
  int edges;        // edges per launch
  int producers;    // workgroups that publish (the others only gather): x of a decode layer is produced by ~32 CUs
  int lean;         // round 3: the retry path touches no global word (the round-2 loop polled p.err after EVERY failed pass -- a second
                    // dependent fabric round trip per retry -- and slept); the timeout is a local spin count only
  const u32x4* stream;  // weight stand-in
  size_t stream_vecs;   // 16-byte vectors available
  int stream_vecs_per_edge;  // per workgroup per edge (0 = idle chip)
  unsigned* err;    // != 0: a spin timed out
  u64* t_out;       // [gridDim.x][2] start / end wall clock of each workgroup
  float* sink;
};

__global__ __launch_bounds__(T) void edges_kernel(Params p) {
  __shared__ float sh_sum;
  __shared__ int sh_fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x, nwg = gridDim.x;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  const u64 t0 = wall_clock64();
  float carry = (float)wg;  // value chain: every edge's payload depends on the previous gather
  float junk = 0.f;
  size_t sv = ((size_t)wg * 977) % (p.stream_vecs ? p.stream_vecs : 1);
  for (int e = 0; e < p.edges; ++e) {
    const unsigned epoch = (unsigned)e + 1;
    u64* buf = p.gran + (size_t)(e & 1) * p.gmax;
    if (wave == 0) {
      // ---- publish this workgroup's granules (write-through 8-byte stores: the data is the flag) ----
      if (wg < p.producers)
        for (int g = lane; g < p.per_wg; g += 64) {
          const float v = carry + (float)g;
          __hip_atomic_store(buf + (size_t)wg * p.per_wg + g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      // ---- sweep ALL granules until every tag carries this epoch ----
      float s = 0.f;
      bool fail = false;
      for (int base = 0; base < p.gmax; base += 64 * 16) {  // 16 granules per lane per pass
        u64 spins = 0;
        while (true) {
          bool ok = true;
          float part = 0.f;
          if (p.lean) {
            // round 3: ALL 16 loads of the pass issued back to back, ONE wait (the guide's pass).  The round-2 loop below guards
            // every load with `idx < gmax`: hipcc turns that into 16 exec-masked branches with `s_waitcnt vmcnt(0)` after each
            // load -- 16 SERIAL fabric round trips per pass, which is what round 2 measured as "7.1 us per edge".
            u64 x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              const int idx = min(base + k * 64 + lane, p.gmax - 1);
              x[k] = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              const bool valid = base + k * 64 + lane < p.gmax;
              ok &= !valid || (unsigned)(x[k] >> 32) == epoch;
              part += valid ? __uint_as_float((unsigned)x[k]) : 0.f;
            }
          } else {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int idx = base + k * 64 + lane;
            if (idx < p.gmax) {
              const u64 x = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok &= (unsigned)(x >> 32) == epoch;
              part += __uint_as_float((unsigned)x);
            }
          }
          }
          if (__all(ok)) {
            s += part;
            break;
          }
          if (p.lean) {
            if (++spins > SPIN_LIMIT) {
              fail = true;
              break;
            }
          } else {
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
              fail = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (fail) break;
      }
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) {
        sh_sum = s;
        if (fail) {
          sh_fail = 1;
          atomicExch(p.err, 1u + (unsigned)e);
        }
      }
    } else if (p.stream_vecs_per_edge > 0) {
      // ---- the weight stream of the next op, running beside the hand-off (non-temporal, nothing depends on it) ----
      const int per_wave = p.stream_vecs_per_edge / 3;
      u32x4 acc = {0u, 0u, 0u, 0u};
      for (int i = lane; i < per_wave; i += 64) {
        const u32x4 v = __builtin_nontemporal_load(p.stream + (sv + (size_t)(wave - 1) * per_wave + i) % p.stream_vecs);
        acc ^= v;
      }
      junk += (float)(acc.x ^ acc.y ^ acc.z ^ acc.w) * 1e-30f;
      sv = (sv + (size_t)nwg * p.stream_vecs_per_edge) % p.stream_vecs;
    }
    __syncthreads();  // the gathered vector (here: its sum) is handed to the other waves through LDS
    if (sh_fail) break;
    carry = sh_sum * 1e-6f + (float)wg;
    __syncthreads();
  }
  const u64 t1 = wall_clock64();
  if (tid == 0) {
    p.t_out[2 * wg] = t0;
    p.t_out[2 * wg + 1] = t1;
  }
  if (junk == 12345.f) p.sink[wg] = junk + carry;
}

static double run(int nwg, int per_wg, int edges, int stream_kb, const u32x4* stream, size_t stream_vecs, int reps, int producers = 0,
                  int lean = 0) {
  Params p{};
  p.per_wg = per_wg;
  p.producers = producers > 0 ? producers : nwg;
  p.lean = lean;
  p.gmax = per_wg * p.producers;
  p.edges = edges;
  p.stream = stream;
  p.stream_vecs = stream_vecs;
  p.stream_vecs_per_edge = stream_kb * 1024 / 16;
  CK(hipMalloc(&p.gran, sizeof(u64) * 2 * p.gmax));
  CK(hipMalloc(&p.err, 4));
  CK(hipMalloc(&p.t_out, sizeof(u64) * 2 * nwg));
  CK(hipMalloc(&p.sink, 4 * nwg));
  std::vector<u64> t(2 * nwg);
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemset(p.gran, 0, sizeof(u64) * 2 * p.gmax));  // re-initialise every call (Guideline 16)
    CK(hipMemset(p.err, 0, 4));
    hipLaunchKernelGGL(edges_kernel, dim3(nwg), dim3(T), 0, 0, p);
    CK(hipDeviceSynchronize());
    unsigned err = 0;
    CK(hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost));
    if (err) {
      fprintf(stderr, "  spin timed out at edge %u (nwg %d): workgroups not co-resident?\n", err - 1, nwg);
      best = -1;
      break;
    }
    CK(hipMemcpy(t.data(), p.t_out, sizeof(u64) * 2 * nwg, hipMemcpyDeviceToHost));
    u64 lo = ~0ull, hi = 0;
    for (int i = 0; i < nwg; ++i) {
      lo = std::min(lo, t[2 * i]);
      hi = std::max(hi, t[2 * i + 1]);
    }
    best = std::min(best, (double)(hi - lo) * 0.01 / edges);  // 100 MHz clock -> us per edge
  }
  CK(hipFree(p.gran));
  CK(hipFree(p.err));
  CK(hipFree(p.t_out));
  CK(hipFree(p.sink));
  return best;
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d,\n", prop.name, ncu);
  const size_t stream_bytes = (size_t)512 << 20;  // > the 256 MB memory-side cache
  u32x4* stream = nullptr;
  CK(hipMalloc(&stream, stream_bytes));
  CK(hipMemset(stream, 1, stream_bytes));
  const size_t sv = stream_bytes / 16;
  const int edges = 240, reps = 5;
  printf(" \"edges_per_launch\": %d, \"unit\": \"us per all-gather edge, one workgroup per CU, best of %d\",\n \"results\": [\n", edges, reps);
  struct Case {
    const char* name;
    int per_wg;
    int stream_kb;
  };
  // x / attention vectors: 1024 fp32 = 4 granules per CU; FFN hidden: 4096 bf16 = 2048 granules = 8 per CU
  const Case cases[] = {{"x_1024f32_idle", 4, 0},      {"h_4096bf16_idle", 8, 0},      {"h_4096f32_idle", 16, 0},
                        {"x_1024f32_stream24KB", 4, 24}, {"h_4096bf16_stream24KB", 8, 24}, {"x_1024f32_stream96KB", 4, 96}};
  const int ncase = (int)(sizeof(cases) / sizeof(cases[0]));
  for (int i = 0; i < ncase; ++i) {
    const double us = run(ncu, cases[i].per_wg, edges, cases[i].stream_kb, stream, sv, reps);
    printf("  {\"case\": \"%s\", \"granules\": %d, \"stream_kb_per_cu_per_edge\": %d, \"us_per_edge\": %.3f}%s\n", cases[i].name,
           cases[i].per_wg * ncu, cases[i].stream_kb, us, i + 1 < ncase ? "," : "");
    fflush(stdout);
  }
  // fewer gatherers: 64 workgroups (one per 4 CUs) -- the attention merge edge, or a two-level scheme
  printf(" ],\n \"x_1024f32_idle_64wg\": %.3f,\n", run(64, 16, edges, 0, stream, sv, reps));
  // round 3: the same edges with a retry path that touches no global word, and with the vector published by
  // the CUs that actually produce it (32 for the 1024-float x of a decode layer: 32 granules each)
  printf(" \"round3_lean\": {\"x_1024f32_256producers\": %.3f, ", run(ncu, 4, edges, 0, stream, sv, reps, 0, 1));
  printf("\"x_1024f32_32producers\": %.3f, ", run(ncu, 32, edges, 0, stream, sv, reps, 32, 1));
  printf("\"x_1024f32_32producers_stream24KB\": %.3f, ", run(ncu, 32, edges, 24, stream, sv, reps, 32, 1));
  printf("\"h_4096f32_256producers\": %.3f, ", run(ncu, 16, edges, 0, stream, sv, reps, 0, 1));
  printf("\"partials_4224f32_64producers\": %.3f, ", run(ncu, 66, edges, 0, stream, sv, reps, 64, 1));
  printf("\"x_1024f32_32producers_64gatherers\": %.3f}\n}\n", run(64, 32, edges, 0, stream, sv, reps, 32, 1));
  CK(hipFree(stream));
  return 0;
}
