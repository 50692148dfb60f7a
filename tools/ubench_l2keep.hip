// Microbenchmark: does what kernel k reads stay in the XCD's L2 for kernel k+1?
//
// The batch-1 AR step is a chain of 62 dependent weight-streaming launches whose bodies are ~2.0 us of floor + 0.5-0.9 us for
// the 2-8 MB of weights.  If the lines a kernel READS survived the launch boundary in the L2 of the XCD that
// read them, every launch could pull the NEXT launch's weights into L2 with a few spare waves, and the next body would start
// on L2 hits instead of HBM misses.  This program measures exactly that: kernel A reads a buffer, kernel B (next in the
// stream) reads W with block b -> slice b, and B's body (first wave in to last wave out, wall_clock64) is compared for
//   cold     A read another buffer, caches flushed by a 1 GB sweep before A   (W comes from HBM)
//   same     A read W with the same block -> slice map                        (same XCD's L2, if it survives the boundary)
//   shifted  A read W with slices shifted by one block                        (another XCD's L2; MALL / HBM for B)
// at 2 / 8 MB (out-proj / FFN-sized), 512 blocks of 256 threads, non-temporal and default loads.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_l2keep tools/ubench_l2keep.hip && tools/bin/ubench_l2keep
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
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

template <bool NT>
__global__ __launch_bounds__(256) void reader(const u32x4* __restrict__ w, int vecs_per_block, int shift, unsigned* __restrict__ sink,
                                              u64* __restrict__ stamps) {
  const u64 t0 = wall_clock64();
  const int nb = gridDim.x;
  const int slice = ((int)blockIdx.x + shift) % nb;
  const u32x4* p = w + (size_t)slice * vecs_per_block;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int i = threadIdx.x; i < vecs_per_block; i += 256) {
    u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    acc ^= v;
  }
  const unsigned x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (x == 0x12345u) sink[blockIdx.x] = x;  // keeps the loads alive
  const u64 t1 = wall_clock64();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

__global__ void sweep(const u32x4* __restrict__ big, size_t vecs, unsigned* sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (size_t)gridDim.x * blockDim.x) acc ^= big[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

int main() {
  const int NB = 512;
  const size_t BIG = (size_t)1 << 30;
  u32x4 *W, *W2, *big;
  unsigned* sink;
  u64* stamps;
  CK(hipMalloc(&W, 16 << 20));
  CK(hipMalloc(&W2, 16 << 20));
  CK(hipMalloc(&big, BIG));
  CK(hipMalloc(&sink, NB * 4));
  CK(hipMalloc(&stamps, NB * 16));
  CK(hipMemset(W, 1, 16 << 20));
  CK(hipMemset(W2, 2, 16 << 20));
  CK(hipMemset(big, 3, BIG));
  std::vector<u64> h(2 * NB);
  printf("{\n");
  bool first = true;
  for (int nt = 0; nt < 2; ++nt)
    for (int mb : {2, 8})
      for (int mode = 0; mode < 3; ++mode) {
        const int vpb = (mb << 20) / 16 / NB;
        std::vector<double> body;
        for (int rep = 0; rep < 12; ++rep) {
          hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, big, BIG / 16, sink);
          const u32x4* a_src = mode == 0 ? W2 : W;
          const int a_shift = mode == 2 ? 1 : 0;
          if (nt) {
            hipLaunchKernelGGL((reader<true>), dim3(NB), dim3(256), 0, 0, a_src, vpb, a_shift, sink, stamps);
            hipLaunchKernelGGL((reader<true>), dim3(NB), dim3(256), 0, 0, W, vpb, 0, sink, stamps);
          } else {
            hipLaunchKernelGGL((reader<false>), dim3(NB), dim3(256), 0, 0, a_src, vpb, a_shift, sink, stamps);
            hipLaunchKernelGGL((reader<false>), dim3(NB), dim3(256), 0, 0, W, vpb, 0, sink, stamps);
          }
          CK(hipDeviceSynchronize());
          CK(hipMemcpy(h.data(), stamps, NB * 16, hipMemcpyDeviceToHost));
          u64 lo = ~0ull, hi = 0;
          for (int b = 0; b < NB; ++b) {
            lo = std::min(lo, h[2 * b]);
            hi = std::max(hi, h[2 * b + 1]);
          }
          if (rep >= 2) body.push_back((double)(hi - lo) * 0.01);  // 100 MHz wall clock -> us
        }
        std::sort(body.begin(), body.end());
        printf("%s \"%s_%dMB_%s\": {\"body_us_median\": %.2f, \"min\": %.2f, \"max\": %.2f}", first ? "" : ",\n", nt ? "nt" : "default", mb,
               mode == 0 ? "cold" : mode == 1 ? "same" : "shifted", body[body.size() / 2], body.front(), body.back());
        first = false;
      }
  printf("\n}\n");
  return 0;
}
