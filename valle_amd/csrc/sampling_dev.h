// Device-side pieces of the AR sampling step shared by sampling.hip (ar_sample_kernel, topk_sample_rows_kernel) and persist.hip
// (the persistent batch-1 step samples inside its launch): Philox4x32-10, block reductions of a 256-thread block, topk_sampling of
// one row held as SAMP_PER logits per thread, arg-max of the row.
//   reference: topk_sampling valle/models/valle.py:1287-1302, top_k_top_p_filtering :1242-1284.
#pragma once
#include "common.h"

namespace vle {

// ---- Philox4x32-10 counter-based RNG -----------------------------------------------------------
__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ inline float philox_uniform(uint64_t seed, uint32_t ctr0, uint32_t ctr1) {
  uint32_t c[4] = {ctr0, ctr1, 0x9E3779B9u, 0xBB67AE85u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

constexpr int SAMP_T = 256;
constexpr int SAMP_PER = 5;  // 256 * 5 >= 1026 logits

__device__ inline unsigned long long block_max_u64(unsigned long long v, unsigned long long* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = red[0];
#pragma unroll
  for (int i = 1; i < SAMP_T / 64; ++i) r = red[i] > r ? red[i] : r;
  return r;
}
__device__ inline int block_sum_i(int v, int* red) {
  v = wave_sum_i(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = 0;
#pragma unroll
  for (int i = 0; i < SAMP_T / 64; ++i) r += red[i];
  return r;
}

// topk_sampling (valle/models/valle.py:1287-1302) of ONE row held by the block as SAMP_PER logits per thread: temperature, top-k
// filter (ties at the k-th value kept, :1259-1260), softmax, inverse-CDF draw with u = Philox(rseed, it).  Block-wide: every thread
// of the SAMP_T-thread block calls it; returns the drawn index (arg-max when top_k == 1).
struct SampScratch {
  unsigned long long* red64;
  int* redi;
  float* redf;
  float* wave_tot;
};
__device__ inline int sample_row(const float (&raw)[SAMP_PER], int V, int top_k, float temperature, unsigned long long rseed, uint32_t it,
                                 int argmax, const SampScratch& sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long* const red64 = sh.red64;
  int* const redi = sh.redi;
  float* const redf = sh.redf;
  float* const wave_tot = sh.wave_tot;
  int sample = argmax;
  if (top_k != 1) {
    float sc[SAMP_PER];
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) sc[j] = temperature != 1.0f ? raw[j] / temperature : raw[j];
    uint32_t thr_key = 0u;  // keep keys >= thr_key
    if (top_k > 1 && top_k < V) {
      // k-th largest via bitwise binary search on the order-preserving key:
      // largest K with count(key >= K) >= top_k.  Ties at the k-th value are all kept, like
      // `logits < topk(logits, k)[0][..., -1]` (valle.py:1259-1260).
      uint32_t cur = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = cur | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < SAMP_PER; ++j) cnt += (tid * SAMP_PER + j < V) && (float_key(sc[j]) >= cand);
        cnt = block_sum_i(cnt, redi);
        if (cnt >= top_k) cur = cand;
      }
      thr_key = cur;
    }
    // softmax over kept entries (max = global max, always kept)
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    __syncthreads();
    if (lane == 0) redf[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    float p[SAMP_PER];
    float local = 0.f;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) {
      const int idx = tid * SAMP_PER + j;
      const bool keep = idx < V && float_key(sc[j]) >= thr_key;
      p[j] = keep ? expf(sc[j] - m) : 0.f;
      local += p[j];
    }
    // block-wide exclusive prefix of `local` (inclusive wave scan + per-wave totals)
    float incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    float base = 0.f, total = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_T / 64; ++i) {
      if (i < w) base += wave_tot[i];
      total += wave_tot[i];
    }
    const float excl = base + incl - local;
    const float u = philox_uniform(rseed, it, 0u);
    const float target = u * total;
    int cand = 0x7fffffff;
    float run = excl;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) {
      const int idx = tid * SAMP_PER + j;
      run += p[j];
      if (p[j] > 0.f && run > target && cand == 0x7fffffff) cand = idx;
    }
    // first index whose inclusive cumulative mass exceeds the target
    const unsigned long long ck = block_max_u64((unsigned long long)(0x7fffffff - cand), red64);
    const int pick = 0x7fffffff - (int)ck;
    sample = pick == 0x7fffffff ? argmax : pick;
  }

  return sample;
}

// arg-max of the row (highest value; among equal values the LOWEST index, torch.argmax's convention) from the per-thread logits
__device__ inline int argmax_row(const float (&raw)[SAMP_PER], int V, unsigned long long* red64) {
  const int tid = threadIdx.x;
  unsigned long long best = 0ull;
#pragma unroll
  for (int j = 0; j < SAMP_PER; ++j) {
    const int idx = tid * SAMP_PER + j;
    if (idx < V) {
      const unsigned long long key = ((unsigned long long)float_key(raw[j]) << 32) | (unsigned)(0x7fffffff - idx);
      best = key > best ? key : best;
    }
  }
  best = block_max_u64(best, red64);
  return 0x7fffffff - (int)(best & 0xffffffffu);
}

}  // namespace vle
