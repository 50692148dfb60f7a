// Device helpers shared by the batch-1 AR-step kernels (gemv1.hip: the launch chain; persist.hip: the persistent step):
// 16-byte weight-vector widening, weight-stream traits, the per-wave dot product, the block-shared LayerNorm and the LDS
// activation row.  Both files call the SAME functions so that the persistent step is bit-identical to the launch chain.
#pragma once
#include "common.h"

namespace vle {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

constexpr int G1_T = 256;  // 4 independent waves per block
constexpr float G1_NEG = -1e30f;

template <typename T>
__device__ inline void widen16(const u32x4_t& v, float (&f)[Elem<T>::VEC]);
template <>
__device__ inline void widen16<float>(const u32x4_t& v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <>
__device__ inline void widen16<bf16_t>(const u32x4_t& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// FP8W: the lane's 8 weights of a chunk are 8 bytes of e4m3fn in v.x / v.y (v_cvt_pk_f32_fp8: two values per op)
template <>
__device__ inline void widen16<bf16w8_t>(const u32x4_t& v, float (&f)[8]) {
  typedef float f32x2v_t __attribute__((ext_vector_type(2)));
  const f32x2v_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
  const f32x2v_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
  f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y; f[6] = d.x; f[7] = d.y;
}

template <>
__device__ inline void widen16<bf16w8t_t>(const u32x4_t& v, float (&f)[8]) {
  widen16<bf16w8_t>(v, f);
}

// weight-stream traits: how a lane fetches its VEC weights of one chunk, and the element type of the KV cache
template <typename T>
struct G1W {
  typedef T cache_t;
  static constexpr bool kScaled = false;
  __device__ static inline u32x4_t load(const T* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
  // rows several workgroups of one XCD read in the same launch (the query rows of the fused QKV + attention launch): default
  // cache policy, so that the first miss leaves the line in that XCD's L2 for the others
  __device__ static inline u32x4_t load_shared(const T* p) { return *reinterpret_cast<const u32x4_t*>(p); }
};
template <>
struct G1W<bf16w8_t> {
  typedef bf16_t cache_t;
  static constexpr bool kScaled = true;
  __device__ static inline u32x4_t load(const bf16w8_t* p) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return u32x4_t{t.x, t.y, 0u, 0u};
  }
  __device__ static inline u32x4_t load_shared(const bf16w8_t* p) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = *reinterpret_cast<const u32x2_t*>(p);
    return u32x4_t{t.x, t.y, 0u, 0u};
  }
};

template <>
struct G1W<bf16w8t_t> {
  typedef bf16_t cache_t;
  static constexpr bool kScaled = true;
  __device__ static inline u32x4_t load(const bf16w8t_t* p) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = *reinterpret_cast<const u32x2_t*>(p);  // default cache policy
    return u32x4_t{t.x, t.y, 0u, 0u};
  }
  __device__ static inline u32x4_t load_shared(const bf16w8t_t* p) { return load(p); }
};

// VEC consecutive fp32 values at p (16-byte aligned)
template <int VEC>
__device__ inline void load_f32_vec(const float* p, float (&f)[VEC]) {
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const f32x4v_t t = *reinterpret_cast<const f32x4v_t*>(p + q * 4);
    f[q * 4 + 0] = t.x; f[q * 4 + 1] = t.y; f[q * 4 + 2] = t.z; f[q * 4 + 3] = t.w;
  }
}

// sum over the lpk consecutive lanes that hold one head (lpk = dh / VEC: a power of two <= 32, wave-uniform), in every lane
__device__ inline float head_group_sum(float v, int lpk) {
  if (lpk >= 2) v += dpp_f32<0xB1>(v);
  if (lpk >= 4) v += dpp_f32<0x4E>(v);
  if (lpk >= 8) v += dpp_f32<0x141>(v);
  if (lpk >= 16) v += dpp_f32<0x140>(v);
  if (lpk >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

template <typename T, int NCH>
__device__ inline float g1_dot(const u32x4_t (&wv)[NCH], const float (&x)[NCH][Elem<T>::VEC]) {
  constexpr int VEC = Elem<T>::VEC;
  typedef float f32x2p_t __attribute__((ext_vector_type(2)));
  f32x2p_t t = f32x2p_t{0.f, 0.f};  // .x: even elements, .y: odd elements -- two chains per row, one v_pk_fma_f32 per pair
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float wf[VEC];
    widen16<T>(wv[c], wf);
#pragma unroll
    for (int j = 0; j < VEC; j += 2) t = __builtin_elementwise_fma(f32x2p_t{wf[j], wf[j + 1]}, f32x2p_t{x[c][j], x[c][j + 1]}, t);
  }
  return t.x + t.y;
}

__device__ inline void g1_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// EPT consecutive fp32 values at p (p aligned to the widest vector EPT allows)
template <int EPT>
__device__ inline void load_ept(const float* p, float (&f)[EPT]) {
  if constexpr (EPT % 4 == 0) {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
      const f32x4v_t t = *reinterpret_cast<const f32x4v_t*>(p + q * 4);
      f[q * 4 + 0] = t.x; f[q * 4 + 1] = t.y; f[q * 4 + 2] = t.z; f[q * 4 + 3] = t.w;
    }
  } else if constexpr (EPT % 2 == 0) {
    typedef float f32x2v_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < EPT / 2; ++q) {
      const f32x2v_t t = *reinterpret_cast<const f32x2v_t*>(p + q * 2);
      f[q * 2 + 0] = t.x; f[q * 2 + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < EPT; ++q) f[q] = p[q];
  }
}
template <int EPT>
__device__ inline void store_ept_lds(float* p, const float (&f)[EPT]) {
  if constexpr (EPT % 4 == 0) {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) *reinterpret_cast<f32x4v_t*>(p + q * 4) = f32x4v_t{f[q * 4], f[q * 4 + 1], f[q * 4 + 2], f[q * 4 + 3]};
  } else {
#pragma unroll
    for (int q = 0; q < EPT; ++q) p[q] = f[q];
  }
}

// sum over the lpk consecutive lanes of a head, lpk a power of two <= 64 (wave-uniform)
__device__ inline float head_group_sum64(float v, int lpk) {
  if (lpk >= 64) return wave_sum_dpp(v);
  return head_group_sum(v, lpk);
}

// LayerNorm of the K-element row whose elements [t * EPT, (t + 1) * EPT) this thread holds (valle/modules/transformer.py:57-74,
// eps 1e-5, biased variance, two-pass fp32) -> sx[K]; red = 8 floats of LDS.  Ends with a barrier: sx is readable.
template <int K, int NT = G1_T>
__device__ inline void g1_block_layernorm(const float (&xv)[K / NT], const float (&gv)[K / NT], const float (&bv)[K / NT], float* sx, float* red) {
  constexpr int EPT = K / NT, NWV = NT / 64;  // red: 2 * NWV floats
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < EPT; ++i) s += xv[i];
  s = wave_sum_dpp(s);
  if (lane == 0) red[w] = s;
  g1_lds_barrier();
  float rs = (red[0] + red[1]) + (red[2] + red[3]);
  if constexpr (NWV == 8) rs += (red[4] + red[5]) + (red[6] + red[7]);
  const float mean = rs * (1.0f / (float)K);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const float t = xv[i] - mean;
    q = fmaf(t, t, q);
  }
  q = wave_sum_dpp(q);
  if (lane == 0) red[NWV + w] = q;
  g1_lds_barrier();
  float rq = (red[NWV] + red[NWV + 1]) + (red[NWV + 2] + red[NWV + 3]);
  if constexpr (NWV == 8) rq += (red[NWV + 4] + red[NWV + 5]) + (red[NWV + 6] + red[NWV + 7]);
  const float rstd = 1.0f / sqrtf(rq * (1.0f / (float)K) + LN_EPS);
  float o[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) o[i] = (xv[i] - mean) * rstd * gv[i] + bv[i];
  store_ept_lds<EPT>(sx + tid * EPT, o);
  g1_lds_barrier();
}

// this lane's K / 64 activations of the wave-level dot products, from the shared row
template <typename T, int NCH>
__device__ inline void g1_read_shared(const float* sx, float (&x)[NCH][Elem<T>::VEC]) {
  constexpr int VEC = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const f32x4v_t t = *reinterpret_cast<const f32x4v_t*>(sx + c * 64 * VEC + lane * VEC + q * 4);
      x[c][q * 4 + 0] = t.x; x[c][q * 4 + 1] = t.y; x[c][q * 4 + 2] = t.z; x[c][q * 4 + 3] = t.w;
    }
}

}  // namespace vle
