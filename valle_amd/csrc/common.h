// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the VALL-E decode engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vle {

constexpr int WAVE = 64;
constexpr float LN_EPS = 1e-5f;  // valle/modules/transformer.py:27

// ---- element types ------------------------------------------------------------------------
struct bf16_t {
  uint16_t v;
};

__host__ __device__ inline float bf16_to_f32(uint16_t v) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
// round-to-nearest-even, same as torch's float -> bfloat16
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
  union {
    uint32_t u;
    float f;
  } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// ---- fp8 weights (engine mode FP8W) -------------------------------------------------------------
// OCP e4m3fn ("fn": finite, no infinities; bias 7, max 448, 0x7f / 0xff = NaN): the format of gfx950's
// v_cvt_pk_f32_fp8 and of torch.float8_e4m3fn.  A weight row is stored as q[k] (e4m3fn) and ONE power-of-two
// scale 2^e, so W'[k] = q[k] * 2^e is exactly representable in bf16: the bandwidth-bound AR step streams q,
// the MFMA-bound passes run on the bf16 copy of the SAME values.
struct bf16w8_t {  // element type tag of the AR-step kernels in FP8W mode: fp8 weights, bf16 KV cache
  uint8_t v;
};

__host__ __device__ inline float e4m3fn_to_f32(uint8_t v) {
  const int e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 15 && m == 7) {
    union { uint32_t u; float f; } c;
    c.u = 0x7fc00000u;
    r = c.f;
  } else if (e == 0) {
    r = (float)m * 0.001953125f;  // m * 2^-9
  } else {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)(e + 120) << 23) | ((uint32_t)m << 20);
    r = c.f;
  }
  return (v & 0x80) ? -r : r;
}
// round-to-nearest-even, saturating at +-448 (callers scale rows so that |x| <= 448; NaN -> 0x7f)
__host__ inline uint8_t f32_to_e4m3fn(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  const uint8_t sign = (uint8_t)((c.u >> 24) & 0x80);
  const uint32_t a = c.u & 0x7fffffffu;
  if (a > 0x7f800000u) return 0x7f;
  if (a >= 0x43e80000u) return sign | 0x7e;  // >= 464 would round past the largest finite value: saturate at 448
  if (a < 0x3c800000u) {                      // < 2^-6: subnormal range, spacing 2^-9
    c.u = a;
    const float t = c.f * 512.0f;             // exact
    const float r = __builtin_rintf(t);       // RNE in the default rounding mode
    return sign | (uint8_t)r;                 // 0..8 (8 = the smallest normal, encoded 0x08)
  }
  uint32_t u = a;
  const uint32_t odd = (u >> 20) & 1u;
  u += 0x7ffffu + odd;                        // RNE at bit 20
  const uint32_t e = (u >> 23) - 120u, m = (u >> 20) & 7u;
  return sign | (uint8_t)((e << 3) | m);
}

// ---- fragment-major ("XF") layout of the AR-step activations at 2..64 utterances --------------------------------
// gemm_skinny.hip reads its [M <= 64][K] bf16 X operand one MFMA B-fragment (16 rows x 8 k per lane) at a time; stored
// row-major, a fragment load touches 16 rows x 64 B (half lines) and the texture path of every CU is the bottleneck
// (measured: AR step of 64 utterances 675 -> 611 ms with this layout).  XF stores each fragment contiguously in lane
// order: element (m, k) of an [MF*16][K] matrix sits at
//     ((c*2 + s)*MF + i)*512 + (fg*16 + fr)*8 + j        c = k / 64, i = m / 16, fr = m % 16, j = k % 8
// where (s, fg) split k % 64 the way the consumer's lanes do: bf16 weights  s = (k%64)/32, fg = (k%32)/8;
// fp8 weights (one 16-byte W vector = 16 consecutive k per lane)  fg = (k%64)/16, s = (k%16)/8.
// Four consecutive k (k % 4 == 0) of one row stay contiguous, so the producers' 8-byte stores survive.
__host__ __device__ inline int64_t xf_index(int m, int k, int MF, bool w8) {
  const int c = k >> 6, kk = k & 63, j = kk & 7;
  const int s = w8 ? ((kk >> 3) & 1) : (kk >> 5);
  const int fg = w8 ? (kk >> 4) : ((kk >> 3) & 3);
  return ((int64_t)(c * 2 + s) * MF + (m >> 4)) * 512 + ((fg << 4) + (m & 15)) * 8 + j;
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static inline float to_f32(float v) { return v; }
  __device__ static inline float from_f32(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static inline float to_f32(bf16_t v) { return bf16_to_f32(v.v); }
  __device__ static inline bf16_t from_f32(float v) {
    bf16_t r;
    r.v = f32_to_bf16(v);
    return r;
  }
};

template <>
struct Elem<bf16w8_t> {
  static constexpr int VEC = 8;  // a lane's chunk is 8 elements as in bf16 mode (8 bytes of fp8)
};
struct bf16w8t_t {  // bf16w8_t with default-policy (cacheable) weight loads: at one utterance the fp8 AR weights (152 MB at
  uint8_t v;        // C2) fit the 256 MB memory-side cache, which non-temporal loads would bypass
};
template <>
struct Elem<bf16w8t_t> {
  static constexpr int VEC = 8;
};

// 16-byte vector load of VEC elements, widened to fp32
template <typename T>
__device__ inline void load_vec16(const T* p, float (&out)[Elem<T>::VEC]);
template <>
__device__ inline void load_vec16<float>(const float* p, float (&out)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x;
  out[1] = v.y;
  out[2] = v.z;
  out[3] = v.w;
}
template <>
__device__ inline void load_vec16<bf16_t>(const bf16_t* p, float (&out)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  out[0] = __uint_as_float(v.x << 16);
  out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16);
  out[3] = __uint_as_float(v.y & 0xffff0000u);
  out[4] = __uint_as_float(v.z << 16);
  out[5] = __uint_as_float(v.z & 0xffff0000u);
  out[6] = __uint_as_float(v.w << 16);
  out[7] = __uint_as_float(v.w & 0xffff0000u);
}

template <typename T>
__device__ inline void store_elem(T* p, float v);
template <>
__device__ inline void store_elem<float>(float* p, float v) {
  *p = v;
}
template <>
__device__ inline void store_elem<bf16_t>(bf16_t* p, float v) {
  p->v = f32_to_bf16(v);
}

// ---- wave / block reductions (wave = 64 lanes) ----------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- DPP wave reductions --------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute (an LDS-crossbar round trip per step); the AR-step kernels are
// latency-bound chains, so their reductions use DPP row operations (VALU rate) for the 16-lane rows
// and v_readlane for the four row totals.  Every lane gets the result; the add order is fixed.
template <int CTRL>
__device__ inline float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ inline float row16_sum_dpp(float v) {  // total of each 16-lane row, in every lane of the row
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror
  v += dpp_f32<0x140>(v);  // row_mirror
  return v;
}
__device__ inline float row16_max_dpp(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}
__device__ inline float readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// Sum / max of v over the 4 lanes {l, l ^ 16, l ^ 32, l ^ 48} (the four 16-lane rows of the wave), the same value in all
// four: gfx950's v_permlane16_swap / v_permlane32_swap instead of two ds_bpermute round trips through the LDS (~100 cycles
// each, on the epilogue's critical path of every batched AR GEMM).  swap(x, x) leaves {own, partner} (in either order) in the
// two results; add and max are commutative, so the partners agree bitwise.
typedef unsigned permlane_u32x2 __attribute__((ext_vector_type(2)));
__device__ inline float rows4_sum(float v) {
  permlane_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float rows4_max(float v) {
  permlane_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ inline float wave_sum_dpp(float v) {
  v = row16_sum_dpp(v);
  return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ inline float wave_max_dpp(float v) {
  v = row16_max_dpp(v);
  return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

// Block-wide sum for blockDim.x == NW*64; `red` is NW floats of LDS; all threads get the result.
template <int NW>
__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from a previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// ---- in-kernel timeline of the AR step (option "ktrace"; SURVEY.md 8d "kernel-gap timeline") ---------------------------------
// rocprofv3's kernel trace serialises the dispatches of a graph replay (every kernel reads ~4.7 us, gaps 0), so the
// un-perturbed timeline is taken by the kernels themselves: every wave stamps the constant-rate wall clock (100 MHz) at
// entry and before its epilogue store into ITS OWN 16-byte slot [step & 31][kernel index][wave id] (plain stores, nothing
// waits for them; atomics on shared min/max words cost ~70 us per kernel).  buf == null (always, outside the
// diagnostic) costs one uniform branch.
constexpr int KT_STEPS = 32, KT_KERNELS = 64, KT_WAVES = 2048;
struct KTrace {
  unsigned long long* buf = nullptr;  // [KT_STEPS][KT_KERNELS][KT_WAVES][4]: entry, two optional phase marks, end
  const int32_t* step = nullptr;      // device word that counts AR iterations
  int idx = 0;                        // kernel index within the step
};
__device__ inline unsigned long long ktrace_begin(const KTrace& t) { return t.buf ? wall_clock64() : 0ull; }
__device__ inline unsigned long long ktrace_mark(const KTrace& t) { return t.buf ? wall_clock64() : 0ull; }
__device__ inline void ktrace_end(const KTrace& t, unsigned long long t0, int wave_id, unsigned long long m1 = 0ull,
                                  unsigned long long m2 = 0ull) {  // call from ONE lane per wave
  if (!t.buf || wave_id >= KT_WAVES || t.idx >= KT_KERNELS) return;
  typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
  const unsigned long long t1 = wall_clock64();
  ull2* p = reinterpret_cast<ull2*>(t.buf + (((size_t)((*t.step) & (KT_STEPS - 1)) * KT_KERNELS + t.idx) * KT_WAVES + wave_id) * 4);
  ull2 a, b;
  a.x = t0; a.y = m1 ? m1 : t1;
  b.x = m2 ? m2 : t1; b.y = t1;
  p[0] = a;
  p[1] = b;
}

// RNG stream of one utterance: splitmix64 of (call seed, request index).  The multinomial draw of iteration `it` is
// Philox(request_seed, it) -- independent of the batch position / slot the utterance happens to occupy, so sampled decodes are
// reproducible across max_batch and scheduling choices, and two requests that use the same slot one after the other differ.
__host__ __device__ inline unsigned long long request_seed(unsigned long long seed, unsigned long long request) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (request + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// order-preserving float <-> uint key (for arg-max with lowest-index tie-break and k-th largest)
__device__ inline uint32_t float_key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#define VLE_HIP_CHECK(expr)                                                 \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) return vle::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

int hip_fail(hipError_t e, const char* expr, const char* file, int line);
void set_global_error(const char* msg);

}  // namespace vle
