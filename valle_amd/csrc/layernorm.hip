// LayerNorm over packed rows (prefill / NAR passes): one wave64 per row, fp32 in, T out.
// Replaces F.layer_norm as called from LayerNorm.forward (valle/modules/transformer.py:57-74) and,
// with the per-stage folded gamma'/beta', AdaptiveLayerNorm.forward (:93-108).
// HBM-bound: 4*d bytes read + sizeof(T)*d written per row; float4 loads, 16 B per lane.
#include "common.h"
#include "kernels.h"

namespace vle {

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_rows_kernel(const float* __restrict__ x,
                                                                 const int32_t* __restrict__ row_map,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ out,
                                                                 int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t src = row_map ? (int64_t)row_map[r] : r;
  const float* xr = x + src * d;
  const int nv = d >> 2;  // d % 4 == 0
  float s = 0.f;
  for (int i = lane; i < nv; i += 64) {
    const float4 v = reinterpret_cast<const float4*>(xr)[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int i = lane; i < nv; i += 64) {
    const float4 v = reinterpret_cast<const float4*>(xr)[i];
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + LN_EPS);
  T* orow = out + r * d;
  for (int i = lane; i < nv; i += 64) {
    const float4 v = reinterpret_cast<const float4*>(xr)[i];
    const float4 g = reinterpret_cast<const float4*>(gamma)[i];
    const float4 bb = reinterpret_cast<const float4*>(beta)[i];
    const float o0 = (v.x - mean) * rstd * g.x + bb.x;
    const float o1 = (v.y - mean) * rstd * g.y + bb.y;
    const float o2 = (v.z - mean) * rstd * g.z + bb.z;
    const float o3 = (v.w - mean) * rstd * g.w + bb.w;
    if constexpr (sizeof(T) == 4) {
      reinterpret_cast<float4*>(orow)[i] = make_float4(o0, o1, o2, o3);
    } else {
      uint2 p;
      p.x = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);
      p.y = (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16);
      reinterpret_cast<uint2*>(orow)[i] = p;
    }
  }
}

// Row held in registers (d = NV * 256): one read of x, gamma / beta requested in the same burst,
// DPP reductions -- one memory round trip instead of three dependent passes.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_rows_reg_kernel(const float* __restrict__ x,
                                                                 const int32_t* __restrict__ row_map,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ out,
                                                                 int64_t rows) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int d = NV * 256;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t src = row_map ? (int64_t)row_map[r] : r;
  const f4* xr = reinterpret_cast<const f4*>(x + src * d) + lane;
  f4 v[NV], g[NV], bb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[i * 64];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = reinterpret_cast<const f4*>(gamma)[i * 64 + lane];
    bb[i] = reinterpret_cast<const f4*>(beta)[i * 64 + lane];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum_dpp(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)d + LN_EPS);
  T* orow = out + r * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const f4 o = (v[i] - mean) * rstd * g[i] + bb[i];
    if constexpr (sizeof(T) == 4) {
      reinterpret_cast<f4*>(orow)[i * 64 + lane] = o;
    } else {
      typedef __bf16 b4 __attribute__((ext_vector_type(4)));
      b4 p;
      p[0] = (__bf16)o.x; p[1] = (__bf16)o.y; p[2] = (__bf16)o.z; p[3] = (__bf16)o.w;  // round-to-nearest-even
      reinterpret_cast<b4*>(orow)[i * 64 + lane] = p;
    }
  }
}

// The same register-resident LayerNorm for the <= 64 rows of the batched AR step, written bf16 in the fragment-major
// layout the weight-streaming GEMM reads (common.h xf_index); identical arithmetic and rounding.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_rows_xf_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16_t* __restrict__ out, int rows,
                                                                int MF, int w8) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef __bf16 b4 __attribute__((ext_vector_type(4)));
  constexpr int d = NV * 256;
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const f4* xr = reinterpret_cast<const f4*>(x + (int64_t)r * d) + lane;
  f4 v[NV], g[NV], bb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[i * 64];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = reinterpret_cast<const f4*>(gamma)[i * 64 + lane];
    bb[i] = reinterpret_cast<const f4*>(beta)[i * 64 + lane];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum_dpp(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)d + LN_EPS);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const f4 o = (v[i] - mean) * rstd * g[i] + bb[i];
    b4 p;
    p[0] = (__bf16)o.x; p[1] = (__bf16)o.y; p[2] = (__bf16)o.z; p[3] = (__bf16)o.w;
    *reinterpret_cast<b4*>(out + xf_index(r, (i * 64 + lane) * 4, MF, w8 != 0)) = p;
  }
}

int launch_layernorm_xf(hipStream_t st, const float* x, const float* gamma, const float* beta, void* out, int rows, int d, int w8) {
  if (rows <= 0) return 0;
  if (rows > 64 || d % 256 != 0) return -1;
  const int MF = (rows + 15) / 16;
  const dim3 grid((rows + 3) / 4), block(256);
#define VLE_LNX(NV) hipLaunchKernelGGL((layernorm_rows_xf_kernel<NV>), grid, block, 0, st, x, gamma, beta, (bf16_t*)out, rows, MF, w8)
  switch (d / 256) {
    case 1: VLE_LNX(1); break;
    case 2: VLE_LNX(2); break;
    case 3: VLE_LNX(3); break;
    case 4: VLE_LNX(4); break;
    case 6: VLE_LNX(6); break;
    case 8: VLE_LNX(8); break;
    default: return -1;
  }
#undef VLE_LNX
  return 0;
}

template <typename T>
static bool layernorm_reg_dispatch(hipStream_t st, const float* x, const int32_t* row_map, const float* gamma, const float* beta,
                                   T* out, int64_t rows, int d) {
  if (d % 256 != 0) return false;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define VLE_LN(NV) hipLaunchKernelGGL((layernorm_rows_reg_kernel<T, NV>), grid, block, 0, st, x, row_map, gamma, beta, out, rows)
  switch (d / 256) {
    case 1: VLE_LN(1); return true;
    case 2: VLE_LN(2); return true;
    case 3: VLE_LN(3); return true;
    case 4: VLE_LN(4); return true;
    case 6: VLE_LN(6); return true;
    case 8: VLE_LN(8); return true;
    default: return false;
  }
#undef VLE_LN
}

int launch_layernorm(hipStream_t st, int dtype, const float* x, const int32_t* row_map, const float* gamma,
                     const float* beta, void* out, int64_t rows, int d) {
  if (rows <= 0) return 0;
  if (d % 4 != 0) return -1;
  if (dtype == DT_F32 ? layernorm_reg_dispatch<float>(st, x, row_map, gamma, beta, (float*)out, rows, d)
                      : layernorm_reg_dispatch<bf16_t>(st, x, row_map, gamma, beta, (bf16_t*)out, rows, d))
    return 0;
  constexpr int NW = 4;
  const dim3 grid((unsigned)((rows + NW - 1) / NW)), block(NW * 64);
  if (dtype == DT_F32)
    hipLaunchKernelGGL((layernorm_rows_kernel<float, NW>), grid, block, 0, st, x, row_map, gamma, beta, (float*)out, rows, d);
  else
    hipLaunchKernelGGL((layernorm_rows_kernel<bf16_t, NW>), grid, block, 0, st, x, row_map, gamma, beta, (bf16_t*)out, rows, d);
  return 0;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ row_map,
                                   float* __restrict__ dst, int rows, int d) {
  const int r = blockIdx.x;
  const float4* s = reinterpret_cast<const float4*>(src + (int64_t)row_map[r] * d);
  float4* o = reinterpret_cast<float4*>(dst + (int64_t)r * d);
  for (int i = threadIdx.x; i < (d >> 2); i += blockDim.x) o[i] = s[i];
}

int launch_gather_rows(hipStream_t st, const float* src, const int32_t* row_map, float* dst, int rows, int d) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, st, src, row_map, dst, rows, d);
  return 0;
}

}  // namespace vle
