// Small glue kernels: the batch > 8 AR step (which runs its linears on the MFMA GEMM path instead of
// the weight-streaming GEMV path) and packing of the result codes.
#include "common.h"
#include "kernels.h"

namespace vle {

template <typename T>
__global__ __launch_bounds__(256) void qkv_split_kernel(const T* __restrict__ qkv, float* __restrict__ q, T* __restrict__ kc,
                                                        T* __restrict__ vc, const int32_t* __restrict__ kv_len, int d, int nhead,
                                                        int ctx_max) {
  const int b = blockIdx.x;
  const int dh = d / nhead;
  const T* src = qkv + (int64_t)b * 3 * d;
  const int slot = kv_len[b];
  for (int j = threadIdx.x; j < d; j += 256) {
    q[(int64_t)b * d + j] = Elem<T>::to_f32(src[j]);
    const int h = j / dh, e = j - h * dh;
    const int64_t o = (((int64_t)b * nhead + h) * ctx_max + slot) * dh + e;
    kc[o] = src[d + j];
    vc[o] = src[2 * d + j];
  }
}

int launch_qkv_split(hipStream_t st, int dtype, const void* qkv, float* q, void* k_cache, void* v_cache, const int32_t* kv_len,
                     int B, int d, int nhead, int ctx_max) {
  if (B <= 0) return 0;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(qkv_split_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)qkv, q, (float*)k_cache,
                       (float*)v_cache, kv_len, d, nhead, ctx_max);
  else
    hipLaunchKernelGGL(qkv_split_kernel<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)qkv, q, (bf16_t*)k_cache,
                       (bf16_t*)v_cache, kv_len, d, nhead, ctx_max);
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                           T* __restrict__ out, int nhead, int dh, int ns) {
  const int b = blockIdx.x;
  const int d = nhead * dh;
  for (int i = threadIdx.x; i < d; i += 256) {
    const int h = i / dh;
    const float* ml = part_ml + ((int64_t)(b * nhead + h) * ns) * 2;
    const float* po = part_o + (int64_t)b * ns * d + i;  // [B][nsplit][d]
    float m = -1e30f;
    for (int s = 0; s < ns; ++s) m = fmaxf(m, ml[2 * s]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < ns; ++s) {
      const float f = expf(ml[2 * s] - m);
      l += ml[2 * s + 1] * f;
      o += po[(int64_t)s * d] * f;
    }
    store_elem<T>(out + (int64_t)b * d + i, o / l);
  }
}

int launch_attn_combine(hipStream_t st, int dtype, const float* part_o, const float* part_ml, void* out, int B, int nhead, int dh,
                        int nsplit) {
  if (B <= 0) return 0;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(attn_combine_kernel<float>, dim3(B), dim3(256), 0, st, part_o, part_ml, (float*)out, nhead, dh, nsplit);
  else
    hipLaunchKernelGGL(attn_combine_kernel<bf16_t>, dim3(B), dim3(256), 0, st, part_o, part_ml, (bf16_t*)out, nhead, dh, nsplit);
  return 0;
}

__global__ void codes_set_first_kernel(const int64_t* __restrict__ first_cb, int64_t fc_stride,
                                       const int32_t* __restrict__ grow_seq, const int32_t* __restrict__ grow_pos, int64_t rows,
                                       int64_t* __restrict__ codes, int64_t g_stride, int Q) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int b = grow_seq[r], g = grow_pos[r];
  codes[((int64_t)b * g_stride + g) * Q] = first_cb[(int64_t)b * fc_stride + g];
}

int launch_codes_set_first(hipStream_t st, const int64_t* first_cb, int64_t fc_stride, const int32_t* grow_seq,
                           const int32_t* grow_pos, int64_t rows, int64_t* codes, int64_t g_stride, int Q) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(codes_set_first_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, first_cb, fc_stride,
                     grow_seq, grow_pos, rows, codes, g_stride, Q);
  return 0;
}

// ---- slot API (continuous batching) -----------------------------------------------------------------
// AR state of newly admitted utterances: [kv_len | audio_pos | n_gen | done | cap | iter][max_B]
__global__ void slot_state_init_kernel(int32_t* state, int max_B, const int32_t* slots, const int32_t* kv_len,
                                       const int32_t* audio_pos, const int32_t* cap, int n, unsigned long long* slot_seed,
                                       unsigned long long seed, unsigned long long first_request) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = slots[i];
  if (slot_seed != nullptr) slot_seed[b] = request_seed(seed, first_request + (unsigned long long)i);  // RNG stream of the REQUEST, not of the slot
  state[0 * max_B + b] = kv_len[i];
  state[1 * max_B + b] = audio_pos[i];
  state[2 * max_B + b] = 0;
  state[3 * max_B + b] = 0;
  state[4 * max_B + b] = cap[i];
  state[5 * max_B + b] = 0;
}

int launch_slot_state_init(hipStream_t st, int32_t* state, int max_B, const int32_t* slots, const int32_t* kv_len,
                           const int32_t* audio_pos, const int32_t* cap, int n, unsigned long long* slot_seed, unsigned long long seed,
                           unsigned long long first_request) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(slot_state_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, state, max_B, slots, kv_len, audio_pos, cap, n,
                     slot_seed, seed, first_request);
  return 0;
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ src_rows, float* __restrict__ dst,
                                    const int32_t* __restrict__ dst_rows, int d) {
  const int r = blockIdx.x;
  const float4* s = reinterpret_cast<const float4*>(src + (int64_t)src_rows[r] * d);
  float4* o = reinterpret_cast<float4*>(dst + (int64_t)dst_rows[r] * d);
  for (int i = threadIdx.x; i < (d >> 2); i += blockDim.x) o[i] = s[i];
}

int launch_scatter_rows(hipStream_t st, const float* src, const int32_t* src_rows, float* dst, const int32_t* dst_rows, int rows, int d) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows), dim3(256), 0, st, src, src_rows, dst, dst_rows, d);
  return 0;
}

// ---- fragment-major copy of a weight matrix for gemm_skinny.hip -----------------------------------------------
// W[N][K] (bf16, or e4m3fn codes) -> per 16-row block nb and 64-deep chunk c the bytes of each MFMA A-fragment in lane
// order, so that a fragment load is one contiguous 1 KB instead of 16 rows x 64 B:
//   bf16: dst[((nb*(K/64) + c)*2 + s)*512 + lane*8 + j] = W[nb*16 + fr][c*64 + s*32 + fg*8 + j]     lane = fg*16 + fr
//   fp8 : dst[(nb*(K/64) + c)*1024 + lane*16 + j]       = W[nb*16 + fr][c*64 + fg*16 + j]
// (rows beyond N repeat row N-1, as the kernel's own clamp does).  One 16-byte vector per thread.
__global__ void pack_w_frag_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int N, int K, int fp8) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // destination vector index
  const int kc = K >> 6, nblk = (N + 15) / 16;
  const int64_t total = (int64_t)nblk * kc * (fp8 ? 64 : 128);
  if (v >= total) return;
  const int lane = (int)(v & 63), fr = lane & 15, fg = lane >> 4;
  int64_t t = v >> 6;
  int s = 0;
  if (!fp8) {
    s = (int)(t & 1);
    t >>= 1;
  }
  const int c = (int)(t % kc), nb = (int)(t / kc);
  const int row = min(nb * 16 + fr, N - 1);
  const int64_t k = fp8 ? (int64_t)c * 64 + fg * 16 : (int64_t)c * 64 + s * 32 + fg * 8;
  const int es = fp8 ? 1 : 2;
  *reinterpret_cast<uint4*>(dst + v * 16) = *reinterpret_cast<const uint4*>(src + ((int64_t)row * K + k) * es);
}

int launch_pack_w_frag(hipStream_t st, const void* src, void* dst, int N, int K, int fp8) {
  if (K % 64 != 0 || N < 1) return -1;
  const int64_t total = (int64_t)((N + 15) / 16) * (K >> 6) * (fp8 ? 64 : 128);
  hipLaunchKernelGGL(pack_w_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const unsigned char*)src,
                     (unsigned char*)dst, N, K, fp8);
  return 0;
}

// Range check of token ids on the engine-owned copies of the inputs (the reference's nn.Embedding raises IndexError,
// valle/modules/embedding.py:34,44): utterance i's rows live at ids + (slot_map ? slot_map[i] : i) * stride0, rows are
// row_stride ids apart and their first `inner` ids are checked; the first id of a row must be < limit0, the others < limit_rest.  An offending id is REPLACED by 0 (the
// gathers downstream stay in bounds) and the flag is raised; the host turns the flag into VLE_EINDEX.
__global__ __launch_bounds__(256) void check_ids_kernel(int64_t* __restrict__ ids, int64_t stride0, int row_stride, int inner,
                                                        const int32_t* __restrict__ lens, const int32_t* __restrict__ slot_map, int n,
                                                        int max_rows, int limit0, int limit_rest, int32_t* __restrict__ flag, int code) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)n * max_rows) return;
  const int i = (int)(t / max_rows), r = (int)(t % max_rows);
  if (r >= lens[i]) return;
  int64_t* row = ids + (int64_t)(slot_map ? slot_map[i] : i) * stride0 + (int64_t)r * row_stride;
  bool bad = false;
  for (int j = 0; j < inner; ++j) {
    const int64_t v = row[j];
    if (v < 0 || v >= (j == 0 ? limit0 : limit_rest)) {
      row[j] = 0;
      bad = true;
    }
  }
  if (bad) atomicOr(flag, code);
}

int launch_check_ids(hipStream_t st, int64_t* ids, int64_t stride0, int row_stride, int inner, const int32_t* lens, const int32_t* slot_map,
                     int n, int max_rows, int limit0, int limit_rest, int32_t* flag, int code) {
  if (n <= 0 || max_rows <= 0) return 0;
  const int64_t total = (int64_t)n * max_rows;
  hipLaunchKernelGGL(check_ids_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ids, stride0, row_stride, inner, lens, slot_map,
                     n, max_rows, limit0, limit_rest, flag, code);
  return 0;
}

// Per-row fp8 quantisation of the activations that feed gemm_fp8.hip (engine mode FP8): row r of x[bf16, rows x K] ->
// e4m3fn codes + ONE power-of-two scale (the smallest 2^e with max|x_row| / 2^e <= 448, the rule of the FP8W weights,
// common.h), so code * scale is exact and the GEMM epilogue's rescaling is exact.  One wave per row, the row in registers
// (NV 16-byte vectors per lane), v_cvt_pk_fp8_f32 (round-to-nearest-even, OCP e4m3fn on gfx950).
template <int NV>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ q,
                                                                float* __restrict__ scale, int64_t rows) {
  constexpr int K = NV * 512;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const uint4* xr = reinterpret_cast<const uint4*>(x + r * K) + lane;
  uint4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[i * 64];
  float f[NV][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    f[i][0] = __uint_as_float(v[i].x << 16); f[i][1] = __uint_as_float(v[i].x & 0xffff0000u);
    f[i][2] = __uint_as_float(v[i].y << 16); f[i][3] = __uint_as_float(v[i].y & 0xffff0000u);
    f[i][4] = __uint_as_float(v[i].z << 16); f[i][5] = __uint_as_float(v[i].z & 0xffff0000u);
    f[i][6] = __uint_as_float(v[i].w << 16); f[i][7] = __uint_as_float(v[i].w & 0xffff0000u);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[i][j]));
  }
  amax = wave_max_dpp(amax);
  float sc = 1.f;
  if (amax > 0.f && amax < INFINITY) {
    int ex = 0;
    const float m = frexpf(amax / 448.0f, &ex);
    sc = ldexpf(1.0f, m == 0.5f ? ex - 1 : ex);
  }
  const float inv = 1.0f / sc;  // exact
  uint2* qr = reinterpret_cast<uint2*>(q + r * K) + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[i][0] * inv, f[i][1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[i][2] * inv, f[i][3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[i][4] * inv, f[i][5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[i][6] * inv, f[i][7] * inv, hi, true);
    qr[i * 64] = uint2{(unsigned)lo, (unsigned)hi};
  }
  if (lane == 0) scale[r] = sc;
}

// returns 0 = launched, 1 = width not instantiated
int launch_quantize_rows_fp8(hipStream_t st, const void* x_bf16, void* q, float* scale, int64_t rows, int K) {
  if (rows <= 0) return 0;
  if (K % 512 != 0) return 1;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define VLE_Q8(NV) hipLaunchKernelGGL((quantize_rows_fp8_kernel<NV>), grid, block, 0, st, (const bf16_t*)x_bf16, (unsigned char*)q, scale, rows)
  switch (K / 512) {
    case 1: VLE_Q8(1); break;
    case 2: VLE_Q8(2); break;
    case 3: VLE_Q8(3); break;
    case 4: VLE_Q8(4); break;
    case 6: VLE_Q8(6); break;
    case 8: VLE_Q8(8); break;
    case 12: VLE_Q8(12); break;
    case 16: VLE_Q8(16); break;
    default: return 1;
  }
#undef VLE_Q8
  return 0;
}

}  // namespace vle
