// Attention kernels.
//   reference: F.multi_head_attention_forward as called from MultiheadAttention.forward
//   (valle/modules/activation.py:408-427): heads = contiguous dh slices of d, softmax(QK^T/sqrt(dh)+mask)V;
//   AR mask (valle/models/valle.py:1019-1033): text rows see text only, audio rows see text + causal
//   audio; NAR: no mask.
//
// (1) attention_rows_kernel  -- prefill / NAR over packed sequences, generic in dtype and head size
//     (dh = 4*DPT): 64 query rows x 4 lanes per row per block, K/V tiles staged once per block in LDS
//     as fp32 and shared by the 64 rows, online softmax in registers.  fp32 VALU math: this is the
//     exact-mode path and the generic fallback.
// (2) the decode attention of the AR step lives in decode_attn.hip.
#include "common.h"
#include "kernels.h"

namespace vle {

// ------------------------------------------------------------------------------------------------
// (1) prefill / NAR attention
// ------------------------------------------------------------------------------------------------
constexpr int AT_KB = 32;  // keys per LDS tile

// 64 query rows per block, 4 lanes each.  (Measured and dropped, round 6: 32 rows per block -- two blocks per CU at one utterance -- fp32 NAR
// 49.1 -> 51.5 ms with the vector staging, 55.2 -> 59.9 without: every block stages all keys, twice the K / V traffic.)
template <typename T, int DPT, bool VS = true>  // VS: vector staging (below; fp32 only)
__global__ __launch_bounds__(256) void attention_rows_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                             const int32_t* __restrict__ seq_off,
                                                             const int32_t* __restrict__ text_len, int d, int nhead,
                                                             int causal) {
  constexpr int DH = DPT * 4;
  __shared__ __attribute__((aligned(16))) float Ks[AT_KB * DH];
  __shared__ __attribute__((aligned(16))) float Vs[AT_KB * DH];

  constexpr int AT_QB = 64, NT = 256;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AT_QB;
  const int off = seq_off[b], len = seq_off[b + 1] - off;
  if (q0 >= len) return;
  const int S = text_len[b];
  const int tid = threadIdx.x, sub = tid & 3, rloc = tid >> 2;
  const int row = q0 + rloc;
  const bool valid = row < len;
  // keys visible to this row: j < klim
  const int klim = !valid ? 0 : (causal ? max(S, row + 1) : len);
  const int qend = min(q0 + AT_QB, len);
  const int kmax = causal ? max(S, qend) : len;  // block-wide bound
  const int d3 = 3 * d;
  const float scale = 1.0f / sqrtf((float)DH);

  float q[DPT], acc[DPT];
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    q[j] = valid ? Elem<T>::to_f32(qkv[(int64_t)(off + row) * d3 + h * DH + sub * DPT + j]) * scale : 0.f;
    acc[j] = 0.f;
  }
  float m = -1e30f, l = 0.f;

  // Staging (round 6): fp32 tiles are fetched as 16-byte vectors, the NEXT tile's vectors are requested before this tile's arithmetic
  // and written to LDS after it (register double buffer): the kernel used to fetch 4-byte elements through an integer division and wait
  // for them between two barriers, once per 32 keys.  What a row computes, and in which order, is untouched (same bits).
  constexpr bool VEC = VS && sizeof(T) == 4 && (AT_KB * DH / 4) % NT == 0;
  constexpr int NV = VEC ? AT_KB * DH / 4 / NT : 1;  // float4 of K (and of V) per thread and tile
  typedef float at_f32x4 __attribute__((ext_vector_type(4)));
  at_f32x4 pk[NV], pv[NV];
  auto fetch = [&](int kt0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx4 = tid + v * NT, kk = idx4 / (DH / 4), e = (idx4 - kk * (DH / 4)) * 4;
      const int key = kt0 + kk;
      const float* base = reinterpret_cast<const float*>(qkv) + (int64_t)(off + (key < len ? key : len - 1)) * d3 + h * DH + e;
      const at_f32x4 k4 = *reinterpret_cast<const at_f32x4*>(base + d), v4 = *reinterpret_cast<const at_f32x4*>(base + 2 * d);
      pk[v] = key < len ? k4 : at_f32x4{0.f, 0.f, 0.f, 0.f};
      pv[v] = key < len ? v4 : at_f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx4 = tid + v * NT;
      *reinterpret_cast<at_f32x4*>(&Ks[idx4 * 4]) = pk[v];
      *reinterpret_cast<at_f32x4*>(&Vs[idx4 * 4]) = pv[v];
    }
  };
  if constexpr (VEC) {
    if (kmax > 0) fetch(0);
  }
  for (int kt0 = 0; kt0 < kmax; kt0 += AT_KB) {
    __syncthreads();
    if constexpr (VEC) {
      stash();
      if (kt0 + AT_KB < kmax) fetch(kt0 + AT_KB);
    } else {
    for (int idx = tid; idx < AT_KB * DH; idx += NT) {
      const int kk = idx / DH, e = idx - kk * DH;
      const int key = kt0 + kk;
      float kv = 0.f, vv = 0.f;
      if (key < len) {
        const T* base = qkv + (int64_t)(off + key) * d3 + h * DH + e;
        kv = Elem<T>::to_f32(base[d]);
        vv = Elem<T>::to_f32(base[2 * d]);
      }
      Ks[idx] = kv;
      Vs[idx] = vv;
    }
    }
    __syncthreads();
    float sc[AT_KB];
    float tmax = -1e30f;
#pragma unroll
    for (int kk = 0; kk < AT_KB; ++kk) {
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < DPT; ++j) part = fmaf(q[j], Ks[kk * DH + sub * DPT + j], part);
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      sc[kk] = (kt0 + kk < klim) ? part : -INFINITY;
      tmax = fmaxf(tmax, sc[kk]);
    }
    const float mn = fmaxf(m, tmax);
    const float f = expf(m - mn);
    l *= f;
#pragma unroll
    for (int j = 0; j < DPT; ++j) acc[j] *= f;
#pragma unroll
    for (int kk = 0; kk < AT_KB; ++kk) {
      const float p = expf(sc[kk] - mn);  // exp(-inf) = 0 for masked keys
      l += p;
#pragma unroll
      for (int j = 0; j < DPT; ++j) acc[j] = fmaf(p, Vs[kk * DH + sub * DPT + j], acc[j]);
    }
    m = mn;
  }
  if (valid) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < DPT; ++j) store_elem<T>(out + (int64_t)(off + row) * d + h * DH + sub * DPT + j, acc[j] * inv);
  }
}

int g_attn_f32_vec = 1;  // "attn_f32_vec": 16-byte, register-double-buffered K / V staging of the fp32 instantiations (0: round 1's element-wise staging)

template <typename T>
static int attention_dispatch(hipStream_t st, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len,
                              int B, int max_len, int d, int nhead, int causal) {
  const int dh = d / nhead;
  const dim3 grid((max_len + 63) / 64, nhead, B), block(256);
#define VLE_AT(DPT)                                                                                                     \
  do {                                                                                                                  \
    if (g_attn_f32_vec) hipLaunchKernelGGL((attention_rows_kernel<T, DPT, true>), grid, block, 0, st, (const T*)qkv, (T*)out, seq_off, text_len, d, nhead, causal); \
    else hipLaunchKernelGGL((attention_rows_kernel<T, DPT, false>), grid, block, 0, st, (const T*)qkv, (T*)out, seq_off, text_len, d, nhead, causal); \
  } while (0)
  switch (dh) {
    case 4: VLE_AT(1); break;
    case 8: VLE_AT(2); break;
    case 16: VLE_AT(4); break;
    case 32: VLE_AT(8); break;
    case 64: VLE_AT(16); break;
    case 96: VLE_AT(24); break;
    case 128: VLE_AT(32); break;
    default: return -1;
  }
#undef VLE_AT
  return 0;
}

int launch_attention(hipStream_t st, int dtype, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len,
                     int B, int max_len, int d, int nhead, int causal) {
  if (B <= 0 || max_len <= 0) return 0;
  if (dtype == DT_F32) return attention_dispatch<float>(st, qkv, out, seq_off, text_len, B, max_len, d, nhead, causal);
  // bf16: the MFMA flash kernels when they have the head size (attn_mfma2.hip, else round 1's attn_mfma.hip)
  {
    const int r2 = launch_attention_mfma2(st, qkv, out, seq_off, text_len, B, max_len, (int64_t)B * max_len, d, nhead, causal);
    if (r2 <= 0) return r2;
  }
  if (launch_attention_mfma(st, qkv, out, seq_off, text_len, B, max_len, d, nhead, causal) == 0) return 0;
  return attention_dispatch<bf16_t>(st, qkv, out, seq_off, text_len, B, max_len, d, nhead, causal);
}

// K/V of the packed prefill rows -> head-major cache [B][H][ctx_max][dh]
template <typename T>
__global__ __launch_bounds__(256) void kv_scatter_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                         const int32_t* __restrict__ row_seq,
                                                         const int32_t* __restrict__ row_pos, int d, int nhead, int ctx_max) {
  const int64_t r = blockIdx.x;
  const int b = row_seq[r], pos = row_pos[r];
  const int dh = d / nhead;
  const T* src = qkv + r * 3 * d;
  for (int j = threadIdx.x; j < d; j += 256) {
    const int h = j / dh, e = j - h * dh;
    const int64_t o = (((int64_t)b * nhead + h) * ctx_max + pos) * dh + e;
    kc[o] = src[d + j];
    vc[o] = src[2 * d + j];
  }
}

int launch_kv_scatter(hipStream_t st, int dtype, const void* qkv, void* k_cache, void* v_cache, const int32_t* row_seq,
                      const int32_t* row_pos, int64_t rows, int d, int nhead, int ctx_max) {
  if (rows <= 0) return 0;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(kv_scatter_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)qkv, (float*)k_cache,
                       (float*)v_cache, row_seq, row_pos, d, nhead, ctx_max);
  else
    hipLaunchKernelGGL(kv_scatter_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)qkv,
                       (bf16_t*)k_cache, (bf16_t*)v_cache, row_seq, row_pos, d, nhead, ctx_max);
  return 0;
}

}  // namespace vle
