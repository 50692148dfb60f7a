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
// (2) decode_attention_kernel -- ONE new query per (utterance, head) against the KV cache, the
//     HBM-bound part of the AR step (2*ctx*d*sizeof(T) bytes per layer per utterance).  Keys are
//     split over NSPLIT blocks x 4 waves so a batch-1 step still spreads its KV stream over many
//     CUs; LPK lanes share one key so that every wave-load is a contiguous 1 KiB run of the
//     head-major cache ([B][H][ctx][dh]); per-lane-group online softmax, merged by shuffles, then
//     through LDS; the un-normalised partial (m, l, o[dh]) goes to HBM and is merged in the
//     out-proj kernel's prologue (skinny.hip PRO_ATTN).
#include "common.h"
#include "kernels.h"

namespace vle {

// ------------------------------------------------------------------------------------------------
// (1) prefill / NAR attention
// ------------------------------------------------------------------------------------------------
constexpr int AT_QB = 64;  // query rows per block
constexpr int AT_KB = 32;  // keys per LDS tile

template <typename T, int DPT>
__global__ __launch_bounds__(256) void attention_rows_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                             const int32_t* __restrict__ seq_off,
                                                             const int32_t* __restrict__ text_len, int d, int nhead,
                                                             int causal) {
  constexpr int DH = DPT * 4;
  __shared__ __attribute__((aligned(16))) float Ks[AT_KB * DH];
  __shared__ __attribute__((aligned(16))) float Vs[AT_KB * DH];

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

  for (int kt0 = 0; kt0 < kmax; kt0 += AT_KB) {
    __syncthreads();
    for (int idx = tid; idx < AT_KB * DH; idx += 256) {
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

template <typename T>
static int attention_dispatch(hipStream_t st, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len,
                              int B, int max_len, int d, int nhead, int causal) {
  const int dh = d / nhead;
  const dim3 grid((max_len + AT_QB - 1) / AT_QB, nhead, B), block(256);
#define VLE_AT(DPT)                                                                                                 \
  hipLaunchKernelGGL((attention_rows_kernel<T, DPT>), grid, block, 0, st, (const T*)qkv, (T*)out, seq_off, text_len, d, \
                     nhead, causal)
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

// ------------------------------------------------------------------------------------------------
// (2) decode attention (one query per utterance and head)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPK>
__global__ __launch_bounds__(256) void decode_attention_kernel(const float* __restrict__ q, const T* __restrict__ kc,
                                                               const T* __restrict__ vc, const int32_t* __restrict__ kv_len,
                                                               float* __restrict__ part_o, float* __restrict__ part_ml,
                                                               int nhead, int dh, int ctx_max, int nsplit) {
  constexpr int KPW = 64 / LPK;  // keys per wave-load
  __shared__ float sm_ml[4][2];
  __shared__ float sm_o[4][LPK * VEC];

  const int h = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int slot = lane / LPK, part = lane % LPK;
  const bool active = part * VEC < dh;
  const int ctx = kv_len[b] + 1;  // the new token's K/V were just written to slot kv_len[b]
  const int C = (ctx + nsplit - 1) / nsplit;
  const int kbeg = s * C, kend = min(ctx, kbeg + C);
  const int d = nhead * dh;
  const float scale = 1.0f / sqrtf((float)dh);

  float qv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) qv[j] = active ? q[(int64_t)b * d + h * dh + part * VEC + j] * scale : 0.f;

  const T* Kb = kc + ((int64_t)b * nhead + h) * ctx_max * dh;
  const T* Vb = vc + ((int64_t)b * nhead + h) * ctx_max * dh;

  float m = -1e30f, l = 0.f, acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;

  auto load_kv = [&](int key, float (&kf)[VEC], float (&vf)[VEC]) {
    if (key < kend && active) {
      if constexpr (VEC * sizeof(T) == 16) {
        load_vec16<T>(Kb + (int64_t)key * dh + part * VEC, kf);
        load_vec16<T>(Vb + (int64_t)key * dh + part * VEC, vf);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          kf[j] = Elem<T>::to_f32(Kb[(int64_t)key * dh + part * VEC + j]);
          vf[j] = Elem<T>::to_f32(Vb[(int64_t)key * dh + part * VEC + j]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) kf[j] = vf[j] = 0.f;
    }
  };
  auto update = [&](int key, const float (&kf)[VEC], const float (&vf)[VEC]) {
    float sc = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) sc = fmaf(qv[j], kf[j], sc);
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) sc += __shfl_xor(sc, o, 64);
    if (key < kend) {
      const float mn = fmaxf(m, sc);
      const float f = expf(m - mn), p = expf(sc - mn);
      l = l * f + p;
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = acc[j] * f + p * vf[j];
      m = mn;
    }
  };
  // two key groups per iteration: 4 independent 16-byte loads in flight per lane
  for (int base = kbeg + w * KPW; base < kend; base += 8 * KPW) {
    const int key0 = base + slot, key1 = base + 4 * KPW + slot;
    float kf0[VEC], vf0[VEC], kf1[VEC], vf1[VEC];
    load_kv(key0, kf0, vf0);
    load_kv(key1, kf1, vf1);
    update(key0, kf0, vf0);
    update(key1, kf1, vf1);
  }
  // merge the KPW key slots of the wave (lanes with the same `part`)
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    const float f1 = expf(m - mn), f2 = expf(m2 - mn);
    l = l * f1 + l2 * f2;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = acc[j] * f1 + __shfl_xor(acc[j], o, 64) * f2;
    m = mn;
  }
  if (slot == 0) {
    if (part == 0) {
      sm_ml[w][0] = m;
      sm_ml[w][1] = l;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm_o[w][part * VEC + j] = acc[j];
  }
  __syncthreads();
  if (w == 0 && slot == 0) {
    float M = fmaxf(fmaxf(sm_ml[0][0], sm_ml[1][0]), fmaxf(sm_ml[2][0], sm_ml[3][0]));
    float L = 0.f, o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const float f = expf(sm_ml[ww][0] - M);
      L += sm_ml[ww][1] * f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += sm_o[ww][part * VEC + j] * f;
    }
    const int64_t pidx = ((int64_t)b * nhead + h) * nsplit + s;
    if (part == 0) {
      part_ml[pidx * 2] = M;
      part_ml[pidx * 2 + 1] = L;
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) part_o[pidx * dh + part * VEC + j] = o[j];
    }
  }
}

template <typename T>
static int decode_dispatch(hipStream_t st, const float* q, const void* kc, const void* vc, const int32_t* kv_len, float* part_o,
                           float* part_ml, int B, int nhead, int dh, int ctx_max, int nsplit) {
  constexpr int VFULL = Elem<T>::VEC;
  const dim3 grid(nhead, nsplit, B), block(256);
#define VLE_DA(VEC, LPK)                                                                                              \
  hipLaunchKernelGGL((decode_attention_kernel<T, VEC, LPK>), grid, block, 0, st, q, (const T*)kc, (const T*)vc, kv_len, \
                     part_o, part_ml, nhead, dh, ctx_max, nsplit)
  if (dh % VFULL == 0) {
    const int nv = dh / VFULL;
    if (nv <= 1) VLE_DA(VFULL, 1);
    else if (nv <= 2) VLE_DA(VFULL, 2);
    else if (nv <= 4) VLE_DA(VFULL, 4);
    else if (nv <= 8) VLE_DA(VFULL, 8);
    else if (nv <= 16) VLE_DA(VFULL, 16);
    else if (nv <= 32) VLE_DA(VFULL, 32);
    else return -1;
  } else {  // odd head sizes (e.g. dh = 4 in bf16): scalar element per lane
    if (dh <= 1) VLE_DA(1, 1);
    else if (dh <= 2) VLE_DA(1, 2);
    else if (dh <= 4) VLE_DA(1, 4);
    else if (dh <= 8) VLE_DA(1, 8);
    else if (dh <= 16) VLE_DA(1, 16);
    else if (dh <= 32) VLE_DA(1, 32);
    else return -1;
  }
#undef VLE_DA
  return 0;
}

int launch_decode_attention(hipStream_t st, int dtype, const float* q, const void* k_cache, const void* v_cache,
                            const int32_t* kv_len, float* part_o, float* part_ml, int B, int nhead, int dh, int ctx_max,
                            int nsplit) {
  if (B <= 0) return 0;
  if (dtype == DT_F32) return decode_dispatch<float>(st, q, k_cache, v_cache, kv_len, part_o, part_ml, B, nhead, dh, ctx_max, nsplit);
  return decode_dispatch<bf16_t>(st, q, k_cache, v_cache, kv_len, part_o, part_ml, B, nhead, dh, ctx_max, nsplit);
}

}  // namespace vle
