// Decode attention of the AR step: ONE new query per (utterance, head) against the KV cache.
//   reference: the last row of F.multi_head_attention_forward (valle/modules/activation.py:408-427)
//   under the prefix-LM mask of valle/models/valle.py:1019-1033 -- the new audio token sees the whole
//   text, the prompt and every earlier generated frame, i.e. every cache slot 0 .. kv_len.
//   The reference recomputes all rows every step (no KV cache, valle.py:1004); only the last row
//   is new information, and that is what this kernel produces.
//
// HBM-bound (2 * ctx * d * sizeof(T) bytes per layer per utterance) and, at batch 1, latency-bound:
// the kernel is one link of the ~60-kernel dependent chain of a decode step.  Design:
//   * cache layout [B][H][ctx_max][dh] (head-major): the keys of one head are contiguous, LPK lanes
//     share one key, a wave-load is a contiguous 1 KiB run;
//   * the KV range of a (utterance, head) is cut into fixed CHUNKs of keys dealt round-robin to the
//     NSPLIT blocks (block s owns chunks s, s + NSPLIT, ...).  The mapping does not depend on the
//     context length, so every lane requests its first chunk (NK keys of K and of V, 2*NK 16-byte
//     loads in flight) in the same burst as kv_len and q instead of after them; keys >= ctx are
//     masked afterwards (cache slots beyond ctx hold finite stale data or the zeros of allocation);
//   * scores: LPK-lane DPP reduction; softmax with one wave-wide running max, so merging the key
//     slots of a wave is a plain sum; 4 waves merge through LDS; the block writes an un-normalised
//     partial (m, l, o[dh]) that the out-proj GEMV merges in its prologue (gemv1.hip / skinny.hip
//     PRO_ATTN): part_o [B][NSPLIT][d], part_ml [B][H][NSPLIT][2].
#include "common.h"
#include "kernels.h"

namespace vle {

constexpr float DA_NEG = -1e30f;

template <int LPK>
__device__ inline float group_sum(float v) {  // sum over the LPK consecutive lanes sharing a key
  if constexpr (LPK >= 2) v += dpp_f32<0xB1>(v);
  if constexpr (LPK >= 4) v += dpp_f32<0x4E>(v);
  if constexpr (LPK >= 8) v += dpp_f32<0x141>(v);
  if constexpr (LPK >= 16) v += dpp_f32<0x140>(v);
  if constexpr (LPK >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

template <typename T, int VEC, int LPK, int NK>
__global__ __launch_bounds__(256) void decode_attn_kernel(const float* __restrict__ q, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, const int32_t* __restrict__ kv_len,
                                                          float* __restrict__ part_o, float* __restrict__ part_ml, int nhead,
                                                          int dh, int ctx_max, int nsplit, T* __restrict__ out_norm,
                                                          const int32_t* __restrict__ done, int out_xf, KTrace kt) {
  const unsigned long long kt0 = ktrace_begin(kt);
  constexpr int KPW = 64 / LPK;         // keys per wave-load
  constexpr int WCH = NK * KPW;         // keys per wave per round
  constexpr int CHUNK = 4 * WCH;        // keys per block per round
  __shared__ float sm_m[4], sm_l[4];
  __shared__ float sm_o[4][LPK * VEC];

  const int h = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int slot = lane / LPK, part = lane % LPK;
  const bool active = part * VEC < dh;
  const int d = nhead * dh;
  const T* Kb = kc + ((int64_t)b * nhead + h) * ctx_max * dh + (active ? part * VEC : 0);
  const T* Vb = vc + ((int64_t)b * nhead + h) * ctx_max * dh + (active ? part * VEC : 0);

  constexpr bool kVec = VEC * sizeof(T) == 16;
  uint4 kraw[kVec ? NK : 1], vraw[kVec ? NK : 1];  // 16-byte vectors stay raw until they are consumed
  float kf1[kVec ? 1 : NK][VEC], vf1[kVec ? 1 : NK][VEC];
  auto issue = [&](int base) {  // loads of the NK keys base + w*WCH + i*KPW + slot (clamped into the cache)
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      int key = base + w * WCH + i * KPW + slot;
      key = key < ctx_max ? key : ctx_max - 1;
      if constexpr (kVec) {
        kraw[i] = *reinterpret_cast<const uint4*>(Kb + (int64_t)key * dh);
        vraw[i] = *reinterpret_cast<const uint4*>(Vb + (int64_t)key * dh);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          kf1[i][j] = Elem<T>::to_f32(Kb[(int64_t)key * dh + j]);
          vf1[i][j] = Elem<T>::to_f32(Vb[(int64_t)key * dh + j]);
        }
      }
    }
  };
  auto widen = [&](const uint4& r, float (&f)[VEC]) {
    if constexpr (sizeof(T) == 4) {
      f[0] = __uint_as_float(r.x); f[1 % VEC] = __uint_as_float(r.y); f[2 % VEC] = __uint_as_float(r.z); f[3 % VEC] = __uint_as_float(r.w);
    } else {
      f[0] = __uint_as_float(r.x << 16); f[1 % VEC] = __uint_as_float(r.x & 0xffff0000u);
      f[2 % VEC] = __uint_as_float(r.y << 16); f[3 % VEC] = __uint_as_float(r.y & 0xffff0000u);
      f[4 % VEC] = __uint_as_float(r.z << 16); f[5 % VEC] = __uint_as_float(r.z & 0xffff0000u);
      f[6 % VEC] = __uint_as_float(r.w << 16); f[7 % VEC] = __uint_as_float(r.w & 0xffff0000u);
    }
  };

  // ---- the burst: first chunk of K/V, kv_len, q ------------------------------------------------------
  int base = s * CHUNK;
  issue(base);
  const int ctx = kv_len[b] + 1;  // the new token's K/V were just written to slot kv_len[b]
  // finished / free utterance of a batch (slot API, ragged lengths): no KV stream for it; its output row stays stale
  // and is never read (sampling skips it too).  Requested with the burst above, so a live utterance pays nothing.
  if (done != nullptr && done[b] != 0) return;
  float qv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) qv[j] = active ? q[(int64_t)b * d + h * dh + part * VEC + j] : 0.f;
  const float scale = 1.0f / sqrtf((float)dh);

  float m = DA_NEG, l = 0.f, acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;

  while (true) {
    float sc[NK];
    float mx = DA_NEG;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      float kf[VEC];
      if constexpr (kVec) widen(kraw[i], kf);
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) kf[j] = kf1[i][j];
      }
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) t = fmaf(qv[j], kf[j], t);
      t = group_sum<LPK>(active ? t : 0.f) * scale;
      const int key = base + w * WCH + i * KPW + slot;
      sc[i] = key < ctx ? t : DA_NEG;
      mx = fmaxf(mx, sc[i]);
    }
    const float mn = fmaxf(m, wave_max_dpp(mx));  // wave-uniform running max
    const float f = __expf(m - mn);               // m = DA_NEG, mn real: 0; both DA_NEG: 1 (all still 0)
    l *= f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] *= f;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int key = base + w * WCH + i * KPW + slot;
      const float p = key < ctx ? __expf(sc[i] - mn) : 0.f;
      l += p;
      float vf[VEC];
      if constexpr (kVec) widen(vraw[i], vf);
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) vf[j] = vf1[i][j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
    }
    m = mn;
    base += nsplit * CHUNK;
    if (base >= ctx) break;  // block-uniform
    issue(base);
  }

  // ---- merge the KPW key slots of the wave (same running max everywhere: plain sums) -----------------
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
  }
  if (slot == 0) {
    if (part == 0) {
      sm_m[w] = m;
      sm_l[w] = l;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm_o[w][part * VEC + j] = acc[j];
  }
  __syncthreads();
  if (tid == 0) ktrace_end(kt, kt0, ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x);  // one stamp per block (past the barrier)
  // ---- merge the 4 waves, write the partial -------------------------------------------------------------
  if (tid < dh || tid == 255) {
    const float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
    float f[4];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) f[ww] = __expf(sm_m[ww] - M);
    if (tid < dh) {
      float o = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) o = fmaf(sm_o[ww][tid], f[ww], o);
      if (out_norm != nullptr) {  // nsplit == 1: this block holds the whole softmax -> normalised T output, no merge kernel
        float L = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) L = fmaf(sm_l[ww], f[ww], L);
        if (out_xf != 0)  // the out-proj GEMM's X, fragment-major (common.h xf_index); rows = gridDim.z utterances
          store_elem<T>(out_norm + xf_index(b, h * dh + tid, ((int)gridDim.z + 15) >> 4, out_xf == 2), o / L);
        else
          store_elem<T>(out_norm + (int64_t)b * d + h * dh + tid, o / L);
      } else {
        part_o[((int64_t)b * nsplit + s) * d + h * dh + tid] = o;
      }
    } else if (out_norm == nullptr) {
      float L = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) L = fmaf(sm_l[ww], f[ww], L);
      float* ml = part_ml + (((int64_t)b * nhead + h) * nsplit + s) * 2;
      ml[0] = M;
      ml[1] = L;
    }
  }
}

template <typename T>
static int decode_dispatch(hipStream_t st, const float* q, const void* kc, const void* vc, const int32_t* kv_len, float* part_o,
                           float* part_ml, int B, int nhead, int dh, int ctx_max, int nsplit, int nk_override, void* out_norm,
                           const int32_t* done, int out_xf, KTrace kt) {
  constexpr int VFULL = Elem<T>::VEC;
  if (dh > 254) return -1;
  const dim3 grid(nhead, nsplit, B), block(256);
  // keys per lane per round: 4 by default, 8 on request (option "attn_nk")
  int lpk = 1;
  if (dh % VFULL == 0) while (lpk * VFULL < dh) lpk *= 2;
  else while (lpk < dh) lpk *= 2;
  const int keys4 = nsplit * 16 * (64 / lpk);
  (void)keys4;
  const bool nk8 = nk_override == 8;  // measured (tools/ar_tune.py, C2 batch 1): 4 keys x 2 rounds beats 8 keys x 1 round
#define VLE_DA(VEC, LPK)                                                                                                    \
  do {                                                                                                                      \
    if (nk8)                                                                                                                \
      hipLaunchKernelGGL((decode_attn_kernel<T, VEC, LPK, 8>), grid, block, 0, st, q, (const T*)kc, (const T*)vc, kv_len, part_o, \
                         part_ml, nhead, dh, ctx_max, nsplit, (T*)out_norm, done, out_xf, kt);                                                              \
    else                                                                                                                    \
      hipLaunchKernelGGL((decode_attn_kernel<T, VEC, LPK, 4>), grid, block, 0, st, q, (const T*)kc, (const T*)vc, kv_len, part_o, \
                         part_ml, nhead, dh, ctx_max, nsplit, (T*)out_norm, done, out_xf, kt);                                                              \
  } while (0)
  if (dh % VFULL == 0) {
    const int nv = dh / VFULL;
    if (nv <= 1) VLE_DA(VFULL, 1);
    else if (nv <= 2) VLE_DA(VFULL, 2);
    else if (nv <= 4) VLE_DA(VFULL, 4);
    else if (nv <= 8) VLE_DA(VFULL, 8);
    else if (nv <= 16) VLE_DA(VFULL, 16);
    else if (nv <= 32) VLE_DA(VFULL, 32);
    else return -1;
  } else {  // odd head sizes (e.g. dh = 4 in bf16): one element per lane
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
                            int nsplit, int nk_override, void* out_norm, const int32_t* done, int out_xf, KTrace kt) {
  if (B <= 0) return 0;
  if (out_xf != 0 && (out_norm == nullptr || dtype != DT_BF16 || B > 64)) return -1;
  if (out_norm != nullptr && nsplit != 1) return -1;
  if (dtype == DT_F32)
    return decode_dispatch<float>(st, q, k_cache, v_cache, kv_len, part_o, part_ml, B, nhead, dh, ctx_max, nsplit, nk_override, out_norm, done, out_xf, kt);
  return decode_dispatch<bf16_t>(st, q, k_cache, v_cache, kv_len, part_o, part_ml, B, nhead, dh, ctx_max, nsplit, nk_override, out_norm, done, out_xf, kt);
}

}  // namespace vle
