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

// ---- optional tail (batched step, one block per (utterance, head)): the out-proj of the layer, fused --------------------------
// out_proj + residual (valle/modules/activation.py:421, transformer.py:297) of the new token needs all H heads of an utterance,
// but it is linear in them: every (utterance, head) block multiplies ITS normalised head output (rounded to bf16 like the
// stand-alone path's X) by its 2 dh-byte slice of every W_o row -- 128 KB from L2, shared by the B blocks of the head -- and
// publishes the d partial sums; the LAST of an utterance's H blocks (ticket; write-through stores + vmcnt drain, no fences:
// gemm_skinny.hip's split-K hand-off) adds the H partials in head order (deterministic), bias and the residual, and is the
// producer of the fused LayerNorm (bf16(x * gamma_next) fragment-major + per-16-column statistics, kernels.h LnProducer).
// One launch and one boundary less per layer; the W_o reads ride under the other blocks' KV streams.
struct AttnOproj {
  const bf16_t* w = nullptr;   // [d][d] row-major bf16 (bf16(W') in FP8W mode)
  const float* bias = nullptr; // [d]
  float* resid = nullptr;      // [B][d] fp32, updated in place
  float* part = nullptr;       // [B][H][d] fp32 partial sums
  int* cnt = nullptr;          // [B] tickets, zero between launches (self-resetting)
  int formal = 0;              // g_gs_formal at launch
  LnProducer lnp;
};

template <int LPKO>
__device__ inline void attn_oproj_tail(const AttnOproj& fo, const float* sm_on, int b, int h, int nhead, int dh, int d) {
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int RPI = 64 / LPKO;  // W_o rows per wave-load: LPKO lanes x 16 bytes cover the head's dh bf16 of one row
  const int part_l = lane % LPKO, sub = lane / LPKO;
  const bool act = part_l * 8 < dh;
  float ov[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ov[j] = act ? sm_on[part_l * 8 + j] : 0.f;
  float* pp = fo.part + ((int64_t)b * nhead + h) * d;
  const bf16_t* wb = fo.w + h * dh + (act ? part_l * 8 : 0);
  // a wave owns rows [w * d/4, (w + 1) * d/4) in chunks of 64: load i of a chunk covers rows i * RPI + sub; after the LPKO-lane
  // butterfly every lane of a group holds its row's sum and lane (part_l == i % LPKO ...) keeps it, so one 256-byte store per chunk
  const int rows_w = d >> 2;
  for (int r0 = w * rows_w; r0 < (w + 1) * rows_w; r0 += 64) {
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {  // = LPKO loads per chunk
      const int row = r0 + i * RPI + sub;
      float wf[8];
      load_vec16<bf16_t>(wb + (int64_t)row * d, wf);
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t = fmaf(wf[j], ov[j], t);
      t = group_sum<LPKO>(act ? t : 0.f);
      keep = part_l == i ? t : keep;  // lane (sub, part_l = i) keeps row r0 + i * RPI + sub
    }
    __hip_atomic_store(pp + r0 + part_l * RPI + sub, keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (fo.formal) {  // "gs_formal": explicit release / acquire around the ticket (gemm_skinny.hip explains both forms)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int t = __hip_atomic_fetch_add(fo.cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fo.formal) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_last = t == nhead - 1;
    if (t == nhead - 1) __hip_atomic_store(fo.cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-reset (graph replay)
  }
  __syncthreads();
  if (!s_last) return;
  // ---- last block of the utterance: sum the heads, finish the residual row, feed the next LayerNorm -------------------------------
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  for (int c = tid * 4; c < d; c += 1024) {
    f32x4_t y = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int hh = 0; hh < nhead; ++hh) {  // fixed order: the sum does not depend on which head finished last
      const float* ph = fo.part + ((int64_t)b * nhead + hh) * d + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] += __hip_atomic_load(ph + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const f32x4_t bias4 = fo.bias ? *reinterpret_cast<const f32x4_t*>(fo.bias + c) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    float* o = fo.resid + (int64_t)b * d + c;
    const f32x4_t x4 = *reinterpret_cast<const f32x4_t*>(o) + (y + bias4);
    *reinterpret_cast<f32x4_t*>(o) = x4;
    if (fo.lnp.gamma != nullptr) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(fo.lnp.gamma + c);
      bf16x4_t o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (__bf16)(x4[r] * g4[r]);
      *reinterpret_cast<bf16x4_t*>(reinterpret_cast<bf16_t*>(fo.lnp.xg_out) + xf_index(b, c, fo.lnp.MF, fo.lnp.w8 != 0)) = o4;
      // the 4 consecutive threads tid % 4 = 0..3 hold one 16-column group (d % 16 == 0: they are active together)
      float sg = (x4[0] + x4[1]) + (x4[2] + x4[3]);
      sg += dpp_f32<0xB1>(sg);
      sg += dpp_f32<0x4E>(sg);
      const float mean = sg * (1.0f / 16.0f);
      float qg = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = x4[r] - mean;
        qg = fmaf(t, t, qg);
      }
      qg += dpp_f32<0xB1>(qg);
      qg += dpp_f32<0x4E>(qg);
      if ((tid & 3) == 0) *reinterpret_cast<f32x2_t*>(fo.lnp.stats_out + ((int64_t)b * (d >> 4) + (c >> 4)) * 2) = f32x2_t{mean, qg};
    }
  }
}

template <typename T, int VEC, int LPK, int NK, bool FO = false, bool NT = false>
__global__ __launch_bounds__(256) void decode_attn_kernel(const float* __restrict__ q, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, const int32_t* __restrict__ kv_len,
                                                          float* __restrict__ part_o, float* __restrict__ part_ml, int nhead,
                                                          int dh, int ctx_max, int nsplit, T* __restrict__ out_norm,
                                                          const int32_t* __restrict__ done, int out_xf, KTrace kt,
                                                          AttnOproj fo = AttnOproj()) {
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
  // Lanes whose key lies beyond the context are masked afterwards; what they load is clamped into the cache.  The streaming variant
  // (NT: many utterances, HBM-bound) reads the context length FIRST and clamps to the last valid row, so those lanes re-read a line
  // the wave fetches anyway -- at 64 utterances the rows past the context were 10 % of the kernel's HBM traffic (r04 PMC: 186.8 MB
  // per launch against 170 MB of valid rows).  The batch-1 variant keeps the length in the same burst as the first keys (latency).
  int key_hi = ctx_max - 1;
  if constexpr (NT) {  // (a free / finished slot's length is whatever it last held: keep the clamp inside the cache whatever it says)
    const int kl = kv_len[b];
    key_hi = kl < 0 ? 0 : (kl < ctx_max ? kl : ctx_max - 1);
  }
  auto issue = [&](int base) {  // loads of the NK keys base + w*WCH + i*KPW + slot (clamped into the cache)
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      int key = base + w * WCH + i * KPW + slot;
      key = key < key_hi ? key : key_hi;
      if constexpr (kVec) {
        if constexpr (NT) {  // a KV stream far larger than the 256 MB memory-side cache: read once per step, do not allocate
          typedef unsigned int da_u32x4 __attribute__((ext_vector_type(4)));
          const da_u32x4 kk = __builtin_nontemporal_load(reinterpret_cast<const da_u32x4*>(Kb + (int64_t)key * dh));
          const da_u32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const da_u32x4*>(Vb + (int64_t)key * dh));
          kraw[i] = uint4{kk.x, kk.y, kk.z, kk.w};
          vraw[i] = uint4{vv.x, vv.y, vv.z, vv.w};
        } else {
          kraw[i] = *reinterpret_cast<const uint4*>(Kb + (int64_t)key * dh);
          vraw[i] = *reinterpret_cast<const uint4*>(Vb + (int64_t)key * dh);
        }
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
  if constexpr (FO) {
    __shared__ float sm_on[LPK * VEC];
    if (tid < dh) {
      const float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
      float o = 0.f, L = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const float f = __expf(sm_m[ww] - M);
        o = fmaf(sm_o[ww][tid], f, o);
        L = fmaf(sm_l[ww], f, L);
      }
      sm_on[tid] = bf16_to_f32(f32_to_bf16(o / L));  // the value the stand-alone out-proj GEMM reads as its bf16 X
    }
    __syncthreads();
    attn_oproj_tail<LPK>(fo, sm_on, b, h, nhead, dh, d);
    return;
  }
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

// "attn_lds_pad": dynamic LDS bytes requested per workgroup of the batched decode attention (never touched): caps the workgroups
// a CU can host (160 KB / pad), so that the B * H workgroups of a launch spread evenly over the CUs instead of wherever the
// dispatcher finds room first (the launch is one wave of equal workgroups: its time is the fullest CU's)
int g_da_lds_pad = 0;
int g_da_nt = -1;  // "attn_nt" (see decode_dispatch)

template <typename T>
static int decode_dispatch(hipStream_t st, const float* q, const void* kc, const void* vc, const int32_t* kv_len, float* part_o,
                           float* part_ml, int B, int nhead, int dh, int ctx_max, int nsplit, int nk_override, void* out_norm,
                           const int32_t* done, int out_xf, KTrace kt, int kv_nt) {
  constexpr int VFULL = Elem<T>::VEC;
  if (dh > 254) return -1;
  const dim3 grid(nhead, nsplit, B), block(256);
  // keys per lane per round: 4 by default, 8 on request (option "attn_nk")
  int lpk = 1;
  if (dh % VFULL == 0) while (lpk * VFULL < dh) lpk *= 2;
  else while (lpk < dh) lpk *= 2;
  const int keys4 = nsplit * 16 * (64 / lpk);
  (void)keys4;
  const bool nk8 = nk_override == 8;
  const unsigned lds_pad = (B > 1 && g_da_lds_pad > 0) ? (unsigned)g_da_lds_pad : 0u;
  // Non-temporal K / V loads when the step's whole KV stream (all layers) exceeds the 256 MB memory-side cache: every byte is read
  // once per step and nothing survives to the next one, so allocating it only evicts what could stay.  Measured (MI355X, C2
  // architecture): 64 utterances AR loop 573 -> 536 ms (decode attention ~5.2 -> ~6 TB/s), 8 utterances 291 -> 283 ms; one
  // utterance (50 MB of KV per step) keeps the default policy -- its cache stays resident.  The caller knows the layer count and
  // passes kv_nt (1 / 0); -1 = decide from this layer's size alone; knob "attn_nt" (g_da_nt >= 0) overrides both.
  const int64_t layer_kv = (int64_t)2 * B * nhead * ctx_max * dh * (int64_t)sizeof(T);
  const bool nt_stream = g_da_nt >= 0 ? g_da_nt != 0 : kv_nt >= 0 ? kv_nt != 0 : layer_kv > ((int64_t)64 << 20);  // measured (tools/ar_tune.py, C2 batch 1): 4 keys x 2 rounds beats 8 keys x 1 round
#define VLE_DA(VEC, LPK)                                                                                                    \
  do {                                                                                                                      \
    if (nt_stream && !nk8)                                                                                                  \
      hipLaunchKernelGGL((decode_attn_kernel<T, VEC, LPK, 4, false, true>), grid, block, lds_pad, st, q, (const T*)kc, (const T*)vc, kv_len, part_o, \
                         part_ml, nhead, dh, ctx_max, nsplit, (T*)out_norm, done, out_xf, kt, AttnOproj());                     \
    else if (nk8)                                                                                                           \
      hipLaunchKernelGGL((decode_attn_kernel<T, VEC, LPK, 8>), grid, block, lds_pad, st, q, (const T*)kc, (const T*)vc, kv_len, part_o, \
                         part_ml, nhead, dh, ctx_max, nsplit, (T*)out_norm, done, out_xf, kt);                                                              \
    else                                                                                                                    \
      hipLaunchKernelGGL((decode_attn_kernel<T, VEC, LPK, 4>), grid, block, lds_pad, st, q, (const T*)kc, (const T*)vc, kv_len, part_o, \
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
                            int nsplit, int nk_override, void* out_norm, const int32_t* done, int out_xf, KTrace kt, int kv_nt) {
  if (B <= 0) return 0;
  if (out_xf != 0 && (out_norm == nullptr || dtype != DT_BF16 || B > 64)) return -1;
  if (out_norm != nullptr && nsplit != 1) return -1;
  if (dtype == DT_F32)
    return decode_dispatch<float>(st, q, k_cache, v_cache, kv_len, part_o, part_ml, B, nhead, dh, ctx_max, nsplit, nk_override, out_norm, done, out_xf, kt, kv_nt);
  return decode_dispatch<bf16_t>(st, q, k_cache, v_cache, kv_len, part_o, part_ml, B, nhead, dh, ctx_max, nsplit, nk_override, out_norm, done, out_xf, kt, kv_nt);
}

// decode attention + out-proj + residual (+ LayerNorm producer) of the batched step in ONE launch: bf16 cache, one block per
// (utterance, head) (nsplit = 1), dh % 8 == 0, dh in {32, 64, 128} (a power-of-two group of lanes covers the head), d % 256 == 0.
// `part` [B][nhead][d] fp32, `cnt` [B] zeroed ints.  Returns 1 when the shape is not covered.
int launch_decode_attention_oproj(hipStream_t st, const float* q, const void* k_cache, const void* v_cache, const int32_t* kv_len, int B,
                                  int nhead, int dh, int ctx_max, const int32_t* done, const void* wo_bf16, const float* bias, float* resid,
                                  float* part, int* cnt, const LnProducer& lnp, KTrace kt) {
  if (B <= 0) return 0;
  const int d = nhead * dh;
  if (!(dh == 32 || dh == 64 || dh == 128) || d % 256 != 0 || !wo_bf16 || !resid || !part || !cnt) return 1;
  AttnOproj fo;
  fo.w = (const bf16_t*)wo_bf16; fo.bias = bias; fo.resid = resid; fo.part = part; fo.cnt = cnt; fo.lnp = lnp; fo.formal = g_gs_formal;
  const dim3 grid(nhead, 1, B), block(256);
#define VLE_DAO(LPK)                                                                                                            \
  hipLaunchKernelGGL((decode_attn_kernel<bf16_t, 8, LPK, 4, true>), grid, block, 0, st, q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, \
                     kv_len, (float*)nullptr, (float*)nullptr, nhead, dh, ctx_max, 1, (bf16_t*)nullptr, done, 0, kt, fo)
  if (dh == 32) VLE_DAO(4);
  else if (dh == 64) VLE_DAO(8);
  else VLE_DAO(16);
#undef VLE_DAO
  return 0;
}

}  // namespace vle
