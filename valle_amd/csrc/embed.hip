// TokenEmbedding gathers fused with SinePositionalEmbedding (x + alpha * pe[pos]).
//   reference: TokenEmbedding.forward valle/modules/embedding.py:43-47,
//              SinePositionalEmbedding.forward embedding.py:93-97 (x_scale = 1),
//              call sites valle/models/valle.py:994-997, 1013-1016, 1064-1066, 1081-1083, 1110-1123.
// One wave per row; float4 (16 B / lane) loads and stores; HBM/L2-bound gathers.
#include "common.h"
#include "kernels.h"

namespace vle {

constexpr int EMB_NW = 4;

__device__ inline float4 f4_fma(float4 e, float a, float4 p) {
  // multiply then add, two roundings, exactly like `x * 1.0 + alpha * pe` in torch (no FMA contraction)
  return make_float4(__fadd_rn(e.x, __fmul_rn(a, p.x)), __fadd_rn(e.y, __fmul_rn(a, p.y)),
                     __fadd_rn(e.z, __fmul_rn(a, p.z)), __fadd_rn(e.w, __fmul_rn(a, p.w)));
}
__device__ inline float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// [text ; bos? ; prompt codebook 0] rows of the AR prefill (valle.py:994-1016)
__global__ __launch_bounds__(EMB_NW * 64) void prefill_embed_kernel(PrefillEmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * EMB_NW + (threadIdx.x >> 6);
  if (r >= a.rows) return;
  const int b = a.row_seq[r], pos = a.row_pos[r];
  const int S = a.text_len[b];
  const float* erow;
  const float* prow;
  float alpha;
  if (pos < S) {
    const int64_t id = a.text[(int64_t)b * a.s_stride + pos];
    erow = a.text_emb + id * a.d;
    prow = a.pe + (int64_t)pos * a.d;
    alpha = *a.alpha_text;
  } else {
    const int ap = pos - S;  // position in the AR audio stream (BOS included)
    int64_t tok;
    if (a.bos && ap == 0)
      tok = 1025;  // NUM_AUDIO_TOKENS + 1, valle.py:1006-1007
    else
      tok = a.prompt[((int64_t)b * a.p_stride + (ap - a.bos)) * a.Q];
    erow = a.audio_emb + tok * a.d;
    prow = a.pe + (int64_t)ap * a.d;
    alpha = *a.alpha_audio;
  }
  float4* o = reinterpret_cast<float4*>(a.x + r * a.d);
  for (int i = lane; i < (a.d >> 2); i += 64)
    o[i] = f4_fma(reinterpret_cast<const float4*>(erow)[i], alpha, reinterpret_cast<const float4*>(prow)[i]);
}

int launch_prefill_embed(hipStream_t st, const PrefillEmbedArgs& a) {
  if (a.rows <= 0) return 0;
  hipLaunchKernelGGL(prefill_embed_kernel, dim3((unsigned)((a.rows + EMB_NW - 1) / EMB_NW)), dim3(EMB_NW * 64), 0, st, a);
  return 0;
}

// y_emb = nar_audio_embeddings[0](y)  (+ sum_j>=1 embeddings[j](prompt[..., j]) on prompt rows when
// prefix_mode != 0; same left-to-right fp32 summation order as valle.py:1064-1066, 1110-1113)
__global__ __launch_bounds__(EMB_NW * 64) void nar_yemb_init_kernel(NarEmbedArgs a, int sum_all) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * EMB_NW + (threadIdx.x >> 6);
  if (r >= a.arows) return;
  const int b = a.arow_seq[r], ap = a.arow_pos[r];
  const int P = a.t.prompt_len[b];
  const bool in_prompt = ap < P;
  const int64_t* prow = a.prompt + ((int64_t)b * a.p_stride + (in_prompt ? ap : 0)) * a.Q;
  const int64_t tok0 = in_prompt ? prow[0] : a.first_cb[(int64_t)b * a.g_stride + (ap - P)];
  const float* e0 = a.audio_embs[0] + tok0 * a.d;
  float4* o = reinterpret_cast<float4*>(a.y_emb + r * a.d);
  for (int i = lane; i < (a.d >> 2); i += 64) {
    float4 v = reinterpret_cast<const float4*>(e0)[i];
    if (sum_all && in_prompt)
      for (int j = 1; j < a.Q; ++j) v = f4_add(v, reinterpret_cast<const float4*>(a.audio_embs[j] + prow[j] * a.d)[i]);
    o[i] = v;
  }
}

int launch_nar_yemb_init(hipStream_t st, const NarEmbedArgs& a, int sum_all) {
  if (a.arows <= 0) return 0;
  hipLaunchKernelGGL(nar_yemb_init_kernel, dim3((unsigned)((a.arows + EMB_NW - 1) / EMB_NW)), dim3(EMB_NW * 64), 0, st, a, sum_all);
  return 0;
}

// prefix_mode 0: y_emb[:, :P] += embeddings[j](prompt[..., j]) after stage j-1 (valle.py:1104-1107)
__global__ __launch_bounds__(EMB_NW * 64) void nar_yemb_add_prompt_kernel(NarEmbedArgs a, int j) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * EMB_NW + (threadIdx.x >> 6);
  if (r >= a.arows) return;
  const int b = a.arow_seq[r], ap = a.arow_pos[r];
  if (ap >= a.t.prompt_len[b]) return;
  const int64_t tok = a.prompt[((int64_t)b * a.p_stride + ap) * a.Q + j];
  const float4* e = reinterpret_cast<const float4*>(a.audio_embs[j] + tok * a.d);
  float4* o = reinterpret_cast<float4*>(a.y_emb + r * a.d);
  for (int i = lane; i < (a.d >> 2); i += 64) o[i] = f4_add(o[i], e[i]);
}

int launch_nar_yemb_add_prompt(hipStream_t st, const NarEmbedArgs& a, int j) {
  if (a.arows <= 0) return 0;
  hipLaunchKernelGGL(nar_yemb_add_prompt_kernel, dim3((unsigned)((a.arows + EMB_NW - 1) / EMB_NW)), dim3(EMB_NW * 64), 0, st, a, j);
  return 0;
}

// xy_pos = concat([nar_text_position(nar_text_embedding(text)), nar_audio_position(y_emb)])
// (valle.py:1081-1083, 1121-1123); text positions and audio positions both restart at 0.
__global__ __launch_bounds__(EMB_NW * 64) void nar_assemble_kernel(NarEmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * EMB_NW + (threadIdx.x >> 6);
  if (r >= a.xrows) return;
  const int b = a.xrow_seq[r], pos = a.xrow_pos[r];
  const int S = a.t.text_len[b];
  const float4* erow;
  const float4* prow;
  float alpha;
  if (pos < S) {
    // prefix_mode 2/4: text = cat(text[:1], text[enrolled_len-1:])  (valle.py:1068-1079)
    const int src = pos == 0 ? 0 : pos + a.t.text_drop[b];
    const int64_t id = a.text[(int64_t)b * a.s_stride + src];
    erow = reinterpret_cast<const float4*>(a.text_emb + id * a.d);
    prow = reinterpret_cast<const float4*>(a.pe + (int64_t)pos * a.d);
    alpha = *a.alpha_text;
  } else {
    const int ap = pos - S;
    erow = reinterpret_cast<const float4*>(a.y_emb + ((int64_t)a.t.aoff[b] + ap) * a.d);
    prow = reinterpret_cast<const float4*>(a.pe + (int64_t)ap * a.d);
    alpha = *a.alpha_audio;
  }
  float4* o = reinterpret_cast<float4*>(a.x + r * a.d);
  for (int i = lane; i < (a.d >> 2); i += 64) o[i] = f4_fma(erow[i], alpha, prow[i]);
}

int launch_nar_assemble(hipStream_t st, const NarEmbedArgs& a) {
  if (a.xrows <= 0) return 0;
  hipLaunchKernelGGL(nar_assemble_kernel, dim3((unsigned)((a.xrows + EMB_NW - 1) / EMB_NW)), dim3(EMB_NW * 64), 0, st, a);
  return 0;
}

}  // namespace vle
