// Stand-alone kernels of the block API (seam B3) that the engine itself runs fused:
//   TokenEmbedding.forward            valle/modules/embedding.py:43-47   (row gather)
//   SinePositionalEmbedding.forward   valle/modules/embedding.py:93-97   (x * x_scale + alpha * pe[:, :T])
//   AdaptiveLayerNorm.forward         valle/modules/transformer.py:93-108 (affine fold of [w, b] = Linear(stage_emb))
// All HBM/L2-bound row kernels: one wave per row, float4 (16 B / lane) accesses.
#include "common.h"
#include "kernels.h"

namespace vle {

constexpr int BO_NW = 4;

__global__ __launch_bounds__(BO_NW * 64) void token_embedding_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                                      float* __restrict__ out, int64_t n, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * BO_NW + (threadIdx.x >> 6);
  if (r >= n) return;
  const float4* src = reinterpret_cast<const float4*>(table + ids[r] * (int64_t)d);
  float4* dst = reinterpret_cast<float4*>(out + r * (int64_t)d);
  for (int i = lane; i < (d >> 2); i += 64) dst[i] = src[i];
}

// out[r][:] += table[ids[r]][:]  -- the accumulating form: "y_emb += embedding(codes)" of the NAR prompt / stage update
// (valle/models/valle.py:1104-1113, 1134)
__global__ __launch_bounds__(BO_NW * 64) void token_embedding_add_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                                          float* __restrict__ out, int64_t n, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * BO_NW + (threadIdx.x >> 6);
  if (r >= n) return;
  const float4* src = reinterpret_cast<const float4*>(table + ids[r] * (int64_t)d);
  float4* dst = reinterpret_cast<float4*>(out + r * (int64_t)d);
  for (int i = lane; i < (d >> 2); i += 64) {
    const float4 a = dst[i], b = src[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

int launch_token_embedding_add(hipStream_t st, const int64_t* ids, const float* table, float* out, int64_t n, int d) {
  if (n <= 0) return 0;
  if (d % 4) return -1;
  hipLaunchKernelGGL(token_embedding_add_kernel, dim3((unsigned)((n + BO_NW - 1) / BO_NW)), dim3(BO_NW * 64), 0, st, ids, table, out, n, d);
  return 0;
}

int launch_token_embedding(hipStream_t st, const int64_t* ids, const float* table, float* out, int64_t n, int d) {
  if (n <= 0) return 0;
  if (d % 4) return -1;
  hipLaunchKernelGGL(token_embedding_kernel, dim3((unsigned)((n + BO_NW - 1) / BO_NW)), dim3(BO_NW * 64), 0, st, ids, table, out, n, d);
  return 0;
}

// out[b][t][:] = x[b][t][:] * x_scale + alpha * pe[t][:]   -- multiply, multiply, add: three roundings as in torch
__global__ __launch_bounds__(BO_NW * 64) void sine_positional_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                                      const float* __restrict__ alpha, float x_scale,
                                                                      float* __restrict__ out, int64_t rows, int T, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * BO_NW + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int t = (int)(r % T);
  const float a = *alpha;
  const float4* xs = reinterpret_cast<const float4*>(x + r * (int64_t)d);
  const float4* ps = reinterpret_cast<const float4*>(pe + (int64_t)t * d);
  float4* dst = reinterpret_cast<float4*>(out + r * (int64_t)d);
  for (int i = lane; i < (d >> 2); i += 64) {
    const float4 v = xs[i], p = ps[i];
    dst[i] = make_float4(__fadd_rn(__fmul_rn(v.x, x_scale), __fmul_rn(a, p.x)), __fadd_rn(__fmul_rn(v.y, x_scale), __fmul_rn(a, p.y)),
                         __fadd_rn(__fmul_rn(v.z, x_scale), __fmul_rn(a, p.z)), __fadd_rn(__fmul_rn(v.w, x_scale), __fmul_rn(a, p.w)));
  }
}

int launch_sine_positional(hipStream_t st, const float* x, const float* pe, const float* alpha, float x_scale, float* out,
                           int64_t B, int T, int d) {
  const int64_t rows = B * T;
  if (rows <= 0) return 0;
  if (d % 4) return -1;
  hipLaunchKernelGGL(sine_positional_kernel, dim3((unsigned)((rows + BO_NW - 1) / BO_NW)), dim3(BO_NW * 64), 0, st, x, pe, alpha,
                     x_scale, out, rows, T, d);
  return 0;
}

// w * (n * g + be) + b  ==  n * (w * g) + (w * be + b): the fp32 fold the engine applies per (stage, norm site)
__global__ void adaln_fold_kernel(const float* __restrict__ wb, const float* __restrict__ g, const float* __restrict__ be,
                                  float* __restrict__ gamma_out, float* __restrict__ beta_out, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  const float w = wb[i], b = wb[d + i];
  gamma_out[i] = __fmul_rn(w, g[i]);
  beta_out[i] = __fadd_rn(__fmul_rn(w, be[i]), b);
}

int launch_adaln_fold(hipStream_t st, const float* wb, const float* g, const float* be, float* gamma_out, float* beta_out, int d) {
  if (d <= 0) return -1;
  hipLaunchKernelGGL(adaln_fold_kernel, dim3((d + 255) / 256), dim3(256), 0, st, wb, g, be, gamma_out, beta_out, d);
  return 0;
}

// Cross-entropy rows of the teacher-forced forward: F.cross_entropy(logits, targets, reduction=
// "sum", ignore_index) of valle/models/valle.py:875, 936-942 and the top-k hit of its Top10Accuracy metrics (:877-879,
// 945-956).  One wave per row: max, log-sum-exp, the target's logit, and the target's rank (number of strictly larger
// logits; ties resolved towards the lower index like torch.topk).  loss[r] = lse - logit[target] (0 for an ignored or
// out-of-range target), hit[r] = 1 if the target is among the top k, -1 if the row is ignored.
__global__ __launch_bounds__(BO_NW * 64) void cross_entropy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                                    float* __restrict__ loss, int32_t* __restrict__ hit, int64_t rows,
                                                                    int V, int ignore_index, int topk) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * BO_NW + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* lg = logits + r * V;
  const int64_t t = targets[r];
  const bool ignored = t == ignore_index || t < 0 || t >= V;
  float m = -INFINITY;
  for (int i = lane; i < V; i += 64) m = fmaxf(m, lg[i]);
  m = wave_max(m);
  const float tl = ignored ? 0.f : lg[t];
  float s = 0.f;
  int above = 0;
  for (int i = lane; i < V; i += 64) {
    const float v = lg[i];
    s += expf(v - m);
    above += (v > tl) || (v == tl && i < (int)t);
  }
  s = wave_sum(s);
  above = wave_sum_i(above);
  if (lane == 0) {
    loss[r] = ignored ? 0.f : (m + logf(s)) - tl;
    hit[r] = ignored ? -1 : (above < topk ? 1 : 0);
  }
}

int launch_cross_entropy(hipStream_t st, const float* logits, const int64_t* targets, float* loss, int32_t* hit, int64_t rows, int V,
                         int ignore_index, int topk) {
  if (rows <= 0) return 0;
  if (V < 1) return -1;
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((unsigned)((rows + BO_NW - 1) / BO_NW)), dim3(BO_NW * 64), 0, st, logits, targets, loss,
                     hit, rows, V, ignore_index, topk);
  return 0;
}

// ---- cross-attention of VALL-F's TransformerDecoderLayer._mha_block (valle/modules/transformer.py:582-597 ->
// MultiheadAttention.forward(x, mem, mem), activation.py:199-431): out[t][h] = softmax(q[t][h] . K[:, h]^T / sqrt(dh)) V[:, h], no
// mask (the decode path runs one un-padded utterance: memory_key_padding_mask is all False, valle.py:604, :626-632).
// q [Tq][d], kv [S][2 d] = [K | V] rows of the memory (text) sequence, heads = contiguous dh-slices.  One wave per (query row,
// head): lane l owns elements l and l + 64 of the head (dh <= 128); per key one 64-lane DPP reduction for the score and an
// online-softmax update of the lane's two output elements.  The text side is short (S <= a few hundred) and VALL-F is not the
// production model: a plain exact kernel, fp32 arithmetic, no tiling.
template <typename T>
__global__ __launch_bounds__(256) void cross_attention_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ out, int Tq,
                                                              int S, int d, int dh) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int h = blockIdx.y;
  if (t >= Tq) return;  // wave-uniform
  const bool a0 = lane < dh, a1 = lane + 64 < dh;
  const T* qp = q + (int64_t)t * d + h * dh;
  const float scale = 1.0f / sqrtf((float)dh);
  const float q0 = a0 ? Elem<T>::to_f32(qp[lane]) * scale : 0.f, q1 = a1 ? Elem<T>::to_f32(qp[lane + 64]) * scale : 0.f;
  float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
  for (int j = 0; j < S; ++j) {
    const T* kp = kv + (int64_t)j * 2 * d + h * dh;
    float part = 0.f;
    if (a0) part = q0 * Elem<T>::to_f32(kp[lane]);
    if (a1) part = fmaf(q1, Elem<T>::to_f32(kp[lane + 64]), part);
    const float sc = wave_sum_dpp(part);  // the same value in every lane
    const float mn = fmaxf(m, sc);
    const float alpha = __expf(m - mn), p = __expf(sc - mn);  // first key: alpha = exp(-inf) = 0
    l = l * alpha + p;
    const T* vp = kp + d;
    if (a0) o0 = fmaf(p, Elem<T>::to_f32(vp[lane]), o0 * alpha);
    if (a1) o1 = fmaf(p, Elem<T>::to_f32(vp[lane + 64]), o1 * alpha);
    m = mn;
  }
  const float inv = 1.0f / l;
  T* op = out + (int64_t)t * d + h * dh;
  if (a0) store_elem<T>(op + lane, o0 * inv);
  if (a1) store_elem<T>(op + lane + 64, o1 * inv);
}

int launch_cross_attention(hipStream_t st, int dtype, const void* q, const void* kv, void* out, int Tq, int S, int d, int nhead) {
  if (Tq <= 0) return 0;
  if (S <= 0 || nhead <= 0 || d % nhead != 0 || d / nhead > 128) return -1;
  const dim3 grid((unsigned)((Tq + 3) / 4), (unsigned)nhead), block(256);
  if (dtype == DT_F32)
    hipLaunchKernelGGL((cross_attention_kernel<float>), grid, block, 0, st, (const float*)q, (const float*)kv, (float*)out, Tq, S, d, d / nhead);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL((cross_attention_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)out, Tq, S, d,
                       d / nhead);
  else
    return -1;
  return 0;
}

}  // namespace vle
