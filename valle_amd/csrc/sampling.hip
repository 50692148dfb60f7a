// AR sampling + stop rule + next-token embedding, and the NAR arg-max / y_emb update.
//   reference: topk_sampling valle/models/valle.py:1287-1302, top_k_top_p_filtering :1242-1284,
//              stop rule :1044-1055, y = concat([y, samples]) :1057, NAR argmax/update :1128-1134.
// Everything stays on the device: per-utterance done flags / lengths live in HBM so the AR step can
// be replayed from a hipGraph with no host round trip (the reference syncs 2-3x per step).
#include "common.h"
#include "kernels.h"
#include "sampling_dev.h"

namespace vle {

// ---- stand-alone operator: one block per row of logits[rows][V] (V <= SAMP_T * SAMP_PER): out[row] = topk_sampling draw --------
__global__ __launch_bounds__(SAMP_T) void topk_sample_rows_kernel(const float* __restrict__ logits, int V, int top_k, float temperature,
                                                                  unsigned long long seed, unsigned it, int64_t* __restrict__ out,
                                                                  int64_t* __restrict__ argmax_out) {
  __shared__ unsigned long long red64[SAMP_T / 64];
  __shared__ int redi[SAMP_T / 64];
  __shared__ float redf[SAMP_T / 64];
  __shared__ float wave_tot[SAMP_T / 64];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* lg = logits + (int64_t)row * V;
  float raw[SAMP_PER];
#pragma unroll
  for (int j = 0; j < SAMP_PER; ++j) {
    const int idx = tid * SAMP_PER + j;
    raw[j] = idx < V ? lg[idx] : -INFINITY;
  }
  const int am = argmax_row(raw, V, red64);
  const int smp = sample_row(raw, V, top_k, temperature, request_seed(seed, (unsigned long long)row), it, am, SampScratch{red64, redi, redf, wave_tot});
  if (tid == 0) {
    out[row] = smp;
    if (argmax_out) argmax_out[row] = am;
  }
}

int launch_topk_sample_rows(hipStream_t st, const float* logits, int64_t rows, int V, int top_k, float temperature, unsigned long long seed,
                            unsigned it, int64_t* out, int64_t* argmax_out) {
  if (rows <= 0) return 0;
  if (V <= 0 || V > SAMP_T * SAMP_PER || rows > 0x7fffffff) return -2;
  hipLaunchKernelGGL(topk_sample_rows_kernel, dim3((unsigned)rows), dim3(SAMP_T), 0, st, logits, V, top_k, temperature, seed, it, out, argmax_out);
  return 0;
}

__global__ __launch_bounds__(SAMP_T) void ar_sample_kernel(ArSampleArgs a) {
  __shared__ unsigned long long red64[SAMP_T / 64];
  __shared__ int redi[SAMP_T / 64];
  __shared__ float redf[SAMP_T / 64];
  __shared__ float wave_tot[SAMP_T / 64];
  __shared__ int sh_next, sh_stop, sh_pos;

  const int b = a.slot_map ? a.slot_map[blockIdx.x] : (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (a.host_prog != nullptr && blockIdx.x == 0 && tid == 0) {
    // done_count is complete as of the previous launch (kernel boundary); [1] is a launch counter only this thread writes
    const int dc = a.s.done_count[0], sc = a.s.done_count[1] + 1;
    a.s.done_count[1] = sc;
    __hip_atomic_store(a.host_prog + 0, dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.host_prog + 1, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (a.s.done[b]) return;
  const unsigned long long kt0 = ktrace_begin(a.kt);
  const int it = a.s.iter[b];
  const int V = a.V;
  const ArDyn dyn = *a.dyn;
  const float* lg = a.logits + (int64_t)b * V;

  float raw[SAMP_PER];
  unsigned long long best = 0ull;
#pragma unroll
  for (int j = 0; j < SAMP_PER; ++j) {
    const int idx = tid * SAMP_PER + j;
    raw[j] = idx < V ? lg[idx] : -INFINITY;
    if (idx < V) {
      // highest value wins; among equal values the LOWEST index wins (torch.argmax convention)
      const unsigned long long key = ((unsigned long long)float_key(raw[j]) << 32) | (unsigned)(0x7fffffff - idx);
      best = key > best ? key : best;
    }
  }
  if (dyn.trace != nullptr && it < dyn.trace_cap) {
    float* tr = dyn.trace + ((int64_t)it * a.B + b) * V;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) {
      const int idx = tid * SAMP_PER + j;
      if (idx < V) tr[idx] = raw[j];
    }
  }
  best = block_max_u64(best, red64);
  const int argmax = 0x7fffffff - (int)(best & 0xffffffffu);

  const unsigned long long rseed = a.slot_seed != nullptr ? a.slot_seed[b] : request_seed(dyn.seed, (unsigned long long)b);
  const int sample = sample_row(raw, V, dyn.top_k, dyn.temperature, rseed, (uint32_t)it, argmax, SampScratch{red64, redi, redf, wave_tot});

  if (tid == 0) {
    ktrace_end(a.kt, kt0, (int)blockIdx.x);  // before a.s.iter moves on: the stamp lands in this step's slot
    const int n = a.s.n_gen[b];
    const int kvl = a.s.kv_len[b] + (a.first ? 0 : 1);
    const int ap = a.s.audio_pos[b] + (a.first ? 0 : 1);
    // valle.py:1044-1048: argmax == EOS  or  sample == EOS  or  (y.shape[1] - P) > 16 * S
    int stop = (!dyn.ignore_eos && ((argmax == 1024) || (sample == 1024))) || (n + a.bos > a.s.cap[b]);
    if (dyn.max_new > 0 && n >= dyn.max_new) stop = 1;
    if (dyn.has_forced) stop = n >= dyn.forced_len[b];
    if (n >= (int)a.g_stride || kvl >= a.ctx_max) stop = 1;  // capacity guard
    int next = sample;
    if (!stop) {
      if (dyn.has_forced) {
        const int64_t f = dyn.forced[(int64_t)b * dyn.forced_stride + n];
        next = (int)f;
        if (f < 0 || f >= (int64_t)a.V + a.bos) {  // outside ar_audio_embedding: the reference's nn.Embedding raises IndexError
          next = 0;
          if (a.id_err) atomicOr(a.id_err, 4);
        }
      }
      a.tokens[(int64_t)b * a.g_stride + n] = next;
      a.sampled[(int64_t)b * a.g_stride + n] = sample;
      a.s.n_gen[b] = n + 1;
      a.s.kv_len[b] = kvl;
      a.s.audio_pos[b] = ap;
    } else {
      if (n < (int)a.g_stride) a.sampled[(int64_t)b * a.g_stride + n] = sample;  // the stopping iteration's own draw (e.g. EOS), for the hooks
      a.s.done[b] = 1;
      atomicAdd(a.s.done_count, 1);
    }
    a.s.iter[b] = it + 1;
    sh_next = next;
    sh_stop = stop;
    sh_pos = ap;
  }
  __syncthreads();
  if (sh_stop) return;
  // next step's input: ar_audio_position(ar_audio_embedding(token))  (valle.py:1013-1015)
  const float4* e = reinterpret_cast<const float4*>(a.audio_emb + (int64_t)sh_next * a.d);
  const float4* pe = reinterpret_cast<const float4*>(a.pe + (int64_t)sh_pos * a.d);
  const float alpha = *a.alpha_audio;
  float4* xo = reinterpret_cast<float4*>(a.x + (int64_t)b * a.d);
  for (int i = tid; i < (a.d >> 2); i += SAMP_T) {
    const float4 ev = e[i], pv = pe[i];
    const float4 xv = make_float4(__fadd_rn(ev.x, __fmul_rn(alpha, pv.x)), __fadd_rn(ev.y, __fmul_rn(alpha, pv.y)),
                                  __fadd_rn(ev.z, __fmul_rn(alpha, pv.z)), __fadd_rn(ev.w, __fmul_rn(alpha, pv.w)));
    xo[i] = xv;
    if (a.lnp.gamma != nullptr) {
      // producer side of the fused LayerNorm of the batched step (kernels.h LnProducer): bf16(x * gamma) of the first layer's
      // LayerNorm in the fragment-major X layout + (mean, M2) of every 16-column group (4 consecutive threads; d % 16 == 0)
      typedef __bf16 b4 __attribute__((ext_vector_type(4)));
      const float4 g = reinterpret_cast<const float4*>(a.lnp.gamma)[i];
      b4 o4;
      o4[0] = (__bf16)(xv.x * g.x); o4[1] = (__bf16)(xv.y * g.y); o4[2] = (__bf16)(xv.z * g.z); o4[3] = (__bf16)(xv.w * g.w);
      *reinterpret_cast<b4*>(reinterpret_cast<bf16_t*>(a.lnp.xg_out) + xf_index(b, i * 4, a.lnp.MF, a.lnp.w8 != 0)) = o4;
      float s = (xv.x + xv.y) + (xv.z + xv.w);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const float mean = s * (1.0f / 16.0f);
      const float t0 = xv.x - mean, t1 = xv.y - mean, t2 = xv.z - mean, t3 = xv.w - mean;
      float q = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, t0 * t0)));
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      if ((i & 3) == 0) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f2*>(a.lnp.stats_out + ((int64_t)b * (a.d >> 4) + (i >> 2)) * 2) = f2{mean, q};
      }
    }
  }
}

int launch_ar_sample(hipStream_t st, const ArSampleArgs& a) {
  if (a.V > SAMP_T * SAMP_PER) return -1;
  hipLaunchKernelGGL(ar_sample_kernel, dim3(a.B), dim3(SAMP_T), 0, st, a);
  return 0;
}

// samples = argmax(logits); codes.append(samples); y_emb[:, P:] += embedding(samples)
// (valle.py:1130-1134).  One wave per generated frame.
constexpr int NARG_NW = 4;
__global__ __launch_bounds__(NARG_NW * 64) void nar_argmax_kernel(NarArgmaxArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * NARG_NW + (threadIdx.x >> 6);
  if (r >= a.rows) return;
  const float* lg = a.logits + r * a.V;
  unsigned long long best = 0ull;
  for (int idx = lane; idx < a.V; idx += 64) {
    const unsigned long long key = ((unsigned long long)float_key(lg[idx]) << 32) | (unsigned)(0x7fffffff - idx);
    best = key > best ? key : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(best, o, 64);
    best = t > best ? t : best;
  }
  const int code = 0x7fffffff - (int)(best & 0xffffffffu);
  const int b = a.grow_seq[r], g = a.grow_pos[r];
  if (lane == 0) a.codes[((int64_t)b * a.g_stride + g) * a.Q + a.col] = code;
  if (a.next_emb != nullptr) {
    int ecode = code;
    if (a.forced != nullptr) {  // teacher-forced NAR (parity hook): the next stage sees the given history
      const int64_t f = a.forced[((int64_t)b * a.f_stride + g) * a.Q + a.col];
      ecode = f < 0 ? 0 : (f >= a.V ? a.V - 1 : (int)f);
    }
    const float4* e = reinterpret_cast<const float4*>(a.next_emb + (int64_t)ecode * a.d);
    float4* y = reinterpret_cast<float4*>(a.y_emb + ((int64_t)a.aoff[b] + a.prompt_len[b] + g) * a.d);
    for (int i = lane; i < (a.d >> 2); i += 64) {
      const float4 yv = y[i], ev = e[i];
      y[i] = make_float4(yv.x + ev.x, yv.y + ev.y, yv.z + ev.z, yv.w + ev.w);
    }
  }
}

int launch_nar_argmax(hipStream_t st, const NarArgmaxArgs& a) {
  if (a.rows <= 0) return 0;
  hipLaunchKernelGGL(nar_argmax_kernel, dim3((unsigned)((a.rows + NARG_NW - 1) / NARG_NW)), dim3(NARG_NW * 64), 0, st, a);
  return 0;
}

}  // namespace vle
