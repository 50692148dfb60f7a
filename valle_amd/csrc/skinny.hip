// AR-step linear layers at batch <= 8: HBM-bound weight streaming (arithmetic intensity ~ B flop/B).
//   out[b][n] = epi( sum_k W[n][k] * pro(x)[b][k] + bias[n] )
// replaces, for ONE new token per utterance, the in-proj / out-proj / linear1 / linear2 / predict
// `linear` calls of the reference's per-step full-sequence forward
// (valle/modules/activation.py:414-421, valle/modules/transformer.py:297-302,332-334,
//  valle/models/valle.py:1039) together with the ops around them:
//   prologue PRO_LN   : LayerNorm of the residual row (transformer.py:57-74) fused in front
//   prologue PRO_ATTN : merge of the split-K decode-attention partials (attention.hip)
//   epilogue SEPI_QKV : + bias, q kept fp32, K/V written straight into the KV cache slot
//   epilogue SEPI_RESID: + bias, residual add in place on the fp32 stream
//   epilogue SEPI_RELU : + bias, ReLU (transformer.py:187, 333)
//
// gfx950 mapping: W is [N][K] row-major = one contiguous K-run per output row.  A wave64 owns RPW
// rows; lane l loads the 16-byte vector l of each 1-KiB row chunk (fully coalesced
// global_load_dwordx4).  All weight loads of the first PRE chunks are issued BEFORE the prologue
// (they do not depend on x), so HBM latency overlaps the LayerNorm / partial merge; activations sit
// in LDS as fp32 and are broadcast-read with ds_read_b128.  No LDS staging of weights: each
// weight byte is used once (guide: "GEMV / M <= 16: load straight to VGPRs").
#include "common.h"
#include "kernels.h"

namespace vle {

constexpr int SK_T = 256;  // 4 waves

template <typename T, int NB, int RPW, int PRO, int EPI>
__global__ __launch_bounds__(SK_T) void skinny_kernel(SkinnyArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int PRE = (8 / RPW) < 1 ? 1 : (8 / RPW);  // register-prefetched chunks (<= 8 vectors / lane)
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [NB][K]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int K = a.K, N = a.N;
  const int row0 = (blockIdx.x * (SK_T / 64) + w) * RPW;
  const T* W = reinterpret_cast<const T*>(a.w);

  // ---- issue the first PRE weight chunks (independent of the prologue) ----------------------
  uint4 pre[RPW][PRE];
#pragma unroll
  for (int p = 0; p < PRE; ++p) {
    const int k0 = (p * 64 + lane) * VEC;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      int row = row0 + r;
      row = row < N ? row : N - 1;
      pre[r][p] = k0 < K ? *reinterpret_cast<const uint4*>(W + (int64_t)row * K + k0) : make_uint4(0, 0, 0, 0);
    }
  }

  // ---- prologue: activations -> LDS (fp32) -----------------------------------------------------
  if constexpr (PRO == PRO_PLAIN) {
    for (int b = 0; b < NB; ++b)
      for (int i = tid; i < (K >> 2); i += SK_T) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < a.B) v = reinterpret_cast<const float4*>(a.x + (int64_t)b * K)[i];
        reinterpret_cast<float4*>(xs + b * K)[i] = v;
      }
  } else if constexpr (PRO == PRO_LN) {
    // one wave per utterance row: wave-level reductions only, no block barrier inside
    for (int b = w; b < NB; b += SK_T / 64) {
      float* xr = xs + b * K;
      if (b >= a.B) {
        for (int i = lane; i < K; i += 64) xr[i] = 0.f;
        continue;
      }
      const float* src = a.x + (int64_t)b * K;
      float s = 0.f;
      for (int i = lane; i < (K >> 2); i += 64) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<float4*>(xr)[i] = v;
        s += (v.x + v.y) + (v.z + v.w);
      }
      const float mean = wave_sum(s) / (float)K;
      float q = 0.f;
      for (int i = lane; i < (K >> 2); i += 64) {
        const float4 v = reinterpret_cast<const float4*>(xr)[i];  // own writes: same lane, no barrier
        const float c0 = v.x - mean, c1 = v.y - mean, c2 = v.z - mean, c3 = v.w - mean;
        q += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + LN_EPS);
      for (int i = lane; i < (K >> 2); i += 64) {
        const float4 v = reinterpret_cast<const float4*>(xr)[i];
        const float4 g = reinterpret_cast<const float4*>(a.gamma)[i];
        const float4 be = reinterpret_cast<const float4*>(a.beta)[i];
        reinterpret_cast<float4*>(xr)[i] =
            make_float4((v.x - mean) * rstd * g.x + be.x, (v.y - mean) * rstd * g.y + be.y,
                        (v.z - mean) * rstd * g.z + be.z, (v.w - mean) * rstd * g.w + be.w);
      }
    }
  } else {  // PRO_ATTN: o = sum_s exp(m_s - m) o_s / sum_s exp(m_s - m) l_s
    const int dh = a.dh, H = a.nhead, ns = a.nsplit;
    for (int b = 0; b < NB; ++b)
      for (int i = tid; i < K; i += SK_T) {
        float val = 0.f;
        if (b < a.B) {
          const int h = i / dh;
          const float* ml = a.part_ml + ((int64_t)(b * H + h) * ns) * 2;
          const float* po = a.part_o + (int64_t)b * ns * K + i;  // [B][nsplit][d]
          float m = -1e30f;
          for (int s = 0; s < ns; ++s) m = fmaxf(m, ml[2 * s]);
          float l = 0.f, o = 0.f;
          for (int s = 0; s < ns; ++s) {
            const float f = expf(ml[2 * s] - m);
            l += ml[2 * s + 1] * f;
            o += po[(int64_t)s * K] * f;
          }
          val = o / l;
        }
        xs[b * K + i] = val;
      }
  }
  __syncthreads();

  // ---- dot products ----------------------------------------------------------------------------
  float acc[RPW][NB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;

  auto consume = [&](const uint4 (&wv)[RPW], int k0) {
    float wf[RPW][VEC];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if constexpr (sizeof(T) == 4) {
        wf[r][0] = __uint_as_float(wv[r].x); wf[r][1] = __uint_as_float(wv[r].y);
        wf[r][2] = __uint_as_float(wv[r].z); wf[r][3] = __uint_as_float(wv[r].w);
      } else {
        wf[r][0] = __uint_as_float(wv[r].x << 16); wf[r][1] = __uint_as_float(wv[r].x & 0xffff0000u);
        wf[r][2] = __uint_as_float(wv[r].y << 16); wf[r][3] = __uint_as_float(wv[r].y & 0xffff0000u);
        wf[r][4] = __uint_as_float(wv[r].z << 16); wf[r][5] = __uint_as_float(wv[r].z & 0xffff0000u);
        wf[r][6] = __uint_as_float(wv[r].w << 16); wf[r][7] = __uint_as_float(wv[r].w & 0xffff0000u);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float xv[VEC];
#pragma unroll
      for (int v4 = 0; v4 < VEC / 4; ++v4) {
        const float4 t = *reinterpret_cast<const float4*>(xs + b * K + k0 + v4 * 4);
        xv[v4 * 4 + 0] = t.x; xv[v4 * 4 + 1] = t.y; xv[v4 * 4 + 2] = t.z; xv[v4 * 4 + 3] = t.w;
      }
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[r][b] = fmaf(wf[r][j], xv[j], acc[r][b]);
    }
  };

#pragma unroll
  for (int p = 0; p < PRE; ++p) {
    const int k0 = (p * 64 + lane) * VEC;
    if (k0 < K) {
      uint4 wv[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) wv[r] = pre[r][p];
      consume(wv, k0);
    }
  }
#pragma unroll 2
  for (int k0 = (PRE * 64 + lane) * VEC; k0 < K; k0 += 64 * VEC) {
    uint4 wv[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      int row = row0 + r;
      row = row < N ? row : N - 1;
      wv[r] = *reinterpret_cast<const uint4*>(W + (int64_t)row * K + k0);
    }
    consume(wv, k0);
  }

#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);

  // ---- epilogue: lane (r * NB + b) owns output (row0 + r, b) ----------------------------------------
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (lane != r * NB + b) continue;
      const int n = row0 + r;
      if (n >= N || b >= a.B) continue;
      float v = acc[r][b] + (a.bias ? a.bias[n] : 0.f);
      if constexpr (EPI == SEPI_STORE) {
        a.out[(int64_t)b * N + n] = v;
      } else if constexpr (EPI == SEPI_RELU) {
        a.out[(int64_t)b * N + n] = fmaxf(v, 0.f);
      } else if constexpr (EPI == SEPI_RESID) {
        a.resid[(int64_t)b * N + n] += v;
      } else {  // SEPI_QKV: rows [0,d) = Q, [d,2d) = K, [2d,3d) = V  (activation.py:128-130)
        const int d = N / 3, which = n / d, j = n - which * d;
        if (which == 0) {
          a.q_out[(int64_t)b * d + j] = v;
        } else {
          const int h = j / a.dh, e = j - h * a.dh;
          const int64_t off = (((int64_t)b * a.nhead + h) * a.ctx_max + a.kv_len[b]) * a.dh + e;
          store_elem<T>(reinterpret_cast<T*>(which == 1 ? a.k_cache : a.v_cache) + off, v);
        }
      }
    }
  }
}

template <typename T, int NB, int RPW, int PRO, int EPI>
static int skinny_launch_one(hipStream_t st, const SkinnyArgs& a) {
  const int rows_per_block = (SK_T / 64) * RPW;
  const dim3 grid((a.N + rows_per_block - 1) / rows_per_block), block(SK_T);
  const size_t lds = (size_t)NB * a.K * sizeof(float);
  auto kfn = skinny_kernel<T, NB, RPW, PRO, EPI>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return -3;
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kfn, grid, block, lds, st, a);
  return 0;
}

// only the (prologue, epilogue) pairs the AR step and the operator API use are instantiated
template <typename T, int NB, int RPW>
static int skinny_launch_pro(hipStream_t st, const SkinnyArgs& a) {
  const int key = a.pro * 8 + a.epi;
  switch (key) {
    case PRO_LN * 8 + SEPI_QKV: return skinny_launch_one<T, NB, RPW, PRO_LN, SEPI_QKV>(st, a);
    case PRO_LN * 8 + SEPI_RELU: return skinny_launch_one<T, NB, RPW, PRO_LN, SEPI_RELU>(st, a);
    case PRO_LN * 8 + SEPI_STORE: return skinny_launch_one<T, NB, RPW, PRO_LN, SEPI_STORE>(st, a);
    case PRO_ATTN * 8 + SEPI_RESID: return skinny_launch_one<T, NB, RPW, PRO_ATTN, SEPI_RESID>(st, a);
    case PRO_PLAIN * 8 + SEPI_RESID: return skinny_launch_one<T, NB, RPW, PRO_PLAIN, SEPI_RESID>(st, a);
    case PRO_PLAIN * 8 + SEPI_STORE: return skinny_launch_one<T, NB, RPW, PRO_PLAIN, SEPI_STORE>(st, a);
    case PRO_PLAIN * 8 + SEPI_RELU: return skinny_launch_one<T, NB, RPW, PRO_PLAIN, SEPI_RELU>(st, a);
    default: return -1;
  }
}

template <typename T, int NB>
static int skinny_launch_rpw(hipStream_t st, const SkinnyArgs& a) {
  // rows per wave: keep >= ~256 blocks in flight, otherwise favour more loads in flight per lane
  const int rows_rpw2 = (SK_T / 64) * 2;
  if (NB <= 4 && a.N / rows_rpw2 >= 320) return skinny_launch_pro<T, NB, 2>(st, a);
  return skinny_launch_pro<T, NB, 1>(st, a);
}

template <typename T>
static int skinny_launch_nb(hipStream_t st, const SkinnyArgs& a) {
  if (a.B <= 1) return skinny_launch_rpw<T, 1>(st, a);
  if (a.B <= 2) return skinny_launch_rpw<T, 2>(st, a);
  if (a.B <= 4) return skinny_launch_rpw<T, 4>(st, a);
  if (a.B <= 8) return skinny_launch_rpw<T, 8>(st, a);
  return -1;
}

int launch_skinny(hipStream_t st, int dtype, const SkinnyArgs& a) {
  if (a.K % 8 != 0 || a.K % 4 != 0) return -1;
  if ((size_t)8 * a.K * sizeof(float) > 160 * 1024 && a.B > 4) return -1;
  if (dtype == DT_F32) return skinny_launch_nb<float>(st, a);
  return skinny_launch_nb<bf16_t>(st, a);
}

}  // namespace vle
