// AR-step linear layers for ONE utterance (batch 1): wave-autonomous weight-streaming GEMV.
//   out[n] = epi( sum_k W[n][k] * pro(x)[k] + bias[n] )
// Replaces, for the one new token of a decode step, the in-proj / out-proj / linear1 / linear2 /
// predict `linear` calls of the reference's per-step full-sequence forward
// (valle/modules/activation.py:414-421, valle/modules/transformer.py:297-302, 332-334,
//  valle/models/valle.py:1039) and the ops around them (same prologue / epilogue set as skinny.hip).
//
// Why a second GEMV next to skinny.hip: at batch 1 a decode step is a chain of ~60 dependent
// kernels of 2-8 MB each; every kernel is latency-bound (one HBM round trip ~1-2 us under load
// against ~1 us of streaming), so the design rule is ONE parallel burst of loads per kernel and
// nothing dependent after it:
//   * a wave owns RPW rows and the whole K extent: lane l holds the 16-byte vector l of every
//     64-vector chunk, ALL RPW x NCH weight vectors are requested up front (non-temporal: each
//     weight byte is used once per step, keep L2/MALL for the KV cache and the small vectors);
//   * bias / residual / LayerNorm affine / x / attention partials / kv_len are requested in the same
//     burst, before any reduction: biases are read once per step and have been evicted by the
//     300 MB weight stream, so a load issued in the epilogue would cost a second HBM latency;
//   * x lives in REGISTERS (K / 64 values per lane), not LDS: no block barrier, no LDS round trip;
//     every wave recomputes the LayerNorm statistics (K values, two DPP reductions) or the
//     split-KV merge for its own lanes -- redundant flops are free here;
//   * reductions use DPP row ops + v_readlane (common.h), not ds_bpermute shuffles.
// Shapes outside the instantiated set (K not a multiple of 64 vectors, batch > 1) use skinny.hip.
#include "common.h"
#include "kernels.h"

namespace vle {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

constexpr int G1_T = 256;  // 4 independent waves per block
constexpr float G1_NEG = -1e30f;

template <typename T>
__device__ inline void widen16(const u32x4_t& v, float (&f)[Elem<T>::VEC]);
template <>
__device__ inline void widen16<float>(const u32x4_t& v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <>
__device__ inline void widen16<bf16_t>(const u32x4_t& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// FP8W: the lane's 8 weights of a chunk are 8 bytes of e4m3fn in v.x / v.y (v_cvt_pk_f32_fp8: two values per op)
template <>
__device__ inline void widen16<bf16w8_t>(const u32x4_t& v, float (&f)[8]) {
  typedef float f32x2v_t __attribute__((ext_vector_type(2)));
  const f32x2v_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
  const f32x2v_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
  f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y; f[6] = d.x; f[7] = d.y;
}

template <>
__device__ inline void widen16<bf16w8t_t>(const u32x4_t& v, float (&f)[8]) {
  widen16<bf16w8_t>(v, f);
}

// weight-stream traits: how a lane fetches its VEC weights of one chunk, and the element type of the KV cache
template <typename T>
struct G1W {
  typedef T cache_t;
  static constexpr bool kScaled = false;
  __device__ static inline u32x4_t load(const T* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
};
template <>
struct G1W<bf16w8_t> {
  typedef bf16_t cache_t;
  static constexpr bool kScaled = true;
  __device__ static inline u32x4_t load(const bf16w8_t* p) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return u32x4_t{t.x, t.y, 0u, 0u};
  }
};

template <>
struct G1W<bf16w8t_t> {
  typedef bf16_t cache_t;
  static constexpr bool kScaled = true;
  __device__ static inline u32x4_t load(const bf16w8t_t* p) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = *reinterpret_cast<const u32x2_t*>(p);  // default cache policy
    return u32x4_t{t.x, t.y, 0u, 0u};
  }
};

// VEC consecutive fp32 values at p (16-byte aligned)
template <int VEC>
__device__ inline void load_f32_vec(const float* p, float (&f)[VEC]) {
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const f32x4v_t t = *reinterpret_cast<const f32x4v_t*>(p + q * 4);
    f[q * 4 + 0] = t.x; f[q * 4 + 1] = t.y; f[q * 4 + 2] = t.z; f[q * 4 + 3] = t.w;
  }
}

template <typename T, int NCH, int RPW, int PRO, int EPI, int NS>
__global__ __launch_bounds__(G1_T) void gemv1_kernel(SkinnyArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CH = 64 * VEC;  // K elements per chunk (one 16-byte vector per lane)
  constexpr int K = NCH * CH;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (G1_T / 64) + (threadIdx.x >> 6);
  const int row0 = wave * RPW;
  const int N = a.N;
  if (row0 >= N) return;  // wave-uniform
  const unsigned long long kt0 = ktrace_begin(a.kt);
  const T* W = reinterpret_cast<const T*>(a.w);

  // ---- the burst, in the order the data is needed (vmcnt retires loads in issue order): activations
  // and their affine / partials, then the weights, then the epilogue operands of the row this lane
  // writes (lane r < RPW owns row0 + r).  sched_barrier(0) pins every request above the first wait.
  float x[NCH][VEC];
  float g[PRO == PRO_LN ? NCH : 1][VEC], be[PRO == PRO_LN ? NCH : 1][VEC];
  constexpr bool kAttn = PRO == PRO_ATTN;
  float ms[kAttn ? NCH : 1][NS], ls[kAttn ? NCH : 1][NS], po[kAttn ? NCH : 1][NS][VEC];
  if constexpr (PRO == PRO_PLAIN) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_f32_vec<VEC>(a.x + c * CH + lane * VEC, x[c]);
  } else if constexpr (PRO == PRO_LN) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_f32_vec<VEC>(a.x + c * CH + lane * VEC, x[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      load_f32_vec<VEC>(a.gamma + c * CH + lane * VEC, g[c]);
      load_f32_vec<VEC>(a.beta + c * CH + lane * VEC, be[c]);
    }
  } else {
    // partials of the decode attention (decode_attn.hip): part_o [NS][d], part_ml [H][NS][2]
    const int dh = a.dh;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k0 = c * CH + lane * VEC;
      const int h = k0 / dh;  // dh % VEC == 0: the lane's vector lies inside one head
      const float* ml = a.part_ml + (int64_t)h * NS * 2;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        ms[c][s] = ml[2 * s];
        ls[c][s] = ml[2 * s + 1];
        load_f32_vec<VEC>(a.part_o + (int64_t)s * K + k0, po[c][s]);
      }
    }
  }
  u32x4_t wv[RPW][NCH];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    int row = row0 + r;
    row = row < N ? row : N - 1;
    const T* wr = W + (int64_t)row * K + lane * VEC;
#pragma unroll
    for (int c = 0; c < NCH; ++c) wv[r][c] = G1W<T>::load(wr + c * CH);
  }
  const int myrow = row0 + lane;
  const bool writer = lane < RPW && myrow < N;
  float bias_v = 0.f, resid_v = 0.f, scale_v = 1.f;
  int kvl = 0;
  if (writer) {
    if (a.bias) bias_v = a.bias[myrow];
    if constexpr (G1W<T>::kScaled) scale_v = a.wscale[myrow];  // the row's power-of-two scale (FP8W)
    if constexpr (EPI == SEPI_RESID) resid_v = a.resid[myrow];
    if constexpr (EPI == SEPI_QKV) kvl = a.kv_len[0];
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: this lane's K / 64 activations, fp32, in registers ------------------------------
  if constexpr (PRO == PRO_LN) {
    // LayerNorm (valle/modules/transformer.py:57-74; eps 1e-5, biased variance), two-pass in fp32
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += x[c][j];
    const float mean = wave_sum_dpp(s) * (1.0f / (float)K);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float t = x[c][j] - mean;
        q = fmaf(t, t, q);
      }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * (1.0f / (float)K) + LN_EPS);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[c][j] = (x[c][j] - mean) * rstd * g[c][j] + be[c][j];
  } else if constexpr (PRO == PRO_ATTN) {
    // merge of the NS split-KV partials:  o = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float M = ms[c][0];
#pragma unroll
      for (int s = 1; s < NS; ++s) M = fmaxf(M, ms[c][s]);
      float L = 0.f, acc[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float f = __expf(ms[c][s] - M);
        L = fmaf(ls[c][s], f, L);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(po[c][s][j], f, acc[j]);
      }
      const float inv = 1.0f / L;
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[c][j] = acc[j] * inv;
    }
  }

  // ---- dot products ----------------------------------------------------------------------------------
  float acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float t0 = 0.f, t1 = 0.f;  // two chains per row: shorter dependent FMA chain
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float wf[VEC];
      widen16<T>(wv[r][c], wf);
#pragma unroll
      for (int j = 0; j < VEC; j += 2) {
        t0 = fmaf(wf[j], x[c][j], t0);
        t1 = fmaf(wf[j + 1], x[c][j + 1], t1);
      }
    }
    acc[r] = t0 + t1;
  }
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float t = wave_sum_dpp(acc[r]);
    mine = lane == r ? t : mine;
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  if (lane == 0) ktrace_end(a.kt, kt0, wave);
  if (!writer) return;
  const float v = G1W<T>::kScaled ? fmaf(mine, scale_v, bias_v) : mine + bias_v;  // * 2^e is exact: one rounding, as in bf16 mode
  if constexpr (EPI == SEPI_STORE) {
    a.out[myrow] = v;
  } else if constexpr (EPI == SEPI_RELU) {
    a.out[myrow] = fmaxf(v, 0.f);
  } else if constexpr (EPI == SEPI_RESID) {
    a.resid[myrow] = resid_v + v;
  } else {  // SEPI_QKV: rows [0,d) = Q, [d,2d) = K, [2d,3d) = V  (valle/modules/activation.py:128-130)
    const int d = N / 3, which = myrow / d, j = myrow - which * d;
    if (which == 0) {
      a.q_out[j] = v;
    } else {
      const int h = j / a.dh, e = j - h * a.dh;
      const int64_t off = ((int64_t)h * a.ctx_max + kvl) * a.dh + e;
      typedef typename G1W<T>::cache_t CT;
      store_elem<CT>(reinterpret_cast<CT*>(which == 1 ? a.k_cache : a.v_cache) + off, v);
    }
  }
}

template <typename T, int NCH, int RPW, int PRO, int EPI, int NS>
static int g1_launch(hipStream_t st, const SkinnyArgs& a) {
  const int waves = (a.N + RPW - 1) / RPW;
  const dim3 grid((waves + G1_T / 64 - 1) / (G1_T / 64)), block(G1_T);
  hipLaunchKernelGGL((gemv1_kernel<T, NCH, RPW, PRO, EPI, NS>), grid, block, 0, st, a);
  return 0;
}

template <typename T, int NCH, int RPW>
static int g1_dispatch_pe(hipStream_t st, const SkinnyArgs& a) {
  const int key = a.pro * 8 + a.epi;
  if constexpr (NCH <= 4) {  // K = d family
    switch (key) {
      case PRO_LN * 8 + SEPI_QKV: return g1_launch<T, NCH, RPW, PRO_LN, SEPI_QKV, 1>(st, a);
      case PRO_LN * 8 + SEPI_RELU: return g1_launch<T, NCH, RPW, PRO_LN, SEPI_RELU, 1>(st, a);
      case PRO_LN * 8 + SEPI_STORE: return g1_launch<T, NCH, RPW, PRO_LN, SEPI_STORE, 1>(st, a);
      case PRO_PLAIN * 8 + SEPI_RESID: return g1_launch<T, NCH, RPW, PRO_PLAIN, SEPI_RESID, 1>(st, a);
      case PRO_PLAIN * 8 + SEPI_STORE: return g1_launch<T, NCH, RPW, PRO_PLAIN, SEPI_STORE, 1>(st, a);
      case PRO_PLAIN * 8 + SEPI_RELU: return g1_launch<T, NCH, RPW, PRO_PLAIN, SEPI_RELU, 1>(st, a);
      case PRO_ATTN * 8 + SEPI_RESID:
        if constexpr (RPW == 1) {
          switch (a.nsplit) {
            case 1: return g1_launch<T, NCH, 1, PRO_ATTN, SEPI_RESID, 1>(st, a);
            case 2: return g1_launch<T, NCH, 1, PRO_ATTN, SEPI_RESID, 2>(st, a);
            case 4: return g1_launch<T, NCH, 1, PRO_ATTN, SEPI_RESID, 4>(st, a);
            case 8: return g1_launch<T, NCH, 1, PRO_ATTN, SEPI_RESID, 8>(st, a);
            case 16:
              if constexpr (NCH * Elem<T>::VEC <= 16) return g1_launch<T, NCH, 1, PRO_ATTN, SEPI_RESID, 16>(st, a);
              return 1;  // the partials would not fit the register file: generic kernel
            default: return 1;
          }
        }
        return 1;
      default: return 1;
    }
  } else {  // K = 4d family: linear2 only
    if (key == PRO_PLAIN * 8 + SEPI_RESID) return g1_launch<T, NCH, RPW, PRO_PLAIN, SEPI_RESID, 1>(st, a);
    return 1;
  }
}

template <typename T, int NCH>
static int g1_dispatch_rpw(hipStream_t st, const SkinnyArgs& a) {
  if constexpr (NCH <= 4) {
    // >= ~1024 waves keeps every CU busy; more rows per wave = more bytes in flight per lane
    int rpw = 1;
    if (a.pro != PRO_ATTN) {
      if (a.N >= 4096 && NCH <= 2) rpw = 4;
      // N = 3 x 1024 k (the QKV GEMV at d = 1024 k): 3 rows per wave give exactly k workgroups per CU.  With 2 rows per wave
      // 384 workgroups leave half the CUs with two and half with one, and the launch ends 1.4 us after its first wave does
      // (profiles/r02_ktrace_b1_timeline.csv, end_spread); measured 253.4 -> 251.1 us per step (tools/ar_tune.py qkv_rpw3)
      else if (a.N % 3 == 0 && (a.N / 3) % 1024 == 0) rpw = 3;
      else if (a.N >= 2048) rpw = 2;
    }
    if (a.rpw_override > 0) rpw = a.rpw_override;
    if (a.pro == PRO_ATTN) rpw = 1;
    if (rpw * NCH > 16) rpw = 1;
    switch (rpw) {
      case 1: return g1_dispatch_pe<T, NCH, 1>(st, a);
      case 2: return g1_dispatch_pe<T, NCH, 2>(st, a);
      case 3: return g1_dispatch_pe<T, NCH, 3>(st, a);
      case 4:
        if constexpr (NCH <= 2) return g1_dispatch_pe<T, NCH, 4>(st, a);
        return g1_dispatch_pe<T, NCH, 2>(st, a);
      default: return 1;
    }
  } else {
    return g1_dispatch_pe<T, NCH, 1>(st, a);
  }
}

template <typename T>
static int g1_dispatch_nch(hipStream_t st, const SkinnyArgs& a) {
  constexpr int CH = 64 * Elem<T>::VEC;
  if (a.K % CH != 0) return 1;
  switch (a.K / CH) {
    case 1: return g1_dispatch_rpw<T, 1>(st, a);
    case 2: return g1_dispatch_rpw<T, 2>(st, a);
    case 3: return g1_dispatch_rpw<T, 3>(st, a);
    case 4: return g1_dispatch_rpw<T, 4>(st, a);
    case 8: return g1_dispatch_rpw<T, 8>(st, a);
    case 12: return g1_dispatch_rpw<T, 12>(st, a);
    case 16: return g1_dispatch_rpw<T, 16>(st, a);
    default: return 1;
  }
}

// returns 0 = launched, 1 = shape not covered (caller falls back to launch_skinny), < 0 = error
int launch_gemv1(hipStream_t st, int dtype, const SkinnyArgs& a) {
  if (a.B != 1 || a.N <= 0) return 1;
  if (a.pro == PRO_ATTN && (a.dh % (dtype == DT_F32 ? 4 : 8) != 0)) return 1;
  if (dtype == DT_FP8W) return !a.wscale ? -1 : a.temporal ? g1_dispatch_nch<bf16w8t_t>(st, a) : g1_dispatch_nch<bf16w8_t>(st, a);
  if (dtype == DT_F32) return g1_dispatch_nch<float>(st, a);
  return g1_dispatch_nch<bf16_t>(st, a);
}

}  // namespace vle
