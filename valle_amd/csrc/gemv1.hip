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
#include "gemv1_dev.h"

namespace vle {


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
  constexpr bool kAttn = PRO == PRO_ATTN || PRO == PRO_ATTN_SELF;
  constexpr bool kSelf = PRO == PRO_ATTN_SELF;  // the partials cover the OLD keys only: the new token's own term is merged here
  float ms[kAttn ? NCH : 1][NS], ls[kAttn ? NCH : 1][NS], po[kAttn ? NCH : 1][NS][VEC];
  float qs[kSelf ? NCH : 1][VEC], ksf[kSelf ? NCH : 1][VEC], vsf[kSelf ? NCH : 1][VEC];
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
      if constexpr (kSelf) {
        load_f32_vec<VEC>(a.q_self + k0, qs[c]);
        load_f32_vec<VEC>(a.k_self + k0, ksf[c]);
        load_f32_vec<VEC>(a.v_self + k0, vsf[c]);
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
  } else if constexpr (kAttn) {
    // merge of the NS split-KV partials:  o = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s
    // PRO_ATTN_SELF: plus the new token's own key as one more partial  (m, l, o) = (q . k / sqrt(dh), 1, v)  -- its score is
    // the dot product over the head's dh columns = dh / VEC consecutive lanes of this chunk (DPP group sum)
    const int lpk = kSelf ? a.dh / VEC : 1;
    const float sscale = 1.0f / sqrtf((float)a.dh);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float M = ms[c][0];
#pragma unroll
      for (int s = 1; s < NS; ++s) M = fmaxf(M, ms[c][s]);
      float sself = 0.f;
      if constexpr (kSelf) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) t = fmaf(qs[c][j], ksf[c][j], t);
        sself = head_group_sum(t, lpk) * sscale;
        M = fmaxf(M, sself);
      }
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
      if constexpr (kSelf) {
        const float f = __expf(sself - M);
        L += f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(vsf[c][j], f, acc[j]);
      }
      const float inv = 1.0f / L;
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[c][j] = (a.act_bf16 & 1) ? bf16_to_f32(f32_to_bf16(acc[j] * inv)) : acc[j] * inv;
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
    a.out[myrow] = (a.act_bf16 & 2) ? bf16_to_f32(f32_to_bf16(fmaxf(v, 0.f))) : fmaxf(v, 0.f);  // bit 1: the hidden row in bf16 (hT_step of the batched path)
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

template <typename T, int NCH>
__device__ inline void g1_layernorm(float (&x)[NCH][Elem<T>::VEC], const float (&g)[NCH][Elem<T>::VEC], const float (&be)[NCH][Elem<T>::VEC]) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int K = NCH * 64 * VEC;
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
}


// Totals of R = 8 / 16 per-lane partial sums over the 64 lanes of the wave, all at once: a butterfly that halves the number of
// values at every step (lanes l and l ^ 2^b split the rows between them), R - 1 exchanges instead of R full wave reductions --
// on the critical path of the fused launch's query rows (16 per wave).  Returns, in EVERY lane, the total of row
// g1_rows_owner(lane) = the bit-reversed low log2(R) bits of the lane.
template <int BIT>
__device__ inline float g1_xchg(float v, int lane) {  // value of lane ^ (1 << BIT), BIT < 4
  if constexpr (BIT == 0) return dpp_f32<0xB1>(v);
  else if constexpr (BIT == 1) return dpp_f32<0x4E>(v);
  else if constexpr (BIT == 2) {
    const float up = dpp_f32<0x104>(v), dn = dpp_f32<0x114>(v);  // row_shl:4 (from lane + 4), row_shr:4 (from lane - 4)
    return (lane & 4) ? dn : up;
  } else return dpp_f32<0x128>(v);  // row_ror:8: the other half of the 16-lane row
}
template <int R, int BIT>
__device__ inline void g1_rows_stage(float (&acc)[R], int lane) {
  constexpr int H = R >> (BIT + 1);  // values kept after this stage
  const bool hi = (lane >> BIT) & 1;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const float keep = hi ? acc[i + H] : acc[i], send = hi ? acc[i] : acc[i + H];
    acc[i] = keep + g1_xchg<BIT>(send, lane);
  }
}
template <int R>
__device__ inline float g1_rows_reduce(float (&acc)[R], int lane) {
  static_assert(R == 8 || R == 16, "butterfly sizes");
  g1_rows_stage<R, 0>(acc, lane);
  g1_rows_stage<R, 1>(acc, lane);
  g1_rows_stage<R, 2>(acc, lane);
  if constexpr (R == 16) g1_rows_stage<R, 3>(acc, lane);
  float v = acc[0];
  if constexpr (R == 8) v += g1_xchg<3>(v, lane);
  return rows4_sum(v);  // lanes l, l ^ 16, l ^ 32, l ^ 48
}
template <int R>
__device__ inline int g1_rows_owner(int lane) {  // the row whose total g1_rows_reduce leaves in this lane
  constexpr int LG = R == 16 ? 4 : 3;
  int r = 0;
#pragma unroll
  for (int b = 0; b < LG; ++b) r |= ((lane >> b) & 1) << (LG - 1 - b);
  return r;
}

// =====================================================================================================================
// Block-shared activations (round 3; "g1_shared" = 1, the default).  The wave-autonomous kernel above has every wave fetch the
// whole activation row itself: K fp32 values (+ the LayerNorm affine: 3 K; + NS split partials: (NS + 3) K) -- at d = 1024 that
// is 12-16 KB per wave through the CU's texture path against 2-8 KB of weights, and the ~64 B/clk of that path, not HBM, is
// what the burst queues on (tools/ubench_boundary.hip: a stand-in that streams the same 2 / 6 / 8 MB of weights with one
// activation dword per lane has a body of 1.15 / 1.85 / 2.1 us where out-proj / QKV / FFN1 measured 2.08 / 2.64 / 2.81).
// Here the 256 threads of a workgroup fetch the row ONCE (thread t owns K / 256 consecutive elements: 3-11 small loads, issued
// BEFORE the weights so that vmcnt, which retires in order, releases them first), build the prologue's result once --
// LayerNorm with two block reductions, or the split-KV merge -- and publish it in LDS; the waves pick their K / 64 values per
// lane from there while the weight burst is still in flight.  The barriers are raw s_barrier + lgkmcnt(0): they never wait
// for the weight loads.

template <typename T, int NCH, int RPW, int PRO, int EPI, int NS>
__global__ __launch_bounds__(G1_T) void gemv1s_kernel(SkinnyArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CH = 64 * VEC;
  constexpr int K = NCH * CH;
  constexpr int EPT = K / G1_T;  // activation elements per thread of the shared prologue (K % 256 == 0 for every instantiated K)
  constexpr bool kAttn = PRO == PRO_ATTN || PRO == PRO_ATTN_SELF;
  constexpr bool kSelf = PRO == PRO_ATTN_SELF;
  __shared__ __attribute__((aligned(16))) float sx[K];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = blockIdx.x * (G1_T / 64) + (tid >> 6);
  const int row0 = wave * RPW;
  const int N = a.N;
  const bool live = row0 < N;  // wave-uniform; dead waves still take part in the prologue and its barriers
  const unsigned long long kt0 = ktrace_begin(a.kt);
  const T* W = reinterpret_cast<const T*>(a.w);

  // ---- the burst: this thread's slice of the activations (small, first: vmcnt retires in issue order), then the weights,
  // then the epilogue operands of the row this lane writes
  float xv[EPT], gv[PRO == PRO_LN ? EPT : 1], bv[PRO == PRO_LN ? EPT : 1];
  float pm[kAttn ? NS : 1], pl[kAttn ? NS : 1], po[kAttn ? NS : 1][EPT];
  float qs[kSelf ? EPT : 1], ksf[kSelf ? EPT : 1], vsf[kSelf ? EPT : 1];
  const int e0 = tid * EPT;
  if constexpr (PRO == PRO_PLAIN) {
    load_ept<EPT>(a.x + e0, xv);
  } else if constexpr (PRO == PRO_LN) {
    load_ept<EPT>(a.x + e0, xv);
    load_ept<EPT>(a.gamma + e0, gv);
    load_ept<EPT>(a.beta + e0, bv);
  } else {
    const int h = e0 / a.dh;  // dh % EPT == 0: the thread's slice lies inside one head
    const float* ml = a.part_ml + (int64_t)h * NS * 2;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      pm[s] = ml[2 * s];
      pl[s] = ml[2 * s + 1];
      load_ept<EPT>(a.part_o + (int64_t)s * K + e0, po[s]);
    }
    if constexpr (kSelf) {
      load_ept<EPT>(a.q_self + e0, qs);
      load_ept<EPT>(a.k_self + e0, ksf);
      load_ept<EPT>(a.v_self + e0, vsf);
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
  const bool writer = live && lane < RPW && myrow < N;
  float bias_v = 0.f, resid_v = 0.f, scale_v = 1.f;
  int kvl = 0;
  if (writer) {
    if (a.bias) bias_v = a.bias[myrow];
    if constexpr (G1W<T>::kScaled) scale_v = a.wscale[myrow];
    if constexpr (EPI == SEPI_RESID) resid_v = a.resid[myrow];
    if constexpr (EPI == SEPI_QKV) kvl = a.kv_len[0];
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue, once per workgroup -> sx[K] -------------------------------------------------------------------------
  if constexpr (PRO == PRO_LN) {
    g1_block_layernorm<K>(xv, gv, bv, sx, red);
  } else if constexpr (PRO == PRO_PLAIN) {
    store_ept_lds<EPT>(sx + e0, xv);
    g1_lds_barrier();
  } else {
    // merge of the NS split-KV partials (+ the new token's own key, PRO_ATTN_SELF) for this thread's EPT columns of one head
    float M = pm[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) M = fmaxf(M, pm[s]);
    float sself = 0.f;
    if constexpr (kSelf) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < EPT; ++i) t = fmaf(qs[i], ksf[i], t);
      sself = head_group_sum64(t, a.dh / EPT) * (1.0f / sqrtf((float)a.dh));
      M = fmaxf(M, sself);
    }
    float L = 0.f, acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float f = __expf(pm[s] - M);
      L = fmaf(pl[s], f, L);
#pragma unroll
      for (int i = 0; i < EPT; ++i) acc[i] = fmaf(po[s][i], f, acc[i]);
    }
    if constexpr (kSelf) {
      const float f = __expf(sself - M);
      L += f;
#pragma unroll
      for (int i = 0; i < EPT; ++i) acc[i] = fmaf(vsf[i], f, acc[i]);
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] *= inv;
    if (a.act_bf16 & 1) {  // the merged attention row in bf16, as the batched path (att_step) and the persistent step's packed edge carry it
#pragma unroll
      for (int i = 0; i < EPT; ++i) acc[i] = bf16_to_f32(f32_to_bf16(acc[i]));
    }
    store_ept_lds<EPT>(sx + e0, acc);
    g1_lds_barrier();
  }
  if (!live) return;
  float x[NCH][VEC];
  g1_read_shared<T, NCH>(sx, x);

  // ---- dot products ----------------------------------------------------------------------------------
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float t = wave_sum_dpp(g1_dot<T, NCH>(wv[r], x));
    mine = lane == r ? t : mine;
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  if (lane == 0) ktrace_end(a.kt, kt0, wave);
  if (!writer) return;
  const float v = G1W<T>::kScaled ? fmaf(mine, scale_v, bias_v) : mine + bias_v;
  if constexpr (EPI == SEPI_STORE) {
    a.out[myrow] = v;
  } else if constexpr (EPI == SEPI_RELU) {
    a.out[myrow] = (a.act_bf16 & 2) ? bf16_to_f32(f32_to_bf16(fmaxf(v, 0.f))) : fmaxf(v, 0.f);  // bit 1: the hidden row in bf16 (hT_step of the batched path)
  } else if constexpr (EPI == SEPI_RESID) {
    a.resid[myrow] = resid_v + v;
  } else {
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

int g_g1_shared = 1;  // "g1_shared": 1 = block-shared activations (gemv1s_kernel), 0 = the wave-autonomous kernel (A/B)

template <typename T, int NCH, int RPW, int PRO, int EPI, int NS>
static int g1_launch(hipStream_t st, const SkinnyArgs& a) {
  const int waves = (a.N + RPW - 1) / RPW;
  const dim3 grid((waves + G1_T / 64 - 1) / (G1_T / 64)), block(G1_T);
  if (g_g1_shared) hipLaunchKernelGGL((gemv1s_kernel<T, NCH, RPW, PRO, EPI, NS>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemv1_kernel<T, NCH, RPW, PRO, EPI, NS>), grid, block, 0, st, a);
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
      case PRO_ATTN_SELF * 8 + SEPI_RESID:
        if constexpr (RPW == 1) {
          if (!a.q_self || !a.k_self || !a.v_self) return -1;
          switch (a.nsplit) {
            case 4: return g1_launch<T, NCH, 1, PRO_ATTN_SELF, SEPI_RESID, 4>(st, a);
            case 8: return g1_launch<T, NCH, 1, PRO_ATTN_SELF, SEPI_RESID, 8>(st, a);
            case 16:
              if constexpr (NCH * Elem<T>::VEC <= 16) return g1_launch<T, NCH, 1, PRO_ATTN_SELF, SEPI_RESID, 16>(st, a);
              return 1;
            default: return 1;
          }
        }
        return 1;
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
    const bool attn_pro = a.pro == PRO_ATTN || a.pro == PRO_ATTN_SELF;
    if (!attn_pro) {
      if (a.N >= 4096 && NCH <= 2) rpw = 4;
      // N = 3 x 1024 k (the QKV GEMV at d = 1024 k): 3 rows per wave give exactly k workgroups per CU.  With 2 rows per wave
      // 384 workgroups leave half the CUs with two and half with one, and the launch ends 1.4 us after its first wave does
      // (profiles/r02_ktrace_b1_timeline.csv, end_spread); measured 253.4 -> 251.1 us per step (tools/ar_tune.py qkv_rpw3)
      else if (a.N % 3 == 0 && (a.N / 3) % 1024 == 0) rpw = 3;
      else if (a.N >= 2048) rpw = 2;
    }
    if (a.rpw_override > 0) rpw = a.rpw_override;
    if (attn_pro) rpw = 1;
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

// Whether launch_gemv1 covers the out-proj GEMV with the PRO_ATTN_SELF prologue at this shape (K = N = d): the fused QKV +
// attention launch leaves the new token's own softmax term to that prologue, and skinny.hip has no such prologue -- the engine
// must not pick the fused launch where this is false.  Mirrors launch_gemv1 / g1_dispatch_nch / g1_dispatch_pe.
bool gemv1_attn_self_supports(int dtype, int d, int dh, int nsplit) {
  const int vec = dtype == DT_F32 ? 4 : 8, ch = 64 * vec;
  if (d <= 0 || dh <= 0 || d % ch != 0 || dh % vec != 0) return false;
  const int nch = d / ch;
  if (!(nch == 1 || nch == 2 || nch == 3 || nch == 4)) return false;  // the K = d family of g1_dispatch_pe
  if (!(nsplit == 4 || nsplit == 8 || (nsplit == 16 && nch * vec <= 16))) return false;
  if (g_g1_shared) {
    const int ept = d / G1_T;
    if (d % G1_T != 0 || ept < 1 || dh % ept != 0) return false;
    const int lpk = dh / ept;
    return lpk >= 1 && lpk <= 64 && (lpk & (lpk - 1)) == 0;
  }
  const int lpk = dh / vec;
  return lpk >= 1 && lpk <= 32 && (lpk & (lpk - 1)) == 0;
}

// returns 0 = launched, 1 = shape not covered (caller falls back to launch_skinny), < 0 = error
int launch_gemv1(hipStream_t st, int dtype, const SkinnyArgs& a) {
  if (a.B != 1 || a.N <= 0) return 1;
  if ((a.pro == PRO_ATTN || a.pro == PRO_ATTN_SELF) && (a.dh % (dtype == DT_F32 ? 4 : 8) != 0)) return 1;
  if ((a.pro == PRO_ATTN || a.pro == PRO_ATTN_SELF) && g_g1_shared) {  // a thread's K / 256 columns lie inside one head
    const int ept = a.K / G1_T;
    if (a.K % G1_T != 0 || ept < 1 || a.dh % ept != 0) return 1;
  }
  if (a.pro == PRO_ATTN_SELF) {  // a head = a power-of-two group of consecutive lanes (of the wave / of the workgroup's threads)
    const int lpk = g_g1_shared ? a.dh / (a.K / G1_T) : a.dh / (dtype == DT_F32 ? 4 : 8);
    if (lpk < 1 || lpk > (g_g1_shared ? 64 : 32) || (lpk & (lpk - 1)) != 0) return 1;
  }
  if (dtype == DT_FP8W) return !a.wscale ? -1 : a.temporal ? g1_dispatch_nch<bf16w8t_t>(st, a) : g1_dispatch_nch<bf16w8_t>(st, a);
  if (dtype == DT_F32) return g1_dispatch_nch<float>(st, a);
  return g1_dispatch_nch<bf16_t>(st, a);
}

// =====================================================================================================================
// LN1 + QKV projection + KV-cache write + decode attention of ONE utterance in ONE launch (kernels.h QkvAttnArgs).
//   reference: norm1 (valle/modules/transformer.py:296-297), in-proj (valle/modules/activation.py:414-421) and the last row
//   of F.multi_head_attention_forward under the prefix-LM mask (valle/models/valle.py:1019-1033) of one decoder layer.
// Why: at batch 1 the step is a chain of dependent launches of ~4 us each (2.4 us body + 1.75 us boundary);
// QKV GEMV -> decode attention is the one pair whose hand-off is HEAD-LOCAL, so it needs no grid-wide exchange:
//   * attention workgroup (head h, split s) recomputes q_h itself -- dh rows of W_q (128 KB at d = 1024), dh / 4 rows per
//     wave, the same wave-autonomous burst as gemv1_kernel; the nsplit workgroups of a head are placed on ONE XCD (block b
//     runs on XCD b % 8) so the rows come from HBM once and are shared through that XCD's L2;
//   * its first chunk of K / V of the cache is requested in the SAME burst (it does not depend on q), keys >= kv_len are
//     masked: the workgroup covers the OLD keys only;
//   * the K and V rows of the GEMV (2d rows) are the remaining workgroups of the launch: they write the cache slot kv_len
//     for later steps and k_new / v_new (fp32, rounded to the cache type) for THIS step;
//   * the new token's own softmax term (q . k_new, v_new) is a (head, one key) problem the out-proj GEMV's prologue solves
//     for free (gemv1_kernel PRO_ATTN_SELF: one more partial with l = 1).
// No workgroup waits for another: a kernel boundary less per layer (62 -> 50 launches per step at L = 12) and one HBM round
// trip instead of two on the critical path.
template <typename T, int NCH, int DH, int RPW, int NW, bool HO, int NK = 4>
__global__ __launch_bounds__(NW * 64) void qkv_attn1_kernel(QkvAttnArgs a) {
  constexpr int NT = NW * 64;  // 4 or 8 waves per workgroup
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CH = 64 * VEC;
  constexpr int K = NCH * CH;  // = d
  typedef typename G1W<T>::cache_t CT;
  constexpr int CVEC = Elem<CT>::VEC;  // cache elements per 16-byte vector
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long kt0 = ktrace_begin(a.kt);
  const T* W = reinterpret_cast<const T*>(a.w);

  // ---- common part of the burst: this thread's slice of the token's residual row and of the LayerNorm affine (the row is
  // normalised ONCE per workgroup and shared through LDS, g1_block_layernorm) -----------------------------------------------
  constexpr int EPT = K / NT;
  __shared__ __attribute__((aligned(16))) float sx[K];
  __shared__ float red[2 * NW];
  float xv[EPT], gv[EPT], bv[EPT];
  load_ept<EPT>(a.x + threadIdx.x * EPT, xv);
  load_ept<EPT>(a.gamma + threadIdx.x * EPT, gv);
  load_ept<EPT>(a.beta + threadIdx.x * EPT, bv);
  float x[NCH][VEC];

  if ((int)blockIdx.x >= a.n_attn) {
    // ================= K / V rows of the in-projection: rows [d, 3d), RPW per wave (gemv1_kernel's structure) =============
    const int wave = ((int)blockIdx.x - a.n_attn) * NW + w;
    const int row0 = (HO ? 0 : K) + wave * RPW;  // hand-off mode: the query rows are GEMV rows too
    const bool live = row0 < 3 * K;  // wave-uniform; dead waves still take part in the shared LayerNorm's barriers
    u32x4_t wv[RPW][NCH];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      int row = row0 + r;
      row = row < 3 * K ? row : 3 * K - 1;
      const T* wr = W + (int64_t)row * K + lane * VEC;
#pragma unroll
      for (int c = 0; c < NCH; ++c) wv[r][c] = G1W<T>::load(wr + c * CH);
    }
    const int myrow = row0 + lane;
    const bool writer = live && lane < RPW && myrow < 3 * K;
    float bias_v = 0.f, scale_v = 1.f;
    int kvl = 0, kvl_epoch = 0;
    if (writer) {
      if (a.bias) bias_v = a.bias[myrow];
      if constexpr (G1W<T>::kScaled) scale_v = a.wscale[myrow];
      kvl = a.kv_len[0];
      if constexpr (HO) kvl_epoch = a.epoch_ptr[0] + 1;
    }
    __builtin_amdgcn_sched_barrier(0);
    g1_block_layernorm<K, NT>(xv, gv, bv, sx, red);
    if (!live) return;
    g1_read_shared<T, NCH>(sx, x);
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const float t = wave_sum_dpp(g1_dot<T, NCH>(wv[r], x));
      mine = lane == r ? t : mine;
    }
    if (lane == 0) ktrace_end(a.kt, kt0, (int)blockIdx.x * NW + w);
    if (!writer) return;
    const float v = G1W<T>::kScaled ? fmaf(mine, scale_v, bias_v) : mine + bias_v;
    const int which = myrow / K, j = myrow - which * K;  // 0 = Q (hand-off mode), 1 = K, 2 = V  (valle/modules/activation.py:128-130)
    if constexpr (HO) {
      if (which == 0) {  // publish: ONE aligned 8-byte write-through store per value, tag = this step's epoch
        a.q_out[j] = v;
        const unsigned epoch = (unsigned)kvl_epoch;
        __hip_atomic_store(a.q_gran + j, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    const int h = j / DH, e = j - h * DH;
    CT* dst = reinterpret_cast<CT*>(which == 1 ? a.k_cache : a.v_cache) + ((int64_t)h * a.ctx_max + kvl) * DH + e;
    store_elem<CT>(dst, v);
    float vr = v;
    if constexpr (sizeof(CT) == 2) vr = bf16_to_f32(f32_to_bf16(v));  // what later steps will read back from the cache
    (which == 1 ? a.k_new : a.v_new)[j] = vr;
    return;
  }

  // ================= attention workgroup (head h, split s) ==============================================================
  const int NS = a.nsplit;
  int h, s;
  {
    const int ab = (int)blockIdx.x;
    if ((a.nhead & 7) == 0) {  // the NS splits of a head on one XCD (XCD = block index % 8)
      const int t = ab >> 3;
      s = t % NS;
      h = (ab & 7) + 8 * (t / NS);
    } else {
      h = ab / NS;
      s = ab - h * NS;
    }
  }
  constexpr int QRT = DH / NW;                                 // query rows per wave
  constexpr int QR = (QRT * NCH <= 32) ? QRT : (32 / NCH);     // ... per pass (<= 32 weight vectors in flight per lane)
  constexpr int QP = QRT / QR;
  static_assert(QRT % QR == 0 && QR >= 1 && QR <= 64, "query-row passes");
  constexpr int LPK = DH / CVEC;   // lanes per key
  constexpr int KPW = 64 / LPK;    // keys per wave-load
  constexpr int WCH = NK * KPW;    // keys per wave per round (NK = 4, or 8 with the q hand-off: no weight registers to share the file with)
  constexpr int CHUNK = NW * WCH;   // keys per workgroup per round
  static_assert(DH % CVEC == 0 && (LPK & (LPK - 1)) == 0 && LPK <= 32, "head size");
  __shared__ float sq[DH];
  __shared__ float sm_m[NW], sm_l[NW];
  __shared__ float sm_o[NW][DH];

  const int qrow0 = h * DH + w * QRT;  // this wave's first query row (= row of W: the Q block is rows [0, d))
  u32x4_t wv[QR][NCH];
  auto load_pass = [&](int p) {
#pragma unroll
    for (int r = 0; r < QR; ++r) {
      const T* wr = W + (int64_t)(qrow0 + p * QR + r) * K + lane * VEC;
#pragma unroll
      for (int c = 0; c < NCH; ++c) wv[r][c] = a.q_temporal ? G1W<T>::load_shared(wr + c * CH) : G1W<T>::load(wr + c * CH);
    }
  };
  const int slot = lane / LPK, part = lane % LPK;
  const CT* Kb = reinterpret_cast<const CT*>(a.k_cache) + (int64_t)h * a.ctx_max * DH + part * CVEC;
  const CT* Vb = reinterpret_cast<const CT*>(a.v_cache) + (int64_t)h * a.ctx_max * DH + part * CVEC;
  uint4 kraw[NK], vraw[NK];
  const int ctx_max = a.ctx_max;
  auto issue = [&](int base) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      int key = base + w * WCH + i * KPW + slot;
      key = key < ctx_max ? key : ctx_max - 1;
      kraw[i] = *reinterpret_cast<const uint4*>(Kb + (int64_t)key * DH);
      vraw[i] = *reinterpret_cast<const uint4*>(Vb + (int64_t)key * DH);
    }
  };
  auto widen = [&](const uint4& r, float (&f)[CVEC]) {
    if constexpr (sizeof(CT) == 4) {
      f[0] = __uint_as_float(r.x); f[1 % CVEC] = __uint_as_float(r.y); f[2 % CVEC] = __uint_as_float(r.z); f[3 % CVEC] = __uint_as_float(r.w);
    } else {
      f[0] = __uint_as_float(r.x << 16); f[1 % CVEC] = __uint_as_float(r.x & 0xffff0000u);
      f[2 % CVEC] = __uint_as_float(r.y << 16); f[3 % CVEC] = __uint_as_float(r.y & 0xffff0000u);
      f[4 % CVEC] = __uint_as_float(r.z << 16); f[5 % CVEC] = __uint_as_float(r.z & 0xffff0000u);
      f[6 % CVEC] = __uint_as_float(r.w << 16); f[7 % CVEC] = __uint_as_float(r.w & 0xffff0000u);
    }
  };

  // ---- the rest of the burst: (no hand-off: the first pass of W_q rows,) the first chunk of K / V, kv_len, the rows' bias / scale
  if constexpr (!HO) load_pass(0);
  int base = s * CHUNK;
  issue(base);
  const int ctx = a.kv_len[0];  // OLD keys only: slots [0, kv_len); slot kv_len is being written by the K / V workgroups
  unsigned epoch = 0;
  if constexpr (HO) epoch = (unsigned)(a.epoch_ptr[0] + 1);
  float qb[QP], qsc[QP];
#pragma unroll
  for (int p = 0; p < QP; ++p) {
    const int r = lane < QR ? lane : 0;
    qb[p] = a.bias ? a.bias[qrow0 + p * QR + r] : 0.f;
    qsc[p] = 1.f;
    if constexpr (G1W<T>::kScaled) qsc[p] = a.wscale[qrow0 + p * QR + r];
  }
  __builtin_amdgcn_sched_barrier(0);

  g1_block_layernorm<K, NT>(xv, gv, bv, sx, red);
  bool have_q = false;
  if constexpr (HO) {
    // poll this head's dh granules (one or two per lane of wave 0; relaxed agent-scope loads bypass L1): ready when every tag
    // carries this step's epoch.  ~1 us after the producing GEMV workgroups stored them; bounded -- a workgroup that gives up
    // computes the rows itself below.
    __shared__ int s_have_q;
    if (w == 0) {
      constexpr int GPL = (DH + 63) / 64;
      const unsigned long long* gq = a.q_gran + h * DH;
      bool ok = false;
      unsigned long long gx[GPL];
      for (unsigned spins = 0; spins < 40000u; ++spins) {
#pragma unroll
        for (int k = 0; k < GPL; ++k) gx[k] = __hip_atomic_load(gq + min(k * 64 + lane, DH - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool mine_ok = true;
#pragma unroll
        for (int k = 0; k < GPL; ++k) mine_ok &= (unsigned)(gx[k] >> 32) == epoch;
        if (__all(mine_ok)) {
          ok = true;
          break;
        }
      }
      if (ok) {
#pragma unroll
        for (int k = 0; k < GPL; ++k)
          if (k * 64 + lane < DH) sq[k * 64 + lane] = __uint_as_float((unsigned)gx[k]);
      }
      if (lane == 0) {
        s_have_q = ok ? 1 : 0;
        if (!ok && a.spin_fail) atomicAdd(a.spin_fail, 1u);
      }
    }
    __syncthreads();
    have_q = s_have_q != 0;
    if (!have_q) load_pass(0);  // fall back: block-uniform
  }
  if (!have_q) {
  g1_read_shared<T, NCH>(sx, x);
#pragma unroll
  for (int p = 0; p < QP; ++p) {
    float acc[QR];
#pragma unroll
    for (int r = 0; r < QR; ++r) acc[r] = g1_dot<T, NCH>(wv[r], x);
    if (p + 1 < QP) load_pass(p + 1);  // the registers are free again
    // (hand-off mode: this is the give-up path; it reduces row by row like the GEMV workgroups whose result it replaces, so that
    // a fallback cannot change a bit of q)
    if constexpr (!HO && (QR == 8 || QR == 16)) {
      const float tot = g1_rows_reduce<QR>(acc, lane);  // every lane: the total of row g1_rows_owner(lane)
      const int r = g1_rows_owner<QR>(lane);
      // bias / scale were loaded per lane for row `lane`: fetch the owner row's through the LDS crossbar (one bpermute each)
      const float qbr = __shfl(qb[p], r, 64), qscr = __shfl(qsc[p], r, 64);
      if (lane < QR) {  // lanes 0 .. QR-1 own each row exactly once (the bit-reversal is a permutation of them)
        const float qv1 = G1W<T>::kScaled ? fmaf(tot, qscr, qbr) : tot + qbr;
        sq[w * QRT + p * QR + r] = qv1;
        if (s == 0) a.q_out[qrow0 + p * QR + r] = qv1;
      }
    } else {
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < QR; ++r) {
        const float t = wave_sum_dpp(acc[r]);
        mine = lane == r ? t : mine;
      }
      if (lane < QR) {
        const float qv1 = G1W<T>::kScaled ? fmaf(mine, qsc[p], qb[p]) : mine + qb[p];
        sq[w * QRT + p * QR + lane] = qv1;
        if (s == 0) a.q_out[qrow0 + p * QR + lane] = qv1;
      }
    }
  }
  }
  const unsigned long long ktm1 = ktrace_mark(a.kt);  // this wave's query rows are done
  __syncthreads();
  float qv[CVEC];
#pragma unroll
  for (int j = 0; j < CVEC; ++j) qv[j] = sq[part * CVEC + j];
  const float scale = 1.0f / sqrtf((float)DH);

  float m = G1_NEG, l = 0.f, oacc[CVEC];
#pragma unroll
  for (int j = 0; j < CVEC; ++j) oacc[j] = 0.f;
  while (true) {
    float sc[NK];
    float mx = G1_NEG;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      float kf[CVEC];
      widen(kraw[i], kf);
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < CVEC; ++j) t = fmaf(qv[j], kf[j], t);
      t = head_group_sum(t, LPK) * scale;
      const int key = base + w * WCH + i * KPW + slot;
      sc[i] = key < ctx ? t : G1_NEG;
      mx = fmaxf(mx, sc[i]);
    }
    const float mn = fmaxf(m, wave_max_dpp(mx));  // wave-uniform running max
    const float f = __expf(m - mn);
    l *= f;
#pragma unroll
    for (int j = 0; j < CVEC; ++j) oacc[j] *= f;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int key = base + w * WCH + i * KPW + slot;
      const float p = key < ctx ? __expf(sc[i] - mn) : 0.f;
      l += p;
      float vf[CVEC];
      widen(vraw[i], vf);
#pragma unroll
      for (int j = 0; j < CVEC; ++j) oacc[j] = fmaf(p, vf[j], oacc[j]);
    }
    m = mn;
    base += NS * CHUNK;
    if (base >= ctx) break;  // block-uniform
    issue(base);
  }
  const unsigned long long ktm2 = ktrace_mark(a.kt);  // key loop done
  // ---- merge the KPW key slots of the wave (one running max per wave: plain sums), then the 4 waves through LDS -------------
  // lanes l ^ 8 by a DPP row rotation, l ^ 16 / l ^ 32 by the permlane swaps (common.h rows4_sum); ds_bpermute only where a
  // head is narrower than 8 lanes
#pragma unroll
  for (int o = LPK; o < 8; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int j = 0; j < CVEC; ++j) oacc[j] += __shfl_xor(oacc[j], o, 64);
  }
  if constexpr (LPK <= 8) {
    l += dpp_f32<0x128>(l);
#pragma unroll
    for (int j = 0; j < CVEC; ++j) oacc[j] += dpp_f32<0x128>(oacc[j]);
  }
  if constexpr (LPK <= 16) {
    l = rows4_sum(l);
#pragma unroll
    for (int j = 0; j < CVEC; ++j) oacc[j] = rows4_sum(oacc[j]);
  } else {  // LPK == 32
    l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int j = 0; j < CVEC; ++j) oacc[j] += __shfl_xor(oacc[j], 32, 64);
  }
  if (slot == 0) {
    if (part == 0) {
      sm_m[w] = m;
      sm_l[w] = l;
    }
#pragma unroll
    for (int j = 0; j < CVEC; ++j) sm_o[w][part * CVEC + j] = oacc[j];
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (lane == 0) ktrace_end(a.kt, kt0, (int)blockIdx.x * NW + w, ktm1, ktm2);
  if (tid < DH || tid == NT - 1) {  // DH <= 128 < NT - 1
    float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
    if constexpr (NW == 8) M = fmaxf(M, fmaxf(fmaxf(sm_m[4], sm_m[5]), fmaxf(sm_m[6], sm_m[7])));
    float f[NW];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) f[ww] = __expf(sm_m[ww] - M);
    if (tid < DH) {
      float o = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) o = fmaf(sm_o[ww][tid], f[ww], o);
      a.part_o[(int64_t)s * K + h * DH + tid] = o;
    } else {
      float L = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) L = fmaf(sm_l[ww], f[ww], L);
      float* ml = a.part_ml + ((int64_t)h * NS + s) * 2;
      ml[0] = M;
      ml[1] = L;
    }
  }
}

bool qkv_attn1_supports(int dtype, int d, int nhead, int dh) {
  const int vec = dtype == DT_F32 ? 4 : 8, ch = 64 * vec;
  if (d <= 0 || nhead <= 0 || nhead * dh != d || d % ch != 0) return false;
  const int nch = d / ch;
  if (!(nch == 1 || nch == 2 || nch == 4)) return false;
  return dh == 64 || dh == 128;
}

template <typename T, int NCH, int DH, int NW>
static int qa_launch(hipStream_t st, const QkvAttnArgs& a) {
  constexpr int RPW = NCH <= 2 ? 4 : 2;
  QkvAttnArgs b = a;
  b.n_attn = a.nhead * a.nsplit;
  const bool ho = a.q_gran != nullptr && a.epoch_ptr != nullptr;
  const int gemv_waves = ((ho ? 3 : 2) * a.d + RPW - 1) / RPW;  // hand-off: the query rows are GEMV rows of the launch too
  const dim3 grid(b.n_attn + (gemv_waves + NW - 1) / NW), block(NW * 64);
  if (ho && a.nk == 8) hipLaunchKernelGGL((qkv_attn1_kernel<T, NCH, DH, RPW, NW, true, 8>), grid, block, 0, st, b);
  else if (ho && a.nk == 2) hipLaunchKernelGGL((qkv_attn1_kernel<T, NCH, DH, RPW, NW, true, 2>), grid, block, 0, st, b);  // the persistent step's split (persist.hip)
  else if (ho) hipLaunchKernelGGL((qkv_attn1_kernel<T, NCH, DH, RPW, NW, true>), grid, block, 0, st, b);
  else hipLaunchKernelGGL((qkv_attn1_kernel<T, NCH, DH, RPW, NW, false>), grid, block, 0, st, b);
  return 0;
}

int g_qa_waves = 4;  // "qa_waves": waves per workgroup of the fused launch (4 or 8).  Measured (MI355X, C2): 4 waves 215.9 us/step, 8 waves
                     // (two per SIMD, half the query rows each) 224.5 -- the 512-thread workgroups start later and their barriers cost more than the overlap buys

template <typename T, int NCH, int DH>
static int qa_pick(hipStream_t st, const QkvAttnArgs& a) {
  if constexpr ((NCH * 64 * Elem<T>::VEC) % 512 == 0) {  // 8 waves: every thread still owns >= 1 element of the shared row
    if (g_qa_waves == 8) return qa_launch<T, NCH, DH, 8>(st, a);
  }
  return qa_launch<T, NCH, DH, 4>(st, a);
}

template <typename T>
static int qa_dispatch(hipStream_t st, const QkvAttnArgs& a) {
  constexpr int CH = 64 * Elem<T>::VEC;
  const int key = (a.d / CH) * 1000 + a.dh;
  switch (key) {
    case 1064: return qa_pick<T, 1, 64>(st, a);
    case 2064: return qa_pick<T, 2, 64>(st, a);
    case 4064: return qa_pick<T, 4, 64>(st, a);
    case 1128: return qa_pick<T, 1, 128>(st, a);
    case 2128: return qa_pick<T, 2, 128>(st, a);
    case 4128: return qa_pick<T, 4, 128>(st, a);
    default: return 1;
  }
}

// returns 0 = launched, 1 = shape not covered (caller launches the QKV GEMV and the decode attention separately), < 0 = error
int launch_qkv_attn1(hipStream_t st, int dtype, const QkvAttnArgs& a) {
  if (!qkv_attn1_supports(dtype, a.d, a.nhead, a.dh)) return 1;
  if (!(a.nsplit == 4 || a.nsplit == 8 || a.nsplit == 16)) return 1;
  if (!a.w || !a.x || !a.gamma || !a.beta || !a.q_out || !a.k_new || !a.v_new || !a.k_cache || !a.v_cache || !a.kv_len || !a.part_o || !a.part_ml)
    return -1;
  if (dtype == DT_FP8W) return !a.wscale ? -1 : a.temporal ? qa_dispatch<bf16w8t_t>(st, a) : qa_dispatch<bf16w8_t>(st, a);
  if (dtype == DT_F32) return qa_dispatch<float>(st, a);
  return qa_dispatch<bf16_t>(st, a);
}

}  // namespace vle
