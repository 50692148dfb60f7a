// The AR decode loop of 2 .. 6 utterances as ONE persistent launch (round 6; option "persist_batch").
//   reference: the AR loop of VALLE.inference (valle/models/valle.py:1012-1057) for every utterance of the batch -- the L pre-norm
//   decoder layers (valle/modules/transformer.py:296-302, 332-334; attention valle/modules/activation.py:408-427), the final norm
//   and ar_predict_layer (valle.py:1035-1039), topk_sampling (:1287-1302), the stop rule (:1044-1048) and the next token's
//   embedding + position (:1013-1015, :1057).
//
// Why.  On the batched launch chain 2 .. 8 utterances cost 330-370 us per AR step whatever the batch (62 dependent launches; DESIGN.md
// 4.2, profiles/r06_small_batch.json): two utterances there are SLOWER than one on pstep_kernel (persist.hip, 128 us per step).
// pstep_kernel's step is hand-off-latency-bound -- per layer 3.3 us of arithmetic against 7.1 us of waiting on six edges -- and its
// weights sit in registers when an operator's input arrives.  This kernel keeps that launch exactly (same grid, same ownership of
// weight rows, same six edges per layer, same request schedule) and lets every edge carry NB activation rows: the weights are
// requested ONCE per step and multiplied with NB rows, the NB rows of an edge travel in the same sweep, and the waiting time is paid
// once per step instead of once per utterance.  What grows with NB is the arithmetic between the hand-offs (dot products, LayerNorm
// statistics, the attention share over each utterance's own cached keys, the draw).
//
// Per utterance the arithmetic is pstep_kernel's default form (bf16 weights, PK 13: hidden row as bf16 pairs, folded LayerNorm, bf16
// activation rows on v_dot2c, XCD-local head-group edges; 2 keys per lane; request schedule 3) instruction for instruction on the
// same lane <-> element mapping, so every logit and token of utterance b is BIT-IDENTICAL to a one-utterance launch of that form on
// utterance b alone (tests/test_persist_gpu.py::test_batched_persistent_launch_is_bit_identical_to_one_utterance_launches).
// Utterances that have stopped stay in the launch (their rows keep flowing with a frozen cache slot, nothing of theirs is stored);
// the launch ends when all have stopped.  Spins are bounded as in persist.hip (PStepArgs::fail -> VLE_EBUSY).
#include "persist_dev.h"
#include <type_traits>

namespace vle {

namespace {

// NB rows of NV consecutive granules each (16-byte loads; row b at byte offset off + b * stride), all in the same pass
template <int NB, int NV>
__device__ inline void gather_rows16(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned stride, unsigned epoch, float (&v)[NB][NV], PsSpin& sp) {
  static_assert(NV % 2 == 0, "pairs of granules");
  sp.passes = 0;
  for (;;) {
    u32x4_t raw[NB][NV / 2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < NV / 2; ++k) raw[b][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + b * stride + 16 * k), 0, 16 /* sc1 */);
    bool ok = true;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < NV / 2; ++k) {
        ok &= raw[b][k].y == epoch && raw[b][k].w == epoch;
        v[b][2 * k] = __uint_as_float(raw[b][k].x);
        v[b][2 * k + 1] = __uint_as_float(raw[b][k].z);
      }
    ++sp.passes;
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
}

// the same + ONE more granule per row at byte offset off1 + b * stride (the same for every lane)
template <int NB, int NV>
__device__ inline void gather_rows16_plus1(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned off1, unsigned stride, unsigned epoch, float (&v)[NB][NV],
                                           float (&v1)[NB], PsSpin& sp) {
  static_assert(NV % 2 == 0, "pairs of granules");
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  sp.passes = 0;
  for (;;) {
    u32x4_t raw[NB][NV / 2];
    u32x2_t r1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int k = 0; k < NV / 2; ++k) raw[b][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + b * stride + 16 * k), 0, 16 /* sc1 */);
      r1[b] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(off1 + b * stride), 0, 16 /* sc1 */);
    }
    bool ok = true;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      ok &= r1[b].y == epoch;
      v1[b] = __uint_as_float(r1[b].x);
#pragma unroll
      for (int k = 0; k < NV / 2; ++k) {
        ok &= raw[b][k].y == epoch && raw[b][k].w == epoch;
        v[b][2 * k] = __uint_as_float(raw[b][k].x);
        v[b][2 * k + 1] = __uint_as_float(raw[b][k].z);
      }
    }
    ++sp.passes;
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
}

// one granule per lane and row from the XCD-local copy (gl) or the write-through copy (g), rows `stride` granules apart:
// PS_LOCAL_TRIES passes on the local copies, one on the others, and so on (gather_one_dual of persist_dev.h for NB rows)
template <int NB>
__device__ inline void gather_one_dual_rows(const gran_t* g, const gran_t* gl, int stride, unsigned epoch, float (&v)[NB], PsSpin& sp) {
  sp.passes = 0;
  for (;;) {
    const bool local = (sp.passes % (PS_LOCAL_TRIES + 1)) != PS_LOCAL_TRIES;
    const gran_t* src = local ? gl : g;
    gran_t raw[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) raw[b] = gran_load(src + (size_t)b * stride);
    bool ok = true;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      ok &= (unsigned)(raw[b] >> 32) == epoch;
      v[b] = __uint_as_float((unsigned)raw[b]);
    }
    ++sp.passes;
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
}

// body(integral_constant<int, 0>) ... body(integral_constant<int, N - 1>): a loop over the utterances that is unrolled whatever its body
// holds (a `#pragma unroll` loop around barriers and block-uniform branches was left rolled for 3 / 4 utterances: the per-utterance
// state arrays then became dynamically indexed and were moved to LDS)
template <int N, typename F>
__device__ inline void for_each_utt(F&& body) {
  if constexpr (N > 0) {
    for_each_utt<N - 1>(body);
    body(std::integral_constant<int, N - 1>{});
  }
}

// The wave totals of persist_dev.h for N independent values IN LOCKSTEP: between two hand-offs a workgroup runs one wave per SIMD, so
// every DPP / permlane step of a reduction waits out its own latency -- N reductions written one after the other cost N times that.
// Step k of all N values is issued before step k + 1 of any; per value the arithmetic is row16_sum_dpp's / rows4_sum's.
template <int N>
__device__ inline void row16_sums_lockstep(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_f32<0xB1>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_f32<0x4E>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_f32<0x141>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_f32<0x140>(v[i]);
}
template <int N>
__device__ inline void rows4_sums_lockstep(float (&v)[N]) {
  permlane_u32x2 r[N];
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __uint_as_float(r[i][0]) + __uint_as_float(r[i][1]);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __uint_as_float(r[i][0]) + __uint_as_float(r[i][1]);
}
// (lane & 15) == c, on an opaque copy of the lane id (see sel_by_lane)
__device__ inline bool lane_col_is(int c) {
  int x = threadIdx.x & 15;
  asm volatile("" : "+v"(x));
  return x == c;
}
// N <= 16 wave totals at once: value i's 16-lane row sums are kept in lane column i of every row, ONE pair of permlane swaps finishes all
// of them -- lanes i, i + 16, i + 32, i + 48 end up with total i (per value: rows4_sum(row16_sum_dpp(v)), persist_dev.h's arithmetic)
template <int N>
__device__ inline float wave_sums_by_column(float (&v)[N]) {
  static_assert(N <= 16, "one lane column per value");
  row16_sums_lockstep<N>(v);
  float r = v[0];
#pragma unroll
  for (int i = 1; i < N; ++i) r = lane_col_is(i) ? v[i] : r;
  return rows4_sum(r);
}

// 4 rows of NB <= 8 utterances (value 4 b + r): lane 4 b + r ends up with its total -- utterances 0 .. 3 through the 16 lane columns of
// one reduction, utterances 4 .. through a second one whose totals the lanes 16 .. 31 keep
template <int NB>
__device__ inline float wave_sums_rows4(float (&v)[4 * NB]) {
  if constexpr (NB <= 4) {
    return wave_sums_by_column<4 * NB>(v);
  } else {
    float a[16], b[4 * NB - 16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v[i];
#pragma unroll
    for (int i = 16; i < 4 * NB; ++i) b[i - 16] = v[i];
    const float ra = wave_sums_by_column<16>(a);
    const float rb = wave_sums_by_column<4 * NB - 16>(b);
    int hi = (threadIdx.x & 63) >> 4;
    asm volatile("" : "+v"(hi));
    return hi == 1 ? rb : ra;
  }
}

// value of utterance (lane >> 2) out of per-utterance values (lanes >= 4 NB: utterance 0's)
template <int NB, typename X>
__device__ inline X sel_by_lane(const X (&v)[NB]) {
  X r = v[0];
#pragma unroll
  for (int b = 1; b < NB; ++b) {
    // (an opaque copy of the lane's utterance per comparison: hipcc turns a chain of selects on ONE index into a dynamically indexed
    //  private array, which it then moves to LDS -- 5 arrays of NB words per thread for 3 / 4 utterances)
    int lbx = (threadIdx.x & 63) >> 2;
    asm volatile("" : "+v"(lbx));
    r = lbx == b ? v[b] : r;
  }
  return r;
}

}  // namespace

template <int NB, bool TR = false>
__global__ __launch_bounds__(PS_T) void pstepb_kernel(PStepArgs a) {
  typedef bf16_t T;
  constexpr int D = 1024, H = 16, NK = 2;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CH = 64 * VEC;
  constexpr int NCH = D / CH;        // K = d
  constexpr int NCH2 = 4 * D / CH;   // K = 4d
  constexpr int NWG = 256;
  constexpr int DH = D / H;
  constexpr int NS = NWG / H;        // key splits per head = workgroups per head
  constexpr int QR = DH / NS;        // rows of each of Q, K, V this workgroup projects
  constexpr int RQ = 3 * QR / 4;     // in-projection rows per wave
  constexpr int R1 = 4 * D / NWG / 4;  // linear1 rows per wave
  constexpr int EPT = D / PS_T;      // elements of a d-vector per thread
  constexpr int EPT2 = 4 * D / PS_T; // ... of the hidden vector
  static_assert(NB >= 2 && NB <= 6, "lane 4 b + r carries row r of utterance b (lanes 0 .. 4 NB - 1); wave w merges the attention partials of utterances w and w + 4");
  static_assert(VEC == 8 && NCH == 2 && NCH2 == 8 && DH == 64 && NS == 16 && QR == 4 && RQ == 3 && R1 == 4 && EPT == 4 && EPT2 == 16, "shape");
  typedef bf16_t CT;
  constexpr int CVEC = 8;
  constexpr int LPK = DH / CVEC, KPW = 64 / LPK, WCH = NK * KPW, CHUNK = 4 * WCH;
  typedef unsigned u32x2v_t __attribute__((ext_vector_type(2)));

  // ---- LDS: one array (> 80 KB: one workgroup per CU) -------------------------------------------------------------------------
  constexpr int SM_FLOATS = 21 * 1024;
  __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
  constexpr int SXF = 2 * 1024;   // floats per utterance's input row (4 d bf16 values)
  constexpr int UBF = 576;        // floats of an utterance's small arrays
  auto sxrow = [&](int b) { return smem + b * SXF; };                  // the operator's input row of utterance b (bf16)
  constexpr int SAMP0 = NB * SXF;  // the sampling step's row [V] and scratch words (3 K floats), then the utterances' small arrays
  auto ub = [&](int b) { return smem + SAMP0 + 3 * 1024 + b * UBF; };  // small arrays of utterance b:
  constexpr int O_RED = 0, O_SQ = 8, O_SK = O_SQ + DH, O_SV = O_SK + DH, O_SMM = O_SV + DH, O_SML = O_SMM + 4, O_SMO = O_SML + 4, O_SPM = O_SMO + 4 * DH,
                O_SPL = O_SPM + NS, O_SPO = O_SPL + NS, O_SRES = O_SPO + NS * QR;
  static_assert(O_SRES + 4 <= UBF && SAMP0 + 3 * 1024 + NB * UBF <= SM_FLOATS, "LDS carve");

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = (int)blockIdx.x;
  const int lb = lane >> 2, lr = lane & 3;  // results of a wave's rows sit in lane 4 b + r: utterance b, row r
  // the NS workgroups of a head on ONE XCD (block b runs on XCD b % 8: speed only)
  const int jj = c >> 3;
  const int h = (c & 7) * (H / 8) + jj / NS, s = jj % NS;

  unsigned live = 0u;  // bit b: utterance b has not stopped
  int itb[NB];         // AR iteration of utterance b: its Philox counter and its row of the logits trace
  int it = 0;          // the launch's step counter: epoch of the granules = it + 1
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    itb[b] = a.iter[b];
    if (!a.done[b]) {
      live |= 1u << b;
      it = itb[b];  // (batch calls: the same for every utterance still running -- they step together from the prefill)
    }
  }
  if (a.epoch_ctr != nullptr) it = a.epoch_ctr[0];  // slot mode: the slots' iteration counters differ, the epochs follow a counter of their own
  if (live == 0u) {  // every utterance has stopped: the remaining launches of the host's queue are no-ops that still report
    if (a.nsteps > 0 && c == 0 && tid == 0) {
      const PStepSample q = ps_sample_load(a.smp);
      if (q.host_prog != nullptr) {
        const int sc = q.s.done_count[1] + a.nsteps;
        q.s.done_count[1] = sc;
        __hip_atomic_store(q.host_prog + 0, q.s.done_count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(q.host_prog + 1, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (a.fail != nullptr && __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // (persist.hip: an earlier launch gave up)
  if (c == 0 && tid == 0 && a.never) smem[SM_FLOATS - 1] = 0.f;  // keeps the whole array allocated

  unsigned epoch = (unsigned)(it + 1);
  int kvl[NB];  // slot of the new token of utterance b; its old keys are [0, kvl[b])
#pragma unroll
  for (int b = 0; b < NB; ++b) kvl[b] = a.kv_len[b];
  const int ctx_max = a.ctx_max;
  const int naps = a.naps;
  const int nap_att = naps & 15, nap_x = (naps >> 4) & 15, nap_x2 = (naps >> 8) & 15, nap_hid = (naps >> 12) & 15, nap_qkv = (naps >> 16) & 15,
            nap_part = (naps >> 20) & 15;
  PsSpin sp{PS_SPINS, a.fail, (a.mode >> 8) & 15, 0u};
  PsTrace pt{nullptr, 0, 0ull};  // in-kernel timeline (option "persist_trace"; persist_dev.h): the same stamps as pstep_kernel's
  if constexpr (TR) pt.p = (a.ptrace && tid == 0) ? a.ptrace + ((size_t)(it & 7) * NWG + c) * PS_PT_SLOTS : nullptr;
  pt_begin<TR>(pt);
  pt_end<TR>(pt, 0u);

  const int GPL1 = ps_gran_per_layer(D, H, NS);  // per utterance
  const int GPL = NB * GPL1;
  const gran_t* const GB = a.gran;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gran, 0, (int)((size_t)(a.L + 1) * GPL * sizeof(gran_t)), 0x00020000);
  // offsets inside a layer's granules: every edge holds NB rows, row b at + b * (the edge's length)
  constexpr int L_PART = H * NS * (2 + DH);
  constexpr int G_X = 0, G_QKV = NB * D, G_PART = NB * 4 * D, G_ATT = G_PART + NB * L_PART, G_X2 = G_ATT + NB * D, G_HID = G_X2 + NB * D;
  constexpr int G_QKVL = G_HID + NB * 4 * D, G_PARTL = G_QKVL + NB * 3 * D;  // XCD-local copies of the two head-group edges
  auto goff = [&](const gran_t* g) { return (unsigned)((const char*)g - (const char*)GB); };

  // ---- register-resident operands, requested ahead (ONE set for all utterances) -------------------------------------------------
  u32x4_t wq[RQ][NCH], wo[NCH], w1[R1][NCH], w2[NCH2];
  float bq = 0.f, bo_v = 0.f, b1_v = 0.f, b2_v = 0.f;
  float sgq = 0.f, sg1_v = 0.f, sgx = 0.f, tbx = 0.f;
  float g1v[EPT], g2v[EPT];
  u32x4_t kraw[NB][NK], vraw[NB][NK];

  const int slot = lane / LPK, part = lane % LPK;
  auto qkv_row = [&](int r) { return (r / QR) * D + h * DH + s * QR + (r % QR); };
  auto wvec = [&](unsigned long long W, int64_t row, int KK, int cc) {
    return ps_load_nt(reinterpret_cast<const u32x4_t PS_GLOBAL*>(as_g<T>(W) + row * KK + lane * VEC + cc * CH));
  };
  // (straight-line requests on selected addresses: persist.hip)
  const bool extra_row = (c == 0 && w == 0 && 4 * NWG < a.V);  // wave-uniform: row 4 * 256 (the EOS row at V = 1025)
  auto issue_wqkv = [&](const PsLayer& p, bool pred) {
    const unsigned long long W = p.wqkv;
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int64_t prow = (r == 1 && extra_row) ? 4 * NWG : 4 * c + w;
      const int64_t row = pred ? prow : (int64_t)qkv_row(w * RQ + r);
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) wq[r][cc] = wvec(W, row, D, cc);
    }
    // the row's (sg, tb): in-projection row (lane & 3) of this wave, or the predict layer's row 4c + w (and row 1024 for its one wave)
    const int rq = qkv_row(w * RQ + (lr < RQ ? lr : RQ - 1));
    const int ri = pred ? 4 * c + w : rq;
    sgq = as_g<float>(p.sgqkv)[ri];
    bq = as_g<float>(p.tbqkv)[ri];
    sgx = as_g<float>(p.sgqkv)[pred ? 4 * NWG : 0];
    tbx = as_g<float>(p.tbqkv)[pred ? 4 * NWG : 0];
    ps_load4(as_g<float>(p.g1) + tid * EPT, g1v);
  };
  // (cache rows inside a multi-step launch: persist.hip issue_kv -- lanes beyond the valid length re-read row nvalid - 1)
  auto issue_kv = [&](const PsLayer& p, int b, int base, int nvalid) {
    const CT PS_GLOBAL* Kb = as_g<CT>(p.kc) + ((int64_t)b * H + h) * ctx_max * DH + part * CVEC;
    const CT PS_GLOBAL* Vb = as_g<CT>(p.vc) + ((int64_t)b * H + h) * ctx_max * DH + part * CVEC;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      int key = base + w * WCH + i * KPW + slot;
      key = key < nvalid ? key : nvalid - 1;
      key = key > 0 ? key : 0;
      kraw[b][i] = *reinterpret_cast<const u32x4_t PS_GLOBAL*>(Kb + (int64_t)key * DH);
      vraw[b][i] = *reinterpret_cast<const u32x4_t PS_GLOBAL*>(Vb + (int64_t)key * DH);
    }
  };
  auto issue_kv_all = [&](const PsLayer& p) {
#pragma unroll
    for (int b = 0; b < NB; ++b) issue_kv(p, b, s * CHUNK, kvl[b]);
  };
  auto issue_wo = [&](const PsLayer& p) {
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) wo[cc] = wvec(p.wo, 4 * c + w, D, cc);
    bo_v = as_g<float>(p.bo)[4 * c + w];
  };
  auto issue_w1_rows = [&](const PsLayer& p, int r0, int r1) {
#pragma unroll
    for (int r = 0; r < R1; ++r) {
      if (r < r0 || r >= r1) continue;
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) w1[r][cc] = wvec(p.w1, 4 * R1 * c + w * R1 + r, D, cc);
    }
    if (r0 == 0) {
      const int r1i = 4 * R1 * c + w * R1 + lr;
      sg1_v = as_g<float>(p.sg1)[r1i];
      b1_v = as_g<float>(p.tb1)[r1i];
      ps_load4(as_g<float>(p.g2) + tid * EPT, g2v);
    }
  };
  auto issue_w2_chunks = [&](const PsLayer& p, int c0, int c1) {
#pragma unroll
    for (int cc = 0; cc < NCH2; ++cc) {
      if (cc < c0 || cc >= c1) continue;
      w2[cc] = wvec(p.w2, 4 * c + w, 4 * D, cc);
    }
    if (c0 == 0) b2_v = as_g<float>(p.b2)[4 * c + w];
  };
  // ---- x of the first layer: the sampling kernel's output (previous launch), one row per utterance ------------------------------
  float xv[NB][EPT];
#pragma unroll
  for (int b = 0; b < NB; ++b) load_ept<EPT>(a.x_in + (size_t)b * D + tid * EPT, xv[b]);
  {
    const PsLayer p0 = ps_layer(a.layers, 0);
    issue_wqkv(p0, false);
    issue_kv_all(p0);
  }
  __builtin_amdgcn_sched_barrier(0);
  auto nap = [&](int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
  };
  // folded LayerNorm, the row side, for the NB rows (persist.hip fold_stats per row; ONE barrier for all of them)
  auto fold_stats = [&](const float (&xr)[NB][EPT], const float (&gv)[EPT], float (&mean)[NB], float (&rstd)[NB]) {
    float sw[NB], qw[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float xg[EPT];
#pragma unroll
      for (int k = 0; k < EPT; ++k) xg[k] = xr[b][k] * gv[k];
      *reinterpret_cast<u32x2v_t*>(reinterpret_cast<unsigned char*>(sxrow(b)) + tid * 8) = u32x2v_t{pack_bf16x2(xg[0], xg[1]), pack_bf16x2(xg[2], xg[3])};
      sw[b] = 0.f;
#pragma unroll
      for (int k = 0; k < EPT; ++k) sw[b] += xr[b][k];
    }
    row16_sums_lockstep<NB>(sw);
    rows4_sums_lockstep<NB>(sw);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      sw[b] *= (1.0f / (64.0f * EPT));  // the wave's mean
      qw[b] = 0.f;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const float t = xr[b][k] - sw[b];
        qw[b] = fmaf(t, t, qw[b]);
      }
    }
    row16_sums_lockstep<NB>(qw);
    rows4_sums_lockstep<NB>(qw);
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float* red = ub(b) + O_RED;
        red[w] = sw[b];
        red[4 + w] = qw[b];
      }
    }
    g1_lds_barrier();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* red = ub(b) + O_RED;
      mean[b] = ((red[0] + red[1]) + (red[2] + red[3])) * 0.25f;
      float m2 = (red[4] + red[5]) + (red[6] + red[7]);
      float dm = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = red[i] - mean[b];
        dm = fmaf(t, t, dm);
      }
      m2 = fmaf(dm, 64.0f * EPT, m2);
      rstd[b] = __builtin_amdgcn_rsqf(fmaf(m2, 1.0f / (float)D, LN_EPS));
    }
  };

  // ---- sampling inside the launch: the utterances' state, carried in registers by every workgroup -------------------------------
  int n_gen[NB], apos[NB], s_cap[NB];
  int dcount = 0;  // utterances done so far (workgroup 0's thread 0 keeps ArState::done_count)
  {
    const PStepSample q = ps_sample_load(a.smp);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      n_gen[b] = q.s.n_gen[b];
      apos[b] = q.s.audio_pos[b];
      s_cap[b] = q.s.cap[b];
    }
    dcount = q.s.done_count[0];
  }
  const int nsteps = a.nsteps;

  for (int step = 0; step < nsteps; ++step) {
  if (step > 0) {
    sp.budget = sp.budget ? PS_SPINS : 0u;
    if constexpr (TR) pt = PsTrace{(a.ptrace && tid == 0) ? a.ptrace + ((size_t)(it & 7) * NWG + c) * PS_PT_SLOTS : nullptr, 0, 0ull};
    pt_begin<TR>(pt);
    pt_end<TR>(pt, 0u);
  }
  for (int l = 0; l < a.L; ++l) {
    const PsLayer p = ps_layer(a.layers, l);
    const PsLayer pn = ps_layer(a.layers, l + 1);
    gran_t* const G = a.gran + (size_t)l * GPL;
    const bool last = l + 1 == a.L;

    // ======== (1) LN1 + in-projection of this head's 3 QR rows, NB input rows ====================================================
    if (l > 0) {
      pt_begin<TR>(pt);
      nap(nap_x);
      gather_rows16<NB, EPT>(rs, goff(G + G_X + tid * EPT), D * 8u, epoch, xv, sp);
      pt_end<TR>(pt, sp.passes);
    }
    if (tid == c) {
#pragma unroll
      for (int b = 0; b < NB; ++b) store_ept_lds<EPT>(ub(b) + O_SRES, xv[b]);
    }
    float kv_new = 0.f;  // lane 4 b + r, r < RQ: this lane's K or V element of utterance b's new token
    {
      float ln_mean[NB], ln_rstd[NB];
      fold_stats(xv, g1v, ln_mean, ln_rstd);
      float t[4 * NB];  // value 4 b + r: row r of utterance b (r = 3: unused)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sxrow(b), xb);
#pragma unroll
        for (int r = 0; r < RQ; ++r) t[4 * b + r] = ps_dot_bf16<NCH>(wq[r], xb);
        t[4 * b + 3] = 0.f;
      }
      const float mine = wave_sums_rows4<NB>(t);  // lane 4 b + r: row r of utterance b
      const float mean_l = sel_by_lane<NB>(ln_mean), rstd_l = sel_by_lane<NB>(ln_rstd);
      if (lane < 4 * NB && lr < RQ) {
        const int r = w * RQ + lr, which = r / QR, e = s * QR + (r % QR);  // e: element of the head
        const float v = fmaf(rstd_l, fmaf(-mean_l, sgq, mine), bq);
        gran_t* gq = G + G_QKV + lb * (3 * D) + h * (3 * DH) + which * DH + e;
        gran_t* gql = G + G_QKVL + lb * (3 * D) + h * (3 * DH) + which * DH + e;
        if (which == 0) {
          gran_store(gq, epoch, v);
          gran_store_local(gql, epoch, __float_as_uint(v));
        } else {
          kv_new = v;
          const float vr = bf16_to_f32(f32_to_bf16(v));  // what later steps will read back from the cache
          gran_store(gq, epoch, vr);
          gran_store_local(gql, epoch, __float_as_uint(vr));
        }
      }
    }
    issue_kv_all(p);

    // ======== (2) q, k_new, v_new of the head; attention over this workgroup's share of each utterance's cached keys ==============
    pt_begin<TR>(pt);
    {
      nap(nap_qkv);
      const int wq_i = w < 3 ? w : 0;
      const int gi = h * (3 * DH) + wq_i * DH + (lane < DH ? lane : 0);
      float t[NB];
      gather_one_dual_rows<NB>(G + G_QKV + gi, G + G_QKVL + gi, 3 * D, epoch, t, sp);
      if (w < 3 && lane < DH) {
#pragma unroll
        for (int b = 0; b < NB; ++b) (ub(b) + (w == 0 ? O_SQ : w == 1 ? O_SK : O_SV))[lane] = t[b];
      }
    }
    g1_lds_barrier();
    pt_end<TR>(pt, sp.passes);
    __builtin_amdgcn_sched_barrier(0);
    {
      auto widen = [&](const u32x4_t& r, float (&f)[CVEC]) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
      };
      const float scale = 1.0f / sqrtf((float)DH);
      // The NB attention shares IN LOCKSTEP (see row16_sums_lockstep): step k of every utterance's chain -- scores, head sums, running
      // maximum, exponentials, weighted values -- is issued before step k + 1 of any.  Per utterance: pstep_kernel's arithmetic.
      float qv[NB][CVEC], m[NB], lsum[NB], oacc[NB][CVEC];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float* const u = ub(b);
#pragma unroll
        for (int j = 0; j < CVEC; ++j) qv[b][j] = u[O_SQ + part * CVEC + j];
        m[b] = G1_NEG;
        lsum[b] = 0.f;
#pragma unroll
        for (int j = 0; j < CVEC; ++j) oacc[b][j] = 0.f;
      }
      // one round = this workgroup's CHUNK keys at `base` of the utterances B0 .. B0 + NU - 1 (their K / V rows are in kraw / vraw)
      auto attn_round = [&](auto b0c, auto nuc, int base) {
        constexpr int B0 = decltype(b0c)::value, NU = decltype(nuc)::value;
        float sc[NU][NK], mx[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < NK; ++i) {
            float kf[CVEC];
            widen(kraw[B0 + u][i], kf);
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < CVEC; ++j) t = fmaf(qv[B0 + u][j], kf[j], t);
            sc[u][i] = t;
          }
        static_assert(LPK == 8, "head_group_sum's three steps");
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < NK; ++i) sc[u][i] += dpp_f32<0xB1>(sc[u][i]);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < NK; ++i) sc[u][i] += dpp_f32<0x4E>(sc[u][i]);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < NK; ++i) sc[u][i] += dpp_f32<0x141>(sc[u][i]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          mx[u] = G1_NEG;
#pragma unroll
          for (int i = 0; i < NK; ++i) {
            const int key = base + w * WCH + i * KPW + slot;
            sc[u][i] = key < kvl[B0 + u] ? sc[u][i] * scale : G1_NEG;
            mx[u] = fmaxf(mx[u], sc[u][i]);
          }
        }
        // wave_max_dpp of every utterance, step by step
#pragma unroll
        for (int u = 0; u < NU; ++u) mx[u] = fmaxf(mx[u], dpp_f32<0xB1>(mx[u]));
#pragma unroll
        for (int u = 0; u < NU; ++u) mx[u] = fmaxf(mx[u], dpp_f32<0x4E>(mx[u]));
#pragma unroll
        for (int u = 0; u < NU; ++u) mx[u] = fmaxf(mx[u], dpp_f32<0x141>(mx[u]));
#pragma unroll
        for (int u = 0; u < NU; ++u) mx[u] = fmaxf(mx[u], dpp_f32<0x140>(mx[u]));
        float mn[NU], f[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const float wm = fmaxf(fmaxf(readlane_f32(mx[u], 0), readlane_f32(mx[u], 16)), fmaxf(readlane_f32(mx[u], 32), readlane_f32(mx[u], 48)));
          mn[u] = fmaxf(m[B0 + u], wm);  // wave-uniform running max
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) f[u] = __expf(m[B0 + u] - mn[u]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          lsum[B0 + u] *= f[u];
#pragma unroll
          for (int j = 0; j < CVEC; ++j) oacc[B0 + u][j] *= f[u];
        }
        float pr[NU][NK];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < NK; ++i) {
            const int key = base + w * WCH + i * KPW + slot;
            pr[u][i] = key < kvl[B0 + u] ? __expf(sc[u][i] - mn[u]) : 0.f;
          }
#pragma unroll
        for (int i = 0; i < NK; ++i)
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            lsum[B0 + u] += pr[u][i];
            float vf[CVEC];
            widen(vraw[B0 + u][i], vf);
#pragma unroll
            for (int j = 0; j < CVEC; ++j) oacc[B0 + u][j] = fmaf(pr[u][i], vf[j], oacc[B0 + u][j]);
          }
#pragma unroll
        for (int u = 0; u < NU; ++u) m[B0 + u] = mn[u];
      };
      attn_round(std::integral_constant<int, 0>{}, std::integral_constant<int, NB>{}, s * CHUNK);
      // contexts beyond NS * CHUNK = 1024 keys: the further rounds per utterance (block-uniform)
      for_each_utt<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        for (int base = s * CHUNK + NS * CHUNK; base < kvl[b]; base += NS * CHUNK) {
          issue_kv(p, b, base, kvl[b]);
          attn_round(bc, std::integral_constant<int, 1>{}, base);
        }
      });
      // merge the KPW key slots of the wave, then the 4 waves through LDS (persist.hip's order)
      {
        float red9[NB * (CVEC + 1)];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          red9[b * (CVEC + 1)] = lsum[b];
#pragma unroll
          for (int j = 0; j < CVEC; ++j) red9[b * (CVEC + 1) + 1 + j] = oacc[b][j];
        }
#pragma unroll
        for (int i = 0; i < NB * (CVEC + 1); ++i) red9[i] += dpp_f32<0x128>(red9[i]);
        rows4_sums_lockstep<NB * (CVEC + 1)>(red9);
        if (slot == 0) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float* const u = ub(b);
            if (part == 0) {
              u[O_SMM + w] = m[b];
              u[O_SML + w] = red9[b * (CVEC + 1)];
            }
#pragma unroll
            for (int j = 0; j < CVEC; ++j) u[O_SMO + w * DH + part * CVEC + j] = red9[b * (CVEC + 1) + 1 + j];
          }
        }
      }
      g1_lds_barrier();
      auto publish_partial = [&](int b) {  // wave w: the workgroup's partials of utterances w, w + 4 (DH lanes: the output; lane 0 also the (max, sum) pair)
        const float* const u = ub(b);
        const float M = fmaxf(fmaxf(u[O_SMM + 0], u[O_SMM + 1]), fmaxf(u[O_SMM + 2], u[O_SMM + 3]));
        float f[4];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) f[ww] = __expf(u[O_SMM + ww] - M);
        gran_t* gp = G + G_PART + b * L_PART + (h * NS + s) * (2 + DH);
        {
          float o = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) o = fmaf(u[O_SMO + ww * DH + lane], f[ww], o);
          gran_store(gp + 2 + lane, epoch, o);
          gran_store_local(gp + (G_PARTL - G_PART) + 2 + lane, epoch, __float_as_uint(o));
        }
        if (lane == 0) {
          float Ls = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) Ls = fmaf(u[O_SML + ww], f[ww], Ls);
          gran_store(gp + 0, epoch, M);
          gran_store(gp + 1, epoch, Ls);
          gran_store_local(gp + (G_PARTL - G_PART) + 0, epoch, __float_as_uint(M));
          gran_store_local(gp + (G_PARTL - G_PART) + 1, epoch, __float_as_uint(Ls));
        }
      };
      if constexpr (NB <= 4) {
        if (w < NB) publish_partial(w);  // wave b: utterance b
      } else {
        for (int b = w; b < NB; b += 4) publish_partial(b);  // wave w: utterances w, w + 4
      }
    }
    issue_wo(p);

    // ======== (3) merge of the head's NS partials + the new token's own key: wave b for utterance b ===============================
    auto merge_partials = [&](int b) {
      float* const u = ub(b);
      const gran_t* gp = G + G_PART + b * L_PART + (size_t)h * NS * (2 + DH);
      constexpr int NL = NS + NS * QR / 2;
      int j = lane, off = 0;
      if (lane >= NS) {
        const int t = lane - NS;
        j = t / (QR / 2);
        off = 2 + s * QR + 2 * (t % (QR / 2));
      }
      if (lane >= NL) {
        j = 0;
        off = 0;
      }
      float t2[2];
      pt_begin<TR>(pt);
      nap(nap_part);
      {
        const unsigned bo = goff(gp + (size_t)j * (2 + DH) + off);
        gather_two_dual(rs, bo, bo + (unsigned)(G_PARTL - G_PART) * 8u, true, epoch, t2, sp);
      }
      if (lane < NS) {
        u[O_SPM + lane] = t2[0];
        u[O_SPL + lane] = t2[1];
      } else if (lane < NL) {
        const int t = lane - NS;
        u[O_SPO + j * QR + 2 * (t % (QR / 2))] = t2[0];
        u[O_SPO + j * QR + 2 * (t % (QR / 2)) + 1] = t2[1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pt_end<TR>(pt, sp.passes);
      float tq = 0.f;
      {
        const int i = lane & 15;
#pragma unroll
        for (int e = 0; e < QR; ++e) tq = fmaf(u[O_SQ + (i * QR + e) % DH], u[O_SK + (i * QR + e) % DH], tq);
      }
      const float sself = head_group_sum64(tq, DH / QR) * (1.0f / sqrtf((float)DH));
      float M, Ls, acc[QR];
      {
        const int q = lane & 15;
        const float mq = u[O_SPM + q], lq = u[O_SPL + q];
        const f32x4v_t oq = *reinterpret_cast<const f32x4v_t*>(u + O_SPO + q * QR);
        M = fmaxf(row16_max_dpp(mq), sself);
        const float f = __expf(mq - M), fs = __expf(sself - M);
        Ls = row16_sum_dpp(lq * f) + fs;
        acc[0] = fmaf(u[O_SV + s * QR + 0], fs, row16_sum_dpp(oq.x * f));
        acc[1] = fmaf(u[O_SV + s * QR + 1], fs, row16_sum_dpp(oq.y * f));
        acc[2] = fmaf(u[O_SV + s * QR + 2], fs, row16_sum_dpp(oq.z * f));
        acc[3] = fmaf(u[O_SV + s * QR + 3], fs, row16_sum_dpp(oq.w * f));
      }
      const float inv = __builtin_amdgcn_rcpf(Ls);
#pragma unroll
      for (int e = 0; e < QR; ++e) acc[e] *= inv;
      float outv = 0.f;
#pragma unroll
      for (int e = 0; e < QR; ++e) outv = lane == e ? acc[e] : outv;
      if (lane < QR) gran_store(G + G_ATT + b * D + h * DH + s * QR + lane, epoch, outv);
    };
    if constexpr (NB <= 4) {
      if (w < NB) merge_partials(w);  // wave b: utterance b
    } else {
      for (int b = w; b < NB; b += 4) merge_partials(b);  // wave w: utterances w, w + 4
    }

    // the new tokens' K / V elements go into the caches HERE, write-through (persist.hip: behind the merge, in front of an all-to-all edge)
    if (lane < 4 * NB && lr < RQ) {
      const int r = w * RQ + lr, which = r / QR, e = s * QR + (r % QR);
      if (which != 0) {
        const int kvl_l = sel_by_lane<NB>(kvl);
        CT PS_GLOBAL* dst = as_gw<CT>(which == 1 ? p.kc : p.vc) + (((int64_t)lb * H + h) * ctx_max + kvl_l) * DH + e;
        __hip_atomic_store(reinterpret_cast<uint16_t*>((unsigned long long)dst), f32_to_bf16(kv_new), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ======== (4) out-proj + residual of rows 4c .. 4c+3, NB rows ================================================================
    g1_lds_barrier();  // (persist.hip: the other waves do not sweep the attention edge while the merging waves still load)
    {
      float raw[NB][EPT];
      pt_begin<TR>(pt);
      issue_w1_rows(p, 0, R1 - 1);
      nap(nap_att);
      gather_rows16<NB, EPT>(rs, goff(G + G_ATT + tid * EPT), D * 8u, epoch, raw, sp);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        *reinterpret_cast<u32x2v_t*>(reinterpret_cast<unsigned char*>(sxrow(b)) + tid * 8) =
            u32x2v_t{pack_bf16x2(raw[b][0], raw[b][1]), pack_bf16x2(raw[b][2], raw[b][3])};
      g1_lds_barrier();
      pt_end<TR>(pt, sp.passes);
      __builtin_amdgcn_sched_barrier(0);
      float t[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sxrow(b), xb);
        t[b] = ps_dot_bf16<NCH>(wo, xb);
      }
      const float mine = wave_sums_by_column<NB>(t);  // lane b: utterance b
      if (lane < NB) {
        const float v = mine + bo_v;
        gran_store(G + G_X2 + lane * D + 4 * c + w, epoch, ub(lane)[O_SRES + w] + v);
      }
    }
    issue_w1_rows(p, R1 - 1, R1);
    issue_w2_chunks(p, 0, NCH2 / 2);

    // ======== (5) LN2 + linear1 + ReLU of rows 16c .. 16c+15, NB rows =============================================================
    {
      pt_begin<TR>(pt);
      nap(nap_x2);
      gather_rows16<NB, EPT>(rs, goff(G + G_X2 + tid * EPT), D * 8u, epoch, xv, sp);
      pt_end<TR>(pt, sp.passes);
      __builtin_amdgcn_sched_barrier(0);
      if (tid == c) {
#pragma unroll
        for (int b = 0; b < NB; ++b) store_ept_lds<EPT>(ub(b) + O_SRES, xv[b]);
      }
      float ln_mean[NB], ln_rstd[NB];
      fold_stats(xv, g2v, ln_mean, ln_rstd);
      float t[4 * NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sxrow(b), xb);
#pragma unroll
        for (int r = 0; r < R1; ++r) t[4 * b + r] = ps_dot_bf16<NCH>(w1[r], xb);
      }
      const float mine = wave_sums_rows4<NB>(t);  // lane 4 b + r: row r of utterance b
      const float mean_l = sel_by_lane<NB>(ln_mean), rstd_l = sel_by_lane<NB>(ln_rstd);
      const float hval = fmaxf(fmaf(rstd_l, fmaf(-mean_l, sg1_v, mine), b1_v), 0.f);
      const float nbv = dpp_f32<0xB1>(hval);  // lane ^ 1
      if (lane < 4 * NB && (lane & 1) == 0)
        gran_store_bits(G + G_HID + lb * (4 * D) + (4 * R1 * c + w * R1 + lr) / 2, epoch, pack_bf16x2(hval, nbv));
    }
    issue_w2_chunks(p, NCH2 / 2, NCH2);

    // ======== (6) linear2 + residual of rows 4c .. 4c+3, NB rows ==================================================================
    {
      constexpr int NHG = EPT2 / 2;
      float raw[NB][NHG];
      pt_begin<TR>(pt);
      nap(nap_hid);
      gather_rows16<NB, NHG>(rs, goff(G + G_HID + tid * NHG), 4 * D * 8u, epoch, raw, sp);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        unsigned char* dst = reinterpret_cast<unsigned char*>(sxrow(b)) + tid * 32;
        *reinterpret_cast<u32x4_t*>(dst) =
            u32x4_t{__float_as_uint(raw[b][0]), __float_as_uint(raw[b][1]), __float_as_uint(raw[b][2]), __float_as_uint(raw[b][3])};
        *reinterpret_cast<u32x4_t*>(dst + 16) =
            u32x4_t{__float_as_uint(raw[b][4]), __float_as_uint(raw[b][5]), __float_as_uint(raw[b][6]), __float_as_uint(raw[b][7])};
      }
      g1_lds_barrier();
      pt_end<TR>(pt, sp.passes);
      __builtin_amdgcn_sched_barrier(0);
      float t[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        u32x4_t xb[NCH2];
        ps_read_bf16<NCH2>(sxrow(b), xb);
        t[b] = ps_dot_bf16<NCH2>(w2, xb);
      }
      const float mine = wave_sums_by_column<NB>(t);  // lane b: utterance b
      if (lane < NB) {
        const float v = mine + b2_v;
        gran_store(G + GPL + G_X + lane * D + 4 * c + w, epoch, ub(lane)[O_SRES + w] + v);  // the next layer's x edge (layer L: the final norm's)
      }
    }
    issue_wqkv(pn, last);
  }

  // ======== final norm + predict layer: rows 4c .. 4c+3 (+ row 1024), NB rows =======================================================
  {
    gran_t* const G = a.gran + (size_t)a.L * GPL;
    pt_begin<TR>(pt);
    nap(nap_x);
    gather_rows16<NB, EPT>(rs, goff(G + G_X + tid * EPT), D * 8u, epoch, xv, sp);
    pt_end<TR>(pt, sp.passes);
    float ln_mean[NB], ln_rstd[NB];
    fold_stats(xv, g1v, ln_mean, ln_rstd);
    constexpr int G_LOG = G_QKV;  // the final block's q/k/v slots carry the logits edges (V <= 3 D)
    float t0 = 0.f, t1 = 0.f;
    {
      float ta[NB], tx[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sxrow(b), xb);
        ta[b] = ps_dot_bf16<NCH>(wq[0], xb);
        tx[b] = extra_row ? ps_dot_bf16<NCH>(wq[1], xb) : 0.f;
      }
      t0 = wave_sums_by_column<NB>(ta);  // lane b: utterance b
      if (extra_row) t1 = wave_sums_by_column<NB>(tx);
    }
    {
      float mean_l = ln_mean[0], rstd_l = ln_rstd[0];  // of utterance `lane`
#pragma unroll
      for (int b = 1; b < NB; ++b) {
        mean_l = lane == b ? ln_mean[b] : mean_l;
        rstd_l = lane == b ? ln_rstd[b] : rstd_l;
      }
      if (lane < NB) {
        const float lg = fmaf(rstd_l, fmaf(-mean_l, sgq, t0), bq);
        a.logits[(size_t)lane * a.V + 4 * c + w] = lg;
        gran_store(G + G_LOG + lane * (3 * D) + 4 * c + w, epoch, lg);
        if (extra_row) {
          const float lx = fmaf(rstd_l, fmaf(-mean_l, sgx, t1), tbx);
          a.logits[(size_t)lane * a.V + 4 * NWG] = lx;
          gran_store(G + G_LOG + lane * (3 * D) + 4 * NWG, epoch, lx);
        }
      }
    }
    {
      // ======== sampling, stop rule, next input rows (ar_sample_kernel, sampling.hip; valle/models/valle.py:1039-1057) ==============
      // Every workgroup gathers the logits of every utterance and draws the SAME tokens (persist.hip).
      float* const slog = smem + SAMP0;   // [V] one utterance's row at a time
      unsigned long long* const red64 = reinterpret_cast<unsigned long long*>(smem + SAMP0 + 2 * 1024);
      int* const redi = reinterpret_cast<int*>(smem + SAMP0 + 2 * 1024 + 16);
      float* const redf = smem + SAMP0 + 2 * 1024 + 32;
      float* const wave_tot = smem + SAMP0 + 2 * 1024 + 48;
      static_assert(SAMP_T == PS_T && SAMP_T * SAMP_PER >= 4 * NWG + 1, "the sampling code's block shape");
      const PStepSample q = ps_sample_load(a.smp);
      const ArDyn dyn = *q.dyn;
      const PsLayer p0 = ps_layer(a.layers, 0);
      issue_wqkv(p0, false);
      float pev[NB][EPT];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int apos1 = apos[b] + 1;
        const int pe_row = apos1 < q.pe_rows ? apos1 : q.pe_rows - 1;  // (persist.hip: the request stays inside the table)
        ps_load4(as_g<float>((unsigned long long)q.pe) + (int64_t)pe_row * D + tid * EPT, pev[b]);
      }
      const float alpha = *q.alpha_audio;
      float lg4[NB][EPT], lgx[NB];
      pt_begin<TR>(pt);
      nap(nap_x);
      gather_rows16_plus1<NB, EPT>(rs, goff(G + G_LOG + tid * EPT), goff(G + G_LOG + 4 * NWG), 3 * D * 8u, epoch, lg4, lgx, sp);
      pt_end<TR>(pt, sp.passes);
      const int V = a.V;
      int next[NB];
      unsigned stopped = 0u;  // utterances that stop in THIS step
      for_each_utt<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        next[b] = 0;
        if (!((live >> b) & 1u)) return;  // block-uniform
        store_ept_lds<EPT>(slog + tid * EPT, lg4[b]);
        if (tid == 0) slog[4 * NWG] = lgx[b];
        __syncthreads();
        float raw[SAMP_PER];
#pragma unroll
        for (int j = 0; j < SAMP_PER; ++j) {
          const int idx = tid * SAMP_PER + j;
          raw[j] = idx < V ? slog[idx] : -INFINITY;
        }
        if (c == 0 && dyn.trace != nullptr && itb[b] < dyn.trace_cap) {
          float* tr = dyn.trace + ((int64_t)itb[b] * a.B + b) * V;
#pragma unroll
          for (int j = 0; j < SAMP_PER; ++j) {
            const int idx = tid * SAMP_PER + j;
            if (idx < V) tr[idx] = raw[j];
          }
        }
        const int argmax = argmax_row(raw, V, red64);
        const unsigned long long rseed = a.slot_seed != nullptr ? a.slot_seed[b] : request_seed(dyn.seed, (unsigned long long)b);
        const int sample = sample_row(raw, V, dyn.top_k, dyn.temperature, rseed, (uint32_t)itb[b], argmax, SampScratch{red64, redi, redf, wave_tot});
        __syncthreads();  // the row and the scratch words are free for the next utterance
        // stop rule (valle.py:1044-1048) and bookkeeping: the same integers in every workgroup; workgroup 0 stores them
        const int kvl1 = kvl[b] + 1;
        int stop = (!dyn.ignore_eos && ((argmax == 1024) || (sample == 1024))) || (n_gen[b] + q.bos > s_cap[b]);
        if (dyn.max_new > 0 && n_gen[b] >= dyn.max_new) stop = 1;
        if (dyn.has_forced) stop = n_gen[b] >= dyn.forced_len[b];
        if (n_gen[b] >= (int)q.g_stride || kvl1 >= ctx_max) stop = 1;  // capacity guard
        int nx = sample;
        bool bad_id = false;
        if (!stop && dyn.has_forced) {
          const int64_t f = dyn.forced[(int64_t)b * dyn.forced_stride + n_gen[b]];
          nx = (int)f;
          if (f < 0 || f >= (int64_t)V + q.bos) {  // outside ar_audio_embedding: the reference's nn.Embedding raises IndexError
            nx = 0;
            bad_id = true;
          }
        }
        if (c == 0 && tid == 0) {
          if (!stop) {
            if (bad_id && q.id_err) atomicOr(q.id_err, 4);
            q.tokens[(int64_t)b * q.g_stride + n_gen[b]] = nx;
            q.sampled[(int64_t)b * q.g_stride + n_gen[b]] = sample;
            q.s.n_gen[b] = n_gen[b] + 1;
            q.s.kv_len[b] = kvl1;
            q.s.audio_pos[b] = apos[b] + 1;
          } else {
            if (n_gen[b] < (int)q.g_stride) q.sampled[(int64_t)b * q.g_stride + n_gen[b]] = sample;  // the stopping iteration's own draw
            q.s.done[b] = 1;
          }
          q.s.iter[b] = itb[b] + 1;
        }
        itb[b] += 1;
        if (stop) {
          stopped |= 1u << b;
        } else {
          next[b] = nx;
          n_gen[b] += 1;
          apos[b] += 1;
          kvl[b] = kvl1;
        }
      });
      const unsigned live_after = live & ~stopped;
      if (c == 0 && tid == 0) {
        const int nd = __builtin_popcount(stopped);
        if (q.host_prog != nullptr) {  // [0] utterances done before this step's successor, [1] sampling steps so far (ar_sample_kernel's words)
          const int sc = q.s.done_count[1] + (live_after == 0u ? nsteps - step : 1);
          q.s.done_count[1] = sc;
          __hip_atomic_store(q.host_prog + 0, dcount + nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(q.host_prog + 1, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (nd) q.s.done_count[0] = dcount + nd;
        if (a.epoch_ctr != nullptr) a.epoch_ctr[0] = it + 1;  // (read again only by the next launch)
      }
      dcount += __builtin_popcount(stopped);
      live = live_after;
      if (live == 0u) return;
      // next step's inputs: ar_audio_position(ar_audio_embedding(token))  (valle.py:1013-1015), the sampling kernel's roundings
      for_each_utt<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if (!((live >> b) & 1u)) return;  // (a stopped utterance keeps its last row: nothing of it is stored any more)
        float ev[EPT];
        load_ept<EPT>(q.audio_emb + (int64_t)next[b] * D + tid * EPT, ev);
#pragma unroll
        for (int k = 0; k < EPT; ++k) xv[b][k] = __fadd_rn(ev[k], __fmul_rn(alpha, pev[b][k]));
        if (c == 0) {
          float* xo = q.x + (size_t)b * D + tid * EPT;
#pragma unroll
          for (int k = 0; k < EPT; ++k) xo[k] = xv[b][k];
        }
      });
      it += 1;
      epoch = (unsigned)(it + 1);
    }
    pt_begin<TR>(pt);
    pt_end<TR>(pt, 0u);
  }
  }  // step
}

bool pstepb_supports(int dtype, int d, int nhead, int dh, int V, int B) {
  return dtype == DT_BF16 && d == 1024 && nhead == 16 && dh == 64 && V > 1024 && V <= 1025 && B >= 2 && B <= PSB_MAX;
}

size_t pstepb_gran_count(int d, int nhead, int L, int B) { return (size_t)B * (L + 1) * ps_gran_per_layer(d, nhead, 256 / nhead); }

typedef void (*PsbKernel)(PStepArgs);
static PsbKernel psb_select(int B, bool traced) {
  if (traced) return B == 2 ? (PsbKernel)pstepb_kernel<2, true> : B == 3 ? (PsbKernel)pstepb_kernel<3, true> : B == 4 ? (PsbKernel)pstepb_kernel<4, true> : nullptr;
  return B == 2 ? (PsbKernel)pstepb_kernel<2> : B == 3 ? (PsbKernel)pstepb_kernel<3> : B == 4 ? (PsbKernel)pstepb_kernel<4> :
         B == 5 ? (PsbKernel)pstepb_kernel<5> : B == 6 ? (PsbKernel)pstepb_kernel<6> : nullptr;
}

// 1 = the occupancy calculator places one workgroup of the B-utterance form on a CU; -1 = it does not fit; 0 = no such form
int pstepb_form_ok(int B, bool traced) {
  const PsbKernel k = psb_select(B, traced);
  if (k == nullptr) return 0;
  int per_cu = 0;
  const hipError_t r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, PS_T, 0);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return per_cu >= 1 ? 1 : -1;
}

// returns 0 = launched, 1 = shape not covered, < 0 = error
int launch_pstepb(hipStream_t st, int dtype, const PStepArgs& a) {
  if (!pstepb_supports(dtype, a.d, a.nhead, a.dh, a.V, a.B)) return 1;
  if (!a.layers || !a.x_in || !a.logits || !a.kv_len || !a.iter || !a.done || !a.gran || a.L < 1) return -1;
  if (a.nsteps < 1 || a.nsteps > 4096 || !a.smp) return -1;  // (the sampling step is always inside this launch)
  const PsbKernel k = psb_select(a.B, a.ptrace != nullptr);
  if (k == nullptr) return -1;
  hipLaunchKernelGGL(k, dim3(256), dim3(PS_T), 0, st, a);
  return 0;
}

}  // namespace vle
