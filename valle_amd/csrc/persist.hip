// The batch-1 AR decode loop as persistent launches of several steps each (options "persist", "persist_sample", "persist_steps";
// kernels.h PStepArgs / PStepSample).
//   reference: the AR loop of VALLE.inference (valle/models/valle.py:1012-1057), per iteration: the L pre-norm decoder layers
//   (valle/modules/transformer.py:296-302, 332-334; attention valle/modules/activation.py:408-427 on the last row of the prefix-LM
//   mask valle.py:1019-1033), the final norm and ar_predict_layer (valle.py:1035-1039), topk_sampling (:1287-1302), the stop rule
//   (:1044-1048) and the next token's embedding + position (:1013-1015, :1057).
//
// Why.  As a launch chain the step is 4 dependent launches per layer (gemv1.hip) + logits + sampling: 50 boundaries of ~1.66 us plus
// 50 bodies that each begin with one un-hidden HBM round trip -- 211 us per step for 336 MB (0.198 of the HBM
// roofline, traffic ratio 1.10: latency, not bytes).  Here a launch is one grid of 256 workgroups, ONE per CU, that never leaves
// the chip for up to `nsteps` AR iterations:
//   * every workgroup owns the same slice of every operator (rows of W: 12 of the in-projection = 4 query + 4 key + 4 value rows
//     of ITS head, 4 of out-proj, 16 of linear1, 4 of linear2, 4 of the predict layer; and one (head, 1/16 of the keys) share of
//     the attention), so its weights never move: they are requested one operator AHEAD, straight into registers (a layer's
//     share is 96 KB per workgroup = 96 VGPRs per lane of the 512 a one-wave-per-SIMD workgroup owns), and have landed when the
//     operator's input arrives -- the HBM round trip that opens every kernel of the chain is gone from the critical path;
//   * an operator's output vector travels producer -> consumers as 8-byte {epoch, value} granules (the guide's recipe R2: the data
//     is the flag; one relaxed agent-scope store per value, consumers re-read their granules until every tag carries this step's
//     epoch = the AR iteration counter + 1; nothing else orders anything).  Six edges per layer:
//        x -> [LN1, in-proj] -> (q, k, v of head h: 16 workgroups of ONE XCD, 192 granules) -> [attention over the cached keys]
//        -> (16 split partials of head h, same 16 workgroups, 96 granules each) -> [merge + the new token's own key]
//        -> (attention output, 1024 granules, all) -> [out-proj + residual] -> (x', 1024, all) -> [LN2, linear1, ReLU]
//        -> (hidden, 4096, all) -> [linear2 + residual] -> (x'', 1024, all) -> next layer;
//     and per step: (x of the final norm, 1024, all) -> [final norm, predict layer] -> (logits, 1025, all) -> [draw, stop rule,
//     embedding: the same in every workgroup] -> layer 0 of the next step.  Buffers are per (layer, edge) and reused from step to
//     step: a granule of step t + 1 is stored only behind an all-to-all edge of step t + 1, i.e. after everybody read step t;
//   * every arithmetic step is the launch chain's own device function on the same lane <-> element mapping (gemv1_dev.h), the
//     attention share is qkv_attn1_kernel's code with 16 splits and the draw is ar_sample_kernel's code (sampling_dev.h): with the
//     three-barrier LayerNorm (mode bit 5 off) the step is BIT-IDENTICAL to the chain run with qa_nsplit = 16, qa_nk = NK
//     (tests/test_persist_gpu.py asserts equality of every logit and token of whole decodes); the folded LayerNorm (bit 5, default)
//     is the same arithmetic up to fp32 re-association.
// Round 5: between hand-offs the step is ISSUE-bound (one wave per SIMD: every instruction on a stage's critical path is time), so
// the default form trades the chain's bit-identity for fewer instructions (template bit D2: bf16 activation rows in LDS + v_dot2c,
// lane-parallel merge, DPP / permlane wave totals, v_rsq / v_rcp: layer loop 2873 -> 1998 instructions, 144.7 -> 127.8 us per step;
// the fp32-row forms above are kept, instantiated and tested); the in-kernel timeline is a template parameter; and the launch is
// instantiated for fp8 weight rows (FP8W: e4m3fn codes + row scales, bit-identical to the fp8w chain in the three-barrier form).
// Spins are bounded: a wave that gives up marks the launch failed (PStepArgs::fail, reported by the engine as an error), stops
// waiting for the rest of the launch and lets every other workgroup run through, so a lost granule cannot hang the device.
// Co-residency: 256 workgroups of 256 threads with > 80 KB of LDS each = one per CU on an otherwise idle MI355X (the engine
// owns its stream; a launch's neighbours in the graph are ordinary dependent launches).
#include "persist_dev.h"

namespace vle {

// PF: 0 = an operator's operands are requested right BEFORE the sweep that precedes it (they land while the edge is in flight, but
//         the sweep cannot return before they have: a wave's loads return in order);
//     1 = the sweep first, THEN the operands of the operator AFTER the next one (gather_vals16's functor): measured 240-280 us per
//         step -- the requests of all 256 workgroups fill the memory system right when everybody's sweep is in flight;
//     2 = at the START of the previous stage (right after its own sweep): they land under that stage's arithmetic, publish and
//         hand-off latency.
// PK: bit 0 = the FFN hidden vector, bit 1 = the attention output travel as bf16 pairs (compile-time: a run-time branch around
//     a sweep that carries requests would make hipcc merge in-flight registers, see below).
//     bit 2 = folded LayerNorm: LN(x) . W[n] = rstd * (sum_k W[n][k] gamma[k] x[k] - mean * sg[n]) + tb[n] with the row constants
//     sg, tb of launch_ps_fold.  The dot products need only x * gamma, which a thread has as soon as its gather returns; the row
//     statistics travel as one (mean, M2) pair per wave through the SAME barrier that publishes x * gamma to the other waves and are
//     combined afterwards (Chan's update: exact two-pass statistics per wave, no cancellation in the combination) -- one workgroup
//     barrier per LayerNorm instead of three.  Same arithmetic as the reference's LayerNorm + Linear up to fp32 re-association
//     (valle/modules/transformer.py:57-74 then F.linear), so NOT bit-identical to the launch chain: tests/test_persist_gpu.py holds
//     it to 1e-4 of the logits' spread against the three-barrier form and to bit-reproducibility against itself.
template <typename T, int D, int H, int NK, int PF, int PK, bool TR = false>
__global__ __launch_bounds__(PS_T) void pstep_kernel(PStepArgs a) {

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
  static_assert(D % CH == 0 && D % PS_T == 0 && DH * H == D && NS * H == NWG && QR * NS == DH, "shape");
  static_assert((3 * QR) % 4 == 0 && D / NWG == 4 && EPT == 4, "rows per wave");
  static_assert(H % 8 == 0, "whole heads per XCD");
  typedef typename G1W<T>::cache_t CT;
  constexpr int CVEC = Elem<CT>::VEC;
  constexpr int LPK = DH / CVEC, KPW = 64 / LPK, WCH = NK * KPW, CHUNK = 4 * WCH;
  static_assert(DH % CVEC == 0 && (LPK & (LPK - 1)) == 0 && LPK <= 32, "head size");
  // FP8W (T = bf16w8t_t): the weight rows are e4m3fn codes (8 bytes per lane and chunk, default cache policy: one utterance's fp8 AR
  // weights fit the memory-side cache) with one power-of-two scale per row, applied as the launch chain does -- fmaf(dot, scale, bias)

  // ---- LDS: one array (> 80 KB: one workgroup per CU) -------------------------------------------------------------------------
  constexpr int SM_FLOATS = 21 * 1024;
  __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
  float* const sx = smem;                       // [4 D] the operator's input vector
  float* const red = smem + 4 * D;              // [8]
  float* const sq = red + 8;                    // [DH] q of the head
  float* const sk = sq + DH;                    // [DH] the new token's key (cache-rounded)
  float* const sv = sk + DH;                    // [DH] ... value
  float* const sm_m = sv + DH;                  // [4]
  float* const sm_l = sm_m + 4;                 // [4]
  float* const sm_o = sm_l + 4;                 // [4][DH]
  float* const spm = sm_o + 4 * DH;             // [NS]
  float* const spl = spm + NS;                  // [NS]
  float* const spo = spl + NS;                  // [NS][QR]
  float* const sres = spo + NS * QR;            // [4] residual values of the rows this workgroup owns
  static_assert(4 * D + 8 + 3 * DH + 8 + 4 * DH + 2 * NS + NS * QR + 4 <= SM_FLOATS, "LDS carve");

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = (int)blockIdx.x;
  // the NS workgroups of a head on ONE XCD (block b runs on XCD b % 8: speed only)
  const int jj = c >> 3;
  const int h = (c & 7) * (H / 8) + jj / NS, s = jj % NS;
  if (a.done[0]) {  // the utterance has stopped: the remaining launches of the host's queue are no-ops
    if (a.nsteps > 0 && c == 0 && tid == 0) {  // ... that still report (ar_sample_kernel's progress words)
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
  // An earlier launch of this call gave up (the GPU is shared): its results are void and the call will end with VLE_EBUSY -- the
  // launches still queued behind it end at once instead of each waiting out its own budget.  The counter is zeroed on the stream
  // at the start of every vle_ar_generate and only ever grows inside one, so every workgroup of a launch reads the same answer
  // unless the give-up happens inside THIS launch (then the others' bounded spins end it).
  if (a.fail != nullptr && __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  if (c == 0 && tid == 0 && a.never) smem[SM_FLOATS - 1] = 0.f;  // keeps the whole array allocated

  int it = a.iter[0];               // AR iteration of the step being computed (advances inside a multi-step launch)
  unsigned epoch = (unsigned)(it + 1);
  int kvl = a.kv_len[0];            // slot of the new token; the old keys are [0, kvl)
  const int ctx_max = a.ctx_max;
  const int mode = a.mode;
  constexpr bool hpack = (PK & 1) != 0, apack = (PK & 2) != 0, LF = (PK & 4) != 0, D2 = (PK & 8) != 0, W8 = G1W<T>::kScaled;
  static_assert(!(W8 && D2), "the v_dot2c forms multiply bf16 weights");
  static_assert(!D2 || LF, "the bf16 activation rows carry x * gamma: folded LayerNorm only");
  typedef unsigned u32x2v_t __attribute__((ext_vector_type(2)));
  const bool glocal = D2 ? true : (mode & 16) != 0;  // (D2 is instantiated with the XCD-local copies only: launch_pstep)
  // s_sleep(8) units ahead of the FIRST sweep of an all-to-all edge (attention output, x, x', hidden): a sweep that comes back
  // without the data costs a whole fabric round trip (~1.1 us) before the next one can see it -- waiting first is cheaper
  const int naps = a.naps;  // "persist_naps": s_sleep(4) units (~0.1 us), 4 bits per edge
  const int nap_att = naps & 15, nap_x = (naps >> 4) & 15, nap_x2 = (naps >> 8) & 15, nap_hid = (naps >> 12) & 15, nap_qkv = (naps >> 16) & 15,
            nap_part = (naps >> 20) & 15;
  PsSpin sp{PS_SPINS, a.fail, (mode >> 8) & 15, 0u};
  PsTrace pt{nullptr, 0, 0ull};
  if constexpr (TR) pt.p = (a.ptrace && tid == 0) ? a.ptrace + ((size_t)(it & 7) * NWG + c) * PS_PT_SLOTS : nullptr;
  pt_begin<TR>(pt);
  pt_end<TR>(pt, 0u);

  const int GPL = ps_gran_per_layer(D, H, NS);
  const gran_t* const GB = a.gran;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gran, 0, (int)((size_t)(a.L + 1) * GPL * sizeof(gran_t)), 0x00020000);
  // offsets inside a layer's granules
  constexpr int G_X = 0, G_QKV = D, G_PART = 4 * D, G_ATT = G_PART + H * NS * (2 + DH), G_X2 = G_ATT + D, G_HID = G_X2 + D;
  constexpr int G_QKVL = G_HID + 4 * D, G_PARTL = G_QKVL + 3 * D;  // XCD-local copies of the two head-group edges

  // ---- register-resident operands, requested ahead -----------------------------------------------------------------------------
  u32x4_t wq[RQ][NCH], wo[NCH], w1[R1][NCH], w2[NCH2];
  float bq = 0.f, bo_v = 0.f, b1_v = 0.f, b2_v = 0.f;
  float scq = 1.f, sco = 1.f, sc1 = 1.f, sc2 = 1.f, scx = 1.f;  // FP8W row scales, fetched beside the biases (scx: row 1024 of the predict layer)
  float sgq = 0.f, sg1_v = 0.f, sgx = 0.f, tbx = 0.f;  // folded LayerNorm: sg of the rows whose tb sits in bq / b1_v; row 1024's pair
  float g1v[EPT], be1v[EPT], g2v[EPT], be2v[EPT];
  u32x4_t kraw[NK], vraw[NK];

  const int slot = lane / LPK, part = lane % LPK;
  // in-projection row r (0 .. 3 QR - 1) of this workgroup: which = r / QR (0 Q, 1 K, 2 V), e = r % QR -> row which * D + h * DH + s * QR + e
  auto qkv_row = [&](int r) { return (r / QR) * D + h * DH + s * QR + (r % QR); };

  // a row's 16-byte vectors of this lane: vector cc of row `row` of W[.][KK]
  auto wvec = [&](unsigned long long W, int64_t row, int KK, int cc) {
    if constexpr (W8) {
      const u32x2v_t t = *reinterpret_cast<const u32x2v_t PS_GLOBAL*>(as_g<T>(W) + row * KK + lane * VEC + cc * CH);
      return u32x4_t{t.x, t.y, 0u, 0u};
    } else {
      return ps_load_nt(reinterpret_cast<const u32x4_t PS_GLOBAL*>(as_g<T>(W) + row * KK + lane * VEC + cc * CH));
    }
  };
  // Every request below is STRAIGHT-LINE code on selected addresses: a load under a branch or an exec mask makes hipcc merge the
  // loaded registers with the other path's by v_mov after an s_waitcnt -- the "prefetch" then waits for its own HBM round trip
  // (measured: 1.0 us at the end of every linear2 stage).
  const bool extra_row = (c == 0 && w == 0 && 4 * NWG < a.V);  // wave-uniform: row 4 * 256 (the EOS row at V = 1025)
  // the in-projection rows of layer `p`, or (pred) the predict layer's: its row 4c + w in wq[0], row 1024 in the one wave that owns
  // it (the others re-request their own row: an L2 hit), and the final norm's affine
  // (the predict layer is entry L of the operand table: wqkv = its weight, g1 / be1 = the final norm's affine, sgqkv / tbqkv = its
  //  folded row constants -- no kernel argument stays live through the layers for it)
  auto issue_wqkv = [&](const PsLayer& p, bool pred) {
    const unsigned long long W = p.wqkv;
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int64_t prow = (r == 1 && extra_row) ? 4 * NWG : 4 * c + w;
      const int64_t row = pred ? prow : (int64_t)qkv_row(w * RQ + r);
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) wq[r][cc] = wvec(W, row, D, cc);
    }
    if constexpr (!LF) {
      bq = as_g<float>(p.bqkv)[qkv_row(w * RQ + (lane < RQ ? lane : RQ - 1))];  // (unused by the predict layer: it has no bias)
    } else {  // the row's (sg, tb): in-projection row of lane r, or the predict layer's row 4c + w (and row 1024 for its one wave)
      const int rq = qkv_row(w * RQ + (lane < RQ ? lane : RQ - 1));
      const int ri = pred ? 4 * c + w : rq;
      sgq = as_g<float>(p.sgqkv)[ri];
      bq = as_g<float>(p.tbqkv)[ri];
      sgx = as_g<float>(p.sgqkv)[pred ? 4 * NWG : 0];  // row 1024's pair (used by the one wave that owns the row)
      tbx = as_g<float>(p.tbqkv)[pred ? 4 * NWG : 0];
    }
    if constexpr (W8) {
      const int rq = qkv_row(w * RQ + (lane < RQ ? lane : RQ - 1));
      scq = as_g<float>(p.sqkv)[pred ? 4 * c + w : rq];
      scx = as_g<float>(p.sqkv)[pred ? 4 * NWG : 0];
    }
    ps_load4(as_g<float>(p.g1) + tid * EPT, g1v);
    if constexpr (!LF) ps_load4(as_g<float>(p.be1) + tid * EPT, be1v);
  };
  // Cache rows inside a multi-step launch: the row a step appends (stage 1, another workgroup, possibly another XCD whose L2 is not
  // coherent with this one's) must be readable one step later without a kernel boundary in between.  The writer stores it
  // write-through (agent scope), and NO load of this launch touches a row before it is written -- lanes beyond the valid length
  // `nvalid` re-read row nvalid - 1 instead of running ahead into unwritten rows -- so the first touch of the row's 128-byte line by
  // any L1 / L2 comes after the data is in memory, and ordinary cached loads see it.  (L2-bypassing loads would do too: measured
  // +0.3 us per layer, the cache is partly L2-resident from step to step.)
  auto issue_kv = [&](const PsLayer& p, int base, int nvalid) {
    const CT PS_GLOBAL* Kb = as_g<CT>(p.kc) + (int64_t)h * ctx_max * DH + part * CVEC;
    const CT PS_GLOBAL* Vb = as_g<CT>(p.vc) + (int64_t)h * ctx_max * DH + part * CVEC;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      int key = base + w * WCH + i * KPW + slot;
      key = key < nvalid ? key : nvalid - 1;
      key = key > 0 ? key : 0;  // nvalid = 0 (nothing cached yet) must not reach in front of the cache
      kraw[i] = *reinterpret_cast<const u32x4_t PS_GLOBAL*>(Kb + (int64_t)key * DH);
      vraw[i] = *reinterpret_cast<const u32x4_t PS_GLOBAL*>(Vb + (int64_t)key * DH);
    }
  };
  auto issue_wo = [&](const PsLayer& p) {
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) wo[cc] = wvec(p.wo, 4 * c + w, D, cc);
    bo_v = as_g<float>(p.bo)[4 * c + w];
    if constexpr (W8) sco = as_g<float>(p.so)[4 * c + w];
  };
  // linear1's rows [r0, r1) of this wave (+ its bias and norm2's affine with the first part); linear2's chunks [c0, c1)
  auto issue_w1_rows = [&](const PsLayer& p, int r0, int r1) {
#pragma unroll
    for (int r = 0; r < R1; ++r) {
      if (r < r0 || r >= r1) continue;
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) w1[r][cc] = wvec(p.w1, 4 * R1 * c + w * R1 + r, D, cc);
    }
    if (r0 == 0) {
      const int r1i = 4 * R1 * c + w * R1 + (lane < R1 ? lane : R1 - 1);
      if constexpr (!LF) {
        b1_v = as_g<float>(p.b1)[r1i];
      } else {
        sg1_v = as_g<float>(p.sg1)[r1i];
        b1_v = as_g<float>(p.tb1)[r1i];
      }
      if constexpr (W8) sc1 = as_g<float>(p.s1)[r1i];
      ps_load4(as_g<float>(p.g2) + tid * EPT, g2v);
      if constexpr (!LF) ps_load4(as_g<float>(p.be2) + tid * EPT, be2v);
    }
  };
  auto issue_w1 = [&](const PsLayer& p) { issue_w1_rows(p, 0, R1); };
  auto issue_w2_chunks = [&](const PsLayer& p, int c0, int c1) {
#pragma unroll
    for (int cc = 0; cc < NCH2; ++cc) {
      if (cc < c0 || cc >= c1) continue;
      w2[cc] = wvec(p.w2, 4 * c + w, 4 * D, cc);
    }
    if (c0 == 0) {
      b2_v = as_g<float>(p.b2)[4 * c + w];
      if constexpr (W8) sc2 = as_g<float>(p.s2)[4 * c + w];
    }
  };
  auto issue_w2 = [&](const PsLayer& p) { issue_w2_chunks(p, 0, NCH2); };
  // ---- x of the first layer: the sampling kernel's output (previous launch) ----------------------------------------------------
  float xv[EPT];
  load_ept<EPT>(a.x_in + tid * EPT, xv);
  {
    const PsLayer p0 = ps_layer(a.layers, 0);
    issue_wqkv(p0, false);
    if constexpr (PF >= 1) issue_kv(p0, s * CHUNK, kvl);
  }
  __builtin_amdgcn_sched_barrier(0);
  auto nap = [&](int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
  };
  // folded LayerNorm, the row side: x * gamma into sx, this wave's (mean, M2) of its 256 elements into red, ONE barrier, then the
  // row's mean and 1 / sqrt(var + eps) from the four pairs.  (valle/modules/transformer.py:57-74: biased variance, eps 1e-5)
  auto fold_stats = [&](const float (&xr)[EPT], const float (&gv)[EPT], float& mean, float& rstd) {
    float xg[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) xg[k] = xr[k] * gv[k];
    if constexpr (D2) *reinterpret_cast<u32x2v_t*>(reinterpret_cast<unsigned char*>(sx) + tid * 8) = u32x2v_t{pack_bf16x2(xg[0], xg[1]), pack_bf16x2(xg[2], xg[3])};
    else store_ept_lds<EPT>(sx + tid * EPT, xg);
    float sw = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) sw += xr[k];
    const float mw = (D2 ? ps_wave_sum_fast(sw) : wave_sum_dpp(sw)) * (1.0f / (64.0f * EPT));
    float qw = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const float t = xr[k] - mw;
      qw = fmaf(t, t, qw);
    }
    qw = D2 ? ps_wave_sum_fast(qw) : wave_sum_dpp(qw);
    if (lane == 0) {
      red[w] = mw;
      red[4 + w] = qw;
    }
    g1_lds_barrier();
    mean = ((red[0] + red[1]) + (red[2] + red[3])) * 0.25f;
    float m2 = (red[4] + red[5]) + (red[6] + red[7]);
    float dm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = red[i] - mean;
      dm = fmaf(t, t, dm);
    }
    m2 = fmaf(dm, 64.0f * EPT, m2);
    // (D2: v_rsq_f32, 1 ulp -- the IEEE square root + division sequences are 25 instructions on every LayerNorm's critical path)
    if constexpr (D2) rstd = __builtin_amdgcn_rsqf(fmaf(m2, 1.0f / (float)D, LN_EPS));
    else rstd = 1.0f / sqrtf(m2 * (1.0f / (float)D) + LN_EPS);
  };

  // ---- sampling inside the launch: the utterance's state, carried in registers by every workgroup ------------------------------
  const bool own_sample = a.nsteps > 0;
  int n_gen = 0, apos = 0, s_cap = 0;
  if (own_sample) {
    const PStepSample q = ps_sample_load(a.smp);
    n_gen = q.s.n_gen[0];
    apos = q.s.audio_pos[0];
    s_cap = q.s.cap[0];
  }
  const int nsteps = own_sample ? a.nsteps : 1;

  for (int step = 0; step < nsteps; ++step) {
  if (step > 0) {
    sp.budget = sp.budget ? PS_SPINS : 0u;  // a wave that gave up stays out; the others get a fresh budget per step
    if constexpr (TR) pt = PsTrace{(a.ptrace && tid == 0) ? a.ptrace + ((size_t)(it & 7) * NWG + c) * PS_PT_SLOTS : nullptr, 0, 0ull};
    pt_begin<TR>(pt);
    pt_end<TR>(pt, 0u);
  }
  for (int l = 0; l < a.L; ++l) {
    const PsLayer p = ps_layer(a.layers, l);
    const PsLayer pn = ps_layer(a.layers, l + 1);  // the next layer's entry (entry L: the predict layer), long before its operands are requested
    gran_t* const G = a.gran + (size_t)l * GPL;
    const bool last = l + 1 == a.L;

    // ======== (1) LN1 + in-projection of this head's 3 QR rows =================================================================
    if (l > 0) {
      pt_begin<TR>(pt);
      nap(nap_x);
      if constexpr (PF == 1) ps_gather<EPT>(GB, rs, G + G_X + tid * EPT, epoch, xv, sp, [&]() { issue_kv(p, s * CHUNK, kvl); });
      else ps_gather<EPT>(GB, rs, G + G_X + tid * EPT, epoch, xv, sp, PsNoop());
      pt_end<TR>(pt, sp.passes);
      if constexpr (PF == 2) issue_kv(p, s * CHUNK, kvl);
    }
    if (tid == c) store_ept_lds<EPT>(sres, xv);  // thread c holds x[4c .. 4c+3]: the residual of the rows this workgroup owns
    float kv_new = 0.f;  // lanes < RQ: this lane's K or V element of the new token (stored into the cache further down)
    float ln_mean = 0.f, ln_rstd = 1.f;
    if constexpr (LF) fold_stats(xv, g1v, ln_mean, ln_rstd);
    else g1_block_layernorm<D, PS_T>(xv, g1v, be1v, sx, red);
    {
      float mine = 0.f;
      if constexpr (D2) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sx, xb);
        float t[RQ];
#pragma unroll
        for (int r = 0; r < RQ; ++r) t[r] = ps_dot_bf16<NCH>(wq[r], xb);
        mine = ps_wave_sums_fast<RQ>(t);  // lane r < RQ: row r
      } else {
        float x[NCH][VEC];
        g1_read_shared<T, NCH>(sx, x);
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
          const float t = wave_sum_dpp(g1_dot<T, NCH>(wq[r], x));
          mine = lane == r ? t : mine;
        }
      }
      if (lane < RQ) {
        const int r = w * RQ + lane, which = r / QR, e = s * QR + (r % QR);  // e: element of the head
        const float dq = W8 ? mine * scq : mine;  // (a power-of-two scale: exact)
        const float v = LF ? fmaf(ln_rstd, fmaf(-ln_mean, sgq, dq), bq) : (W8 ? fmaf(mine, scq, bq) : mine + bq);
        gran_t* gq = G + G_QKV + h * (3 * DH) + which * DH + e;
        gran_t* gql = G + G_QKVL + h * (3 * DH) + which * DH + e;
        if (which == 0) {
          gran_store(gq, epoch, v);
          if (glocal) gran_store_local(gql, epoch, __float_as_uint(v));
        } else {
          kv_new = v;
          float vr = v;
          if constexpr (sizeof(CT) == 2) vr = bf16_to_f32(f32_to_bf16(v));  // what later steps will read back from the cache
          gran_store(gq, epoch, vr);
          if (glocal) gran_store_local(gql, epoch, __float_as_uint(vr));
        }
      }
    }
    if constexpr (PF == 0 || PF == 3) issue_kv(p, s * CHUNK, kvl);

    // ======== (2) q, k_new, v_new of the head; attention over this workgroup's share of the cached keys ==========================
    pt_begin<TR>(pt);
    {
      // every wave sweeps (wave 3 repeats wave 0's granules and drops them): the requests the sweep carries stay straight-line code
      nap(nap_qkv);
      const int wq_i = w < 3 ? w : 0;
      const int gi = h * (3 * DH) + wq_i * DH + (lane < DH ? lane : 0);
      float t;
      if constexpr (PF == 1) t = gather_one_dual(G + G_QKV + gi, glocal ? G + G_QKVL + gi : nullptr, epoch, sp, [&]() { issue_wo(p); });
      else t = gather_one_dual(G + G_QKV + gi, glocal ? G + G_QKVL + gi : nullptr, epoch, sp, PsNoop());
      if (w < 3 && lane < DH) (w == 0 ? sq : w == 1 ? sk : sv)[lane] = t;
    }
    g1_lds_barrier();
    pt_end<TR>(pt, sp.passes);
    if constexpr (PF == 2) issue_wo(p);
    __builtin_amdgcn_sched_barrier(0);
    {
      auto widen = [&](const u32x4_t& r, float (&f)[CVEC]) {
        if constexpr (sizeof(CT) == 4) {
          f[0] = __uint_as_float(r.x); f[1 % CVEC] = __uint_as_float(r.y); f[2 % CVEC] = __uint_as_float(r.z); f[3 % CVEC] = __uint_as_float(r.w);
        } else {
          f[0] = __uint_as_float(r.x << 16); f[1 % CVEC] = __uint_as_float(r.x & 0xffff0000u);
          f[2 % CVEC] = __uint_as_float(r.y << 16); f[3 % CVEC] = __uint_as_float(r.y & 0xffff0000u);
          f[4 % CVEC] = __uint_as_float(r.z << 16); f[5 % CVEC] = __uint_as_float(r.z & 0xffff0000u);
          f[6 % CVEC] = __uint_as_float(r.w << 16); f[7 % CVEC] = __uint_as_float(r.w & 0xffff0000u);
        }
      };
      float qv[CVEC];
#pragma unroll
      for (int j = 0; j < CVEC; ++j) qv[j] = sq[part * CVEC + j];
      const float scale = 1.0f / sqrtf((float)DH);
      const int ctx = kvl;
      int base = s * CHUNK;
      float m = G1_NEG, lsum = 0.f, oacc[CVEC];
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
        lsum *= f;
#pragma unroll
        for (int j = 0; j < CVEC; ++j) oacc[j] *= f;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          const int key = base + w * WCH + i * KPW + slot;
          const float pr = key < ctx ? __expf(sc[i] - mn) : 0.f;
          lsum += pr;
          float vf[CVEC];
          widen(vraw[i], vf);
#pragma unroll
          for (int j = 0; j < CVEC; ++j) oacc[j] = fmaf(pr, vf[j], oacc[j]);
        }
        m = mn;
        base += NS * CHUNK;
        if (base >= ctx) break;  // block-uniform
        issue_kv(p, base, kvl);
      }
      // merge the KPW key slots of the wave, then the 4 waves through LDS (qkv_attn1_kernel's order)
#pragma unroll
      for (int o = LPK; o < 8; o <<= 1) {
        lsum += __shfl_xor(lsum, o, 64);
#pragma unroll
        for (int j = 0; j < CVEC; ++j) oacc[j] += __shfl_xor(oacc[j], o, 64);
      }
      if constexpr (LPK <= 8) {
        lsum += dpp_f32<0x128>(lsum);
#pragma unroll
        for (int j = 0; j < CVEC; ++j) oacc[j] += dpp_f32<0x128>(oacc[j]);
      }
      if constexpr (LPK <= 16) {
        lsum = rows4_sum(lsum);
#pragma unroll
        for (int j = 0; j < CVEC; ++j) oacc[j] = rows4_sum(oacc[j]);
      } else {
        lsum += __shfl_xor(lsum, 32, 64);
#pragma unroll
        for (int j = 0; j < CVEC; ++j) oacc[j] += __shfl_xor(oacc[j], 32, 64);
      }
      if (slot == 0) {
        if (part == 0) {
          sm_m[w] = m;
          sm_l[w] = lsum;
        }
#pragma unroll
        for (int j = 0; j < CVEC; ++j) sm_o[w * DH + part * CVEC + j] = oacc[j];
      }
      g1_lds_barrier();
      if (tid < DH || tid == PS_T - 1) {
        const float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
        float f[4];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) f[ww] = __expf(sm_m[ww] - M);
        gran_t* gp = G + G_PART + (h * NS + s) * (2 + DH);
        if (tid < DH) {
          float o = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) o = fmaf(sm_o[ww * DH + tid], f[ww], o);
          gran_store(gp + 2 + tid, epoch, o);
          if (glocal) gran_store_local(gp + (G_PARTL - G_PART) + 2 + tid, epoch, __float_as_uint(o));
        } else {
          float Ls = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) Ls = fmaf(sm_l[ww], f[ww], Ls);
          gran_store(gp + 0, epoch, M);
          gran_store(gp + 1, epoch, Ls);
          if (glocal) {
            gran_store_local(gp + (G_PARTL - G_PART) + 0, epoch, __float_as_uint(M));
            gran_store_local(gp + (G_PARTL - G_PART) + 1, epoch, __float_as_uint(Ls));
          }
        }
      }
    }
    if constexpr (PF == 0 || PF == 3) issue_wo(p);

    // ======== (3) merge of the head's NS partials + the new token's own key, for this workgroup's QR output columns (wave 0) =====
    if (w == 0) {
      // lanes 0 .. NS-1: (m, l) of split j = lane; lanes NS .. NS + NS*QR/2 - 1: two of the QR columns of split j
      const gran_t* gp = G + G_PART + (size_t)h * NS * (2 + DH);
      constexpr int NL = NS + NS * QR / 2;
      static_assert(NL <= 64 && QR % 2 == 0, "one wave gathers the partial pieces");
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
        const unsigned bo = (unsigned)((const char*)(gp + (size_t)j * (2 + DH) + off) - (const char*)GB);
        gather_two_dual(rs, bo, bo + (unsigned)(G_PARTL - G_PART) * 8u, glocal, epoch, t2, sp);
      }
      if (lane < NS) {
        spm[lane] = t2[0];
        spl[lane] = t2[1];
      } else if (lane < NL) {
        const int t = lane - NS;
        spo[j * QR + 2 * (t % (QR / 2))] = t2[0];
        spo[j * QR + 2 * (t % (QR / 2)) + 1] = t2[1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pt_end<TR>(pt, sp.passes);
      // the chain's PRO_ATTN_SELF prologue for thread (head-local index s): gemv1.hip gemv1s_kernel, EPT = QR, NS splits
      float tq = 0.f;
      {
        const int i = lane & 15;  // DH / EPT = 16 threads per head in the chain: lane i holds elements [4 i, 4 i + 4)
#pragma unroll
        for (int e = 0; e < QR; ++e) tq = fmaf(sq[(i * QR + e) % DH], sk[(i * QR + e) % DH], tq);
      }
      const float sself = head_group_sum64(tq, DH / QR) * (1.0f / sqrtf((float)DH));
      float M, Ls, acc[QR];
      if constexpr (D2 && NS == 16 && QR == 4) {
        // one split per lane of a 16-lane DPP row (every row of the wave computes the same): one exponential instruction for the 16
        // splits instead of 16 in a row, the sums by row reductions -- ~50 instructions instead of ~150 on the merging wave's path
        const int q = lane & 15;
        const float mq = spm[q], lq = spl[q];
        const f32x4v_t oq = *reinterpret_cast<const f32x4v_t*>(spo + q * QR);
        M = fmaxf(row16_max_dpp(mq), sself);
        const float f = __expf(mq - M), fs = __expf(sself - M);
        Ls = row16_sum_dpp(lq * f) + fs;
        acc[0] = fmaf(sv[s * QR + 0], fs, row16_sum_dpp(oq.x * f));
        acc[1] = fmaf(sv[s * QR + 1], fs, row16_sum_dpp(oq.y * f));
        acc[2] = fmaf(sv[s * QR + 2], fs, row16_sum_dpp(oq.z * f));
        acc[3] = fmaf(sv[s * QR + 3], fs, row16_sum_dpp(oq.w * f));
      } else {
        M = spm[0];
#pragma unroll
        for (int q = 1; q < NS; ++q) M = fmaxf(M, spm[q]);
        M = fmaxf(M, sself);
        Ls = 0.f;
#pragma unroll
        for (int e = 0; e < QR; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const float f = __expf(spm[q] - M);
          Ls = fmaf(spl[q], f, Ls);
#pragma unroll
          for (int e = 0; e < QR; ++e) acc[e] = fmaf(spo[q * QR + e], f, acc[e]);
        }
        {
          const float f = __expf(sself - M);
          Ls += f;
#pragma unroll
          for (int e = 0; e < QR; ++e) acc[e] = fmaf(sv[s * QR + e], f, acc[e]);
        }
      }
      const float inv = D2 ? __builtin_amdgcn_rcpf(Ls) : 1.0f / Ls;
#pragma unroll
      for (int e = 0; e < QR; ++e) acc[e] *= inv;
      if constexpr (!apack) {
        float outv = 0.f;
#pragma unroll
        for (int e = 0; e < QR; ++e) outv = lane == e ? acc[e] : outv;
        if (lane < QR) gran_store(G + G_ATT + h * DH + s * QR + lane, epoch, outv);
      } else {  // bf16 pairs (the chain with act_bf16 & 1 rounds the merged row the same way)
        const unsigned pk = lane == 0 ? pack_bf16x2(acc[0], acc[1]) : pack_bf16x2(acc[2], acc[3]);
        if (lane < 2) gran_store_bits(G + G_ATT + (h * DH + s * QR) / 2 + lane, epoch, pk);
      }
    }

    // The new token's K / V elements go into the cache HERE, write-through (agent scope: readable by the next step of a multi-step
    // launch, see issue_kv): the store's acknowledgement from memory (~1 us) counts in this wave's vmcnt like a load, and the next
    // wait is the attention-output sweep, an all-to-all edge that takes longer than that anyway.  Right after the in-projection
    // (where the launch chain stores them) it sat in front of the q/k/v sweep, an XCD-local edge of 0.5 us: +0.2 us per layer.
    if (lane < RQ) {
      const int r = w * RQ + lane, which = r / QR, e = s * QR + (r % QR);
      if (which != 0) {
        CT PS_GLOBAL* dst = as_gw<CT>(which == 1 ? p.kc : p.vc) + ((int64_t)h * ctx_max + kvl) * DH + e;
        if constexpr (sizeof(CT) == 2) __hip_atomic_store(reinterpret_cast<uint16_t*>((unsigned long long)dst), f32_to_bf16(kv_new), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(reinterpret_cast<float*>((unsigned long long)dst), kv_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ======== (4) out-proj + residual of rows 4c .. 4c+3 ==========================================================================
    g1_lds_barrier();  // waves 1 .. 3 do not sweep the attention edge while wave 0 still merges: their polls would sit in front of
                       // its loads in the CU's memory queue (measured: 257 -> 241 us per step)
    {
      constexpr int NA = apack ? EPT / 2 : EPT;
      float av[EPT], raw[NA];
      pt_begin<TR>(pt);
      if constexpr (PF == 3) issue_w1_rows(p, 0, R1 - 1);  // in place of most of the nap: ~6 MB chip-wide land inside the edge's own latency
      nap(nap_att);
      if constexpr (PF == 1) ps_gather<NA>(GB, rs, G + G_ATT + tid * NA, epoch, raw, sp, [&]() { issue_w1(p); });
      else ps_gather<NA>(GB, rs, G + G_ATT + tid * NA, epoch, raw, sp, PsNoop());
      if constexpr (D2) {  // the row as bf16 in LDS (already bf16 pairs when the edge travels packed)
        static_assert(EPT == 4, "two bf16 pairs per thread");
        u32x2v_t pk;
        if constexpr (apack) pk = u32x2v_t{__float_as_uint(raw[0]), __float_as_uint(raw[1])};
        else pk = u32x2v_t{pack_bf16x2(raw[0], raw[1]), pack_bf16x2(raw[2], raw[3])};
        *reinterpret_cast<u32x2v_t*>(reinterpret_cast<unsigned char*>(sx) + tid * 8) = pk;
      } else {
        if constexpr (apack) {
#pragma unroll
          for (int k = 0; k < EPT / 2; ++k) {
            const unsigned u = __float_as_uint(raw[k]);
            av[2 * k] = __uint_as_float(u << 16);
            av[2 * k + 1] = __uint_as_float(u & 0xffff0000u);
          }
        } else {
#pragma unroll
          for (int k = 0; k < EPT; ++k) av[k] = raw[k];
        }
        store_ept_lds<EPT>(sx + tid * EPT, av);
      }
      g1_lds_barrier();
      pt_end<TR>(pt, sp.passes);
      if constexpr (PF == 2) issue_w1(p);
      __builtin_amdgcn_sched_barrier(0);
      float mine;
      if constexpr (D2) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sx, xb);
        mine = ps_wave_sum_fast(ps_dot_bf16<NCH>(wo, xb));
      } else {
        float x[NCH][VEC];
        g1_read_shared<T, NCH>(sx, x);
        mine = wave_sum_dpp(g1_dot<T, NCH>(wo, x));
      }
      if (lane == 0) {
        const float v = W8 ? fmaf(mine, sco, bo_v) : mine + bo_v;
        gran_store(G + G_X2 + 4 * c + w, epoch, sres[w] + v);
      }
    }
    if constexpr (PF == 0) issue_w1(p);
    if constexpr (PF == 3) {
      issue_w1_rows(p, R1 - 1, R1);
      issue_w2_chunks(p, 0, NCH2 / 2);
    }

    // ======== (5) LN2 + linear1 + ReLU of rows 16c .. 16c+15 ======================================================================
    {
      pt_begin<TR>(pt);
      nap(nap_x2);
      if constexpr (PF == 1) ps_gather<EPT>(GB, rs, G + G_X2 + tid * EPT, epoch, xv, sp, [&]() { issue_w2(p); });
      else ps_gather<EPT>(GB, rs, G + G_X2 + tid * EPT, epoch, xv, sp, PsNoop());
      pt_end<TR>(pt, sp.passes);
      if constexpr (PF == 2) issue_w2(p);
      __builtin_amdgcn_sched_barrier(0);
      if (tid == c) store_ept_lds<EPT>(sres, xv);
      float ln_mean = 0.f, ln_rstd = 1.f;
      if constexpr (LF) fold_stats(xv, g2v, ln_mean, ln_rstd);
      else g1_block_layernorm<D, PS_T>(xv, g2v, be2v, sx, red);
      float mine = 0.f;
      if constexpr (D2) {
        u32x4_t xb[NCH];
        ps_read_bf16<NCH>(sx, xb);
        float t[R1];
#pragma unroll
        for (int r = 0; r < R1; ++r) t[r] = ps_dot_bf16<NCH>(w1[r], xb);
        mine = ps_wave_sums_fast<R1>(t);  // lane r < R1: row r
      } else {
        float x[NCH][VEC];
        g1_read_shared<T, NCH>(sx, x);
#pragma unroll
        for (int r = 0; r < R1; ++r) {
          const float t = wave_sum_dpp(g1_dot<T, NCH>(w1[r], x));
          mine = lane == r ? t : mine;
        }
      }
      const float d1 = W8 ? mine * sc1 : mine;
      const float hval = fmaxf(LF ? fmaf(ln_rstd, fmaf(-ln_mean, sg1_v, d1), b1_v) : (W8 ? fmaf(mine, sc1, b1_v) : mine + b1_v), 0.f);
      if constexpr (!hpack) {
        if (lane < R1) gran_store(G + G_HID + 4 * R1 * c + w * R1 + lane, epoch, hval);
      } else {  // two bf16 values per granule: half the bytes of the widest edge (the batched path keeps the hidden rows in bf16 too)
        const float nb = dpp_f32<0xB1>(hval);  // lane ^ 1
        if (lane < R1 && (lane & 1) == 0) gran_store_bits(G + G_HID + (4 * R1 * c + w * R1 + lane) / 2, epoch, pack_bf16x2(hval, nb));
      }
    }
    if constexpr (PF == 0) issue_w2(p);
    if constexpr (PF == 3) issue_w2_chunks(p, NCH2 / 2, NCH2);

    // ======== (6) linear2 + residual of rows 4c .. 4c+3 ===========================================================================
    {
      constexpr int NHG = hpack ? EPT2 / 2 : EPT2;
      float hv[EPT2], raw[NHG];
      pt_begin<TR>(pt);
      nap(nap_hid);
      if constexpr (PF == 1) ps_gather<NHG>(GB, rs, G + G_HID + tid * NHG, epoch, raw, sp, [&]() { issue_wqkv(pn, last); });
      else ps_gather<NHG>(GB, rs, G + G_HID + tid * NHG, epoch, raw, sp, PsNoop());
      if constexpr (D2) {  // the 16 hidden values of this thread as 8 bf16 pairs: 32 bytes of the bf16 row
        static_assert(EPT2 == 16, "eight bf16 pairs per thread");
        unsigned pk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if constexpr (hpack) pk[k] = __float_as_uint(raw[k]);
          else pk[k] = pack_bf16x2(raw[2 * k], raw[2 * k + 1]);
        }
        unsigned char* dst = reinterpret_cast<unsigned char*>(sx) + tid * 32;
        *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4_t*>(dst + 16) = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
      } else {
        if constexpr (hpack) {
#pragma unroll
          for (int k = 0; k < EPT2 / 2; ++k) {
            const unsigned u = __float_as_uint(raw[k]);
            hv[2 * k] = __uint_as_float(u << 16);
            hv[2 * k + 1] = __uint_as_float(u & 0xffff0000u);
          }
        } else {
#pragma unroll
          for (int k = 0; k < EPT2; ++k) hv[k] = raw[k];
        }
        store_ept_lds<EPT2>(sx + tid * EPT2, hv);
      }
      g1_lds_barrier();
      pt_end<TR>(pt, sp.passes);
      if constexpr (PF == 2) issue_wqkv(pn, last);
      __builtin_amdgcn_sched_barrier(0);
      float mine;
      if constexpr (D2) {
        u32x4_t xb[NCH2];
        ps_read_bf16<NCH2>(sx, xb);
        mine = ps_wave_sum_fast(ps_dot_bf16<NCH2>(w2, xb));
      } else {
        float x[NCH2][VEC];
        g1_read_shared<T, NCH2>(sx, x);
        mine = wave_sum_dpp(g1_dot<T, NCH2>(w2, x));
      }
      if (lane == 0) {
        const float v = W8 ? fmaf(mine, sc2, b2_v) : mine + b2_v;
        gran_store(G + GPL + G_X + 4 * c + w, epoch, sres[w] + v);  // the next layer's x edge (layer L: the final norm's)
      }
    }
    if constexpr (PF == 0 || PF == 3) issue_wqkv(pn, last);
  }

  // ======== final norm + predict layer: rows 4c .. 4c+3 (+ row 1024) ================================================================
  {
    gran_t* const G = a.gran + (size_t)a.L * GPL;
    pt_begin<TR>(pt);
    nap(nap_x);
    ps_gather<EPT>(GB, rs, G + G_X + tid * EPT, epoch, xv, sp, PsNoop());
    pt_end<TR>(pt, sp.passes);
    float ln_mean = 0.f, ln_rstd = 1.f;
    if constexpr (LF) fold_stats(xv, g1v, ln_mean, ln_rstd);
    else g1_block_layernorm<D, PS_T>(xv, g1v, be1v, sx, red);
    float x[D2 ? 1 : NCH][VEC];
    u32x4_t xb[D2 ? NCH : 1];
    float t0;
    if constexpr (D2) {
      ps_read_bf16<NCH>(sx, xb);
      t0 = ps_wave_sum_fast(ps_dot_bf16<NCH>(wq[0], xb));
    } else {
      g1_read_shared<T, NCH>(sx, x);
      t0 = wave_sum_dpp(g1_dot<T, NCH>(wq[0], x));
    }
    constexpr int G_LOG = G_QKV;  // the final block's q/k/v slots carry the logits edge (V <= 3 D)
    if (lane == 0) {
      const float d0 = W8 ? t0 * scq : t0;
      const float lg = LF ? fmaf(ln_rstd, fmaf(-ln_mean, sgq, d0), bq) : d0 + 0.f;
      a.logits[4 * c + w] = lg;
      if (own_sample) gran_store(G + G_LOG + 4 * c + w, epoch, lg);
    }
    if (extra_row) {
      float t1;
      if constexpr (D2) t1 = ps_wave_sum_fast(ps_dot_bf16<NCH>(wq[1], xb));
      else t1 = wave_sum_dpp(g1_dot<T, NCH>(wq[1], x));
      if (lane == 0) {
        const float dx = W8 ? t1 * scx : t1;
        const float lg = LF ? fmaf(ln_rstd, fmaf(-ln_mean, sgx, dx), tbx) : dx + 0.f;
        a.logits[4 * NWG] = lg;
        if (own_sample) gran_store(G + G_LOG + 4 * NWG, epoch, lg);
      }
    }
    if (own_sample) {
      // ======== sampling, stop rule, next input row (ar_sample_kernel, sampling.hip; valle/models/valle.py:1039-1057) ==============
      // Every workgroup gathers the V logits and draws the SAME token (same logits, same Philox counter), so nobody waits for a
      // "sampler": the step-to-step dependence costs one all-to-all edge.  The next step's first operands are requested first.
      float* const slog = smem + 8 * 1024;   // [V] the row, then regrouped SAMP_PER per thread
      unsigned long long* const red64 = reinterpret_cast<unsigned long long*>(smem + 10 * 1024);
      int* const redi = reinterpret_cast<int*>(smem + 10 * 1024 + 16);
      float* const redf = smem + 10 * 1024 + 32;
      float* const wave_tot = smem + 10 * 1024 + 48;
      static_assert(SAMP_T == PS_T && SAMP_T * SAMP_PER >= 4 * NWG + 1, "the sampling code's block shape");
      const PStepSample q = ps_sample_load(a.smp);
      const ArDyn dyn = *q.dyn;
      const PsLayer p0 = ps_layer(a.layers, 0);
      issue_wqkv(p0, false);
      if constexpr (PF == 1 || PF == 2) issue_kv(p0, s * CHUNK, kvl + 1);  // (these schedules request a layer's keys during its x sweep)
      const int apos1 = apos + 1, kvl1 = kvl + 1;
      float pev[EPT];
      // (at the capacity guard apos1 can be one row past the table -- prepend_bos, full prompt, n_gen == max_gen --: that step stops
      //  and never uses the row, but the request is issued before the stop rule is known: keep it inside the table)
      const int pe_row = apos1 < q.pe_rows ? apos1 : q.pe_rows - 1;
      ps_load4(as_g<float>((unsigned long long)q.pe) + (int64_t)pe_row * D + tid * EPT, pev);
      const float alpha = *q.alpha_audio;
      float lg4[EPT], lgx;
      pt_begin<TR>(pt);
      nap(nap_x);
      gather_vals16_plus1<EPT>(rs, (unsigned)((const char*)(G + G_LOG + tid * EPT) - (const char*)GB),
                               (unsigned)((const char*)(G + G_LOG + 4 * NWG) - (const char*)GB), epoch, lg4, lgx, sp);
      pt_end<TR>(pt, sp.passes);
      store_ept_lds<EPT>(slog + tid * EPT, lg4);
      if (tid == 0) slog[4 * NWG] = lgx;
      __syncthreads();
      const int V = a.V;
      float raw[SAMP_PER];
#pragma unroll
      for (int j = 0; j < SAMP_PER; ++j) {
        const int idx = tid * SAMP_PER + j;
        raw[j] = idx < V ? slog[idx] : -INFINITY;
      }
      if (c == 0 && dyn.trace != nullptr && it < dyn.trace_cap) {
        float* tr = dyn.trace + (int64_t)it * V;
#pragma unroll
        for (int j = 0; j < SAMP_PER; ++j) {
          const int idx = tid * SAMP_PER + j;
          if (idx < V) tr[idx] = raw[j];
        }
      }
      const int argmax = argmax_row(raw, V, red64);
      const int sample = sample_row(raw, V, dyn.top_k, dyn.temperature, request_seed(dyn.seed, 0ull), (uint32_t)it, argmax,
                                    SampScratch{red64, redi, redf, wave_tot});
      // stop rule (valle.py:1044-1048) and bookkeeping: the same integers in every workgroup; workgroup 0 stores them
      int stop = (!dyn.ignore_eos && ((argmax == 1024) || (sample == 1024))) || (n_gen + q.bos > s_cap);
      if (dyn.max_new > 0 && n_gen >= dyn.max_new) stop = 1;
      if (dyn.has_forced) stop = n_gen >= dyn.forced_len[0];
      if (n_gen >= (int)q.g_stride || kvl1 >= ctx_max) stop = 1;  // capacity guard
      int next = sample;
      bool bad_id = false;
      if (!stop && dyn.has_forced) {
        const int64_t f = dyn.forced[n_gen];
        next = (int)f;
        if (f < 0 || f >= (int64_t)V + q.bos) {  // outside ar_audio_embedding: the reference's nn.Embedding raises IndexError
          next = 0;
          bad_id = true;
        }
      }
      if (c == 0 && tid == 0) {
        if (q.host_prog != nullptr) {  // [0] utterances done before this step, [1] sampling steps so far (ar_sample_kernel's words)
          const int sc = q.s.done_count[1] + (stop ? nsteps - step : 1);
          q.s.done_count[1] = sc;
          __hip_atomic_store(q.host_prog + 0, q.s.done_count[0] + (stop ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(q.host_prog + 1, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!stop) {
          if (bad_id && q.id_err) atomicOr(q.id_err, 4);
          q.tokens[n_gen] = next;
          q.sampled[n_gen] = sample;
          q.s.n_gen[0] = n_gen + 1;
          q.s.kv_len[0] = kvl1;
          q.s.audio_pos[0] = apos1;
        } else {
          if (n_gen < (int)q.g_stride) q.sampled[n_gen] = sample;  // the stopping iteration's own draw (e.g. EOS), for the hooks
          q.s.done[0] = 1;
          atomicAdd(q.s.done_count, 1);
        }
        q.s.iter[0] = it + 1;
      }
      if (stop) return;
      // next step's input: ar_audio_position(ar_audio_embedding(token))  (valle.py:1013-1015), the sampling kernel's roundings
      float ev[EPT];
      load_ept<EPT>(q.audio_emb + (int64_t)next * D + tid * EPT, ev);
#pragma unroll
      for (int k = 0; k < EPT; ++k) xv[k] = __fadd_rn(ev[k], __fmul_rn(alpha, pev[k]));
      if (c == 0) {
        float* xo = q.x + tid * EPT;
#pragma unroll
        for (int k = 0; k < EPT; ++k) xo[k] = xv[k];
      }
      n_gen += 1;
      apos = apos1;
      kvl = kvl1;
      it += 1;
      epoch = (unsigned)(it + 1);
    }
    pt_begin<TR>(pt);
    pt_end<TR>(pt, 0u);
  }
  }  // step
}

// Row constants of the folded LayerNorm: one wave per row of bf16 W[N][K]; fp64 sums (one-time work at engine set-up).
__global__ __launch_bounds__(256) void ps_fold_kernel(const bf16_t* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ bias, float* __restrict__ sg, float* __restrict__ tb, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const bf16_t* wr = W + (int64_t)n * K;
  double s = 0.0, t = 0.0;
  for (int k = lane; k < K; k += 64) {
    const double wv = (double)bf16_to_f32(wr[k].v);
    s += wv * (double)gamma[k];
    t += wv * (double)beta[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    t += __shfl_xor(t, o, 64);
  }
  if (lane == 0) {
    sg[n] = (float)s;
    tb[n] = (float)(t + (bias ? (double)bias[n] : 0.0));
  }
}

int launch_ps_fold(hipStream_t st, const void* W, const float* gamma, const float* beta, const float* bias, float* sg, float* tb, int N, int K) {
  if (!W || !gamma || !beta || !sg || !tb || N < 1 || K < 1) return -1;
  hipLaunchKernelGGL(ps_fold_kernel, dim3((N + 3) / 4), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(W), gamma, beta, bias, sg, tb, N, K);
  return 0;
}

bool pstep_supports(int dtype, int d, int nhead, int dh, int V) {
  return (dtype == DT_BF16 || dtype == DT_FP8W || dtype == DT_F32) && d == 1024 && nhead == 16 && dh == 64 && V > 1024 && V <= 1025;
}

size_t pstep_gran_count(int d, int nhead, int L) { return (size_t)(L + 1) * ps_gran_per_layer(d, nhead, 256 / nhead); }

// PK of a mode: bits 4 / 8: hidden / attention rows as bf16 pairs; 32: folded LayerNorm; 64: bf16 activation rows + v_dot2c (D2)
static int ps_pk_of(int mode) { return ((mode >> 2) & 3) | ((mode >> 3) & 4) | ((mode >> 3) & 8); }

// The instantiations that ship (round 6: the forms the measurement ladder of DESIGN.md 4.1 dropped -- 4 keys per lane, request
// schedules 1 / 2, the attention row as bf16 pairs, D2 / folded forms without the packed hidden row -- are no longer compiled):
//   bf16 weights   PK 0  fp32 rows, nothing packed, three-barrier LayerNorm   } bit-identical to the launch chain at the same
//                  PK 1  + hidden row as bf16 pairs                           } decomposition (tests/test_persist_gpu.py)
//                  PK 5  + folded LayerNorm
//                  PK 13 + bf16 activation rows on v_dot2c (the default; also with the in-kernel timeline)
//   fp8 weights    PK 1, PK 5 (the default)          fp32: PK 0
// each for 2 keys per lane and the request schedules 0 (all at once) and 3 (spread over three sweeps, the default).
typedef void (*PsKernel)(PStepArgs);
template <typename WT, int PK, bool TR = false>
static PsKernel ps_pf(int pf) {
  return pf == 3 ? (PsKernel)pstep_kernel<WT, 1024, 16, 2, 3, PK, TR> : pf == 0 && !TR ? (PsKernel)pstep_kernel<WT, 1024, 16, 2, 0, PK, false> : nullptr;
}
static PsKernel ps_select(int dtype, int mode, int nk, int pf, bool traced) {
  if (nk != 2) return nullptr;
  const int pk = ps_pk_of(mode);
  if (dtype == DT_FP8W) return traced ? nullptr : pk == 1 ? ps_pf<bf16w8t_t, 1>(pf) : pk == 5 ? ps_pf<bf16w8t_t, 5>(pf) : nullptr;
  if (dtype == DT_F32) return traced || pk != 0 ? nullptr : ps_pf<float, 0>(pf);
  if (dtype != DT_BF16) return nullptr;
  if (pk == 13 && !(mode & 16)) return nullptr;  // (the D2 forms were built on the XCD-local group edges only)
  if (traced) return pk == 13 ? ps_pf<bf16_t, 13, true>(pf) : nullptr;
  return pk == 0 ? ps_pf<bf16_t, 0>(pf) : pk == 1 ? ps_pf<bf16_t, 1>(pf) : pk == 5 ? ps_pf<bf16_t, 5>(pf) : pk == 13 ? ps_pf<bf16_t, 13>(pf) : nullptr;
}

// 1 = this (weight type, mode, schedule) is an instantiated form AND the occupancy calculator places (at least) one of ITS workgroups
// on a CU -- asked about the kernel that would be launched, not about a stand-in; 0 = no such form; -1 = it does not fit
int pstep_form_ok(int dtype, int mode, int nk, int pf, bool traced) {
  const PsKernel k = ps_select(dtype, mode, nk, pf, traced);
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
int launch_pstep(hipStream_t st, int dtype, const PStepArgs& a) {
  if (!pstep_supports(dtype, a.d, a.nhead, a.dh, a.V)) return 1;
  if (!a.layers || !a.x_in || !a.logits || !a.kv_len || !a.iter || !a.done || !a.gran || a.L < 1) return -1;
  if (a.nsteps < 0 || a.nsteps > 4096 || (a.nsteps > 0 && !a.smp)) return -1;
  const PsKernel k = ps_select(dtype, a.mode, a.nk, a.pf, a.ptrace != nullptr);
  if (k == nullptr) return -1;
  hipLaunchKernelGGL(k, dim3(256), dim3(PS_T), 0, st, a);
  return 0;
}

}  // namespace vle
