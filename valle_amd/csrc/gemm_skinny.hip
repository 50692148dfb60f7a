// AR-step linear layers for 2..64 utterances (bf16): weight-streaming MFMA GEMM with a skinny M.
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] ),   M = batch <= 64
//   reference ops, for ONE new token per utterance: in-proj / out-proj (valle/modules/activation.py:
//   414-421), linear1 / linear2 (valle/modules/transformer.py:332-334), ar_predict_layer (valle.py:1039).
//
// The step is HBM-bound on W (each weight byte is used by all M rows once) and a tiled GEMM is the
// wrong shape for it: with M = 64 a 64x64 tile grid has only N/64 = 16..64 workgroups, each walking
// K serially (measured 26-59 us per GEMM on gemm.hip).  Here the unit of work is ONE 16-row
// fragment of W:
//   * a workgroup owns W rows n0..n0+15; its 4 waves split K in four and combine through LDS,
//     so N/16 = 64..256 workgroups stream disjoint 32..128 KB slabs of W;
//   * W goes global -> registers directly as the MFMA A operand (no LDS: nothing shares it): per
//     64-deep k-chunk two 16-byte vectors per lane, the 4 lanes of a row covering one full 64-byte
//     sector per instruction;
//   * X (the M activation rows, bf16) is the B operand, read through L1/L2 in the same pattern;
//     v_mfma_f32_16x16x32_bf16 computes C^T[n][m], so a lane ends up with 4 consecutive output
//     columns of one utterance: vector epilogue, and K/V go straight into the cache slot (EPI QKV),
//     which removes the separate split kernel;
//   * up to 4 k-chunks (40 x 16-byte loads per lane at M = 64) are requested before the first MFMA;
//   * split-K across workgroups (gridDim.y = KS) when N/16 alone would leave most of the 256 CUs idle
//     (N = d: out-proj, FFN2, logits -> 64 row fragments): every slice writes its fp32 partial tile
//     (4 KB) to a workspace, takes a ticket on the fragment's counter, and the LAST arriver sums the
//     KS partials in slice order (deterministic, independent of arrival order) and runs the epilogue;
//     the counter resets itself, so a captured graph replays without a memset.
// The X operand is read fragment-major when the producer wrote it that way (x_xf, common.h xf_index): stored row-major a
// fragment load touches 16 rows x 64 B and the launch is bound by the texture path; fragment-major it is one contiguous
// 1 KB (measured at 64 utterances: AR loop 667 -> 613 ms).  W is read the same way from a fragment-major copy the
// engine makes at vle_finalize_weights for engines with max_batch >= 2 (w_packed; +304 MB at C2, 3 % of the AR loop).
// Measured and rejected (round 1, MI355X, M = 64, per launch in a dependent graph chain): staging the X slice in
// LDS once per workgroup (full-line loads, waves as 1-2 row fragments x k-ranges) 9.2-9.8 us vs 7.2-7.3 us here --
// the load -> LDS -> barrier -> ds_read prologue serialises what this kernel requests as one burst; LayerNorm fused
// into that prologue 13.1 us vs 1.9 (LayerNorm launch) + 7.2.  All four shapes cost ~7.2 us regardless of W size
// (2-8 MB): the launch is latency-bound on the X fragments (4x the bytes of W through each CU's texture path).
// Round 3 took that apart (tools/ubench_xload.hip, tools/ktrace_dist.py): the bare burst -- 32 KB of W from HBM + 128 KB of X from
// L2 per workgroup -- lands in 2.7 us; the other ~1.4 us ahead of the MFMAs were this kernel's own instruction stream (per-load
// layout decisions, three kernarg round trips, MFMAs waiting for the epilogue's operands).  The FAST bodies below fix the layout at
// compile time: QKV 6.1 -> 5.1, FFN1 6.2 -> 4.8, FFN2 7.2 -> 6.3 us per launch at 64 utterances (DESIGN.md 4.2 vi).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));
typedef float gs_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gs_u32x4 __attribute__((ext_vector_type(4)));

// epilogue of one lane: v = 4 consecutive output columns ncol..ncol+3 of row m (bias already added)
template <int EPI>
__device__ inline void gs_epilogue(const GemmSkinnyArgs& a, gs_f32x4 v, int m, int ncol, gs_f32x4 gamma4, gs_f32x4 old4, int kvl) {
  const int M = a.M, N = a.N;
  if (m >= M || ncol >= N) return;
  const bool vec = ncol + 3 < N && (N & 3) == 0;
  if constexpr (EPI == GS_EPI_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if constexpr (EPI == GS_EPI_STORE || EPI == GS_EPI_RELU) {
    if (a.out_xf != 0 && vec) {  // [M][N] is the next GEMM's X: fragment-major, 4 consecutive columns stay one 8-byte store
      gs_bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
      *reinterpret_cast<gs_bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + xf_index(m, ncol, (M + 15) >> 4, a.out_xf == 2)) = o4;
      return;
    }
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (int64_t)m * N + ncol;
    if (vec) {
      gs_bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
      *reinterpret_cast<gs_bf16x4*>(o) = o4;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ncol + r < N) store_elem<bf16_t>(o + r, v[r]);
    }
  } else if constexpr (EPI == GS_EPI_F32) {
    float* o = reinterpret_cast<float*>(a.out) + (int64_t)m * N + ncol;
    if (vec) *reinterpret_cast<gs_f32x4*>(o) = v;
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ncol + r < N) o[r] = v[r];
    }
  } else if constexpr (EPI == GS_EPI_RESID) {
    float* o = a.resid + (int64_t)m * N + ncol;
    if (vec) {
      const gs_f32x4 x4 = old4 + v;  // the residual's old value was requested with the kernel's first burst
      *reinterpret_cast<gs_f32x4*>(o) = x4;
      if (a.lnp.gamma != nullptr) {
        // producer side of the fused LayerNorm (kernels.h LnProducer): the 4 lanes fg = 0..3 of a row hold this workgroup's 16
        // columns of it (N % 16 == 0, so the four lanes are active together): bf16(x * gamma_next) into the next GEMM's X,
        // and the group's (mean, M2) into slot blockIdx.x of the row
        gs_bf16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (__bf16)(x4[r] * gamma4[r]);
        *reinterpret_cast<gs_bf16x4*>(reinterpret_cast<bf16_t*>(a.lnp.xg_out) + xf_index(m, ncol, a.lnp.MF, a.lnp.w8 != 0)) = o4;
        const float mean = rows4_sum((x4[0] + x4[1]) + (x4[2] + x4[3])) * (1.0f / 16.0f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = x4[r] - mean;
          q = fmaf(t, t, q);
        }
        q = rows4_sum(q);
        if ((threadIdx.x & 63) < 16) {
          typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<gs_f32x2*>(a.lnp.stats_out + ((int64_t)m * (N >> 4) + (ncol >> 4)) * 2) = gs_f32x2{mean, q};
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ncol + r < N) o[r] += v[r];
    }
  } else {  // GS_EPI_QKV: rows [0,d) = Q, [d,2d) = K, [2d,3d) = V (valle/modules/activation.py:128-130); d % 4 == 0
    const int d = N / 3, which = ncol / d, j = ncol - which * d;
    if (which == 0) {
      *reinterpret_cast<gs_f32x4*>(a.q_out + (int64_t)m * d + j) = v;
    } else {
      const int h = j / a.dh, e = j - h * a.dh;  // dh % 4 == 0: the 4 columns stay inside one head
      const int64_t off = (((int64_t)m * a.nhead + h) * a.ctx_max + kvl) * a.dh + e;  // kv_len[m], requested up front
      gs_bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
      *reinterpret_cast<gs_bf16x4*>(reinterpret_cast<bf16_t*>(which == 1 ? a.k_cache : a.v_cache) + off) = o4;
    }
  }
}

// 16-byte write-through (sc1) store / L1-bypassing (sc1) loads of the split-K hand-off.  The guide's pitfall 7 ("bulk publish 3-20x
// slow: data held 16 B per lane re-issued as narrow sc1 stores"): a dword sc1 store is one fabric write, ~6x the dwordx4 time per
// byte.  hipcc has no 16-byte atomic: inline asm (the compiler does not count these operations -- the store is drained by the
// explicit vmcnt(0) that follows it, the loads carry their own wait).
__device__ inline void gs_store16_wt(float* p, gs_f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ inline void gs_load16x4_sc1(const float* p0, const float* p1, const float* p2, const float* p3, gs_f32x4& t0, gs_f32x4& t1,
                                       gs_f32x4& t2, gs_f32x4& t3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}

// granule form of a lane's 4 partial sums: {tag, v0, tag, v1} {tag, v2, tag, v3}, two write-through 16-byte stores (each 8-byte
// half validates itself, so a torn 16-byte write is still never read as valid-with-old-data)
__device__ inline void gs_store_gran(unsigned long long* p, gs_f32x4 v, unsigned tag) {
  const gs_u32x4 a = gs_u32x4{tag, __float_as_uint(v[0]), tag, __float_as_uint(v[1])};
  const gs_u32x4 b = gs_u32x4{tag, __float_as_uint(v[2]), tag, __float_as_uint(v[3])};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(p), "v"(a), "v"(b) : "memory");
}
__device__ inline void gs_load_gran(const unsigned long long* p, gs_u32x4& a, gs_u32x4& b) {  // no wait: the caller drains vmcnt
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}

// FP8W: 8 e4m3fn weights (two dwords) -> the bf16 MFMA operand; exact (every e4m3fn value is a bf16 value)
__device__ inline gs_bf16x8 gs_fp8x8_to_bf16(unsigned int lo, unsigned int hi) {
  typedef float f32x2v_t __attribute__((ext_vector_type(2)));
  const f32x2v_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
  const f32x2v_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
  gs_bf16x8 o;
  o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)b.x; o[3] = (__bf16)b.y;
  o[4] = (__bf16)c.x; o[5] = (__bf16)c.y; o[6] = (__bf16)d.x; o[7] = (__bf16)d.y;
  return o;
}

// W8 (engine mode FP8W): W is e4m3fn [N][K] with one power-of-two scale per row (applied in the epilogue, exact).
// A lane then takes ONE 16-byte vector per 64-deep chunk = the 16 consecutive k of its quarter of the chunk; the two
// MFMA k-halves use bytes 0-7 and 8-15, and X is read with the same k assignment (any k permutation is a valid dot
// product as long as both operands agree), so W is still fetched as full 64-byte sectors per row.
// FAST (round 3): the production configuration -- fragment-major W and X, whole rounds of 4 chunks per wave, N % 16 == 0, no
// diagnostics -- with every layout decision taken at compile time.  The general body below re-decides them per load (uniform
// branches on kernel arguments, 64-bit selects between the row-major and fragment-major addresses, clamps): ~6 instructions
// and a branch per 16-byte load, ~1200 instructions ahead of the first MFMA, executed by the ONE wave a SIMD holds -- the
// "burst" left the CU over ~1.5 us (tools/ubench_xload.hip: the same 160 KB requested back to back lands in ~1.2 us).
// NF (FAST bodies, NF * MF <= 4): W row fragments per workgroup.  At d = 1536 the grids are 288 / 384 workgroups on 256 CUs, so
// some CUs host two workgroups and each pulls the whole X again (2 x (24 + 96) KB at 32 utterances, fp8 W); with NF = 2 a
// workgroup streams 2 x 16 rows of W against ONE pass over X (48 + 96 KB) and the grid (144 / 192) fits the chip.  Wave w then
// finishes (W fragment w / MF, row fragment w % MF) -- the per-fragment arithmetic is unchanged (bit-identical).
template <int MF, int EPI, bool W8, int FAST = 0, int NF = 1>  // FAST: 0 = general body, 1 = whole rounds of 4 chunks, 2 = + one round of 2
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmSkinnyArgs a) {
  static_assert(NF == 1 || (FAST != 0 && NF * MF <= 4), "NF > 1: FAST bodies only, one finishing wave per (W fragment, row fragment)");
  // k-chunks (64 deep) requested per round
  constexpr int G = 4;  // (8 / 16 at M <= 32 / 16 measured slower: 256 VGPRs + AGPR spills leave one workgroup per CU)
  __shared__ __attribute__((aligned(16))) float red[4][NF * MF][64][4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (FAST) {
    // one kernarg round trip: every argument the load burst needs is requested with the first s_load batch (left alone the
    // compiler fetched the segment in three dependent steps, ~0.1-0.2 us each, ahead of the first global load)
    asm volatile("" ::"s"(a.x), "s"(a.w), "s"(a.bias), "s"(a.wscale), "s"(a.K), "s"(a.N), "s"(a.M), "s"(a.lnc.stats), "s"(a.lnc.wg),
                 "s"(a.resid), "s"(a.lnp.gamma), "s"(a.kv_len), "s"(a.kt.buf), "s"(a.ks_grid), "s"(a.gran_epoch), "s"(a.ws_gran),
                 "s"(a.gran_idx), "s"(a.ws_part), "s"(a.ws_cnt));
  }
  const unsigned long long kt0 = ktrace_begin(a.kt);
  const int fr = lane & 15, fg = lane >> 4;
  const int ff = NF > 1 ? min(wave / MF, NF - 1) : 0;  // the W fragment (of this workgroup's NF) whose tile this wave finishes
  const int n0 = ((int)blockIdx.x * NF + ff) * 16;
  const int K = a.K, N = a.N, M = a.M;
  const int KS = FAST ? a.ks_grid : (int)gridDim.y, ks = blockIdx.y;
  // granule hand-off: this launch's epoch word (written by the previous step's sampling kernel), requested with the other
  // epilogue operands behind the burst -- as a per-lane vector load: forced into an SGPR up here it cost a full round trip
  // ahead of the first weight request
  int gran_ep = 0;
  const int Kw = K / (4 * KS);  // this wave's share of K (multiple of 64)
  // Every workgroup reads ALL of X (<= 128 KB, L2-resident) while streaming its own 16 rows of W.  If all workgroups walked X
  // in the same order they would hit the same L2 channels at the same moment; so the K quarter a wave takes and the order of
  // the row fragments inside a chunk are rotated by the workgroup index (sums stay in K order: the combine indexes by quarter).
  const int wq = (!FAST && a.rot) ? (wave + (int)blockIdx.x) & 3 : wave;
  const int frot = (!FAST && a.rot) ? ((int)blockIdx.x >> 2) & 3 : 0;
  auto fi = [&](int i) -> int { return FAST ? i : MF == 4 ? (i ^ frot) : MF == 2 ? (i ^ (frot & 1)) : MF == 3 ? (i + frot) % 3 : 0; };
  const int kbeg = (ks * 4 + wq) * Kw;
  const int nrow = min(n0 + fr, N - 1);
  constexpr int KOFS = W8 ? 16 : 8;   // first k of this lane inside a chunk = fg * KOFS
  constexpr int SSTEP = W8 ? 8 : 32;  // k distance between the lane's two MFMA k-halves
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.w) + (int64_t)nrow * K + kbeg + fg * 8;            // bf16 W
  const unsigned char* wp8 = reinterpret_cast<const unsigned char*>(a.w) + (int64_t)nrow * K + kbeg + fg * 16;  // fp8 W
  const bf16_t* xp[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) xp[i] = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)min(fi(i) * 16 + fr, M - 1) * K + kbeg + fg * KOFS;
  const bool xfrag = a.x_xf != 0;  // X stored fragment-major (common.h xf_index): one contiguous 1 KB per fragment load
  const bf16_t* xfb = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)(kbeg / 64) * 2 * MF * 512 + lane * 8;

  // epilogue operands: requested right AFTER the first round's W / X loads (vmcnt retires in issue order, and the MFMAs need
  // only W and X: issued first, the 32-40 KB of group statistics / bias / residual per workgroup used to sit in the texture
  // path ahead of the weight requests -- round 3, same finding as gemv1.hip's block-shared activations)
  const int ncol = n0 + fg * 4;  // first of this lane's 4 output columns
  gs_f32x4 bias4 = gs_f32x4{0.f, 0.f, 0.f, 0.f};
  gs_f32x4 scale4 = gs_f32x4{1.f, 1.f, 1.f, 1.f};
  gs_f32x4 wg4 = gs_f32x4{0.f, 0.f, 0.f, 0.f}, gamma4 = gs_f32x4{1.f, 1.f, 1.f, 1.f};
  constexpr int LN_MAXQ = 16;  // float4 (= 2 slots) per lane: rows up to 4 * 32 * 16 = 2048 wide
  gs_f32x4 lst[LN_MAXQ];
  constexpr bool kFastLn = EPI == GS_EPI_QKV || EPI == GS_EPI_RELU || EPI == GS_EPI_F32;
  const bool ln_in = FAST ? kFastLn : a.lnc.stats != nullptr;
  // float4 per lane = (nslots / 4 lanes) / 2.  Kept a run-time value in the FAST body too (there it is 8): the merge loops of
  // the epilogue then compile to the same instruction sequence in both bodies -- with a constant trip count hipcc contracted
  // the mul / add chains into different fma's and the two bodies differed in the last bit of rstd (3e-3 on the logits)
  const int ln_nq = ln_in ? a.lnc.nslots >> 3 : 0;
  gs_f32x4 old4 = gs_f32x4{0.f, 0.f, 0.f, 0.f};
  int kvl = 0;
  auto request_epilogue_operands = [&]() {
    if constexpr (FAST) {
      // no branches: the compiler then knows how many loads follow the W / X burst and lets the MFMAs wait for the burst only
      // (vmcnt(n)); behind a uniform branch it waited for these operands too.  FAST launches have a bias, LayerNorm statistics
      // exactly when the epilogue is one that consumes them (64 / 96 slots: 8 / 12 vectors per lane), and rows are clamped, not skipped.
      constexpr bool LN = EPI == GS_EPI_QKV || EPI == GS_EPI_RELU || EPI == GS_EPI_F32;
      const int mrow = min((NF > 1 ? wave % MF : wave) * 16 + fr, M - 1);
      bias4 = *reinterpret_cast<const gs_f32x4*>(a.bias + ncol);
      if constexpr (W8) scale4 = *reinterpret_cast<const gs_f32x4*>(a.wscale + ncol);
      if constexpr (LN) {
        wg4 = *reinterpret_cast<const gs_f32x4*>(a.lnc.wg + ncol);
        constexpr int NSLOTS = FAST == 2 ? 96 : 64;  // the launcher admits exactly these (d = 1024 / 1536)
        const float* sp = a.lnc.stats + (int64_t)mrow * (2 * NSLOTS) + fg * 4;
#pragma unroll
        for (int j = 0; j < NSLOTS / 8; ++j) lst[j] = *reinterpret_cast<const gs_f32x4*>(sp + j * 16);
      }
      if constexpr (EPI == GS_EPI_RESID) {
        gamma4 = *reinterpret_cast<const gs_f32x4*>((a.lnp.gamma != nullptr ? a.lnp.gamma : a.bias) + ncol);
        old4 = *reinterpret_cast<const gs_f32x4*>(a.resid + (int64_t)mrow * N + ncol);
      }
      if constexpr (EPI == GS_EPI_QKV) kvl = a.kv_len[mrow];
      if constexpr (EPI == GS_EPI_RESID)  // (the only epilogue the engine splits across workgroups)
        gran_ep = *(a.gran_epoch != nullptr ? a.gran_epoch : reinterpret_cast<const int32_t*>(a.bias));
      return;
    }
    if (a.ws_gran != nullptr) gran_ep = a.gran_epoch[0];
    if (a.bias != nullptr) {
      if (FAST || ncol + 3 < N) bias4 = *reinterpret_cast<const gs_f32x4*>(a.bias + ncol);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = ncol + r < N ? a.bias[ncol + r] : 0.f;
      }
    }
    if constexpr (W8) {
      if (FAST || ncol + 3 < N) scale4 = *reinterpret_cast<const gs_f32x4*>(a.wscale + ncol);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) scale4[r] = ncol + r < N ? a.wscale[ncol + r] : 1.f;
      }
    }
    // fused LayerNorm (kernels.h): consumer operands (wg of this lane's 4 columns; the group statistics of the row this lane
    // finishes, 16 rows per wave, the lane's quarter of the row's slots) and the producer's gamma
    if (ln_in) {
      if (FAST || ncol + 3 < N) wg4 = *reinterpret_cast<const gs_f32x4*>(a.lnc.wg + ncol);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) wg4[r] = ncol + r < N ? a.lnc.wg[ncol + r] : 0.f;
      }
      if (wave < MF) {
        // lane (fr, fg) takes the slot pairs 4 j + fg, j = 0 .. nslots / 8: the four lanes of a row read one contiguous 64 bytes
        // per instruction (16 segments per wave-load; a lane walking its own quarter of the row touched 64 lines per instruction)
        const float* sp = a.lnc.stats + ((int64_t)min(wave * 16 + fr, M - 1) * a.lnc.nslots) * 2 + fg * 4;
#pragma unroll
        for (int j0 = 0; j0 < LN_MAXQ; j0 += 4)
          if (j0 < ln_nq) {
#pragma unroll
            for (int j = j0; j < j0 + 4; ++j)
              if (FAST || j < ln_nq) lst[j] = *reinterpret_cast<const gs_f32x4*>(sp + j * 16);
          }
      }
    }
    if constexpr (EPI == GS_EPI_RESID) {
      if (a.lnp.gamma != nullptr && (FAST || ncol + 3 < N)) gamma4 = *reinterpret_cast<const gs_f32x4*>(a.lnp.gamma + ncol);
      // the residual row this lane finishes (wave i: fragment i): nobody else writes it during this launch
      if (wave < MF && wave * 16 + fr < M && (FAST || (ncol + 3 < N && (N & 3) == 0)))
        old4 = *reinterpret_cast<const gs_f32x4*>(a.resid + (int64_t)(wave * 16 + fr) * N + ncol);
    }
    if constexpr (EPI == GS_EPI_QKV) {
      if (wave < MF) kvl = a.kv_len[min(wave * 16 + fr, M - 1)];
    }
  };

  // merge of the row's 16-column groups (mean_g, M2_g = sum (x - mean_g)^2).  Every group has 16 elements, so the union is
  //   mean = average of the group means,   M2 = sum_g M2_g + 16 sum_g (mean_g - mean)^2
  // (Chan et al. for equal counts): two short passes over the lane's share of the slots, no dependent chain of running means
  // (the general pairwise update cost ~15 instructions per group).  Fixed order, and the cross-lane sums over fg = 0..3 are
  // commutative pairwise adds: all four lanes of a row agree bitwise.  ~150 dependent instructions for the one wave a SIMD
  // holds (0.4 us).
  float ln_mean = 0.f, ln_rstd = 1.f;
  auto ln_merge = [&]() {
    float msum = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j)
      if (j < ln_nq) msum += lst[j][0] + lst[j][2];
    msum = rows4_sum(msum);
    const float cnt = 16.f * (float)a.lnc.nslots;  // = K of the producer's rows
    const float mean = msum * __builtin_amdgcn_rcpf((float)a.lnc.nslots);  // (v_rcp / v_rsq: the IEEE sequences are 45 of this epilogue's ~150 dependent instructions)
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j)
      if (j < ln_nq) {
        const float d0 = lst[j][0] - mean, d1 = lst[j][2] - mean;
        m2 += fmaf(16.f, fmaf(d1, d1, d0 * d0), lst[j][1] + lst[j][3]);
      }
    m2 = rows4_sum(m2);
    ln_mean = mean;
    ln_rstd = __builtin_amdgcn_rsqf(fmaf(m2, __builtin_amdgcn_rcpf(cnt), LN_EPS));
  };
  // Measured and rejected (MI355X, 64 utterances, tools/ktrace_dist.py): requesting the statistics AHEAD of the W / X burst and
  // merging them while the burst is in flight shrinks the epilogue after the MFMAs 1.2 -> 0.5 us, but the MFMAs start 0.7 us
  // later (the statistics are the last thing the previous kernel wrote, and the slowest to arrive): C3 553.2 vs 553.8 k tokens/s.

  gs_f32x4 acc[NF][MF];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[f][i] = gs_f32x4{0.f, 0.f, 0.f, 0.f};

  const int chunks = Kw >> 6;
  if constexpr (FAST != 0) {
    // rounds of 4 chunks, then (FAST == 2) one round of 2: at d = 1536 a wave's share of K is 6 chunks.  One base per operand,
    // every offset a compile-time constant; the epilogue's operands are requested behind the first round's burst.
    auto round = [&](auto gn_c, const int c0) {
      constexpr int GN = decltype(gn_c)::value;
      gs_u32x4 wv[GN][NF][2], xv[GN][MF][2];
      constexpr int CB = W8 ? 1024 : 2048;  // bytes of one fragment-major chunk of W
      const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w) +
                                ((int64_t)blockIdx.x * NF * (K >> 6) + (kbeg >> 6) + c0) * CB + lane * 16;
      const int64_t wfs = (int64_t)(K >> 6) * CB;  // bytes between the workgroup's consecutive W fragments
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + (int64_t)((kbeg >> 6) + c0) * (2 * MF * 1024) + lane * 16;
#pragma unroll
      for (int g = 0; g < GN; ++g) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          wv[g][f][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wb + f * wfs + g * CB));
          if constexpr (!W8) wv[g][f][1] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wb + f * wfs + g * CB + 1024));
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          xv[g][i][0] = *reinterpret_cast<const gs_u32x4*>(xb + ((g * 2 + 0) * MF + i) * 1024);
          xv[g][i][1] = *reinterpret_cast<const gs_u32x4*>(xb + ((g * 2 + 1) * MF + i) * 1024);
        }
      }
      if (c0 == 0) request_epilogue_operands();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < GN; ++g) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            gs_bf16x8 wa;
            if constexpr (W8) wa = gs_fp8x8_to_bf16(wv[g][f][0][2 * s], wv[g][f][0][2 * s + 1]);
            else wa = __builtin_bit_cast(gs_bf16x8, wv[g][f][s]);
#pragma unroll
            for (int i = 0; i < MF; ++i)
              acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(gs_bf16x8, xv[g][i][s]), acc[f][i], 0, 0, 0);
          }
        }
      }
    };
    int c0 = 0;
    for (; c0 + 4 <= chunks; c0 += 4) round(std::integral_constant<int, 4>{}, c0);
    if constexpr (FAST == 2) round(std::integral_constant<int, 2>{}, c0);
  } else {
  for (int c0 = 0; c0 < chunks; c0 += G) {
    gs_u32x4 wv[G][2], xv[G][MF][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = min(c0 + g, chunks - 1);  // clamped: a short last round re-reads its final chunk, unused below
      if (a.dbg & 2) {  // timing diagnostic: no W traffic
        wv[g][0] = gs_u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        wv[g][1] = wv[g][0];
      } else if (a.w_packed) {  // fragment-major copy (misc.hip pack_w_frag_kernel): one contiguous 1 KB per fragment load
        const int64_t cc = (int64_t)blockIdx.x * (K >> 6) + (kbeg >> 6) + c;
        if constexpr (W8) {
          wv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(reinterpret_cast<const unsigned char*>(a.w) + cc * 1024 + lane * 16));
        } else {
          const bf16_t* wf = reinterpret_cast<const bf16_t*>(a.w) + cc * 1024 + lane * 8;
          wv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wf));
          wv[g][1] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wf + 512));
        }
      } else if constexpr (W8) {
        wv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wp8 + c * 64));
      } else {
        wv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wp + c * 64));
        wv[g][1] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wp + c * 64 + 32));
      }
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        if (a.dbg & 1) {  // timing diagnostic (option "gs_dbg"): no X traffic -- results are meaningless
          xv[g][i][0] = gs_u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          xv[g][i][1] = xv[g][i][0];
          continue;
        }
        xv[g][i][0] = *reinterpret_cast<const gs_u32x4*>(xfrag ? xfb + ((c * 2 + 0) * MF + fi(i)) * 512 : xp[i] + c * 64);
        xv[g][i][1] = *reinterpret_cast<const gs_u32x4*>(xfrag ? xfb + ((c * 2 + 1) * MF + fi(i)) * 512 : xp[i] + c * 64 + SSTEP);
      }
    }
    if (c0 == 0) request_epilogue_operands();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (c0 + g < chunks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          gs_bf16x8 wa;
          if constexpr (W8) wa = gs_fp8x8_to_bf16(wv[g][0][2 * s], wv[g][0][2 * s + 1]);
          else wa = __builtin_bit_cast(gs_bf16x8, wv[g][s]);
#pragma unroll
          for (int i = 0; i < MF; ++i)
            acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(gs_bf16x8, xv[g][i][s]), acc[0][i], 0, 0, 0);
        }
      }
    }
  }
  }

  // ---- combine the four K quarters through LDS; wave i finishes tile i = (W fragment i / MF, row fragment i % MF) ------------
  constexpr int NT = NF * MF;  // 16 x 16 output tiles of this workgroup
  const unsigned long long ktm1 = ktrace_mark(a.kt);  // MFMAs issued (the loads they wait for have landed)
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < MF; ++i) *reinterpret_cast<gs_f32x4*>(&red[wq][f * MF + fi(i)][lane][0]) = acc[f][i];  // [K quarter][tile]
  __syncthreads();
  const unsigned long long ktm2 = ktrace_mark(a.kt);  // past the block barrier
  const int i = wave;  // waves >= NT only take part in the barriers below
  const int ri = NF > 1 ? wave % MF : wave;  // the tile's row fragment
  gs_f32x4 v = gs_f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave < NT) {
    v = *reinterpret_cast<const gs_f32x4*>(&red[0][i][lane][0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const gs_f32x4*>(&red[w][i][lane][0]);
  }
  if (KS > 1) {  // uniform per launch
    // Cross-workgroup hand-off, two forms (option / tune knob "gs_formal"):
    //  0 (default) -- the write-through form, /opt/skills/guides/cdna_hip_programming.md Guideline 16 recipe R1: partials go out
    //     as write-through (sc1) stores, EVERY storing wave drains them (asm vmcnt(0): an acknowledged sc1 store is at the memory
    //     side, visible to every XCD), block barrier, one lane takes a relaxed agent-scope ticket; the last arriver reads the
    //     partials with sc1 loads (they bypass its CU's L1 and were never in its L2: the producers' sc1 stores drop the line).
    //     No cache-wide operation anywhere.
    //  1 -- the same data path with the C++ memory model spelled out: release fence (buffer_wbl2 sc1 -- writes the XCD's whole
    //     L2 back) before the ticket, acquire fence (buffer_inv sc1) after it.  Measured on MI355X: the fences
    //     add whole-cache work to a ~10 us kernel; results are bit-identical, which is why form 0 is the default.
    //  granules ("gs_gran", engine launches that pass an epoch word; KS <= 4) -- the guide's recipe R2, "the data is the flag":
    //     slices 0 .. KS-2 publish their tile as 8-byte {tag, value} granules (write-through) and are DONE -- no drain, no ticket;
    //     slice KS-1 (dispatched last: blockIdx.y is the slow index, so its producers are already resident or finished) polls
    //     the KS-1 tiles with L1-bypassing loads until every granule carries this launch's tag, adds them in slice order with
    //     its own tile last -- the order of the ticket form, so the sums are bit-identical -- and runs the epilogue.  Removes
    //     the store drain and the ticket's round trip from the critical path.  The tag = (AR iteration, layer) + 1 differs
    //     from that of the launch that used this workspace before (>= 2 layers: the layer index alone alternates).
    if (a.ws_gran != nullptr) {
      const unsigned tag = (unsigned)gran_ep * 64u + (unsigned)a.gran_idx + 1u;
      unsigned long long* gt = a.ws_gran + (((int64_t)blockIdx.x * KS) * NT + i) * 256 + lane * 4;  // tile (slice 0, i): 4 granules per lane
      if (ks != KS - 1) {
        if (wave < NT) gs_store_gran(gt + (int64_t)ks * NT * 256, v, tag);
        return;
      }
      if (wave < NT) {
        gs_u32x4 ga[3], gb[3];
        bool ok = false;
        for (unsigned spins = 0; spins < 400000u && !ok; ++spins) {
#pragma unroll
          for (int q = 0; q < 3; ++q) gs_load_gran(gt + (int64_t)min(q, KS - 2) * NT * 256, ga[q], gb[q]);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          bool mine = true;
#pragma unroll
          for (int q = 0; q < 3; ++q) mine &= ga[q].x == tag && ga[q].z == tag && gb[q].x == tag && gb[q].z == tag;
          ok = __all(mine);
        }
        if (!ok && lane == 0 && a.gran_fail != nullptr) atomicAdd(a.gran_fail, 1u);
        const gs_f32x4 own = v;
        v = gs_f32x4{__uint_as_float(ga[0].y), __uint_as_float(ga[0].w), __uint_as_float(gb[0].y), __uint_as_float(gb[0].w)};
#pragma unroll
        for (int q = 1; q < 3; ++q)
          if (q < KS - 1) v += gs_f32x4{__uint_as_float(ga[q].y), __uint_as_float(ga[q].w), __uint_as_float(gb[q].y), __uint_as_float(gb[q].w)};
        v += own;
      }
    } else {
    __shared__ int s_last;
    float* part = a.ws_part + ((int64_t)blockIdx.x * KS * NT) * 256;  // tile (ks, i): [lane][4] fp32, one 16-byte vector per lane
    if (wave < NT) gs_store16_wt(part + (ks * NT + i) * 256 + lane * 4, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (a.formal) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop its own wait after buffer_wbl2 (guide, compiler hazard)
      }
      const int t = __hip_atomic_fetch_add(a.ws_cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.formal) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_last = t == KS - 1;
      if (t == KS - 1) __hip_atomic_store(a.ws_cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-reset
    }
    __syncthreads();
    if (!s_last) return;
    if (wave < NT) {
      v = gs_f32x4{0.f, 0.f, 0.f, 0.f};
      for (int q0 = 0; q0 < KS; q0 += 4) {  // fixed slice order: the sum does not depend on who arrived last; 4 loads in flight
        const float* pb = part + i * 256 + lane * 4;
        gs_f32x4 t0, t1, t2, t3;
        gs_load16x4_sc1(pb + min(q0, KS - 1) * NT * 256, pb + min(q0 + 1, KS - 1) * NT * 256, pb + min(q0 + 2, KS - 1) * NT * 256,
                        pb + min(q0 + 3, KS - 1) * NT * 256, t0, t1, t2, t3);
        v += t0;
        if (q0 + 1 < KS) v += t1;
        if (q0 + 2 < KS) v += t2;
        if (q0 + 3 < KS) v += t3;
      }
    }
    }
  }
  if (wave >= NT) return;
  if (ln_in) {
    ln_merge();
    const float mean = ln_mean, rstd = ln_rstd;
    if constexpr (W8) v = v * scale4;
    // explicit fma's: left to -ffp-contract the compile-time-layout body fused the last step and the general body did not
    v = __builtin_elementwise_fma(__builtin_elementwise_fma(gs_f32x4{-mean, -mean, -mean, -mean}, wg4, v), gs_f32x4{rstd, rstd, rstd, rstd}, bias4);  // bias4 = wb = W beta + bias
  } else {
    if constexpr (W8) v = v * scale4 + bias4;  // * 2^e is exact
    else v += bias4;
  }
  // lane (fg, fr) holds C[m = 16 i + fr][n = n0 + 4 fg + r]
  gs_epilogue<EPI>(a, v, ri * 16 + fr, ncol, gamma4, old4, kvl);
  if (lane == 0) ktrace_end(a.kt, kt0, ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 4 + wave, ktm1, ktm2);
}

// K slices across workgroups: enough to give every CU a workgroup (target), each wave keeping >= 64 of K
// Measured at M = 64 (MI355X, per launch incl. the ticket hand-off): K = 4096, N = 1024 (FFN2) 23.4 -> 12.8 us
// with 4 slices; at K = 1024 a slice is one 64-deep chunk per wave and the hand-off costs more than the idle
// CUs did (out-proj 9.8 -> 12.2 us, QKV 14.6 -> 20.4 us): split only long K.
int gemm_skinny_ksplit(int N, int K, int target_wgs) {
  if (K < 2048) return 1;
  const int nblk = (N + 15) / 16;
  int ks = 1;
  while (nblk * ks < target_wgs && ks < 16 && K % (4 * 64 * ks * 2) == 0) ks *= 2;
  return ks;
}

// [tickets][partial tiles, fp32][partial tiles as {tag, value} granules]
size_t gemm_skinny_workspace_bytes() { return GS_WS_CNT_BYTES + (size_t)GS_WS_MAX_TILES * 64 * 16 * (sizeof(float) + 8); }

int g_gs_fast = 1;  // "gs_fast": 0 = always the general body (A/B)
// "gs_gran": 1 = split-K hand-off through {tag, value} granules where the caller provides an epoch (engine FFN2).  Measured
// (MI355X, 64 utterances, d = 1024, FFN2 with 4 slices; tools/ktrace_dist.py): bit-identical sums, no time-outs, and SLOWER --
// FFN2 6.25 -> 7.2 us per launch, AR step 670 -> 679 us: the finisher's first poll (6 x 16 B per lane from the memory side,
// ~1 us) usually arrives before the other slices' write-through stores have landed and a second one follows, where the ticket's
// last arriver reads exactly once, after the fact.  Default 0 (the ticket, recipe R1).
int g_gs_gran = 0;

template <int MF, bool W8, int FAST, int NF = 1>
static int gs_launch_f(hipStream_t st, const GemmSkinnyArgs& a0, int KS) {
  const dim3 grid((a0.N + 15) / 16 / NF, KS), block(256);
  GemmSkinnyArgs a = a0;
  a.ks_grid = KS;
  if (NF > 1) a.ws_gran = nullptr;  // (the granule hand-off indexes single-fragment workgroups)
  switch (a.epi) {
    case GS_EPI_STORE: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_STORE, W8, FAST, NF>), grid, block, 0, st, a); break;
    case GS_EPI_RELU: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_RELU, W8, FAST, NF>), grid, block, 0, st, a); break;
    case GS_EPI_RESID: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_RESID, W8, FAST, NF>), grid, block, 0, st, a); break;
    case GS_EPI_F32: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_F32, W8, FAST, NF>), grid, block, 0, st, a); break;
    case GS_EPI_QKV: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_QKV, W8, FAST, NF>), grid, block, 0, st, a); break;
    default: return -1;
  }
  return 0;
}

int g_gs_nf = 1;  // "gs_nf": 0 = one W fragment per workgroup always (A/B); 1 = two where the grid would not fit the chip

template <int MF, bool W8>
static int gs_launch_w(hipStream_t st, const GemmSkinnyArgs& a, int KS) {
  // the compile-time layout (see the kernel): fragment-major W and X (X padded to MF * 16 rows), an even number of chunks per
  // wave, whole 16-column fragments, a bias, 64-slot-multiple LayerNorm statistics exactly on the consuming epilogues, no diagnostics
  const bool ln_epi = a.epi == GS_EPI_QKV || a.epi == GS_EPI_RELU || a.epi == GS_EPI_F32;
  const int chunks = a.K / (4 * KS) / 64;
  const bool fast = g_gs_fast && a.w_packed && a.x_xf != 0 && a.dbg == 0 && a.rot == 0 && a.N % 16 == 0 && (a.K / (4 * KS)) % 128 == 0 &&
                    a.bias != nullptr && (a.lnc.stats != nullptr) == ln_epi && (!ln_epi || a.lnc.nslots == (chunks % 4 == 0 ? 64 : 96)) &&
                    (a.epi != GS_EPI_RESID || a.resid != nullptr) && (a.epi != GS_EPI_QKV || a.kv_len != nullptr);
  if (!fast) return gs_launch_f<MF, W8, 0>(st, a, KS);
  if constexpr (MF <= 2) {
    // two W fragments per workgroup where one each would put more workgroups in flight than the chip has CUs (kernel header)
    const int nfrag = a.N / 16;
    if (g_gs_nf && nfrag % 2 == 0 && nfrag * KS > 256)
      return chunks % 4 == 0 ? gs_launch_f<MF, W8, 1, 2>(st, a, KS) : gs_launch_f<MF, W8, 2, 2>(st, a, KS);
  }
  return chunks % 4 == 0 ? gs_launch_f<MF, W8, 1>(st, a, KS) : gs_launch_f<MF, W8, 2>(st, a, KS);
}

template <int MF>
static int gs_launch(hipStream_t st, const GemmSkinnyArgs& a, int KS) {
  return a.wscale != nullptr ? gs_launch_w<MF, true>(st, a, KS) : gs_launch_w<MF, false>(st, a, KS);
}

bool gemm_skinny_supports(int M, int N, int K, int epi, int dh) {
  if (M < 1 || M > 64 || N < 1 || K < 256 || K % 256 != 0) return false;
  if (epi == GS_EPI_QKV && (N % 12 != 0 || dh % 4 != 0)) return false;
  return true;
}


// =====================================================================================================================
// M-split variant (round 3): N / 16 < #CUs (out-proj, FFN2, logits: N = d -> 64 row fragments at d = 1024).
// The kernel above fills the chip there by cutting K across workgroups, and pays for it: the partial tiles go through a
// write-through hand-off + ticket and the last arriver's combine -- 3.1 us of FFN2's 8.3 us, and out-proj (K too short to
// split) runs on 64 workgroups only (profiles/r03_ktrace_b64_timeline.csv).  Here a workgroup owns ONE 16-row fragment of W
// and ONE 16-utterance fragment of the batch, for ALL of K: grid = (N / 16, M / 16).  Its NW = 4 / 8 / 16 waves split K
// (one round of <= 4 chunks each) and combine through LDS -- no cross-workgroup exchange at all.  The M / 16 workgroups of a
// W fragment run on the same XCD when N / 16 is a multiple of 8 (block -> XCD = linear index % 8) and share the rows through
// its L2 (default cache policy instead of non-temporal loads); each reads only ITS 16 rows of X.  Texture-path bytes per CU:
// W 2 K / NW x NW + X 2 K x 16 = 64 K bytes (vs 160 KB at d = 1024 above), on 256 CUs instead of 64.
template <int NW, int EPI, bool W8, int FAST = 0>  // FAST = 1..4: the wave's chunks, ONE round, compile-time layout (gemm_skinny_kernel's FAST body)
__global__ __launch_bounds__(NW * 64) void gemm_skinny_ms_kernel(GemmSkinnyArgs a) {
  constexpr int G = FAST != 0 ? FAST : 4;
  __shared__ __attribute__((aligned(16))) float red[NW][64][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (FAST) {
    asm volatile("" ::"s"(a.x), "s"(a.w), "s"(a.bias), "s"(a.wscale), "s"(a.K), "s"(a.N), "s"(a.M), "s"(a.lnc.stats), "s"(a.lnc.wg),
                 "s"(a.resid), "s"(a.lnp.gamma), "s"(a.kv_len), "s"(a.kt.buf), "s"(a.ks_grid));
  }
  const unsigned long long kt0 = ktrace_begin(a.kt);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * 16, mi = blockIdx.y, m0 = mi * 16;
  const int K = a.K, N = a.N, M = a.M, MFt = FAST ? a.ks_grid : (int)gridDim.y;
  const int Kw = K / NW;  // multiple of 64 (launcher)
  const int kbeg = wave * Kw;
  const int nrow = min(n0 + fr, N - 1);
  constexpr int KOFS = W8 ? 16 : 8;
  constexpr int SSTEP = W8 ? 8 : 32;
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.w) + (int64_t)nrow * K + kbeg + fg * 8;
  const unsigned char* wp8 = reinterpret_cast<const unsigned char*>(a.w) + (int64_t)nrow * K + kbeg + fg * 16;
  const bf16_t* xp = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)min(m0 + fr, M - 1) * K + kbeg + fg * KOFS;
  const bool xfrag = a.x_xf != 0;
  const bf16_t* xfb = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)(kbeg / 64) * 2 * MFt * 512 + lane * 8;

  const int ncol = n0 + fg * 4;
  gs_f32x4 bias4 = gs_f32x4{0.f, 0.f, 0.f, 0.f}, scale4 = gs_f32x4{1.f, 1.f, 1.f, 1.f};
  gs_f32x4 wg4 = gs_f32x4{0.f, 0.f, 0.f, 0.f}, gamma4 = gs_f32x4{1.f, 1.f, 1.f, 1.f}, old4 = gs_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int LN_MAXQ = 16;
  gs_f32x4 lst[LN_MAXQ];
  const bool ln_in = a.lnc.stats != nullptr;
  const int ln_nq = ln_in ? a.lnc.nslots >> 3 : 0;
  int kvl = 0;
  auto request_epilogue_operands = [&]() {  // wave 0 finishes the tile
    if (wave != 0) return;
    if (a.bias != nullptr) {
      if (FAST || ncol + 3 < N) bias4 = *reinterpret_cast<const gs_f32x4*>(a.bias + ncol);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = ncol + r < N ? a.bias[ncol + r] : 0.f;
      }
    }
    if constexpr (W8) {
      if (FAST || ncol + 3 < N) scale4 = *reinterpret_cast<const gs_f32x4*>(a.wscale + ncol);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) scale4[r] = ncol + r < N ? a.wscale[ncol + r] : 1.f;
      }
    }
    if (ln_in) {
      if (ncol + 3 < N) wg4 = *reinterpret_cast<const gs_f32x4*>(a.lnc.wg + ncol);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) wg4[r] = ncol + r < N ? a.lnc.wg[ncol + r] : 0.f;
      }
      const float* sp = a.lnc.stats + ((int64_t)min(m0 + fr, M - 1) * a.lnc.nslots + fg * (a.lnc.nslots >> 2)) * 2;
#pragma unroll
      for (int j = 0; j < LN_MAXQ; ++j)
        if (j < ln_nq) lst[j] = *reinterpret_cast<const gs_f32x4*>(sp + j * 4);
    }
    if constexpr (EPI == GS_EPI_RESID) {
      if (a.lnp.gamma != nullptr && ncol + 3 < N) gamma4 = *reinterpret_cast<const gs_f32x4*>(a.lnp.gamma + ncol);
      if (m0 + fr < M && ncol + 3 < N && (N & 3) == 0) old4 = *reinterpret_cast<const gs_f32x4*>(a.resid + (int64_t)(m0 + fr) * N + ncol);
    }
    if constexpr (EPI == GS_EPI_QKV) kvl = a.kv_len[min(m0 + fr, M - 1)];
  };

  gs_f32x4 acc = gs_f32x4{0.f, 0.f, 0.f, 0.f};
  const int chunks = Kw >> 6;
  for (int c0 = 0; c0 < chunks; c0 += G) {
    gs_u32x4 wv[G][2], xv[G][2];
    if constexpr (FAST) {
      const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w) +
                                ((int64_t)blockIdx.x * (K >> 6) + (kbeg >> 6) + c0) * (W8 ? 1024 : 2048) + lane * 16;
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + ((int64_t)((kbeg >> 6) + c0) * 2 * MFt + mi) * 1024 + lane * 16;
      const int64_t xs = (int64_t)MFt * 1024;  // bytes between the two k-halves of a chunk (and half a chunk)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        wv[g][0] = *reinterpret_cast<const gs_u32x4*>(wb + g * (W8 ? 1024 : 2048));
        if constexpr (!W8) wv[g][1] = *reinterpret_cast<const gs_u32x4*>(wb + g * 2048 + 1024);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        xv[g][0] = *reinterpret_cast<const gs_u32x4*>(xb + (g * 2 + 0) * xs);
        xv[g][1] = *reinterpret_cast<const gs_u32x4*>(xb + (g * 2 + 1) * xs);
      }
    } else {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = min(c0 + g, chunks - 1);
      if (a.w_packed) {
        const int64_t cc = (int64_t)blockIdx.x * (K >> 6) + (kbeg >> 6) + c;
        if constexpr (W8) {
          wv[g][0] = *reinterpret_cast<const gs_u32x4*>(reinterpret_cast<const unsigned char*>(a.w) + cc * 1024 + lane * 16);
        } else {
          const bf16_t* wf = reinterpret_cast<const bf16_t*>(a.w) + cc * 1024 + lane * 8;
          if (a.ms_nt) {
            wv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wf));
            wv[g][1] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wf + 512));
          } else {
            wv[g][0] = *reinterpret_cast<const gs_u32x4*>(wf);
            wv[g][1] = *reinterpret_cast<const gs_u32x4*>(wf + 512);
          }
        }
      } else if constexpr (W8) {
        wv[g][0] = *reinterpret_cast<const gs_u32x4*>(wp8 + c * 64);
      } else {
        wv[g][0] = *reinterpret_cast<const gs_u32x4*>(wp + c * 64);
        wv[g][1] = *reinterpret_cast<const gs_u32x4*>(wp + c * 64 + 32);
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = min(c0 + g, chunks - 1);
      xv[g][0] = *reinterpret_cast<const gs_u32x4*>(xfrag ? xfb + ((c * 2 + 0) * MFt + mi) * 512 : xp + c * 64);
      xv[g][1] = *reinterpret_cast<const gs_u32x4*>(xfrag ? xfb + ((c * 2 + 1) * MFt + mi) * 512 : xp + c * 64 + SSTEP);
    }
    }
    if (c0 == 0) request_epilogue_operands();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (FAST || c0 + g < chunks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          gs_bf16x8 wa;
          if constexpr (W8) wa = gs_fp8x8_to_bf16(wv[g][0][2 * s], wv[g][0][2 * s + 1]);
          else wa = __builtin_bit_cast(gs_bf16x8, wv[g][s]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(gs_bf16x8, xv[g][s]), acc, 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long ktm1 = ktrace_mark(a.kt);
  *reinterpret_cast<gs_f32x4*>(&red[wave][lane][0]) = acc;
  __syncthreads();
  const unsigned long long ktm2 = ktrace_mark(a.kt);
  if (wave != 0) return;
  gs_f32x4 v = *reinterpret_cast<const gs_f32x4*>(&red[0][lane][0]);
#pragma unroll
  for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const gs_f32x4*>(&red[w][lane][0]);  // fixed order
  if (ln_in) {  // merge of the row's group statistics: see gemm_skinny_kernel
    float msum = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j)
      if (j < ln_nq) msum += lst[j][0] + lst[j][2];
    msum = rows4_sum(msum);
    const float cnt = 16.f * (float)a.lnc.nslots;
    const float mean = msum * __builtin_amdgcn_rcpf((float)a.lnc.nslots);  // (v_rcp / v_rsq: the IEEE sequences are 45 of this epilogue's ~150 dependent instructions)
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j)
      if (j < ln_nq) {
        const float d0 = lst[j][0] - mean, d1 = lst[j][2] - mean;
        m2 += fmaf(16.f, fmaf(d1, d1, d0 * d0), lst[j][1] + lst[j][3]);
      }
    m2 = rows4_sum(m2);
    const float rstd = __builtin_amdgcn_rsqf(fmaf(m2, __builtin_amdgcn_rcpf(cnt), LN_EPS));
    if constexpr (W8) v = v * scale4;
    // explicit fma's: left to -ffp-contract the compile-time-layout body fused the last step and the general body did not
    v = __builtin_elementwise_fma(__builtin_elementwise_fma(gs_f32x4{-mean, -mean, -mean, -mean}, wg4, v), gs_f32x4{rstd, rstd, rstd, rstd}, bias4);
  } else {
    if constexpr (W8) v = v * scale4 + bias4;
    else v += bias4;
  }
  gs_epilogue<EPI>(a, v, m0 + fr, ncol, gamma4, old4, kvl);
  if (lane == 0) ktrace_end(a.kt, kt0, (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, ktm1, ktm2);
}

// "gs_msplit": 0 = never (split-K across workgroups, A/B); 1 (default) = where N / 16 leaves CUs idle AND K <= 2048 (out-proj,
// logits); 2 = also for long K (FFN2); 3 = 2 with non-temporal W loads.  Measured at 64 utterances, d = 1024 (MI355X, ktrace):
// out-proj 5.08 -> 2.57 us per launch; FFN2 (K = 4096: 128 KB of W + 128 KB of X per workgroup) 8.3 -> 12.6-13.0 us -- each
// of the four workgroups of a W fragment ends up pulling its own copy of the rows (nt or not), 32 MB instead of 8 MB per launch,
// so long K keeps the split-K kernel; AR loop of C3 600 -> 577 ms with 1, 642 with 2, 648 with 3.
int g_gs_msplit = 1;

// "gs_ms_pad": dynamic LDS bytes requested (never touched) by the 1024-thread M-split workgroups: > 80 KB leaves room for ONE such
// workgroup per CU, so that the 256 of them cannot be packed two to a CU while other CUs stay empty
int g_gs_ms_pad = 0;

template <int NW, int EPI, bool W8>
static int gs_ms_launch_one(hipStream_t st, const GemmSkinnyArgs& a0, dim3 grid, dim3 block) {
  unsigned pad = (NW == 16 && g_gs_ms_pad > 0) ? (unsigned)g_gs_ms_pad : 0u;
  if (pad > 0) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_ms_kernel<NW, EPI, W8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              140 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        pad = 0;
      } else {
        attr_set = true;
      }
    }
  }
  GemmSkinnyArgs a = a0;
  a.ks_grid = (int)grid.y;
  // the compile-time layout (gemm_skinny_kernel's FAST body): fragment-major operands, whole rounds of 4 chunks per wave
  const int chunks = a.K / NW / 64;
  const bool fast = g_gs_fast && pad == 0 && a.w_packed && a.x_xf != 0 && a.ms_nt == 0 && a.dbg == 0 && a.N % 16 == 0 && (a.K / NW) % 64 == 0 &&
                    chunks >= 1 && chunks <= 4;
  if (fast && chunks == 4) hipLaunchKernelGGL((gemm_skinny_ms_kernel<NW, EPI, W8, 4>), grid, block, 0, st, a);
  else if (fast && chunks == 3) hipLaunchKernelGGL((gemm_skinny_ms_kernel<NW, EPI, W8, 3>), grid, block, 0, st, a);
  else if (fast && chunks == 2) hipLaunchKernelGGL((gemm_skinny_ms_kernel<NW, EPI, W8, 2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_skinny_ms_kernel<NW, EPI, W8>), grid, block, pad, st, a);
  return 0;
}

template <int NW, bool W8>
static int gs_ms_launch_w(hipStream_t st, const GemmSkinnyArgs& a) {
  const dim3 grid((a.N + 15) / 16, (a.M + 15) / 16), block(NW * 64);
  switch (a.epi) {
    case GS_EPI_STORE: return gs_ms_launch_one<NW, GS_EPI_STORE, W8>(st, a, grid, block);
    case GS_EPI_RELU: return gs_ms_launch_one<NW, GS_EPI_RELU, W8>(st, a, grid, block);
    case GS_EPI_RESID: return gs_ms_launch_one<NW, GS_EPI_RESID, W8>(st, a, grid, block);
    case GS_EPI_F32: return gs_ms_launch_one<NW, GS_EPI_F32, W8>(st, a, grid, block);
    case GS_EPI_QKV: return gs_ms_launch_one<NW, GS_EPI_QKV, W8>(st, a, grid, block);
    default: return -1;
  }
}

// M-split when it fills the chip better than N / 16 row fragments alone: returns the waves per workgroup (4 / 8 / 16) or 0
static int gs_msplit_waves(const GemmSkinnyArgs& a) {
  if (!g_gs_msplit || a.ksplit > 0 || a.M <= 16) return 0;
  if (g_gs_msplit == 1 && a.K > 2048) return 0;  // long K stays on the split-K kernel (see g_gs_msplit)
  const int nfrag = (a.N + 15) / 16, mf = (a.M + 15) / 16;
  if (nfrag >= 160 || nfrag * mf > 384) return 0;  // QKV / FFN1 already have a workgroup per CU
  for (int nw : {4, 8, 16})
    if (a.K % (64 * nw) == 0 && a.K / (64 * nw) <= 4) return nw;
  return a.K % (64 * 16) == 0 ? 16 : 0;
}

int g_gs_formal = 0;  // "gs_formal": split-K hand-off with explicit agent-scope release / acquire fences (see the kernel)

// returns 0 = launched, 1 = shape not covered
int launch_gemm_skinny(hipStream_t st, const GemmSkinnyArgs& a) {
  if (!gemm_skinny_supports(a.M, a.N, a.K, a.epi, a.dh)) return 1;
  if (a.lnc.stats != nullptr && (a.lnc.wg == nullptr || a.bias == nullptr || a.lnc.nslots * 16 != a.K || a.lnc.nslots % 8 != 0 || a.lnc.nslots > 128)) return -1;
  if (a.lnp.gamma != nullptr && (a.epi != GS_EPI_RESID || a.N % 16 != 0 || !a.lnp.xg_out || !a.lnp.stats_out || a.lnp.MF < 1)) return -1;
  int KS = 1;
  if (a.workspace != nullptr) {  // [GS_WS_CNT_BYTES of zeroed tickets][partial tiles]
    KS = a.ksplit > 0 ? a.ksplit : gemm_skinny_ksplit(a.N, a.K, a.target_wgs > 0 ? a.target_wgs : 256);
    const int nblk = (a.N + 15) / 16;
    while (KS > 1 && (a.K % (256 * KS) != 0 || nblk * KS > GS_WS_MAX_TILES || nblk > GS_WS_CNT_BYTES / 4)) KS >>= 1;
  }
  if (const int nw = gs_msplit_waves(a)) {
    const bool w8 = a.wscale != nullptr;
    GemmSkinnyArgs a2 = a;
    a2.ms_nt = g_gs_msplit == 3;
    const GemmSkinnyArgs& a = a2;
    switch (nw) {
      case 4: return w8 ? gs_ms_launch_w<4, true>(st, a) : gs_ms_launch_w<4, false>(st, a);
      case 8: return w8 ? gs_ms_launch_w<8, true>(st, a) : gs_ms_launch_w<8, false>(st, a);
      default: return w8 ? gs_ms_launch_w<16, true>(st, a) : gs_ms_launch_w<16, false>(st, a);
    }
  }
  GemmSkinnyArgs b = a;
  b.formal = g_gs_formal;
  b.ws_cnt = reinterpret_cast<int*>(a.workspace);
  b.ws_part = a.workspace ? reinterpret_cast<float*>(reinterpret_cast<char*>(a.workspace) + GS_WS_CNT_BYTES) : nullptr;
  b.ws_gran = (g_gs_gran && a.workspace && a.gran_epoch != nullptr && a.epi == GS_EPI_RESID && KS >= 2 && KS <= 4 && !g_gs_formal)
                  ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.workspace) + GS_WS_CNT_BYTES +
                                                          (size_t)GS_WS_MAX_TILES * 64 * 16 * sizeof(float))
                  : nullptr;
  if (a.M <= 16) return gs_launch<1>(st, b, KS);
  if (a.M <= 32) return gs_launch<2>(st, b, KS);
  if (a.M <= 48) return gs_launch<3>(st, b, KS);
  return gs_launch<4>(st, b, KS);
}

}  // namespace vle
