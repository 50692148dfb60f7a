// Host-side launchers of the gfx950 kernels (one .hip file per family).  All pointers are
// device pointers; `dtype` is VLE_DTYPE_F32 (0) or VLE_DTYPE_BF16 (1) and selects the element
// type T of every `void*` operand marked T.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace vle {

constexpr int DT_F32 = 0;
constexpr int DT_BF16 = 1;
constexpr int DT_FP8W = 2;  // AR-step linear weights e4m3fn + per-row 2^e scale; everything else as DT_BF16
inline size_t dtype_size(int dtype) { return dtype == DT_F32 ? 4 : 2; }

// ---- layernorm.hip ---------------------------------------------------------------------------
// out[T][r] = LN(x[f32][row_map ? row_map[r] : r]) * gamma + beta
int launch_layernorm(hipStream_t st, int dtype, const float* x, const int32_t* row_map, const float* gamma,
                     const float* beta, void* out, int64_t rows, int d);
// dst[f32][r] = src[f32][row_map[r]]
int launch_gather_rows(hipStream_t st, const float* src, const int32_t* row_map, float* dst, int rows, int d);
// LayerNorm of the AR-step rows (<= 64) written bf16 in the fragment-major layout of common.h (MF = ceil(rows / 16))
int launch_layernorm_xf(hipStream_t st, const float* x, const float* gamma, const float* beta, void* out, int rows, int d, int w8);
// block_ops.hip: stand-alone forms of ops the engine runs fused (block API, SURVEY.md 8b seam B3)
int launch_token_embedding(hipStream_t st, const int64_t* ids, const float* table, float* out, int64_t n, int d);
int launch_token_embedding_add(hipStream_t st, const int64_t* ids, const float* table, float* out, int64_t n, int d);  // out += table[ids]
int launch_sine_positional(hipStream_t st, const float* x, const float* pe, const float* alpha, float x_scale, float* out,
                           int64_t B, int T, int d);
int launch_adaln_fold(hipStream_t st, const float* wb, const float* g, const float* be, float* gamma_out, float* beta_out, int d);
int launch_cross_entropy(hipStream_t st, const float* logits, const int64_t* targets, float* loss, int32_t* hit, int64_t rows, int V,
                         int ignore_index, int topk);

// ---- gemm.hip -------------------------------------------------------------------------------
enum { EPI_STORE = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_F32 = 3,
       // LayerNorm folded into the packed-row GEMMs (GemmLn below; bf16, gemm_glds.hip / gemm_8ph.hip only)
       EPI_RESID_LNP = 4,   // EPI_RESID + producer side: bf16(x_new * gamma) and the statistics of every 64-column group of x_new
       EPI_STORE_LNC = 5,   // consumer side: A holds x * gamma; out = bf16(rstd * (acc - mean * sg[n]) + bias[n])
       EPI_RELU_LNC = 6 };  // ... with ReLU
// LayerNorm of the residual stream without a LayerNorm launch (valle/modules/transformer.py:57-74, 93-108 with the AdaLN fold):
//   LN(x) W^T + b  =  rstd * ((x * gamma) W^T - mean * sg) + tb,   sg[n] = sum_k W[n][k] gamma[k],  tb[n] = sum_k W[n][k] beta[k] + b[n].
// The GEMM that completes the residual stream x (out-proj, linear2: EPI_RESID_LNP) also writes xg = bf16(x * gamma) of the NEXT
// norm site and, per row and 64-column group, (mean, M2) of the group (exact two-pass over the 64 fp32 values); the GEMM that
// reads the normalised row (in-proj, linear1: EPI_*_LNC) multiplies xg, combines the row's K / 64 group pairs (Chan) and applies
// the affine in its epilogue -- `bias` must then be tb; sg and tb reach the epilogue through LDS (an LDS-DMA issued first thing in
// the prologue: no registers through the main loop, no exposed load in the epilogue).  The layout of the statistics does not depend
// on tile sizes: any mix of launches (tile policy, leftover-row launches) may produce and consume it.
struct GemmLn {
  const float* gamma = nullptr;   // producer: [N] of the next norm site
  void* xg = nullptr;             // producer: bf16 [M][N]
  float* stats_out = nullptr;     // producer: [N / 64][stats_ld][2], GROUP-major: a consumer wave's lanes own consecutive rows, so its
                                  //           request for one group is one contiguous 512 bytes (row-major cost it 64 lines per instruction:
                                  //           1.3 us of address traffic in front of the first tile of every workgroup, measured)
  const float* stats_in = nullptr;  // consumer: [K / 64][stats_ld][2]
  int64_t stats_ld = 0;           // rows between two groups of the statistics (the buffer's row capacity)
  const float* sg = nullptr;      // consumer: [N]
};
constexpr int LN_GROUP = 64;
// out = epi(A[T, M x K] @ W[T, N x K]^T + bias)
int launch_gemm(hipStream_t st, int dtype, const void* A, const void* W, const float* bias, void* out, float* resid,
                int64_t M, int N, int K, int epi, const GemmLn* ln = nullptr);
// the LNP / LNC epilogues exist for a layer of width d over M packed rows (bf16, M >= 128, every tile inside N whatever the tile
// policy picks for N = d, 3 d, 4 d; the consumer combines up to 24 group pairs: d <= 1536)
bool gemm_ln_supports(int dtype, int64_t M, int d);

int launch_gemm_f32_strided(hipStream_t st, const float* A, int64_t lda, const float* W, const float* bias, float* out, float* resid,
                            int64_t M, int N, int K, int epi);

// gemm_glds.hip: bf16, M >= 128, K % 64 == 0: LDS-DMA multi-stage pipeline; returns 1 when the shape is not covered
int launch_gemm_glds(hipStream_t st, const void* A, const void* W, const float* bias, void* out, float* resid, int64_t M, int N,
                     int K, int epi, const GemmLn* ln = nullptr);

// gemm_glds.hip on fp32 operands (token-exact mode, M >= 128): 0 launched, 1 shape not covered / switched off ("f32_glds")
int launch_gemm_glds_f32(hipStream_t st, const float* A, const float* W, const float* bias, void* out, float* resid, int64_t M, int N, int K, int epi);
extern int g_f32_glds;
extern int g_attn_f32_vec;  // attention.hip: 16-byte register-double-buffered K / V staging of the fp32 attention (0: element-wise, A/B)
extern int g_attn_qw;
extern int g_g8_colgroup;
extern int g_g8_stagger;
extern int g_g8_persist;  // gemm_8ph.hip: 1 = persistent tile loop (one workgroup per CU walks several tiles), 0 = one tile per workgroup
extern int g_glds_8ph;
extern int g_glds_t64;
extern int g_glds_tail;  // gemm_glds.hip: 1 = the rows past the last full 256-row tile of a gemm_8ph launch as a second small launch when that saves a round
extern int g_glds_swz;
// gemm_8ph.hip: bf16, 256 x 256 tile, phase-split schedule; returns 1 when the shape is not covered
int launch_gemm_8ph(hipStream_t st, const void* A, const void* W, const float* bias, void* out, float* resid, int64_t M, int N,
                    int K, int epi, const GemmLn* ln = nullptr);
// gemm_fp8.hip: e4m3fn x e4m3fn on v_mfma_scale_f32_16x16x128_f8f6f4, per-row power-of-two scales on both operands
// (engine mode FP8); returns 1 when the shape is not covered
int launch_gemm_fp8(hipStream_t st, const void* A8, const float* a_scale, const void* W8, const float* w_scale, const float* bias,
                    void* out, float* resid, int64_t M, int N, int K, int epi);
// misc.hip: bf16 rows -> e4m3fn codes + one power-of-two scale per row; returns 1 when K is not instantiated
int launch_quantize_rows_fp8(hipStream_t st, const void* x_bf16, void* q, float* scale, int64_t rows, int K);
extern int g_glds_prio;
extern int g_glds_w8;
extern int g_glds_big;  // gemm_glds.hip tile policy: 0 never the 8-wave 256 x 128 tile, -1 default threshold, n > 0 threshold

// ---- gemm_skinny.hip (AR-step weight-streaming MFMA GEMM, bf16, 2 <= M = batch <= 64) ------------
enum { GS_EPI_STORE = 0, GS_EPI_RELU = 1, GS_EPI_RESID = 2, GS_EPI_F32 = 3, GS_EPI_QKV = 4 };
// LayerNorm fused across the GEMMs of the batched AR step (no LayerNorm launch):
//   y[n] = rstd * (sum_k W[n][k] * (gamma[k] x[k]) - mean * wg[n]) + wb[n],   wg = W gamma,  wb = W beta + bias
// The PRODUCER of a residual row (RESID epilogue, sampling kernel) also writes bf16(x * gamma_next) in the fragment-major X
// layout and, per 16-column group, the group's (mean, M2 = sum (x - mean)^2); the CONSUMER merges the d/16 groups of a row in a
// fixed order (Chan's parallel-variance update: exact two-pass quality, deterministic) and applies the algebra in its epilogue.
struct LnProducer {
  const float* gamma = nullptr;  // [d] of the LayerNorm that reads this residual next; null = plain epilogue
  void* xg_out = nullptr;        // bf16, fragment-major [MF * 16][d]
  float* stats_out = nullptr;    // [rows][d / 16][2]
  int w8 = 0;                    // xf_index variant the consuming GEMM uses (FP8W weights)
  int MF = 0;                    // row fragments of the step's batch
};
struct LnConsumer {
  const float* stats = nullptr;  // [rows][nslots][2]; null = X already normalised
  const float* wg = nullptr;     // [N]; `bias` then holds wb
  int nslots = 0;                // K / 16
};
struct GemmSkinnyArgs {
  LnProducer lnp;
  LnConsumer lnc;
  KTrace kt;  // diagnostic timeline (option "ktrace")
  int dbg = 0;  // timing diagnostics (option "gs_dbg"): 1 = no X loads, 2 = no W loads (results meaningless)
  int ms_nt = 0;   // M-split kernel: non-temporal W loads (A/B; filled by the launcher)
  int formal = 0;  // split-K hand-off with explicit release / acquire fences (filled by the launcher from g_gs_formal)
  int rot = 0;  // rotate the order in which a workgroup walks X by its index (option "gs_rot")
  int ks_grid = 1;  // = gridDim.y, filled by the launcher (the FAST body reads it with the other arguments instead of the implicit ones)
  const void* x = nullptr;     // bf16 [M][K]
  const void* w = nullptr;     // bf16 [N][K]
  const float* bias = nullptr; // f32 [N] or null
  const float* wscale = nullptr;  // FP8W: w is e4m3fn [N][K], row n scaled by wscale[n] (a power of two); null = bf16 w
  int M = 0, N = 0, K = 0, epi = GS_EPI_STORE;
  void* out = nullptr;         // bf16 [M][N] (STORE / RELU) or f32 [M][N] (F32)
  float* resid = nullptr;      // f32 [M][N] += (.)   (RESID)
  float* q_out = nullptr;      // f32 [M][d]          (QKV)
  void* k_cache = nullptr;     // bf16 [M][H][ctx_max][dh] (this layer)
  void* v_cache = nullptr;
  const int32_t* kv_len = nullptr;
  int ctx_max = 0, nhead = 1, dh = 4;
  // split-K across workgroups (N/16 < #CUs): caller-owned device workspace of gemm_skinny_workspace_bytes(),
  // its first GS_WS_CNT_BYTES zeroed once (the tickets reset themselves); null = no split
  void* workspace = nullptr;
  int ksplit = 0;      // 0 = chosen from N, K and target_wgs
  int target_wgs = 0;  // 0 = 256
  int w_packed = 0;        // w is the fragment-major copy made by launch_pack_w_frag (engine only)
  int x_xf = 0;            // X is stored fragment-major (common.h xf_index) instead of row-major
  int out_xf = 0;          // STORE / RELU epilogues write `out` fragment-major for the next GEMM: 0 no, 1 bf16-W consumer, 2 fp8-W
  int* ws_cnt = nullptr;   // (filled by the launcher)
  float* ws_part = nullptr;
  // split-K hand-off through self-validating granules instead of the ticket (engine only; gemm_skinny.hip "gs_gran"): a device
  // word that differs between consecutive launches on this workspace when combined with gran_idx (the AR iteration counter +
  // the layer index, >= 2 layers), and a counter of spin time-outs (stays 0)
  const int32_t* gran_epoch = nullptr;
  int gran_idx = 0;
  unsigned* gran_fail = nullptr;
  unsigned long long* ws_gran = nullptr;  // (filled by the launcher)
};
constexpr int GS_WS_CNT_BYTES = 4096;  // 1024 row-fragment tickets
constexpr int GS_WS_MAX_TILES = 2048;  // partial 16 x 64 fp32 tiles (4 KB each)
extern int g_da_nt;       // decode_attn.hip: non-temporal K / V loads (-1 auto, 0, 1)
extern int g_da_lds_pad;  // decode_attn.hip: dynamic LDS bytes per workgroup of the batched decode attention (occupancy cap)
extern int g_gs_formal;
extern int g_gs_gran;  // gemm_skinny.hip: split-K hand-off through granules where the caller provides an epoch (default 0: measured slower)
extern int g_gs_nf;    // gemm_skinny.hip: two W fragments per workgroup where the one-fragment grid exceeds the chip (default 1)
extern int g_gs_fast;  // gemm_skinny.hip: compile-time-layout body of the split-K skinny GEMM where the launch qualifies (default 1)
extern int g_gs_ms_pad;
extern int g_gs_msplit;  // gemm_skinny.hip: M-split kernel for N / 16 < #CUs (default 1)
size_t gemm_skinny_workspace_bytes();
int gemm_skinny_ksplit(int N, int K, int target_wgs);
bool gemm_skinny_supports(int M, int N, int K, int epi, int dh);
int launch_gemm_skinny(hipStream_t st, const GemmSkinnyArgs& a);  // 1 = shape not covered

// ---- skinny.hip (AR-step weight-streaming GEMV, M = batch <= 8) --------------------------------
enum { PRO_PLAIN = 0, PRO_LN = 1, PRO_ATTN = 2, PRO_ATTN_SELF = 3 };  // ATTN_SELF: the partials exclude the new token (qkv_attn1), merged here
enum { SEPI_STORE = 0, SEPI_RELU = 1, SEPI_RESID = 2, SEPI_QKV = 3 };
struct SkinnyArgs {
  const void* w = nullptr;     // T [N][K]
  const float* bias = nullptr; // f32 [N] or null
  const float* wscale = nullptr;  // dtype DT_FP8W (gemv1 only): w is e4m3fn [N][K], row n scaled by wscale[n]
  const void* w8 = nullptr;       // engine: the e4m3fn copy of w (FP8W), tried first by launch_ar_linear
  int temporal = 0;               // DT_FP8W gemv1: default-policy weight loads (memory-side cache) instead of non-temporal
  int N = 0, K = 0, B = 0;
  int pro = PRO_PLAIN, epi = SEPI_STORE;
  const float* x = nullptr;      // f32 [B][K]           (PRO_PLAIN / PRO_LN)
  const float* gamma = nullptr;  // f32 [K]              (PRO_LN)
  const float* beta = nullptr;
  const float* part_o = nullptr; // f32 [B][nsplit][d] (PRO_ATTN): un-normalised partial outputs
  const float* part_ml = nullptr;// f32 [B][H][nsplit][2]  (running max, running sum)
  int nsplit = 1, nhead = 1, dh = 1;
  const float* q_self = nullptr; // f32 [d] (PRO_ATTN_SELF, batch 1): the new token's query, key and value rows (cache-rounded) --
  const float* k_self = nullptr; //   its own term of the softmax is one more "partial" (m = q.k / sqrt(dh), l = 1, o = v)
  const float* v_self = nullptr;
  float* out = nullptr;          // f32 [B][N]           (SEPI_STORE / SEPI_RELU)
  float* resid = nullptr;        // f32 [B][N] += (.)    (SEPI_RESID)
  float* q_out = nullptr;        // f32 [B][d]           (SEPI_QKV)
  void* k_cache = nullptr;       // T [B][H][ctx_max][dh] (this layer)
  void* v_cache = nullptr;
  const int32_t* kv_len = nullptr; // [B] slot the new token's K/V go to
  int ctx_max = 0;
  int rpw_override = 0;          // tuning hook of the batch-1 path (rows per wave), 0 = heuristic
  int act_bf16 = 0;              // gemv1 only: 1 = the merged attention row (PRO_ATTN*), 2 = the ReLU output (SEPI_RELU) rounded to bf16
  KTrace kt;                     // diagnostic timeline (gemv1 only)
};
int launch_skinny(hipStream_t st, int dtype, const SkinnyArgs& a);
// gemv1.hip: batch-1 wave-autonomous variant; returns 1 when the shape is not instantiated (use launch_skinny)
int launch_gemv1(hipStream_t st, int dtype, const SkinnyArgs& a);
extern int g_qa_waves;   // 4 / 8 waves per workgroup of the fused QKV + attention launch
extern int g_g1_shared;  // 1 = block-shared activations (default), 0 = wave-autonomous (A/B)

// gemv1.hip: LN1 + QKV projection + KV-cache write + decode attention over the OLD keys of ONE utterance in ONE launch
// (batch-1 AR step: 5 -> 4 launches per layer).  Workgroups [0, H * nsplit) each recompute their head's dh query rows (the
// splits of a head sit on one XCD and share the rows through its L2) and stream their key chunks of the cache; the other
// workgroups are the K / V rows of the GEMV (cache write + k_new / v_new).  Nothing is exchanged inside the launch: the new
// token's own softmax term is merged by the out-proj GEMV's prologue (PRO_ATTN_SELF).  Returns 1 = shape not covered.
struct QkvAttnArgs {
  const void* w = nullptr;        // T [3d][d] in_proj_weight (valle/modules/activation.py:128-130)
  const float* bias = nullptr;    // f32 [3d]
  const float* wscale = nullptr;  // FP8W row scales
  const float* x = nullptr;       // f32 [d] residual stream of the new token
  const float* gamma = nullptr;   // norm1
  const float* beta = nullptr;
  float* q_out = nullptr;         // f32 [d]
  float* k_new = nullptr;         // f32 [d], rounded to the cache type
  float* v_new = nullptr;
  void* k_cache = nullptr;        // cache_t [H][ctx_max][dh] of this layer
  void* v_cache = nullptr;
  const int32_t* kv_len = nullptr;
  float* part_o = nullptr;        // [nsplit][d]
  float* part_ml = nullptr;       // [H][nsplit][2]
  int d = 0, nhead = 0, dh = 0, ctx_max = 0, nsplit = 8;
  int n_attn = 0;                 // (filled by the launcher) workgroups with the attention role
  int temporal = 0;               // FP8W: default-policy weight loads
  int nk = 4;                     // keys per lane per round of the attention workgroups (8: hand-off mode only)
  int q_temporal = 1;             // the query rows (read by the nsplit workgroups of a head) with the default cache policy
  // In-launch hand-off of q (round 3, option "qa_handoff"): the query rows are ordinary GEMV rows spread over all CUs (each written
  // once, as an 8-byte {tag = epoch, value} granule: the guide's recipe R2, "the data is the flag"); the attention workgroups only
  // stream their K / V chunks and poll their head's dh granules.  Bounded spin: on a timeout a workgroup recomputes its head's
  // query rows itself (the hand-off-free path), so a result never depends on the hand-off succeeding.
  unsigned long long* q_gran = nullptr;  // [d] granules of this layer (zeroed whenever *epoch_ptr restarts); null = no hand-off
  const int32_t* epoch_ptr = nullptr;    // device word that grows by one per AR step (ArState::iter): epoch = *epoch_ptr + 1
  unsigned* spin_fail = nullptr;         // optional: counts workgroups that fell back
  KTrace kt;
};
bool qkv_attn1_supports(int dtype, int d, int nhead, int dh);
bool gemv1_attn_self_supports(int dtype, int d, int dh, int nsplit);  // the out-proj GEMV that must follow the fused launch
int launch_qkv_attn1(hipStream_t st, int dtype, const QkvAttnArgs& a);

// ---- persist.hip: the batch-1 AR step as ONE persistent launch (256 workgroups, one per CU; option "persist") -------------------
struct PLayer {  // one decoder layer's operands (device table, one entry per layer)
  const void *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;  // bf16 [N][K]
  const float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
  const float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
  void *kc = nullptr, *vc = nullptr;  // this layer's KV cache [H][ctx_max][dh] (batch 1)
  // folded LayerNorm (persist_mode bit 5): per output row n of the in-projection / linear1, sg[n] = sum_k W[n][k] gamma[k] and
  // tb[n] = sum_k W[n][k] beta[k] + bias[n]  (launch_ps_fold)
  const float *sgqkv = nullptr, *tbqkv = nullptr, *sg1 = nullptr, *tb1 = nullptr;
  // FP8W (the weights above are e4m3fn codes then): the rows' power-of-two scales; null in bf16 mode
  const float *sqkv = nullptr, *so = nullptr, *s1 = nullptr, *s2 = nullptr;
};
constexpr int PS_PT_SLOTS = 512;
constexpr int PS_MODE_DEFAULT = 0x174;    // hidden vector as bf16 pairs, XCD-local group edges, folded LayerNorm, bf16 activation rows + v_dot2c (D2), 1 sleep unit between sweeps
constexpr int PS_NAPS_DEFAULT = 0x325756;  // re-timed for the D2 forms (round 5: 130.9 -> 129.6 us per step on the probe; round 4's 0x335854 was timed for the fp32-row forms)
struct PStepArgs {
  const PLayer* layers = nullptr;  // device [L + 1]: the decoder layers, then the predict layer as a pseudo-layer (wqkv = ar_predict_layer.weight
                                   // bf16 [V][d], g1 / be1 = the final LayerNorm, sgqkv / tbqkv = its folded row constants [V], bqkv = any valid [3 d])
  int L = 0, d = 0, nhead = 0, dh = 0, V = 0, ctx_max = 0;
  const float* x_in = nullptr;     // f32 [d]: the token's embedding + position (sampling kernel)
  float* logits = nullptr;         // f32 [V]
  const int32_t* kv_len = nullptr; // [1] cache slot of the new token
  const int32_t* iter = nullptr;   // [1] AR iteration counter: epoch = iter + 1
  const int32_t* done = nullptr;   // [1] the utterance has stopped: the launch is a no-op
  unsigned long long* gran = nullptr;  // [(L + 1) * pstep_gran_per_layer] {epoch, value} granules (zeroed whenever iter restarts)
  unsigned* fail = nullptr;        // waves that gave up waiting (expected 0; the engine reports an error otherwise)
  unsigned long long* ptrace = nullptr;  // [8][256][PS_PT_SLOTS] optional timeline
  int never = 0;                   // always 0 (keeps the LDS carve allocated)
  // "persist_mode": bit 2 (4) the FFN hidden vector, bit 3 (8) the attention output travel as bf16 pairs; bit 4 (16) the two
  // head-group edges also through XCD-local (default-policy) granules; bit 5 (32) folded LayerNorm: the dot products run on
  // x * gamma while the row statistics are still being combined, rstd * (dot - mean * sg) + tb afterwards -- one workgroup barrier
  // per LayerNorm instead of three (not bit-identical to the launch chain: fp32 re-association); bit 6 (64) D2: the operators' input
  // rows in LDS as bf16, dot products on v_dot2c_f32_bf16 (needs bit 5; keys per lane 2, request schedules 0 / 3); bits 8..11 s_sleep
  // units between two sweeps of an edge
  int mode = PS_MODE_DEFAULT;
  // "persist_naps": s_sleep(4) units (~0.1 us each) ahead of the FIRST sweep of an edge, 4 bits each: attention output, x, x',
  // hidden, q/k/v, partials.  A sweep that comes back without the data costs a fabric round trip (~1.1 us) before the next one
  // can see it; the values time the first sweep to land just after the producers' stores (tools/persist_probe.py sweeps)
  int naps = PS_NAPS_DEFAULT;
  // the sampling step inside the launch: nsteps > 0 AR iterations per launch, `smp` = DEVICE copy of the PStepSample block (read with
  // scalar loads where it is needed -- as kernel arguments its 30 words stayed live through the whole step and spilled)
  int nsteps = 0;
  const struct PStepSample* smp = nullptr;
  int nk = 2;                      // "persist_nk": keys per lane per round of the attention share (2: 1024 keys in one round; 4)
  int pf = 3;                      // "persist_pf": when the compute waves request an operator's operands (persist.hip): 0 = in one burst right
                                   // before the sweep that precedes the operator, 3 = linear1 / linear2 spread over three sweeps (default)
  // slot mode of the batched launch (vle_slots_step on engines of 2 .. PSB_MAX slots): the slots' iteration counters differ (each restarts
  // at its admission), so the granules' epoch comes from a counter of its own that only ever grows between two vle_slots_begin calls
  // (null: epoch = iteration + 1), and slot b draws from the RNG stream of the request it holds (null: request_seed(dyn.seed, b))
  int32_t* epoch_ctr = nullptr;
  const unsigned long long* slot_seed = nullptr;
  int B = 1;                       // utterances in the launch: 1 = pstep_kernel (persist.hip); 2 .. PSB_MAX = pstepb_kernel (persist_nb.hip: x_in
                                   // [B][d], logits [B][V], kv_len / iter / done [B], gran sized by pstepb_gran_count, the caches [B][H][ctx_max][dh])
};
bool pstep_supports(int dtype, int d, int nhead, int dh, int V);
// 1 = (weight type, persist_mode, keys per lane, request schedule, timeline) is an instantiated form of pstep_kernel and the occupancy
// calculator places one of ITS workgroups per CU (the persistent grid needs all 256 resident); 0 = no such form; -1 = does not fit
int pstep_form_ok(int dtype, int mode, int nk, int pf, bool traced);
size_t pstep_gran_count(int d, int nhead, int L);
int launch_pstep(hipStream_t st, int dtype, const PStepArgs& a);  // 0 launched, 1 shape not covered, < 0 error
// persist_nb.hip: the same launch for 2 .. PSB_MAX utterances (bf16 weights, the default form only: PS_MODE_DEFAULT, 2 keys per lane,
// request schedule 3, sampling inside the launch); per utterance bit-identical to pstep_kernel's default form
constexpr int PSB_MAX = 6;
bool pstepb_supports(int dtype, int d, int nhead, int dh, int V, int B);
int pstepb_form_ok(int B, bool traced);  // 1 = one workgroup of the B-utterance form fits a CU; 0 = no such form; -1 = does not fit
size_t pstepb_gran_count(int d, int nhead, int L, int B);
int launch_pstepb(hipStream_t st, int dtype, const PStepArgs& a);  // 0 launched, 1 shape not covered, < 0 error
// sg[n] = sum_k W[n][k] gamma[k], tb[n] = sum_k W[n][k] beta[k] + (bias ? bias[n] : 0) for the N rows of bf16 W[N][K] (fp64 sums)
int launch_ps_fold(hipStream_t st, const void* W, const float* gamma, const float* beta, const float* bias, float* sg, float* tb, int N, int K);

// ---- attention.hip --------------------------------------------------------------------------
// prefill (causal=1: prefix-LM mask) / NAR (causal=0) attention over packed sequences
int launch_attention(hipStream_t st, int dtype, const void* qkv, void* out, const int32_t* seq_off,
                     const int32_t* text_len, int B, int max_len, int d, int nhead, int causal);
// attn_mfma.hip: bf16 MFMA flash kernel, dh in {32,64,96,128}; returns 1 when the head size is not covered
int launch_attention_mfma(hipStream_t st, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len, int B,
                          int max_len, int d, int nhead, int causal);
// attn_mfma2.hip: second-generation flash kernel (V^T pre-pass, double-buffered LDS, 128-query blocks); `rows` = an upper bound on
// the packed rows of qkv (sizes the V^T scratch); returns 1 when not covered, < 0 on error
int launch_attention_mfma2(hipStream_t st, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len, int B,
                           int max_len, int64_t rows, int d, int nhead, int causal);
int attn2_reserve(int64_t rows, int B, int d, void** scratch_out = nullptr);  // pre-pass modes only (attn_mode <= 2)
extern int g_g8_dbg;
extern int g_g8_nt;
extern int g_glds_epi;
extern int g_attn_v2, g_attn_xcd, g_attn_q128, g_attn_mode, g_attn_defer, g_attn_ring, g_attn_lsum;
// copy K,V of packed prefill rows into the cache: cache[b][h][pos][e]
int launch_kv_scatter(hipStream_t st, int dtype, const void* qkv, void* k_cache, void* v_cache, const int32_t* row_seq,
                      const int32_t* row_pos, int64_t rows, int d, int nhead, int ctx_max);
// decode_attn.hip: one new query per utterance against the KV cache; writes split partials
// part_o [B][nsplit][d], part_ml [B][H][nsplit][2]
// `done` (int32 [B] or null): utterances whose flag is set are skipped (no KV traffic, output row left stale)
int launch_decode_attention(hipStream_t st, int dtype, const float* q, const void* k_cache, const void* v_cache,
                            const int32_t* kv_len, float* part_o, float* part_ml, int B, int nhead, int dh, int ctx_max,
                            int nsplit, int nk_override = 0, void* out_norm = nullptr, const int32_t* done = nullptr,
                            int out_xf = 0, KTrace kt = KTrace(), int kv_nt = -1);  // kv_nt: non-temporal K / V loads (1 / 0; -1 = by this layer's size)
// the same + the layer's out-proj, residual and LayerNorm producer in one launch (decode_attn.hip AttnOproj); 1 = shape not covered
int launch_decode_attention_oproj(hipStream_t st, const float* q, const void* k_cache, const void* v_cache, const int32_t* kv_len, int B,
                                  int nhead, int dh, int ctx_max, const int32_t* done, const void* wo_bf16, const float* bias, float* resid,
                                  float* part, int* cnt, const LnProducer& lnp, KTrace kt);  // out_xf: out_norm fragment-major (0 no, 1 bf16-W consumer, 2 fp8-W consumer)
// nk_override: keys per lane per round (0 = auto, 4, 8); out_norm (T [B][d], nsplit == 1 only): write the
// normalised attention output directly instead of partials

// ---- embed.hip ------------------------------------------------------------------------------
struct PrefillEmbedArgs {
  const int64_t* text; int64_t s_stride;          // [B][s_stride]
  const int64_t* prompt; int64_t p_stride; int Q; // [B][p_stride][Q]
  const int32_t* text_len; const int32_t* row_seq; const int32_t* row_pos;
  const float* text_emb; const float* audio_emb; const float* pe;
  const float* alpha_text; const float* alpha_audio; // device scalars
  int bos; int d; int64_t rows; float* x;
};
int launch_prefill_embed(hipStream_t st, const PrefillEmbedArgs& a);

struct NarSeqTables {                 // all device int32 [B] unless noted
  const int32_t* text_len;   // S'_b  (after prefix_mode 2/4 slicing)
  const int32_t* text_drop;  // source index shift for text positions >= 1
  const int32_t* prompt_len; // P_b
  const int32_t* gen_len;    // G_b
  const int32_t* aoff;       // first row of utterance b in the packed audio rows   [B+1]
  const int32_t* xoff;       // first row of utterance b in the packed NAR rows     [B+1]
};
struct NarEmbedArgs {
  NarSeqTables t;
  const int64_t* text; int64_t s_stride;
  const int64_t* prompt; int64_t p_stride; int Q;   // prompt codes [B][p_stride][Q]
  const int64_t* first_cb; int64_t g_stride;        // generated first codebook [B][g_stride]
  const int32_t* arow_seq; const int32_t* arow_pos; // packed audio rows -> (b, a)
  const int32_t* xrow_seq; const int32_t* xrow_pos; // packed NAR rows   -> (b, pos)
  const float* const* audio_embs;                   // device array of Q table pointers
  const float* text_emb; const float* pe; const float* alpha_text; const float* alpha_audio;
  int d; int64_t arows; int64_t xrows;
  float* y_emb;  // [arows][d]
  float* x;      // [xrows][d]
};
int launch_nar_yemb_init(hipStream_t st, const NarEmbedArgs& a, int sum_all_prompt_codebooks);
int launch_nar_yemb_add_prompt(hipStream_t st, const NarEmbedArgs& a, int j);
int launch_nar_assemble(hipStream_t st, const NarEmbedArgs& a);

// ---- sampling.hip ---------------------------------------------------------------------------
struct ArState {           // device pointers
  int32_t* kv_len;     // [B] KV slot of the token being fed this step
  int32_t* audio_pos;  // [B] position (audio stream, incl. BOS) of that token
  int32_t* n_gen;      // [B] frames appended so far
  int32_t* done;       // [B]
  int32_t* cap;        // [B] 16 * S_b   (valle.py:1047)
  int32_t* iter;       // [B] AR loop iterations executed
  int32_t* done_count; // [1]
};
// Per-call sampling parameters live in DEVICE memory (one struct per engine) so that a captured
// hipGraph of the AR step does not bake them in.
struct ArDyn {
  int32_t top_k; float temperature; uint64_t seed;
  int32_t max_new; int32_t has_forced;
  int32_t ignore_eos; int32_t pad0;        // benchmark hook: only the length cap stops an utterance
  const int64_t* forced; int64_t forced_stride; const int32_t* forced_len;
  float* trace; int64_t trace_cap;         // [trace_cap][B][V] or null
};
struct ArSampleArgs {
  ArState s;
  const ArDyn* dyn;
  const float* logits;  // [B][V]
  int V; int B; int d;
  int bos; int first;
  int64_t* tokens; int64_t g_stride;       // [B][g_stride] token history actually fed
  int64_t* sampled;                        // [B][g_stride] engine's own samples
  const float* audio_emb; const float* pe; const float* alpha_audio;
  float* x;                                // [B][d] next step's input
  int ctx_max;
  const int32_t* slot_map = nullptr;       // slot API: block i serves utterance slot_map[i] (B = number of listed slots)
  const unsigned long long* slot_seed = nullptr;  // slot API: RNG seed of the request each slot holds (null: request_seed(dyn.seed, b))
  int32_t* id_err = nullptr;               // |= 4 when a forced token is outside the audio vocabulary (it is replaced by 0)
  KTrace kt;
  LnProducer lnp;                          // batched step with fused LayerNorm: also emit bf16(x * gamma) + group statistics
  // host-visible progress (pinned, mapped): [0] = utterances done before this launch, [1] = sample launches so far.  The host
  // polls these words instead of putting a D2H copy + event on the stream after every graph replay.
  int32_t* host_prog = nullptr;
};
// VALL-F cross-attention (block API): q [Tq x d], kv [S x 2d] = [K | V] of the memory sequence, no mask; dtype DT_F32 / DT_BF16
int launch_cross_attention(hipStream_t st, int dtype, const void* q, const void* kv, void* out, int Tq, int S, int d, int nhead);
int launch_ar_sample(hipStream_t st, const ArSampleArgs& a);
// The persistent batch-1 step with the sampling step INSIDE the launch (persist.hip): nsteps AR iterations per launch.  After the
// predict layer every workgroup gathers the 1025 logits, draws the token with the sampling kernel's own code (sampling_dev.h: the
// same Philox stream, so every workgroup draws the same token), applies the stop rule and builds the next step's input row itself;
// workgroup 0 alone writes the utterance's state, the token history and the host progress words (ar_sample_kernel's stores).
struct PStepSample {  // PStepArgs::smp (device memory); PStepArgs::nsteps = 0: the launch ends with the logits of one step and launch_ar_sample follows
  ArState s{};
  const ArDyn* dyn = nullptr;
  int bos = 0;
  int pe_rows = 0;                         // rows of the position table `pe` (requests are kept inside it)
  int64_t* tokens = nullptr; int64_t* sampled = nullptr; int64_t g_stride = 0;
  const float* audio_emb = nullptr; const float* pe = nullptr; const float* alpha_audio = nullptr;
  float* x = nullptr;                      // [d] the input row of the step after this launch's last
  int32_t* id_err = nullptr;
  int32_t* host_prog = nullptr;
};
// stand-alone topk_sampling (valle.py:1287-1302) per row of logits[rows][V]: out[row] = draw with Philox(request_seed(seed, row), it),
// argmax_out[row] (nullable) = arg-max of the raw row
int launch_topk_sample_rows(hipStream_t st, const float* logits, int64_t rows, int V, int top_k, float temperature, unsigned long long seed,
                            unsigned it, int64_t* out, int64_t* argmax_out);
// slot API (continuous batching): per-slot AR state of newly admitted utterances; rows of X scattered to slot rows
int launch_slot_state_init(hipStream_t st, int32_t* state, int max_B, const int32_t* slots, const int32_t* kv_len,
                           const int32_t* audio_pos, const int32_t* cap, int n, unsigned long long* slot_seed = nullptr,
                           unsigned long long seed = 0, unsigned long long first_request = 0);
// fragment-major copy of W[N][K] (bf16 or fp8 codes) for gemm_skinny.hip; dst holds ceil(N/16)*16*K elements
int launch_pack_w_frag(hipStream_t st, const void* src, void* dst, int N, int K, int fp8);
int launch_scatter_rows(hipStream_t st, const float* src, const int32_t* src_rows, float* dst, const int32_t* dst_rows, int rows, int d);

struct NarArgmaxArgs {
  const float* logits; int V;          // [rows][V]
  int64_t rows;
  const int32_t* grow_seq; const int32_t* grow_pos; // generated rows -> (b, g)
  const int32_t* aoff; const int32_t* prompt_len;
  int64_t* codes; int64_t g_stride; int Q; int col;  // codes[b][g][col] = argmax
  const float* next_emb;  // table added to y_emb rows (null at the last stage)
  float* y_emb; int d;
  const int64_t* forced = nullptr; int64_t f_stride = 0;  // parity hook: y_emb += next_emb[forced[b][g][col]] instead of the own arg-max
};
int launch_nar_argmax(hipStream_t st, const NarArgmaxArgs& a);

// ---- misc.hip (batch > 8 AR-step glue + output packing) ------------------------------------------
// qkv[T][B][3d] -> q f32 [B][d], K/V -> cache slot kv_len[b]
int launch_qkv_split(hipStream_t st, int dtype, const void* qkv, float* q, void* k_cache, void* v_cache,
                     const int32_t* kv_len, int B, int d, int nhead, int ctx_max);
// merge decode-attention partials -> out[T][B][d]
int launch_attn_combine(hipStream_t st, int dtype, const float* part_o, const float* part_ml, void* out, int B, int nhead,
                        int dh, int nsplit);
// token-id range check + sanitise on the engine-owned input copies (flag |= code when an id is out of range; see misc.hip)
int launch_check_ids(hipStream_t st, int64_t* ids, int64_t stride0, int row_stride, int inner, const int32_t* lens, const int32_t* slot_map,
                     int n, int max_rows, int limit0, int limit_rest, int32_t* flag, int code);
// codes[b][g][0] = first_cb[b][g] for the generated rows
int launch_codes_set_first(hipStream_t st, const int64_t* first_cb, int64_t fc_stride, const int32_t* grow_seq,
                           const int32_t* grow_pos, int64_t rows, int64_t* codes, int64_t g_stride, int Q);

}  // namespace vle
