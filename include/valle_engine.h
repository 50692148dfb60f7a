/*
 * valle_engine.h -- C ABI of the MI355X-native VALL-E decode engine (libvalle_engine.so).
 *
 * The reference (lifeiteng/vall-e) has no FFI layer: its seam for this path is the Python
 * method surface VALLE.inference()/continual() and the Transformer block modules under it.
 * Each entry point below names the reference interface it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types; every function returns 0 on success or a
 *     negative VLE_E* code; vle_last_error() returns the message of the last failing call.
 *   - "DEVICE" pointers are HBM addresses owned by the caller (e.g. tensor.data_ptr());
 *     "HOST" pointers are ordinary host memory.  `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream); the engine orders its work against it with events.
 *   - the engine owns packed weights, the KV cache, workspaces, RNG state and captured
 *     hipGraphs (all sized at vle_create); one engine per (GPU, caller thread).
 *   - batch is an extension: the reference asserts batch == 1 (valle/models/valle.py:989);
 *     a batch here is B independent utterances (the oracle for it is B reference calls).
 */
#ifndef VALLE_ENGINE_H_
#define VALLE_ENGINE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLE_OK 0
#define VLE_EINVAL (-1)   /* bad argument / unsupported configuration                     */
#define VLE_ESTATE (-2)   /* call out of order (e.g. generate before prefill)             */
#define VLE_EHIP (-3)     /* a HIP runtime call failed                                     */
#define VLE_EKEY (-4)     /* unknown / mis-shaped state_dict key                           */
#define VLE_ENOTOKEN (-5) /* EOS at the very first AR step: the reference raises SyntaxError
                             ("well trained model shouldn't reach here", valle.py:1049-1052) */
#define VLE_EINDEX (-6)   /* a token id is outside its vocabulary (text >= 512, first codebook > 1024, other codebooks
                             >= 1024, or negative): the reference's nn.Embedding raises IndexError
                             (valle/modules/embedding.py:34,44).  The engine replaced the id by 0 before using it. */
#define VLE_EBUSY (-7)    /* vle_ar_generate / vle_slots_step: the persistent AR launch (one workgroup per CU for the whole AR loop;
                             1 .. 6 utterances) could not keep the whole GPU -- another workload held CUs for > 0.1 s and a wave
                             gave up waiting.  This call's tokens are invalid.  Repeat vle_ar_prefill + vle_ar_generate (slot mode:
                             vle_slots_begin and admit the utterances again): the engine runs its next such calls on the launch
                             chain (2, then 4 ... 64 calls while it keeps happening) and re-arms the persistent launch by itself.
                             valle_amd.VALLE.inference does the repeat (valle_amd/model.py). */

/* arithmetic mode of the whole path */
#define VLE_DTYPE_F32 0  /* fp32 weights / KV / accumulate: token-id-exact vs the reference */
#define VLE_DTYPE_BF16 1 /* bf16 weights + KV, fp32 residual stream and accumulators       */
#define VLE_DTYPE_FP8 3  /* EXPERIMENTAL -- not the mode BASELINE configs[4] is quoted on (that is VLE_DTYPE_FP8W below).  FP8W plus
                            fp8 ACTIVATIONS on the MFMA-bound passes (prefill, NAR): every Linear there runs e4m3fn x e4m3fn on
                            CDNA4's block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) with the activations quantised per
                            ROW (one power-of-two scale, unit MX block scales).  Per-row activation scales hold only a 15 % sigma
                            max / 3 % mean logit bar at d1536-L24 (the bf16 / FP8W bar is 5 %); per-32-column MX scales fused into
                            the producers were not built.  The AR step is FP8W's. */
#define VLE_DTYPE_FP8W 2 /* BASELINE configs[4]'s weight format, the mode that config is quoted on.  BF16 mode on fp8-representable weights: every Linear weight row is replaced by
                          * W' = e4m3fn(w / 2^e) * 2^e (one power-of-two scale per row, so W' is exact in bf16); the
                          * HBM-bound AR step streams the 1-byte codes, prefill / NAR run bf16 MFMA on bf16(W') */

typedef struct vle_engine vle_engine;

/* Constructor surface of VALLE(d_model, nhead, num_layers, norm_first, add_prenet, prefix_mode,
 * share_embedding, nar_scale_factor, prepend_bos, num_quantizers) -- valle/models/valle.py:727-760
 * / :54-84 -- plus capacity limits.  Only the production shape runs natively (norm_first = 1,
 * add_prenet = 0, nar_scale_factor = 1); anything else returns VLE_EINVAL. */
typedef struct vle_config {
  int32_t d_model;
  int32_t nhead;
  int32_t num_layers;
  int32_t num_quantizers; /* 1..8 */
  int32_t prefix_mode;    /* 0, 1, 2, 4 (valle/models/__init__.py:64-69) */
  int32_t prepend_bos;    /* 0 / 1 */
  int32_t norm_first;     /* must be 1 */
  int32_t add_prenet;     /* must be 0 */
  int32_t dtype_mode;     /* VLE_DTYPE_* */
  int32_t max_batch;      /* utterances decoded together */
  int32_t max_text;       /* S_max  (text tokens incl. BOS/EOS) */
  int32_t max_prompt;     /* P_max  (prompt frames) */
  int32_t max_gen;        /* G_max  (generated frames); 0 => 16*max_text + 1 */
  int32_t device;         /* HIP device ordinal */
  int32_t use_graph;      /* 1: capture the AR step in a hipGraph (default), 0: eager launches */
  int32_t steps_per_graph;/* AR steps per captured graph (0 => default 8) */
  int32_t reserved[8];
} vle_config;

/* get_model(params) + VALLE.__init__  (valle/models/__init__.py:98-136) */
int vle_create(const vle_config* cfg, vle_engine** out);
void vle_destroy(vle_engine* e);
const char* vle_last_error(const vle_engine* e); /* e may be NULL: last vle_create error */

/* model.load_state_dict(ckpt["model"], strict=True)  (valle/bin/infer.py:135-138).
 * `key` is exactly a reference state_dict key, `data` fp32, HOST memory,
 * C-contiguous with `shape[ndim]`.  Extra key "position.pe" (max_pos, d_model): the sinusoid table
 * SinePositionalEmbedding.extend_pe builds (valle/modules/embedding.py:75-91); optional -- the
 * engine computes it itself when absent. */
int vle_load_tensor(vle_engine* e, const char* key, const float* data, const int64_t* shape, int ndim);
/* strict=True check (every key present), AdaLN folding in fp32 (gamma' = w*gamma,
 * beta' = w*beta + b per stage and norm site, valle/modules/transformer.py:93-108), conversion to the
 * compute dtype and upload. */
int vle_finalize_weights(vle_engine* e);

/* ---- VALLE.inference()  (valle/models/valle.py:961-1137), split in its three phases ---------- */

/* valle.py:993-1038 for the first loop iteration: embed text + prompt (first codebook), run the
 * prefix-LM masked decoder over [text; prompt], fill the KV cache, leave step-0 logits ready.
 *   text          DEVICE int64 [B, s_stride]        (x;  ids in [0,512))
 *   text_lens     HOST   int32 [B]                  (x_lens)
 *   prompt_codes  DEVICE int64 [B, p_stride, Q]     (y;  codes in [0,1024))
 *   prompt_lens   HOST   int32 [B]                                                        */
int vle_ar_prefill(vle_engine* e, void* stream, const int64_t* text, int64_t s_stride, const int32_t* text_lens,
                   const int64_t* prompt_codes, int64_t p_stride, const int32_t* prompt_lens, int32_t B);

/* The AR `while True` loop, valle.py:1012-1057, with the reference's stop rule (:1044-1055) evaluated
 * per utterance on the device.  top_k / temperature as topk_sampling (:1287-1302); top_k == 1 is the
 * reference's greedy (lowest index wins ties).  `seed` feeds the engine's counter-based RNG (the
 * reference draws from torch's global generator; sampled streams are not comparable, logits are).
 *   max_new       0 => reference rule only; >0 => additionally stop after max_new frames
 *   forced        DEVICE int64 [B, forced_stride] or NULL: teacher-forced token history (parity hook;
 *                 utterance b then runs exactly forced_lens[b] steps)
 *   forced_lens   HOST int32 [B] or NULL
 *   codes0        DEVICE int64 [B, g_stride] or NULL: first-codebook tokens
 *   gen_lens      HOST int32 [B]: frames generated per utterance (G_b)
 * Blocks the host until every utterance stopped.  Returns VLE_ENOTOKEN if some utterance produced
 * no frame and prepend_bos == 0 (the reference's SyntaxError case). */
int vle_ar_generate(vle_engine* e, void* stream, int32_t top_k, float temperature, uint64_t seed, int32_t max_new,
                    const int64_t* forced, int64_t forced_stride, const int32_t* forced_lens,
                    int64_t* codes0, int64_t g_stride, int32_t* gen_lens);

/* The 7 NAR stages, valle.py:1062-1137, on the utterances of the last prefill/generate.
 *   enroll_lens   HOST int32 [B] or NULL (required for prefix_mode 2/4, valle.py:1068-1079)
 *   codes         DEVICE int64 [B, g_stride, Q]: all Q codebooks, frames >= G_b untouched */
int vle_nar_decode(vle_engine* e, void* stream, const int32_t* enroll_lens, int64_t* codes, int64_t g_stride);

/* Grow the capacities chosen at vle_create IN PLACE (each becomes max(old, requested); max_gen 0 = 16 * max_text + 1): the
 * weights, their packed / quantised copies and the tuned options stay; only the capacity-dependent buffers (KV cache,
 * activations, tables, traces) are re-created and the captured graphs dropped.  Call between decodes (any prefill / slot state
 * is lost).  pe: HOST fp32 [pe_rows, d] sinusoid table built like SinePositionalEmbedding.extend_pe
 * (valle/modules/embedding.py:75-91), needed when the position range grows (NULL = the engine's own restatement of it).
 * Before vle_finalize_weights it only records the new sizes. */
int vle_reserve(vle_engine* e, int32_t max_batch, int32_t max_text, int32_t max_prompt, int32_t max_gen, const float* pe,
                int64_t pe_rows);

/* Parity hook (the NAR analogue of vle_ar_generate's `forced`): the NEXT vle_nar_decode teacher-forces its stage
 * history -- after stage i the embedding added to y_emb (valle.py:1133-1134) is the one of forced_codes[b][g][i+1]
 * instead of the stage's own arg-max; the codes written are still the engine's own arg-max.  This is what makes the
 * per-stage logits of a reduced-precision mode comparable with the reference's.
 *   forced_codes  DEVICE int64 [B, f_stride, Q] (must stay valid until that vle_nar_decode returned), or NULL to clear */
int vle_nar_force(vle_engine* e, const int64_t* forced_codes, int64_t f_stride);

/* ---- VALLE.continual()  (valle/models/valle.py:1139-1238): NAR only ------------------------- */
/*   y_codes DEVICE int64 [B, t_stride, 8], y_lens HOST int32 [B]; prefix_b = min(T_b/2, 225) (:1173);
 *   codes DEVICE int64 [B, g_stride, 8] receives (T_b - prefix_b) frames; gen_lens HOST int32 [B]. */
int vle_nar_continual(vle_engine* e, void* stream, const int64_t* text, int64_t s_stride, const int32_t* text_lens,
                      const int64_t* y_codes, int64_t t_stride, const int32_t* y_lens, int32_t B,
                      int64_t* codes, int64_t g_stride, int32_t* gen_lens);

/* ---- slot API: continuous batching (callers either side of inference(): valle/bin/infer.py:223-269 decodes one
 * utterance at a time) ------------------------------------------------------------------------------------------
 * The engine's max_batch positions are SLOTS, each free or holding one utterance.  vle_slots_begin frees them all;
 * vle_slots_prefill admits n new utterances into the listed free slots (prefill of those only + their first sample);
 * vle_slots_step advances every live slot by nsteps AR steps (finished / free slots cost no KV traffic) and returns the
 * per-slot done flags and generated lengths (HOST int32 [max_batch]); vle_slots_harvest runs the 7 NAR stages for the
 * listed FINISHED slots into codes[slot][g][q] (DEVICE int64, row pitch g_stride * num_quantizers) and frees them.
 * Numerics and stop rule are those of vle_ar_prefill / vle_ar_generate / vle_nar_decode: what an utterance decodes to
 * does not depend on what shares the batch.  top_k / temperature / seed apply to the samples drawn by that call. */
int vle_slots_begin(vle_engine* e, void* stream);
int vle_slots_prefill(vle_engine* e, void* stream, int32_t n, const int32_t* slots, const int64_t* text, int64_t s_stride,
                      const int32_t* text_lens, const int64_t* prompt_codes, int64_t p_stride, const int32_t* prompt_lens,
                      int32_t top_k, float temperature, uint64_t seed);
int vle_slots_step(vle_engine* e, void* stream, int32_t nsteps, int32_t top_k, float temperature, uint64_t seed,
                   int32_t* done_out, int32_t* gen_lens_out);
int vle_slots_harvest(vle_engine* e, void* stream, int32_t n, const int32_t* slots, const int32_t* enroll_lens, int64_t* codes,
                      int64_t g_stride);

/* ---- parity / measurement hooks -------------------------------------------------------------- */
/* options: "trace_ar_logits" (0/1: keep every AR step's fp32 logits), "trace_nar_logits" (0/1),
 *          "nsplit" (0 = auto, else 1|2|4|8|16: KV split of the decode attention),
 *          "no_gemv1" (1: batch-1 AR step on the generic skinny kernel instead of gemv1.hip), "gemv1_rpw" (rows per wave, tuning),
 *          "no_gemm_skinny" (1: batch 2..64 AR step on the v0 kernels), "attn_nk" (0|4|8 keys per lane per round),
 *          "steps_per_graph" (n > 0 overrides vle_config.steps_per_graph),
 *          "ignore_eos" (1: benchmark hook for random-init weights -- only the 16*S length cap stops an utterance),
 *          "profile_kernels" (n > 0: time each kernel of the next n AR steps with hipEvents on the
 *           engine stream, launches become eager; 0: off);
 *          kernel A/B knobs, each described with its measurement where the kernel is defined:
 *          "qkv_attn", "qa_handoff", "qa_nsplit", "g1_shared" (batch-1 step); "gs_fast" (compile-time-layout bodies of the
 *          batched GEMMs, default 1), "gs_nf" (two W fragments per workgroup where the grid exceeds the chip, default 1), "gs_msplit",
 *          "gs_formal", "gs_gran" (split-K hand-off through granules, default 0),
 *          "gs_fuse_ln", "attn_oproj", "attn_nt", "attn_lds_pad" (batched step); "ktrace" (in-kernel timeline);
 *          the persistent batch-1 AR launch (valle_amd/csrc/persist.hip): "persist" (default 1: d1024-h16 at one
 *          utterance runs it in every engine mode; 0 = the launch chain), "persist_sample" (default 1: topk_sampling, the stop rule and the next token's
 *          embedding inside the launch) with "persist_steps" (default 32 AR iterations per launch), "persist_mode" (bit field: 4 / 8
 *          hidden / attention rows as bf16 pairs, 16 XCD-local copies of the head-group edges, 32 folded LayerNorm, 64 bf16 activation
 *          rows + v_dot2c_f32_bf16 dot products; default 0x174),
 *          "persist_pf" (0 | 3 operand request schedule), "persist_nk" (2 keys per lane) -- since round 6 only the shipped forms are compiled:
 *          other values (pf 1 / 2, nk 4, packing modes the measurements dropped) make the call run the launch chain --, "persist_naps" (first-sweep waits, 4 bits
 *          per edge; -1 = the engine mode's measured default: bf16 0x325756, fp8 weight rows 0x214645, fp32 0x217645), "persist_trace" (in-kernel timeline), "act_bf16" (the chain's matching roundings).
 *          "persist_batch" (default 1: calls of 2 .. 6 utterances on bf16 engines -- vle_ar_generate and, on engines of 2 .. 6 slots,
 *          vle_slots_step -- run the batched persistent launch, valle_amd/csrc/persist_nb.hip, DESIGN.md 4.2: the default form only; per utterance
 *          bit-identical to the one-utterance launch; 0 = the launch chain; first-sweep waits per batch 0x405745 / 0x305752 / 0x317780 / 0x006876 / 0x007860),
 *          "persist_rearm" (any value: forget the back-off after VLE_EBUSY), "persist_inject_fail" (n: the next n persistent calls
 *          end as if a wave had given up -- the test hook of the VLE_EBUSY path).
 *   debug words of vle_debug_fetch for it: "persist_active" (the next batch-1 call would run it), "persist_capable" (a ONE-utterance call on this engine
 *   would, whatever the batch of the last prefill), "persist_batch_capable" (the largest batch, 0 or 2 .. 6, a call on this engine would run on the
 *   batched persistent launch: valle_amd.VALLE.inference_batch decodes two utterances one after the other only where this says 0), "persist_ran" (the LAST
 *   vle_ar_generate / vle_slots_step did), "persist_fail" (waves that gave up in the last call; 0 in a healthy run), "persist_fallbacks" (calls that
 *   ended with VLE_EBUSY since vle_create), "persist_backoff" (batch-1 calls left on the launch chain before it is re-armed),
 *   "persist_sample_active", "ar_launches", "persist_trace".
 *   Environment (debugging, read once per process): VLE_GUARD_ALLOC=1|2 every engine allocation in its own virtual-memory mapping,
 *   ending (1) / starting (2) at the mapping's edge with an unmapped granule behind / in front; VLE_ALLOC_LOG=1 lists every engine
 *   allocation on stderr (a GPU memory-access fault's address then names its buffer); VLE_POISON_ALLOC=<byte> fills every engine
 *   allocation with that byte first (memory of a fresh box is not zero: reads of never-written state behave the same everywhere). */
int vle_set_option(vle_engine* e, const char* name, int64_t value);
/* what: "ar_logits"  -> fp32 [n_steps, B, 1025] (row t = logits of AR loop iteration t)
 *       "nar_logits:<stage>" -> fp32 [sum_b G_b, 1024]
 *       "ar_sampled" -> int64 [B, G_max] the engine's own samples (differs from codes0 only when forced)
 *       "kv_len" -> int32 [B]
 *       "kernel_times" -> double [16]: total ms then launch counts for the AR-step kernel families
 *                         {qkv, decode-attention, out-proj, ffn1, ffn2, logits, sample, -}
 *   returns the number of bytes written (>= 0) or an error code */
int64_t vle_debug_fetch(vle_engine* e, const char* what, void* host_dst, size_t bytes);
/* phase timings of the last call, milliseconds measured with hipEvents on the engine's stream:
 * out[0] prefill, out[1] AR steps, out[2] NAR, out[3] number of AR steps executed */
int vle_last_timings(vle_engine* e, double* out4);
/* Debugging aid for CALLER-owned buffers (inputs, forced tokens, outputs): `bytes` of device memory in a virtual-memory mapping of
 * their own, ENDING at the mapping's end (at_start = 0) or starting at its start (1), with an unmapped granule on either side -- an
 * overrun by any kernel of the path faults at once instead of landing in a neighbour's memory.  Never freed.  (No reference
 * counterpart: the reference's tensors are bounds-checked by ATen.) */
int vle_debug_guard_alloc(int32_t device, size_t bytes, int32_t at_start, void** out);
/* algorithmic bytes moved per AR step for the current batch at context length ctx */
int64_t vle_ar_step_bytes(const vle_engine* e, int32_t B, int64_t sum_ctx);

/* The weight format of VLE_DTYPE_FP8W, on HOST buffers (what vle_finalize_weights applies to every Linear weight):
 * per row n of w[f32, N x K]: scale_out[n] = the smallest power of two with max|w[n]| / scale <= 448,
 * q_out[n][k] = round-to-nearest-even e4m3fn(w[n][k] / scale) (OCP e4m3fn = torch.float8_e4m3fn),
 * deq_out[n][k] = q * scale = W'.  q_out / deq_out may be NULL. */
int vle_quantize_fp8w(const float* w, int64_t N, int64_t K, uint8_t* q_out, float* scale_out, float* deq_out);

/* ---- Transformer-block operator surface (valle/modules/transformer.py, activation.py) -------- */
/* Stand-alone kernels behind the block API (B3 in SURVEY.md 8b); all pointers DEVICE, row-major.
 * dtype = VLE_DTYPE_*: element type of `x`/`w`/`out` where marked T; fp32 where marked f32. */

/* LayerNorm.forward (transformer.py:57-74): out[T] = LN(x[f32]) * gamma + beta, eps 1e-5 */
int vle_op_layernorm(void* stream, int dtype, const float* x, const float* gamma, const float* beta, void* out,
                     int64_t rows, int32_t d);
/* nn.Linear / in_proj / out_proj (activation.py:414-421, transformer.py:333):
 * out = epilogue(a[T, M x K] @ w[T, N x K]^T + bias[f32, N]);
 * epilogue 0: store T; 1: ReLU, store T; 2: resid[f32, M x N] += (.) in place; 3: store f32 */
int vle_op_linear(void* stream, int dtype, const void* a, const void* w, const float* bias, void* out, float* resid,
                  int64_t M, int32_t N, int32_t K, int epilogue);
/* vle_op_linear with a caller-owned DEVICE workspace for the split-K form of the M <= 64 weight-streaming
 * GEMM (gemm_skinny.hip; N / 16 row fragments alone would leave CUs idle when N = d).  The workspace holds
 * vle_op_linear_workspace_bytes() bytes, its first 4096 zeroed once by the caller (the tickets reset
 * themselves); calls sharing a workspace must be ordered on one stream.  Deterministic: partials are
 * summed in slice order.  ksplit: 0 = chosen from the shape; other shapes ignore the workspace. */
int vle_op_linear_ws(void* stream, int dtype, const void* a, const void* w, const float* bias, void* out, float* resid,
                     int64_t M, int32_t N, int32_t K, int epilogue, void* workspace, int32_t ksplit);
/* LayerNorm folded into the packed-row GEMMs of the prefill / NAR passes (valle/modules/transformer.py:57-74 LayerNorm, :93-108
 * AdaptiveLayerNorm, followed by F.linear): LN(x) W^T + b = rstd * ((x * gamma) W^T - mean * sg) + tb with sg = W gamma,
 * tb = W beta + b.  The two halves as stand-alone operators, bf16 operands:
 *   producer: resid[M][N] (fp32) += a[M][K] @ w[N][K]^T + bias; xg[M][N] = bf16(resid * gamma);
 *             stats[N / 64][M][2] = (mean, sum of squared deviations) of every 64-column group of the completed row, group-major;
 *   consumer: out[M][N] = bf16(act(rstd_m * (xg[M][K] @ w[N][K]^T - mean_m * sg[n]) + tb[n])), the row's mean / rstd combined
 *             from stats[K / 64][M][2] (eps 1e-5, biased variance), act = ReLU when relu != 0.
 * M >= 128; the normalised width (producer N, consumer K) a multiple of 256 and <= 1536; the other dimension a multiple of 256
 * (consumer N) / of 128 (producer K). */
int vle_op_linear_ln_producer(void* stream, const void* a, const void* w, const float* bias, float* resid, const float* gamma,
                              void* xg, float* stats, int64_t M, int32_t N, int32_t K);
int vle_op_linear_ln_consumer(void* stream, const void* xg, const void* w, const float* tb, const float* sg, const float* stats,
                              void* out, int64_t M, int32_t N, int32_t K, int32_t relu);
int64_t vle_op_linear_workspace_bytes(void);
/* The two weight-streaming kernels of the AR step on FP8W weights (w8 e4m3fn [N x K], wscale f32 [N], DEVICE):
 * vle_op_linear_skinny_fp8w: one utterance (gemv1.hip), contract of vle_op_linear_skinny with M = 1;
 * vle_op_linear_fp8w: 1 <= M <= 64 rows of bf16 activations (gemm_skinny.hip), contract of vle_op_linear_ws.
 * Both equal the bf16 kernels run on W' = w8 * wscale (bit-identical for the GEMV, fp32 accumulation order aside). */
int vle_op_linear_skinny_fp8w(void* stream, const float* x, const float* gamma, const float* beta, const void* w8,
                              const float* wscale, const float* bias, float* out, float* resid, int32_t N, int32_t K, int epilogue);
int vle_op_linear_fp8w(void* stream, const void* a, const void* w8, const float* wscale, const float* bias, void* out, float* resid,
                       int64_t M, int32_t N, int32_t K, int epilogue, void* workspace, int32_t ksplit);
/* Kernel-selection knobs of the stand-alone operators, process-global (tests and microbenchmarks):
 * "glds_big" -1 default | 0 never | n: full-tile count from which the bf16 GEMM uses its 8-wave 256 x 128 tile;
 * "glds_8ph" 0 never | -1 | n: tile count from which it uses the phase-split 256 x 256 kernel; "glds_t64" n: 128 x 64 tiles from n of them, 64 x 64 below (default 160); "glds_tail" 1 | 0: the rows past
 * its last full 256-row tile as a second small launch when that saves a round of tiles; "glds_w8", "glds_swz",
 * "glds_prio" 0 | 1: A/B switches of the tile kernels; "g8_persist" bits 1 | 2 | 4: which launches of the 256 x 256 kernel take its
 * persistent tile loop (bf16-output forms | residual forms K < 2048 | K >= 2048; default 1); "f32_glds" 1 | 0: fp32 packed-row GEMMs on the
 * LDS-DMA ring | the register-staged kernel (bit-identical); "attn_f32_vec" 1 | 0: 16-byte double-buffered K / V staging of the fp32 attention
 * (bit-identical).  The three are also engine options (vle_set_option), process-global. */
int vle_op_tune(const char* name, int64_t value);
/* Same contract on the skinny (M <= 8, fp32 activations) weight-streaming path of the AR step:
 * x[f32, M x K]; optional fused LayerNorm prologue when gamma != NULL;
 * epilogue 0: out f32 = (.); 1: ReLU -> out f32; 2: resid += (.) */
int vle_op_linear_skinny(void* stream, int dtype, const float* x, const float* gamma, const float* beta, const void* w,
                         const float* bias, float* out, float* resid, int32_t M, int32_t N, int32_t K, int epilogue);
/* F.multi_head_attention_forward core (activation.py:408-427) on packed sequences:
 * qkv[T, rows x 3d] (rows of utterance b = seq_off[b] .. seq_off[b+1]), out[T, rows x d];
 * row i of an utterance attends keys j < max(text_len[b], causal ? i+1 : len_b)  -- the prefix-LM
 * mask of valle.py:1019-1033 when causal = 1, no mask (NAR) when causal = 0. */
int vle_op_attention(void* stream, int dtype, const void* qkv, void* out, const int32_t* seq_off_dev,
                     const int32_t* text_len_dev, int32_t B, int32_t max_len, int32_t d, int32_t nhead, int causal);

/* Cross-attention of VALL-F's decoder layers (valle/modules/transformer.py:582-597, MultiheadAttention.forward(x, mem, mem),
 * valle/modules/activation.py:199-431): q [Tq x d] (the projected, un-scaled queries), kv [S x 2d] = [K | V] rows of the memory
 * (text) sequence, heads = contiguous d / nhead slices (<= 128), no mask; out [Tq x d]; dtype VLE_DTYPE_F32 / _BF16 for all three. */
int vle_op_cross_attention(void* stream, int dtype, const void* q, const void* kv, void* out, int32_t Tq, int32_t S, int32_t d,
                           int32_t nhead);
/* The same attention for ONE new query per utterance against the head-major KV cache
 * [B][H][ctx_max][dh] (element type T): the last row of F.multi_head_attention_forward under the
 * prefix-LM mask of valle.py:1019-1033 (the new token sees cache slots 0 .. kv_len[b]).  The reference
 * recomputes every row each step (valle.py:1004 TODO); this is the KV-cache form of that row.
 *   q f32 [B][d]; kv_len_dev int32 [B] (slot of the newest key); nsplit in {1,2,4,8,16}
 *   workspace f32 [B * nsplit * (d + 2 * nhead)]: the split partials (part_o [B][nsplit][d] then
 *   part_ml [B][nhead][nsplit][2]), left there for vle_op_attn_out_proj
 *   out f32 [B][d] or NULL: merged, normalised attention output */
int vle_op_decode_attention(void* stream, int dtype, const float* q, const void* k_cache, const void* v_cache,
                            const int32_t* kv_len_dev, float* workspace, float* out, int32_t B, int32_t nhead, int32_t dh,
                            int32_t ctx_max, int32_t nsplit);
/* out_proj of the AR step with the split merge fused in front (activation.py:421 `linear(attn_output,
 * out_proj_weight, out_proj_bias)` + the residual add of transformer.py:297): resid[f32, B x d] +=
 * merge(workspace) @ w[T, d x d]^T + bias; B <= 8. */
int vle_op_attn_out_proj(void* stream, int dtype, const float* workspace, const void* w, const float* bias, float* resid,
                         int32_t B, int32_t nhead, int32_t dh, int32_t nsplit);

/* The attention half of ONE utterance's decode step as the batch-1 AR step runs it (two launches; gemv1.hip qkv_attn1_kernel +
 * the out-proj GEMV with the PRO_ATTN_SELF prologue):  x += out_proj(MHA(LayerNorm(x)))  for the one new token, i.e.
 * `x + self._sa_block(self.norm1(x))` of valle/modules/transformer.py:296-297 on the last row, with the in-projection
 * (valle/modules/activation.py:414-421) appending the token's K / V to cache slot kv_len[0] and the softmax running over slots
 * 0 .. kv_len[0] (prefix-LM mask, valle.py:1019-1033).
 *   x f32 [d] in/out; gamma / beta f32 [d] (norm1); w_in T [3d][d], b_in f32 [3d]; w_out T [d][d], b_out f32 [d];
 *   k_cache / v_cache cache type (T; bf16 when T is bf16) [nhead][ctx_max][dh]; kv_len_dev int32 [1] < ctx_max;
 *   workspace f32 [3 d + nsplit * (d + 2 * nhead)]; nsplit in {4, 8, 16}; dtype VLE_DTYPE_F32 / _BF16.
 * Shapes the fused launch does not cover (dh not in {64, 128}, d / (64 * vector) not in {1, 2, 4}) fail with VLE_EINVAL: the
 * engine then takes vle_op_linear_skinny + vle_op_decode_attention + vle_op_attn_out_proj. */
int vle_op_attn_step1(void* stream, int dtype, float* x, const float* gamma, const float* beta, const void* w_in, const float* b_in,
                      const void* w_out, const float* b_out, void* k_cache, void* v_cache, const int32_t* kv_len_dev, float* workspace,
                      int32_t nhead, int32_t dh, int32_t ctx_max, int32_t nsplit);

/* TokenEmbedding.forward (valle/modules/embedding.py:43-47): out[f32, n x d] = table[f32, V x d][ids[i64, n]].
 * ids must lie in [0, V) (like nn.Embedding on a device, no range check on the hot path). */
int vle_op_token_embedding(void* stream, const int64_t* ids, const float* table, float* out, int64_t n, int32_t d);
/* inout[r] += table[ids[r]]: `y_emb += nar_audio_embeddings[j](codes)` of the NAR prompt / stage update (valle.py:1104-1113, 1134) */
int vle_op_token_embedding_add(void* stream, const int64_t* ids, const float* table, float* inout, int64_t n, int32_t d);
/* SinePositionalEmbedding.forward (embedding.py:93-97): out[b][t] = x[b][t] * x_scale + alpha[0] * pe[t];
 * x/out f32 [B x T x d], pe f32 [>= T x d] (built as embedding.py:75-91), alpha f32 DEVICE scalar. */
int vle_op_sine_positional(void* stream, const float* x, const float* pe, const float* alpha_dev, float x_scale, float* out,
                           int64_t B, int32_t T, int32_t d);
/* Rows of the teacher-forced forward's losses and metrics (VALLE.forward, valle/models/valle.py:875, 877-879, 936-956):
 * loss[r] = logsumexp(logits[r]) - logits[r][targets[r]] (0 when targets[r] == ignore_index or is outside [0, V)),
 * hit[r] = 1 / 0 whether the target is among the topk largest logits (ties towards the lower index), -1 for an ignored
 * row.  logits f32 [rows x V], targets i64 [rows], loss f32 [rows], hit i32 [rows], all DEVICE. */
/* Engine mode FP8 (BASELINE configs[4], "CDNA4 fp8 MFMA"), stand-alone:
 * vle_op_quantize_rows_fp8: x[bf16, rows x K] -> q[e4m3fn, rows x K] + scale[f32, rows] (the smallest power of two with
 *   max|x_row| / scale <= 448; round-to-nearest-even);  K = 512 * {1,2,3,4,6,8,12,16}.
 * vle_op_linear_fp8: epilogue((a8 @ w8^T) * a_scale[m] * w_scale[n] + bias) on v_mfma_scale_f32_16x16x128_f8f6f4 (unit
 *   block scales); a8 [M x K], w8 [N x K] e4m3fn codes; out bf16 (STORE / RELU) or f32 (F32), resid f32 += (RESID);
 *   K % 128 == 0, N % 4 == 0.  Replaces the same `linear` calls as vle_op_linear. */
int vle_op_quantize_rows_fp8(void* stream, const void* x_bf16, void* q_out, float* scale_out, int64_t rows, int32_t K);
int vle_op_linear_fp8(void* stream, const void* a8, const float* a_scale, const void* w8, const float* w_scale, const float* bias,
                      void* out, float* resid, int64_t M, int32_t N, int32_t K, int epilogue);
int vle_op_cross_entropy(void* stream, const float* logits, const int64_t* targets, float* loss, int32_t* hit, int64_t rows, int32_t V,
                         int32_t ignore_index, int32_t topk);
/* topk_sampling (valle/models/valle.py:1287-1302; top_k_top_p_filtering :1242-1284 with top_p = 1.0) of each row of
 * logits [f32, rows x V], V <= 1280: divide by `temperature` (if != 1), keep the top_k largest (ties at the k-th value kept;
 * top_k <= 0 or >= V keeps all, top_k == 1 is the arg-max), softmax, draw samples[row] (int64) with
 * u = Philox4x32-10(request seed of (seed, row), step) -- the stream of request `row` of vle_ar_generate(seed) at AR step `step`.
 * argmax [int64, rows] (nullable) receives the arg-max of the raw row (lowest index among ties): the stop rule :1044-1048 needs both. */
int vle_op_topk_sample(void* stream, const float* logits, int64_t rows, int32_t V, int32_t top_k, float temperature, uint64_t seed,
                       uint32_t step, int64_t* samples, int64_t* argmax);
/* AdaptiveLayerNorm.forward (transformer.py:93-108) as an affine fold: with wb = project_layer(stage_emb)
 * [f32, 2d] = [w ; b] and the inner norm's (g, be): gamma_out = w * g, beta_out = w * be + b, so that
 * vle_op_layernorm(x, gamma_out, beta_out) == w * LayerNorm(x) + b (the fold vle_finalize_weights applies). */
int vle_op_adaln_fold(void* stream, const float* wb, const float* g, const float* be, float* gamma_out, float* beta_out,
                      int32_t d);

/* ---- EnCodec 24 kHz decoder: codes -> waveform --------------------------------------------------
 * Replaces audio_tokenizer.decode([(codes.transpose(2, 1), None)]) (valle/bin/infer.py:261-263; AudioTokenizer.decode ->
 * EncodecModel.decode, valle/data/tokenizer.py:241-242; third-party `encodec`, encodec_model_24khz at 6 kbps = 8 codebooks).
 * Parity is UNPINNED (no weights / package offline): the implementation follows the published architecture and is checked
 * against oracle/encodec_oracle.py.
 *   vle_codec_load_tensor: HOST fp32 tensors under the encodec state-dict names -- quantizer.vq.layers.{q}._codebook.embed
 *     (1024, 128); decoder.model.{0,15}.conv.conv.{weight_g, weight_v, bias}; decoder.model.1.lstm.{weight,bias}_{ih,hh}_l{0,1};
 *     decoder.model.{3,6,9,12}.convtr.convtr.{weight_g, weight_v, bias}; decoder.model.{4,7,10,13}.{block.1, block.3,
 *     shortcut}.conv.conv.* (a plain `weight`, or torch's parametrizations.weight.original{0,1}, is accepted for weight_g/v)
 *   vle_codec_finalize: folds weight_norm, re-lays every convolution out as a GEMM operand, uploads
 *   vle_codec_decode: codes DEVICE int64 [T, n_q] (one utterance, the layout VALLE.inference returns) -> wav DEVICE f32 [320 T],
 *     24 kHz mono; asynchronous w.r.t. the host like the engine's calls. */
typedef struct vle_codec vle_codec;
int vle_codec_create(int32_t device, int32_t n_q, vle_codec** out);
int vle_codec_load_tensor(vle_codec* c, const char* key, const float* host_data, const int64_t* shape, int ndim);
int vle_codec_finalize(vle_codec* c);
int vle_codec_decode(vle_codec* c, void* stream, const int64_t* codes, int64_t T, float* wav);
const char* vle_codec_last_error(const vle_codec* c);
void vle_codec_destroy(vle_codec* c);

#ifdef __cplusplus
}
#endif
#endif /* VALLE_ENGINE_H_ */
