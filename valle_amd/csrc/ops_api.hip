// C ABI of the stand-alone Transformer-block operators (valle_engine.h, "operator surface"):
// thin argument checks around the kernel launchers, no engine instance needed.
#include <string>

#include "common.h"
#include "kernels.h"
#include "valle_engine.h"

using namespace vle;

static int op_fail(const char* m) {
  set_global_error(m);
  return VLE_EINVAL;
}
static int op_done(int r, const char* what) {
  if (r != 0) {
    set_global_error(what);
    return r == -3 ? VLE_EHIP : VLE_EINVAL;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
  return VLE_OK;
}

extern "C" int vle_op_layernorm(void* stream, int dtype, const float* x, const float* gamma, const float* beta, void* out,
                                int64_t rows, int32_t d) {
  if (!x || !gamma || !beta || !out || d % 4) return op_fail("vle_op_layernorm: bad argument");
  return op_done(launch_layernorm((hipStream_t)stream, dtype, x, nullptr, gamma, beta, out, rows, d), "vle_op_layernorm");
}

extern "C" int vle_op_tune(const char* name, int64_t value) {
  if (!name) return op_fail("vle_op_tune: null name");
  const std::string n = name;
  if (n == "glds_big" && value >= -1) vle::g_glds_big = (int)value;
  else if (n == "glds_w8" && value >= 0 && value <= 1) vle::g_glds_w8 = (int)value;
  else if (n == "glds_prio" && value >= 0 && value <= 1) vle::g_glds_prio = (int)value;
  else if (n == "glds_swz" && value >= 0 && value <= 1) vle::g_glds_swz = (int)value;
  else if (n == "glds_8ph" && value >= -1) vle::g_glds_8ph = (int)value;
  else if (n == "glds_tail" && value >= 0 && value <= 1) vle::g_glds_tail = (int)value;
  else if (n == "glds_t64" && value >= 0 && value <= 100000) vle::g_glds_t64 = (int)value;
  else if (n == "g8_stagger" && value >= 0 && value <= 1) vle::g_g8_stagger = (int)value;
  else if (n == "g8_colgroup" && value >= 0 && value <= 16) vle::g_g8_colgroup = (int)value;
  else if (n == "g8_persist" && value >= 0 && value <= 7) vle::g_g8_persist = (int)value;
  else if (n == "f32_glds" && value >= 0 && value <= 1) vle::g_f32_glds = (int)value;
  else if (n == "attn_f32_vec" && value >= 0 && value <= 1) vle::g_attn_f32_vec = (int)value;
  else if (n == "attn_qw" && value >= 0 && value <= 2) vle::g_attn_qw = (int)value;
  else if (n == "attn_v2" && value >= 0 && value <= 2) vle::g_attn_v2 = (int)value;
  else if (n == "attn_xcd" && value >= 0 && value <= 1) vle::g_attn_xcd = (int)value;
  else if (n == "attn_q128" && value >= -1 && value <= 1) vle::g_attn_q128 = (int)value;
  else if (n == "glds_epi" && value >= 0 && value <= 1) vle::g_glds_epi = (int)value;
  else if (n == "g8_dbg" && value >= 0 && value <= 7) vle::g_g8_dbg = (int)value;
  else if (n == "g8_nt" && value >= 0 && value <= 3) vle::g_g8_nt = (int)value;
  else if (n == "attn_mode" && value >= 0 && value <= 3) vle::g_attn_mode = (int)value;
  else if (n == "attn_ring" && (value == 0 || value == 2 || value == 4)) vle::g_attn_ring = (int)value;
  else if (n == "qa_waves" && (value == 4 || value == 8)) vle::g_qa_waves = (int)value;
  else if (n == "g1_shared" && value >= 0 && value <= 1) vle::g_g1_shared = (int)value;
  else if (n == "gs_msplit" && value >= 0 && value <= 3) vle::g_gs_msplit = (int)value;
  else if (n == "gs_formal" && value >= 0 && value <= 1) vle::g_gs_formal = (int)value;
  else if (n == "gs_fast" && value >= 0 && value <= 1) vle::g_gs_fast = (int)value;
  else if (n == "gs_gran" && value >= 0 && value <= 1) vle::g_gs_gran = (int)value;
  else if (n == "gs_nf" && value >= 0 && value <= 1) vle::g_gs_nf = (int)value;
  else if (n == "attn_lsum" && value >= 0 && value <= 1) vle::g_attn_lsum = (int)value;
  else if (n == "attn_defer" && value >= 0 && value <= 16) vle::g_attn_defer = (int)value;
  else return op_fail("vle_op_tune: unknown knob or value out of range");
  return VLE_OK;
}

static int op_linear(void* stream, int dtype, const void* a, const void* w, const float* bias, void* out, float* resid, int64_t M,
                     int32_t N, int32_t K, int epilogue, void* workspace, int32_t ksplit, const char* who) {
  if (!a || !w) return op_fail("vle_op_linear: null operand");
  if (epilogue == EPI_RESID ? !resid : !out) return op_fail("vle_op_linear: missing output");
  if (dtype == DT_BF16 && M <= 64 && gemm_skinny_supports((int)M, N, K, epilogue, 4)) {  // the AR-step path of 2..64 utterances
    GemmSkinnyArgs g;
    g.x = a; g.w = w; g.bias = bias; g.M = (int)M; g.N = N; g.K = K; g.epi = epilogue; g.out = out; g.resid = resid;
    g.workspace = workspace; g.ksplit = ksplit;
    return op_done(launch_gemm_skinny((hipStream_t)stream, g), who);
  }
  return op_done(launch_gemm((hipStream_t)stream, dtype, a, w, bias, out, resid, M, N, K, epilogue), who);
}

extern "C" int vle_op_linear(void* stream, int dtype, const void* a, const void* w, const float* bias, void* out, float* resid,
                             int64_t M, int32_t N, int32_t K, int epilogue) {
  return op_linear(stream, dtype, a, w, bias, out, resid, M, N, K, epilogue, nullptr, 0, "vle_op_linear");
}

extern "C" int vle_op_linear_ws(void* stream, int dtype, const void* a, const void* w, const float* bias, void* out, float* resid,
                                int64_t M, int32_t N, int32_t K, int epilogue, void* workspace, int32_t ksplit) {
  if (ksplit < 0 || ksplit > 16 || (ksplit & (ksplit - 1))) return op_fail("vle_op_linear_ws: ksplit must be 0, 1, 2, 4, 8 or 16");
  return op_linear(stream, dtype, a, w, bias, out, resid, M, N, K, epilogue, workspace, ksplit, "vle_op_linear_ws");
}

// LayerNorm folded into the packed-row GEMMs (kernels.h GemmLn), the two halves as stand-alone operators (bf16):
//   producer: resid[M][N] += a @ w^T + bias; xg = bf16(resid * gamma); stats[N / 64][M][2] = (mean, M2) of every 64-column group (group-major)
//   consumer: out[M][N] = bf16(act(rstd * (xg @ w^T - mean * sg) + tb)) with the rows' mean / rstd combined from stats[K / 64][M][2]
extern "C" int vle_op_linear_ln_producer(void* stream, const void* a, const void* w, const float* bias, float* resid, const float* gamma,
                                         void* xg, float* stats, int64_t M, int32_t N, int32_t K) {
  if (!a || !w || !resid || !gamma || !xg || !stats) return op_fail("vle_op_linear_ln_producer: null operand");
  if (!gemm_ln_supports(DT_BF16, M, N) || K % 128 != 0 || K < 256) return op_fail("vle_op_linear_ln_producer: shape not covered (M >= 128, N % 256 == 0, N <= 1536, K % 128 == 0)");
  GemmLn ln;
  ln.gamma = gamma; ln.xg = xg; ln.stats_out = stats; ln.stats_ld = M;
  return op_done(launch_gemm((hipStream_t)stream, DT_BF16, a, w, bias, nullptr, resid, M, N, K, EPI_RESID_LNP, &ln), "vle_op_linear_ln_producer");
}
extern "C" int vle_op_linear_ln_consumer(void* stream, const void* xg, const void* w, const float* tb, const float* sg, const float* stats,
                                         void* out, int64_t M, int32_t N, int32_t K, int32_t relu) {
  if (!xg || !w || !tb || !sg || !stats || !out) return op_fail("vle_op_linear_ln_consumer: null operand");
  if (!gemm_ln_supports(DT_BF16, M, K) || N % 256 != 0) return op_fail("vle_op_linear_ln_consumer: shape not covered (M >= 128, K % 256 == 0, K <= 1536, N % 256 == 0)");
  GemmLn ln;
  ln.stats_in = stats; ln.stats_ld = M; ln.sg = sg;
  return op_done(launch_gemm((hipStream_t)stream, DT_BF16, xg, w, tb, out, nullptr, M, N, K, relu ? EPI_RELU_LNC : EPI_STORE_LNC, &ln), "vle_op_linear_ln_consumer");
}

extern "C" int vle_op_linear_skinny_fp8w(void* stream, const float* x, const float* gamma, const float* beta, const void* w8,
                                         const float* wscale, const float* bias, float* out, float* resid, int32_t N, int32_t K,
                                         int epilogue) {
  if (!x || !w8 || !wscale) return op_fail("vle_op_linear_skinny_fp8w: null operand");
  SkinnyArgs a;
  a.w = w8; a.wscale = wscale; a.bias = bias; a.N = N; a.K = K; a.B = 1;
  a.pro = gamma ? PRO_LN : PRO_PLAIN;
  a.x = x; a.gamma = gamma; a.beta = beta;
  a.epi = epilogue == 0 ? SEPI_STORE : epilogue == 1 ? SEPI_RELU : SEPI_RESID;
  a.out = out; a.resid = resid;
  if (a.epi == SEPI_RESID ? !resid : !out) return op_fail("vle_op_linear_skinny_fp8w: missing output");
  if (a.pro == PRO_LN && a.epi == SEPI_RESID) return op_fail("vle_op_linear_skinny_fp8w: LN + residual is not instantiated");
  const int r = launch_gemv1((hipStream_t)stream, DT_FP8W, a);
  if (r == 1) return op_fail("vle_op_linear_skinny_fp8w: shape not instantiated (K must be a multiple of 512)");
  return op_done(r, "vle_op_linear_skinny_fp8w");
}

extern "C" int vle_op_linear_fp8w(void* stream, const void* a, const void* w8, const float* wscale, const float* bias, void* out,
                                  float* resid, int64_t M, int32_t N, int32_t K, int epilogue, void* workspace, int32_t ksplit) {
  if (!a || !w8 || !wscale) return op_fail("vle_op_linear_fp8w: null operand");
  if (epilogue == EPI_RESID ? !resid : !out) return op_fail("vle_op_linear_fp8w: missing output");
  if (ksplit < 0 || ksplit > 16 || (ksplit & (ksplit - 1))) return op_fail("vle_op_linear_fp8w: ksplit must be 0, 1, 2, 4, 8 or 16");
  if (M < 1 || M > 64 || !gemm_skinny_supports((int)M, N, K, epilogue, 4)) return op_fail("vle_op_linear_fp8w: shape not covered (M <= 64, K % 256 == 0)");
  GemmSkinnyArgs g;
  g.x = a; g.w = w8; g.wscale = wscale; g.bias = bias; g.M = (int)M; g.N = N; g.K = K; g.epi = epilogue; g.out = out; g.resid = resid;
  g.workspace = workspace; g.ksplit = ksplit;
  return op_done(launch_gemm_skinny((hipStream_t)stream, g), "vle_op_linear_fp8w");
}

extern "C" int64_t vle_op_linear_workspace_bytes(void) { return (int64_t)gemm_skinny_workspace_bytes(); }

extern "C" int vle_op_linear_skinny(void* stream, int dtype, const float* x, const float* gamma, const float* beta, const void* w,
                                    const float* bias, float* out, float* resid, int32_t M, int32_t N, int32_t K, int epilogue) {
  if (!x || !w || M < 1 || M > 8) return op_fail("vle_op_linear_skinny: bad argument (M must be 1..8)");
  SkinnyArgs a;
  a.w = w; a.bias = bias; a.N = N; a.K = K; a.B = M;
  a.pro = gamma ? PRO_LN : PRO_PLAIN;
  a.x = x; a.gamma = gamma; a.beta = beta;
  a.epi = epilogue == 0 ? SEPI_STORE : epilogue == 1 ? SEPI_RELU : SEPI_RESID;
  a.out = out; a.resid = resid;
  if (a.epi == SEPI_RESID ? !resid : !out) return op_fail("vle_op_linear_skinny: missing output");
  if (a.pro == PRO_LN && a.epi == SEPI_RESID) return op_fail("vle_op_linear_skinny: LN + residual is not instantiated");
  if (M == 1) {  // batch 1 runs on the wave-autonomous GEMV when it has the shape (as the engine does)
    const int r = launch_gemv1((hipStream_t)stream, dtype, a);
    if (r <= 0) return op_done(r, "vle_op_linear_skinny");
  }
  return op_done(launch_skinny((hipStream_t)stream, dtype, a), "vle_op_linear_skinny");
}

extern "C" int vle_op_attention(void* stream, int dtype, const void* qkv, void* out, const int32_t* seq_off_dev,
                                const int32_t* text_len_dev, int32_t B, int32_t max_len, int32_t d, int32_t nhead, int causal) {
  if (!qkv || !out || !seq_off_dev || !text_len_dev || nhead < 1 || d % nhead) return op_fail("vle_op_attention: bad argument");
  return op_done(launch_attention((hipStream_t)stream, dtype, qkv, out, seq_off_dev, text_len_dev, B, max_len, d, nhead, causal),
                 "vle_op_attention");
}

extern "C" int vle_op_cross_attention(void* stream, int dtype, const void* q, const void* kv, void* out, int32_t Tq, int32_t S, int32_t d,
                                      int32_t nhead) {
  if (!q || !kv || !out || Tq < 0 || S < 1 || nhead < 1 || d % nhead) return op_fail("vle_op_cross_attention: bad argument");
  const int r = launch_cross_attention((hipStream_t)stream, dtype, q, kv, out, Tq, S, d, nhead);
  if (r == -1) return op_fail("vle_op_cross_attention: head size must be <= 128, dtype f32 or bf16");
  return op_done(r, "vle_op_cross_attention");
}

extern "C" int vle_op_decode_attention(void* stream, int dtype, const float* q, const void* k_cache, const void* v_cache,
                                       const int32_t* kv_len_dev, float* workspace, float* out, int32_t B, int32_t nhead,
                                       int32_t dh, int32_t ctx_max, int32_t nsplit) {
  if (!q || !k_cache || !v_cache || !kv_len_dev || !workspace || B < 1 || nhead < 1 || dh < 1 || ctx_max < 1)
    return op_fail("vle_op_decode_attention: bad argument");
  if (nsplit < 1 || nsplit > 16 || (nsplit & (nsplit - 1))) return op_fail("vle_op_decode_attention: nsplit must be 1, 2, 4, 8 or 16");
  const int64_t d = (int64_t)nhead * dh;
  float* part_o = workspace;
  float* part_ml = workspace + (int64_t)B * nsplit * d;
  int r = launch_decode_attention((hipStream_t)stream, dtype, q, k_cache, v_cache, kv_len_dev, part_o, part_ml, B, nhead, dh, ctx_max,
                                  nsplit);
  if (r == 0 && out) r = launch_attn_combine((hipStream_t)stream, DT_F32, part_o, part_ml, out, B, nhead, dh, nsplit);
  return op_done(r, "vle_op_decode_attention");
}

extern "C" int vle_op_attn_out_proj(void* stream, int dtype, const float* workspace, const void* w, const float* bias,
                                    float* resid, int32_t B, int32_t nhead, int32_t dh, int32_t nsplit) {
  if (!workspace || !w || !resid || B < 1 || B > 8 || nhead < 1 || dh < 1) return op_fail("vle_op_attn_out_proj: bad argument");
  if (nsplit < 1 || nsplit > 16 || (nsplit & (nsplit - 1))) return op_fail("vle_op_attn_out_proj: nsplit must be 1, 2, 4, 8 or 16");
  const int d = nhead * dh;
  SkinnyArgs a;
  a.w = w; a.bias = bias; a.N = d; a.K = d; a.B = B; a.pro = PRO_ATTN; a.epi = SEPI_RESID;
  a.part_o = workspace; a.part_ml = workspace + (int64_t)B * nsplit * d; a.nsplit = nsplit; a.nhead = nhead; a.dh = dh;
  a.resid = resid;
  if (B == 1) {
    const int r = launch_gemv1((hipStream_t)stream, dtype, a);
    if (r <= 0) return op_done(r, "vle_op_attn_out_proj");
  }
  return op_done(launch_skinny((hipStream_t)stream, dtype, a), "vle_op_attn_out_proj");
}

extern "C" int vle_op_attn_step1(void* stream, int dtype, float* x, const float* gamma, const float* beta, const void* w_in, const float* b_in,
                                 const void* w_out, const float* b_out, void* k_cache, void* v_cache, const int32_t* kv_len_dev,
                                 float* workspace, int32_t nhead, int32_t dh, int32_t ctx_max, int32_t nsplit) {
  if (!x || !gamma || !beta || !w_in || !w_out || !k_cache || !v_cache || !kv_len_dev || !workspace || nhead < 1 || dh < 1 || ctx_max < 1)
    return op_fail("vle_op_attn_step1: bad argument");
  if (!(dtype == DT_F32 || dtype == DT_BF16)) return op_fail("vle_op_attn_step1: dtype must be f32 or bf16");
  if (!(nsplit == 4 || nsplit == 8 || nsplit == 16)) return op_fail("vle_op_attn_step1: nsplit must be 4, 8 or 16");
  const int d = nhead * dh;
  if (!qkv_attn1_supports(dtype, d, nhead, dh) || (nsplit == 16 && d / 64 > 16)) return op_fail("vle_op_attn_step1: shape not covered by the fused launch");
  QkvAttnArgs q;
  q.w = w_in; q.bias = b_in; q.x = x; q.gamma = gamma; q.beta = beta;
  q.q_out = workspace; q.k_new = workspace + d; q.v_new = workspace + 2 * d;
  q.part_o = workspace + 3 * d; q.part_ml = workspace + 3 * d + (int64_t)nsplit * d;
  q.k_cache = k_cache; q.v_cache = v_cache; q.kv_len = kv_len_dev; q.d = d; q.nhead = nhead; q.dh = dh; q.ctx_max = ctx_max; q.nsplit = nsplit;
  int r = launch_qkv_attn1((hipStream_t)stream, dtype, q);
  if (r != 0) return op_done(r < 0 ? r : -1, "vle_op_attn_step1");
  SkinnyArgs a;
  a.w = w_out; a.bias = b_out; a.N = d; a.K = d; a.B = 1; a.pro = PRO_ATTN_SELF; a.epi = SEPI_RESID;
  a.part_o = q.part_o; a.part_ml = q.part_ml; a.nsplit = nsplit; a.nhead = nhead; a.dh = dh; a.resid = x;
  a.q_self = q.q_out; a.k_self = q.k_new; a.v_self = q.v_new;
  r = launch_gemv1((hipStream_t)stream, dtype, a);
  return op_done(r == 0 ? 0 : (r < 0 ? r : -1), "vle_op_attn_step1");
}

extern "C" int vle_op_token_embedding(void* stream, const int64_t* ids, const float* table, float* out, int64_t n, int32_t d) {
  if (!ids || !table || !out || n < 0 || d < 4 || d % 4) return op_fail("vle_op_token_embedding: bad argument");
  return op_done(launch_token_embedding((hipStream_t)stream, ids, table, out, n, d), "vle_op_token_embedding");
}

extern "C" int vle_op_token_embedding_add(void* stream, const int64_t* ids, const float* table, float* inout, int64_t n, int32_t d) {
  if (!ids || !table || !inout || n < 0 || d < 4 || d % 4) return op_fail("vle_op_token_embedding_add: bad argument");
  return op_done(launch_token_embedding_add((hipStream_t)stream, ids, table, inout, n, d), "vle_op_token_embedding_add");
}

extern "C" int vle_op_sine_positional(void* stream, const float* x, const float* pe, const float* alpha_dev, float x_scale, float* out,
                                      int64_t B, int32_t T, int32_t d) {
  if (!x || !pe || !alpha_dev || !out || B < 0 || T < 1 || d < 4 || d % 4) return op_fail("vle_op_sine_positional: bad argument");
  return op_done(launch_sine_positional((hipStream_t)stream, x, pe, alpha_dev, x_scale, out, B, T, d), "vle_op_sine_positional");
}

extern "C" int vle_op_adaln_fold(void* stream, const float* wb, const float* g, const float* be, float* gamma_out, float* beta_out,
                                 int32_t d) {
  if (!wb || !g || !be || !gamma_out || !beta_out || d < 1) return op_fail("vle_op_adaln_fold: bad argument");
  return op_done(launch_adaln_fold((hipStream_t)stream, wb, g, be, gamma_out, beta_out, d), "vle_op_adaln_fold");
}

extern "C" int vle_op_cross_entropy(void* stream, const float* logits, const int64_t* targets, float* loss, int32_t* hit, int64_t rows,
                                    int32_t V, int32_t ignore_index, int32_t topk) {
  if (!logits || !targets || !loss || !hit || rows < 0 || V < 1 || topk < 1) return op_fail("vle_op_cross_entropy: bad argument");
  return op_done(launch_cross_entropy((hipStream_t)stream, logits, targets, loss, hit, rows, V, ignore_index, topk), "vle_op_cross_entropy");
}

extern "C" int vle_op_topk_sample(void* stream, const float* logits, int64_t rows, int32_t V, int32_t top_k, float temperature, uint64_t seed,
                                  uint32_t step, int64_t* samples, int64_t* argmax) {
  if (!logits || !samples || rows < 0 || V < 1) return op_fail("vle_op_topk_sample: bad argument");
  if (!(temperature > 0.f)) return op_fail("vle_op_topk_sample: temperature must be positive");
  const int r = launch_topk_sample_rows((hipStream_t)stream, logits, rows, V, top_k, temperature, seed, step, samples, argmax);
  if (r == -2) return op_fail("vle_op_topk_sample: V must be <= 1280 (the audio vocabulary is 1025)");
  return op_done(r, "vle_op_topk_sample");
}

extern "C" int vle_op_quantize_rows_fp8(void* stream, const void* x_bf16, void* q_out, float* scale_out, int64_t rows, int32_t K) {
  if (!x_bf16 || !q_out || !scale_out || rows < 0 || K < 512) return op_fail("vle_op_quantize_rows_fp8: bad argument");
  const int r = launch_quantize_rows_fp8((hipStream_t)stream, x_bf16, q_out, scale_out, rows, K);
  if (r == 1) return op_fail("vle_op_quantize_rows_fp8: K must be 512 * {1,2,3,4,6,8,12,16}");
  return op_done(r, "vle_op_quantize_rows_fp8");
}

extern "C" int vle_op_linear_fp8(void* stream, const void* a8, const float* a_scale, const void* w8, const float* w_scale, const float* bias,
                                 void* out, float* resid, int64_t M, int32_t N, int32_t K, int epilogue) {
  if (!a8 || !a_scale || !w8 || !w_scale || M < 1 || N < 4 || K < 128) return op_fail("vle_op_linear_fp8: bad argument");
  if (epilogue == EPI_RESID ? resid == nullptr : out == nullptr) return op_fail("vle_op_linear_fp8: output missing");
  const int r = launch_gemm_fp8((hipStream_t)stream, a8, a_scale, w8, w_scale, bias, out, resid, M, N, K, epilogue);
  if (r == 1) return op_fail("vle_op_linear_fp8: shape not covered (K % 128, N % 4)");
  return op_done(r, "vle_op_linear_fp8");
}
