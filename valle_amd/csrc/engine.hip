// Host side of libvalle_engine.so: weight packing, buffers, the prefill / AR-step / NAR schedules,
// hipGraph capture of the AR step, and the C ABI of include/valle_engine.h.
//
// Reference being replaced: VALLE.inference() / continual() -- valle/models/valle.py:961-1238 -- and
// the modules under it (valle/modules/{transformer,activation,embedding}.py).  The reference has no
// KV cache (valle.py:1004 "TODO: Managing decoder steps avoid repetitive computation"); the cache
// here is exact because of the prefix-LM mask (valle.py:1019-1033): text rows never see audio
// columns and audio rows are causal, so the hidden state of a position never changes once computed.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "valle_engine.h"

namespace vle {

static std::mutex g_err_mu;
// live engines of this process: the kernel-selection knobs of vle_set_option that are process-wide (gs_*, g1_shared, qa_waves,
// attn_nt, ...) change what EVERY engine would capture, so they drop the captured AR-step graphs of all of them
static std::mutex g_engines_mu;
static std::vector<struct ::vle_engine*> g_engines;
static std::string g_last_error;

void set_global_error(const char* msg) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_last_error = msg;
}
int hip_fail(hipError_t e, const char* expr, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, expr);
  set_global_error(buf);
  return VLE_EHIP;
}

constexpr int NUM_TEXT_TOKENS = 512;    // valle/models/macros.py:2
constexpr int NUM_AUDIO_TOKENS = 1024;  // valle/models/macros.py:5
constexpr int V_AR = NUM_AUDIO_TOKENS + 1;
constexpr int SKINNY_MAX_B = 8;

struct LayerW {
  void *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;  // T
  float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
  float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;  // AR LayerNorm affine
  // FP8W (AR decoder only): e4m3fn codes [N][K] + one power-of-two scale per row; wqkv.. then hold bf16(W')
  void *wqkv8 = nullptr, *wo8 = nullptr, *w18 = nullptr, *w28 = nullptr;
  // fragment-major copies for gemm_skinny.hip (AR decoder of an engine with max_batch >= 2): of the bf16 weights, or of
  // the fp8 codes in FP8W mode
  void *wqkv_p = nullptr, *wo_p = nullptr, *w1_p = nullptr, *w2_p = nullptr;
  float *sqkv = nullptr, *so = nullptr, *s1 = nullptr, *s2 = nullptr;
  // fused LayerNorm of the batched step (kernels.h LnConsumer): wg = W gamma, wb = W beta + bias, fp32, of the effective weights
  float *wg_qkv = nullptr, *wb_qkv = nullptr, *wg_1 = nullptr, *wb_1 = nullptr;
};

}  // namespace vle

using namespace vle;

struct vle_engine {
  vle_config cfg{};
  int d = 0, H = 0, dh = 0, L = 0, Q = 0, bos = 0, dtype = 0;
  int max_B = 0, max_S = 0, max_P = 0, max_G = 0, ctx_max = 0, max_pos = 0;
  int64_t max_rows = 0;  // packed rows of the biggest pass
  std::string err;
  hipStream_t st = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<void*> allocs;      // weights and everything that lives as long as the engine
  std::vector<void*> buf_allocs;  // capacity-dependent buffers (KV cache, activations, traces): vle_reserve frees and re-creates them
  bool in_buffers = false;        // dev_alloc target
  // Small allocations (biases, LayerNorm affines, folded row constants, granule buffers, operand tables, state words) are carved out of
  // a few large chunks instead of one hipMalloc -- i.e. one 4 KB page and one small page-table fragment -- each (dev_alloc below).
  struct Arena { char* cur = nullptr; size_t left = 0; } arena_w, arena_b;
  bool finalized = false;
  bool broken = false;            // vle_reserve failed half-way: buffers are gone, every entry point refuses (VLE_ESTATE)

  // ---- host staging of the state dict --------------------------------------------------------
  std::map<std::string, std::vector<float>> host_w;
  std::map<std::string, std::vector<int64_t>> host_shape;

  // ---- device weights ---------------------------------------------------------------------------
  float *ar_text_emb = nullptr, *nar_text_emb = nullptr, *ar_audio_emb = nullptr;
  float* nar_audio_emb[8] = {};
  const float** nar_audio_emb_tab = nullptr;  // device array of the Q table pointers
  float* alphas = nullptr;                    // [4]: ar_text, ar_audio, nar_text, nar_audio
  float* pe = nullptr;
  std::vector<LayerW> ar, nar;
  float *ar_norm_g = nullptr, *ar_norm_b = nullptr;
  void* ar_predict = nullptr;
  void* ar_predict8 = nullptr;     // FP8W
  void* ar_predict_p = nullptr;    // fragment-major copy (see LayerW)
  bool opt_gs_wpack = true;        // option "gs_wpack": gemm_skinny reads the fragment-major weight copies
  float* ar_predict_s = nullptr;
  bool w8 = false;                 // dtype_mode == VLE_DTYPE_FP8W / FP8: dtype stays DT_BF16 for activations / KV / MFMA passes
  bool a8 = false;                 // dtype_mode == VLE_DTYPE_FP8: the packed passes run gemm_fp8.hip on per-row-quantised activations
  unsigned char* A8 = nullptr;     // [max_rows][4 d] e4m3fn codes of the current GEMM's activations
  float* a8_scale = nullptr;       // [max_rows]
  bool opt_fp8_gemm = true;        // option "fp8_gemm": 0 = FP8 mode runs the bf16 kernels on bf16(W') like FP8W (A/B)
  void* nar_predict[7] = {};
  // folded AdaLN affine, [stage][site] with site = 2*l (norm1), 2*l+1 (norm2), 2*L (final)
  std::vector<std::vector<float*>> nar_gamma, nar_beta;

  // ---- buffers ------------------------------------------------------------------------------------
  void *kcache = nullptr, *vcache = nullptr;  // T [L][B][H][ctx_max][dh]
  float *x_step = nullptr, *q_step = nullptr, *h_step = nullptr, *part_o = nullptr, *part_ml = nullptr, *logits = nullptr;
  unsigned long long* qgran = nullptr;       // [L][d] {epoch, q} granules of the fused launch's in-launch hand-off (zeroed at every prefill)
  unsigned* qa_spin_fail = nullptr;          // workgroups that gave up waiting and recomputed their query rows (diagnostic)
  float *k_new = nullptr, *v_new = nullptr;  // [B][d] the new token's K / V rows (cache-rounded) of the fused batch-1 QKV + attention launch
  void *xn_step = nullptr, *qkv_step = nullptr, *att_step = nullptr, *hT_step = nullptr;  // batch > 8 path
  int32_t* state_dev = nullptr;  // kv_len, audio_pos, n_gen, done, cap, iter [max_B] each, then done_count
  ArState S{};
  ArDyn* dyn_dev = nullptr;
  int64_t *tokens = nullptr, *sampled = nullptr;  // [max_B][max_G]
  int64_t *text_ids = nullptr, *prompt_codes = nullptr;  // [max_B][max_S], [max_B][max_P + max_G][Q]
  int32_t* forced_len_dev = nullptr;
  unsigned long long* slot_seed_dev = nullptr;  // [max_B] RNG seed of the request in each slot (slot API)
  unsigned long long admitted = 0;              // requests admitted since vle_slots_begin
  const int64_t* nar_forced = nullptr; int64_t nar_forced_stride = 0;  // vle_nar_force: consumed by the next NAR call
  int32_t* id_err_dev = nullptr;  // token-id range flag (1 text, 2 prompt / continuation codes, 4 forced tokens): VLE_EINDEX
  hipEvent_t ev_chk = nullptr;
  float *X = nullptr, *yemb = nullptr, *nar_logits = nullptr;
  // LayerNorm folded into the packed-row GEMMs of the prefill / NAR passes (kernels.h GemmLn; option "ln_fold", default 1):
  float* ln_rows_stats = nullptr;  // [d / 64][max_rows][2] (mean, M2) of every 64-column group of the residual rows, group-major (buffer)
  float* nar_fold = nullptr;       // [Q - 1][L][14 d]: per (stage, layer) sg / tb of the in-projection (3 d each) and of linear1 (4 d each)
  bool opt_ln_fold = true;         // measured (round 5): one utterance's NAR 8.37 -> 8.24 ms, 64 utterances' kernel time - 2 %
  void *Xn = nullptr, *QKV = nullptr, *ATT = nullptr, *Hb = nullptr;
  int32_t* tables_dev = nullptr;  // row tables
  int32_t* tables_host = nullptr; // pinned mirror
  int64_t tables_cap = 0;
  int32_t* poll_host = nullptr;   // pinned [64]
  int32_t* prog_host = nullptr;   // pinned + mapped [16]: progress words the sample kernel writes (ArSampleArgs::host_prog)
  int32_t* prog_dev = nullptr;    // device address of prog_host
  bool opt_host_prog = true;      // option "host_prog": 0 = poll through a D2H copy + event per graph replay (round-1 behaviour)
  float* trace_ar = nullptr; int64_t trace_ar_cap = 0;
  float* trace_nar = nullptr;     // [Q-1][sumG_max][1024]
  bool opt_trace_ar = false, opt_trace_nar = false;
  // per-kernel timing of the AR step with hipEvents on the engine stream (option "profile_kernels"):
  // forces eager launches; tags: 0 qkv, 1 decode-attention, 2 out-proj, 3 ffn1, 4 ffn2, 5 logits, 6 sample
  bool opt_profile = false;
  // option "ktrace": the batch-1 step kernels stamp the wall clock into ktrace_buf (common.h KTrace); kt_idx counts launches of a step
  unsigned long long* ktrace_buf = nullptr;
  bool opt_ktrace = false;
  int kt_idx = 0;
  KTrace next_kt() {
    KTrace k;
    if (opt_ktrace && ktrace_buf) {
      k.buf = ktrace_buf; k.step = S.iter; k.idx = kt_idx++;
    }
    return k;
  }
  bool opt_no_gemm_skinny = false;  // option "no_gemm_skinny": batch 2..64 on the v0 kernels (A/B measurements)
  int opt_w8_temporal = -1;         // option "w8_temporal": FP8W batch-1 GEMV loads its weights with the default cache policy
                                    // (-1 = when the fp8 AR weights fit the 256 MB memory-side cache with room for the KV stream)
  bool opt_gs_xf = true;            // option "gs_xf": AR-step activations of the gemm_skinny path in the fragment-major layout
  int opt_gs_target = 0;            // option "gs_target_wgs": workgroups the split-K of gemm_skinny aims for (0 = 256)
  void* gs_ws = nullptr;            // split-K tickets + partial tiles of gemm_skinny (zeroed once)
  int opt_gs_dbg = 0;               // option "gs_dbg": timing diagnostics of gemm_skinny (1 = no X loads, 2 = no W loads)
  int opt_gs_rot = 0;               // option "gs_rot": gemm_skinny workgroups walk X in rotated orders (A/B knob)
  bool opt_gs_fuse_ln = true;       // option "gs_fuse_ln": LayerNorm folded into the gemm_skinny launches (no LayerNorm kernels in the step)
  float* ln_stats = nullptr;        // [64][d / 16][2] group statistics of the residual rows (kernels.h LnProducer)
  float* ao_part = nullptr;         // [B][H][d] partial out-proj sums of the fused attention + out-proj launch (decode_attn.hip AttnOproj)
  int* ao_cnt = nullptr;            // [B] its tickets (zeroed once, self-resetting)
  bool opt_attn_oproj = false;      // option "attn_oproj": batched step, out-proj folded into the decode-attention launch
  float *wg_pred = nullptr, *wb_pred = nullptr;  // final LayerNorm + ar_predict_layer
  bool opt_ignore_eos = false;  // option "ignore_eos": synthetic-weight benchmarks run every utterance to the length cap
  bool opt_no_gemv1 = false;  // option "no_gemv1": force the generic skinny kernel at batch 1 (A/B measurements)
  int opt_nsplit = 0;         // option "nsplit": 0 = chosen per batch
  int opt_nk = 0;             // option "attn_nk": keys per lane per round of the decode attention (0 auto, 4, 8)
  int opt_spg = 0;            // option "steps_per_graph": overrides cfg.steps_per_graph when > 0
  int opt_qkv_attn = 1;       // option "qkv_attn": batch 1, QKV GEMV + decode attention in one launch (gemv1.hip qkv_attn1_kernel)
  int opt_qa_nsplit = 8;      // option "qa_nsplit": KV splits per head of that launch (4, 8, 16).  With the q hand-off (no redundant query rows) 8
                              // splits cover 1024 keys in one round: 212.4 us per step over the whole 753-step run against 215.3 with 4 (which wins
                              // below context 512: 208.1 vs 212.3); without the hand-off: 215.9 / 218.8 / 263 us for 4 / 8 / 16
  int opt_qa_nk = 4;          // option "qa_nk": keys per lane per round of the fused launch's attention workgroups (4 / 8)
  int opt_qa_handoff = 1;     // option "qa_handoff": q reaches the attention workgroups through granules instead of being recomputed per split
  int opt_qa_qtemporal = 1;   // option "qa_qtemporal": its query-row loads with the default cache policy (shared by a head's splits through L2)
  // ---- the batch-1 step as one persistent launch (persist.hip; option "persist") ----
  int opt_persist = 1;        // option "persist": 1 = batch-1 AR steps run pstep_kernel (+ the sampling launch) where the shape is covered
  int opt_persist_batch = 1;  // option "persist_batch": 1 = calls of 2 .. PSB_MAX utterances run pstepb_kernel (persist_nb.hip) where the shape is covered
  int ps_gran_B = 1;          // utterances the granule buffer was sized for
  int ps_table_B = 0;         // batch the operand table was built for (the caches' layer stride depends on it)
  int32_t* ps_epoch = nullptr; // [1] device: epoch counter of the batched launch in slot mode (PStepArgs::epoch_ctr)
  bool ps_fold_valid = false; // ps_fold holds the row constants (they depend on the weights only: not recomputed when the table is rebuilt for another batch)
  mutable int psb_form_key = -1, psb_form_res = 0;  // pstepb_form_ok(B), cached
  int opt_ps_nk = 2, opt_ps_pf = 3;  // options "persist_nk", "persist_pf" (PStepArgs)
  int opt_ps_naps = -1;              // option "persist_naps" (-1: the engine mode's own timing, ps_naps_of)
  bool ps_device_ok = false;  // the device has the 256 CUs the persistent grid needs
  mutable int ps_form_key = -1, ps_form_res = 0;  // ps_form_ok(): the last (mode, schedule, weight type) asked about and pstep_form_ok's answer
  PLayer* ps_table = nullptr;               // device [L] operand table (rebuilt when the KV cache moves)
  PStepSample* ps_sample = nullptr;         // device copy of the in-launch sampling step's operands (PStepArgs::smp)
  unsigned char* ps_host = nullptr;         // pinned staging of both: they reach the device by stream-ordered copies on the engine's stream
  size_t ps_host_bytes = 0;
  float* ps_fold = nullptr;                 // [L][14 d] + [2][V_AR + 3]: row constants of the folded LayerNorm (launch_ps_fold)
  unsigned long long* ps_gran = nullptr;    // {epoch, value} granules of the step's edges (zeroed at every prefill)
  size_t ps_gran_n = 0;
  unsigned long long* ps_ptrace = nullptr;  // [8][256][PS_PT_SLOTS] in-kernel timeline (option "persist_trace")
  bool opt_ps_trace = false;
  int opt_ps_mode = PS_MODE_DEFAULT;  // option "persist_mode": PStepArgs::mode
  bool opt_ps_sample = true;          // option "persist_sample": the sampling step inside the persistent launch, several AR iterations
                                      // per launch (0: one step per launch + the sampling kernel, as the launch chain does)
  int opt_ps_steps = 32;              // option "persist_steps": AR iterations per persistent launch ("steps_per_graph" > 0 overrides):
                                      // 146.7 us per step at 8, 145.6 at 32 (a launch's first step starts with cold operands)
  int opt_act_bf16 = 2;       // option "act_bf16" (bf16 engines only): batch-1 chain, 1 = merged attention row, 2 = FFN hidden row rounded to
                              // bf16 -- what the persistent step's packed edges carry (persist_mode bits 8 / 4), so that the chain (profiling,
                              // slot mode, shapes the persistent step lacks) and the persistent step compute the same numbers
  const void* ps_table_kc = nullptr; int ps_table_ctx = 0;  // what the table was built for
  PStepSample ps_sample_sent{}; bool ps_sample_valid = false;  // what ps_sample holds (re-uploaded only when it changes)
  // A persistent launch that could not keep the whole GPU (a wave gave up waiting, PStepArgs::fail) ends the call with VLE_EBUSY;
  // the engine then stays on the launch chain for `ps_backoff` batch-1 prefills (2, 4, ... 64: doubling while it keeps happening,
  // back to 2 after a clean persistent decode) and re-arms the persistent launch by itself.
  int ps_backoff = 0, ps_backoff_next = 2;
  unsigned ps_fallbacks = 0;   // calls that ended with VLE_EBUSY since vle_create (debug item "persist_fallbacks")
  unsigned ps_last_fail = 0;   // give-up count of the last call that ran the persistent launch (debug item "persist_fail")
  bool ps_last_call = false;   // the last vle_ar_generate ran the persistent launch (debug item "persist_ran")
  int dbg_inject_psfail = 0;   // option "persist_inject_fail": the next n persistent calls get a non-zero give-up counter (tests)
  int opt_rpw = 0;            // option "gemv1_rpw": rows per wave override of gemv1 (tuning)
  int opt_rpw_qkv = 0;        // option "gemv1_rpw_qkv": the same for the QKV GEMV only
  int opt_rpw_ffn1 = 0;       // option "gemv1_rpw_ffn1": ... for the FFN1 GEMV only
  std::vector<hipEvent_t> prof_pool;
  size_t prof_used = 0;
  std::vector<int> prof_tags;
  double prof_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double prof_calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- per-call state -------------------------------------------------------------------------------
  int B = 0;
  int nseq = 0;            // sequences of the packed pass being enqueued when it differs from B (slot API); 0 = B
  bool slot_mode = false;  // vle_slots_*: B = max_B slots, each free (done = 1) or holding one utterance
  std::vector<int32_t> slot_state;  // host copy of the device AR state after the last vle_slots_step
  bool have_prefill = false, have_gen = false;
  std::vector<int32_t> S_len, P_len, G_len;
  int64_t sumG_last = 0;
  int nsplit = 1;
  std::map<int, std::pair<hipGraphExec_t, hipGraphExec_t>> graphs;  // B -> (multi, single)
  double t_prefill = 0, t_ar = 0, t_nar = 0, n_steps = 0;
  int n_ar_launches = 0;  // AR-loop submissions of the last vle_ar_generate (graph replays / eager steps), debug item "ar_launches"

  int fail(int code, const std::string& m) {
    err = m;
    return code;
  }
};

namespace {

#define E_HIP(e, expr)                                                                           \
  do {                                                                                           \
    hipError_t _r = (expr);                                                                      \
    if (_r != hipSuccess) {                                                                      \
      char _b[512];                                                                              \
      snprintf(_b, sizeof(_b), "HIP error %d (%s) at %s:%d: %s", (int)_r, hipGetErrorString(_r), __FILE__, __LINE__, #expr); \
      (void)hipGetLastError(); /* reported through the C ABI: do not leave it for the caller's next runtime check (torch) */ \
      return (e)->fail(VLE_EHIP, _b);                                                            \
    }                                                                                            \
  } while (0)

#define E_LAUNCH(e, expr)                                                                        \
  do {                                                                                           \
    int _r = (expr);                                                                             \
    if (_r != 0) {                                                                               \
      char _b[512];                                                                              \
      snprintf(_b, sizeof(_b), "kernel launch rejected (%d) at %s:%d: %s", _r, __FILE__, __LINE__, #expr); \
      return (e)->fail(_r == -3 ? VLE_EHIP : VLE_EINVAL, _b);                                    \
    }                                                                                            \
  } while (0)

// Debugging aid (environment VLE_GUARD_ALLOC=1): every device allocation of the engine gets its own virtual-memory mapping and ENDS at
// the end of it, with an unmapped granule behind -- a kernel that reads or writes past the end of any engine buffer faults at once and
// deterministically instead of once in twenty processes (hipMalloc sub-allocates: an overrun usually lands in somebody's mapped
// memory).  Each mapping is listed on stderr so the faulting address names its buffer.  VLE_GUARD_ALLOC=2 puts the buffer at the
// START of its mapping behind an unmapped granule instead (under-runs).  Nothing is freed in this mode.
static bool guard_alloc_enabled() {
  static const int on = [] { const char* v = getenv("VLE_GUARD_ALLOC"); return v ? atoi(v) : 0; }();
  return on != 0;
}
static int guard_alloc(int device, void** out, size_t bytes, const char* tag, int mode_override = 0) {
  static const int env_mode = [] { const char* v = getenv("VLE_GUARD_ALLOC"); return v ? atoi(v) : 0; }();
  const int mode = mode_override ? mode_override : env_mode;
  static std::atomic<int> seq{0};
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return -1;
  const size_t mapped = (bytes + gran - 1) / gran * gran;
  void* va = nullptr;
  if (hipMemAddressReserve(&va, mapped + 2 * gran, gran, nullptr, 0) != hipSuccess) return -1;
  char* base = (char*)va + gran;  // one unmapped granule on either side
  hipMemGenericAllocationHandle_t h;
  if (hipMemCreate(&h, mapped, &prop, 0) != hipSuccess) return -1;
  if (hipMemMap(base, mapped, 0, h, 0) != hipSuccess) return -1;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(base, mapped, &acc, 1) != hipSuccess) return -1;
  const size_t off = mode == 2 ? 0 : ((mapped - bytes) & ~(size_t)(mode_override ? 15 : 255));  // caller buffers: 16-byte aligned like torch's
  *out = base + off;
  fprintf(stderr, "[guard] #%d %s %zu bytes at %p .. %p (mapping %p .. %p)\n", seq.fetch_add(1), tag, bytes, (void*)(base + off), (void*)(base + off + bytes),
          (void*)base, (void*)(base + mapped));
  fflush(stderr);
  return 0;
}

// Two more debugging aids for ordinary hipMalloc allocations.  VLE_ALLOC_LOG=1: every engine allocation is listed on stderr
// (sequence number, kind, bytes, address range) so that the address of a GPU memory-access fault names its buffer.
// VLE_POISON_ALLOC=<byte>: every engine allocation is filled with that byte before the engine initialises what it means to
// initialise -- a fresh box hands out memory that holds whatever the last tenant left (the driver wipes on release, so a
// second process on the same box sees zeros): a read of something the engine forgot to write then behaves the same in every
// process instead of once per box (0xff: NaN floats, -1 integers, non-canonical pointers; 0x7f: huge finite floats / integers).
static void debug_alloc_note(void* q, size_t bytes, const char* tag) {
  static const int log_on = [] { const char* v = getenv("VLE_ALLOC_LOG"); return v ? atoi(v) : 0; }();
  static const int poison = [] { const char* v = getenv("VLE_POISON_ALLOC"); return v ? (int)strtol(v, nullptr, 0) : -1; }();
  // VLE_POISON_RANGE=a:b restricts the poison to allocations number a .. b - 1 of the process (bisecting WHICH buffer a NaN came from)
  static const std::pair<int, int> range = [] {
    const char* v = getenv("VLE_POISON_RANGE");
    int a = 0, b = 1 << 30;
    if (v) sscanf(v, "%d:%d", &a, &b);
    return std::make_pair(a, b);
  }();
  static std::atomic<int> seq{0};
  const int n = seq.fetch_add(1);
  if (poison >= 0 && n >= range.first && n < range.second) {
    // synchronous: an allocation made inside a call (the logit trace of vle_ar_generate) is written by the engine's stream right away,
    // and a fill still in flight on the null stream would land on top of those writes (seen: NaN "logits" that were never computed)
    (void)hipMemset(q, poison & 0xff, bytes);
    (void)hipDeviceSynchronize();
  }
  if (log_on) {
    fprintf(stderr, "[alloc] #%d %s %zu bytes at %p .. %p\n", n, tag, bytes, q, (void*)((char*)q + bytes));
    fflush(stderr);
  }
}

static void debug_host_note(void* q, size_t bytes, const char* tag) {
  static const int log_on = [] { const char* v = getenv("VLE_ALLOC_LOG"); return v ? atoi(v) : 0; }();
  if (log_on) {
    fprintf(stderr, "[alloc] host %s %zu bytes at %p .. %p\n", tag, bytes, q, (void*)((char*)q + bytes));
    fflush(stderr);
  }
}

template <typename T>
int dev_alloc(vle_engine* e, T** p, size_t count) {
  void* q = nullptr;
  if (guard_alloc_enabled()) {
    const int gr = guard_alloc(e->cfg.device, &q, std::max<size_t>(count, 1) * sizeof(T), e->in_buffers ? "buffer" : "weight");
    if (gr != 0) return e->fail(VLE_EHIP, "guard allocation failed (VLE_GUARD_ALLOC)");
    *p = (T*)q;  // never freed: a debugging mode
    return 0;
  }
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  // Arena (round 6; VERDICT r5 next #3a): every allocation below 1 MB comes out of 8 MB chunks at 256-byte alignment.  The batch-1
  // persistent step touches ~120 small per-layer vectors per iteration; as separate hipMallocs each sat in a page of its own (the
  // guarded-mapping runs of round 5 -- every buffer in its own mapping -- priced address translation at ~20 % of that step).  The chunks
  // are ordinary entries of allocs / buf_allocs, so vle_reserve and vle_destroy free them as before.  VLE_ARENA=0: one hipMalloc each (A/B).
  static const bool arena_on = [] { const char* v = getenv("VLE_ARENA"); return v == nullptr || atoi(v) != 0; }();
  constexpr size_t ARENA_MAX = (size_t)1 << 20, ARENA_CHUNK = (size_t)8 << 20, ARENA_ALIGN = 256;
  if (arena_on && bytes < ARENA_MAX) {
    vle_engine::Arena& ar = e->in_buffers ? e->arena_b : e->arena_w;
    const size_t need = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
    if (ar.left < need) {
      E_HIP(e, hipMalloc(&q, ARENA_CHUNK));
      (e->in_buffers ? e->buf_allocs : e->allocs).push_back(q);
      ar.cur = (char*)q;
      ar.left = ARENA_CHUNK;
    }
    q = ar.cur;
    ar.cur += need;
    ar.left -= need;
    *p = (T*)q;
    debug_alloc_note(q, bytes, e->in_buffers ? "buffer (arena)" : "weight (arena)");
    return 0;
  }
  E_HIP(e, hipMalloc(&q, bytes));
  (e->in_buffers ? e->buf_allocs : e->allocs).push_back(q);
  *p = (T*)q;
  debug_alloc_note(q, bytes, e->in_buffers ? "buffer" : "weight");
  return 0;
}

int upload_f32(vle_engine* e, float** dst, const float* src, size_t n) {
  int r = dev_alloc(e, dst, n);
  if (r) return r;
  E_HIP(e, hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

// Host-side weight preparation (quantisation, bf16 conversion, W gamma / W beta sums) is row-parallel: up to 16 threads over
// contiguous row ranges -- loading the 1.6 B parameters of BASELINE configs[4] in engine mode FP8 is otherwise a minute of
// single-threaded loops.  Results do not depend on the thread count (every row is computed by one thread, in the same order).
template <typename F>
void parallel_rows(int64_t n, int64_t work_per_row, F f) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 16u), n);
  if (nt <= 1 || n * work_per_row < (int64_t)1 << 18) {
    f((int64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  const int64_t per = (n + nt - 1) / nt;
  int64_t done_to = 0;  // rows [0, done_to) are owned by started threads
  for (int t = 0; t < nt; ++t) {
    const int64_t a = t * per, b = std::min<int64_t>(n, a + per);
    if (a >= b) break;
    try {
      th.emplace_back([=]() { f(a, b); });
    } catch (const std::system_error&) {  // no more threads (container limits): nothing may escape through the C ABI
      break;
    }
    done_to = b;
  }
  if (done_to < n) f(done_to, n);  // the rest on this thread
  for (auto& t : th) t.join();
}

// FP8W weight format (common.h): per row, scale = the smallest power of two with max|w| / scale <= 448,
// q = RNE_e4m3fn(w / scale); W' = q * scale is what every kernel of the mode computes with.
void quantize_rows_fp8w_range(const float* w, int64_t n0, int64_t n1, int64_t K, uint8_t* q, float* scale, float* deq);
void quantize_rows_fp8w(const float* w, int64_t N, int64_t K, uint8_t* q, float* scale, float* deq) {
  parallel_rows(N, K, [=](int64_t a, int64_t b) { quantize_rows_fp8w_range(w, a, b, K, q, scale, deq); });
}
void quantize_rows_fp8w_range(const float* w, int64_t n0, int64_t n1, int64_t K, uint8_t* q, float* scale, float* deq) {
  for (int64_t n = n0; n < n1; ++n) {
    const float* row = w + n * K;
    float amax = 0.f;
    for (int64_t k = 0; k < K; ++k) amax = std::max(amax, std::fabs(row[k]));
    float sc = 1.f;
    if (amax > 0.f && std::isfinite(amax)) {
      int ex = 0;
      const float m = std::frexp(amax / 448.0f, &ex);  // amax / 448 = m * 2^ex, m in [0.5, 1)
      sc = std::ldexp(1.0f, m == 0.5f ? ex - 1 : ex);
    }
    scale[n] = sc;
    const float inv = 1.0f / sc;  // exact
    for (int64_t k = 0; k < K; ++k) {
      const uint8_t c = f32_to_e4m3fn(row[k] * inv);
      if (q) q[n * K + k] = c;
      if (deq) deq[n * K + k] = e4m3fn_to_f32(c) * sc;
    }
  }
}

// FP8W upload of one Linear weight [N][K]: bf16(W') -> *dst (prefill / NAR / fallback kernels); when q8 != null also
// the e4m3fn codes and the row scales for the weight-streaming AR step
int upload_fp8w(vle_engine* e, void** dst, void** q8, float** sc, const float* src, int64_t N, int64_t K) {
  std::vector<uint8_t> q((size_t)(N * K));
  std::vector<float> scale((size_t)N), deq((size_t)(N * K));
  quantize_rows_fp8w(src, N, K, q.data(), scale.data(), deq.data());
  std::vector<uint16_t> tmp((size_t)(N * K));
  {
    uint16_t* tp = tmp.data();
    const float* dp = deq.data();
    parallel_rows(N, K, [=](int64_t a, int64_t b) {
      for (int64_t i = a * K; i < b * K; ++i) tp[i] = f32_to_bf16(dp[i]);  // exact: 4 significant bits * 2^e
    });
  }
  uint16_t* p = nullptr;
  int r = dev_alloc(e, &p, tmp.size());
  if (r) return r;
  E_HIP(e, hipMemcpy(p, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  *dst = p;
  if (q8) {
    uint8_t* pq = nullptr;
    if ((r = dev_alloc(e, &pq, q.size()))) return r;
    E_HIP(e, hipMemcpy(pq, q.data(), q.size(), hipMemcpyHostToDevice));
    *q8 = pq;
    if ((r = upload_f32(e, sc, scale.data(), scale.size()))) return r;
  }
  return 0;
}

// wg[n] = sum_k Weff[n][k] * gamma[k], wb[n] = sum_k Weff[n][k] * beta[k] + bias[n] for the fused LayerNorm of the batched
// step; Weff = the values the GEMM multiplies with (bf16-rounded weights, or W' in FP8W mode)
int upload_wg_wb(vle_engine* e, const float* w, int64_t N, int64_t K, const float* gamma, const float* beta, const float* bias,
                 float** wg_dev, float** wb_dev) {
  std::vector<float> deq;
  if (e->w8) {
    deq.resize((size_t)(N * K));
    std::vector<float> sc((size_t)N);
    quantize_rows_fp8w(w, N, K, nullptr, sc.data(), deq.data());
    w = deq.data();
  }
  std::vector<float> wg((size_t)N), wb((size_t)N);
  {
    float *wgp = wg.data(), *wbp = wb.data();
    const bool w8 = e->w8;
    parallel_rows(N, K, [=](int64_t a, int64_t b) {
      for (int64_t n = a; n < b; ++n) {
        const float* row = w + n * K;
        double ag = 0.0, ab = 0.0;
        for (int64_t k = 0; k < K; ++k) {
          const double wv = w8 ? (double)row[k] : (double)bf16_to_f32(f32_to_bf16(row[k]));
          ag += wv * (double)gamma[k];
          ab += wv * (double)beta[k];
        }
        wgp[n] = (float)ag;
        wbp[n] = (float)(ab + (bias ? (double)bias[n] : 0.0));
      }
    });
  }
  int r = upload_f32(e, wg_dev, wg.data(), wg.size());
  if (r) return r;
  return upload_f32(e, wb_dev, wb.data(), wb.size());
}

// fp32 host tensor -> compute dtype on the device
int upload_T(vle_engine* e, void** dst, const float* src, size_t n) {
  if (e->dtype == DT_F32) return upload_f32(e, (float**)dst, src, n);
  std::vector<uint16_t> tmp(n);
  {
    uint16_t* tp = tmp.data();
    const int64_t blk = 4096, nb = ((int64_t)n + blk - 1) / blk;
    parallel_rows(nb, blk, [=](int64_t a, int64_t b) {
      for (int64_t i = a * blk; i < std::min<int64_t>((int64_t)n, b * blk); ++i) tp[i] = f32_to_bf16(src[i]);
    });
  }
  uint16_t* p = nullptr;
  int r = dev_alloc(e, &p, n);
  if (r) return r;
  E_HIP(e, hipMemcpy(p, tmp.data(), n * 2, hipMemcpyHostToDevice));
  *dst = p;
  return 0;
}

const std::vector<float>* find_w(vle_engine* e, const std::string& key, std::initializer_list<int64_t> shape) {
  auto it = e->host_w.find(key);
  if (it == e->host_w.end()) {
    e->err = "missing state_dict key: " + key;
    return nullptr;
  }
  const auto& sh = e->host_shape[key];
  if (sh.size() != shape.size() || !std::equal(sh.begin(), sh.end(), shape.begin())) {
    e->err = "shape mismatch for state_dict key: " + key;
    return nullptr;
  }
  return &it->second;
}

// SinePositionalEmbedding.extend_pe, valle/modules/embedding.py:75-91 (fallback when the host
// wrapper did not pass the torch-built table as "position.pe")
void build_pe(std::vector<float>& pe, int max_pos, int d) {
  pe.assign((size_t)max_pos * d, 0.f);
  for (int i = 0; i < d; i += 2) {
    const float div = (float)std::exp((double)((float)i * (float)(-(std::log(10000.0) / d))));
    for (int p = 0; p < max_pos; ++p) {
      const float ang = (float)p * div;
      pe[(size_t)p * d + i] = (float)std::sin((double)ang);
      if (i + 1 < d) pe[(size_t)p * d + i + 1] = (float)std::cos((double)ang);
    }
  }
}

int choose_nsplit(const vle_engine* e, int B) {
  // spread the KV stream of a small batch over >= ~128-256 blocks, but never over more blocks than
  // the context has key chunks (decode_attn.hip: a block owns fixed chunks of 16 wave-loads)
  // Batched bf16 step: never split -- one block per (utterance, head) writes the normalised row itself, a split needs the combine launch
  // behind it, and the step is launch-count-bound at small batches.  Measured (round 6, profiles/r06_small_batch.json): 2 utterances
  // 368.7 -> 328.9 us per step, 3 utterances 378.1 -> 330.0 (two splits were chosen below 64 (utterance, head) blocks).
  if (B >= 2 && e->dtype == DT_BF16 && !e->opt_no_gemm_skinny) return 1;
  int ns = 1;
  while (ns < 16 && (int64_t)B * e->H * ns < 64) ns *= 2;  // measured at B=1, H=16: 4 splits best (op_chain_bench)
  const int vec = e->dtype == DT_F32 ? 4 : 8;
  int lpk = 1;
  if (e->dh % vec == 0) while (lpk * vec < e->dh) lpk *= 2;
  else while (lpk < e->dh) lpk *= 2;
  const int chunk = 16 * (64 / lpk);
  const int chunks = (e->ctx_max + chunk - 1) / chunk;
  int cap = 1;
  while (cap * 2 <= chunks) cap *= 2;
  return std::min(ns, cap);
}

// batch 1: the wave-autonomous GEMV (gemv1.hip) when it has the shape, else the generic skinny kernel
int launch_ar_linear(vle_engine* e, const SkinnyArgs& a) {
  if (a.B == 1 && !e->opt_no_gemv1) {
    SkinnyArgs t = a;
    t.rpw_override = (a.epi == SEPI_QKV && e->opt_rpw_qkv > 0) ? e->opt_rpw_qkv : (a.epi == SEPI_RELU && e->opt_rpw_ffn1 > 0) ? e->opt_rpw_ffn1 : e->opt_rpw;
    t.kt = e->next_kt();
    if (e->w8 && a.w8 != nullptr) {  // FP8W: stream the e4m3fn codes; shapes gemv1 lacks fall through to bf16(W')
      t.w = a.w8;
      // measured at C2 (152 MB of codes): AR loop 173.4 -> 166-171 ms with cacheable loads; non-temporal otherwise
      const int64_t w8_bytes = (int64_t)e->L * 12 * e->d * e->d + (int64_t)V_AR * e->d;
      t.temporal = e->opt_w8_temporal >= 0 ? e->opt_w8_temporal : (w8_bytes <= (int64_t)192 << 20 ? 1 : 0);
      const int r8 = launch_gemv1(e->st, DT_FP8W, t);
      if (r8 <= 0) return r8;
      t.w = a.w;
    }
    const int r = launch_gemv1(e->st, e->dtype, t);
    if (r <= 0) return r;
  }
  return launch_skinny(e->st, e->dtype, a);
}

}  // namespace

// =================================================================================================
// create / destroy / load
// =================================================================================================
extern "C" int vle_create(const vle_config* c, vle_engine** out) {
  if (!c || !out) {
    set_global_error("vle_create: null argument");
    return VLE_EINVAL;
  }
  auto bad = [&](const char* m) {
    set_global_error(m);
    return VLE_EINVAL;
  };
  if (c->norm_first != 1 || c->add_prenet != 0) return bad("only norm_first=1, add_prenet=0 run natively");
  if (c->d_model <= 0 || c->nhead <= 0 || c->d_model % c->nhead) return bad("bad d_model / nhead");
  if (c->d_model % 32 != 0) return bad("d_model must be a multiple of 32");
  const int dh = c->d_model / c->nhead;
  if (!(dh == 4 || dh == 8 || dh == 16 || dh == 32 || dh == 64 || dh == 96 || dh == 128)) return bad("unsupported head size");
  if (c->dtype_mode == VLE_DTYPE_BF16 && c->d_model % 64 != 0) return bad("bf16 mode needs d_model % 64 == 0");
  if (c->num_quantizers < 1 || c->num_quantizers > 8) return bad("num_quantizers must be 1..8");
  if (!(c->prefix_mode == 0 || c->prefix_mode == 1 || c->prefix_mode == 2 || c->prefix_mode == 4)) return bad("bad prefix_mode");
  if (c->dtype_mode != VLE_DTYPE_F32 && c->dtype_mode != VLE_DTYPE_BF16 && c->dtype_mode != VLE_DTYPE_FP8W && c->dtype_mode != VLE_DTYPE_FP8)
    return bad("bad dtype_mode");
  if (c->dtype_mode == VLE_DTYPE_FP8W && c->d_model % 64 != 0) return bad("fp8w mode needs d_model % 64 == 0");
  if (c->dtype_mode == VLE_DTYPE_FP8 && c->d_model % 512 != 0) return bad("fp8 mode needs d_model % 512 == 0");
  if (c->max_batch < 1 || c->max_text < 1 || c->max_prompt < 0) return bad("bad capacity");

  vle_engine* e = new vle_engine();
  e->cfg = *c;
  e->d = c->d_model; e->H = c->nhead; e->dh = dh; e->L = c->num_layers; e->Q = c->num_quantizers;
  e->bos = c->prepend_bos ? 1 : 0;
  e->dtype = (c->dtype_mode == VLE_DTYPE_FP8W || c->dtype_mode == VLE_DTYPE_FP8) ? DT_BF16 : c->dtype_mode;
  e->w8 = c->dtype_mode == VLE_DTYPE_FP8W || c->dtype_mode == VLE_DTYPE_FP8;
  e->a8 = c->dtype_mode == VLE_DTYPE_FP8;
  e->max_B = c->max_batch; e->max_S = c->max_text; e->max_P = c->max_prompt;
  e->max_G = c->max_gen > 0 ? c->max_gen : 16 * c->max_text + 1;
  e->ctx_max = e->max_S + e->max_P + 1 + e->max_G;
  e->max_pos = std::max(e->max_S, e->max_P + 1 + e->max_G) + 1;
  e->max_rows = (int64_t)e->max_B * (e->max_S + e->max_P + 1 + e->max_G);
  if (const hipError_t sr = hipSetDevice(c->device); sr != hipSuccess) {  // no such device / no GPU on this box
    delete e;
    set_global_error((std::string("hipSetDevice failed: ") + hipGetErrorString(sr)).c_str());
    return VLE_EHIP;
  }
  {  // the persistent batch-1 step is a grid of 256 co-resident workgroups, one per CU (persist.hip)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) (void)hipGetLastError();
    e->ps_device_ok = cus >= 256;  // (that the selected pstep_kernel form fits a CU is asked per form: ps_form_ok below)
  }
  auto chk = [&](hipError_t r, const char* what) {
    if (r != hipSuccess) {
      std::string m = std::string(what) + ": " + hipGetErrorString(r);
      set_global_error(m.c_str());
      return false;
    }
    return true;
  };
  bool ok = chk(hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking), "hipStreamCreate");
  ok = ok && chk(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming), "hipEventCreate");
  ok = ok && chk(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming), "hipEventCreate");
  ok = ok && chk(hipEventCreateWithFlags(&e->ev_chk, hipEventDisableTiming), "hipEventCreate");
  for (int i = 0; i < 6 && ok; ++i) ok = chk(hipEventCreate(&e->ev_t[i]), "hipEventCreate");
  if (!ok) {
    vle_destroy(e);
    return VLE_EHIP;
  }
  {
    std::lock_guard<std::mutex> lk(g_engines_mu);
    g_engines.push_back(e);
  }
  *out = e;
  return VLE_OK;
}

// Free every capacity-dependent buffer and forget its address (vle_reserve re-creates them; vle_destroy ends them).
static void release_buffers(vle_engine* e) {
  for (void* p : e->buf_allocs) (void)hipFree(p);
  e->buf_allocs.clear();
  e->arena_b = vle_engine::Arena{};
  if (e->tables_host) (void)hipHostFree(e->tables_host);
  if (e->ps_host) (void)hipHostFree(e->ps_host);
  e->ps_host = nullptr; e->ps_host_bytes = 0;
  if (e->poll_host) (void)hipHostFree(e->poll_host);
  if (e->prog_host) (void)hipHostFree(e->prog_host);
  e->tables_host = e->poll_host = e->prog_host = nullptr;
  e->prog_dev = nullptr;
  e->kcache = e->vcache = nullptr;
  e->x_step = e->q_step = e->h_step = e->part_o = e->part_ml = e->logits = nullptr;
  e->k_new = e->v_new = nullptr; e->qgran = nullptr; e->qa_spin_fail = nullptr;
  e->ps_sample_valid = false;
  e->ps_table = nullptr; e->ps_sample = nullptr; e->ps_fold = nullptr; e->ps_gran = nullptr; e->ps_gran_n = 0; e->ps_ptrace = nullptr; e->ps_table_kc = nullptr; e->ps_table_ctx = 0; e->ps_table_B = 0; e->ps_fold_valid = false; e->ps_epoch = nullptr;
  e->xn_step = e->qkv_step = e->att_step = e->hT_step = nullptr;
  e->gs_ws = nullptr; e->ln_stats = nullptr; e->ao_part = nullptr; e->ao_cnt = nullptr;
  e->state_dev = nullptr; e->S = ArState{}; e->dyn_dev = nullptr;
  e->tokens = e->sampled = e->text_ids = e->prompt_codes = nullptr;
  e->forced_len_dev = nullptr; e->slot_seed_dev = nullptr; e->id_err_dev = nullptr;
  e->X = e->yemb = e->nar_logits = nullptr; e->ln_rows_stats = nullptr;
  e->Xn = e->QKV = e->ATT = e->Hb = nullptr;
  e->A8 = nullptr; e->a8_scale = nullptr;
  e->tables_dev = nullptr; e->tables_cap = 0;
  e->trace_ar = nullptr; e->trace_ar_cap = 0; e->trace_nar = nullptr; e->ktrace_buf = nullptr;
  e->nar_forced = nullptr; e->nar_forced_stride = 0;
  e->have_prefill = e->have_gen = false;
  e->slot_mode = false;
}

static void drop_graphs(vle_engine* e) {
  (void)hipSetDevice(e->cfg.device);
  if (e->st) (void)hipStreamSynchronize(e->st);
  for (auto& kv : e->graphs) {
    if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
    if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
  }
  e->graphs.clear();
}
// a PROCESS-WIDE kernel-selection knob changed: no engine may keep a graph captured under the old selection.  (Not thread-safe
// against a concurrent vle_ar_generate of ANOTHER engine: the knobs are tuning / A-B tools, set between calls.)
static void drop_all_graphs(vle_engine* caller) {
  std::lock_guard<std::mutex> lk(g_engines_mu);
  for (vle_engine* o : g_engines) drop_graphs(o);
  if (caller) (void)hipSetDevice(caller->cfg.device);
}

extern "C" void vle_destroy(vle_engine* e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> lk(g_engines_mu);
    g_engines.erase(std::remove(g_engines.begin(), g_engines.end(), e), g_engines.end());
  }
  (void)hipSetDevice(e->cfg.device);
  if (e->st) (void)hipStreamSynchronize(e->st);
  for (auto& kv : e->graphs) {
    if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
    if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
  }
  for (hipEvent_t ev : e->prof_pool) (void)hipEventDestroy(ev);
  for (void* p : e->allocs) (void)hipFree(p);
  release_buffers(e);
  for (int i = 0; i < 6; ++i)
    if (e->ev_t[i]) (void)hipEventDestroy(e->ev_t[i]);
  if (e->ev_in) (void)hipEventDestroy(e->ev_in);
  if (e->ev_out) (void)hipEventDestroy(e->ev_out);
  if (e->ev_chk) (void)hipEventDestroy(e->ev_chk);
  if (e->st) (void)hipStreamDestroy(e->st);
  delete e;
}

extern "C" const char* vle_last_error(const vle_engine* e) {
  if (e) return e->err.c_str();
  std::lock_guard<std::mutex> lk(g_err_mu);
  static thread_local std::string copy;
  copy = g_last_error;
  return copy.c_str();
}

extern "C" int vle_load_tensor(vle_engine* e, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!e || !key || !data || !shape || ndim < 1 || ndim > 4) return VLE_EINVAL;
  if (e->finalized || e->broken) return e->fail(VLE_ESTATE, e->broken ? "engine is unusable after a failed vle_reserve: destroy it" : "vle_load_tensor after vle_finalize_weights");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  e->host_w[key].assign(data, data + n);
  e->host_shape[key].assign(shape, shape + ndim);
  return VLE_OK;
}

static int load_layer(vle_engine* e, const std::string& p, LayerW& w, bool adaptive) {
  const int64_t d = e->d;
  const std::vector<float>* t;
#define GET(name, ...)                          \
  t = find_w(e, p + name, {__VA_ARGS__});       \
  if (!t) return VLE_EKEY
  int r;
  // one Linear weight: compute dtype, or (FP8W) bf16(W') plus, for the AR decoder, the fp8 codes + row scales
  auto up = [&](void** dst, void** q8, float** sc, int64_t N, int64_t K) -> int {
    if (!e->w8) return upload_T(e, dst, t->data(), t->size());
    return upload_fp8w(e, dst, (adaptive && !e->a8) ? nullptr : q8, sc, t->data(), N, K);  // FP8 mode: the NAR decoder's codes too
  };
  GET(".self_attn.in_proj_weight", 3 * d, d);
  if ((r = up(&w.wqkv, &w.wqkv8, &w.sqkv, 3 * d, d))) return r;
  GET(".self_attn.in_proj_bias", 3 * d);
  if ((r = upload_f32(e, &w.bqkv, t->data(), t->size()))) return r;
  GET(".self_attn.out_proj.weight", d, d);
  if ((r = up(&w.wo, &w.wo8, &w.so, d, d))) return r;
  GET(".self_attn.out_proj.bias", d);
  if ((r = upload_f32(e, &w.bo, t->data(), t->size()))) return r;
  GET(".linear1.weight", 4 * d, d);
  if ((r = up(&w.w1, &w.w18, &w.s1, 4 * d, d))) return r;
  GET(".linear1.bias", 4 * d);
  if ((r = upload_f32(e, &w.b1, t->data(), t->size()))) return r;
  GET(".linear2.weight", d, 4 * d);
  if ((r = up(&w.w2, &w.w28, &w.s2, d, 4 * d))) return r;
  GET(".linear2.bias", d);
  if ((r = upload_f32(e, &w.b2, t->data(), t->size()))) return r;
  if (!adaptive) {
    GET(".norm1.weight", d);
    if ((r = upload_f32(e, &w.g1, t->data(), t->size()))) return r;
    GET(".norm1.bias", d);
    if ((r = upload_f32(e, &w.be1, t->data(), t->size()))) return r;
    GET(".norm2.weight", d);
    if ((r = upload_f32(e, &w.g2, t->data(), t->size()))) return r;
    GET(".norm2.bias", d);
    if ((r = upload_f32(e, &w.be2, t->data(), t->size()))) return r;
    if (e->dtype == DT_BF16 && e->d % 256 == 0) {  // fused-LayerNorm vectors of the batched step (made for every engine: vle_reserve may grow max_batch)
      const auto* wq = find_w(e, p + ".self_attn.in_proj_weight", {3 * d, d});
      const auto* bq = find_w(e, p + ".self_attn.in_proj_bias", {3 * d});
      const auto* w1 = find_w(e, p + ".linear1.weight", {4 * d, d});
      const auto* b1 = find_w(e, p + ".linear1.bias", {4 * d});
      const auto* g1 = find_w(e, p + ".norm1.weight", {d});
      const auto* be1 = find_w(e, p + ".norm1.bias", {d});
      const auto* g2 = find_w(e, p + ".norm2.weight", {d});
      const auto* be2 = find_w(e, p + ".norm2.bias", {d});
      if (!wq || !bq || !w1 || !b1 || !g1 || !be1 || !g2 || !be2) return VLE_EKEY;
      if ((r = upload_wg_wb(e, wq->data(), 3 * d, d, g1->data(), be1->data(), bq->data(), &w.wg_qkv, &w.wb_qkv))) return r;
      if ((r = upload_wg_wb(e, w1->data(), 4 * d, d, g2->data(), be2->data(), b1->data(), &w.wg_1, &w.wb_1))) return r;
    }
  }
#undef GET
  return 0;
}

// AdaptiveLayerNorm (valle/modules/transformer.py:93-108): [w, b] = project_layer(stage_emb);
// w * (xhat * gamma + beta) + b  ==  xhat * (w * gamma) + (w * beta + b).  project_layer(stage_emb)
// does not depend on the input, so it is folded once per (stage, norm site), in fp32.
static int fold_adaln(vle_engine* e, const std::string& site, const std::vector<float>& stage_emb, float** g_out,
                      float** b_out) {
  const int64_t d = e->d;
  const auto* pw = find_w(e, site + ".project_layer.weight", {2 * d, d});
  const auto* pb = find_w(e, site + ".project_layer.bias", {2 * d});
  const auto* g = find_w(e, site + ".norm.weight", {d});
  const auto* b = find_w(e, site + ".norm.bias", {d});
  if (!pw || !pb || !g || !b) return VLE_EKEY;
  std::vector<float> gg(d), bb(d);
  for (int64_t o = 0; o < 2 * d; ++o) {
    double acc = 0.0;
    const float* row = pw->data() + o * d;
    for (int64_t k = 0; k < d; ++k) acc += (double)row[k] * (double)stage_emb[k];
    const float wb = (float)(acc + (double)(*pb)[o]);
    if (o < d) {
      gg[o] = wb * (*g)[o];
      bb[o] = wb * (*b)[o];  // + b part added below
    } else {
      bb[o - d] += wb;
    }
  }
  int r;
  if ((r = upload_f32(e, g_out, gg.data(), d))) return r;
  return upload_f32(e, b_out, bb.data(), d);
}

static int alloc_buffers(vle_engine* e);

// fragment-major copies of the AR decoder's weights for gemm_skinny.hip, made once the batched step exists (max_batch >= 2)
static int make_weight_packs(vle_engine* e) {
  const int64_t d = e->d;
  if (!(e->max_B >= 2 && e->dtype == DT_BF16 && e->d % 256 == 0) || e->ar.empty() || e->ar[0].wqkv_p != nullptr) return 0;
  const bool was = e->in_buffers;
  e->in_buffers = false;  // weights
  int r = 0;
  auto pack = [&](const void* src, void** dst, int64_t N, int64_t K) -> int {
    if (!src) return 0;
    unsigned char* p = nullptr;
    const size_t bytes = (size_t)((N + 15) / 16 * 16) * K * (e->w8 ? 1 : 2);
    int rr = dev_alloc(e, &p, bytes);
    if (rr) return rr;
    E_LAUNCH(e, launch_pack_w_frag(e->st, src, p, (int)N, (int)K, e->w8 ? 1 : 0));
    *dst = p;
    return 0;
  };
  for (int l = 0; l < e->L && !r; ++l) {
    LayerW& w = e->ar[l];
    if ((r = pack(e->w8 ? w.wqkv8 : w.wqkv, &w.wqkv_p, 3 * d, d))) break;
    if ((r = pack(e->w8 ? w.wo8 : w.wo, &w.wo_p, d, d))) break;
    if ((r = pack(e->w8 ? w.w18 : w.w1, &w.w1_p, 4 * d, d))) break;
    if ((r = pack(e->w8 ? w.w28 : w.w2, &w.w2_p, d, 4 * d))) break;
  }
  if (!r) r = pack(e->w8 ? e->ar_predict8 : e->ar_predict, &e->ar_predict_p, V_AR, d);
  e->in_buffers = was;
  if (r) return r;
  E_HIP(e, hipStreamSynchronize(e->st));
  return 0;
}

extern "C" int vle_finalize_weights(vle_engine* e) {
  if (!e) return VLE_EINVAL;
  if (e->finalized) return VLE_OK;
  if (e->broken) return e->fail(VLE_ESTATE, "engine is unusable after a failed vle_reserve: destroy it");
  E_HIP(e, hipSetDevice(e->cfg.device));
  const int64_t d = e->d;
  const std::vector<float>* t;
  int r;
#define GETK(key, ...)                   \
  t = find_w(e, key, {__VA_ARGS__});     \
  if (!t) return VLE_EKEY
  GETK("ar_text_embedding.word_embeddings.weight", NUM_TEXT_TOKENS, d);
  if ((r = upload_f32(e, &e->ar_text_emb, t->data(), t->size()))) return r;
  GETK("ar_audio_embedding.word_embeddings.weight", V_AR + e->bos, d);
  if ((r = upload_f32(e, &e->ar_audio_emb, t->data(), t->size()))) return r;
  float al[4] = {1.f, 1.f, 1.f, 1.f};
  GETK("ar_text_position.alpha", 1);
  al[0] = (*t)[0];
  GETK("ar_audio_position.alpha", 1);
  al[1] = (*t)[0];
  e->ar.resize(e->L);
  for (int l = 0; l < e->L; ++l)
    if ((r = load_layer(e, "ar_decoder.layers." + std::to_string(l), e->ar[l], false))) return r;
  GETK("ar_decoder.norm.weight", d);
  if ((r = upload_f32(e, &e->ar_norm_g, t->data(), t->size()))) return r;
  GETK("ar_decoder.norm.bias", d);
  if ((r = upload_f32(e, &e->ar_norm_b, t->data(), t->size()))) return r;
  GETK("ar_predict_layer.weight", V_AR, d);
  if (e->w8) r = upload_fp8w(e, &e->ar_predict, &e->ar_predict8, &e->ar_predict_s, t->data(), V_AR, d);
  else r = upload_T(e, &e->ar_predict, t->data(), t->size());
  if (r) return r;
  if (e->dtype == DT_BF16 && e->d % 256 == 0) {
    const auto* ng = find_w(e, "ar_decoder.norm.weight", {d});
    const auto* nb = find_w(e, "ar_decoder.norm.bias", {d});
    if (!ng || !nb) return VLE_EKEY;
    if ((r = upload_wg_wb(e, t->data(), V_AR, d, ng->data(), nb->data(), nullptr, &e->wg_pred, &e->wb_pred))) return r;
  }

  if (e->Q > 1) {
    GETK("nar_text_embedding.word_embeddings.weight", NUM_TEXT_TOKENS, d);
    if ((r = upload_f32(e, &e->nar_text_emb, t->data(), t->size()))) return r;
    for (int j = 0; j < e->Q; ++j) {
      GETK("nar_audio_embeddings." + std::to_string(j) + ".word_embeddings.weight", j == 0 ? V_AR : NUM_AUDIO_TOKENS, d);
      if ((r = upload_f32(e, &e->nar_audio_emb[j], t->data(), t->size()))) return r;
    }
    const float* tab[8];
    for (int j = 0; j < 8; ++j) tab[j] = e->nar_audio_emb[j];
    if ((r = dev_alloc(e, &e->nar_audio_emb_tab, 8))) return r;
    E_HIP(e, hipMemcpy((void*)e->nar_audio_emb_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    GETK("nar_text_position.alpha", 1);
    al[2] = (*t)[0];
    GETK("nar_audio_position.alpha", 1);
    al[3] = (*t)[0];
    e->nar.resize(e->L);
    for (int l = 0; l < e->L; ++l)
      if ((r = load_layer(e, "nar_decoder.layers." + std::to_string(l), e->nar[l], true))) return r;
    e->nar_gamma.assign(e->Q - 1, std::vector<float*>(2 * e->L + 1, nullptr));
    e->nar_beta.assign(e->Q - 1, std::vector<float*>(2 * e->L + 1, nullptr));
    for (int i = 0; i < e->Q - 1; ++i) {
      GETK("nar_stage_embeddings." + std::to_string(i) + ".word_embeddings.weight", 1, d);
      const std::vector<float> stage = *t;
      for (int l = 0; l < e->L; ++l) {
        const std::string p = "nar_decoder.layers." + std::to_string(l);
        if ((r = fold_adaln(e, p + ".norm1", stage, &e->nar_gamma[i][2 * l], &e->nar_beta[i][2 * l]))) return r;
        if ((r = fold_adaln(e, p + ".norm2", stage, &e->nar_gamma[i][2 * l + 1], &e->nar_beta[i][2 * l + 1]))) return r;
      }
      if ((r = fold_adaln(e, "nar_decoder.norm", stage, &e->nar_gamma[i][2 * e->L], &e->nar_beta[i][2 * e->L]))) return r;
      GETK("nar_predict_layers." + std::to_string(i) + ".weight", NUM_AUDIO_TOKENS, d);
      if (e->w8) r = upload_fp8w(e, &e->nar_predict[i], nullptr, nullptr, t->data(), NUM_AUDIO_TOKENS, d);
      else r = upload_T(e, &e->nar_predict[i], t->data(), t->size());
      if (r) return r;
    }
  }
#undef GETK
  if ((r = upload_f32(e, &e->alphas, al, 4))) return r;
  {
    auto it = e->host_w.find("position.pe");
    std::vector<float> pe;
    if (it != e->host_w.end() && e->host_shape["position.pe"].size() == 2 && e->host_shape["position.pe"][1] == d &&
        e->host_shape["position.pe"][0] >= e->max_pos) {
      pe.assign(it->second.begin(), it->second.begin() + (size_t)e->max_pos * d);
    } else {
      build_pe(pe, e->max_pos, e->d);
    }
    if ((r = upload_f32(e, &e->pe, pe.data(), pe.size()))) return r;
  }
  e->host_w.clear();
  e->host_shape.clear();
  e->in_buffers = true;
  r = alloc_buffers(e);
  e->in_buffers = false;
  if (r) return r;
  if ((r = make_weight_packs(e))) return r;
  if (e->Q > 1 && e->dtype == DT_BF16 && d % 256 == 0 && d <= 1536) {
    // row constants of the folded (Ada)LayerNorm of the NAR passes: per stage and layer, sg = W gamma_s, tb = W beta_s + b of the
    // in-projection and of linear1 on the folded AdaLN affine of that stage (fold_adaln above), fp64 sums on the device
    if ((r = dev_alloc(e, &e->nar_fold, (size_t)(e->Q - 1) * e->L * 14 * d))) return r;
    for (int i = 0; i < e->Q - 1; ++i)
      for (int l = 0; l < e->L; ++l) {
        float* f = e->nar_fold + ((size_t)i * e->L + l) * 14 * d;
        E_LAUNCH(e, launch_ps_fold(e->st, e->nar[l].wqkv, e->nar_gamma[i][2 * l], e->nar_beta[i][2 * l], e->nar[l].bqkv, f, f + 3 * d, (int)(3 * d), (int)d));
        E_LAUNCH(e, launch_ps_fold(e->st, e->nar[l].w1, e->nar_gamma[i][2 * l + 1], e->nar_beta[i][2 * l + 1], e->nar[l].b1, f + 6 * d, f + 10 * d, (int)(4 * d), (int)d));
      }
    E_HIP(e, hipStreamSynchronize(e->st));
  }
  e->in_buffers = true;  // from here on dev_alloc serves capacity-dependent buffers (traces, diagnostics)
  e->finalized = true;
  return VLE_OK;
}

static int alloc_buffers(vle_engine* e) {
  const size_t es = dtype_size(e->dtype);
  const int64_t d = e->d, B = e->max_B;
  int r;
  const size_t kv_elems = (size_t)e->L * B * e->H * e->ctx_max * e->dh;
  char* p = nullptr;
  if ((r = dev_alloc(e, &p, kv_elems * es))) return r;
  e->kcache = p;
  if ((r = dev_alloc(e, &p, kv_elems * es))) return r;
  e->vcache = p;
  // decode_attn.hip reads (and masks) cache slots beyond the context: they must hold finite values
  E_HIP(e, hipMemset(e->kcache, 0, kv_elems * es));
  E_HIP(e, hipMemset(e->vcache, 0, kv_elems * es));
  if ((r = dev_alloc(e, &e->x_step, B * d))) return r;
  if ((r = dev_alloc(e, &e->q_step, B * d))) return r;
  if ((r = dev_alloc(e, &e->h_step, B * 4 * d))) return r;
  if ((r = dev_alloc(e, &e->k_new, B * d))) return r;
  if ((r = dev_alloc(e, &e->v_new, B * d))) return r;
  if ((r = dev_alloc(e, &e->qgran, (size_t)e->L * d))) return r;
  E_HIP(e, hipMemset(e->qgran, 0, (size_t)e->L * d * sizeof(unsigned long long)));
  if ((r = dev_alloc(e, &e->qa_spin_fail, 4))) return r;
  E_HIP(e, hipMemset(e->qa_spin_fail, 0, 4 * sizeof(unsigned)));
  if (pstep_supports(e->w8 ? DT_FP8W : e->dtype, e->d, e->H, e->dh, V_AR)) {
    e->ps_gran_B = (e->dtype == DT_BF16 && !e->w8) ? std::max(1, std::min(e->max_B, PSB_MAX)) : 1;  // the batched persistent launch: [B] rows per edge
    e->ps_gran_n = (size_t)e->ps_gran_B * pstep_gran_count(e->d, e->H, e->L);
    if ((r = dev_alloc(e, &e->ps_gran, e->ps_gran_n))) return r;
    E_HIP(e, hipMemset(e->ps_gran, 0, e->ps_gran_n * sizeof(unsigned long long)));
    if ((r = dev_alloc(e, &e->ps_table, (size_t)e->L + 1))) return r;
    if ((r = dev_alloc(e, &e->ps_fold, (size_t)e->L * 14 * d + 2 * (V_AR + 3)))) return r;
    if ((r = dev_alloc(e, &e->ps_sample, (size_t)1))) return r;
    if ((r = dev_alloc(e, &e->ps_epoch, (size_t)4))) return r;
    e->ps_host_bytes = ((size_t)e->L + 1) * sizeof(PLayer) + sizeof(PStepSample);
    E_HIP(e, hipHostMalloc((void**)&e->ps_host, e->ps_host_bytes, hipHostMallocDefault));
    debug_host_note(e->ps_host, e->ps_host_bytes, "pinned ps_host");
  }
  if ((r = dev_alloc(e, &e->part_o, (size_t)B * e->H * 16 * e->dh))) return r;
  if ((r = dev_alloc(e, &e->part_ml, (size_t)B * e->H * 16 * 2))) return r;
  if ((r = dev_alloc(e, &e->logits, B * V_AR))) return r;
  E_HIP(e, hipMemset(e->x_step, 0, (size_t)B * d * sizeof(float)));
  E_HIP(e, hipMemset(e->q_step, 0, (size_t)B * d * sizeof(float)));
  E_HIP(e, hipMemset(e->h_step, 0, (size_t)B * 4 * d * sizeof(float)));
  E_HIP(e, hipMemset(e->k_new, 0, (size_t)B * d * sizeof(float)));
  E_HIP(e, hipMemset(e->v_new, 0, (size_t)B * d * sizeof(float)));
  E_HIP(e, hipMemset(e->part_o, 0, (size_t)B * e->H * 16 * e->dh * sizeof(float)));
  E_HIP(e, hipMemset(e->part_ml, 0, (size_t)B * e->H * 16 * 2 * sizeof(float)));
  E_HIP(e, hipMemset(e->logits, 0, (size_t)B * V_AR * sizeof(float)));
  {  // GEMM-path step buffers (every batch size: the GEMV path also uses them when it cannot hold B rows in LDS)
    const size_t Bp = (size_t)(B + 15) / 16 * 16;  // the fragment-major layout (common.h xf_index) holds whole 16-row fragments
    if ((r = dev_alloc(e, &p, Bp * d * es))) return r;
    e->xn_step = p;
    if ((r = dev_alloc(e, &p, (size_t)B * 3 * d * es))) return r;
    e->qkv_step = p;
    if ((r = dev_alloc(e, &p, Bp * d * es))) return r;
    e->att_step = p;
    if ((r = dev_alloc(e, &p, Bp * 4 * d * es))) return r;
    e->hT_step = p;
    if ((r = dev_alloc(e, &p, gemm_skinny_workspace_bytes()))) return r;
    E_HIP(e, hipMemset(p, 0, gemm_skinny_workspace_bytes()));
    e->gs_ws = p;
    // free / finished slots and the rows that pad the batch to whole 16-row fragments keep whatever they held: start from finite
    // values everywhere (MFMA rows are independent, so a NaN there cannot reach a live row -- this is hygiene, not a fix)
    E_HIP(e, hipMemset(e->xn_step, 0, Bp * d * es));
    E_HIP(e, hipMemset(e->qkv_step, 0, (size_t)B * 3 * d * es));
    E_HIP(e, hipMemset(e->att_step, 0, Bp * d * es));
    E_HIP(e, hipMemset(e->hT_step, 0, Bp * 4 * d * es));
    if ((r = dev_alloc(e, &e->ln_stats, (size_t)64 * (d / 16 + 1) * 2))) return r;
    E_HIP(e, hipMemset(e->ln_stats, 0, (size_t)64 * (d / 16 + 1) * 2 * sizeof(float)));
    if ((r = dev_alloc(e, &e->ao_part, (size_t)B * e->H * d))) return r;
    E_HIP(e, hipMemset(e->ao_part, 0, (size_t)B * e->H * d * sizeof(float)));
    if ((r = dev_alloc(e, &p, (size_t)B * sizeof(int)))) return r;
    E_HIP(e, hipMemset(p, 0, (size_t)B * sizeof(int)));
    e->ao_cnt = (int*)p;
  }
  if ((r = dev_alloc(e, &e->state_dev, 6 * B + 8))) return r;
  e->S.kv_len = e->state_dev;
  e->S.audio_pos = e->state_dev + B;
  e->S.n_gen = e->state_dev + 2 * B;
  e->S.done = e->state_dev + 3 * B;
  e->S.cap = e->state_dev + 4 * B;
  e->S.iter = e->state_dev + 5 * B;
  e->S.done_count = e->state_dev + 6 * B;
  if ((r = dev_alloc(e, &e->dyn_dev, 1))) return r;
  if ((r = dev_alloc(e, &e->tokens, (size_t)B * e->max_G))) return r;
  if ((r = dev_alloc(e, &e->sampled, (size_t)B * e->max_G))) return r;
  E_HIP(e, hipMemset(e->tokens, 0, (size_t)B * e->max_G * sizeof(int64_t)));
  E_HIP(e, hipMemset(e->sampled, 0, (size_t)B * e->max_G * sizeof(int64_t)));
  if ((r = dev_alloc(e, &e->text_ids, (size_t)B * e->max_S))) return r;
  if ((r = dev_alloc(e, &e->prompt_codes, (size_t)B * (e->max_P + e->max_G) * 8))) return r;
  if ((r = dev_alloc(e, &e->forced_len_dev, B))) return r;
  if ((r = dev_alloc(e, &e->slot_seed_dev, B))) return r;
  if ((r = dev_alloc(e, &e->id_err_dev, 4))) return r;
  E_HIP(e, hipMemset(e->id_err_dev, 0, 4 * sizeof(int32_t)));
  const int64_t R = e->max_rows;
  if ((r = dev_alloc(e, &e->X, (size_t)R * d))) return r;
  if (e->dtype == DT_BF16 && d % 256 == 0 && d <= 1536) {
    if ((r = dev_alloc(e, &e->ln_rows_stats, (size_t)R * (d / LN_GROUP) * 2))) return r;
  }
  if ((r = dev_alloc(e, &p, (size_t)R * d * es))) return r;
  e->Xn = p;
  if ((r = dev_alloc(e, &p, (size_t)R * 3 * d * es))) return r;
  e->QKV = p;
  if ((r = dev_alloc(e, &p, (size_t)R * d * es))) return r;
  e->ATT = p;
  if ((r = dev_alloc(e, &p, (size_t)R * 4 * d * es))) return r;
  e->Hb = p;
  if (e->a8) {
    if ((r = dev_alloc(e, &e->A8, (size_t)R * 4 * d))) return r;
    if ((r = dev_alloc(e, &e->a8_scale, (size_t)R))) return r;
  }
  if (e->Q > 1) {
    if ((r = dev_alloc(e, &e->yemb, (size_t)B * (e->max_P + e->max_G) * d))) return r;
    if ((r = dev_alloc(e, &e->nar_logits, (size_t)B * e->max_G * NUM_AUDIO_TOKENS))) return r;
  }
  // row tables: generous upper bound (a handful of int32 per packed row)
  e->tables_cap = 8 * R + 64 * B + 1024;
  if ((r = dev_alloc(e, &e->tables_dev, e->tables_cap))) return r;
  E_HIP(e, hipHostMalloc((void**)&e->tables_host, e->tables_cap * sizeof(int32_t), hipHostMallocDefault));
  E_HIP(e, hipHostMalloc((void**)&e->poll_host, 64 * sizeof(int32_t), hipHostMallocDefault));
  debug_host_note(e->tables_host, e->tables_cap * sizeof(int32_t), "pinned tables_host");
  debug_host_note(e->poll_host, 64 * sizeof(int32_t), "pinned poll_host");
  // progress words the sampling kernel writes while graphs are in flight: they must be COHERENT (fine-grained) host memory or the
  // host sees nothing until a synchronisation; without a device mapping the AR loop falls back to the event / D2H poll
  e->prog_dev = nullptr;
  if (hipHostMalloc((void**)&e->prog_host, 16 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    memset(e->prog_host, 0, 16 * sizeof(int32_t));
    if (hipHostGetDevicePointer((void**)&e->prog_dev, e->prog_host, 0) != hipSuccess) e->prog_dev = nullptr;
    debug_host_note(e->prog_host, 16 * sizeof(int32_t), "pinned+mapped prog_host");
    debug_host_note(e->prog_dev, 16 * sizeof(int32_t), "device alias of prog_host");
  } else {
    (void)hipGetLastError();
    e->prog_host = nullptr;
  }
  return 0;
}

// =================================================================================================
// schedules
// =================================================================================================
namespace {

struct TableBuilder {  // packs int32 tables into the pinned mirror, one H2D copy
  vle_engine* e;
  int64_t used = 0;
  int32_t* take(int64_t n, int32_t** host) {
    if (used + n > e->tables_cap) return nullptr;
    *host = e->tables_host + used;
    int32_t* dev = e->tables_dev + used;
    used += (n + 3) & ~int64_t(3);
    return dev;
  }
};

int enter(vle_engine* e, void* caller_stream) {
  E_HIP(e, hipSetDevice(e->cfg.device));
  E_HIP(e, hipStreamSynchronize(e->st));  // the pinned table mirror is reused by every call
  E_HIP(e, hipEventRecord(e->ev_in, (hipStream_t)caller_stream));
  E_HIP(e, hipStreamWaitEvent(e->st, e->ev_in, 0));
  return 0;
}
int leave(vle_engine* e, void* caller_stream) {
  E_HIP(e, hipEventRecord(e->ev_out, e->st));
  E_HIP(e, hipStreamWaitEvent((hipStream_t)caller_stream, e->ev_out, 0));
  return 0;
}

inline void* cache_layer(vle_engine* e, void* base, int l) {
  return (char*)base + (size_t)l * e->B * e->H * e->ctx_max * e->dh * dtype_size(e->dtype);
}

const char* kIdErrMsg = "token id out of range: text ids must be in [0, 512), first-codebook ids in [0, 1024], the other "
                        "codebooks in [0, 1024) (the reference's nn.Embedding raises IndexError)";
constexpr int POLL_IDERR = 40;  // poll_host slot of the id-range flag
constexpr int POLL_PSFAIL = 41; // ... of the persistent step's give-up counter

// one transformer layer over packed rows (prefill: AR weights + prefix-LM mask; NAR: no mask, folded AdaLN)
// one Linear over packed rows: engine mode FP8 quantises the bf16 activations per row and runs the fp8 MFMA GEMM on the
// layer's e4m3fn codes (gemm_fp8.hip); every other mode (and shapes that kernel does not cover) the bf16 / fp32 GEMM
int gemm_rows(vle_engine* e, const void* A, const void* w, const void* w8, const float* wscale, const float* bias, void* out,
              float* resid, int64_t rows, int N, int K, int epi) {
  if (e->a8 && e->opt_fp8_gemm && w8 != nullptr && wscale != nullptr) {
    const int q = launch_quantize_rows_fp8(e->st, A, e->A8, e->a8_scale, rows, K);
    if (q < 0) return q;
    if (q == 0) {
      const int g = launch_gemm_fp8(e->st, e->A8, e->a8_scale, w8, wscale, bias, out, resid, rows, N, K, epi);
      if (g <= 0) return g;
    }
  }
  return launch_gemm(e->st, e->dtype, A, w, bias, out, resid, rows, N, K, epi);
}

// LayerNorm folded into the GEMMs of one layer over packed rows (kernels.h GemmLn): row constants of this layer's two norm sites,
// whether X's norm1 operand (bf16(x * gamma1) in Xn + group statistics) was left by the previous layer's linear2, and the gamma of
// the NEXT layer's norm1 (null: this is the last layer -- its linear2 is a plain residual GEMM)
struct LnFoldLayer {
  const float *sg_qkv = nullptr, *tb_qkv = nullptr, *sg_1 = nullptr, *tb_1 = nullptr;
  bool in_folded = false;
  const float* next_g1 = nullptr;
};
bool use_ln_fold(const vle_engine* e, int64_t rows) {
  return e->opt_ln_fold && !e->a8 && e->ln_rows_stats != nullptr && gemm_ln_supports(e->dtype, rows, e->d);
}

int enqueue_layer_rows(vle_engine* e, const LayerW& w, const float* g1, const float* b1, const float* g2, const float* b2,
                       int64_t rows, const int32_t* seq_off, const int32_t* text_len, int max_len, int causal,
                       void* kc, void* vc, const int32_t* row_seq, const int32_t* row_pos, const LnFoldLayer* f = nullptr) {
  const int d = e->d;
  hipStream_t st = e->st;
  if (f != nullptr) {
    // 5 launches instead of 7: the two LayerNorm passes over the rows (read 4 d + write 2 d bytes per row each, and at one
    // utterance 5 us + a launch boundary each) ride on the residual GEMMs' epilogues
    GemmLn ln;
    if (f->in_folded) {
      ln.stats_in = e->ln_rows_stats; ln.stats_ld = e->max_rows; ln.sg = f->sg_qkv;
      E_LAUNCH(e, launch_gemm(st, e->dtype, e->Xn, w.wqkv, f->tb_qkv, e->QKV, nullptr, rows, 3 * d, d, EPI_STORE_LNC, &ln));
    } else {
      E_LAUNCH(e, launch_layernorm(st, e->dtype, e->X, nullptr, g1, b1, e->Xn, rows, d));
      E_LAUNCH(e, launch_gemm(st, e->dtype, e->Xn, w.wqkv, w.bqkv, e->QKV, nullptr, rows, 3 * d, d, EPI_STORE));
    }
    if (kc) E_LAUNCH(e, launch_kv_scatter(st, e->dtype, e->QKV, kc, vc, row_seq, row_pos, rows, d, e->H, e->ctx_max));
    E_LAUNCH(e, launch_attention(st, e->dtype, e->QKV, e->ATT, seq_off, text_len, e->nseq > 0 ? e->nseq : e->B, max_len, d, e->H, causal));
    ln = GemmLn();
    ln.gamma = g2; ln.xg = e->Xn; ln.stats_out = e->ln_rows_stats; ln.stats_ld = e->max_rows;
    E_LAUNCH(e, launch_gemm(st, e->dtype, e->ATT, w.wo, w.bo, nullptr, e->X, rows, d, d, EPI_RESID_LNP, &ln));
    ln = GemmLn();
    ln.stats_in = e->ln_rows_stats; ln.stats_ld = e->max_rows; ln.sg = f->sg_1;
    E_LAUNCH(e, launch_gemm(st, e->dtype, e->Xn, w.w1, f->tb_1, e->Hb, nullptr, rows, 4 * d, d, EPI_RELU_LNC, &ln));
    if (f->next_g1 != nullptr) {
      ln = GemmLn();
      ln.gamma = f->next_g1; ln.xg = e->Xn; ln.stats_out = e->ln_rows_stats; ln.stats_ld = e->max_rows;
      E_LAUNCH(e, launch_gemm(st, e->dtype, e->Hb, w.w2, w.b2, nullptr, e->X, rows, d, 4 * d, EPI_RESID_LNP, &ln));
    } else {
      E_LAUNCH(e, launch_gemm(st, e->dtype, e->Hb, w.w2, w.b2, nullptr, e->X, rows, d, 4 * d, EPI_RESID));
    }
    return 0;
  }
  E_LAUNCH(e, launch_layernorm(st, e->dtype, e->X, nullptr, g1, b1, e->Xn, rows, d));
  E_LAUNCH(e, gemm_rows(e, e->Xn, w.wqkv, w.wqkv8, w.sqkv, w.bqkv, e->QKV, nullptr, rows, 3 * d, d, EPI_STORE));
  if (kc) E_LAUNCH(e, launch_kv_scatter(st, e->dtype, e->QKV, kc, vc, row_seq, row_pos, rows, d, e->H, e->ctx_max));
  E_LAUNCH(e, launch_attention(st, e->dtype, e->QKV, e->ATT, seq_off, text_len, e->nseq > 0 ? e->nseq : e->B, max_len, d, e->H, causal));
  E_LAUNCH(e, gemm_rows(e, e->ATT, w.wo, w.wo8, w.so, w.bo, nullptr, e->X, rows, d, d, EPI_RESID));
  E_LAUNCH(e, launch_layernorm(st, e->dtype, e->X, nullptr, g2, b2, e->Xn, rows, d));
  E_LAUNCH(e, gemm_rows(e, e->Xn, w.w1, w.w18, w.s1, w.b1, e->Hb, nullptr, rows, 4 * d, d, EPI_RELU));
  E_LAUNCH(e, gemm_rows(e, e->Hb, w.w2, w.w28, w.s2, w.b2, nullptr, e->X, rows, d, 4 * d, EPI_RESID));
  return 0;
}

// event pair around one launch when profiling (events come from a pool, resolved after the sync)
struct ProfScope {
  vle_engine* e;
  bool on;
  ProfScope(vle_engine* e_, int tag) : e(e_), on(e_->opt_profile) {
    if (!on) return;
    if (e->prof_used + 2 > e->prof_pool.size()) {
      on = false;
      return;
    }
    e->prof_tags.push_back(tag);
    (void)hipEventRecord(e->prof_pool[e->prof_used], e->st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e->prof_pool[e->prof_used + 1], e->st);
    e->prof_used += 2;
  }
};

// bf16, 2..64 utterances: LayerNorm kernel + weight-streaming MFMA GEMM (gemm_skinny.hip)
bool use_mfma_skinny(const vle_engine* e) {
  return e->dtype == DT_BF16 && e->B >= 2 && e->B <= 64 && !e->opt_no_gemm_skinny && e->d % 256 == 0 && e->dh % 4 == 0;
}

// step activations of the gemm_skinny path in the fragment-major layout (LayerNorm kernel instantiated for these widths)
bool use_xf(const vle_engine* e) {
  const int nv = e->d / 256;
  return e->opt_gs_xf && use_mfma_skinny(e) && e->d % 256 == 0 && (nv <= 4 || nv == 6 || nv == 8);
}

// LayerNorm folded into the gemm_skinny launches of the batched step (kernels.h LnProducer / LnConsumer)
bool use_fuse_ln(const vle_engine* e) {
  return e->opt_gs_fuse_ln && use_xf(e) && !e->ar.empty() && e->ar[0].wg_qkv != nullptr && e->wg_pred != nullptr && e->ln_stats != nullptr &&
         e->d <= 2048;
}
LnProducer ln_producer(const vle_engine* e, const float* gamma) {
  LnProducer p;
  p.gamma = gamma; p.xg_out = e->xn_step; p.stats_out = e->ln_stats; p.w8 = e->w8 ? 1 : 0; p.MF = (e->B + 15) / 16;
  return p;
}

bool use_skinny(const vle_engine* e) {
  if (use_mfma_skinny(e)) return false;
  return e->B <= SKINNY_MAX_B && (size_t)(e->B <= 1 ? 1 : e->B <= 2 ? 2 : e->B <= 4 ? 4 : 8) * 4 * e->d * sizeof(float) <= 160 * 1024;
}

// logits of the current x_step rows: final LayerNorm (valle.py:151) + ar_predict_layer (valle.py:1039)
// `fused`: x_step's producer (the last layer's FFN2 epilogue) already left bf16(x * gamma_final) + group statistics
int enqueue_ar_logits(vle_engine* e, bool fused = false) {
  hipStream_t st = e->st;
  ProfScope ps(e, 5);
  if (use_skinny(e)) {
    SkinnyArgs a;
    a.w = e->ar_predict; a.w8 = e->ar_predict8; a.wscale = e->ar_predict_s; a.bias = nullptr; a.N = V_AR; a.K = e->d; a.B = e->B;
    a.pro = PRO_LN; a.epi = SEPI_STORE; a.x = e->x_step; a.gamma = e->ar_norm_g; a.beta = e->ar_norm_b; a.out = e->logits;
    E_LAUNCH(e, launch_ar_linear(e, a));
  } else {
    const bool xf = use_xf(e);
    fused = fused && use_fuse_ln(e);
    // un-fused calls (prefill, slot admission) normalise into a scratch buffer: with the fused step xn_step holds the live
    // utterances' bf16(x * gamma) for their NEXT step and must survive
    void* xn = fused ? e->xn_step : e->att_step;
    if (!fused) {
      if (xf) E_LAUNCH(e, launch_layernorm_xf(st, e->x_step, e->ar_norm_g, e->ar_norm_b, xn, e->B, e->d, e->w8 ? 1 : 0));
      else E_LAUNCH(e, launch_layernorm(st, e->dtype, e->x_step, nullptr, e->ar_norm_g, e->ar_norm_b, xn, e->B, e->d));
    }
    if (use_mfma_skinny(e)) {
      GemmSkinnyArgs g;
      g.workspace = e->gs_ws; g.target_wgs = e->opt_gs_target; g.x_xf = xf ? 1 : 0;
      if (fused) {
        g.lnc.stats = e->ln_stats; g.lnc.wg = e->wg_pred; g.lnc.nslots = e->d / 16; g.bias = e->wb_pred;
      }
      g.x = xn; g.w = e->w8 ? e->ar_predict8 : e->ar_predict; g.wscale = e->w8 ? e->ar_predict_s : nullptr; g.M = e->B; g.N = V_AR;
      if (e->opt_gs_wpack && e->ar_predict_p) { g.w = e->ar_predict_p; g.w_packed = 1; } g.K = e->d; g.epi = GS_EPI_F32; g.out = e->logits;
      g.kt = e->next_kt(); g.rot = e->opt_gs_rot; g.dbg = e->opt_gs_dbg;
        E_LAUNCH(e, launch_gemm_skinny(st, g));
    } else {
      E_LAUNCH(e, launch_gemm(st, e->dtype, xn, e->ar_predict, nullptr, e->logits, nullptr, e->B, V_AR, e->d, EPI_F32));
    }
  }
  return 0;
}

int enqueue_ar_sample(vle_engine* e, int first, const int32_t* slot_map = nullptr, int nslots = 0) {
  ProfScope ps(e, 6);
  ArSampleArgs a{};
  a.s = e->S; a.dyn = e->dyn_dev; a.logits = e->logits; a.V = V_AR; a.B = slot_map ? nslots : e->B; a.d = e->d; a.bos = e->bos; a.first = first;
  a.slot_map = slot_map; a.id_err = e->id_err_dev;
  if (!first) a.kt = e->next_kt();
  if (e->opt_host_prog && !slot_map) a.host_prog = e->prog_dev;  // null without a coherent mapping (alloc_buffers)
  if (e->slot_mode) a.slot_seed = e->slot_seed_dev;
  if (use_mfma_skinny(e) && use_fuse_ln(e)) a.lnp = ln_producer(e, e->ar[0].g1);
  a.tokens = e->tokens; a.g_stride = e->max_G; a.sampled = e->sampled;
  a.audio_emb = e->ar_audio_emb; a.pe = e->pe; a.alpha_audio = e->alphas + 1; a.x = e->x_step; a.ctx_max = e->ctx_max;
  E_LAUNCH(e, launch_ar_sample(e->st, a));
  return 0;
}

// the step's KV stream (all layers, this batch, full capacity) against the 256 MB memory-side cache: decode_attn.hip reads it
// non-temporally when nothing of it can survive until the next step
int kv_stream_nt(const vle_engine* e) {
  const int64_t bytes = (int64_t)2 * e->L * e->B * e->H * e->ctx_max * e->dh * (int64_t)dtype_size(e->dtype);
  return bytes > ((int64_t)192 << 20) ? 1 : 0;
}

// FP8W engines run the persistent launch in the forms instantiated for fp8 weight rows: the hidden row as bf16 pairs, three-barrier or
// folded LayerNorm, fp32 activation rows (the v_dot2c forms multiply bf16 weights), keys per lane 2, request schedules 0 / 3
// first-sweep waits per engine mode (tools/persist_probe.py --dtype sweeps, round 5: the stages' lengths differ with the weight format):
// bf16 D2 forms 0x325756 (126.9 us per step), fp8 weight rows 0x214645 (128.0 -> 125.7), fp32 0x217645 (191.7 -> 187.1)
// the batched launch (persist_nb.hip): the stages between two hand-offs grow with the number of rows, so the first sweeps are
// timed per batch (tools/persist_nb_probe.py, coordinate descent over the six edges)
constexpr int PSB_NAPS_DEFAULT[PSB_MAX + 1] = {0, 0, 0x405745, 0x305752, 0x317780, 0x006876, 0x007860};  // profiles/r06_persist_nb_probe*.json: 157 / 188 / 221 / 265 / 298 us per step
int ps_naps_of(const vle_engine* e) {
  if (e->opt_ps_naps >= 0) return e->opt_ps_naps;
  if (e->B > 1 && e->B <= PSB_MAX) return PSB_NAPS_DEFAULT[e->B];
  return e->dtype == DT_F32 ? 0x217645 : e->w8 ? 0x214645 : PS_NAPS_DEFAULT;
}
int ps_mode_of(const vle_engine* e) {
  if (e->dtype == DT_F32) return e->opt_ps_mode & ~(64 | 32 | 8 | 4);  // fp32 (token-exact) mode: nothing packed, three-barrier LayerNorm
  return e->w8 ? (e->opt_ps_mode & ~(64 | 8)) : e->opt_ps_mode;
}
bool ps_w8_mode_ok(const vle_engine* e) {
  return e->ar_predict8 != nullptr && e->ar_predict_s != nullptr && !e->ar.empty() && e->ar[0].wqkv8 != nullptr;
}
// The form launch_pstep would pick for this engine's weight type and options exists AND fits one workgroup per CU (persist.hip
// pstep_form_ok; the answer is cached per option set).  Option sets without an instantiated form -- persist_nk = 4, persist_pf = 1 / 2,
// packing modes the ladder dropped -- run the launch chain instead of failing the call.
bool ps_form_ok(const vle_engine* e) {
  const int key = (ps_mode_of(e) & 0xffff) | (e->opt_ps_nk << 16) | (e->opt_ps_pf << 20) | ((e->opt_ps_trace ? 1 : 0) << 24) | ((e->w8 ? DT_FP8W : e->dtype) << 25);
  if (e->ps_form_key != key) {
    e->ps_form_key = key;
    e->ps_form_res = pstep_form_ok(e->w8 ? DT_FP8W : e->dtype, ps_mode_of(e), e->opt_ps_nk, e->opt_ps_pf, e->opt_ps_trace);
  }
  return e->ps_form_res == 1;
}

// The persistent step (persist.hip) covers this call: batch 1, the covered shape, bf16 or fp8 weights, its table built for this cache
// ... or 2 .. PSB_MAX utterances on pstepb_kernel (persist_nb.hip): bf16 weights, the default form only (PS_MODE_DEFAULT's packing /
// LayerNorm / v_dot2c bits, 2 keys per lane, request schedule 3), sampling inside the launch
bool psb_covers(const vle_engine* e, int B) {
  if (!e->opt_persist_batch || B < 2 || B > PSB_MAX || B > e->ps_gran_B) return false;
  if (e->dtype != DT_BF16 || e->w8 || !e->opt_ps_sample || e->opt_ps_nk != 2 || e->opt_ps_pf != 3) return false;
  if ((e->opt_ps_mode & 0xfc) != (PS_MODE_DEFAULT & 0xfc)) return false;
  if (!pstepb_supports(e->dtype, e->d, e->H, e->dh, V_AR, B)) return false;
  if (e->psb_form_key != B * 2 + (e->opt_ps_trace ? 1 : 0)) {
    e->psb_form_key = B * 2 + (e->opt_ps_trace ? 1 : 0);
    e->psb_form_res = pstepb_form_ok(B, e->opt_ps_trace);
  }
  return e->psb_form_res == 1;
}
bool persist_ready(const vle_engine* e) {
  return e->opt_persist && e->ps_backoff == 0 && e->ps_device_ok && (e->B == 1 || psb_covers(e, e->B)) && !e->slot_mode && !e->opt_profile && e->ps_table != nullptr &&
         e->ps_gran != nullptr && (!e->w8 || ps_w8_mode_ok(e)) && ps_form_ok(e) &&
         e->ps_table_kc == e->kcache && e->ps_table_ctx == e->ctx_max && e->ps_table_B == e->B && (int)e->ar.size() == e->L;
}

// (re)build the operand table for a batch-1 call and forget the granules' old tags (the iteration counter restarts at every
// prefill).  Called from vle_ar_prefill: never inside a stream capture.
int persist_prepare(vle_engine* e) {
  if (!e->ps_table || !e->ps_gran || !e->ps_fold || !e->ps_sample || !e->ps_host || !(e->B == 1 || psb_covers(e, e->B))) return 0;
  if (e->w8 && (e->ar_predict8 == nullptr || e->ar.empty() || e->ar[0].wqkv8 == nullptr)) return 0;
  if (e->ps_table_kc != e->kcache || e->ps_table_ctx != e->ctx_max || e->ps_table_B != e->B) {
    std::vector<PLayer> tab(e->L + 1);
    const int64_t d = e->d;
    for (int l = 0; l < e->L; ++l) {
      const LayerW& w = e->ar[l];
      PLayer& t = tab[l];
      float* f = e->ps_fold + (size_t)l * 14 * d;
      t.sgqkv = f; t.tbqkv = f + 3 * d; t.sg1 = f + 6 * d; t.tb1 = f + 10 * d;
      if (e->dtype == DT_BF16 && !e->ps_fold_valid) {  // (fp32 mode runs the three-barrier form only: no row constants)
        E_LAUNCH(e, launch_ps_fold(e->st, w.wqkv, w.g1, w.be1, w.bqkv, f, f + 3 * d, (int)(3 * d), (int)d));
        E_LAUNCH(e, launch_ps_fold(e->st, w.w1, w.g2, w.be2, w.b1, f + 6 * d, f + 10 * d, (int)(4 * d), (int)d));
      }
      t.wqkv = w.wqkv; t.wo = w.wo; t.w1 = w.w1; t.w2 = w.w2;
      if (e->w8) {  // the e4m3fn codes + row scales (the folded-LayerNorm constants above come from bf16(W') = the same values)
        t.wqkv = w.wqkv8; t.wo = w.wo8; t.w1 = w.w18; t.w2 = w.w28;
        t.sqkv = w.sqkv; t.so = w.so; t.s1 = w.s1; t.s2 = w.s2;
      }
      t.bqkv = w.bqkv; t.bo = w.bo; t.b1 = w.b1; t.b2 = w.b2;
      t.g1 = w.g1; t.be1 = w.be1; t.g2 = w.g2; t.be2 = w.be2;
      t.kc = cache_layer(e, e->kcache, l); t.vc = cache_layer(e, e->vcache, l);
    }
    {  // entry L: the predict layer behind the final norm
      float* f = e->ps_fold + (size_t)e->L * 14 * d;
      if (e->dtype == DT_BF16 && !e->ps_fold_valid) E_LAUNCH(e, launch_ps_fold(e->st, e->ar_predict, e->ar_norm_g, e->ar_norm_b, nullptr, f, f + V_AR + 3, V_AR, (int)d));
      e->ps_fold_valid = true;
      PLayer& t = tab[e->L];
      t = tab[e->L - 1];  // every pointer valid
      t.wqkv = e->ar_predict; t.g1 = e->ar_norm_g; t.be1 = e->ar_norm_b; t.sgqkv = f; t.tbqkv = f + V_AR + 3;
      if (e->w8) {
        t.wqkv = e->ar_predict8; t.sqkv = e->ar_predict_s;
      }
    }
    // pinned staging + a copy ON THE ENGINE'S STREAM: the kernels that read the table through scalar loads are ordered behind it by
    // the stream itself (a synchronous copy from pageable memory may be a host write through the PCIe aperture)
    E_HIP(e, hipStreamSynchronize(e->st));  // no earlier copy still reads the staging area
    memcpy(e->ps_host, tab.data(), tab.size() * sizeof(PLayer));
    E_HIP(e, hipMemcpyAsync(e->ps_table, e->ps_host, tab.size() * sizeof(PLayer), hipMemcpyHostToDevice, e->st));
    e->ps_table_kc = e->kcache; e->ps_table_ctx = e->ctx_max; e->ps_table_B = e->B;
  }
  {
    PStepSample q;
    memset(&q, 0, sizeof(q));  // padding bytes too: the block is compared with what was uploaded last
    q.s = e->S; q.dyn = e->dyn_dev; q.bos = e->bos; q.pe_rows = e->max_pos;
    q.tokens = e->tokens; q.sampled = e->sampled; q.g_stride = e->max_G;
    q.audio_emb = e->ar_audio_emb; q.pe = e->pe; q.alpha_audio = e->alphas + 1; q.x = e->x_step;
    q.id_err = e->id_err_dev;
    q.host_prog = (e->opt_host_prog && !e->slot_mode) ? e->prog_dev : nullptr;  // (vle_slots_step runs a fixed number of steps: nobody polls)
    if (!e->ps_sample_valid || memcmp(&q, &e->ps_sample_sent, sizeof(q)) != 0) {  // unchanged between prefills: no stall of the stream
      unsigned char* hq = e->ps_host + ((size_t)e->L + 1) * sizeof(PLayer);
      E_HIP(e, hipStreamSynchronize(e->st));
      memcpy(hq, &q, sizeof(q));
      E_HIP(e, hipMemcpyAsync(e->ps_sample, hq, sizeof(q), hipMemcpyHostToDevice, e->st));
      memcpy(&e->ps_sample_sent, &q, sizeof(q));
      e->ps_sample_valid = true;
    }
  }
  E_HIP(e, hipMemsetAsync(e->ps_gran, 0, (e->ps_gran_n / e->ps_gran_B) * e->B * sizeof(unsigned long long), e->st));
  if (e->ps_epoch) E_HIP(e, hipMemsetAsync(e->ps_epoch, 0, 4 * sizeof(int32_t), e->st));
  return 0;
}

// Slot mode (continuous batching) on engines of 2 .. PSB_MAX slots: vle_slots_step advances every live slot on the batched persistent
// launch -- the slots are its utterances, a free slot is a stopped one (tables and granules prepared by vle_slots_begin)
bool slot_persist_ready(const vle_engine* e) {
  return e->slot_mode && e->opt_persist && e->ps_backoff == 0 && e->ps_device_ok && !e->opt_profile && psb_covers(e, e->max_B) && e->B == e->max_B &&
         e->ps_table != nullptr && e->ps_gran != nullptr && e->ps_epoch != nullptr && ps_form_ok(e) && e->ps_table_kc == e->kcache &&
         e->ps_table_ctx == e->ctx_max && e->ps_table_B == e->B && (int)e->ar.size() == e->L;
}

int enqueue_persist_step(vle_engine* e, int nsteps = 1) {
  PStepArgs a;
  a.layers = e->ps_table; a.L = e->L; a.d = e->d; a.nhead = e->H; a.dh = e->dh; a.V = V_AR; a.ctx_max = e->ctx_max;
  a.x_in = e->x_step; a.logits = e->logits;
  a.kv_len = e->S.kv_len; a.iter = e->S.iter; a.done = e->S.done; a.gran = e->ps_gran;
  a.fail = e->qa_spin_fail ? e->qa_spin_fail + 2 : nullptr;
  a.ptrace = e->opt_ps_trace ? e->ps_ptrace : nullptr;
  a.mode = ps_mode_of(e); a.nk = e->opt_ps_nk; a.pf = e->opt_ps_pf; a.naps = ps_naps_of(e);
  if (e->opt_ps_sample) {
    a.nsteps = nsteps;
    a.smp = e->ps_sample;
  }
  a.B = e->B;
  if (e->slot_mode) {
    a.epoch_ctr = e->ps_epoch;
    a.slot_seed = e->slot_seed_dev;
  }
  const int r = e->B > 1 ? launch_pstepb(e->st, e->dtype, a) : launch_pstep(e->st, e->w8 ? DT_FP8W : e->dtype, a);
  if (r != 0) return e->fail(VLE_EINVAL, "launch_pstep rejected the step");
  return 0;
}

// One AR step = one new token per utterance through the L layers + logits + sampling.
int enqueue_ar_step(vle_engine* e) {
  hipStream_t st = e->st;
  const int d = e->d;
  e->kt_idx = 0;
  if (persist_ready(e)) {
    int pr = enqueue_persist_step(e);
    if (pr) return pr;
    return e->opt_ps_sample ? 0 : enqueue_ar_sample(e, 0, nullptr, 0);
  }
  const bool sk = use_skinny(e);
  const bool gs = use_mfma_skinny(e);
  for (int l = 0; l < e->L; ++l) {
    const LayerW& w = e->ar[l];
    void* kc = cache_layer(e, e->kcache, l);
    void* vc = cache_layer(e, e->vcache, l);
    if (gs) {
      GemmSkinnyArgs g;
      g.M = e->B;
      g.workspace = e->gs_ws; g.target_wgs = e->opt_gs_target;
      if (e->L >= 2 && e->L <= 63) {  // split-K hand-off through granules: (AR iteration, layer) tags (kernels.h gran_epoch)
        g.gran_epoch = e->S.iter; g.gran_idx = l; g.gran_fail = e->qa_spin_fail ? e->qa_spin_fail + 1 : nullptr;
      }
      const bool xf = use_xf(e);                 // step activations fragment-major (common.h xf_index)
      const int xfw = xf ? (e->w8 ? 2 : 1) : 0;  // producer-side code: which k split the consuming GEMM uses
      const bool fuse = use_fuse_ln(e);          // no LayerNorm launches: producers emit x * gamma + statistics (kernels.h)
      LnConsumer lnc;
      lnc.stats = e->ln_stats; lnc.nslots = d / 16;
      {
        ProfScope ps(e, 0);
        if (!fuse) {
          if (xf) E_LAUNCH(e, launch_layernorm_xf(st, e->x_step, w.g1, w.be1, e->xn_step, e->B, d, e->w8 ? 1 : 0));
          else E_LAUNCH(e, launch_layernorm(st, e->dtype, e->x_step, nullptr, w.g1, w.be1, e->xn_step, e->B, d));
        }
        g.x_xf = xf ? 1 : 0;
        g.x = e->xn_step; g.w = e->w8 ? w.wqkv8 : w.wqkv; g.wscale = e->w8 ? w.sqkv : nullptr; g.bias = w.bqkv; g.N = 3 * d; g.K = d; g.epi = GS_EPI_QKV;
        if (fuse) {
          g.lnc = lnc; g.lnc.wg = w.wg_qkv; g.bias = w.wb_qkv;
        }
        const bool wpk = e->opt_gs_wpack && w.wqkv_p != nullptr;
        g.w_packed = wpk ? 1 : 0;
        if (wpk) g.w = w.wqkv_p;
        g.q_out = e->q_step; g.k_cache = kc; g.v_cache = vc; g.kv_len = e->S.kv_len; g.ctx_max = e->ctx_max; g.nhead = e->H; g.dh = e->dh;
        g.kt = e->next_kt(); g.rot = e->opt_gs_rot; g.dbg = e->opt_gs_dbg;
        E_LAUNCH(e, launch_gemm_skinny(st, g));
      }
      const bool direct = e->nsplit == 1;  // one block holds a whole (utterance, head): it normalises itself
      bool oproj_done = false;
      if (e->opt_attn_oproj && direct && e->dtype == DT_BF16 && e->ao_part != nullptr) {  // attention + out-proj + residual in one launch
        ProfScope ps(e, 1);
        const int fr = launch_decode_attention_oproj(st, e->q_step, kc, vc, e->S.kv_len, e->B, e->H, e->dh, e->ctx_max, e->S.done, w.wo, w.bo,
                                                     e->x_step, e->ao_part, e->ao_cnt, fuse ? ln_producer(e, w.g2) : LnProducer(), e->next_kt());
        if (fr < 0) return e->fail(VLE_EHIP, "launch_decode_attention_oproj failed");
        oproj_done = fr == 0;
      }
      if (!oproj_done) {
        ProfScope ps(e, 1);
        E_LAUNCH(e, launch_decode_attention(st, e->dtype, e->q_step, kc, vc, e->S.kv_len, e->part_o, e->part_ml, e->B, e->H, e->dh,
                                            e->ctx_max, e->nsplit, e->opt_nk, direct ? e->att_step : nullptr, e->S.done,
                                            direct ? xfw : 0, e->next_kt(), kv_stream_nt(e)));
      }
      if (!oproj_done) {
        ProfScope ps(e, 2);
        if (!direct) E_LAUNCH(e, launch_attn_combine(st, e->dtype, e->part_o, e->part_ml, e->att_step, e->B, e->H, e->dh, e->nsplit));
        g.x_xf = (xf && direct) ? 1 : 0;  // the merge kernel of the split path writes row-major
        g.lnc = LnConsumer();
        g.x = e->att_step; g.w = g.w_packed ? w.wo_p : (e->w8 ? w.wo8 : w.wo); g.wscale = e->w8 ? w.so : nullptr; g.bias = w.bo; g.N = d; g.K = d; g.epi = GS_EPI_RESID; g.resid = e->x_step;
        if (fuse) g.lnp = ln_producer(e, w.g2);  // the residual it completes is read by this layer's norm2 next
        g.kt = e->next_kt(); g.rot = e->opt_gs_rot; g.dbg = e->opt_gs_dbg;
        E_LAUNCH(e, launch_gemm_skinny(st, g));
        g.lnp = LnProducer();
      }
      {
        ProfScope ps(e, 3);
        if (!fuse) {
          if (xf) E_LAUNCH(e, launch_layernorm_xf(st, e->x_step, w.g2, w.be2, e->xn_step, e->B, d, e->w8 ? 1 : 0));
          else E_LAUNCH(e, launch_layernorm(st, e->dtype, e->x_step, nullptr, w.g2, w.be2, e->xn_step, e->B, d));
        }
        g.x_xf = xf ? 1 : 0; g.out_xf = xfw;  // FFN1 writes the hidden rows for FFN2 in the same layout
        g.x = e->xn_step; g.w = g.w_packed ? w.w1_p : (e->w8 ? w.w18 : w.w1); g.wscale = e->w8 ? w.s1 : nullptr; g.bias = w.b1; g.N = 4 * d; g.K = d; g.epi = GS_EPI_RELU; g.out = e->hT_step;
        if (fuse) {
          g.lnc = lnc; g.lnc.wg = w.wg_1; g.bias = w.wb_1;
        }
        g.kt = e->next_kt(); g.rot = e->opt_gs_rot; g.dbg = e->opt_gs_dbg;
        E_LAUNCH(e, launch_gemm_skinny(st, g));
        g.lnc = LnConsumer();
      }
      {
        ProfScope ps(e, 4);
        g.x_xf = xf ? 1 : 0; g.out_xf = 0;
        g.x = e->hT_step; g.w = g.w_packed ? w.w2_p : (e->w8 ? w.w28 : w.w2); g.wscale = e->w8 ? w.s2 : nullptr; g.bias = w.b2; g.N = d; g.K = 4 * d; g.epi = GS_EPI_RESID; g.resid = e->x_step;
        if (fuse) g.lnp = ln_producer(e, l + 1 < e->L ? e->ar[l + 1].g1 : e->ar_norm_g);  // next layer's norm1, or the final norm
        g.kt = e->next_kt(); g.rot = e->opt_gs_rot; g.dbg = e->opt_gs_dbg;
        E_LAUNCH(e, launch_gemm_skinny(st, g));
        g.lnp = LnProducer();
      }
      continue;
    }
    // batch 1: LN1 + QKV GEMV + KV write + decode attention over the old keys in ONE launch; the out-proj prologue below merges
    // the new token's own term (gemv1.hip qkv_attn1_kernel / PRO_ATTN_SELF).  Shapes it lacks take the two launches.
    bool fused_qa = false;
    if (sk && e->B == 1 && e->opt_qkv_attn && !e->opt_no_gemv1) {
      const bool codes = e->w8 && w.wqkv8 != nullptr;
      const int qdt = codes ? DT_FP8W : e->dtype;
      // the out-proj GEMV merges the new token's own term (PRO_ATTN_SELF): only gemv1 has that prologue, so the fused launch is
      // taken only where that GEMV exists too (e.g. not at fp32 d256-h2: 128-wide heads over 1-element thread slices)
      if (qkv_attn1_supports(qdt, d, e->H, e->dh) && (e->opt_qa_nsplit == 4 || e->opt_qa_nsplit == 8 || e->opt_qa_nsplit == 16) &&
          gemv1_attn_self_supports(codes ? DT_FP8W : e->dtype, d, e->dh, e->opt_qa_nsplit)) {
        ProfScope ps(e, 0);
        QkvAttnArgs q;
        q.w = codes ? w.wqkv8 : w.wqkv; q.wscale = codes ? w.sqkv : nullptr; q.bias = w.bqkv;
        q.x = e->x_step; q.gamma = w.g1; q.beta = w.be1; q.q_out = e->q_step; q.k_new = e->k_new; q.v_new = e->v_new;
        q.k_cache = kc; q.v_cache = vc; q.kv_len = e->S.kv_len; q.part_o = e->part_o; q.part_ml = e->part_ml;
        q.d = d; q.nhead = e->H; q.dh = e->dh; q.ctx_max = e->ctx_max; q.nsplit = e->opt_qa_nsplit;
        if (codes) {
          const int64_t w8_bytes = (int64_t)e->L * 12 * e->d * e->d + (int64_t)V_AR * e->d;
          q.temporal = e->opt_w8_temporal >= 0 ? e->opt_w8_temporal : (w8_bytes <= (int64_t)192 << 20 ? 1 : 0);
        }
        q.q_temporal = e->opt_qa_qtemporal;
        if (e->opt_qa_handoff && e->qgran != nullptr) {
          q.q_gran = e->qgran + (size_t)l * d; q.epoch_ptr = e->S.iter; q.spin_fail = e->qa_spin_fail;
          q.nk = e->opt_qa_nk == 8 ? 8 : e->opt_qa_nk == 2 ? 2 : 4;
        }
        q.kt = e->next_kt();
        const int fr = launch_qkv_attn1(st, qdt, q);
        if (fr < 0) return e->fail(VLE_EHIP, "launch_qkv_attn1 failed");
        fused_qa = fr == 0;
      }
    }
    if (fused_qa) {
      // nothing: both launches are done
    } else if (sk) {
      ProfScope ps(e, 0);
      SkinnyArgs a;
      a.w = w.wqkv; a.w8 = w.wqkv8; a.wscale = w.sqkv; a.bias = w.bqkv; a.N = 3 * d; a.K = d; a.B = e->B; a.pro = PRO_LN; a.epi = SEPI_QKV;
      a.x = e->x_step; a.gamma = w.g1; a.beta = w.be1; a.q_out = e->q_step; a.k_cache = kc; a.v_cache = vc;
      a.kv_len = e->S.kv_len; a.ctx_max = e->ctx_max; a.nhead = e->H; a.dh = e->dh;
      E_LAUNCH(e, launch_ar_linear(e, a));
    } else {
      ProfScope ps(e, 0);
      E_LAUNCH(e, launch_layernorm(st, e->dtype, e->x_step, nullptr, w.g1, w.be1, e->xn_step, e->B, d));
      E_LAUNCH(e, launch_gemm(st, e->dtype, e->xn_step, w.wqkv, w.bqkv, e->qkv_step, nullptr, e->B, 3 * d, d, EPI_STORE));
      E_LAUNCH(e, launch_qkv_split(st, e->dtype, e->qkv_step, e->q_step, kc, vc, e->S.kv_len, e->B, d, e->H, e->ctx_max));
    }
    if (!fused_qa) {
      ProfScope ps(e, 1);
      E_LAUNCH(e, launch_decode_attention(st, e->dtype, e->q_step, kc, vc, e->S.kv_len, e->part_o, e->part_ml, e->B, e->H, e->dh,
                                          e->ctx_max, e->nsplit, e->opt_nk, nullptr, e->B > 1 ? e->S.done : nullptr, 0, e->next_kt(), kv_stream_nt(e)));
    }
    if (sk) {
      {
        ProfScope ps(e, 2);
        SkinnyArgs a;
        a.w = w.wo; a.w8 = w.wo8; a.wscale = w.so; a.bias = w.bo; a.N = d; a.K = d; a.B = e->B; a.pro = PRO_ATTN; a.epi = SEPI_RESID;
        a.part_o = e->part_o; a.part_ml = e->part_ml; a.nsplit = e->nsplit; a.nhead = e->H; a.dh = e->dh; a.resid = e->x_step;
        if (fused_qa) {  // the partials exclude the new token: merge its own term here
          a.pro = PRO_ATTN_SELF; a.nsplit = e->opt_qa_nsplit; a.q_self = e->q_step; a.k_self = e->k_new; a.v_self = e->v_new;
        }
        a.act_bf16 = e->dtype == DT_BF16 ? (e->opt_act_bf16 & 1) : 0;
        E_LAUNCH(e, launch_ar_linear(e, a));
      }
      {
        ProfScope ps(e, 3);
        SkinnyArgs f1;
        f1.w = w.w1; f1.w8 = w.w18; f1.wscale = w.s1; f1.bias = w.b1; f1.N = 4 * d; f1.K = d; f1.B = e->B; f1.pro = PRO_LN; f1.epi = SEPI_RELU;
        f1.x = e->x_step; f1.gamma = w.g2; f1.beta = w.be2; f1.out = e->h_step; f1.act_bf16 = e->dtype == DT_BF16 ? (e->opt_act_bf16 & 2) : 0;
        E_LAUNCH(e, launch_ar_linear(e, f1));
      }
      {
        ProfScope ps(e, 4);
        SkinnyArgs f2;
        f2.w = w.w2; f2.w8 = w.w28; f2.wscale = w.s2; f2.bias = w.b2; f2.N = d; f2.K = 4 * d; f2.B = e->B; f2.pro = PRO_PLAIN; f2.epi = SEPI_RESID;
        f2.x = e->h_step; f2.resid = e->x_step;
        E_LAUNCH(e, launch_ar_linear(e, f2));
      }
    } else {
      {
        ProfScope ps(e, 2);
        E_LAUNCH(e, launch_attn_combine(st, e->dtype, e->part_o, e->part_ml, e->att_step, e->B, e->H, e->dh, e->nsplit));
        E_LAUNCH(e, launch_gemm(st, e->dtype, e->att_step, w.wo, w.bo, nullptr, e->x_step, e->B, d, d, EPI_RESID));
      }
      {
        ProfScope ps(e, 3);
        E_LAUNCH(e, launch_layernorm(st, e->dtype, e->x_step, nullptr, w.g2, w.be2, e->xn_step, e->B, d));
        E_LAUNCH(e, launch_gemm(st, e->dtype, e->xn_step, w.w1, w.b1, e->hT_step, nullptr, e->B, 4 * d, d, EPI_RELU));
      }
      {
        ProfScope ps(e, 4);
        E_LAUNCH(e, launch_gemm(st, e->dtype, e->hT_step, w.w2, w.b2, nullptr, e->x_step, e->B, d, 4 * d, EPI_RESID));
      }
    }
  }
  int r;
  if ((r = enqueue_ar_logits(e, gs))) return r;
  return enqueue_ar_sample(e, 0);
}

int capture_graph(vle_engine* e, int steps, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  E_HIP(e, hipStreamBeginCapture(e->st, hipStreamCaptureModeThreadLocal));
  int r = 0;
  if (persist_ready(e) && e->opt_ps_sample) {
    e->kt_idx = 0;
    r = enqueue_persist_step(e, steps);  // ONE launch runs all `steps` iterations, sampling included
  } else {
    for (int i = 0; i < steps && r == 0; ++i) r = enqueue_ar_step(e);
  }
  hipError_t ce = hipStreamEndCapture(e->st, &g);
  if (r) {
    if (g) (void)hipGraphDestroy(g);
    return r;
  }
  E_HIP(e, ce);
  hipError_t ie = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  E_HIP(e, ie);
  return 0;
}

}  // namespace

extern "C" int vle_ar_prefill(vle_engine* e, void* stream, const int64_t* text, int64_t s_stride, const int32_t* text_lens,
                              const int64_t* prompt_codes, int64_t p_stride, const int32_t* prompt_lens, int32_t B) {
  if (!e) return VLE_EINVAL;
  if (!e->finalized) return e->fail(VLE_ESTATE, e->broken ? "engine is unusable after a failed vle_reserve: destroy it" : "weights not finalized");
  if (!text || !text_lens || !prompt_codes || !prompt_lens) return e->fail(VLE_EINVAL, "null argument");
  if (B < 1 || B > e->max_B) return e->fail(VLE_EINVAL, "batch exceeds max_batch");
  for (int b = 0; b < B; ++b) {
    if (text_lens[b] < 1 || text_lens[b] > e->max_S || text_lens[b] > s_stride) return e->fail(VLE_EINVAL, "text_lens out of range");
    if (prompt_lens[b] < 0 || prompt_lens[b] > e->max_P || prompt_lens[b] > p_stride) return e->fail(VLE_EINVAL, "prompt_lens out of range");
    if (prompt_lens[b] + e->bos < 1) return e->fail(VLE_EINVAL, "empty prompt needs prepend_bos");
  }
  int r;
  if ((r = enter(e, stream))) return r;
  hipStream_t st = e->st;
  e->B = B;
  e->slot_mode = false;
  e->have_prefill = e->have_gen = false;
  e->nsplit = e->opt_nsplit > 0 ? e->opt_nsplit : choose_nsplit(e, B);
  e->S_len.assign(text_lens, text_lens + B);
  e->P_len.assign(prompt_lens, prompt_lens + B);
  E_HIP(e, hipEventRecord(e->ev_t[0], st));

  // engine-owned copies of the inputs (the NAR phase needs them again)
  E_HIP(e, hipMemcpy2DAsync(e->text_ids, e->max_S * sizeof(int64_t), text, s_stride * sizeof(int64_t),
                            std::min<int64_t>(s_stride, e->max_S) * sizeof(int64_t), B, hipMemcpyDeviceToDevice, st));
  const int64_t pp = (int64_t)(e->max_P + e->max_G) * e->Q;
  if (p_stride > 0)
    E_HIP(e, hipMemcpy2DAsync(e->prompt_codes, pp * sizeof(int64_t), prompt_codes, p_stride * e->Q * sizeof(int64_t),
                              std::min<int64_t>(p_stride, e->max_P) * e->Q * sizeof(int64_t), B, hipMemcpyDeviceToDevice, st));

  // tables
  TableBuilder tb{e};
  int64_t rows = 0;
  int max_len = 0;
  for (int b = 0; b < B; ++b) {
    const int n = text_lens[b] + e->bos + prompt_lens[b];
    rows += n;
    max_len = std::max(max_len, n);
  }
  int32_t *h_seq_off, *h_text_len, *h_row_seq, *h_row_pos, *h_last, *h_state, *h_prompt_len;
  int32_t* d_seq_off = tb.take(B + 1, &h_seq_off);
  int32_t* d_text_len = tb.take(B, &h_text_len);
  int32_t* d_prompt_len = tb.take(B, &h_prompt_len);
  int32_t* d_row_seq = tb.take(rows, &h_row_seq);
  int32_t* d_row_pos = tb.take(rows, &h_row_pos);
  int32_t* d_last = tb.take(B, &h_last);
  int32_t* d_state = tb.take(6 * e->max_B + 8, &h_state);
  if (!d_seq_off || !d_text_len || !d_prompt_len || !d_row_seq || !d_row_pos || !d_last || !d_state) return e->fail(VLE_EINVAL, "table overflow");
  int64_t off = 0;
  memset(h_state, 0, (6 * e->max_B + 8) * sizeof(int32_t));
  for (int b = 0; b < B; ++b) {
    const int n = text_lens[b] + e->bos + prompt_lens[b];
    h_seq_off[b] = (int32_t)off;
    h_text_len[b] = text_lens[b];
    h_prompt_len[b] = prompt_lens[b];
    for (int p = 0; p < n; ++p) {
      h_row_seq[off + p] = b;
      h_row_pos[off + p] = p;
    }
    off += n;
    h_last[b] = (int32_t)(off - 1);
    h_state[0 * e->max_B + b] = n;                              // kv_len: next free slot
    h_state[1 * e->max_B + b] = e->bos + prompt_lens[b];        // audio_pos of the next token
    h_state[4 * e->max_B + b] = 16 * text_lens[b];              // cap (valle.py:1047)
  }
  h_seq_off[B] = (int32_t)off;
  E_HIP(e, hipMemcpyAsync(e->tables_dev, e->tables_host, tb.used * sizeof(int32_t), hipMemcpyHostToDevice, st));
  E_HIP(e, hipMemcpyAsync(e->state_dev, d_state, (6 * e->max_B + 8) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  // the AR iteration counters restart at 0: so do the epochs of the fused launch's q granules -- forget the old tags
  if (e->qgran) E_HIP(e, hipMemsetAsync(e->qgran, 0, (size_t)e->L * e->d * sizeof(unsigned long long), st));
  if ((r = persist_prepare(e))) return r;
  // id range check on the engine-owned copies (sanitises them); the flag is read at the end of this call, by which
  // time these three tiny operations have long completed -- no stall of the stream
  E_HIP(e, hipMemsetAsync(e->id_err_dev, 0, sizeof(int32_t), st));
  E_LAUNCH(e, launch_check_ids(st, e->text_ids, e->max_S, 1, 1, d_text_len, nullptr, B, e->max_S, NUM_TEXT_TOKENS, 0, e->id_err_dev, 1));
  E_LAUNCH(e, launch_check_ids(st, e->prompt_codes, pp, e->Q, e->Q, d_prompt_len, nullptr, B, e->max_P, V_AR, NUM_AUDIO_TOKENS, e->id_err_dev, 2));
  E_HIP(e, hipMemcpyAsync(e->poll_host + POLL_IDERR, e->id_err_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  E_HIP(e, hipEventRecord(e->ev_chk, st));

  PrefillEmbedArgs pa{};
  pa.text = e->text_ids; pa.s_stride = e->max_S; pa.prompt = e->prompt_codes; pa.p_stride = e->max_P + e->max_G; pa.Q = e->Q;
  pa.text_len = d_text_len; pa.row_seq = d_row_seq; pa.row_pos = d_row_pos;
  pa.text_emb = e->ar_text_emb; pa.audio_emb = e->ar_audio_emb; pa.pe = e->pe;
  pa.alpha_text = e->alphas + 0; pa.alpha_audio = e->alphas + 1; pa.bos = e->bos; pa.d = e->d; pa.rows = rows; pa.x = e->X;
  E_LAUNCH(e, launch_prefill_embed(st, pa));
  const bool pf_fold = use_ln_fold(e, rows) && !e->ar.empty() && e->ar[0].wg_qkv != nullptr;
  for (int l = 0; l < e->L; ++l) {
    const LayerW& w = e->ar[l];
    LnFoldLayer f;
    f.sg_qkv = w.wg_qkv; f.tb_qkv = w.wb_qkv; f.sg_1 = w.wg_1; f.tb_1 = w.wb_1;
    f.in_folded = l > 0;
    f.next_g1 = l + 1 < e->L ? e->ar[l + 1].g1 : nullptr;
    if ((r = enqueue_layer_rows(e, w, w.g1, w.be1, w.g2, w.be2, rows, d_seq_off, d_text_len, max_len, 1,
                                cache_layer(e, e->kcache, l), cache_layer(e, e->vcache, l), d_row_seq, d_row_pos, pf_fold ? &f : nullptr)))
      return r;
  }
  E_LAUNCH(e, launch_gather_rows(st, e->X, d_last, e->x_step, B, e->d));
  if ((r = enqueue_ar_logits(e))) return r;
  E_HIP(e, hipEventRecord(e->ev_t[1], st));
  E_HIP(e, hipEventSynchronize(e->ev_chk));
  if (e->poll_host[POLL_IDERR] != 0) {
    (void)leave(e, stream);
    return e->fail(VLE_EINDEX, kIdErrMsg);
  }
  e->have_prefill = true;
  return leave(e, stream);
}

extern "C" int vle_ar_generate(vle_engine* e, void* stream, int32_t top_k, float temperature, uint64_t seed, int32_t max_new,
                               const int64_t* forced, int64_t forced_stride, const int32_t* forced_lens, int64_t* codes0,
                               int64_t g_stride, int32_t* gen_lens) {
  if (!e) return VLE_EINVAL;
  if (!e->have_prefill) return e->fail(VLE_ESTATE, "vle_ar_generate needs vle_ar_prefill first");
  if (!gen_lens) return e->fail(VLE_EINVAL, "gen_lens is null");
  if (!(temperature > 0.f)) return e->fail(VLE_EINVAL, "temperature must be > 0");
  if ((forced != nullptr) != (forced_lens != nullptr)) return e->fail(VLE_EINVAL, "forced and forced_lens go together");
  int r;
  if ((r = enter(e, stream))) return r;
  hipStream_t st = e->st;
  const int B = e->B;
  if (e->prog_host) e->prog_host[0] = e->prog_host[1] = 0;  // enter() synchronised the stream: no kernel is writing them

  // upper bound on loop iterations (stop rule valle.py:1047): n + bos > 16 S  first holds at n = 16 S + 1 - bos
  int bound = 0;
  for (int b = 0; b < B; ++b) {
    int nb = 16 * e->S_len[b] + 1 - e->bos;
    if (max_new > 0) nb = std::min(nb, (int)max_new);
    if (forced) {
      if (forced_lens[b] < 0 || forced_lens[b] > forced_stride) return e->fail(VLE_EINVAL, "forced_lens out of range");
      nb = forced_lens[b];
    }
    nb = std::min(nb, e->max_G);
    nb = std::min(nb, e->ctx_max - (e->S_len[b] + e->bos + e->P_len[b]));
    bound = std::max(bound, nb);
  }
  ArDyn dyn{};
  dyn.top_k = top_k; dyn.temperature = temperature; dyn.seed = seed; dyn.max_new = max_new; dyn.ignore_eos = e->opt_ignore_eos ? 1 : 0;
  dyn.has_forced = forced ? 1 : 0; dyn.forced = forced; dyn.forced_stride = forced_stride; dyn.forced_len = e->forced_len_dev;
  if (e->opt_trace_ar) {
    const int64_t need = (int64_t)(bound + 1);
    if (e->trace_ar_cap < need * B) {
      float* p = nullptr;
      if ((r = dev_alloc(e, &p, (size_t)need * e->max_B * V_AR))) return r;
      e->trace_ar = p;
      e->trace_ar_cap = need * e->max_B;
    }
    dyn.trace = e->trace_ar;
    dyn.trace_cap = e->trace_ar_cap / B;
  }
  // dyn / forced_lens go through the pinned mirror (valid until the stream consumed them: we sync below)
  static_assert(sizeof(ArDyn) % 4 == 0, "ArDyn packing");
  int32_t* hp = e->tables_host + e->tables_cap - (int64_t)(sizeof(ArDyn) / 4 + e->max_B + 4);
  memcpy(hp, &dyn, sizeof(ArDyn));
  E_HIP(e, hipMemcpyAsync(e->dyn_dev, hp, sizeof(ArDyn), hipMemcpyHostToDevice, st));
  if (forced) {
    int32_t* hf = hp + sizeof(ArDyn) / 4;
    memcpy(hf, forced_lens, B * sizeof(int32_t));
    E_HIP(e, hipMemcpyAsync(e->forced_len_dev, hf, B * sizeof(int32_t), hipMemcpyHostToDevice, st));
  }
  E_HIP(e, hipMemsetAsync(e->id_err_dev, 0, sizeof(int32_t), st));
  // the persistent launch's give-up counter belongs to THIS call (a launch that finds it non-zero ends at once: pstep_kernel)
  const bool ps_call = persist_ready(e);
  e->ps_last_call = ps_call;
  if (e->qa_spin_fail) E_HIP(e, hipMemsetAsync(e->qa_spin_fail + 2, 0, sizeof(unsigned), st));
  E_HIP(e, hipEventRecord(e->ev_t[2], st));

  // iteration 0: sample from the prefill's logits
  if ((r = enqueue_ar_sample(e, 1))) return r;
  int steps_done = 0;
  const int spg = e->opt_spg > 0 ? e->opt_spg : (persist_ready(e) && e->opt_ps_sample) ? e->opt_ps_steps : (e->cfg.steps_per_graph > 0 ? e->cfg.steps_per_graph : 8);
  const bool use_graph = e->cfg.use_graph != 0 && !e->opt_profile;
  if (e->opt_profile) {
    e->prof_used = 0;
    e->prof_tags.clear();
    for (int i = 0; i < 8; ++i) e->prof_ms[i] = e->prof_calls[i] = 0;
  }
  hipGraphExec_t g_multi = nullptr, g_single = nullptr;
  if (use_graph && bound > 0) {
    const int gkey = B * 64 + e->nsplit + (ps_call ? 1 << 20 : 0);  // the persistent launch and the chain are different graphs
    auto it = e->graphs.find(gkey);
    if (it == e->graphs.end()) {
      // first step runs eagerly (sets kernel attributes outside capture), then capture
      if ((r = enqueue_ar_step(e))) return r;
      steps_done = 1;
      if ((r = capture_graph(e, spg, &g_multi))) return r;
      if ((r = capture_graph(e, 1, &g_single))) return r;
      e->graphs[gkey] = {g_multi, g_single};
    } else {
      g_multi = it->second.first;
      g_single = it->second.second;
    }
  }
  // run; the host must not stall the stream, and must not run unboundedly ahead of it either (utterances stop early)
  constexpr int RING = 8, LAG = 3;
  hipEvent_t evs[RING];
  for (int i = 0; i < RING; ++i) evs[i] = nullptr;
  const bool hostprog = e->opt_host_prog && e->prog_dev != nullptr;  // no coherent mapping: event / D2H poll instead
  if (!hostprog)
    for (int i = 0; i < RING; ++i) E_HIP(e, hipEventCreateWithFlags(&evs[i], hipEventDisableTiming));
  volatile int32_t* hprog = e->prog_host;
  int launches = 0;
  bool all_done = false, stalled = false;
  // with the sampling step inside the persistent launch a launch ends by itself at the stop rule (length cap, max_new, forced
  // length, capacities: pstep_kernel applies ar_sample_kernel's rule): the tail of the loop is one more multi-step launch, not up to
  // spg - 1 single steps
  const bool self_stopping = use_graph && persist_ready(e) && e->opt_ps_sample;
  while (steps_done < bound && !all_done) {
    if (use_graph) {
      if (bound - steps_done >= spg || self_stopping) {
        E_HIP(e, hipGraphLaunch(g_multi, st));
        steps_done += spg;
      } else {
        E_HIP(e, hipGraphLaunch(g_single, st));
        steps_done += 1;
      }
    } else {
      if ((r = enqueue_ar_step(e))) return r;
      steps_done += 1;
    }
    ++launches;
    if (hostprog) {
      // progress words written by the sample kernel into pinned host memory: nothing goes on the stream between two
      // replays.  hp[1] counts sample launches (iteration 0 included), hp[0] the utterances done one launch earlier.
      const int ahead = LAG * (use_graph ? spg : 8);
      const auto t_start = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (steps_done + 1 - hprog[1] > ahead && hprog[0] < B) {
        if ((++spins & 0xfff) == 0) {
          if (hipStreamQuery(st) == hipSuccess) break;  // everything enqueued has run (also: the kernels were not built to report)
          if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(30)) {
            stalled = true;
            break;
          }
        }
        __builtin_ia32_pause();
      }
      if (stalled) break;
      if (hprog[0] >= B) all_done = true;
    } else {
      const int slot = (launches - 1) % RING;
      E_HIP(e, hipMemcpyAsync(e->poll_host + slot, e->S.done_count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      E_HIP(e, hipEventRecord(evs[slot], st));
      if (launches > LAG) {
        const int old = (launches - 1 - LAG) % RING;
        E_HIP(e, hipEventSynchronize(evs[old]));
        if (e->poll_host[old] >= B) all_done = true;
      }
    }
  }
  if (stalled) return e->fail(VLE_EHIP, "AR loop made no progress for 30 s");
  E_HIP(e, hipEventRecord(e->ev_t[3], st));
  // results
  std::vector<int32_t> st_host(6 * e->max_B + 8);
  E_HIP(e, hipMemcpyAsync(e->tables_host, e->state_dev, st_host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  E_HIP(e, hipMemcpyAsync(e->poll_host + POLL_IDERR, e->id_err_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  e->poll_host[POLL_PSFAIL] = 0;
  if (ps_call && e->dbg_inject_psfail > 0 && e->qa_spin_fail) {  // test hook: as if one wave had given up
    --e->dbg_inject_psfail;
    E_HIP(e, hipMemsetAsync(e->qa_spin_fail + 2, 1, 1, st));
  }
  if (ps_call && e->qa_spin_fail) E_HIP(e, hipMemcpyAsync(e->poll_host + POLL_PSFAIL, e->qa_spin_fail + 2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (codes0)
    E_HIP(e, hipMemcpy2DAsync(codes0, g_stride * sizeof(int64_t), e->tokens, e->max_G * sizeof(int64_t),
                              std::min<int64_t>(g_stride, e->max_G) * sizeof(int64_t), B, hipMemcpyDeviceToDevice, st));
  E_HIP(e, hipStreamSynchronize(st));
  for (int i = 0; i < RING; ++i)
    if (evs[i]) (void)hipEventDestroy(evs[i]);
  if (e->opt_profile) {
    for (size_t i = 0; i < e->prof_tags.size(); ++i) {
      float pm = 0.f;
      if (hipEventElapsedTime(&pm, e->prof_pool[2 * i], e->prof_pool[2 * i + 1]) == hipSuccess) {
        e->prof_ms[e->prof_tags[i]] += pm;
        e->prof_calls[e->prof_tags[i]] += 1;
      }
    }
  }
  memcpy(st_host.data(), e->tables_host, st_host.size() * sizeof(int32_t));
  e->G_len.resize(B);
  bool no_token = false, not_done = false;
  for (int b = 0; b < B; ++b) {
    e->G_len[b] = st_host[2 * e->max_B + b];
    gen_lens[b] = e->G_len[b];
    if (e->G_len[b] == 0 && !e->bos && !forced && max_new <= 0) no_token = true;
    if (!st_host[3 * e->max_B + b]) not_done = true;
  }
  e->n_steps = std::min(steps_done, bound);
  e->n_ar_launches = launches;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e->ev_t[2], e->ev_t[3]) == hipSuccess) e->t_ar = ms;
  if (hipEventElapsedTime(&ms, e->ev_t[0], e->ev_t[1]) == hipSuccess) e->t_prefill = ms;
  e->have_gen = true;
  if ((r = leave(e, stream))) return r;
  if (e->poll_host[POLL_IDERR] != 0) return e->fail(VLE_EINDEX, "forced token id outside the audio vocabulary (the reference's nn.Embedding raises IndexError)");
  if (ps_call) {
    e->ps_last_fail = (unsigned)e->poll_host[POLL_PSFAIL];
    if (e->ps_last_fail != 0) {
      ++e->ps_fallbacks;
      e->ps_backoff = e->ps_backoff_next;
      e->ps_backoff_next = std::min(64, 2 * e->ps_backoff_next);
      e->have_prefill = e->have_gen = false;  // the KV cache and the token history of this call are not to be used
      return e->fail(VLE_EBUSY, "the persistent AR launch could not keep the whole GPU (a wave gave up waiting for an in-launch hand-off: GPU shared "
                                "with another workload?): this call's results are invalid; repeat vle_ar_prefill + vle_ar_generate -- the engine "
                                "runs the launch chain for its next batch-1 calls and re-arms the persistent launch by itself");
    }
    e->ps_backoff_next = 2;
  }
  if (not_done) return e->fail(VLE_ESTATE, "AR loop ended with unfinished utterances (capacity too small?)");
  if (!ps_call && B <= PSB_MAX && !e->slot_mode && e->ps_backoff > 0) --e->ps_backoff;  // one more batch-1 call on the chain after VLE_EBUSY; at 0 the persistent launch is re-armed
  if (no_token) return e->fail(VLE_ENOTOKEN, "well trained model shouldn't reach here.");
  return VLE_OK;
}

// The NAR stages on (text, prompt, first codebook).  `mode`: prefix mode to apply (continual() maps
// 2/4 to 1); drop[b] = enrolled_len - 2 for prefix_mode 2/4 inference, else 0.
// `slots` (slot API): the utterances to decode, as slot ids; every per-utterance table is then indexed by slot id
// (sized max_B) except the attention's sequence offsets, which follow the order of the list.  null = utterances 0..B-1.
static int run_nar(vle_engine* e, const std::vector<int32_t>& drop, int mode, int64_t* codes, int64_t g_stride,
                   const std::vector<int32_t>* slots = nullptr) {
  hipStream_t st = e->st;
  const int Q = e->Q, d = e->d;
  const int nseq = slots ? (int)slots->size() : e->B;  // sequences in this pass
  const int B = slots ? e->max_B : e->B;                // extent of the per-utterance tables
  auto sl = [&](int i) { return slots ? (*slots)[i] : i; };
  struct NseqScope {  // enqueue_layer_rows launches the attention over e->nseq sequences
    vle_engine* e;
    ~NseqScope() { e->nseq = 0; }
  } nseq_scope{e};
  e->nseq = nseq;
  int r;
  TableBuilder tb{e};
  std::vector<int> Sn(B), N(B);
  int64_t xrows = 0, arows = 0, grows = 0;
  int max_len = 0;
  for (int i = 0; i < nseq; ++i) {
    const int b = sl(i);
    Sn[b] = e->S_len[b] - drop[b];
    N[b] = Sn[b] + e->P_len[b] + e->G_len[b];
    xrows += N[b];
    arows += e->P_len[b] + e->G_len[b];
    grows += e->G_len[b];
    max_len = std::max(max_len, N[b]);
  }
  e->sumG_last = grows;
  E_HIP(e, hipEventRecord(e->ev_t[4], st));
  if (grows == 0 || Q == 1) {
    if (Q == 1 && grows > 0) {
      // only the first codebook exists (valle.py:1060-1061)
      int32_t *h_gs, *h_gp;
      int32_t* d_gs = tb.take(grows, &h_gs);
      int32_t* d_gp = tb.take(grows, &h_gp);
      if (!d_gs || !d_gp) return e->fail(VLE_EINVAL, "table overflow");
      int64_t o = 0;
      for (int i = 0; i < nseq; ++i) {
        const int b = sl(i);
        for (int g = 0; g < e->G_len[b]; ++g, ++o) {
          h_gs[o] = b;
          h_gp[o] = g;
        }
      }
      E_HIP(e, hipMemcpyAsync(e->tables_dev, e->tables_host, tb.used * sizeof(int32_t), hipMemcpyHostToDevice, st));
      E_LAUNCH(e, launch_codes_set_first(st, e->tokens, e->max_G, d_gs, d_gp, grows, codes, g_stride, Q));
    }
    E_HIP(e, hipEventRecord(e->ev_t[5], st));
    return 0;
  }
  int32_t *h_tl, *h_td, *h_pl, *h_gl, *h_aoff, *h_xoff, *h_as, *h_ap, *h_xs, *h_xp, *h_gs, *h_gp, *h_gmap, *h_tl_zero;
  int32_t* d_tl = tb.take(B, &h_tl);
  int32_t* d_td = tb.take(B, &h_td);
  int32_t* d_pl = tb.take(B, &h_pl);
  int32_t* d_gl = tb.take(B, &h_gl);
  int32_t* d_aoff = tb.take(B + 1, &h_aoff);
  int32_t* d_xoff = tb.take(nseq + 1, &h_xoff);
  int32_t* d_as = tb.take(arows, &h_as);
  int32_t* d_ap = tb.take(arows, &h_ap);
  int32_t* d_xs = tb.take(xrows, &h_xs);
  int32_t* d_xp = tb.take(xrows, &h_xp);
  int32_t* d_gs = tb.take(grows, &h_gs);
  int32_t* d_gp = tb.take(grows, &h_gp);
  int32_t* d_gmap = tb.take(grows, &h_gmap);
  int32_t* d_tl_zero = tb.take(nseq, &h_tl_zero);
  if (!d_tl || !d_td || !d_pl || !d_gl || !d_aoff || !d_xoff || !d_as || !d_ap || !d_xs || !d_xp || !d_gs || !d_gp || !d_gmap ||
      !d_tl_zero)
    return e->fail(VLE_EINVAL, "table overflow");
  int64_t xo = 0, ao = 0, go = 0;
  if (slots) {
    memset(h_tl, 0, B * sizeof(int32_t)); memset(h_td, 0, B * sizeof(int32_t)); memset(h_pl, 0, B * sizeof(int32_t));
    memset(h_gl, 0, B * sizeof(int32_t)); memset(h_aoff, 0, (B + 1) * sizeof(int32_t));
  }
  for (int i = 0; i < nseq; ++i) {
    const int b = sl(i);
    h_tl[b] = Sn[b]; h_td[b] = drop[b]; h_pl[b] = e->P_len[b]; h_gl[b] = e->G_len[b]; h_tl_zero[i] = 0;
    h_aoff[b] = (int32_t)ao; h_xoff[i] = (int32_t)xo;
    for (int a = 0; a < e->P_len[b] + e->G_len[b]; ++a) {
      h_as[ao + a] = b;
      h_ap[ao + a] = a;
    }
    for (int p = 0; p < N[b]; ++p) {
      h_xs[xo + p] = b;
      h_xp[xo + p] = p;
    }
    for (int g = 0; g < e->G_len[b]; ++g) {
      h_gs[go + g] = b;
      h_gp[go + g] = g;
      h_gmap[go + g] = (int32_t)(xo + Sn[b] + e->P_len[b] + g);  // xy_dec[:, text_len + prefix_len:] (valle.py:1128)
    }
    ao += e->P_len[b] + e->G_len[b]; xo += N[b]; go += e->G_len[b];
  }
  h_aoff[B] = (int32_t)ao; h_xoff[nseq] = (int32_t)xo;
  E_HIP(e, hipMemcpyAsync(e->tables_dev, e->tables_host, tb.used * sizeof(int32_t), hipMemcpyHostToDevice, st));

  NarEmbedArgs na{};
  na.t = NarSeqTables{d_tl, d_td, d_pl, d_gl, d_aoff, d_xoff};
  na.text = e->text_ids; na.s_stride = e->max_S; na.prompt = e->prompt_codes; na.p_stride = e->max_P + e->max_G; na.Q = Q;
  na.first_cb = e->tokens; na.g_stride = e->max_G;
  na.arow_seq = d_as; na.arow_pos = d_ap; na.xrow_seq = d_xs; na.xrow_pos = d_xp;
  na.audio_embs = e->nar_audio_emb_tab; na.text_emb = e->nar_text_emb; na.pe = e->pe;
  na.alpha_text = e->alphas + 2; na.alpha_audio = e->alphas + 3; na.d = d; na.arows = arows; na.xrows = xrows;
  na.y_emb = e->yemb; na.x = e->X;

  E_LAUNCH(e, launch_codes_set_first(st, e->tokens, e->max_G, d_gs, d_gp, grows, codes, g_stride, Q));
  E_LAUNCH(e, launch_nar_yemb_init(st, na, mode != 0 ? 1 : 0));
  for (int i = 0; i < Q - 1; ++i) {
    E_LAUNCH(e, launch_nar_assemble(st, na));
    const bool nar_fold = use_ln_fold(e, xrows) && e->nar_fold != nullptr;
    for (int l = 0; l < e->L; ++l) {
      LnFoldLayer f;
      const float* fc = nar_fold ? e->nar_fold + ((size_t)i * e->L + l) * 14 * d : nullptr;
      f.sg_qkv = fc; f.tb_qkv = fc + 3 * d; f.sg_1 = fc + 6 * d; f.tb_1 = fc + 10 * d;
      f.in_folded = l > 0;
      f.next_g1 = l + 1 < e->L ? e->nar_gamma[i][2 * (l + 1)] : nullptr;
      if ((r = enqueue_layer_rows(e, e->nar[l], e->nar_gamma[i][2 * l], e->nar_beta[i][2 * l], e->nar_gamma[i][2 * l + 1],
                                  e->nar_beta[i][2 * l + 1], xrows, d_xoff, d_tl_zero, max_len, 0, nullptr, nullptr, nullptr, nullptr,
                                  nar_fold ? &f : nullptr)))
        return r;
    }
    // final AdaLN only on the generated rows, then nar_predict_layers[i] (valle.py:1128)
    E_LAUNCH(e, launch_layernorm(st, e->dtype, e->X, d_gmap, e->nar_gamma[i][2 * e->L], e->nar_beta[i][2 * e->L], e->Xn, grows, d));
    float* lg = e->nar_logits;
    if (e->opt_trace_nar && e->trace_nar) lg = e->trace_nar + (size_t)i * e->max_B * e->max_G * NUM_AUDIO_TOKENS;
    E_LAUNCH(e, launch_gemm(st, e->dtype, e->Xn, e->nar_predict[i], nullptr, lg, nullptr, grows, NUM_AUDIO_TOKENS, d, EPI_F32));
    NarArgmaxArgs aa{};
    aa.logits = lg; aa.V = NUM_AUDIO_TOKENS; aa.rows = grows; aa.grow_seq = d_gs; aa.grow_pos = d_gp; aa.aoff = d_aoff;
    aa.prompt_len = d_pl; aa.codes = codes; aa.g_stride = g_stride; aa.Q = Q; aa.col = i + 1;
    aa.next_emb = i < Q - 2 ? e->nar_audio_emb[i + 1] : nullptr;  // valle.py:1133-1134
    aa.y_emb = e->yemb; aa.d = d;
    aa.forced = e->nar_forced; aa.f_stride = e->nar_forced_stride;
    E_LAUNCH(e, launch_nar_argmax(st, aa));
    if (mode == 0 && i < Q - 2) E_LAUNCH(e, launch_nar_yemb_add_prompt(st, na, i + 1));  // valle.py:1104-1107
  }
  E_HIP(e, hipEventRecord(e->ev_t[5], st));
  return 0;
}

static int finish_nar_timing(vle_engine* e) {
  E_HIP(e, hipStreamSynchronize(e->st));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e->ev_t[4], e->ev_t[5]) == hipSuccess) e->t_nar = ms;
  return 0;
}

extern "C" int vle_nar_decode(vle_engine* e, void* stream, const int32_t* enroll_lens, int64_t* codes, int64_t g_stride) {
  if (!e) return VLE_EINVAL;
  // vle_nar_force is one-shot: whatever way this call ends (argument errors included) the caller-owned pointer is forgotten
  struct ForcedGuard {
    vle_engine* e;
    ~ForcedGuard() { e->nar_forced = nullptr; e->nar_forced_stride = 0; }
  } forced_guard{e};
  if (!e->have_gen) return e->fail(VLE_ESTATE, "vle_nar_decode needs vle_ar_generate first");
  if (!codes) return e->fail(VLE_EINVAL, "codes is null");
  const int mode = e->cfg.prefix_mode;
  std::vector<int32_t> drop(e->B, 0);
  if (mode == 2 || mode == 4) {
    if (!enroll_lens) return e->fail(VLE_EINVAL, "prefix_mode 2/4 needs enroll_lens (valle.py:1068-1079)");
    for (int b = 0; b < e->B; ++b) {
      // text = cat(text[:, :1], text[:, enrolled_len-1:]); text_len -= enrolled_len - 2
      if (enroll_lens[b] < 2 || enroll_lens[b] - 1 > e->S_len[b]) return e->fail(VLE_EINVAL, "enroll_lens out of range");
      drop[b] = enroll_lens[b] - 2;
    }
  }
  for (int b = 0; b < e->B; ++b) {
    if (e->G_len[b] > g_stride) return e->fail(VLE_EINVAL, "g_stride smaller than generated length");
    if (e->nar_forced && e->G_len[b] > e->nar_forced_stride) return e->fail(VLE_EINVAL, "vle_nar_force: f_stride smaller than generated length");
  }
  int r;
  if ((r = enter(e, stream))) return r;
  if ((r = run_nar(e, drop, mode, codes, g_stride))) return r;
  if ((r = finish_nar_timing(e))) return r;
  return leave(e, stream);
}

extern "C" int vle_nar_force(vle_engine* e, const int64_t* forced_codes, int64_t f_stride) {
  if (!e) return VLE_EINVAL;
  if (forced_codes && f_stride < 1) return e->fail(VLE_EINVAL, "f_stride must be >= 1");
  e->nar_forced = forced_codes;
  e->nar_forced_stride = f_stride;
  return VLE_OK;
}

extern "C" int vle_nar_continual(vle_engine* e, void* stream, const int64_t* text, int64_t s_stride, const int32_t* text_lens,
                                 const int64_t* y_codes, int64_t t_stride, const int32_t* y_lens, int32_t B, int64_t* codes,
                                 int64_t g_stride, int32_t* gen_lens) {
  if (!e) return VLE_EINVAL;
  if (!e->finalized) return e->fail(VLE_ESTATE, e->broken ? "engine is unusable after a failed vle_reserve: destroy it" : "weights not finalized");
  e->nar_forced = nullptr; e->nar_forced_stride = 0;  // vle_nar_force applies to vle_nar_decode only
  if (e->Q != 8) return e->fail(VLE_EINVAL, "continual() asserts num_quantizers == 8 (valle.py:1160)");
  if (!text || !text_lens || !y_codes || !y_lens || !codes || !gen_lens) return e->fail(VLE_EINVAL, "null argument");
  if (B < 1 || B > e->max_B) return e->fail(VLE_EINVAL, "batch exceeds max_batch");
  // validate into locals; the engine's per-call state is committed only after every check passed
  std::vector<int32_t> S_new(B, 0), P_new(B, 0), G_new(B, 0);
  for (int b = 0; b < B; ++b) {
    if (text_lens[b] < 1 || text_lens[b] > e->max_S || text_lens[b] > s_stride) return e->fail(VLE_EINVAL, "text_lens out of range");
    if (y_lens[b] < 1 || y_lens[b] > t_stride || y_lens[b] > e->max_P + e->max_G) return e->fail(VLE_EINVAL, "y_lens out of range");
    const int prefix = std::min((int)(y_lens[b] * 0.5), 3 * 75);  // valle.py:1173
    S_new[b] = text_lens[b]; P_new[b] = prefix; G_new[b] = y_lens[b] - prefix;
    if (P_new[b] > e->max_P + e->max_G || G_new[b] > e->max_G || G_new[b] > g_stride) return e->fail(VLE_EINVAL, "capacity exceeded");
  }
  int r;
  if ((r = enter(e, stream))) return r;
  hipStream_t st = e->st;
  // commit point: this call replaces whatever prefill / generate / slot state the engine held
  e->B = B;
  e->slot_mode = false;
  e->have_prefill = e->have_gen = false;
  e->S_len = S_new; e->P_len = P_new; e->G_len = G_new;
  for (int b = 0; b < B; ++b) gen_lens[b] = G_new[b];
  E_HIP(e, hipMemcpy2DAsync(e->text_ids, e->max_S * sizeof(int64_t), text, s_stride * sizeof(int64_t),
                            std::min<int64_t>(s_stride, e->max_S) * sizeof(int64_t), B, hipMemcpyDeviceToDevice, st));
  const int64_t pp = (int64_t)(e->max_P + e->max_G) * 8;
  E_HIP(e, hipMemcpy2DAsync(e->prompt_codes, pp * sizeof(int64_t), y_codes, t_stride * 8 * sizeof(int64_t),
                            std::min<int64_t>(t_stride, e->max_P + e->max_G) * 8 * sizeof(int64_t), B, hipMemcpyDeviceToDevice, st));
  {  // id range check on the engine-owned copies: all 8 codebooks of the prefix rows, the first codebook of every row
    TableBuilder tb0{e};
    int32_t *h_s, *h_p, *h_t;
    int32_t* d_s = tb0.take(B, &h_s);
    int32_t* d_p = tb0.take(B, &h_p);
    int32_t* d_t = tb0.take(B, &h_t);
    if (!d_s || !d_p || !d_t) return e->fail(VLE_EINVAL, "table overflow");
    for (int b = 0; b < B; ++b) {
      h_s[b] = S_new[b]; h_p[b] = P_new[b]; h_t[b] = y_lens[b];
    }
    E_HIP(e, hipMemcpyAsync(e->tables_dev, e->tables_host, tb0.used * sizeof(int32_t), hipMemcpyHostToDevice, st));
    E_HIP(e, hipMemsetAsync(e->id_err_dev, 0, sizeof(int32_t), st));
    E_LAUNCH(e, launch_check_ids(st, e->text_ids, e->max_S, 1, 1, d_s, nullptr, B, e->max_S, NUM_TEXT_TOKENS, 0, e->id_err_dev, 1));
    E_LAUNCH(e, launch_check_ids(st, e->prompt_codes, pp, 8, 8, d_p, nullptr, B, e->max_P + e->max_G, V_AR, NUM_AUDIO_TOKENS, e->id_err_dev, 2));
    E_LAUNCH(e, launch_check_ids(st, e->prompt_codes, pp, 8, 1, d_t, nullptr, B, e->max_P + e->max_G, V_AR, 0, e->id_err_dev, 2));
    E_HIP(e, hipMemcpyAsync(e->poll_host + POLL_IDERR, e->id_err_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    E_HIP(e, hipStreamSynchronize(st));  // run_nar below rebuilds the pinned table mirror
    if (e->poll_host[POLL_IDERR] != 0) {
      (void)leave(e, stream);
      return e->fail(VLE_EINDEX, kIdErrMsg);
    }
  }
  // first codebook of the continuation: codes = [y[:, prefix_len:, 0]] (valle.py:1178), from the checked copy
  for (int b = 0; b < B; ++b)
    if (e->G_len[b] > 0)
      E_HIP(e, hipMemcpy2DAsync(e->tokens + (size_t)b * e->max_G, sizeof(int64_t),
                                e->prompt_codes + ((size_t)b * pp + (size_t)e->P_len[b] * 8), 8 * sizeof(int64_t), sizeof(int64_t),
                                e->G_len[b], hipMemcpyDeviceToDevice, st));
  std::vector<int32_t> drop(B, 0);
  const int mode = e->cfg.prefix_mode == 0 ? 0 : 1;
  if ((r = run_nar(e, drop, mode, codes, g_stride))) return r;
  if ((r = finish_nar_timing(e))) return r;
  return leave(e, stream);
}

// =================================================================================================
// hooks
// =================================================================================================
// ---- slot API: continuous batching -----------------------------------------------------------
// The engine's max_batch utterance positions become SLOTS, each free or holding one utterance.  The caller admits new
// utterances into free slots (prefill of just those), advances every live slot by a few AR steps, and harvests finished
// slots (their 7 NAR stages), so a finished utterance frees its slot -- and its share of the KV stream, which the decode
// attention skips for done slots -- while the others keep decoding.  Same kernels, same captured step graph
// (B = max_batch), same numerics as vle_ar_prefill / vle_ar_generate / vle_nar_decode: an utterance's tokens do not
// depend on what shares the batch with it.

static int set_dyn(vle_engine* e, int32_t top_k, float temperature, uint64_t seed, int32_t max_new) {
  ArDyn dyn{};
  dyn.top_k = top_k; dyn.temperature = temperature; dyn.seed = seed; dyn.max_new = max_new; dyn.ignore_eos = e->opt_ignore_eos ? 1 : 0;
  dyn.forced_len = e->forced_len_dev;
  int32_t* hp = e->tables_host + e->tables_cap - (int64_t)(sizeof(ArDyn) / 4 + e->max_B + 4);
  memcpy(hp, &dyn, sizeof(ArDyn));
  E_HIP(e, hipMemcpyAsync(e->dyn_dev, hp, sizeof(ArDyn), hipMemcpyHostToDevice, e->st));
  return 0;
}

extern "C" int vle_slots_begin(vle_engine* e, void* stream) {
  if (!e) return VLE_EINVAL;
  if (!e->finalized) return e->fail(VLE_ESTATE, e->broken ? "engine is unusable after a failed vle_reserve: destroy it" : "weights not finalized");
  int r;
  if ((r = enter(e, stream))) return r;
  e->B = e->max_B;
  e->nsplit = e->opt_nsplit > 0 ? e->opt_nsplit : choose_nsplit(e, e->B);
  e->slot_mode = true;
  e->admitted = 0;
  e->have_prefill = e->have_gen = false;
  e->S_len.assign(e->max_B, 0);
  e->P_len.assign(e->max_B, 0);
  e->G_len.assign(e->max_B, 0);
  e->slot_state.assign(6 * e->max_B + 8, 0);
  for (int b = 0; b < e->max_B; ++b) e->slot_state[3 * e->max_B + b] = 1;  // every slot free = done
  memcpy(e->tables_host, e->slot_state.data(), e->slot_state.size() * sizeof(int32_t));
  E_HIP(e, hipMemcpyAsync(e->state_dev, e->tables_host, e->slot_state.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->st));
  // free slots run through the (row-independent) GEMMs of every step: give them finite inputs
  E_HIP(e, hipMemsetAsync(e->x_step, 0, (size_t)e->max_B * e->d * sizeof(float), e->st));
  // after VLE_EBUSY the engine stays on the launch chain for `ps_backoff` calls: a slot session counts as one (the session that follows
  // the failed one runs the chain, the one after it the persistent launch again; doubling while it keeps happening)
  if (e->ps_backoff > 0 && e->max_B >= 2 && e->max_B <= PSB_MAX) --e->ps_backoff;
  if ((r = persist_prepare(e))) return r;  // 2 .. PSB_MAX slots: the steps run the batched persistent launch (slot_persist_ready)
  return leave(e, stream);
}

extern "C" int vle_slots_prefill(vle_engine* e, void* stream, int32_t n, const int32_t* slots, const int64_t* text, int64_t s_stride,
                                 const int32_t* text_lens, const int64_t* prompt_codes, int64_t p_stride, const int32_t* prompt_lens,
                                 int32_t top_k, float temperature, uint64_t seed) {
  if (!e) return VLE_EINVAL;
  if (!e->slot_mode) return e->fail(VLE_ESTATE, "vle_slots_prefill needs vle_slots_begin first");
  if (n < 1 || n > e->max_B || !slots || !text || !text_lens || !prompt_codes || !prompt_lens) return e->fail(VLE_EINVAL, "bad argument");
  if (!(temperature > 0.f)) return e->fail(VLE_EINVAL, "temperature must be > 0");
  std::vector<char> seen(e->max_B, 0);
  for (int i = 0; i < n; ++i) {
    const int b = slots[i];
    if (b < 0 || b >= e->max_B || seen[b]) return e->fail(VLE_EINVAL, "slot id out of range or listed twice");
    seen[b] = 1;
    if (!e->slot_state[3 * e->max_B + b]) return e->fail(VLE_ESTATE, "slot still holds a live utterance");
    if (text_lens[i] < 1 || text_lens[i] > e->max_S || text_lens[i] > s_stride) return e->fail(VLE_EINVAL, "text_lens out of range");
    if (prompt_lens[i] < 0 || prompt_lens[i] > e->max_P || prompt_lens[i] > p_stride) return e->fail(VLE_EINVAL, "prompt_lens out of range");
    if (prompt_lens[i] + e->bos < 1) return e->fail(VLE_EINVAL, "empty prompt needs prepend_bos");
  }
  int r;
  if ((r = enter(e, stream))) return r;
  hipStream_t st = e->st;
  // engine-owned copies of the inputs, at the slot's row (the NAR phase needs them again)
  const int64_t pp = (int64_t)(e->max_P + e->max_G) * e->Q;
  for (int i = 0; i < n; ++i) {
    const int b = slots[i];
    E_HIP(e, hipMemcpyAsync(e->text_ids + (int64_t)b * e->max_S, text + (int64_t)i * s_stride, text_lens[i] * sizeof(int64_t),
                            hipMemcpyDeviceToDevice, st));
    if (prompt_lens[i] > 0)
      E_HIP(e, hipMemcpyAsync(e->prompt_codes + (int64_t)b * pp, prompt_codes + (int64_t)i * p_stride * e->Q,
                              (size_t)prompt_lens[i] * e->Q * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    e->S_len[b] = text_lens[i];
    e->P_len[b] = prompt_lens[i];
    e->G_len[b] = 0;
  }
  TableBuilder tb{e};
  int64_t rows = 0;
  int max_len = 0;
  for (int i = 0; i < n; ++i) {
    const int len = text_lens[i] + e->bos + prompt_lens[i];
    rows += len;
    max_len = std::max(max_len, len);
  }
  int32_t *h_seq_off, *h_tl_seq, *h_tl_slot, *h_row_seq, *h_row_pos, *h_last, *h_slots, *h_kv, *h_ap, *h_cap, *h_pl_seq;
  int32_t* d_seq_off = tb.take(n + 1, &h_seq_off);
  int32_t* d_tl_seq = tb.take(n, &h_tl_seq);            // attention: by sequence order
  int32_t* d_pl_seq = tb.take(n, &h_pl_seq);
  int32_t* d_tl_slot = tb.take(e->max_B, &h_tl_slot);   // embedding: by slot id (row_seq holds slot ids)
  int32_t* d_row_seq = tb.take(rows, &h_row_seq);
  int32_t* d_row_pos = tb.take(rows, &h_row_pos);
  int32_t* d_last = tb.take(n, &h_last);
  int32_t* d_slots = tb.take(n, &h_slots);
  int32_t* d_kv = tb.take(n, &h_kv);
  int32_t* d_ap = tb.take(n, &h_ap);
  int32_t* d_cap = tb.take(n, &h_cap);
  if (!d_seq_off || !d_tl_seq || !d_pl_seq || !d_tl_slot || !d_row_seq || !d_row_pos || !d_last || !d_slots || !d_kv || !d_ap || !d_cap)
    return e->fail(VLE_EINVAL, "table overflow");
  for (int b = 0; b < e->max_B; ++b) h_tl_slot[b] = e->S_len[b];
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    const int b = slots[i], len = text_lens[i] + e->bos + prompt_lens[i];
    h_seq_off[i] = (int32_t)off;
    h_tl_seq[i] = text_lens[i];
    h_pl_seq[i] = prompt_lens[i];
    for (int p = 0; p < len; ++p) {
      h_row_seq[off + p] = b;
      h_row_pos[off + p] = p;
    }
    off += len;
    h_last[i] = (int32_t)(off - 1);
    h_slots[i] = b;
    h_kv[i] = len;                            // kv_len: next free cache position
    h_ap[i] = e->bos + prompt_lens[i];        // audio_pos of the next token
    h_cap[i] = 16 * text_lens[i];             // valle.py:1047
  }
  h_seq_off[n] = (int32_t)off;
  E_HIP(e, hipMemcpyAsync(e->tables_dev, e->tables_host, tb.used * sizeof(int32_t), hipMemcpyHostToDevice, st));
  // id range check on the slots' copies BEFORE any slot state changes on the device
  E_HIP(e, hipMemsetAsync(e->id_err_dev, 0, sizeof(int32_t), st));
  E_LAUNCH(e, launch_check_ids(st, e->text_ids, e->max_S, 1, 1, d_tl_seq, d_slots, n, e->max_S, NUM_TEXT_TOKENS, 0, e->id_err_dev, 1));
  E_LAUNCH(e, launch_check_ids(st, e->prompt_codes, pp, e->Q, e->Q, d_pl_seq, d_slots, n, e->max_P, V_AR, NUM_AUDIO_TOKENS, e->id_err_dev, 2));
  E_HIP(e, hipMemcpyAsync(e->poll_host + POLL_IDERR, e->id_err_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  E_HIP(e, hipStreamSynchronize(st));
  if (e->poll_host[POLL_IDERR] != 0) {
    for (int i = 0; i < n; ++i) e->S_len[slots[i]] = 0;  // the slots stay free
    (void)leave(e, stream);
    return e->fail(VLE_EINDEX, kIdErrMsg);
  }
  E_LAUNCH(e, launch_slot_state_init(st, e->state_dev, e->max_B, d_slots, d_kv, d_ap, d_cap, n, e->slot_seed_dev, seed, e->admitted));
  if (e->qgran && e->max_B == 1) E_HIP(e, hipMemsetAsync(e->qgran, 0, (size_t)e->L * e->d * sizeof(unsigned long long), st));  // see vle_ar_prefill
  e->admitted += (unsigned long long)n;

  PrefillEmbedArgs pa{};
  pa.text = e->text_ids; pa.s_stride = e->max_S; pa.prompt = e->prompt_codes; pa.p_stride = e->max_P + e->max_G; pa.Q = e->Q;
  pa.text_len = d_tl_slot; pa.row_seq = d_row_seq; pa.row_pos = d_row_pos;
  pa.text_emb = e->ar_text_emb; pa.audio_emb = e->ar_audio_emb; pa.pe = e->pe;
  pa.alpha_text = e->alphas + 0; pa.alpha_audio = e->alphas + 1; pa.bos = e->bos; pa.d = e->d; pa.rows = rows; pa.x = e->X;
  E_LAUNCH(e, launch_prefill_embed(st, pa));
  e->nseq = n;
  for (int l = 0; l < e->L && r == 0; ++l) {
    const LayerW& w = e->ar[l];
    r = enqueue_layer_rows(e, w, w.g1, w.be1, w.g2, w.be2, rows, d_seq_off, d_tl_seq, max_len, 1, cache_layer(e, e->kcache, l),
                           cache_layer(e, e->vcache, l), d_row_seq, d_row_pos);
  }
  e->nseq = 0;
  if (r) return r;
  // bring the new slots to the state every live slot is in between steps ("x_step = embedding of the next input"):
  // last prefill row -> x_step[slot]; logits of all rows (the live slots' logits are scratch at this point: a step
  // rewrites them before sampling); first sample for the new slots only
  E_LAUNCH(e, launch_scatter_rows(st, e->X, d_last, e->x_step, d_slots, n, e->d));
  if ((r = enqueue_ar_logits(e))) return r;
  if ((r = set_dyn(e, top_k, temperature, seed, 0))) return r;
  if ((r = enqueue_ar_sample(e, 1, d_slots, n))) return r;
  E_HIP(e, hipStreamSynchronize(st));  // the pinned table mirror is reused by the next call
  for (int i = 0; i < n; ++i) e->slot_state[3 * e->max_B + slots[i]] = 0;
  return leave(e, stream);
}

extern "C" int vle_slots_step(vle_engine* e, void* stream, int32_t nsteps, int32_t top_k, float temperature, uint64_t seed,
                              int32_t* done_out, int32_t* gen_lens_out) {
  if (!e) return VLE_EINVAL;
  if (!e->slot_mode) return e->fail(VLE_ESTATE, "vle_slots_step needs vle_slots_begin first");
  if (nsteps < 0 || !(temperature > 0.f)) return e->fail(VLE_EINVAL, "bad argument");
  int r;
  if ((r = enter(e, stream))) return r;
  hipStream_t st = e->st;
  if ((r = set_dyn(e, top_k, temperature, seed, 0))) return r;
  const int B = e->B;
  const int spg = e->opt_spg > 0 ? e->opt_spg : (e->cfg.steps_per_graph > 0 ? e->cfg.steps_per_graph : 8);
  const bool use_graph = e->cfg.use_graph != 0 && !e->opt_profile;
  int steps_done = 0;
  const bool ps_call = slot_persist_ready(e) && nsteps > 0;
  e->ps_last_call = ps_call;
  if (ps_call) {
    // ONE launch per call runs the nsteps iterations of every live slot, sampling and stop rule included (persist_nb.hip); a slot
    // that stops inside the launch stays in it with a frozen cache position
    if (e->qa_spin_fail) E_HIP(e, hipMemsetAsync(e->qa_spin_fail + 2, 0, sizeof(unsigned), st));
    e->kt_idx = 0;
    while (steps_done < nsteps) {
      const int n = std::min(nsteps - steps_done, 4096);
      if ((r = enqueue_persist_step(e, n))) return r;
      steps_done += n;
    }
    e->poll_host[POLL_PSFAIL] = 0;
    if (e->dbg_inject_psfail > 0 && e->qa_spin_fail) {  // test hook: as if one wave had given up
      --e->dbg_inject_psfail;
      E_HIP(e, hipMemsetAsync(e->qa_spin_fail + 2, 1, 1, st));
    }
    if (e->qa_spin_fail) E_HIP(e, hipMemcpyAsync(e->poll_host + POLL_PSFAIL, e->qa_spin_fail + 2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  }
  hipGraphExec_t g_multi = nullptr, g_single = nullptr;
  if (use_graph && nsteps > 0 && !ps_call) {
    auto it = e->graphs.find(B * 64 + e->nsplit);
    if (it == e->graphs.end()) {
      if ((r = enqueue_ar_step(e))) return r;  // first step eagerly (kernel attributes are set outside capture)
      steps_done = 1;
      if ((r = capture_graph(e, spg, &g_multi))) return r;
      if ((r = capture_graph(e, 1, &g_single))) return r;
      e->graphs[B * 64 + e->nsplit] = {g_multi, g_single};
    } else {
      g_multi = it->second.first;
      g_single = it->second.second;
    }
  }
  while (steps_done < nsteps) {
    if (use_graph && nsteps - steps_done >= spg) {
      E_HIP(e, hipGraphLaunch(g_multi, st));
      steps_done += spg;
    } else if (use_graph) {
      E_HIP(e, hipGraphLaunch(g_single, st));
      steps_done += 1;
    } else {
      if ((r = enqueue_ar_step(e))) return r;
      steps_done += 1;
    }
  }
  E_HIP(e, hipMemcpyAsync(e->tables_host, e->state_dev, e->slot_state.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  E_HIP(e, hipStreamSynchronize(st));
  memcpy(e->slot_state.data(), e->tables_host, e->slot_state.size() * sizeof(int32_t));
  for (int b = 0; b < e->max_B; ++b) {
    e->G_len[b] = e->slot_state[2 * e->max_B + b];
    if (done_out) done_out[b] = e->slot_state[3 * e->max_B + b];
    if (gen_lens_out) gen_lens_out[b] = e->G_len[b];
  }
  if (ps_call) {
    e->ps_last_fail = (unsigned)e->poll_host[POLL_PSFAIL];
    if (e->ps_last_fail != 0) {  // the launch could not keep the whole GPU: the slots' state is void
      ++e->ps_fallbacks;
      e->ps_backoff = e->ps_backoff_next;
      e->ps_backoff_next = std::min(64, 2 * e->ps_backoff_next);
      e->slot_mode = false;
      (void)leave(e, stream);
      return e->fail(VLE_EBUSY, "the persistent AR launch could not keep the whole GPU (a wave gave up waiting for an in-launch hand-off: GPU shared "
                                "with another workload?): the slots' utterances are lost; call vle_slots_begin and admit them again -- the engine "
                                "runs the launch chain for its next calls and re-arms the persistent launch by itself");
    }
    e->ps_backoff_next = 2;
  }
  return leave(e, stream);
}

extern "C" int vle_slots_harvest(vle_engine* e, void* stream, int32_t n, const int32_t* slots, const int32_t* enroll_lens,
                                 int64_t* codes, int64_t g_stride) {
  if (!e) return VLE_EINVAL;
  e->nar_forced = nullptr; e->nar_forced_stride = 0;  // vle_nar_force applies to vle_nar_decode only
  if (!e->slot_mode) return e->fail(VLE_ESTATE, "vle_slots_harvest needs vle_slots_begin first");
  if (n < 1 || n > e->max_B || !slots || !codes) return e->fail(VLE_EINVAL, "bad argument");
  const int mode = e->cfg.prefix_mode;
  std::vector<int32_t> drop(e->max_B, 0), list(slots, slots + n);
  for (int i = 0; i < n; ++i) {
    const int b = slots[i];
    if (b < 0 || b >= e->max_B) return e->fail(VLE_EINVAL, "slot id out of range");
    if (e->S_len[b] < 1 || !e->slot_state[3 * e->max_B + b]) return e->fail(VLE_ESTATE, "slot is not a finished utterance");
    if (e->G_len[b] > g_stride) return e->fail(VLE_EINVAL, "g_stride smaller than generated length");
    if (mode == 2 || mode == 4) {
      if (!enroll_lens) return e->fail(VLE_EINVAL, "prefix_mode 2/4 needs enroll_lens (valle.py:1068-1079)");
      if (enroll_lens[i] < 2 || enroll_lens[i] - 1 > e->S_len[b]) return e->fail(VLE_EINVAL, "enroll_lens out of range");
      drop[b] = enroll_lens[i] - 2;
    }
  }
  int r;
  if ((r = enter(e, stream))) return r;
  if ((r = run_nar(e, drop, mode, codes, g_stride, &list))) return r;
  if ((r = finish_nar_timing(e))) return r;
  for (int i = 0; i < n; ++i) e->S_len[slots[i]] = 0;  // the slot is free again
  return leave(e, stream);
}

// Grow the engine's capacities in place: weights (and their packed / quantised copies) stay on the device, only the
// capacity-dependent buffers are re-created.  `pe` (HOST fp32 [pe_rows][d], the SinePositionalEmbedding table built by the
// caller like valle/modules/embedding.py:75-91) is needed when the position range grows; null = the engine's own restatement.
extern "C" int vle_reserve(vle_engine* e, int32_t max_batch, int32_t max_text, int32_t max_prompt, int32_t max_gen, const float* pe,
                           int64_t pe_rows) {
  if (!e) return VLE_EINVAL;
  if (max_batch < 1 || max_text < 1 || max_prompt < 0 || max_gen < 0) return e->fail(VLE_EINVAL, "bad capacity");
  if (e->broken) return e->fail(VLE_ESTATE, "engine is unusable after a failed vle_reserve: destroy it");
  const int nB = std::max(e->max_B, (int)max_batch), nS = std::max(e->max_S, (int)max_text), nP = std::max(e->max_P, (int)max_prompt);
  // like vle_create: the 16 * max_text + 1 default (the reference's length cap, valle.py:1047) applies only when the caller
  // names no generation capacity; an explicit max_gen is never inflated
  const int nG = std::max(e->max_G, max_gen > 0 ? (int)max_gen : 16 * nS + 1);
  if (!e->finalized) {  // nothing allocated yet
    e->max_B = nB; e->max_S = nS; e->max_P = nP; e->max_G = nG;
    e->cfg.max_batch = nB; e->cfg.max_text = nS; e->cfg.max_prompt = nP; e->cfg.max_gen = nG;
    e->ctx_max = nS + nP + 1 + nG;
    e->max_pos = std::max(nS, nP + 1 + nG) + 1;
    e->max_rows = (int64_t)nB * (nS + nP + 1 + nG);
    return VLE_OK;
  }
  if (nB == e->max_B && nS == e->max_S && nP == e->max_P && nG == e->max_G) return VLE_OK;
  E_HIP(e, hipSetDevice(e->cfg.device));
  E_HIP(e, hipStreamSynchronize(e->st));
  for (auto& kv : e->graphs) {
    if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
    if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
  }
  e->graphs.clear();
  // From here until the new buffers exist the engine owns nothing capacity-dependent: if any allocation below fails (growing is
  // exactly when memory runs out) the engine is marked broken -- finalized = false, every entry point answers VLE_ESTATE -- instead
  // of keeping pointers into freed memory.  The caller destroys it and builds a new one (valle_amd.Engine.reserve does).
  release_buffers(e);
  e->finalized = false;
  e->broken = true;
  const int old_pos = e->max_pos;
  e->max_B = nB; e->max_S = nS; e->max_P = nP; e->max_G = nG;
  e->cfg.max_batch = nB; e->cfg.max_text = nS; e->cfg.max_prompt = nP; e->cfg.max_gen = nG;
  e->ctx_max = nS + nP + 1 + nG;
  e->max_pos = std::max(nS, nP + 1 + nG) + 1;
  e->max_rows = (int64_t)nB * (nS + nP + 1 + nG);
  int r;
  if (e->max_pos > old_pos) {  // a longer sinusoid table (the old one stays allocated until vle_destroy: a few MB)
    std::vector<float> tab;
    if (pe != nullptr && pe_rows >= e->max_pos) tab.assign(pe, pe + (size_t)e->max_pos * e->d);
    else build_pe(tab, e->max_pos, e->d);
    e->in_buffers = false;
    float* new_pe = nullptr;
    r = upload_f32(e, &new_pe, tab.data(), tab.size());
    e->in_buffers = true;
    if (r) return r;
    e->pe = new_pe;
  }
  e->in_buffers = true;
  if ((r = alloc_buffers(e))) return r;
  if ((r = make_weight_packs(e))) return r;
  if (e->opt_trace_nar && e->Q > 1) {
    float* p = nullptr;
    if ((r = dev_alloc(e, &p, (size_t)(e->Q - 1) * e->max_B * e->max_G * NUM_AUDIO_TOKENS))) return r;
    e->trace_nar = p;
  }
  if (e->opt_ktrace) {
    unsigned long long* p = nullptr;
    if ((r = dev_alloc(e, &p, (size_t)KT_STEPS * KT_KERNELS * KT_WAVES * 4))) return r;
    e->ktrace_buf = p;
    (void)hipMemset(e->ktrace_buf, 0xFF, (size_t)KT_STEPS * KT_KERNELS * KT_WAVES * 4 * sizeof(unsigned long long));
  }
  e->broken = false;
  e->finalized = true;
  return VLE_OK;
}

extern "C" int vle_set_option(vle_engine* e, const char* name, int64_t value) {
  if (!e || !name) return VLE_EINVAL;
  const std::string n = name;
  if (n == "trace_ar_logits") {
    e->opt_trace_ar = value != 0;
    return VLE_OK;
  }
  if (n == "trace_nar_logits") {
    e->opt_trace_nar = value != 0;
    if (e->opt_trace_nar && !e->trace_nar && e->finalized && e->Q > 1) {
      float* p = nullptr;
      int r = dev_alloc(e, &p, (size_t)(e->Q - 1) * e->max_B * e->max_G * NUM_AUDIO_TOKENS);
      if (r) return r;
      e->trace_nar = p;
    }
    return VLE_OK;
  }
  if (n == "profile_kernels") {
    e->opt_profile = value != 0;
    const size_t want = value > 0 ? (size_t)value * 2 * (5 * e->L + 2) : 0;  // value = steps to cover
    while (e->prof_pool.size() < want) {
      hipEvent_t ev;
      if (hipEventCreate(&ev) != hipSuccess) return e->fail(VLE_EHIP, "hipEventCreate failed");
      e->prof_pool.push_back(ev);
    }
    return VLE_OK;
  }
  if (n == "host_prog") {
    e->opt_host_prog = value != 0;
    (void)hipStreamSynchronize(e->st);
    for (auto& kv : e->graphs) {
      if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
      if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
    }
    e->graphs.clear();
    return VLE_OK;
  }
  if (n == "qkv_attn" || n == "qa_nsplit" || n == "qa_qtemporal" || n == "qa_handoff" || n == "qa_nk") {  // changes the captured graphs: drop them
    if (n == "qa_nsplit" && !(value == 4 || value == 8 || value == 16)) return e->fail(VLE_EINVAL, "qa_nsplit must be 4, 8 or 16");
    if (n == "qa_nk" && !(value == 2 || value == 4 || value == 8)) return e->fail(VLE_EINVAL, "qa_nk must be 2, 4 or 8");
    (n == "qkv_attn" ? e->opt_qkv_attn : n == "qa_nsplit" ? e->opt_qa_nsplit : n == "qa_handoff" ? e->opt_qa_handoff : n == "qa_nk" ? e->opt_qa_nk : e->opt_qa_qtemporal) = (int)value;
    (void)hipStreamSynchronize(e->st);
    for (auto& kv : e->graphs) {
      if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
      if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
    }
    e->graphs.clear();
    return VLE_OK;
  }
  if (n == "no_gemv1" || n == "gemv1_rpw" || n == "gemv1_rpw_qkv" || n == "gemv1_rpw_ffn1" || n == "attn_oproj" || n == "attn_nk" || n == "steps_per_graph" || n == "no_gemm_skinny" || n == "gs_target_wgs" || n == "gs_xf" || n == "gs_wpack" || n == "w8_temporal" || n == "gs_fuse_ln" || n == "gs_rot" || n == "gs_dbg") {  // change the captured graphs: drop them
    if (n == "no_gemv1") e->opt_no_gemv1 = value != 0;
    else if (n == "gs_fuse_ln") e->opt_gs_fuse_ln = value != 0;
    else if (n == "gs_rot") e->opt_gs_rot = (int)value;
    else if (n == "gs_dbg") e->opt_gs_dbg = (int)value;
    else if (n == "gs_xf") e->opt_gs_xf = value != 0;
    else if (n == "gs_wpack") e->opt_gs_wpack = value != 0;
    else if (n == "w8_temporal") e->opt_w8_temporal = (int)value;
    else if (n == "gs_target_wgs") e->opt_gs_target = (int)value;  // 1 = no split-K
    else if (n == "no_gemm_skinny") e->opt_no_gemm_skinny = value != 0;
    else if (n == "attn_nk") e->opt_nk = (int)value;
    else if (n == "steps_per_graph") e->opt_spg = (int)value;
    else if (n == "gemv1_rpw_qkv") e->opt_rpw_qkv = (int)value;
    else if (n == "gemv1_rpw_ffn1") e->opt_rpw_ffn1 = (int)value;
    else if (n == "attn_oproj") e->opt_attn_oproj = value != 0;
    else e->opt_rpw = (int)value;
    (void)hipStreamSynchronize(e->st);
    for (auto& kv : e->graphs) {
      if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
      if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
    }
    e->graphs.clear();
    return VLE_OK;
  }
  if (n == "attn_qw") {
    g_attn_qw = (int)value;
    return VLE_OK;
  }
  if (n == "attn_lsum") {
    g_attn_lsum = value != 0;
    return VLE_OK;
  }
  if (n == "attn_v2" || n == "attn_xcd" || n == "attn_q128" || n == "attn_mode" || n == "attn_defer" || n == "attn_ring") {
    (n == "attn_v2" ? g_attn_v2 : n == "attn_xcd" ? g_attn_xcd : n == "attn_q128" ? g_attn_q128 : n == "attn_mode" ? g_attn_mode : n == "attn_defer" ? g_attn_defer : g_attn_ring) = (int)value;
    return VLE_OK;
  }
  if (n == "gs_formal" || n == "g1_shared" || n == "qa_waves" || n == "gs_msplit" || n == "gs_fast" || n == "gs_gran" || n == "gs_nf" || n == "attn_lds_pad" || n == "attn_nt" || n == "gs_ms_pad") {  // process-global kernel selection / argument: drop the captured graphs
    if (n == "qa_waves") {
      if (!(value == 4 || value == 8)) return e->fail(VLE_EINVAL, "qa_waves must be 4 or 8");
      g_qa_waves = (int)value;
    } else {
      if (n == "gs_ms_pad") g_gs_ms_pad = (int)std::max<int64_t>(0, std::min<int64_t>(value, 120 * 1024));
      else if (n == "attn_nt") g_da_nt = (int)value;
      else if (n == "attn_lds_pad") g_da_lds_pad = (int)std::max<int64_t>(0, std::min<int64_t>(value, 60 * 1024));
      else if (n == "gs_msplit") g_gs_msplit = (int)value;
      else if (n == "gs_fast") g_gs_fast = value != 0;
      else if (n == "gs_gran") g_gs_gran = value != 0;
      else if (n == "gs_nf") g_gs_nf = value != 0;
      else (n == "gs_formal" ? g_gs_formal : g_g1_shared) = value != 0;
    }
    drop_all_graphs(e);  // every live engine of the process, not only the caller
    return VLE_OK;
  }
  if (n == "g8_nt") {
    g_g8_nt = (int)value & 3;
    return VLE_OK;
  }
  if (n == "g8_persist") {  // process-global: gemm_8ph.hip's persistent tile loop (A/B)
    g_g8_persist = (int)value & 7;
    return VLE_OK;
  }
  if (n == "attn_f32_vec") {
    g_attn_f32_vec = value != 0;
    return VLE_OK;
  }
  if (n == "f32_glds") {  // process-global: fp32 packed-row GEMMs on gemm_glds.hip's ring (1) or gemm.hip (0)
    g_f32_glds = value != 0;
    return VLE_OK;
  }
  if (n == "glds_swz" || n == "glds_8ph" || n == "g8_stagger" || n == "g8_colgroup" || n == "glds_tail" || n == "glds_t64") {
    (n == "glds_swz" ? g_glds_swz : n == "glds_8ph" ? g_glds_8ph : n == "g8_stagger" ? g_g8_stagger : n == "glds_tail" ? g_glds_tail : n == "glds_t64" ? g_glds_t64 : g_g8_colgroup) = (int)value;
    return VLE_OK;
  }
  if (n == "glds_big" || n == "glds_w8" || n == "glds_prio") {  // process-global tile policy of gemm_glds.hip (same knobs as vle_op_tune)
    (n == "glds_big" ? g_glds_big : n == "glds_w8" ? g_glds_w8 : g_glds_prio) = (int)value;
    return VLE_OK;
  }
  if (n == "persist" || n == "persist_pf" || n == "persist_nk" || n == "persist_trace" || n == "persist_mode" || n == "persist_naps" || n == "act_bf16" ||
      n == "persist_sample" || n == "persist_steps" || n == "persist_batch") {  // change the captured graphs: drop them
    if (n == "persist") e->opt_persist = value != 0;
    else if (n == "persist_batch") e->opt_persist_batch = value != 0;  // 2 .. PSB_MAX utterances on the persistent launch (0: the launch chain)
    else if (n == "persist_sample") e->opt_ps_sample = value != 0;
    else if (n == "persist_steps") {
      if (value < 1 || value > 4096) return e->fail(VLE_EINVAL, "persist_steps must be 1 .. 4096");
      e->opt_ps_steps = (int)value;
    }
    else if (n == "act_bf16") e->opt_act_bf16 = (int)value & 3;
    else if (n == "persist_mode") e->opt_ps_mode = (int)value;
    else if (n == "persist_naps") e->opt_ps_naps = (int)value;  // (-1: back to the engine mode's default)
    else if (n == "persist_pf") {
      if (value < 0 || value > 3) return e->fail(VLE_EINVAL, "persist_pf must be 0 .. 3");
      e->opt_ps_pf = (int)value;
    } else if (n == "persist_nk") {
      if (!(value == 2 || value == 4)) return e->fail(VLE_EINVAL, "persist_nk must be 2 or 4");
      e->opt_ps_nk = (int)value;
    } else {
      e->opt_ps_trace = value != 0;
      if (e->opt_ps_trace && !e->ps_ptrace && e->finalized) {
        unsigned long long* p = nullptr;
        int r = dev_alloc(e, &p, (size_t)8 * 256 * PS_PT_SLOTS);
        if (r) return r;
        e->ps_ptrace = p;
      }
      if (e->ps_ptrace) (void)hipMemset(e->ps_ptrace, 0, (size_t)8 * 256 * PS_PT_SLOTS * sizeof(unsigned long long));
    }
    (void)hipStreamSynchronize(e->st);
    for (auto& kv : e->graphs) {
      if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
      if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
    }
    e->graphs.clear();
    return VLE_OK;
  }
  if (n == "ktrace") {  // changes the kernels' arguments: drop the captured graphs
    e->opt_ktrace = value != 0;
    if (e->opt_ktrace && !e->ktrace_buf && e->finalized) {
      unsigned long long* p = nullptr;
      int r = dev_alloc(e, &p, (size_t)KT_STEPS * KT_KERNELS * KT_WAVES * 4);
      if (r) return r;
      e->ktrace_buf = p;
    }
    if (e->ktrace_buf) (void)hipMemset(e->ktrace_buf, 0xFF, (size_t)KT_STEPS * KT_KERNELS * KT_WAVES * 4 * sizeof(unsigned long long));
    (void)hipStreamSynchronize(e->st);
    for (auto& kv : e->graphs) {
      if (kv.second.first) (void)hipGraphExecDestroy(kv.second.first);
      if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
    }
    e->graphs.clear();
    return VLE_OK;
  }
  if (n == "persist_inject_fail") {  // test hook: the next `value` persistent calls end as if a wave had given up (VLE_EBUSY)
    e->dbg_inject_psfail = (int)std::max<int64_t>(0, value);
    return VLE_OK;
  }
  if (n == "persist_rearm") {  // forget the back-off: the next batch-1 prefill may take the persistent launch again
    e->ps_backoff = 0; e->ps_backoff_next = 2;
    return VLE_OK;
  }
  if (n == "ln_fold") {  // 1 (default): the LayerNorms of the prefill / NAR passes ride on the residual GEMMs' epilogues; 0: LayerNorm launches
    e->opt_ln_fold = value != 0;
    return VLE_OK;
  }
  if (n == "fp8_gemm") {
    e->opt_fp8_gemm = value != 0;
    return VLE_OK;
  }
  if (n == "ignore_eos") {
    e->opt_ignore_eos = value != 0;
    return VLE_OK;
  }
  if (n == "nsplit") {
    if (value < 0 || value > 16 || (value & (value - 1))) return e->fail(VLE_EINVAL, "nsplit must be 0 (auto), 1, 2, 4, 8 or 16");
    e->opt_nsplit = (int)value;
    if (value > 0) e->nsplit = (int)value;
    return VLE_OK;
  }
  return e->fail(VLE_EINVAL, "unknown option: " + n);
}

extern "C" int64_t vle_debug_fetch(vle_engine* e, const char* what, void* host_dst, size_t bytes) {
  if (!e || !what || !host_dst) return VLE_EINVAL;
  (void)hipSetDevice(e->cfg.device);
  (void)hipStreamSynchronize(e->st);
  const std::string w = what;
  const void* src = nullptr;
  size_t n = 0;
  if (w == "ar_logits") {
    if (!e->trace_ar) return e->fail(VLE_ESTATE, "trace_ar_logits was not enabled");
    src = e->trace_ar;
    n = (size_t)((int64_t)e->n_steps + 1) * e->B * V_AR * sizeof(float);
  } else if (w.rfind("nar_logits:", 0) == 0) {
    const int stage = atoi(w.c_str() + 11);
    if (!e->trace_nar || stage < 0 || stage >= e->Q - 1) return e->fail(VLE_ESTATE, "trace_nar_logits not enabled / bad stage");
    src = e->trace_nar + (size_t)stage * e->max_B * e->max_G * NUM_AUDIO_TOKENS;
    n = (size_t)e->sumG_last * NUM_AUDIO_TOKENS * sizeof(float);
  } else if (w == "ar_sampled") {
    src = e->sampled;
    n = (size_t)e->B * e->max_G * sizeof(int64_t);
  } else if (w == "kv_len") {
    src = e->S.kv_len;
    n = (size_t)e->B * sizeof(int32_t);
  } else if (w == "kernel_times") {  // 8 x total ms, then 8 x calls (doubles)
    double buf[16];
    for (int i = 0; i < 8; ++i) {
      buf[i] = e->prof_ms[i];
      buf[8 + i] = e->prof_calls[i];
    }
    const size_t nb = std::min(bytes, sizeof(buf));
    memcpy(host_dst, buf, nb);
    return (int64_t)nb;
  } else if (w == "ktrace") {
    if (!e->ktrace_buf) return e->fail(VLE_ESTATE, "ktrace was not enabled");
    src = e->ktrace_buf;
    n = (size_t)KT_STEPS * KT_KERNELS * KT_WAVES * 4 * sizeof(unsigned long long);
  } else if (w == "qa_spin_fail") {  // workgroups of the fused launch that gave up waiting for q and recomputed it (expected: 0)
    if (!e->qa_spin_fail) return e->fail(VLE_ESTATE, "no hand-off counter");
    src = e->qa_spin_fail;
    n = sizeof(unsigned);
  } else if (w == "gs_gran_fail") {  // split-K finishers that gave up waiting for a partial tile (expected: 0)
    if (!e->qa_spin_fail) return e->fail(VLE_ESTATE, "no hand-off counter");
    src = e->qa_spin_fail + 1;
    n = sizeof(unsigned);
  } else if (w == "persist_fail") {  // waves of the persistent step that gave up waiting (expected: 0)
    if (!e->qa_spin_fail) return e->fail(VLE_ESTATE, "no hand-off counter");
    src = e->qa_spin_fail + 2;
    n = sizeof(unsigned);
  } else if (w == "ar_launches" || w == "persist_sample_active") {
    // ar_launches: stream submissions of the last AR loop after the first (eager) step -- with the sampling step inside the persistent
    // launch each is ONE pstep_kernel launch of several iterations; persist_sample_active: 1 when that is how the next call would run
    const int32_t v = w == "ar_launches" ? e->n_ar_launches : (persist_ready(e) && e->opt_ps_sample ? 1 : 0);
    const size_t nb = std::min(bytes, sizeof(v));
    memcpy(host_dst, &v, nb);
    return (int64_t)nb;
  } else if (w == "persist_ran" || w == "persist_fallbacks" || w == "persist_backoff") {
    // persist_ran: 1 when the last vle_ar_generate ran the persistent launch; persist_fallbacks: calls that ended with VLE_EBUSY since
    // vle_create; persist_backoff: batch-1 prefills left on the launch chain before the persistent launch is re-armed
    const int32_t v = w == "persist_ran" ? (e->ps_last_call ? 1 : 0) : w == "persist_fallbacks" ? (int32_t)e->ps_fallbacks : e->ps_backoff;
    const size_t nb = std::min(bytes, sizeof(v));
    memcpy(host_dst, &v, nb);
    return (int64_t)nb;
  } else if (w == "persist_capable") {
    // 1 when a ONE-utterance call on this engine would run the persistent launch (shape, device, options, tables, not backed off) whatever
    // the batch of the last prefill was: what VALLE.inference_batch asks before it decodes two utterances one after the other
    const bool v1 = e->opt_persist && e->ps_backoff == 0 && e->ps_device_ok && !e->slot_mode && !e->opt_profile && e->ps_table != nullptr &&
                    e->ps_gran != nullptr && (!e->w8 || ps_w8_mode_ok(e)) && ps_form_ok(e) && (int)e->ar.size() == e->L;
    const int32_t v = v1 ? 1 : 0;
    const size_t nb = std::min(bytes, sizeof(v));
    memcpy(host_dst, &v, nb);
    return (int64_t)nb;
  } else if (w == "persist_batch_capable") {
    // the largest batch (0, or 2 .. PSB_MAX) a call on this engine would run on the batched persistent launch (persist_nb.hip) whatever
    // the batch of the last prefill was: VALLE.inference_batch decodes two utterances one after the other only where this says 0
    int32_t v = 0;
    if (e->opt_persist && e->ps_backoff == 0 && e->ps_device_ok && !e->slot_mode && !e->opt_profile && e->ps_table != nullptr && e->ps_gran != nullptr &&
        ps_form_ok(e) && (int)e->ar.size() == e->L)
      for (int b = 2; b <= PSB_MAX; ++b)
        if (psb_covers(e, b)) v = b;
    const size_t nb = std::min(bytes, sizeof(v));
    memcpy(host_dst, &v, nb);
    return (int64_t)nb;
  } else if (w == "persist_active") {  // 1 when the next batch-1 step would run the persistent launch
    const int32_t v = persist_ready(e) ? 1 : 0;
    const size_t nb = std::min(bytes, sizeof(v));
    memcpy(host_dst, &v, nb);
    return (int64_t)nb;
  } else if (w == "persist_trace") {
    if (!e->ps_ptrace) return e->fail(VLE_ESTATE, "persist_trace was not enabled");
    src = e->ps_ptrace;
    n = (size_t)8 * 256 * PS_PT_SLOTS * sizeof(unsigned long long);
  } else if (w == "last_logits") {
    src = e->logits;
    n = (size_t)e->B * V_AR * sizeof(float);
  } else {
    return e->fail(VLE_EINVAL, "unknown debug item: " + w);
  }
  n = std::min(n, bytes);
  if (hipMemcpy(host_dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) return e->fail(VLE_EHIP, "debug copy failed");
  return (int64_t)n;
}

// Debugging aid for CALLER-owned buffers (the engine's own go through VLE_GUARD_ALLOC): a device allocation in its own virtual-memory
// mapping with an unmapped granule on either side.  at_start = 0: the buffer ENDS at the end of its mapping (a kernel that reads or
// writes past the end of x / y / forced tokens / codes faults at once); 1: it STARTS at the mapping's start (under-runs).  Never freed.
extern "C" int vle_debug_guard_alloc(int32_t device, size_t bytes, int32_t at_start, void** out) {
  if (!out || bytes == 0) return VLE_EINVAL;
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return VLE_EHIP;
  }
  if (guard_alloc(device, out, bytes, "caller", at_start ? 2 : 1) != 0) {
    (void)hipGetLastError();
    set_global_error("vle_debug_guard_alloc: virtual-memory allocation failed");
    return VLE_EHIP;
  }
  return VLE_OK;
}

extern "C" int vle_last_timings(vle_engine* e, double* out4) {
  if (!e || !out4) return VLE_EINVAL;
  out4[0] = e->t_prefill; out4[1] = e->t_ar; out4[2] = e->t_nar; out4[3] = e->n_steps;
  return VLE_OK;
}

// bytes of one AR step = W_AR * w + sum_b (2 L c_b d a  +  2 L d a)
extern "C" int vle_quantize_fp8w(const float* w, int64_t N, int64_t K, uint8_t* q_out, float* scale_out, float* deq_out) {
  if (!w || !scale_out || N < 1 || K < 1) return VLE_EINVAL;
  quantize_rows_fp8w(w, N, K, q_out, scale_out, deq_out);
  return VLE_OK;
}

extern "C" int64_t vle_ar_step_bytes(const vle_engine* e, int32_t B, int64_t sum_ctx) {
  if (!e) return VLE_EINVAL;
  const int64_t d = e->d, L = e->L, es = (int64_t)dtype_size(e->dtype);
  if (e->w8) {  // FP8W: matrices 1 byte / element + one fp32 scale per row; vectors fp32-equivalent counted as in bf16 mode
    const int64_t mats = L * 12 * d * d + (int64_t)V_AR * d, rows = L * 9 * d + V_AR, vecs = L * 13 * d + 2 * d;
    return mats + 4 * rows + vecs * es + 2 * L * d * es * (sum_ctx + B);
  }
  const int64_t w_ar = L * (12 * d * d + 13 * d) + 2 * d + (int64_t)V_AR * d;
  return w_ar * es + 2 * L * d * es * (sum_ctx + B);
}
