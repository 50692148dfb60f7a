// Device helpers of the persistent AR launches (persist.hip: one utterance; persist_nb.hip: 2-6 utterances per launch): granule
// loads / stores, the bounded sweeps, the bf16-row dot product and the wave totals of the D2 forms, the in-kernel timeline.
// (Moved out of persist.hip unchanged in round 6 so that both kernels run the SAME functions.)
#pragma once
#include "common.h"
#include "kernels.h"
#include "gemv1_dev.h"
#include "sampling_dev.h"

namespace vle {

namespace {

constexpr int PS_T = 256;        // 4 waves, one per SIMD
constexpr unsigned PS_SPINS = 1u << 18;  // polling passes before a wave gives up (>= 0.1 s)

typedef unsigned long long gran_t;

// Every pointer of the step comes out of the layer table in device memory: hipcc would treat it as a FLAT address (flat_load:
// both counters, no scalar path).  The table is read through the constant address space (uniform index -> s_load) and the
// operands through global-address-space pointers (global_load, vmcnt only).
#define PS_GLOBAL __attribute__((address_space(1)))
#define PS_CONST __attribute__((address_space(4)))
template <typename X>
__device__ inline const X PS_GLOBAL* as_g(unsigned long long v) { return (const X PS_GLOBAL*)v; }
template <typename X>
__device__ inline X PS_GLOBAL* as_gw(unsigned long long v) { return (X PS_GLOBAL*)v; }
__device__ inline u32x4_t ps_load_nt(const u32x4_t PS_GLOBAL* p) { return __builtin_nontemporal_load(p); }  // weights: read once per step
__device__ inline void ps_load4(const float PS_GLOBAL* p, float (&f)[4]) {
  const f32x4v_t t = *reinterpret_cast<const f32x4v_t PS_GLOBAL*>(p);
  f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
}
struct PsLayer {  // one entry of the table, as addresses
  unsigned long long wqkv, wo, w1, w2, bqkv, bo, b1, b2, g1, be1, g2, be2, kc, vc, sgqkv, tbqkv, sg1, tb1, sqkv, so, s1, s2;
};
__device__ inline PsLayer ps_layer(const PLayer* tab, int l) {
  static_assert(sizeof(PLayer) == 22 * 8, "PLayer is 22 pointers");
  const unsigned long long PS_CONST* t = (const unsigned long long PS_CONST*)tab + (size_t)l * 22;
  PsLayer p;
  p.wqkv = t[0]; p.wo = t[1]; p.w1 = t[2]; p.w2 = t[3]; p.bqkv = t[4]; p.bo = t[5]; p.b1 = t[6]; p.b2 = t[7];
  p.g1 = t[8]; p.be1 = t[9]; p.g2 = t[10]; p.be2 = t[11]; p.kc = t[12]; p.vc = t[13];
  p.sgqkv = t[14]; p.tbqkv = t[15]; p.sg1 = t[16]; p.tb1 = t[17];
  p.sqkv = t[18]; p.so = t[19]; p.s1 = t[20]; p.s2 = t[21];  // (the bf16 instantiations never use them: the loads are dropped)
  return p;
}

// the in-launch sampling step's operand block (device memory) through scalar loads
__device__ inline PStepSample ps_sample_load(const PStepSample* p) {
  static_assert(sizeof(PStepSample) == 18 * 8, "PStepSample is 18 eight-byte words");
  const unsigned long long PS_CONST* t = (const unsigned long long PS_CONST*)p;
  unsigned long long wds[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) wds[i] = t[i];
  PStepSample q;
  __builtin_memcpy(&q, wds, sizeof(q));
  return q;
}

__device__ inline gran_t gran_load(const gran_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gran_store(gran_t* p, unsigned epoch, float v) {
  __hip_atomic_store(p, ((gran_t)epoch << 32) | (gran_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void gran_store_bits(gran_t* p, unsigned epoch, unsigned bits) {
  __hip_atomic_store(p, ((gran_t)epoch << 32) | (gran_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same granule with the DEFAULT cache policy: the line stays (dirty) in the storing CU's XCD L2, where the L1-bypassing polls of
// the other CUs of that XCD hit it one L2 round trip later -- a fast path for the edges whose producers and consumers are placed on
// one XCD.  Never the only copy: another XCD cannot see it, so the write-through granule above is always stored too.
__device__ inline void gran_store_local(gran_t* p, unsigned epoch, unsigned bits) {
  *reinterpret_cast<gran_t PS_GLOBAL*>((unsigned long long)p) = ((gran_t)epoch << 32) | (gran_t)bits;
}
// two fp32 -> one bf16 pair, round-to-nearest-even: v_cvt_pk_bf16_f32 (for finite values the bits of common.h f32_to_bf16, which the
// launch chain applies to the same rows -- 10 integer instructions per pair there)
__device__ inline unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
  const pk_bf16x2 v = pk_bf16x2{(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// D2 (PK bit 3): the operator's input row sits in LDS as bf16 and the dot products run on v_dot2c_f32_bf16 -- two multiply-adds per
// lane and instruction on the 16-byte weight vectors as they arrive, no widening of either operand (per 8 weights: 4 instructions
// instead of 8 shifts / masks + 4 packed FMAs, and half the LDS reads).  The hidden row and (mode bit 3) the attention row already
// travel as bf16 pairs: for linear2 / out-proj the products are the same numbers in another summation order; the in-projection,
// linear1 and the predict layer round x * gamma (the folded LayerNorm's operand) to bf16 first -- what the batched step's MFMA
// GEMMs do with the same rows (gemm_skinny.hip, LnProducer).
typedef __bf16 ps_bf16x2 __attribute__((ext_vector_type(2)));
template <int NCH>
__device__ inline float ps_dot_bf16(const u32x4_t (&wv)[NCH], const u32x4_t (&xv)[NCH]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four independent chains (a dot2c accumulates in place)
  auto bf2 = [](unsigned u) { return __builtin_bit_cast(ps_bf16x2, u); };  // (by VALUE: __builtin_bit_cast of a vector-element lvalue reads element 0)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const unsigned w0 = wv[c].x, w1 = wv[c].y, w2 = wv[c].z, w3 = wv[c].w, x0 = xv[c].x, x1 = xv[c].y, x2 = xv[c].z, x3 = xv[c].w;
    a0 = __builtin_amdgcn_fdot2_f32_bf16(bf2(w0), bf2(x0), a0, false);
    a1 = __builtin_amdgcn_fdot2_f32_bf16(bf2(w1), bf2(x1), a1, false);
    a2 = __builtin_amdgcn_fdot2_f32_bf16(bf2(w2), bf2(x2), a2, false);
    a3 = __builtin_amdgcn_fdot2_f32_bf16(bf2(w3), bf2(x3), a3, false);
  }
  return (a0 + a1) + (a2 + a3);
}
// Wave totals for the D2 forms (no bit-identity with the launch chain to keep: its wave_sum_dpp reads the four row totals back
// through v_readlane -- 15 instructions per total).  One value: DPP row sums + the two permlane swaps (8).  R <= 4 values (the rows of
// one operator): after the row sums, lane column c keeps value c & 3 and ONE pair of swaps finishes all of them -- lane r < R ends
// up with total r, which is where the callers want it (R row sums + R - 1 selects + 4 instead of 15 R).
__device__ inline float ps_wave_sum_fast(float v) { return rows4_sum(row16_sum_dpp(v)); }
template <int R>
__device__ inline float ps_wave_sums_fast(const float (&t)[R]) {
  static_assert(R >= 1 && R <= 4, "one value per lane column modulo 4");
  const int sel = threadIdx.x & 3;
  float v = row16_sum_dpp(t[0]);
#pragma unroll
  for (int r = 1; r < R; ++r) {
    const float u = row16_sum_dpp(t[r]);
    v = sel == r ? u : v;
  }
  return rows4_sum(v);  // lanes with (lane & 3) == r < R: total r (other columns of R < 4: total 0's copy)
}
// this lane's 8 bf16 activations of chunk c (elements c * 512 + lane * 8 ..) from the bf16 row at sxh
template <int NCH>
__device__ inline void ps_read_bf16(const float* sx, u32x4_t (&xv)[NCH]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NCH; ++c) xv[c] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned char*>(sx) + (c * 512 + lane * 8) * 2);
}

// Spin state of a wave: `budget` polling passes left in this launch (0 = gave up: never waits again).
struct PsSpin {
  unsigned budget;
  unsigned* fail;
  int sleep;        // s_sleep units between two polling passes
  unsigned passes;  // passes of the last gather (timeline diagnostic)
};
__device__ inline bool ps_retry(PsSpin& sp) {  // wave-uniform; false = stop waiting
  if (sp.budget == 0) return false;
  if (--sp.budget == 0) {
    if ((threadIdx.x & 63) == 0 && sp.fail) atomicAdd(sp.fail, 1u);
    return false;
  }
  for (int i = 0; i < sp.sleep; ++i) __builtin_amdgcn_s_sleep(1);
  return true;
}

// NV consecutive granules at g -> v[NV], re-read until all 64 lanes of the wave see this step's epoch on every tag
template <int NV>
__device__ inline void gather_vals(const gran_t* g, unsigned epoch, float (&v)[NV], PsSpin& sp) {
  sp.passes = 0;
  for (;;) {
    gran_t raw[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) raw[k] = gran_load(g + k);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      ok &= (unsigned)(raw[k] >> 32) == epoch;
      v[k] = __uint_as_float((unsigned)raw[k]);
    }
    ++sp.passes;
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
}
// the same with 16-byte loads (two granules each; the 8-byte halves are what the producers store: observed untorn on gfx950,
// MI355X_MICROARCH.md "Valid forms"): half the load instructions of a sweep.  `off` = byte offset of g in the granule buffer.
// `after_first_issue` runs between the first sweep's loads and their first use: whatever it requests (the operands of a LATER
// operator) is younger than the sweep, so the sweep does not wait for it -- a wave's loads return in order, and requests issued
// BEFORE a sweep delay it by their whole HBM round trip.
struct PsNoop {
  __device__ void operator()() const {}
};
template <int NV, typename F>
__device__ inline void gather_vals16(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned epoch, float (&v)[NV], PsSpin& sp, F&& after_first_issue) {
  static_assert(NV % 2 == 0, "pairs of granules");
  sp.passes = 1;
  {
    u32x4_t raw[NV / 2];
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16 * k), 0, 16 /* sc1 */);
    after_first_issue();
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
      ok &= raw[k].y == epoch && raw[k].w == epoch;
      v[2 * k] = __uint_as_float(raw[k].x);
      v[2 * k + 1] = __uint_as_float(raw[k].z);
    }
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
  for (;;) {
    u32x4_t raw[NV / 2];
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16 * k), 0, 16 /* sc1 */);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
      ok &= raw[k].y == epoch && raw[k].w == epoch;
      v[2 * k] = __uint_as_float(raw[k].x);
      v[2 * k + 1] = __uint_as_float(raw[k].z);
    }
    ++sp.passes;
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
}

// NV granules per lane (16-byte loads) + ONE more granule at byte offset off1 (the same for every lane), all in the same pass
template <int NV>
__device__ inline void gather_vals16_plus1(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned off1, unsigned epoch, float (&v)[NV], float& v1, PsSpin& sp) {
  static_assert(NV % 2 == 0, "pairs of granules");
  sp.passes = 0;
  for (;;) {
    u32x4_t raw[NV / 2];
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16 * k), 0, 16 /* sc1 */);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t r1 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off1, 0, 16 /* sc1 */);
    bool ok = r1.y == epoch;
    v1 = __uint_as_float(r1.x);
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
      ok &= raw[k].y == epoch && raw[k].w == epoch;
      v[2 * k] = __uint_as_float(raw[k].x);
      v[2 * k + 1] = __uint_as_float(raw[k].z);
    }
    ++sp.passes;
    if (__all(ok)) return;
    if (!ps_retry(sp)) return;
  }
}

template <int NV, typename F>
__device__ inline void ps_gather(const gran_t* gbase, __amdgpu_buffer_rsrc_t rs, const gran_t* g, unsigned epoch, float (&v)[NV], PsSpin& sp, F&& after_first_issue) {
  static_assert(NV % 2 == 0, "16-byte sweeps");
  gather_vals16<NV>(rs, (unsigned)((const char*)g - (const char*)gbase), epoch, v, sp, after_first_issue);
}
// One granule per lane from the XCD-local copy (gl; null = none) or the write-through copy (g): PS_LOCAL_TRIES passes on the local
// copy, one on the other, and so on -- whatever the placement, the write-through copy is found.
constexpr int PS_LOCAL_TRIES = 6;
template <typename F>
__device__ inline float gather_one_dual(const gran_t* g, const gran_t* gl, unsigned epoch, PsSpin& sp, F&& after_first_issue) {
  sp.passes = 1;
  {
    const gran_t raw = gran_load(gl != nullptr ? gl : g);
    after_first_issue();
    if (__all((unsigned)(raw >> 32) == epoch)) return __uint_as_float((unsigned)raw);
    if (!ps_retry(sp)) return __uint_as_float((unsigned)raw);
  }
  for (;;) {
    const bool local = gl != nullptr && (sp.passes % (PS_LOCAL_TRIES + 1)) != PS_LOCAL_TRIES;
    const gran_t raw = gran_load(local ? gl : g);
    ++sp.passes;
    if (__all((unsigned)(raw >> 32) == epoch)) return __uint_as_float((unsigned)raw);
    if (!ps_retry(sp)) return __uint_as_float((unsigned)raw);
  }
}
// (Measured and dropped: a producer wave storing only the XCD-local copy before its own sweep and the write-through copy once it has
// its data -- the q/k/v edge went 0.74 -> 0.63 us but the deferred acknowledgement then sat in front of the next stage: 144.5 -> 145.5 us.)
// two granules per lane (16 bytes), same alternation; offsets in bytes into the granule buffer
__device__ inline void gather_two_dual(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned off_local, bool have_local, unsigned epoch, float (&v)[2],
                                       PsSpin& sp) {
  sp.passes = 0;
  for (;;) {
    const bool local = have_local && (sp.passes % (PS_LOCAL_TRIES + 1)) != PS_LOCAL_TRIES;
    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(local ? off_local : off), 0, 16 /* sc1 */);
    v[0] = __uint_as_float(raw.x);
    v[1] = __uint_as_float(raw.z);
    ++sp.passes;
    if (__all(raw.y == epoch && raw.w == epoch)) return;
    if (!ps_retry(sp)) return;
  }
}

// timeline (option "persist_trace"): per hand-off {wall clock when the wave began to wait, polling passes, wall clock when it had the data}
struct PsTrace {
  unsigned long long* p;  // this workgroup's slots, thread 0 only; null otherwise
  int i;
  unsigned long long t0;
};
// TR (compile time): the untraced instantiations carry none of this -- a run-time `if (t.p)` at every hand-off cost 28 exec-masked
// branches per layer and kept the trace state live in scalar registers through the whole step (layer loop 2873 -> 2500 instructions,
// 323 -> 130 v_readlane of spilled scalars)
template <bool TR>
__device__ inline void pt_begin(PsTrace& t) {
  if constexpr (TR) {
    if (t.p) t.t0 = wall_clock64();
  }
}
template <bool TR>
__device__ inline void pt_end(PsTrace& t, unsigned passes) {
  if constexpr (!TR) return;
  if (t.p && t.i + 3 <= PS_PT_SLOTS) {
    t.p[t.i] = t.t0; t.p[t.i + 1] = passes; t.p[t.i + 2] = wall_clock64();
    t.i += 3;
  }
}

}  // namespace

// granules of one layer (per utterance)
__host__ __device__ inline int ps_gran_per_layer(int d, int H, int NS) { return d + 3 * d + H * NS * (2 + d / H) + d + d + 4 * d + 3 * d + H * NS * (2 + d / H); }

}  // namespace vle
