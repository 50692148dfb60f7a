// bf16 GEMM of the prefill and the 7 NAR passes on the LDS-DMA path:
//   out = epi(A[M x K] @ W[N x K]^T + bias)      (same contract and epilogues as gemm.hip)
//   reference ops: in-proj / out-proj `linear` (valle/modules/activation.py:414-421), FFN
//   linear1/linear2 (valle/modules/transformer.py:332-334), nar_predict_layers (valle.py:1128).
//
// Why a second GEMM: at M ~ 1 k rows (one utterance: text + prompt + generated frames) a GEMM is
// one tile per CU, i.e. one 4-wave workgroup per CU with nothing else to hide latency.  gemm.hip
// keeps ONE k-step of global loads in flight through registers, so every 64-deep k-step pays a
// full L2/HBM round trip (measured 60-140 TF).  Here the tiles go global -> LDS directly
// (global_load_lds_dwordx4, no VGPR round trip) into a ring of STAGES buffers with STAGES-1
// k-steps in flight; the wait is a counted s_waitcnt vmcnt(N) + raw s_barrier (a __syncthreads()
// would drain the DMA queue), one barrier per k-step.
//   * LDS image of a stage: rows x 128 B (64 bf16 of K), 16-byte slot c of row r holds global
//     vector c ^ (r & 7): LDS-DMA writes lane-linearly (wave base + lane*16), so the swizzle is
//     applied on the SOURCE address and again on the ds_read_b128 side (same involution);
//   * v_mfma_f32_16x16x32_bf16, 4 waves as 2 x 2, wave tile (BM/2) x (BN/2);
//   * 16-row fragments that lie entirely beyond M skip their MFMAs (M = 1025 leaves a 1-row tail tile);
//   * XCD-aware tile order as in gemm.hip.
#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 gg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gg_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gg_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gg_bf16x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ inline void gg_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// SWZ selects the XOR key of the 16-byte slot swizzle inside a 128-byte LDS row: 0 = row & 7, 1 = (row >> 1) & 7.
// A 128-byte row spans half of the 64 banks and consecutive rows alternate halves, so the 16 rows of one ds_read_b128
// lane group fall 8 + 8 into the two halves: key (row >> 1) & 7 gives the 8 rows of a half 8 distinct slots, key row & 7
// gives rows r and r + 8 (same half) the same slot.
template <int SWZ>
__device__ inline int gg_key(int row) {
  return SWZ ? ((row >> 1) & 7) : (row & 7);
}

// one 16-byte fragment pair -> accumulator: bf16: 8 k per lane, one MFMA; fp32: 4 k per lane, four exact-fp32 MFMAs (x, y, z, w)
template <typename T>
__device__ inline gg_f32x4 gg_mma(const gg_bf16x8& w, const gg_bf16x8& a, gg_f32x4 c) {
  if constexpr (sizeof(T) == 2) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
  } else {
    const gg_f32x4 wf = __builtin_bit_cast(gg_f32x4, w), af = __builtin_bit_cast(gg_f32x4, a);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], af[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], af[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], af[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], af[3], c, 0, 0, 0);
    return c;
  }
}

// T: bf16_t, or float (round 6) -- the token-exact engine mode's packed-row GEMMs on the same pipeline: a stage row is still 128 bytes
// (32 fp32 of K), a fragment read is still one 16-byte vector per lane (4 consecutive k), fed to four v_mfma_f32_16x16x4_f32 (components
// x, y, z, w) exactly as gemm.hip's register-staged kernel does, in the same stage / half / component order: every output element is
// the same chain of exact fp32 FMAs, so the two kernels are BIT-IDENTICAL (tests/test_ops_gpu.py) and the token-exact mode's ids cannot
// move.  fp32 MFMA runs at the vector rate (157 TF/s), so what the ring buys is latency hiding: gemm.hip keeps one k-step in flight.
template <typename T, int BM, int BN, int EPI, int NW, bool PRIO = false, int SWZ = 0>
__global__ __launch_bounds__(NW * 64) void gemm_glds_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                        const float* __restrict__ bias, void* __restrict__ out_,
                                                        float* __restrict__ resid, int64_t M, int N, int K, int glds_legacy_epilogue, GemmLn ln) {
  // LayerNorm folded into the GEMMs (kernels.h GemmLn): LNP = this launch completes the residual stream and leaves bf16(x * gamma) +
  // group statistics for the next norm site; LNC = this launch reads x * gamma and applies rstd * (acc - mean * sg) + tb
  constexpr bool LNP = EPI == EPI_RESID_LNP, LNC = EPI == EPI_STORE_LNC || EPI == EPI_RELU_LNC;
  constexpr bool RESID = EPI == EPI_RESID || LNP, RELU = EPI == EPI_RELU || EPI == EPI_RELU_LNC;
  constexpr bool F32IN = sizeof(T) == 4;
  static_assert(!F32IN || !(LNP || LNC), "the LayerNorm-folded epilogues are bf16 only");
  constexpr int KE = 128 / (int)sizeof(T);  // K elements per stage row
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int STAGES = (4 * STAGE_BYTES <= 144 * 1024) ? 4 : 3;
  constexpr int D = STAGES - 1;                 // k-steps in flight
  constexpr int NIA = BM / (8 * NW), NIB = BN / (8 * NW);   // LDS-DMA instructions per wave per stage (8 rows each)
  constexpr int NI = NIA + NIB;
  constexpr int WMW = NW / 2;                   // waves along M (x 2 along N): 2 x 2, or 4 x 2 for the 8-wave 256-row tile
  constexpr int WM = BM / WMW, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  // LNC: behind the ring (never a K-tile target): (mean, M2) of the two halves of the tile's rows, then sg and tb of the tile's columns
  constexpr int LN_STATS = BM * 16, LN_LDS = LNC ? LN_STATS + 2 * BN * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE_BYTES + LN_LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches / LDS bases
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;  // wave >> 1 in [0, WMW)
  const int nbx = gridDim.x, nby = gridDim.y;
  const int nblk = nbx * nby;
  int bid = blockIdx.y * nbx + blockIdx.x;
  {
    // XCD-aware order over the full-height tiles (block b runs on XCD b % 8: give each XCD a contiguous
    // run of tiles sharing W panels in its L2); the cheap tail-row tiles keep the highest ids so they
    // are dispatched last, behind the first round of full tiles
    const int nfull = (int)(M / BM) * nbx;
    const int nr = bid < nfull ? nfull : nblk;
    if (bid < nfull || nfull == 0) {
      const int q = nr / 8, r = nr % 8, xcd = bid % 8, idx = bid / 8;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
  }
  const int64_t m0 = (int64_t)(bid / nbx) * BM;
  const int n0 = (bid % nbx) * BN;

  // per-lane source of the DMA pieces: piece p of a wave covers tile rows (p*4 + wave)*8 .. +8;
  // lane -> (row = lane>>3, slot = lane&7), source vector = slot ^ (row & 7)
  const int prow = lane >> 3;
  const unsigned char* srcA[NIA];
  const unsigned char* srcB[NIB];
#pragma unroll
  for (int p = 0; p < NIA; ++p) {
    int64_t gm = m0 + (p * NW + wave) * 8 + prow;
    gm = gm < M ? gm : M - 1;
    srcA[p] = reinterpret_cast<const unsigned char*>(A + gm * K) + ((lane & 7) ^ gg_key<SWZ>((p * NW + wave) * 8 + prow)) * 16;
  }
#pragma unroll
  for (int p = 0; p < NIB; ++p) {
    int gn = n0 + (p * NW + wave) * 8 + prow;
    gn = gn < N ? gn : N - 1;
    srcB[p] = reinterpret_cast<const unsigned char*>(W + (int64_t)gn * K) + ((lane & 7) ^ gg_key<SWZ>((p * NW + wave) * 8 + prow)) * 16;
  }
  auto issue = [&](int kt) {
    unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int p = 0; p < NIA; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[p] + (int64_t)kt * 128),
                                       (__attribute__((address_space(3))) void*)(st + (p * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < NIB; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[p] + (int64_t)kt * 128),
                                       (__attribute__((address_space(3))) void*)(st + BM * 128 + (p * NW + wave) * 1024), 16, 0, 0);
  };

  const int fr = lane & 15, fg = lane >> 4;
  // bias of this lane's 4 consecutive output columns per n-fragment: requested first (ahead of the
  // DMA queue, so its wait never drains the pipeline), clamped address, consumed in the epilogue
  gg_f32x4 bias4[LNC ? 1 : FN];
  if constexpr (LNC) {
    // sg / tb of this tile's BN columns into LDS, the first requests of the workgroup (BN / 4 lanes x 16 bytes each, waves 0 and 1)
    bias4[0] = gg_f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave < 2 && lane < BN / 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((wave == 0 ? ln.sg : bias) + n0 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(smem + STAGES * STAGE_BYTES + LN_STATS + wave * BN * 4), 16, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = min(n0 + wn0 + j * 16 + fg * 4, N - 4);
      bias4[j] = bias != nullptr ? *reinterpret_cast<const gg_f32x4*>(bias + n) : gg_f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  gg_f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = gg_f32x4{0.f, 0.f, 0.f, 0.f};
  // 16-row fragments of this wave with at least one row < M (wave-uniform; < FM only in the tail tile)
  int nlive = (int)((M - (m0 + wm0) + 15) / 16);
  nlive = nlive < 0 ? 0 : (nlive > FM ? FM : nlive);

  const int KT = K / KE;
  // LNC: the row statistics.  Thread t < 2 BM owns HALF of row m0 + t % BM (half t / BM of its K / 64 groups): the (mean, M2) pairs are
  // requested AHEAD of the K-tile queue (a wave's loads return in order: the wait for them never drains the pipeline), straight-line
  // and on clamped addresses (no branch around a request), one coalesced 8-byte load per group (group-major layout), and combined
  // after the main loop (the compiler sinks the arithmetic to its use):
  //   mean_h = avg(mean_g),  M2_h = sum(M2_g) + 64 * sum((mean_g - mean_h)^2)   (Chan; the squares taken around the half's first group)
  // The two halves meet in the epilogue (LDS behind the ring).
  constexpr int LNV = 12;  // groups per half row: K <= 1536
  typedef float gg_f32x2 __attribute__((ext_vector_type(2)));
  static_assert(!LNC || NW * 64 >= 2 * BM, "two threads per tile row");
  gg_f32x2 lnp[LNC ? LNV : 1];
  const int ln_ng = K / 128;  // groups per half row
  if constexpr (LNC) {
    int64_t m = m0 + tid % BM;
    m = m < M ? m : M - 1;
    const int half = (tid / BM) & 1;
    const gg_f32x2* sp = reinterpret_cast<const gg_f32x2*>(ln.stats_in) + (int64_t)half * ln_ng * ln.stats_ld + m;
#pragma unroll
    for (int g = 0; g < LNV; ++g) lnp[g] = sp[(int64_t)(g < ln_ng ? g : ln_ng - 1) * ln.stats_ld];
  }
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < KT) issue(s);
  float ln_mean = 0.f, ln_m2 = 0.f;  // of this thread's half row
  if constexpr (LNC) {
    const float ref = lnp[0][0];
    float s1 = 0.f, s2 = 0.f, q = 0.f;
#pragma unroll
    for (int g = 0; g < LNV; ++g) {
      const bool on = g < ln_ng;
      const float d0 = on ? lnp[g][0] - ref : 0.f;
      s1 += d0;
      s2 = fmaf(d0, d0, s2);
      q += on ? lnp[g][1] : 0.f;
    }
    const float sm = s1 * __builtin_amdgcn_rcpf((float)ln_ng);  // mean_h - ref (no IEEE division sequences in these epilogues: every
                                                                // wave of the workgroup runs them at once, issue-bound)
    ln_mean = ref + sm;
    ln_m2 = q + (float)LN_GROUP * (s2 - s1 * sm);
  }

  for (int kt = 0; kt < KT; ++kt) {
    // tile kt has landed once at most min(D-1, KT-1-kt) younger tiles of this wave are outstanding
    const int younger = KT - 1 - kt;
    if (younger >= D - 1) gg_wait_vm<(D - 1) * NI>();
    else if (D >= 3 && younger == 1) gg_wait_vm<NI>();
    else gg_wait_vm<0>();
    __builtin_amdgcn_s_barrier();  // every wave's pieces of tile kt landed; everyone finished reading tile kt-1
    if (kt + D < KT) issue(kt + D);  // into the buffer of tile kt-1

    const unsigned char* As = smem + (kt % STAGES) * STAGE_BYTES;
    const unsigned char* Bs = As + BM * 128;
    if (nlive == FM) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        gg_bf16x8 af[FM], bfr[FN];
        const int c = ks * 4 + fg;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm0 + i * 16 + fr;
          af[i] = *reinterpret_cast<const gg_bf16x8*>(As + row * 128 + ((c ^ gg_key<SWZ>(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn0 + j * 16 + fr;
          bfr[j] = *reinterpret_cast<const gg_bf16x8*>(Bs + row * 128 + ((c ^ gg_key<SWZ>(row)) << 4));
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);  // MFMA cluster ahead of the other waves' loads (knob "glds_prio")
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)  // W fragment as the A operand: C^T, see the epilogue
            acc[i][j] = gg_mma<T>(bfr[j], af[i], acc[i][j]);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      }
    } else if (nlive > 0) {  // tail tile (rows beyond M): only fragment rows that exist
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fg;
        gg_bf16x8 bfr[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn0 + j * 16 + fr;
          bfr[j] = *reinterpret_cast<const gg_bf16x8*>(Bs + row * 128 + ((c ^ gg_key<SWZ>(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          if (i < nlive) {
            const int row = wm0 + i * 16 + fr;
            const gg_bf16x8 a = *reinterpret_cast<const gg_bf16x8*>(As + row * 128 + ((c ^ gg_key<SWZ>(row)) << 4));
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = gg_mma<T>(bfr[j], a, acc[i][j]);
          }
        }
      }
    }
  }

  // epilogue.  The MFMAs above compute C^T (W fragment as the A operand), so lane (fg, fr) holds
  // C[m = 16 i + fr][n = 16 j + 4 fg + r], r = 0..3: four CONSECUTIVE columns of one row.
  //
  // Tiles that lie inside N go through LDS (free once the last k-step is consumed; measured on gemm_8ph.hip, which does the
  // same: a store instruction issued from the fragments covers 16 rows x 32 bytes and that store tail cost 20-50 % of the
  // kernel).  The tile is written as a row-major image -- 16-byte chunk c of row r at chunk c ^ (r & 7), and for 2-byte
  // elements the two 8-byte halves of a chunk swapped when (r >> 3) & 1: conflict-free for the fragment writes -- and read back
  // as whole rows: every global access is BN x element-size contiguous bytes of one output row, 16 bytes per lane; the
  // residual's old values are loaded in that row form too (requested before the barrier).
  if ((n0 + BN <= N && !glds_legacy_epilogue) || LNP || LNC) {  // (the LN epilogues exist in this form only: gemm_ln_supports)
    constexpr bool F32OUT = RESID || EPI == EPI_F32 || F32IN;  // (fp32 activations: every epilogue writes floats)
    constexpr int ES = F32OUT ? 4 : 2, RB = BN * ES, CPR = RB / 16, RPI = 64 / CPR, IT = BM / NW / RPI;
    static_assert(BM * RB <= STAGES * STAGE_BYTES && (BM / NW) % RPI == 0, "epilogue image must fit the LDS ring");
    static_assert(!LNP || CPR % 16 == 0, "a 16-lane row of the row-form pass covers one 64-column group");
    __syncthreads();  // every wave has consumed the last k-step (and its own requests have landed: the loop's last waits are vmcnt(0))
    unsigned char* const E = smem;
    if constexpr (LNC) {  // the rows' (mean, rstd) from their owner threads to the fragment layout
      float* const S = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);
      if (tid < 2 * BM) {
        S[4 * (tid % BM) + 2 * (tid / BM)] = ln_mean;
        S[4 * (tid % BM) + 2 * (tid / BM) + 1] = ln_m2;
      }
      __syncthreads();
    }
    const float ln_invk = __builtin_amdgcn_rcpf((float)K);
    gg_f32x4 sg4[LNC ? FN : 1], tb4[LNC ? FN : 1];  // this lane's columns of sg / tb, out of LDS once
    if constexpr (LNC) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        sg4[j] = *reinterpret_cast<const gg_f32x4*>(smem + STAGES * STAGE_BYTES + LN_STATS + (wn0 + j * 16 + fg * 4) * 4);
        tb4[j] = *reinterpret_cast<const gg_f32x4*>(smem + STAGES * STAGE_BYTES + LN_STATS + BN * 4 + (wn0 + j * 16 + fg * 4) * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float mean = 0.f, rstd = 1.f;
      if constexpr (LNC) {
        // the row's two halves (K / 2 elements each): mean = (m0 + m1) / 2, M2 = q0 + q1 + (K / 4) (m0 - m1)^2; once per row of the
        // lane, v_rcp / v_rsq (1 ulp) instead of IEEE division + square root sequences (30 instructions each: the first build of this
        // epilogue spent 600 instructions per wave on them, 1.2 us per tile with all eight waves in it)
        const gg_f32x4 hh = *reinterpret_cast<const gg_f32x4*>(smem + STAGES * STAGE_BYTES + (wm0 + i * 16 + fr) * 16);
        const float dm = hh[0] - hh[2];
        mean = 0.5f * (hh[0] + hh[2]);
        rstd = __builtin_amdgcn_rsqf(fmaf(hh[1] + hh[3] + 0.25f * (float)K * dm * dm, ln_invk, LN_EPS));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wm0 + i * 16 + fr, col = wn0 + j * 16 + fg * 4;
        gg_f32x4 v;
        if constexpr (LNC) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, fmaf(-mean, sg4[j][r], acc[i][j][r]), tb4[j][r]);
        } else {
          v = acc[i][j] + bias4[LNC ? 0 : j];
        }
        if constexpr (RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr (F32OUT) {
          *reinterpret_cast<gg_f32x4*>(E + row * RB + (((col >> 2) ^ (fr & 7)) << 4)) = v;
        } else {
          gg_bf16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];  // v_cvt_pk_bf16_f32: round-to-nearest-even
          *reinterpret_cast<gg_bf16x4*>(E + row * RB + (((col >> 3) ^ (fr & 7)) << 4) + ((((col >> 2) & 1) ^ (fr >> 3)) << 3)) = o4;
        }
      }
    }
    const int l = lane % CPR, rsub = lane / CPR;
    gg_f32x4 old[RESID ? IT : 1];
    gg_f32x4 gamma4 = gg_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RESID) {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int64_t m = m0 + wave * (BM / NW) + it * RPI + rsub;
        old[it] = *reinterpret_cast<const gg_f32x4*>(resid + (m < M ? m : M - 1) * N + n0 + l * 4);
      }
      if constexpr (LNP) gamma4 = *reinterpret_cast<const gg_f32x4*>(ln.gamma + n0 + l * 4);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = wave * (BM / NW) + it * RPI + rsub;
      const int64_t m = m0 + r;
      gg_u32x4 v = *reinterpret_cast<const gg_u32x4*>(E + r * RB + ((l ^ (r & 7)) << 4));
      if constexpr (!F32OUT) {
        if ((r >> 3) & 1) v = gg_u32x4{v[2], v[3], v[0], v[1]};
      }
      if constexpr (LNP) {
        // the completed residual row x: store it, and leave the next norm site's operand bf16(x * gamma) and this 64-column group's
        // (mean, M2) -- exact two-pass over the group's 64 fp32 values (16 lanes x 4: one DPP row), before any guard (whole rows of lanes)
        const gg_f32x4 x = old[it] + gg_f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        const float gmean = row16_sum_dpp((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / (float)LN_GROUP);
        const float d0 = x[0] - gmean, d1 = x[1] - gmean, d2 = x[2] - gmean, d3 = x[3] - gmean;
        const float gm2 = row16_sum_dpp(fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0))));
        if (m < M) {
          *reinterpret_cast<gg_f32x4*>(resid + m * N + n0 + l * 4) = x;
          gg_bf16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = (__bf16)(x[r] * gamma4[r]);
          *reinterpret_cast<gg_bf16x4*>(reinterpret_cast<bf16_t*>(ln.xg) + m * N + n0 + l * 4) = o4;
          if ((l & 15) == 0) {
            *reinterpret_cast<gg_f32x2*>(ln.stats_out + ((int64_t)((n0 + l * 4) / LN_GROUP) * ln.stats_ld + m) * 2) = gg_f32x2{gmean, gm2};
          }
        }
      } else if (m < M) {
        if constexpr (EPI == EPI_RESID) {
          const gg_f32x4 f = gg_f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
          *reinterpret_cast<gg_f32x4*>(resid + m * N + n0 + l * 4) = old[it] + f;
        } else if constexpr (F32OUT) {
          *reinterpret_cast<gg_u32x4*>(reinterpret_cast<float*>(out_) + m * N + n0 + l * 4) = v;
        } else {
          *reinterpret_cast<gg_u32x4*>(reinterpret_cast<bf16_t*>(out_) + m * N + n0 + l * 8) = v;
        }
      }
    }
    return;
  }
  // Tiles that reach beyond N (and the A/B knob "glds_epi" = 0): straight from the fragments -- one 8-byte (bf16) or 16-byte
  // (fp32) access per fragment.  N % 4 == 0.
  // Values are finished in one straight-line block (a single wait for the bias / residual loads);
  // the conditional blocks contain only the memory instructions, so no store waits for another.
  const bool full = m0 + BM <= M && n0 + BN <= N;  // block-uniform: no per-element bounds checks
  if constexpr (EPI == EPI_RESID) {
    gg_f32x4 old[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int64_t m = m0 + wm0 + i * 16 + fr;
        const int n = n0 + wn0 + j * 16 + fg * 4;
        old[i][j] = gg_f32x4{0.f, 0.f, 0.f, 0.f};
        if (full || (m < M && n < N)) old[i][j] = *reinterpret_cast<const gg_f32x4*>(resid + m * N + n);
      }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = old[i][j] + (acc[i][j] + bias4[LNC ? 0 : j]);
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acc[i][j] += bias4[LNC ? 0 : j];
        if constexpr (EPI == EPI_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
        }
      }
  }
  // pin the finished values here: without this the compiler sinks the adds (and their vmcnt(0)) into
  // every conditional store block, where each wait also drains the previous store
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]));
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int64_t m = m0 + wm0 + i * 16 + fr;
      const int n = n0 + wn0 + j * 16 + fg * 4;
      if (full || (m < M && n < N)) {
        if constexpr (EPI == EPI_RESID) {
          *reinterpret_cast<gg_f32x4*>(resid + m * N + n) = acc[i][j];
        } else if constexpr (EPI == EPI_F32 || F32IN) {
          *reinterpret_cast<gg_f32x4*>(reinterpret_cast<float*>(out_) + m * N + n) = acc[i][j];
        } else {
          gg_bf16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = (__bf16)acc[i][j][r];  // v_cvt_pk_bf16_f32: round-to-nearest-even
          *reinterpret_cast<gg_bf16x4*>(reinterpret_cast<bf16_t*>(out_) + m * N + n) = o4;
        }
      }
    }
}

// tile policy knob (vle_op_tune "glds_big"): 0 = never use the 8-wave 256 x 128 tile, -1 = default threshold
// (>= 512 full tiles), n > 0 = threshold n
int g_glds_big = -1;
int g_glds_8ph = -1;  // "glds_8ph": 0 = never gemm_8ph.hip, -1 = from 128 full 256 x 256 tiles (measured: ahead of the 256 x 128
                      // tile from there on, 463 vs 401 TF/s at 128 tiles, 970 vs 740 at 4112), n > 0 = from n tiles
int g_glds_epi = 1;   // "glds_epi": 1 = epilogue staged through LDS (whole-row stores), 0 = stores straight from the fragments (A/B knob)
int g_glds_swz = 0;   // "glds_swz": 1 = slot key (row >> 1) & 7 on the 8-wave tiles (A/B knob)
int g_glds_prio = 0;  // "glds_prio": s_setprio(1) around the MFMA cluster of the 8-wave tiles (A/B knob)
int g_glds_w8 = 1;  // "glds_w8": 8-wave workgroups on the 128-row tiles as well (batch-1 NAR 12.0 -> 10.0 ms); 0 = 4 waves

// the per-row pointers of a GemmLn, `rows` rows further down (leftover-row launches)
static GemmLn gg_ln_rows(const GemmLn* ln, int64_t rows, int N, int K) {
  GemmLn o = ln ? *ln : GemmLn();
  if (o.xg) o.xg = (bf16_t*)o.xg + rows * N;
  if (o.stats_out) o.stats_out += rows * 2;  // group-major: the row index is the fast one
  if (o.stats_in) o.stats_in += rows * 2;
  return o;
}

template <int BM, int BN, int NW = 4>
static int gg_launch(hipStream_t st, const bf16_t* A, const bf16_t* W, const float* bias, void* out, float* resid, int64_t M,
                     int N, int K, int epi, const GemmLn* lnp = nullptr) {
  const dim3 grid((N + BN - 1) / BN, (unsigned)((M + BM - 1) / BM)), block(NW * 64);
  const GemmLn ln = lnp ? *lnp : GemmLn();
#define VLE_GG(E)                                                                                                       \
  do {                                                                                                                  \
    if (NW == 8 && g_glds_swz)                                                                                          \
      hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, BM, BN, E, NW, false, (NW == 8) ? 1 : 0>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, g_glds_epi == 0, ln); \
    else if (NW == 8 && g_glds_prio)                                                                                    \
      hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, BM, BN, E, NW, (NW == 8), 0>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, g_glds_epi == 0, ln); \
    else                                                                                                                \
      hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, BM, BN, E, NW, false, 0>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, g_glds_epi == 0, ln);  \
  } while (0)
#define VLE_GG_LN(E) hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, BM, BN, E, NW, false, 0>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, 0, ln)
  switch (epi) {
    case EPI_STORE: VLE_GG(EPI_STORE); break;
    case EPI_RELU: VLE_GG(EPI_RELU); break;
    case EPI_RESID: VLE_GG(EPI_RESID); break;
    case EPI_F32: VLE_GG(EPI_F32); break;
    // LayerNorm folded into the GEMM (kernels.h GemmLn): the default body only (the swizzle / priority A-B variants stay un-folded)
    case EPI_RESID_LNP: if (!ln.gamma || !ln.xg || !ln.stats_out || ln.stats_ld < M || N % BN) return -1; VLE_GG_LN(EPI_RESID_LNP); break;
    case EPI_STORE_LNC: if (!ln.stats_in || !ln.sg || !bias || ln.stats_ld < M || N % BN || K > 1536 || K % 128) return -1; VLE_GG_LN(EPI_STORE_LNC); break;
    case EPI_RELU_LNC: if (!ln.stats_in || !ln.sg || !bias || ln.stats_ld < M || N % BN || K > 1536 || K % 128) return -1; VLE_GG_LN(EPI_RELU_LNC); break;
    default: return -1;
  }
#undef VLE_GG_LN
#undef VLE_GG
  return 0;
}

// returns 0 = launched, 1 = shape not covered (caller uses gemm.hip)
// "glds_tail" (default 1): rows beyond the last full 256-row tile of a gemm_8ph.hip launch go to a second, small launch when that
// saves a whole round of 256 x 256 tiles over the 256 CUs.  The NAR stages of B utterances of T = 1025 rows have M = 1024 B + B: the
// B leftover rows made a 257th tile row whose N / 256 tiles ran alone in an extra round -- 1028 tiles = 4 rounds + 4 tiles for
// linear2 / out-proj at 64 utterances (a fifth of the launch), 13 rounds instead of 12 for the in-projection, 17 instead of 16 for
// linear1.  Rows are independent, so the split changes no number.
int g_glds_tail = 1;
// "glds_t64": 128 x 64 tiles from this many of them, 64 x 64 tiles below.  96 until round 4; at 144 tiles of 128 x 64 (the N = 1024
// GEMMs of one utterance's NAR rows, M = 1025: 56 % of the CUs) the 272 tiles of 64 x 64 are faster -- NAR 8.52 -> 8.21 ms.
int g_glds_t64 = 160;

// fp32 operands (the token-exact engine mode, M >= 128 packed rows): 64 x 64 tiles, 4 waves, a 4-stage ring of 16 KB = 64 KB of LDS, so two
// workgroups share a CU (one's epilogue and barriers under the other's MFMAs) and the 17th tile row of M = 1025 costs no round of its
// own.  fp32 MFMA is 16x slower than bf16: the launch is MFMA-bound at any tile size, and the ring exists to keep that pipe fed.
// Bit-identical to gemm.hip's fp32 kernel (same per-element FMA chain).  "f32_glds" = 0: gemm.hip (A/B).
int g_f32_glds = 1;
int launch_gemm_glds_f32(hipStream_t st, const float* A, const float* W, const float* bias, void* out, float* resid, int64_t M, int N, int K, int epi) {
  if (!g_f32_glds || K % 32 != 0 || K < 32 || M < 128 || N < 4 || N % 4 != 0) return 1;
  constexpr int BM = 64, BN = 64, NW = 4;
  const dim3 grid((N + BN - 1) / BN, (unsigned)((M + BM - 1) / BM)), block(NW * 64);
  const GemmLn ln;
#define VLE_GGF(E) hipLaunchKernelGGL((gemm_glds_kernel<float, BM, BN, E, NW, false, 0>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, 0, ln)
  switch (epi) {
    case EPI_STORE: VLE_GGF(EPI_STORE); break;
    case EPI_RELU: VLE_GGF(EPI_RELU); break;
    case EPI_RESID: VLE_GGF(EPI_RESID); break;
    case EPI_F32: VLE_GGF(EPI_F32); break;
    default: return 1;
  }
#undef VLE_GGF
  return 0;
}

bool gemm_ln_supports(int dtype, int64_t M, int d) {
  return dtype == DT_BF16 && M >= 128 && d % 256 == 0 && d >= 256 && d <= 1536 && g_glds_epi != 0;
}

int launch_gemm_glds(hipStream_t st, const void* A, const void* W, const float* bias, void* out, float* resid, int64_t M, int N,
                     int K, int epi, const GemmLn* ln) {
  if (K % 64 != 0 || K < 64 || M < 1 || N < 1 || N % 4 != 0) return 1;
  if (epi >= EPI_RESID_LNP && (ln == nullptr || N % 256 != 0)) return -1;
  if (M < 128) return 1;  // tiny-M launches (AR step at batch > 8) stay on gemm.hip for now
  // tile choice: estimated time ~ rounds of full-cost tiles over 256 CUs x per-tile cost (~ BM*BN/eff)
  const int64_t full128 = (M / 128) * ((N + 127) / 128) + ((M % 128) ? ((N + 127) / 128) : 0);
  const int64_t t128x64 = ((M + 127) / 128) * ((N + 63) / 64);
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* w = (const bf16_t*)W;
  // many tiles (batched prefill / NAR rows): 256 x 128 with 8 waves -- two waves per SIMD cover each other's
  // ds_read -> MFMA latency, and the W panel is re-read half as often
  // enough 256 x 256 tiles: the 4-phase-per-K-tile schedule of gemm_8ph.hip
  if (g_glds_8ph != 0 && (M / 256) * (N / 256) >= (g_glds_8ph > 0 ? g_glds_8ph : 128)) {
    const int64_t rem = M % 256, ncol = N / 256, cus = 256;
    const bool split = g_glds_tail != 0 && rem > 0 && N % 256 == 0 && ((M / 256) * ncol + cus - 1) / cus < (((M + 255) / 256) * ncol + cus - 1) / cus;
    const int64_t Mmain = split ? M - rem : M;
    if (launch_gemm_8ph(st, A, W, bias, out, resid, Mmain, N, K, epi, ln) == 0) {
      if (!split) return 0;
      const bf16_t* a2 = a + Mmain * K;
      void* out2 = out == nullptr ? nullptr : (epi == EPI_F32 || epi == EPI_RESID || epi == EPI_RESID_LNP) ? (void*)((float*)out + Mmain * N) : (void*)((bf16_t*)out + Mmain * N);
      float* resid2 = resid == nullptr ? nullptr : resid + Mmain * N;
      const GemmLn ln2 = gg_ln_rows(ln, Mmain, N, K);
      if (rem >= 128) return launch_gemm_glds(st, a2, W, bias, out2, resid2, rem, N, K, epi, ln ? &ln2 : nullptr);
      return gg_launch<64, 64>(st, a2, w, bias, out2, resid2, rem, N, K, epi, ln ? &ln2 : nullptr);
    }
  }
  const int64_t t256 = (M / 256) * ((N + 127) / 128);
  if (g_glds_big != 0 && t256 >= (g_glds_big > 0 ? g_glds_big : 512)) return gg_launch<256, 128, 8>(st, a, w, bias, out, resid, M, N, K, epi, ln);
  if (full128 >= 160 && g_glds_tail != 0) {
    // the same for the 128 x 128 tiles: one utterance's NAR rows (M = 1025 = 8 x 128 + 1) made linear1 9 x 32 = 288 tiles, a second
    // round for 32 one-row tiles
    const int64_t rem = M % 128, ncol = (N + 127) / 128, cus = 256;
    if (rem > 0 && rem <= 64 && M >= 256 && ((M / 128) * ncol + cus - 1) / cus < (((M + 127) / 128) * ncol + cus - 1) / cus) {
      const int64_t Mmain = M - rem;
      const int r = launch_gemm_glds(st, A, W, bias, out, resid, Mmain, N, K, epi, ln);
      if (r != 0) return r;
      void* out2 = out == nullptr ? nullptr : (epi == EPI_F32 || epi == EPI_RESID || epi == EPI_RESID_LNP) ? (void*)((float*)out + Mmain * N) : (void*)((bf16_t*)out + Mmain * N);
      const GemmLn ln2 = gg_ln_rows(ln, Mmain, N, K);
      return gg_launch<64, 64>(st, a + Mmain * K, w, bias, out2, resid == nullptr ? nullptr : resid + Mmain * N, rem, N, K, epi, ln ? &ln2 : nullptr);
    }
  }
  if (g_glds_w8) {  // 8 waves on the one-tile-per-CU shapes too (wave tile 32 x 64 / 32 x 32)
    if (full128 >= 160) return gg_launch<128, 128, 8>(st, a, w, bias, out, resid, M, N, K, epi, ln);
    if (t128x64 >= g_glds_t64) return gg_launch<128, 64, 8>(st, a, w, bias, out, resid, M, N, K, epi, ln);
  }
  if (full128 >= 160) return gg_launch<128, 128>(st, a, w, bias, out, resid, M, N, K, epi, ln);
  if (t128x64 >= g_glds_t64) return gg_launch<128, 64>(st, a, w, bias, out, resid, M, N, K, epi, ln);
  return gg_launch<64, 64>(st, a, w, bias, out, resid, M, N, K, epi, ln);
}

}  // namespace vle
