// bf16 GEMM for the batched prefill / NAR rows (tens of thousands of rows): 256 x 256 tile, 8 waves, LDS-DMA staging
// in HALF-tiles, the k-loop cut into 4 phases per 64-deep K-tile so that LDS reads, DMA issue and MFMA of different
// waves overlap ("8-phase" schedule of /opt/skills/guides/cdna_hip_programming.md, 2 K-tiles per trip).
//   out = epi(A[M x K] @ W[N x K]^T + bias)      same contract and epilogues as gemm_glds.hip
//   reference ops: in-proj / out-proj `linear` (valle/modules/activation.py:414-421), FFN linear1 / linear2
//   (valle/modules/transformer.py:332-334) over the packed rows of valle.py:1035-1038, 1125-1127.
//
// Why another GEMM: gemm_glds.hip is one barrier per K-tile, every wave in lock step (all read LDS, then all issue
// MFMAs): measured 750-870 TF/s at 65 600 rows.  Here
//   * a wave's 128 x 64 outputs are FOUR 64 x 32 quadrants, one from each pair of tile halves:
//       rows  {wr*64 .. +64} of the tile's top half and of its bottom half,  wr = wave >> 2
//       cols  {wc*32 .. +32} of the tile's left half and of its right half,  wc = wave & 3
//     so in a given phase ALL waves read the same half-tiles, and a half-tile is dead for everyone at a known phase;
//   * per K-tile:  phase 1: read A-top (8 x ds_read_b128) + B-left (4), MFMA quadrant (top, left)     16 MFMAs
//                  phase 2: read B-right (4),                          MFMA (top, right)
//                  phase 3: read A-bot (8),                            MFMA (bot, right)
//                  phase 4: no reads (B-left stayed in registers),     MFMA (bot, left)
//     = 24 LDS reads per 64 MFMAs (gemm_glds.hip: 32 per 64);
//   * every phase issues ONE half-tile of LDS-DMA (2 x global_load_lds_dwordx4 per lane): phases 1 / 2 -> A-bot /
//     B-right of tile t+1 (other buffer), phases 3 / 4 -> B-left / A-top of tile t+2 (this buffer).  Issue order = order
//     of first use, so ONE counted wait per K-tile (phase 4: vmcnt(4) leaves the two newest half-tiles in flight)
//     retires exactly tile t+1; it sits before phase 4's first barrier and the data is first read in the next phase
//     (a DMA is ordered for another wave's ds_read only by the issuer's vmcnt followed by a barrier the reader passed);
//   * a half is overwritten no earlier than two phases after its last read: those reads were retired by that phase's
//     lgkmcnt(0) and every wave has passed two more barriers (enough also for the staggered wave groups);
//   * s_setprio(1) around each MFMA cluster: with the waves spread over different parts of a phase the scheduler can
//     prefer the ones entering MFMAs;
//   * LDS image of a half-tile: 128 rows x 128 B, 16-byte slot c of row r holds source vector c ^ (r & 7) (swizzle on
//     the DMA source address, same XOR on the ds_read side), 2 buffers x 4 halves x 16 KB = 128 KB.
// Shapes: N % 256 == 0, K % 128 == 0, K >= 256; rows beyond M are clamped on load and masked on store.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 g8_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g8_bf16x4 __attribute__((ext_vector_type(4)));
typedef float g8_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int g8_u32x4 __attribute__((ext_vector_type(4)));

constexpr int G8_HALF = 128 * 128;       // bytes of one half-tile (128 rows x 64 bf16)
constexpr int G8_BUF = 4 * G8_HALF;      // A-top, A-bot, B-left, B-right
enum { G8_ATOP = 0, G8_ABOT = 1, G8_BLEFT = 2, G8_BRIGHT = 3 };

template <int N>
__device__ inline void g8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// STAGGER: waves 4-7 run one barrier (half a phase) behind waves 0-3, so on every SIMD one wave is in its MFMA cluster
// while the other issues its LDS reads / DMA.  SWZ: slot key of the swizzle, 0 = row & 7, 1 = (row >> 1) & 7 (a 128-byte
// row covers half of the 64 banks and consecutive rows alternate halves: key 1 gives the 8 + 8 rows of a 16-lane
// ds_read_b128 group distinct slots inside each half).
template <int SWZ>
__device__ inline int g8_key(int row) {
  return SWZ ? ((row >> 1) & 7) : (row & 7);
}

template <int EPI, bool STAGGER, int SWZ>
__global__ __launch_bounds__(512) void gemm_8ph_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                       const float* __restrict__ bias, void* __restrict__ out_,
                                                       float* __restrict__ resid, int64_t M, int N, int K, int GC, int dbg, GemmLn ln) {
  // LayerNorm folded into the GEMMs (kernels.h GemmLn; same scheme as gemm_glds.hip): LNP = this launch completes the residual stream
  // and leaves bf16(x * gamma) + group statistics for the next norm site; LNC = this launch reads x * gamma and applies
  // rstd * (acc - mean * sg) + tb in its epilogue
  constexpr bool LNP = EPI == EPI_RESID_LNP, LNC = EPI == EPI_STORE_LNC || EPI == EPI_RELU_LNC;
  constexpr bool RESID = EPI == EPI_RESID || LNP, RELU = EPI == EPI_RELU || EPI == EPI_RELU_LNC, BF16OUT = EPI == EPI_STORE || RELU || LNC;
  // LNC: behind the two buffers (never a K-tile target): (mean, M2) of the two halves of the tile's rows, then sg and tb of its columns
  constexpr int LN_STATS = 256 * 16, LN_LDS = LNC ? LN_STATS + 2 * 256 * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G8_BUF + LN_LDS];  // the ONLY LDS object (a second one makes
                                                                                    // hipcc drain vmcnt before every ds_read)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;

  // XCD-aware tile order (block b runs on XCD b % 8: give each XCD a contiguous run of tiles sharing W panels)
  const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
  int bid = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Tile order inside that run: column groups of GC tiles swept down the rows.  In plain row-major order the 32 tiles
  // an XCD runs at a time span all N / 256 column panels, W (8 MB at N = 4096) cycles through the 4 MB L2 once per wave
  // of tiles and the kernel pulls ~9x its algorithmic bytes from the fabric (PMC: 1.23 GB per FFN1 launch); with GC = 4
  // an XCD keeps 4 W panels (2 MB) resident and streams the A panels past them.
  int tr, tc;
  if (GC > 1 && nbx % GC == 0) {
    const int per = (int)gridDim.y * GC, cg = bid / per, rem = bid - cg * per;
    tr = rem / GC;
    tc = cg * GC + rem % GC;
  } else {
    tr = bid / nbx;
    tc = bid % nbx;
  }
  const int64_t m0 = (int64_t)tr * 256;
  const int n0 = tc * 256;

  // ---- DMA sources: half-tile piece j (0, 1) of this lane covers half rows (j*8 + wave)*8 + (lane >> 3) ----
  const int prow = lane >> 3;
  const unsigned char* src[4][2];  // [half][piece], advanced by kt * 128 bytes
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int hr = (j * 8 + wave) * 8 + prow;                 // row inside the half (0..127)
    const int vec = (lane & 7) ^ g8_key<SWZ>(hr);             // swizzled source vector
    int64_t ga = m0 + hr, gb = m0 + 128 + hr;
    ga = ga < M ? ga : M - 1;
    gb = gb < M ? gb : M - 1;
    src[G8_ATOP][j] = reinterpret_cast<const unsigned char*>(A + ga * K) + vec * 16;
    src[G8_ABOT][j] = reinterpret_cast<const unsigned char*>(A + gb * K) + vec * 16;
    src[G8_BLEFT][j] = reinterpret_cast<const unsigned char*>(W + (int64_t)(n0 + hr) * K) + vec * 16;
    src[G8_BRIGHT][j] = reinterpret_cast<const unsigned char*>(W + (int64_t)(n0 + 128 + hr) * K) + vec * 16;
  }
  auto issue = [&](int half, int kt) {  // 2 x 1 KB per wave: one half-tile of K-tile kt into buffer kt & 1
    unsigned char* dst = smem + (kt & 1) * G8_BUF + half * G8_HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[half][j] + (int64_t)kt * 128),
                                       (__attribute__((address_space(3))) void*)(dst + (j * 8 + wave) * 1024), 16, 0, 0);
  };

  // bias of this lane's 4 consecutive output columns per n-fragment (requested ahead of the DMA queue)
  g8_f32x4 bias4[LNC ? 1 : 2][LNC ? 1 : 2];
  if constexpr (LNC) {
    // sg / tb of this tile's 256 columns into LDS, the first requests of the workgroup (a wave each: 64 lanes x 16 bytes): no
    // registers through the main loop, no exposed load in the epilogue
    bias4[0][0] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((wave == 0 ? ln.sg : bias) + n0 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(smem + 2 * G8_BUF + LN_STATS + wave * 1024), 16, 0, 0);
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + h * 128 + wc * 32 + j * 16 + fg * 4;
        bias4[h][j] = bias != nullptr ? *reinterpret_cast<const g8_f32x4*>(bias + n) : g8_f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }

  g8_f32x4 acc[2][2][4][2];  // [row half][col half][m-fragment][n-fragment]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment readers: 64-deep K-tile = 2 MFMA k-steps; slot = (ks*4 + fg) ^ (row & 7)
  g8_bf16x8 af[4][2], bl[2][2], brt[2][2];
  auto read_a = [&](const unsigned char* half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wr * 64 + i * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const g8_bf16x8*>(half + row * 128 + (((ks * 4 + fg) ^ g8_key<SWZ>(row)) << 4));
    }
  };
  auto read_b = [&](const unsigned char* half, g8_bf16x8 (&bf)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wc * 32 + j * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bf[j][ks] = *reinterpret_cast<const g8_bf16x8*>(half + row * 128 + (((ks * 4 + fg) ^ g8_key<SWZ>(row)) << 4));
    }
  };
  auto mma = [&](g8_f32x4 (&c)[4][2], const g8_bf16x8 (&bf)[2][2]) {  // one 64 x 32 quadrant x K = 64: 16 MFMAs
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)  // W fragment as the A operand: C^T (a lane owns 4 consecutive output columns)
          c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks], af[i][ks], c[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_sync = [&]() {  // reads of this phase retired for this wave, and every wave is here
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int KT = (dbg & 2) ? 2 : K / 64;  // diagnostic (g8_dbg bit 1): prologue + two K-tiles + epilogue only
  // LNC: the row statistics, as in gemm_glds.hip: thread t owns HALF of row m0 + t % 256 (half t / 256 of its K / 64 groups); the
  // (mean, M2) pairs are requested AHEAD of the K-tile queue (in-order return: the wait for them never drains the pipeline), one
  // coalesced 8-byte load per group (group-major layout) on clamped addresses; the two halves meet in the epilogue
  constexpr int LNV = 12;
  typedef float g8_f32x2 __attribute__((ext_vector_type(2)));
  g8_f32x2 lnp[LNC ? LNV : 1];
  const int ln_ng = K / 128;  // groups per half row
  if constexpr (LNC) {
    int64_t m = m0 + (tid & 255);
    m = m < M ? m : M - 1;
    const g8_f32x2* sp = reinterpret_cast<const g8_f32x2*>(ln.stats_in) + (int64_t)(tid >> 8) * ln_ng * ln.stats_ld + m;
#pragma unroll
    for (int g = 0; g < LNV; ++g) lnp[g] = sp[(int64_t)(g < ln_ng ? g : ln_ng - 1) * ln.stats_ld];
  }
  // Issue schedule (one half-tile per phase, in order of first use, into a half whose last read lies >= 2 phases back,
  // so it also holds when the two wave groups are half a phase apart):
  //   phase 1: A-bot of tile t+1      phase 2: B-right of tile t+1     (other buffer)
  //   phase 3: B-left of tile t+2     phase 4: A-top of tile t+2       (this buffer: both last read in phase 1)
  // and ONE counted wait per K-tile in phase 4: vmcnt(4) leaves the two halves of tile t+2 in flight and retires all of
  // tile t+1; it precedes phase 4's first barrier, the data is first read in the next phase.
  issue(G8_BLEFT, 0); issue(G8_ATOP, 0); issue(G8_BRIGHT, 0); issue(G8_ABOT, 0);
  issue(G8_BLEFT, 1); issue(G8_ATOP, 1);
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
    const float sm = s1 * __builtin_amdgcn_rcpf((float)ln_ng);  // mean_h - ref
    ln_mean = ref + sm;
    ln_m2 = q + (float)LN_GROUP * (s2 - s1 * sm);
  }
  g8_wait_vm<4>();  // tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();  // pairs with the first in-loop barrier of waves 0-3

  for (int kt = 0; kt < KT; ++kt) {
    const unsigned char* buf = smem + (kt & 1) * G8_BUF;
    // ---- phase 1: (top, left) --------------------------------------------------------------------
    read_b(buf + G8_BLEFT * G8_HALF, bl);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf + G8_ATOP * G8_HALF);
    if (kt + 1 < KT) issue(G8_ABOT, kt + 1);
    phase_sync();
    mma(acc[0][0], bl);
    phase_end();
    // ---- phase 2: (top, right) -------------------------------------------------------------------
    read_b(buf + G8_BRIGHT * G8_HALF, brt);
    if (kt + 1 < KT) issue(G8_BRIGHT, kt + 1);
    phase_sync();
    mma(acc[0][1], brt);
    phase_end();
    // ---- phase 3: (bot, right) -------------------------------------------------------------------
    read_a(buf + G8_ABOT * G8_HALF);
    if (kt + 2 < KT) issue(G8_BLEFT, kt + 2);
    phase_sync();
    mma(acc[1][1], brt);
    phase_end();
    // ---- phase 4: (bot, left), and the one counted wait of the K-tile --------------------------------
    if (kt + 2 < KT) {
      issue(G8_ATOP, kt + 2);
      g8_wait_vm<4>();
    } else {
      g8_wait_vm<0>();
    }
    phase_sync();
    mma(acc[1][0], bl);
    phase_end();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the last barrier of waves 4-7

  // ---- epilogue: lane (fg, fr) holds C[m = .. + 16 i + fr][n = .. + 16 j + 4 fg + r], r = 0..3 -----------------
  // Through LDS (free once the last K-tile is consumed): written straight from the fragments, a store instruction covers 16 rows x
  // 32 bytes (8-byte pieces of 16 different rows) and the store tail costs 20-50 % of the kernel (diagnostic knob g8_dbg = 1:
  // QKV 960 -> 1243 TF/s, out-proj 560 -> 1090 without the stores).  Staged as a row-major image, every global access is a whole
  // row of the tile: 512 bytes of bf16 (two rows per wave-instruction) or 1 KB of fp32 (one row), 16 bytes per lane.
  //   bf16 image  [256][256]: 16-byte chunk c of row r at chunk c ^ (r & 7), its 8-byte halves swapped when (r >> 3) & 1 --
  //               conflict-free for the fragment writes (ds_write_b64: 16 rows of one column chunk) and the row reads;
  //   fp32 image  [128][256], one row half of the tile per pass: chunk c of row r at c ^ (r & 7) (ds_write_b128 groups of 8 rows).
  if (!(dbg & 4)) {
    unsigned char* const E = smem;
    if constexpr (BF16OUT) {
      if constexpr (LNC) {
        float* const S = reinterpret_cast<float*>(smem + 2 * G8_BUF);  // the rows' (mean, rstd): owner threads -> fragment layout
        S[4 * (tid & 255) + 2 * (tid >> 8)] = ln_mean;
        S[4 * (tid & 255) + 2 * (tid >> 8) + 1] = ln_m2;
        __syncthreads();
      }
      g8_f32x4 sg4[LNC ? 2 : 1][LNC ? 2 : 1], tb4[LNC ? 2 : 1][LNC ? 2 : 1];  // this lane's columns of sg / tb, out of LDS once
      if constexpr (LNC) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = b * 128 + wc * 32 + j * 16 + fg * 4;
            sg4[b][j] = *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + LN_STATS + col * 4);
            tb4[b][j] = *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + LN_STATS + 1024 + col * 4);
          }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = a * 128 + wr * 64 + i * 16 + fr;
          float mean = 0.f, rstd = 1.f;
          if constexpr (LNC) {  // the row's two halves (K / 2 elements each): mean = (m0 + m1) / 2, M2 = q0 + q1 + (K / 4) (m0 - m1)^2
            const g8_f32x4 hh = *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + row * 16);
            const float dm = hh[0] - hh[2];
            mean = 0.5f * (hh[0] + hh[2]);
            rstd = __builtin_amdgcn_rsqf(fmaf(hh[1] + hh[3] + 0.25f * (float)K * dm * dm, __builtin_amdgcn_rcpf((float)K), LN_EPS));  // v_rcp / v_rsq: no IEEE sequences
          }
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c = b * 16 + wc * 4 + j * 2 + (fg >> 1);
              g8_f32x4 v;
              if constexpr (LNC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, fmaf(-mean, sg4[b][j][r], acc[a][b][i][j][r]), tb4[b][j][r]);
              } else {
                v = acc[a][b][i][j] + bias4[LNC ? 0 : b][LNC ? 0 : j];
              }
              if constexpr (RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              g8_bf16x4 o4;
#pragma unroll
              for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
              *reinterpret_cast<g8_bf16x4*>(E + row * 512 + ((c ^ (fr & 7)) << 4) + (((fg & 1) ^ (fr >> 3)) << 3)) = o4;
            }
        }
      __syncthreads();
      bf16_t* const outp = reinterpret_cast<bf16_t*>(out_);
      const int l = lane & 31;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = wave * 32 + it * 2 + (lane >> 5);
        g8_u32x4 v = *reinterpret_cast<const g8_u32x4*>(E + r * 512 + ((l ^ (r & 7)) << 4));
        if ((r >> 3) & 1) v = g8_u32x4{v[2], v[3], v[0], v[1]};
        const int64_t m = m0 + r;
        if (m < M && !(dbg & 1)) {
          if (dbg & 8) __builtin_nontemporal_store(v, reinterpret_cast<g8_u32x4*>(outp + m * N + n0 + l * 8));
          else *reinterpret_cast<g8_u32x4*>(outp + m * N + n0 + l * 8) = v;
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (a == 1) __syncthreads();  // the rows of pass 0 have been read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr * 64 + i * 16 + fr;
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c = b * 32 + wc * 8 + j * 4 + fg;
              *reinterpret_cast<g8_f32x4*>(E + row * 1024 + ((c ^ (fr & 7)) << 4)) = acc[a][b][i][j] + bias4[LNC ? 0 : b][LNC ? 0 : j];
            }
        }
        float* const base = (RESID ? resid : reinterpret_cast<float*>(out_)) + n0 + lane * 4;
        g8_f32x4 old[16];
        g8_f32x4 gamma4 = g8_f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (LNP) gamma4 = *reinterpret_cast<const g8_f32x4*>(ln.gamma + n0 + lane * 4);
        if constexpr (RESID) {  // the 16 rows' old values: requested before the barrier
#pragma unroll
          for (int it = 0; it < 16; ++it) {
            const int64_t m = m0 + a * 128 + wave * 16 + it;
            old[it] = (dbg & 16) ? __builtin_nontemporal_load(reinterpret_cast<const g8_f32x4*>(base + (m < M ? m : M - 1) * N))
                                 : *reinterpret_cast<const g8_f32x4*>(base + (m < M ? m : M - 1) * N);
          }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int r = wave * 16 + it;
          g8_f32x4 v = *reinterpret_cast<const g8_f32x4*>(E + r * 1024 + ((lane ^ (r & 7)) << 4));
          if constexpr (RESID) v = old[it] + v;
          const int64_t m = m0 + a * 128 + r;
          float gmean = 0.f, gm2 = 0.f;
          if constexpr (LNP) {  // (mean, M2) of this lane row's 64-column group of the completed residual row: exact two-pass, one DPP row
            gmean = row16_sum_dpp((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / (float)LN_GROUP);
            const float d0 = v[0] - gmean, d1 = v[1] - gmean, d2 = v[2] - gmean, d3 = v[3] - gmean;
            gm2 = row16_sum_dpp(fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0))));
          }
          if (m < M && !(dbg & 1)) {
            if (dbg & 16) __builtin_nontemporal_store(v, reinterpret_cast<g8_f32x4*>(base + m * N));
            else *reinterpret_cast<g8_f32x4*>(base + m * N) = v;
            if constexpr (LNP) {
              g8_bf16x4 o4;
#pragma unroll
              for (int q = 0; q < 4; ++q) o4[q] = (__bf16)(v[q] * gamma4[q]);
              *reinterpret_cast<g8_bf16x4*>(reinterpret_cast<bf16_t*>(ln.xg) + m * N + n0 + lane * 4) = o4;
              if ((lane & 15) == 0)
                *reinterpret_cast<g8_f32x2*>(ln.stats_out + ((int64_t)((n0 + lane * 4) / LN_GROUP) * ln.stats_ld + m) * 2) = g8_f32x2{gmean, gm2};
            }
          }
        }
      }
    }
    return;
  }
  // ---- legacy epilogue (g8_dbg bit 2): stores straight from the fragments ------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + a * 128 + wr * 64 + i * 16 + fr;
      if (m >= M) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + b * 128 + wc * 32 + j * 16 + fg * 4;
          g8_f32x4 v = acc[a][b][i][j] + bias4[LNC ? 0 : b][LNC ? 0 : j];
          if ((dbg & 1) && v[0] != 1.2345e30f) continue;  // diagnostic (g8_dbg bit 0): no epilogue stores
          if constexpr (EPI == EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if constexpr (EPI == EPI_RESID) {
            g8_f32x4* o = reinterpret_cast<g8_f32x4*>(resid + m * N + n);
            *o = *o + v;
          } else if constexpr (EPI == EPI_F32) {
            *reinterpret_cast<g8_f32x4*>(reinterpret_cast<float*>(out_) + m * N + n) = v;
          } else {
            g8_bf16x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
            *reinterpret_cast<g8_bf16x4*>(reinterpret_cast<bf16_t*>(out_) + m * N + n) = o4;
          }
        }
    }
}

// =====================================================================================================================
// Persistent tile loop (round 6; VERDICT r2-r5 "next": "a persistent tile loop whose epilogue shares the CU with the next tile's
// first K-tiles").  One workgroup per CU walks tiles w, w + G, w + 2G ... of the same XCD-aware order; the main loop is the one
// above, what changes is everything AROUND it:
//   * the K-loop's DMA queue runs ACROSS tiles: the four half-tile requests the last two K-tiles of a tile have no use for
//     (phases 3 / 4 of K-tile KT-2, phases 1 / 2 of K-tile KT-1) fetch K-tile 0 of the NEXT tile into buffer 0 (KT is even), so the
//     next main loop starts on data that landed during this tile's epilogue -- no launch ramp, no kernarg fetch, no exposed first
//     round trip (the one-tile kernel pays ~2.5 us of those per tile on a ~33 us K = 1024 tile);
//   * the epilogue stages through buffer 1 ONLY (64 KB: a 128-row half of a bf16 tile, a 128 x 128 quarter of an fp32 tile per
//     pass), because buffer 0 is already receiving; its barriers are raw s_barrier + lgkmcnt(0) -- a __syncthreads() would drain
//     vmcnt and wait for every store of the previous pass to be acknowledged;
//   * its global stores are never waited for: they drain while the next tile's MFMAs run (the one-tile kernel's workgroup cannot
//     end, and the CU cannot take its next tile, before they have left);
//   * the residual's old values are requested one pass AHEAD of the stores that precede them in program order (two register sets),
//     so a pass waits for loads that were issued before the previous pass's stores (a wave's memory operations retire in order);
//   * bias / sg / tb / row statistics of a tile are requested when the previous epilogue has issued its last store and are retired by
//     the first counted wait of the main loop.
// Per-element arithmetic, the group statistics' reduction order and every rounding are those of gemm_8ph_kernel: the two kernels
// are bit-identical (tests/test_ops_gpu.py::test_gemm_8ph_persistent_tile_loop_is_bit_identical).
template <int EPI, int LNV = 8>  // LNV: statistics groups per half row the LNC forms combine: K <= 1024 (with 12 for K = 1536 the epilogue's next-tile
                                // statistics do not fit beside the accumulators -- 7 spilled registers whose reloads drain vmcnt: those launches keep
                                // the one-tile kernel)
__global__ __launch_bounds__(512) void gemm_8ph_pl_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ out_,
                                                          float* __restrict__ resid, int64_t M, int N, int K, int ntr, int ntc, GemmLn ln) {
  constexpr int SWZ = 0;
  constexpr bool LNP = EPI == EPI_RESID_LNP, LNC = EPI == EPI_STORE_LNC || EPI == EPI_RELU_LNC;
  constexpr bool RESID = EPI == EPI_RESID || LNP, RELU = EPI == EPI_RELU || EPI == EPI_RELU_LNC, BF16OUT = EPI == EPI_STORE || RELU || LNC;
  constexpr int LN_STATS = 256 * 16, LN_LDS = LNC ? LN_STATS + 2 * 256 * 4 : 256 * 4;  // behind the two buffers: LNC row statistics + sg + tb; else the tile's bias
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G8_BUF + LN_LDS];  // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntiles = ntr * ntc, G = gridDim.x;
  const int prow = lane >> 3;
  const int KT = K / 64;  // even (K % 128 == 0)

  // tile id -> (row, column) of the tile grid: the XCD-aware order of gemm_8ph_kernel (workgroup w runs on XCD w % 8, and with G a
  // multiple of 8 so do all of its tiles)
  auto tile_rc = [&](int t, int& tr, int& tc) {
    const int q = ntiles / 8, r = ntiles % 8, xcd = t % 8, idx = t / 8;
    const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tr = b / ntc;
    tc = b - tr * ntc;
  };

  const unsigned char* src[4][2];  // [half][piece] of the tile the DMA queue is currently fetching for that half
  auto set_src = [&](int half, int64_t m0, int n0) {
    // (opaque copies: the next tile's coordinates are known at the top of the tile loop, and hipcc would compute all eight pointers
    // there and carry them through the K loop -- 16 registers the loop does not have)
    asm volatile("" : "+s"(m0), "+s"(n0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hr = (j * 8 + wave) * 8 + prow;
      const int vec = (lane & 7) ^ g8_key<SWZ>(hr);
      if (half == G8_ATOP || half == G8_ABOT) {
        int64_t g = m0 + (half == G8_ABOT ? 128 : 0) + hr;
        g = g < M ? g : M - 1;
        src[half][j] = reinterpret_cast<const unsigned char*>(A + g * K) + vec * 16;
      } else {
        src[half][j] = reinterpret_cast<const unsigned char*>(W + (int64_t)(n0 + (half == G8_BRIGHT ? 128 : 0) + hr) * K) + vec * 16;
      }
    }
  };
  auto issue = [&](int half, int kt) {  // half-tile of K-tile kt (of the tile src[half] points at) into buffer kt & 1
    unsigned char* dst = smem + (kt & 1) * G8_BUF + half * G8_HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[half][j] + (int64_t)kt * 128),
                                       (__attribute__((address_space(3))) void*)(dst + (j * 8 + wave) * 1024), 16, 0, 0);
  };

  g8_bf16x8 af[4][2], bl[2][2], brt[2][2];
  auto read_a = [&](const unsigned char* half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wr * 64 + i * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const g8_bf16x8*>(half + row * 128 + (((ks * 4 + fg) ^ g8_key<SWZ>(row)) << 4));
    }
  };
  auto read_b = [&](const unsigned char* half, g8_bf16x8 (&bf)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wc * 32 + j * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bf[j][ks] = *reinterpret_cast<const g8_bf16x8*>(half + row * 128 + (((ks * 4 + fg) ^ g8_key<SWZ>(row)) << 4));
    }
  };
  auto mma = [&](g8_f32x4 (&c)[4][2], const g8_bf16x8 (&bf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks], af[i][ks], c[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_sync = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lds_barrier = [&]() {  // the epilogue's barrier: LDS traffic of this wave retired, no vmcnt drain
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  typedef float g8_f32x2 __attribute__((ext_vector_type(2)));
  const int ln_ng = K / 128;  // statistics groups per half row
  // per-tile operands: the bias (non-LNC) or sg / tb (LNC) of the tile's 256 columns go to LDS by DMA -- no registers through the main
  // loop, no load in the epilogue; they are older than every K-tile the loop waits for --; LNC: this thread's half-row statistics
  const bool has_bias = bias != nullptr;
  float ln_mean = 0.f, ln_m2 = 0.f;
  // LNC: thread t owns HALF of row m0 + t % 256 (half t / 256 of its K / 64 statistics groups): one coalesced 8-byte load per group
  // (group-major layout) on clamped addresses, reduced to the half row's (mean, M2) (Chan; squares taken around the first group)
  auto stats_request = [&](int64_t m0, g8_f32x2 (&lnp)[LNV]) {
    int64_t m = m0 + (tid & 255);
    m = m < M ? m : M - 1;
    const g8_f32x2* sp = reinterpret_cast<const g8_f32x2*>(ln.stats_in) + (int64_t)(tid >> 8) * ln_ng * ln.stats_ld + m;
#pragma unroll
    for (int g = 0; g < LNV; ++g) lnp[g] = sp[(int64_t)(g < ln_ng ? g : ln_ng - 1) * ln.stats_ld];
  };
  auto stats_reduce = [&](const g8_f32x2 (&lnp)[LNV], float& mean, float& m2) {
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
    const float sm = s1 * __builtin_amdgcn_rcpf((float)ln_ng);
    mean = ref + sm;
    m2 = q + (float)LN_GROUP * (s2 - s1 * sm);
  };
  // the tile's column constants into LDS by DMA: sg / tb (LNC) or the bias
  auto request_tile_columns = [&](int n0) {
    int lane = tid & 63;  // (opaque: hipcc otherwise keeps the per-lane source pointer in scratch across the tile loop)
    asm volatile("" : "+v"(lane));
    if constexpr (LNC) {
      if (wave < 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((wave == 0 ? ln.sg : bias) + n0 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + 2 * G8_BUF + LN_STATS + wave * 1024), 16, 0, 0);
    } else {
      if (wave == 0 && has_bias)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bias + n0 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + 2 * G8_BUF), 16, 0, 0);
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  int tr, tc;
  tile_rc(t, tr, tc);
  int64_t m0 = (int64_t)tr * 256;
  int n0 = tc * 256;
  // ---- prologue of the workgroup's first tile (as gemm_8ph_kernel) ----
  set_src(G8_ATOP, m0, n0); set_src(G8_ABOT, m0, n0); set_src(G8_BLEFT, m0, n0); set_src(G8_BRIGHT, m0, n0);
  issue(G8_BLEFT, 0); issue(G8_ATOP, 0); issue(G8_BRIGHT, 0); issue(G8_ABOT, 0);
  request_tile_columns(n0);
  if constexpr (LNC) {
    g8_f32x2 lnp[LNV];
    stats_request(m0, lnp);
    stats_reduce(lnp, ln_mean, ln_m2);
  }
  issue(G8_BLEFT, 1); issue(G8_ATOP, 1);
  g8_wait_vm<4>();
  __builtin_amdgcn_s_barrier();

  for (;;) {
    __builtin_amdgcn_sched_barrier(0);
    const int tn = t + G;
    const bool has_next = tn < ntiles;
    int ntr_ = 0, ntc_ = 0;
    if (has_next) tile_rc(tn, ntr_, ntc_);
    const int64_t nm0 = (int64_t)ntr_ * 256;
    const int nn0 = ntc_ * 256;

    g8_f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][b][i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};

    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 half a phase behind (pairs with the first in-loop barrier of waves 0-3)
    for (int kt = 0; kt < KT; ++kt) {
      const bool LAST = kt == KT - 1;
      const unsigned char* buf = smem + (kt & 1) * G8_BUF;
      // ---- phase 1 ----
      read_b(buf + G8_BLEFT * G8_HALF, bl);
      __builtin_amdgcn_sched_barrier(0);
      read_a(buf + G8_ATOP * G8_HALF);
      if (!LAST) issue(G8_ABOT, kt + 1);
      else if (has_next) {
        set_src(G8_ABOT, nm0, nn0);
        issue(G8_ABOT, 0);
      }
      phase_sync();
      mma(acc[0][0], bl);
      phase_end();
      // ---- phase 2 ----
      read_b(buf + G8_BRIGHT * G8_HALF, brt);
      if (!LAST) issue(G8_BRIGHT, kt + 1);
      else if (has_next) { set_src(G8_BRIGHT, nm0, nn0); issue(G8_BRIGHT, 0); }
      phase_sync();
      mma(acc[0][1], brt);
      phase_end();
      // ---- phase 3 ----
      read_a(buf + G8_ABOT * G8_HALF);
      if (!LAST) {
        if (kt + 2 < KT) issue(G8_BLEFT, kt + 2);
        else if (has_next) { set_src(G8_BLEFT, nm0, nn0); issue(G8_BLEFT, 0); }  // kt == KT - 2
      }
      phase_sync();
      mma(acc[1][1], brt);
      phase_end();
      // ---- phase 4: the one counted wait of the K-tile: everything but the two newest half-tiles (K-tile kt + 2, or the next tile's
      //      K-tile 0) has landed, i.e. all of K-tile kt + 1.  (Measured and dropped, round 6: TWO counted waits per K-tile -- vmcnt(8) here
      //      for B-left / A-top only, vmcnt(4) in the next phase 1 for A-bot / B-right, which this wait retires only 2-3 phases after
      //      their request -- bit-identical and 1-3 % SLOWER, profiles/r06_gemm_8ph_persist.json: this wait is not where the loop stalls.) ----
      if (!LAST) {
        if (kt + 2 < KT) {
          issue(G8_ATOP, kt + 2);
          g8_wait_vm<4>();
        } else if (has_next) {
          set_src(G8_ATOP, nm0, nn0);
          issue(G8_ATOP, 0);
          g8_wait_vm<4>();
        } else {
          g8_wait_vm<0>();
        }
      }  // (LAST: nothing of this tile is in flight any more)
      phase_sync();
      mma(acc[1][0], bl);
      phase_end();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the last barrier of waves 4-7: everybody has read buffer 1 for the last time

    // ---- epilogue through buffer 1 ----
    // (thread-index terms re-derived from an opaque copy: left alone, hipcc hoists the epilogue's ~40 per-lane address terms out of the
    // tile loop and keeps them in scratch across the main loop -- scratch accesses count in vmcnt like any other memory operation)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    const int wave_e = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int wr_e = wave_e >> 2, wc_e = wave_e & 3, fr_e = lane_e & 15, fg_e = lane_e >> 4;
    // Global traffic of the epilogue goes through buffer descriptors of the TILE (wave-uniform base = the tile's first element, extent =
    // its valid rows): one 32-bit lane offset per access (the per-iteration row step is a scalar add onto it -- NOT the instruction's
    // scalar-offset field, which the raw-buffer range check leaves out), and rows beyond M are dropped / read as zero by the hardware
    // bounds check -- no 64-bit address per row (the flat form kept ~16 of them per pass live
    // and spilled the old values), no branch around a store, so hipcc counts the queue exactly (vmcnt(8) for a pass's old values).
    unsigned char* const E = smem + G8_BUF;
    const int rows_valid = (int)((M - m0) < 256 ? (M - m0) : 256);
    if constexpr (BF16OUT) {
      g8_f32x4 sg4[LNC ? 2 : 1][LNC ? 2 : 1], tb4[LNC ? 2 : 1][LNC ? 2 : 1];
      // LNC: the NEXT tile's row statistics are requested here -- ahead of every store of this epilogue, so that the wait for them
      // (with this wave's pieces of the next K-tile 0, before pass 0's stores) never waits for a store.  (Requested during the last
      // K-tile instead -- a phase or four earlier -- they cost 9 spilled registers whose reloads drain vmcnt inside the K loop.)
      g8_f32x2 lnq[LNC ? LNV : 1];
      float n_mean = 0.f, n_m2 = 0.f;
      if constexpr (LNC) {
        if (has_next) stats_request(nm0, lnq);
      }
      if constexpr (LNC) {
        float* const S = reinterpret_cast<float*>(smem + 2 * G8_BUF);
        S[4 * (tid_e & 255) + 2 * (tid_e >> 8)] = ln_mean;
        S[4 * (tid_e & 255) + 2 * (tid_e >> 8) + 1] = ln_m2;
        lds_barrier();  // (sg / tb landed long ago: they are older than every K-tile the loop waited for)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = b * 128 + wc_e * 32 + j * 16 + fg_e * 4;
            sg4[b][j] = *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + LN_STATS + col * 4);
            tb4[b][j] = *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + LN_STATS + 1024 + col * 4);
          }
      }
      g8_f32x4 bias4[LNC ? 1 : 2][LNC ? 1 : 2];
      if constexpr (!LNC) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bias4[b][j] = has_bias ? *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + (b * 128 + wc_e * 32 + j * 16 + fg_e * 4) * 4) : g8_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<bf16_t*>(out_) + m0 * N + n0), 0,
                                                                            ((rows_valid - 1) * N + 256) * 2, 0x00020000);
      const int l = lane_e & 31;
      const int voff = ((wave_e * 16 + (lane_e >> 5)) * N + l * 8) * 2;  // row wave * 16 + (lane >> 5) of the half, 16 bytes per lane
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (a == 1) lds_barrier();  // the rows of pass 0 have been read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr_e * 64 + i * 16 + fr_e;  // inside the 128-row half
          float mean = 0.f, rstd = 1.f;
          if constexpr (LNC) {
            const g8_f32x4 hh = *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + (a * 128 + row) * 16);
            const float dm = hh[0] - hh[2];
            mean = 0.5f * (hh[0] + hh[2]);
            rstd = __builtin_amdgcn_rsqf(fmaf(hh[1] + hh[3] + 0.25f * (float)K * dm * dm, __builtin_amdgcn_rcpf((float)K), LN_EPS));
          }
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c = b * 16 + wc_e * 4 + j * 2 + (fg_e >> 1);
              g8_f32x4 v;
              if constexpr (LNC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, fmaf(-mean, sg4[b][j][r], acc[a][b][i][j][r]), tb4[b][j][r]);
              } else {
                v = acc[a][b][i][j] + bias4[LNC ? 0 : b][LNC ? 0 : j];
              }
              if constexpr (RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              g8_bf16x4 o4;
#pragma unroll
              for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
              *reinterpret_cast<g8_bf16x4*>(E + row * 512 + ((c ^ (fr_e & 7)) << 4) + (((fg_e & 1) ^ (fr_e >> 3)) << 3)) = o4;
            }
        }
        // this wave's pieces of the next tile's K-tile 0 (requested >= 2.5 phases ago) are retired HERE, before the first store: behind the
        // stores a counted wait could not tell them apart (the last lds_barrier of the epilogue then orders them for the other waves)
        if (a == 0) {
          g8_wait_vm<0>();
          if constexpr (LNC) {
            if (has_next) stats_reduce(lnq, n_mean, n_m2);
            asm volatile("" : "+v"(n_mean), "+v"(n_m2));  // HERE, on the values the wait above retired -- not sunk below the stores
          }
        }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int r = wave_e * 16 + it * 2 + (lane_e >> 5);
          g8_u32x4 v = *reinterpret_cast<const g8_u32x4*>(E + r * 512 + ((l ^ (r & 7)) << 4));
          if ((r >> 3) & 1) v = g8_u32x4{v[2], v[3], v[0], v[1]};
          __builtin_amdgcn_raw_buffer_store_b128(v, ors, voff + (a * 128 + it * 2) * N * 2, 0, 0);  // (row step in the LANE offset: the bounds check covers it)
        }
      }
      if constexpr (LNC) { ln_mean = n_mean; ln_m2 = n_m2; }
    } else {
      // fp32 output: four 128 x 128 quarters (row half a, column half b), 64 KB each.  Lane l = lane & 31 owns 4 consecutive columns
      // of a quarter row (16 lanes = one 64-column statistics group), a wave-instruction covers two rows.
      const int l = lane_e & 31, rsub = lane_e >> 5;
      float* const fbase = (RESID ? resid : reinterpret_cast<float*>(out_)) + m0 * N + n0;
      const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)fbase, 0, ((rows_valid - 1) * N + 256) * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(LNP ? (void*)(reinterpret_cast<bf16_t*>(ln.xg) + m0 * N + n0) : (void*)fbase, 0,
                                                                            LNP ? ((rows_valid - 1) * N + 256) * 2 : 0, 0x00020000);
      const int voff = ((wave_e * 16 + rsub) * N + l * 4) * 4;  // fp32 element (row wave * 16 + rsub, column 4 l) of a quarter
      g8_f32x4 gamma2[2] = {g8_f32x4{0.f, 0.f, 0.f, 0.f}, g8_f32x4{0.f, 0.f, 0.f, 0.f}};  // requested FIRST: the waits for the old values then count past it
      if constexpr (LNP) {
        gamma2[0] = *reinterpret_cast<const g8_f32x4*>(ln.gamma + n0 + l * 4);
        gamma2[1] = *reinterpret_cast<const g8_f32x4*>(ln.gamma + n0 + 128 + l * 4);
      }
      g8_u32x4 old[2][RESID ? 8 : 1];  // two sets: a pass's old values are requested one pass ahead (three sets, two passes ahead, do not fit: 26 spills)
      auto request_old = [&](int p) {  // pass p = 2 a + b
        if constexpr (RESID) {
#pragma unroll
          for (int it = 0; it < 8; ++it)
            old[p & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(frs, voff + (((p >> 1) * 128 + it * 2) * N + (p & 1) * 128) * 4, 0, 0);
        }
      };
      request_old(0);
      {  // the bias joins the accumulators here, once
        g8_f32x4 bias4[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bias4[b][j] = has_bias ? *reinterpret_cast<const g8_f32x4*>(smem + 2 * G8_BUF + (b * 128 + wc_e * 32 + j * 16 + fg_e * 4) * 4) : g8_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[a][b][i][j] = acc[a][b][i][j] + bias4[b][j];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int a = p >> 1, b = p & 1;
        const g8_f32x4 gamma4 = gamma2[b];
        if (p > 0) lds_barrier();  // the rows of the previous pass have been read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr_e * 64 + i * 16 + fr_e;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = wc_e * 8 + j * 4 + fg_e;
            *reinterpret_cast<g8_f32x4*>(E + row * 512 + ((c ^ (fr_e & 7)) << 4)) = acc[a][b][i][j];
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // (not above the fragment writes: their accumulators are the registers the old values land in)
        if (p < 3) request_old(p + 1);  // ahead of this pass's stores in the wave's memory queue
        // (RESID: the wait for pass 0's old values retires this wave's older pieces of the next tile's K-tile 0 with them -- a wave's
        // memory operations retire in order; without old values they are retired explicitly before the first store)
        if constexpr (!RESID) { if (p == 0) g8_wait_vm<0>(); }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int r = wave_e * 16 + it * 2 + rsub;
          g8_f32x4 v = *reinterpret_cast<const g8_f32x4*>(E + r * 512 + ((l ^ (r & 7)) << 4));
          if constexpr (RESID) v = __builtin_bit_cast(g8_f32x4, old[p & 1][it]) + v;
          float gmean = 0.f, gm2 = 0.f;
          if constexpr (LNP) {
            gmean = row16_sum_dpp((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / (float)LN_GROUP);
            const float d0 = v[0] - gmean, d1 = v[1] - gmean, d2 = v[2] - gmean, d3 = v[3] - gmean;
            gm2 = row16_sum_dpp(fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0))));
          }
          const int soff = ((a * 128 + it * 2) * N + b * 128) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g8_u32x4, v), frs, voff + soff, 0, 0);
          if constexpr (LNP) {
            g8_bf16x4 o4;
#pragma unroll
            for (int q = 0; q < 4; ++q) o4[q] = (__bf16)(v[q] * gamma4[q]);
            typedef unsigned int g8_u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(g8_u32x2, o4), xrs, (voff + soff) >> 1, 0, 0);
            const int64_t m = m0 + a * 128 + r;
            if (m < M && (lane_e & 15) == 0)
              *reinterpret_cast<g8_f32x2*>(ln.stats_out + ((int64_t)((n0 + b * 128 + l * 4) / LN_GROUP) * ln.stats_ld + m) * 2) = g8_f32x2{gmean, gm2};
          }
        }
      }
    }
    if (!has_next) return;
    // ---- next tile: its K-tile 0 is in buffer 0 (requested during the last two K-tiles, landed during the epilogue) ----
    t = tn; m0 = nm0; n0 = nn0;
    lds_barrier();  // every wave has read its rows out of buffer 1 (and S): K-tile 1 may land there, sg / tb may be replaced
    request_tile_columns(n0);
    // (the four source sets are re-derived here rather than carried through the epilogue: 16 registers it needs)
    set_src(G8_ATOP, m0, n0); set_src(G8_ABOT, m0, n0); set_src(G8_BLEFT, m0, n0); set_src(G8_BRIGHT, m0, n0);
    issue(G8_BLEFT, 1); issue(G8_ATOP, 1);
    // no wait here: K-tile 0 was retired inside the epilogue (above), and a counted wait would also wait for the epilogue's stores
  }
}

// "g8_persist": which launches take the persistent tile loop (gemm_8ph_pl_kernel) -- bit 0: the bf16-output forms (STORE / RELU / the
// LayerNorm consumers), bit 1: the fp32 / residual forms at K < 2048 (out-proj), bit 2: those at K >= 2048 (linear2); 0 = one tile per
// workgroup everywhere (A/B).  Default 1: measured at 64 utterances to the length cap (profiles/r06_gemm_8ph_persist.json), NAR phase
// 176.3 ms (0) -> 168.4 (1) / 169.5 (3) / 169.3 (7): the residual forms gain nothing in the engine -- their epilogue waits for the
// residual's old values (HBM, two register sets = one pass ahead; three sets spill), which the one-tile kernel's next workgroup hides
int g_g8_persist = 1;
int g_g8_nt = 0;        // "g8_nt": non-temporal epilogue traffic -- 1 the bf16 output tiles, 2 the fp32 residual read-modify-write, 3 both
int g_g8_dbg = 0;       // "g8_dbg": diagnostics -- 1 no epilogue stores, 2 two K-tiles only (what do prologue / epilogue cost?), 4 legacy epilogue
int g_g8_colgroup = 0;  // "g8_colgroup": column tiles per group of the tile order (0 / 1 = row-major)
int g_g8_stagger = 1;  // "g8_stagger": waves 4-7 half a phase behind waves 0-3 (measured +7-10 %: 905 -> 970 TF/s); 0 = lock step

// returns 0 = launched, 1 = shape not covered
int launch_gemm_8ph(hipStream_t st, const void* A, const void* W, const float* bias, void* out, float* resid, int64_t M, int N,
                    int K, int epi, const GemmLn* lnp) {
  if (N % 256 != 0 || K % 128 != 0 || K < 256 || M < 256) return 1;
  const dim3 grid(N / 256, (unsigned)((M + 255) / 256)), block(512);
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* w = (const bf16_t*)W;
  const GemmLn ln = lnp ? *lnp : GemmLn();
  const bool lnc_epi = epi == EPI_STORE_LNC || epi == EPI_RELU_LNC;
  const bool bf16_epi = epi == EPI_STORE || epi == EPI_RELU || lnc_epi;
  const int pbit = bf16_epi ? 1 : K < 2048 ? 2 : 4;
  if ((g_g8_persist & pbit) && g_g8_stagger && !g_glds_swz && g_g8_colgroup <= 1 && g_g8_dbg == 0 && g_g8_nt == 0 && !(lnc_epi && K > 1024)) {
    // the persistent tile loop: one workgroup per CU (more tiles than CUs: each walks several; fewer: one each, as before)
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) (void)hipGetLastError();
      return n >= 8 ? n / 8 * 8 : 256;  // a multiple of 8: a workgroup's tiles stay on its XCD's share of the tile order
    }();
    const int ntr = (int)((M + 255) / 256), ntc = N / 256;
    const dim3 pgrid((unsigned)std::min(ntr * ntc, cus));
#define VLE_G8P(E) hipLaunchKernelGGL((gemm_8ph_pl_kernel<E>), pgrid, block, 0, st, a, w, bias, out, resid, M, N, K, ntr, ntc, ln)
    switch (epi) {
      case EPI_STORE: VLE_G8P(EPI_STORE); return 0;
      case EPI_RELU: VLE_G8P(EPI_RELU); return 0;
      case EPI_RESID: VLE_G8P(EPI_RESID); return 0;
      case EPI_F32: VLE_G8P(EPI_F32); return 0;
      case EPI_RESID_LNP: if (!ln.gamma || !ln.xg || !ln.stats_out || ln.stats_ld < M) return -1; VLE_G8P(EPI_RESID_LNP); return 0;
      case EPI_STORE_LNC: if (!ln.stats_in || !ln.sg || !bias || ln.stats_ld < M) return -1; VLE_G8P(EPI_STORE_LNC); return 0;
      case EPI_RELU_LNC: if (!ln.stats_in || !ln.sg || !bias || ln.stats_ld < M) return -1; VLE_G8P(EPI_RELU_LNC); return 0;
      default: return -1;
    }
#undef VLE_G8P
  }
#define VLE_G8(E, ST, SW) hipLaunchKernelGGL((gemm_8ph_kernel<E, ST, SW>), grid, block, 0, st, a, w, bias, out, resid, M, N, K, g_g8_colgroup, g_g8_dbg | (g_g8_nt << 3), ln)
  if (epi >= EPI_RESID_LNP) {  // LayerNorm folded into the GEMM (kernels.h GemmLn): the default schedule only (staggered, row & 7 swizzle)
    if ((g_g8_dbg & 7) != 0) return -1;
    switch (epi) {
      case EPI_RESID_LNP: if (!ln.gamma || !ln.xg || !ln.stats_out || ln.stats_ld < M) return -1; VLE_G8(EPI_RESID_LNP, true, 0); break;
      case EPI_STORE_LNC: if (!ln.stats_in || !ln.sg || !bias || ln.stats_ld < M || K > 1536) return -1; VLE_G8(EPI_STORE_LNC, true, 0); break;
      case EPI_RELU_LNC: if (!ln.stats_in || !ln.sg || !bias || ln.stats_ld < M || K > 1536) return -1; VLE_G8(EPI_RELU_LNC, true, 0); break;
      default: return -1;
    }
    return 0;
  }
#define VLE_G8E(ST, SW)                           \
  switch (epi) {                                  \
    case EPI_STORE: VLE_G8(EPI_STORE, ST, SW); break; \
    case EPI_RELU: VLE_G8(EPI_RELU, ST, SW); break;   \
    case EPI_RESID: VLE_G8(EPI_RESID, ST, SW); break; \
    case EPI_F32: VLE_G8(EPI_F32, ST, SW); break;     \
    default: return -1;                           \
  }
  if (g_g8_stagger && g_glds_swz) { VLE_G8E(true, 1) }
  else if (g_g8_stagger) { VLE_G8E(true, 0) }
  else if (g_glds_swz) { VLE_G8E(false, 1) }
  else { VLE_G8E(false, 0) }
#undef VLE_G8E
#undef VLE_G8
  return 0;
}

}  // namespace vle
