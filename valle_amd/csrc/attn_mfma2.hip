// Flash-style MFMA attention for the prefill and the 7 NAR passes, second generation (bf16, head size 32 / 64 / 96 / 128).
//   reference: F.multi_head_attention_forward as called from MultiheadAttention.forward
//   (valle/modules/activation.py:408-427): softmax(Q K^T / sqrt(dh) + mask) V per head; AR mask = prefix-LM
//   (valle/models/valle.py:1019-1033), NAR: none.  Same contract as attn_mfma.hip, which it replaces where it applies.
//
// What round 1's kernel spent its time on (profiles/README.md: 18.6 % MFMA utilisation, K/V tiles re-read 5x from the
// fabric): sixteen 2-byte LDS stores per thread per tile to build V^T, two block barriers per 64-key tile around a single
// LDS buffer, and ~9 VALU instructions per score.  Here:
//   * V^T is built ONCE per (sequence, head) by a pre-pass (vt_pack_kernel) into a global scratch, key-permuted the way the
//     second MFMA wants it, instead of once per (query block, tile): the main kernel stages K rows and V^T rows with plain
//     16-byte loads / stores;
//   * two LDS buffers, ONE barrier per tile: tile t+1 is written while tile t is being multiplied, the global loads of
//     tile t+2 are issued a whole tile ahead of the store that needs them;
//   * 128 queries per block (each K / V^T fragment read from LDS feeds two MFMAs, the tiles are re-read from L2 half as often);
//   * softmax in ~5 VALU per score: the 1/sqrt(dh) log2 e factor rides on the exp2 argument's FMA (max is taken on raw scores),
//     the causal / length mask is evaluated only on tiles that cross a boundary (wave-uniform test), and the accumulator is
//     rescaled only when some row's maximum actually grew (wave-uniform test; exact: alpha = 1 otherwise).
// Both products are still computed transposed (S^T = K Q^T, O^T = V^T P^T) so that P leaves the first MFMA in the operand
// layout of the second (see attn_mfma.hip for the lane maps; am_vpos below is the key permutation of a V^T row).
#include <mutex>

#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 a2_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 a2_bf16x4 __attribute__((ext_vector_type(4)));
typedef float a2_f32x4 __attribute__((ext_vector_type(4)));
typedef float a2_f32x2 __attribute__((ext_vector_type(2)));
typedef short a2_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int a2_u32x4 __attribute__((ext_vector_type(4)));

constexpr float A2_NEG = -1e30f;

__device__ __host__ inline int a2_vpos(int key) {  // slot of key (0..63) inside a 64-key V^T tile row
  return ((key >> 5) << 5) + (((key & 15) >> 2) << 3) + (((key >> 4) & 1) << 2) + (key & 3);
}
// first V^T column of sequence b: 64-aligned, and far enough from the previous sequence's last (padded) tile
__device__ __host__ inline int64_t a2_vt_start(int off, int b) { return (int64_t)(off & ~63) + 128 * (int64_t)b; }

// ---- pre-pass: VT[h * DH + e][a2_vt_start(b) + tile * 64 + a2_vpos(key)] = V[b][tile * 64 + key][h][e] (0 beyond the sequence) ----
template <int DH>
__global__ __launch_bounds__(256) void vt_pack_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ vt,
                                                      const int32_t* __restrict__ seq_off, int d, int64_t rp) {
  constexpr int STR = 64 + 8;  // elements per LDS row (padding: the 2-byte scatter spreads over the banks)
  __shared__ uint16_t tile[DH * STR];
  const int b = blockIdx.z, h = blockIdx.y, kt0 = blockIdx.x * 64;
  const int off = seq_off[b], len = seq_off[b + 1] - off;
  if (kt0 >= len) return;
  const int tid = threadIdx.x;
  const int key = tid & 63;
  const bool live = kt0 + key < len;
  const bf16_t* src = qkv + (int64_t)(off + min(kt0 + key, len - 1)) * 3 * d + 2 * d + h * DH;
  for (int vv = tid >> 6; vv < DH / 8; vv += 4) {
    a2_u32x4 v = *reinterpret_cast<const a2_u32x4*>(src + vv * 8);
    if (!live) v = a2_u32x4{0u, 0u, 0u, 0u};
    uint16_t* dst = tile + (vv * 8) * STR + a2_vpos(key);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      dst[(2 * t) * STR] = (uint16_t)(v[t] & 0xffffu);
      dst[(2 * t + 1) * STR] = (uint16_t)(v[t] >> 16);
    }
  }
  __syncthreads();
  const int64_t col0 = a2_vt_start(off, b) + kt0;
  for (int idx = tid; idx < DH * 8; idx += 256) {
    const int e = idx >> 3, vec = idx & 7;
    *reinterpret_cast<a2_u32x4*>(vt + ((int64_t)h * DH + e) * rp + col0 + vec * 8) = *reinterpret_cast<const a2_u32x4*>(tile + e * STR + vec * 8);
  }
}

// XOR key of the 16-byte slot swizzle of a row of NV slots (LDS-DMA modes): conflict-free for the ds_read_b128 fragment reads
// AND the ds_read_b64_tr_b16 transpose reads under the lane groups of MI355X_MICROARCH.md's LDS table (checked exhaustively on
// the host): rows of 8 / 16 slots (dh 64 / 128) row & (NV - 1); rows of 4 / 12 slots (dh 32 / 96) (row >> 1) & 3 -- the key
// must stay inside an aligned group of 4 slots there, and consecutive rows start half a bank set apart.
template <int NV>
__device__ __host__ constexpr int a2_key(int row) {
  return (NV == 8 || NV == 16) ? (row & (NV - 1)) : ((row >> 1) & 3);
}

template <int N>
__device__ inline void a2_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: tiles staged through registers (global -> VGPR -> ds_write_b128), rows padded by 16 bytes (round 2's first layout: a
//         ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table): with a
//         row stride of 2 DH + 16 bytes two lanes of a group share a bank -- 8 LDS cycles per fragment read instead of 4);
// MODE 1: same staging, rows padded by 32 bytes: conflict-free for every head size (stride = 32 mod 64 bytes);
// MODE 2: tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction: no VGPR round trip, no
//         ds_write_b128 at ~13 cycles each), dense rows, 16-byte slots XOR-swizzled on the DMA source address and on the ds_read
//         side (key = row mod (vectors per row)): conflict-free; two buffers, tile t+1 in flight while tile t is multiplied,
//         counted-free wait (vmcnt(0): the only loads in the loop are the DMA pieces) + raw s_barrier per tile.  DH 64 / 128.
// MODE 3: MODE 2 without the V^T pre-pass: V tiles are staged ROW-major exactly like K tiles (same DMA, same swizzle) and the
//         second MFMA's operand is read with ds_read_b64_tr_b16, gfx950's LDS transpose read (two per fragment: 2 x 4 keys).
//         NS = ring depth: 2 (one tile in flight under the current one's math).  4 (three tiles ahead, counted vmcnt) is a
//         measured dead end kept as knob "attn_ring": one 1025-row sequence (272 blocks, nobody else to hide the DMA latency)
//         runs 17.1 us per launch either way, the batched passes lose occupancy (C3 NAR shape 451 -> 338 TF/s).
template <int DH, int QW, int MODE, int NS = 2, bool LSUM = false>
__global__ __launch_bounds__(256) void attn2_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
                                                    const int32_t* __restrict__ seq_off, const int32_t* __restrict__ text_len, int d,
                                                    int nhead, int causal, int64_t rp, int xcd_remap, float defer_exp2) {
  constexpr bool GLDS = MODE >= 2;
  constexpr bool TRV = MODE == 3;  // V staged ROW-major like K and read through ds_read_b64_tr_b16: no V^T pre-pass
  static_assert(MODE != 2 || DH == 64 || DH == 128, "LDS-DMA of the V^T image: 8 slots per V^T row, 8 or 16 per K row");
  constexpr int NV = DH / 8;           // 16-byte vectors per K row
  constexpr int PAD = GLDS ? 0 : (MODE == 1 ? 32 : 16);
  constexpr int KSTR = DH * 2 + PAD;   // bytes per K row in LDS
  constexpr int VSTR = 64 * 2 + PAD;   // bytes per V^T row
  constexpr int NLD = 64 * NV / 256;   // staged vectors per thread per tile (K and V^T each): DH / 32
  constexpr int KS = DH / 32, EB = DH / 16;
  constexpr int VSWZ = MODE == 2 ? 7 : 0;  // V^T image of MODE 2: slot ^= row & 7 (K rows: a2_key)
  static_assert(NS == 2 || (GLDS && NS == 4), "deeper rings exist for the LDS-DMA staging only");
  constexpr int NPW = NV / 4 + (TRV ? NV / 4 : DH / 32);  // DMA pieces per wave per tile
  constexpr int VBYTES = TRV ? 64 * KSTR : DH * VSTR;      // the V (or V^T) image of a tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (64 * KSTR + VBYTES)];

  // XCD-aware block order (block L runs on XCD L % 8, each with its own 4 MB L2): the query blocks of one (sequence, head) read
  // the same K / V -- give every XCD a CONTIGUOUS run of the (b, h, query-block) order so that they meet in one L2 instead of
  // pulling the tiles over the fabric 8 times (PMC, round 1: 5x the algorithmic bytes fetched)
  int qblk, h, b;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nb = gx * gy * (int)gridDim.z;
    int L = ((int)blockIdx.z * gy + (int)blockIdx.y) * gx + (int)blockIdx.x;
    if (xcd_remap) {
      const int q = nb / 8, r = nb % 8, xcd = L % 8, idx = L / 8;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    qblk = L % gx;
    h = (L / gx) % gy;
    b = L / (gx * gy);
  }
  const int q0 = qblk * 64 * QW;
  const int off = seq_off[b], len = seq_off[b + 1] - off;
  if (q0 >= len) return;
  const int S = text_len[b];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  int qrow[QW], klim[QW];
  bool qvalid[QW];
  int kfull = 0x7fffffff;  // keys below kfull are visible to every query of this lane (invalid rows do not constrain)
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    qrow[f] = q0 + (w * QW + f) * 16 + c;
    qvalid[f] = qrow[f] < len;
    klim[f] = !qvalid[f] ? 0 : (causal ? max(S, qrow[f] + 1) : len);  // keys j < klim are visible
    if (qvalid[f]) kfull = min(kfull, klim[f]);
  }
  const int kmax = causal ? max(S, min(q0 + 64 * QW, len)) : len;      // block-wide bound
  const int wq0 = __builtin_amdgcn_readfirstlane(q0 + w * QW * 16);     // first query row of this wave
  const bool wave_live = wq0 < len;                                      // wave-uniform
  const int kmax_w = causal ? max(S, min(wq0 + QW * 16, len)) : len;     // keys any row of this wave can see
  const int d3 = 3 * d;
  const bf16_t* base = qkv + (int64_t)off * d3 + h * DH;
  const bf16_t* vbase = vt + (int64_t)h * DH * rp + a2_vt_start(off, b);

  a2_bf16x8 qf[QW][KS];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    const bf16_t* qp = base + (int64_t)min(qrow[f], len - 1) * d3 + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[f][ks] = *reinterpret_cast<const a2_bf16x8*>(qp + ks * 32);
  }

  constexpr int BUF = 64 * KSTR + VBYTES;
  a2_u32x4 kreg[NLD], vreg[NLD];
  auto gload = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 256;
      const int kkey = idx / NV, kv = idx - kkey * NV;
      kreg[i] = *reinterpret_cast<const a2_u32x4*>(base + (int64_t)min(kt0 + kkey, len - 1) * d3 + d + kv * 8);
      const int e = idx >> 3, vec = idx & 7;  // V^T: DH rows x 8 vectors of 8 (permuted) keys
      vreg[i] = *reinterpret_cast<const a2_u32x4*>(vbase + (int64_t)e * rp + kt0 + vec * 8);
    }
  };
  auto lstore = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 256;
      const int kkey = idx / NV, kv = idx - kkey * NV;
      *reinterpret_cast<a2_u32x4*>(buf + kkey * KSTR + kv * 16) = kreg[i];
      const int e = idx >> 3, vec = idx & 7;
      *reinterpret_cast<a2_u32x4*>(buf + 64 * KSTR + e * VSTR + vec * 16) = vreg[i];
    }
  };
  // LDS-DMA staging (MODE 2): a tile is NV pieces of K (piece q: K-tile vectors 64 q .. 64 q + 63 in row-major order, i.e. rows
  // 64 q / NV ..) and DH / 8 pieces of V^T; wave w issues pieces w, w + 4, ...; lane l of a piece lands at LDS offset 16 l, so the
  // swizzle is applied to WHAT the lane fetches: slot s of row r holds vector s ^ (r & mask)
  const int wv = __builtin_amdgcn_readfirstlane(w);  // provably wave-uniform (LDS base of the DMA goes through M0)
  auto dma = [&](int kt0, unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      const int idx = (i * 4 + w) * 64 + lane;
      const int row = idx / NV, slot = idx % NV;
      const bf16_t* src = base + (__umul24((unsigned)min(kt0 + row, len - 1), (unsigned)d3) + (unsigned)(d + ((slot ^ (GLDS ? a2_key<NV>(row) : 0)) * 8)));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + (i * 4 + wv) * 1024), 16, 0, 0);
    }
    if constexpr (TRV) {  // V rows exactly like the K rows, 2 d bytes... d elements further along the qkv row
#pragma unroll
      for (int i = 0; i < NV / 4; ++i) {
        const int idx = (i * 4 + w) * 64 + lane;
        const int row = idx / NV, slot = idx % NV;
        const bf16_t* src = base + (__umul24((unsigned)min(kt0 + row, len - 1), (unsigned)d3) + (unsigned)(2 * d + ((slot ^ a2_key<NV>(row)) * 8)));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + 64 * KSTR + (i * 4 + wv) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < DH / 32; ++i) {
        const int idx = (i * 4 + w) * 64 + lane;
        const int e = idx >> 3, slot = idx & 7;
        const bf16_t* src = vbase + (int64_t)e * rp + kt0 + ((slot ^ (e & VSWZ)) * 8);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + 64 * KSTR + (i * 4 + wv) * 1024), 16, 0, 0);
      }
    }
  };

  const float sl2 = 1.4426950408889634f / sqrtf((float)DH);  // log2(e) / sqrt(dh)
  const float defer = defer_exp2 / sl2;                       // threshold on raw scores
  float m[QW], l[QW];  // m: running max of the RAW scores of the row (shared by its 4 lanes); l: this lane's part of the row sum
  // LSUM (round 3): the row sums ride on the MFMA pipe -- O^T += V^T P^T with one more "row block" of V^T that is all ones gives
  // sum_k P[k][q] in every row of ol[f]: 2 MFMAs per fragment and tile instead of 16 VALU adds (the kernel is bound by the VALU
  // issue port, 5.9 VALU instructions per MFMA, with the MFMA pipe 37 % busy).  The sum is then over the bf16-rounded P the
  // numerator uses too (numerator and denominator carry the same rounding).
  a2_f32x4 ol[QW];
  a2_f32x4 o[QW][EB];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    m[f] = A2_NEG;
    l[f] = 0.f;
    ol[f] = a2_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int eb = 0; eb < EB; ++eb) o[f][eb] = a2_f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // (Measured and rejected: issuing the scores of tile t+1 ahead of the softmax of tile t -- K ring one tile ahead of the V^T
  //  ring -- 365 -> 328 TF/s at the C3 NAR shape: the extra score registers cost more occupancy than the overlap returns.)
  const int ntile = (kmax + 63) >> 6;
  if constexpr (GLDS) {
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
      if (p < ntile) dma(p * 64, smem + p * BUF);
  } else {
    gload(0);
    lstore(smem);
    __syncthreads();
    if (ntile > 1) gload(64);
  }
  for (int t = 0; t < ntile; ++t) {
    const int kt0 = t * 64;
    const unsigned char* Ks = smem + (t & (NS - 1)) * BUF;
    const unsigned char* Vt = Ks + 64 * KSTR;
    if constexpr (GLDS) {
      // this wave's pieces of tile t landed (and, at t = 0, its Q fragments): at most the pieces of the younger tiles in flight
      if constexpr (NS == 2) {
        a2_wait_vm<0>();
      } else {
        const int younger = min(NS - 2, ntile - 1 - t);
        if (younger >= 2) a2_wait_vm<2 * NPW>();
        else if (younger == 1) a2_wait_vm<NPW>();
        else a2_wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();    // everyone's did; everyone finished reading tile t-1
      if (t + NS - 1 < ntile) dma(kt0 + 64 * (NS - 1), smem + ((t + NS - 1) & (NS - 1)) * BUF);  // into tile t-1's buffer
    }

    // a wave whose query rows all lie beyond the sequence (the ragged last block: N = 1025 leaves ONE row for a block of 64 / 128)
    // or whose rows see none of this tile's keys (causal) only stages tiles and meets the barriers
    if (wave_live && kt0 < kmax_w) {
    // ---- S^T = K Q^T: one K fragment read feeds the QW query fragments ---------------------------------
    a2_f32x4 s[QW][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int f = 0; f < QW; ++f) s[f][kb] = a2_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const a2_bf16x8 a = *reinterpret_cast<const a2_bf16x8*>(Ks + (kb * 16 + c) * KSTR + (((ks * 4 + g) ^ (GLDS ? a2_key<NV>(c) : 0)) * 16));
#pragma unroll
        for (int f = 0; f < QW; ++f) s[f][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[f][ks], s[f][kb], 0, 0, 0);
      }
    }
    // ---- online softmax (exp2 domain; lane (g, c) holds keys 16 kb + 4 g + r of query c) ------------------
    const bool all_visible = __all(kt0 + 64 <= kfull);  // wave-uniform: no key of this tile is masked for any query of the wave
    a2_bf16x8 pf[QW][2];
#pragma unroll
    for (int f = 0; f < QW; ++f) {
      if (!all_visible) {
        const int rel = klim[f] - kt0 - g * 4;  // key kt0 + 16 kb + 4 g + r is visible iff 16 kb + r < rel (compile-time left side)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kb * 16 + r >= rel) s[f][kb][r] = A2_NEG;
      }
      // this lane's maximum over its 16 keys of the row; m[f] is the row's reference, shared by its 4 lanes
      float mt = fmaxf(fmaxf(fmaxf(s[f][0][0], s[f][0][1]), fmaxf(s[f][0][2], s[f][0][3])),
                       fmaxf(fmaxf(s[f][1][0], s[f][1][1]), fmaxf(s[f][1][2], s[f][1][3])));
      mt = fmaxf(mt, fmaxf(fmaxf(fmaxf(s[f][2][0], s[f][2][1]), fmaxf(s[f][2][2], s[f][2][3])),
                           fmaxf(fmaxf(s[f][3][0], s[f][3][1]), fmaxf(s[f][3][2], s[f][3][3]))));
      // Deferred maximum: as long as no score of the wave exceeds its row's reference by more than `defer` (raw-score units;
      // 8 in the exp2 domain => P <= 256, nothing for bf16 P / fp32 sums) the reference stays: no cross-lane reduction, no
      // rescale.  Softmax does not depend on the reference (O and l carry the same factor); defer = 0 is the exact-maximum form.
      if (!__all(mt <= m[f] + defer)) {  // wave-uniform; always taken on the first tile (m = -1e30)
        mt = rows4_max(mt);
        const float mn = fmaxf(m[f], mt);
        const float alpha = __builtin_amdgcn_exp2f((m[f] - mn) * sl2);  // 1 exactly for the rows whose maximum did not grow
        m[f] = mn;
        if constexpr (LSUM) ol[f] *= alpha;
        else l[f] *= alpha;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) o[f][eb] *= alpha;
      }
      const a2_f32x2 nm2 = {-m[f] * sl2, -m[f] * sl2}, sl22 = {sl2, sl2};
      float ps[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        // pairs of scores: one v_pk_fma_f32 scales two (the VALU port, not the MFMA pipe, bounds this kernel: ~6 VALU per MFMA)
        const a2_f32x2 t0 = a2_f32x2{s[f][kb][0], s[f][kb][1]} * sl22 + nm2, t1 = a2_f32x2{s[f][kb][2], s[f][kb][3]} * sl22 + nm2;
        s[f][kb][0] = __builtin_amdgcn_exp2f(t0[0]);  // masked scores: exp2(-1e30 sl2 + ..) = 0
        s[f][kb][1] = __builtin_amdgcn_exp2f(t0[1]);
        s[f][kb][2] = __builtin_amdgcn_exp2f(t1[0]);
        s[f][kb][3] = __builtin_amdgcn_exp2f(t1[1]);
        if constexpr (!LSUM) ps[kb] = (s[f][kb][0] + s[f][kb][1]) + (s[f][kb][2] + s[f][kb][3]);
      }
      if constexpr (!LSUM) {
        const float rowsum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        l[f] += rowsum;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) pf[f][j][tt] = (__bf16)s[f][2 * j + (tt >> 2)][tt & 3];
    }
    // ---- O^T += V^T P^T -----------------------------------------------------------------------------------
    if constexpr (LSUM) {
      typedef short a2_s16x8l __attribute__((ext_vector_type(8)));
      const a2_bf16x8 ones = __builtin_bit_cast(a2_bf16x8, a2_s16x8l{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int f = 0; f < QW; ++f) ol[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[f][j], ol[f], 0, 0, 0);
    }
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        a2_bf16x8 a;
        if constexpr (TRV) {
          // ds_read_b64_tr_b16 (lane map probed on the hardware, tools/ubench_trread.hip): inside a 16-lane group, lane 4 a + b
          // receives element b of the 4 x u16 at the addresses supplied by lanes a, 4 + a, 8 + a, 12 + a.  Lane c supplies the
          // 8 bytes of key (k0 + c / 4), head columns 16 eb + 4 (c % 4) .. + 3; it then receives V[k0 + 0..3][16 eb + c]: four
          // consecutive keys of ITS row of V^T -- the 4-key groups the score MFMA left in this lane (keys 16 kb + 4 g + r).
          const int key = j * 32 + g * 4 + (c >> 2);  // second read: + 16 (same swizzle key: a2_key has period <= 16)
          const unsigned char* va = Vt + key * KSTR + (((eb * 2 + ((c & 3) >> 1)) ^ a2_key<NV>(key)) << 4) + ((c & 1) << 3);
          const a2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a2_s16x4*)(va));
          const a2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a2_s16x4*)(va + 16 * KSTR));
          typedef short a2_s16x8 __attribute__((ext_vector_type(8)));
          const a2_s16x8 both = a2_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          a = __builtin_bit_cast(a2_bf16x8, both);
        } else {
          a = *reinterpret_cast<const a2_bf16x8*>(Vt + (eb * 16 + c) * VSTR + (((j * 4 + g) ^ (c & VSWZ)) * 16));
        }
#pragma unroll
        for (int f = 0; f < QW; ++f) o[f][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[f][j], o[f][eb], 0, 0, 0);
      }
    }  // wave_live

    // ---- tile t+1 (in registers since the previous iteration) -> the other buffer; tile t+2 -> registers -------
    if constexpr (!GLDS) {
      if (t + 1 < ntile) {
        lstore(smem + ((t + 1) & 1) * BUF);  // last read during iteration t-1, i.e. before the barrier that ended it
        if (t + 2 < ntile) gload(kt0 + 128);
      }
      __syncthreads();  // tile t+1 visible; everyone is done reading tile t
    }
  }

#pragma unroll
  for (int f = 0; f < QW; ++f) {
    float lf = l[f];
    if constexpr (LSUM) lf = ol[f][0];  // every row of ol holds the whole row sum of query c already
    else lf = rows4_sum(lf);
    if (qvalid[f]) {
      const float inv = 1.0f / lf;
      bf16_t* op = out + (int64_t)(off + qrow[f]) * d + h * DH + g * 4;
#pragma unroll
      for (int eb = 0; eb < EB; ++eb) {
        a2_bf16x4 r4;
#pragma unroll
        for (int r = 0; r < 4; ++r) r4[r] = (__bf16)(o[f][eb][r] * inv);
        *reinterpret_cast<a2_bf16x4*>(op + eb * 16) = r4;
      }
    }
  }
}

// ---- V^T scratch of the PRE-PASS modes (attn_mode <= 2, A/B diagnostics only; the default attn_mode = 3 reads V through LDS
// transpose reads and never touches it): one per device, reserved lazily by the launcher, grown on demand (never during stream
// capture: attention is not part of the captured AR step), freed at process exit.  The pre-pass modes are therefore meant for ONE
// engine per device at a time; the pointer is handed out under the lock.
namespace {
struct VtScratch {
  void* p = nullptr;
  size_t bytes = 0;
};
std::mutex g_vt_mu;
VtScratch g_vt[16];
}  // namespace

int attn2_reserve(int64_t rows, int B, int d, void** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -3;
  const size_t need = (size_t)d * (size_t)(rows + 128 * (int64_t)(B + 1) + 64) * 2;
  std::lock_guard<std::mutex> lk(g_vt_mu);
  if (g_vt[dev].bytes >= need) {
    if (out) *out = g_vt[dev].p;
    return 0;
  }
  if (g_vt[dev].p) {
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    (void)hipFree(g_vt[dev].p);
    g_vt[dev] = VtScratch();
  }
  void* p = nullptr;
  if (hipMalloc(&p, need) != hipSuccess) return -3;
  if (hipMemset(p, 0, need) != hipSuccess) return -3;  // columns between sequences are read (and multiplied by P = 0): keep them finite
  g_vt[dev].p = p;
  g_vt[dev].bytes = need;
  if (out) *out = p;
  return 0;
}

int g_attn_v2 = 1;    // "attn_v2": 0 = round 1's kernel (attn_mfma.hip) for A/B
int g_attn_xcd = 1;   // "attn_xcd": XCD-aware block order (A/B)
int g_attn_mode = 3;  // "attn_mode": tile staging of attn2_kernel -- 0 registers + 16-byte row padding (round 2's first layout), 1 registers +
                      // 32-byte padding (conflict-free), 2 LDS-DMA + XOR swizzle where the head size allows (64 / 128), else 1;
                      // 3 = 2 with V staged row-major and read by ds_read_b64_tr_b16 (no V^T pre-pass)
int g_attn_lsum = 1;  // "attn_lsum": row sums of the softmax on the MFMA pipe (an all-ones row block of V^T) instead of VALU adds (mode 3)
int g_attn_ring = 0;  // "attn_ring": LDS ring depth of the LDS-DMA staging: 0 / 2 = two buffers, 4 = four (A/B knob, see attn2_kernel)
int g_attn_defer = 8; // "attn_defer": deferred-maximum threshold of attn2_kernel in exp2-domain units (0 = exact running maximum)
int g_attn_q128 = -1; // "attn_q128": 128-query blocks (each K / V^T fragment read from LDS feeds two MFMAs): 0 never, 1 always, -1 (default) =
                      // on the un-masked passes with at least 1024 such blocks (4 per CU) -- measured (tools/attn_bench.py): C3 NAR 461 -> 474
                      // TF/s, C5's dh 96 368 -> 438; a single sequence (144 blocks) and the causal prefill are faster with 64

// returns 0 = launched, 1 = not covered (caller uses attn_mfma.hip / the generic kernel), < 0 error.  `rows` = packed rows of qkv.
int launch_attention_mfma2(hipStream_t st, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len, int B,
                           int max_len, int64_t rows, int d, int nhead, int causal) {
  const int dh = d / nhead;
  if (!g_attn_v2 || d % 8 != 0 || !(dh == 32 || dh == 64 || dh == 96 || dh == 128)) return 1;
  if (rows <= 0) return 1;
  // measured (tools/attn_bench.py, MI355X): ahead of round 1's kernel on the un-masked NAR passes (C3 shape 305 -> 365 TF/s, C5's
  // dh 96 308 -> 319), behind it on the short causal prefill (120 vs 114) -- g_attn_v2 = 2 forces this kernel for every shape
  const int mode = (g_attn_mode == 2 && !(dh == 64 || dh == 128)) ? 1 : g_attn_mode;  // the V^T DMA image exists for 64 / 128 only
  const bool trv = mode == 3;  // V through LDS transpose reads: no V^T scratch, no pre-pass
  // ... without the pre-pass it is ahead there too (C3 prefill 128 -> 159 TF/s): the old kernel keeps the short / causal passes only
  // when the pre-pass form is selected (attn_mode <= 2)
  if (g_attn_v2 == 1 && !trv && (causal || max_len < 512)) return 1;
  bf16_t* vt = nullptr;
  if (!trv) {
    void* sp = nullptr;
    int r = attn2_reserve(rows, B, d, &sp);
    if (r) return r;
    vt = (bf16_t*)sp;
  }
  const int64_t rp = rows + 128 * (int64_t)(B + 1) + 64;  // >= every sequence's last padded column
  const int64_t rp8 = rp & ~(int64_t)7;                    // row pitch: a multiple of 8 elements (16-byte vectors)
  const dim3 pgrid((max_len + 63) / 64, nhead, B), block(256);
  const bool q128 = g_attn_q128 < 0 ? (!causal && (int64_t)B * nhead * ((max_len + 127) / 128) >= 1024) : g_attn_q128 != 0;
  const dim3 grid(q128 ? (max_len + 127) / 128 : (max_len + 63) / 64, nhead, B);
#define VLE_A2M(DH, QW, MODE, NS)                                                                                                  \
  hipLaunchKernelGGL((attn2_kernel<DH, QW, MODE, NS>), grid, block, 0, st, (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, seq_off, \
                     text_len, d, nhead, causal, rp8, g_attn_xcd, (float)g_attn_defer)
#define VLE_A2K(DH, QW)                                  \
  do {                                                   \
    if (mode == 3 && g_attn_lsum) {                      \
      hipLaunchKernelGGL((attn2_kernel<DH, QW, 3, 2, true>), grid, block, 0, st, (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, seq_off, \
                         text_len, d, nhead, causal, rp8, g_attn_xcd, (float)g_attn_defer);                                                   \
    } else if (mode == 3) {                              \
      VLE_A2M(DH, QW, 3, 2);                             \
    } else if (mode == 2) {                              \
      if constexpr (DH == 64 || (DH == 128 && QW == 1)) { \
        if (ring4) VLE_A2M(DH, QW, 2, 4);                \
        else VLE_A2M(DH, QW, 2, 2);                      \
      } else if constexpr (DH == 128) VLE_A2M(DH, QW, 2, 2); \
    } else if (mode == 1) VLE_A2M(DH, QW, 1, 2);         \
    else VLE_A2M(DH, QW, 0, 2);                          \
  } while (0)
#define VLE_A2(DH)                                                                                                              \
  do {                                                                                                                          \
    if (!trv) hipLaunchKernelGGL((vt_pack_kernel<DH>), pgrid, block, 0, st, (const bf16_t*)qkv, vt, seq_off, d, rp8);       \
    if (q128) VLE_A2K(DH, 2);                                                                                                   \
    else VLE_A2K(DH, 1);                                                                                                        \
  } while (0)
  const int64_t nblocks = (int64_t)grid.x * grid.y * grid.z;
  const bool ring4 = g_attn_ring == 4;
  (void)nblocks;
  switch (dh) {
    case 32: VLE_A2(32); break;
    case 64: VLE_A2(64); break;
    case 96: VLE_A2(96); break;
    case 128: VLE_A2(128); break;
    default: return 1;
  }
#undef VLE_A2
#undef VLE_A2M
#undef VLE_A2K
  return 0;
}

}  // namespace vle
