// Flash-style MFMA attention for the prefill and the 7 NAR passes (bf16 mode, head size 32/64/96/128).
//   reference: F.multi_head_attention_forward as called from MultiheadAttention.forward
//   (valle/modules/activation.py:408-427): softmax(Q K^T / sqrt(dh) + mask) V per head, heads =
//   contiguous dh slices; AR mask = prefix-LM (valle/models/valle.py:1019-1033), NAR: none.
//   The reference materialises the (h, T, T) score matrix; here it never leaves registers.
//
// gfx950 mapping (v_mfma_f32_16x16x32_bf16, wave64).  A block = 4 waves = 64 query rows of one
// (utterance, head); a wave owns 16 queries.  Both products are computed TRANSPOSED so that the
// softmax probabilities come out of the first MFMA already in the operand layout of the second:
//   S^T[key][q] = K Q^T   A = K tile rows (LDS, ds_read_b128), B = Q rows (registers, loaded once)
//                         C layout: lane (g = lane>>4, c = lane&15) holds keys 16*kb + 4*g + r of
//                         query c  -> the softmax of a query lives in 4 lanes (c, c+16, c+32, c+48):
//                         in-lane max/sum over 16 values + two cross-lane steps per 64-key tile;
//   O^T[e][q]   = V^T P^T A = V^T rows (LDS), B = P^T: the 8 bf16 a lane needs for key-step j are
//                         exactly its own p[2j][0..3], p[2j+1][0..3] -- no shuffle, no LDS round trip.
//                         The contraction index of an MFMA may be permuted freely as long as A and B
//                         agree, so V^T is staged in LDS with its keys permuted to that order
//                         (pos(key) below); that scatter replaces a transpose read.
// K/V tiles (64 keys) are staged through LDS once per block and shared by the 4 waves; the next
// tile's global loads are in flight while the current one is multiplied.  Softmax runs in the
// exp2 domain (scores pre-multiplied by log2 e / sqrt(dh)); P is rounded to bf16 for the second
// MFMA (fp32 accumulate), the running max / sum / output stay fp32.
#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 am_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 am_bf16x4 __attribute__((ext_vector_type(4)));
typedef float am_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int am_u32x4 __attribute__((ext_vector_type(4)));

constexpr float AM_NEG = -1e30f;

__device__ inline int am_vpos(int key) {  // slot of key (0..63) inside a V^T row of the LDS tile
  return ((key >> 5) << 5) + (((key & 15) >> 2) << 3) + (((key >> 4) & 1) << 2) + (key & 3);
}

// QW = 16-query fragments per wave: 1 (64 queries per block) or 2 (128: every K / V fragment read from LDS feeds two
// MFMAs and the K/V tiles are re-read from L2 half as often -- for the batched passes, where there are blocks to spare).
// PAD = bytes of padding per LDS row: 16 (round 1) makes two lanes of a ds_read_b128 lane group share a bank, 32 is conflict-free
// for every head size (see attn_mfma2.hip, MODE 0 / 1).
template <int DH, int QW, int PAD>
__global__ __launch_bounds__(256) void attn_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                        const int32_t* __restrict__ seq_off,
                                                        const int32_t* __restrict__ text_len, int d, int nhead, int causal) {
  constexpr int NV = DH / 8;           // 16-byte vectors per K/V row
  constexpr int KSTR = DH * 2 + PAD;   // bytes per K row in LDS
  constexpr int VSTR = 64 * 2 + PAD;   // bytes per V^T row
  constexpr int NLD = 64 * NV / 256;   // staged vectors per thread per tile (K and V each)
  constexpr int KS = DH / 32, EB = DH / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * KSTR + DH * VSTR];
  unsigned char* const Ks = smem;
  unsigned char* const Vt = smem + 64 * KSTR;

  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64 * QW;
  const int off = seq_off[b], len = seq_off[b + 1] - off;
  if (q0 >= len) return;
  const int S = text_len[b];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  int qrow[QW], klim[QW];
  bool qvalid[QW];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    qrow[f] = q0 + (w * QW + f) * 16 + c;
    qvalid[f] = qrow[f] < len;
    klim[f] = !qvalid[f] ? 0 : (causal ? max(S, qrow[f] + 1) : len);  // keys j < klim are visible
  }
  const int kmax = causal ? max(S, min(q0 + 64 * QW, len)) : len;      // block-wide bound
  const int d3 = 3 * d;
  const bf16_t* base = qkv + (int64_t)off * d3 + h * DH;

  // Q rows of this wave as the B operand (kept for the whole kernel)
  am_bf16x8 qf[QW][KS];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    const bf16_t* qp = base + (int64_t)min(qrow[f], len - 1) * d3 + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[f][ks] = *reinterpret_cast<const am_bf16x8*>(qp + ks * 32);
  }

  am_u32x4 kreg[NLD], vreg[NLD];
// global -> registers for the K/V tile starting at key kt0 (rows clamped into the sequence)
#define AM_GLOAD(kt0_)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                      \
    const int idx = tid + i * 256;                                                                       \
    const int kkey = idx / NV, kv = idx - kkey * NV;                                                     \
    kreg[i] = *reinterpret_cast<const am_u32x4*>(base + (int64_t)min((kt0_) + kkey, len - 1) * d3 + d + kv * 8); \
    const int vkey = idx & 63, vv = idx >> 6; /* key-fastest: the V^T scatter then hits distinct banks */ \
    vreg[i] = *reinterpret_cast<const am_u32x4*>(base + (int64_t)min((kt0_) + vkey, len - 1) * d3 + 2 * d + vv * 8); \
  }
// registers -> LDS: K rows as they are, V transposed with permuted key slots
#define AM_LSTORE()                                                                                      \
  _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                      \
    const int idx = tid + i * 256;                                                                       \
    const int kkey = idx / NV, kv = idx - kkey * NV;                                                     \
    *reinterpret_cast<am_u32x4*>(Ks + kkey * KSTR + kv * 16) = kreg[i];                                  \
    const int vkey = idx & 63, vv = idx >> 6;                                                            \
    unsigned char* dst = Vt + (vv * 8) * VSTR + am_vpos(vkey) * 2;                                       \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                      \
      const uint32_t wv = vreg[i][t];                                                                    \
      *reinterpret_cast<uint16_t*>(dst + (2 * t) * VSTR) = (uint16_t)(wv & 0xffffu);                     \
      *reinterpret_cast<uint16_t*>(dst + (2 * t + 1) * VSTR) = (uint16_t)(wv >> 16);                     \
    }                                                                                                    \
  }

  const float sl2 = 1.4426950408889634f / sqrtf((float)DH);  // log2(e) / sqrt(dh)
  float m[QW], l[QW];
  am_f32x4 o[QW][EB];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    m[f] = AM_NEG;
    l[f] = 0.f;
#pragma unroll
    for (int eb = 0; eb < EB; ++eb) o[f][eb] = am_f32x4{0.f, 0.f, 0.f, 0.f};
  }

  AM_GLOAD(0)
  AM_LSTORE()
  __syncthreads();
  for (int kt0 = 0; kt0 < kmax; kt0 += 64) {
    const bool has_next = kt0 + 64 < kmax;
    if (has_next) { AM_GLOAD(kt0 + 64) }

    // ---- S^T = K Q^T: one K fragment read feeds the QW query fragments ---------------------------------
    am_f32x4 s[QW][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int f = 0; f < QW; ++f) s[f][kb] = am_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const am_bf16x8 a = *reinterpret_cast<const am_bf16x8*>(Ks + (kb * 16 + c) * KSTR + (ks * 4 + g) * 16);
#pragma unroll
        for (int f = 0; f < QW; ++f) s[f][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[f][ks], s[f][kb], 0, 0, 0);
      }
    }
    // ---- online softmax of query c over this tile's keys (exp2 domain) -----------------------------------
    am_bf16x8 pf[QW][2];
#pragma unroll
    for (int f = 0; f < QW; ++f) {
      float mt = AM_NEG;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt0 + kb * 16 + g * 4 + r;
          const float v = key < klim[f] ? s[f][kb][r] * sl2 : AM_NEG;
          s[f][kb][r] = v;
          mt = fmaxf(mt, v);
        }
      mt = rows4_max(mt);
      const float mn = fmaxf(m[f], mt);
      const float alpha = __builtin_amdgcn_exp2f(m[f] - mn);
      m[f] = mn;
      float rowsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt0 + kb * 16 + g * 4 + r;
          const float p = key < klim[f] ? __builtin_amdgcn_exp2f(s[f][kb][r] - mn) : 0.f;
          s[f][kb][r] = p;
          rowsum += p;
        }
      l[f] = fmaf(l[f], alpha, rowsum);  // per-lane partial of the row sum (the 4 lanes of a query share m)
#pragma unroll
      for (int eb = 0; eb < EB; ++eb) o[f][eb] *= alpha;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) pf[f][j][t] = (__bf16)s[f][2 * j + (t >> 2)][t & 3];
    }
    // ---- O^T += V^T P^T -----------------------------------------------------------------------------------
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const am_bf16x8 a = *reinterpret_cast<const am_bf16x8*>(Vt + (eb * 16 + c) * VSTR + (j * 32 + g * 8) * 2);
#pragma unroll
        for (int f = 0; f < QW; ++f) o[f][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[f][j], o[f][eb], 0, 0, 0);
      }

    __syncthreads();  // every wave is done reading this tile
    if (has_next) { AM_LSTORE() }
    __syncthreads();
  }

#pragma unroll
  for (int f = 0; f < QW; ++f) {
    float lf = l[f];
    lf = rows4_sum(lf);
    if (qvalid[f]) {
      const float inv = 1.0f / lf;
      bf16_t* op = out + (int64_t)(off + qrow[f]) * d + h * DH + g * 4;
#pragma unroll
      for (int eb = 0; eb < EB; ++eb) {
        am_bf16x4 r4;
#pragma unroll
        for (int r = 0; r < 4; ++r) r4[r] = (__bf16)(o[f][eb][r] * inv);
        *reinterpret_cast<am_bf16x4*>(op + eb * 16) = r4;
      }
    }
  }
}

int g_attn_qw = 0;  // "attn_qw": 0 / 1 = 64-query blocks, 2 = 128-query blocks (measured neutral at C3: NAR 221.0 vs 222.8 ms --
                    // the kernel is bound by its softmax VALU work and barriers, not by the K/V re-reads the PMC pass shows)

// returns 0 = launched, 1 = head size not covered (caller uses the generic kernel)
int launch_attention_mfma(hipStream_t st, const void* qkv, void* out, const int32_t* seq_off, const int32_t* text_len, int B,
                          int max_len, int d, int nhead, int causal) {
  const int dh = d / nhead;
  if (d % 8 != 0) return 1;
  const int qw = g_attn_qw == 2 ? 2 : 1;
  const dim3 grid((max_len + 64 * qw - 1) / (64 * qw), nhead, B), block(256);
#define VLE_AMP(DH, QW, PAD)                                                                                                  \
  hipLaunchKernelGGL((attn_mfma_kernel<DH, QW, PAD>), grid, block, 0, st, (const bf16_t*)qkv, (bf16_t*)out, seq_off, text_len, d, \
                     nhead, causal)
#define VLE_AM(DH)                            \
  do {                                        \
    if (qw == 2) {                            \
      if (g_attn_mode == 0) VLE_AMP(DH, 2, 16); \
      else VLE_AMP(DH, 2, 32);                \
    } else {                                  \
      if (g_attn_mode == 0) VLE_AMP(DH, 1, 16); \
      else VLE_AMP(DH, 1, 32);                \
    }                                         \
  } while (0)
  switch (dh) {
    case 32: VLE_AM(32); break;
    case 64: VLE_AM(64); break;
    case 96: VLE_AM(96); break;
    case 128: VLE_AM(128); break;
    default: return 1;
  }
#undef VLE_AM
#undef VLE_AMP
  return 0;
}
#undef AM_GLOAD
#undef AM_LSTORE

}  // namespace vle
