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
//   * up to 4 k-chunks (40 x 16-byte loads per lane at M = 64) are requested before the first MFMA.
#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));
typedef float gs_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gs_u32x4 __attribute__((ext_vector_type(4)));

template <int MF, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmSkinnyArgs a) {
  constexpr int G = 4;  // k-chunks (64 deep) requested per round
  __shared__ __attribute__((aligned(16))) float red[4][MF][64][4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int K = a.K, N = a.N, M = a.M;
  const int Kw = K >> 2;  // this wave's share of K (multiple of 64)
  const int nrow = min(n0 + fr, N - 1);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.w) + (int64_t)nrow * K + wave * Kw + fg * 8;
  const bf16_t* xp[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) xp[i] = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)min(i * 16 + fr, M - 1) * K + wave * Kw + fg * 8;

  // epilogue operands requested up front
  const int ncol = n0 + fg * 4;  // first of this lane's 4 output columns
  gs_f32x4 bias4 = gs_f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    if (ncol + 3 < N) bias4 = *reinterpret_cast<const gs_f32x4*>(a.bias + ncol);
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = ncol + r < N ? a.bias[ncol + r] : 0.f;
    }
  }

  gs_f32x4 acc[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) acc[i] = gs_f32x4{0.f, 0.f, 0.f, 0.f};

  const int chunks = Kw >> 6;
  for (int c0 = 0; c0 < chunks; c0 += G) {
    gs_u32x4 wv[G][2], xv[G][MF][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = min(c0 + g, chunks - 1);  // clamped: a short last round re-reads its final chunk, unused below
      wv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wp + c * 64));
      wv[g][1] = __builtin_nontemporal_load(reinterpret_cast<const gs_u32x4*>(wp + c * 64 + 32));
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        xv[g][i][0] = *reinterpret_cast<const gs_u32x4*>(xp[i] + c * 64);
        xv[g][i][1] = *reinterpret_cast<const gs_u32x4*>(xp[i] + c * 64 + 32);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (c0 + g < chunks) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < MF; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gs_bf16x8, wv[g][s]),
                                                             __builtin_bit_cast(gs_bf16x8, xv[g][i][s]), acc[i], 0, 0, 0);
      }
    }
  }

  // ---- combine the four K quarters through LDS; wave i finishes m-fragment i ------------------------
#pragma unroll
  for (int i = 0; i < MF; ++i) *reinterpret_cast<gs_f32x4*>(&red[wave][i][lane][0]) = acc[i];
  __syncthreads();
  if (wave >= MF) return;
  const int i = wave;
  gs_f32x4 v = *reinterpret_cast<const gs_f32x4*>(&red[0][i][lane][0]);
#pragma unroll
  for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const gs_f32x4*>(&red[w][i][lane][0]);
  v += bias4;
  // lane (fg, fr) holds C[m = 16 i + fr][n = n0 + 4 fg + r]
  const int m = i * 16 + fr;
  if (m >= M || ncol >= N) return;
  const bool vec = ncol + 3 < N && (N & 3) == 0;
  if constexpr (EPI == GS_EPI_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if constexpr (EPI == GS_EPI_STORE || EPI == GS_EPI_RELU) {
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
    if (vec) *reinterpret_cast<gs_f32x4*>(o) = *reinterpret_cast<const gs_f32x4*>(o) + v;
    else {
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
      const int64_t off = (((int64_t)m * a.nhead + h) * a.ctx_max + a.kv_len[m]) * a.dh + e;
      gs_bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
      *reinterpret_cast<gs_bf16x4*>(reinterpret_cast<bf16_t*>(which == 1 ? a.k_cache : a.v_cache) + off) = o4;
    }
  }
}

template <int MF>
static int gs_launch(hipStream_t st, const GemmSkinnyArgs& a) {
  const dim3 grid((a.N + 15) / 16), block(256);
  switch (a.epi) {
    case GS_EPI_STORE: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_STORE>), grid, block, 0, st, a); break;
    case GS_EPI_RELU: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_RELU>), grid, block, 0, st, a); break;
    case GS_EPI_RESID: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_RESID>), grid, block, 0, st, a); break;
    case GS_EPI_F32: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_F32>), grid, block, 0, st, a); break;
    case GS_EPI_QKV: hipLaunchKernelGGL((gemm_skinny_kernel<MF, GS_EPI_QKV>), grid, block, 0, st, a); break;
    default: return -1;
  }
  return 0;
}

bool gemm_skinny_supports(int M, int N, int K, int epi, int dh) {
  if (M < 1 || M > 64 || N < 1 || K < 256 || K % 256 != 0) return false;
  if (epi == GS_EPI_QKV && (N % 12 != 0 || dh % 4 != 0)) return false;
  return true;
}

// returns 0 = launched, 1 = shape not covered
int launch_gemm_skinny(hipStream_t st, const GemmSkinnyArgs& a) {
  if (!gemm_skinny_supports(a.M, a.N, a.K, a.epi, a.dh)) return 1;
  if (a.M <= 16) return gs_launch<1>(st, a);
  if (a.M <= 32) return gs_launch<2>(st, a);
  if (a.M <= 48) return gs_launch<3>(st, a);
  return gs_launch<4>(st, a);
}

}  // namespace vle
