// fp8 (OCP e4m3fn) GEMM of the prefill and the 7 NAR passes on CDNA4's block-scaled fp8 MFMA -- BASELINE.json configs[4]
// ("fp8 weights (CDNA4 fp8 MFMA)"), engine mode FP8:
//   out = epi( (A8[M x K] @ W8[N x K]^T) * sa[m] * sw[n] + bias[n] )
//   A8: activations quantised per row (misc.hip quantize_rows_fp8_kernel: e4m3fn codes + one power-of-two scale per row),
//   W8: the FP8W weight format (common.h: e4m3fn codes + one power-of-two scale per row) -- the SAME codes the AR step streams.
//   reference ops: in-proj / out-proj `linear` (valle/modules/activation.py:414-421), FFN linear1 / linear2
//   (valle/modules/transformer.py:332-334), nar_predict_layers (valle/models/valle.py:1128).
//
// v_mfma_scale_f32_16x16x128_f8f6f4 is the only fp8 MFMA that runs above the bf16 rate on gfx950 (MX, K = 128: 2x bf16;
// the un-scaled 16x16x32 fp8 form runs AT the bf16 rate).  Its block scales (E8M0 per 32 k) are all set to 2^0 here: both
// operands carry ONE scale per row, applied exactly in the fp32 epilogue (powers of two).
//
// Pipeline = gemm_glds.hip byte for byte: a k-step is a 128-BYTE slab of every tile row (there 64 bf16, here 128 fp8), tiles
// go global -> LDS by LDS-DMA into a 3-4 stage ring with counted vmcnt + one raw s_barrier per k-step, 16-byte slots
// XOR-swizzled on the DMA source and on the ds_read_b128 side.  A lane's two 16-byte reads of a k-step (slots fg and 4 + fg
// of its row) form the 32-byte MFMA operand; A and W fragments are read with the same slot assignment, so whatever k order the
// instruction assigns to a lane's 32 bytes, both operands agree -- a dot product is invariant under a common k permutation
// (with unit block scales).  Per k-step and fragment pair: ONE K = 128 MFMA instead of two K = 32 bf16 MFMAs over half the K,
// i.e. half the MFMA time and half the LDS bytes per flop.
#include "common.h"
#include "kernels.h"

namespace vle {

typedef int g8_i32x8 __attribute__((ext_vector_type(8)));
typedef int g8_i32x4 __attribute__((ext_vector_type(4)));
typedef float g8_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g8_bf16x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ inline void g8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_fp8_kernel(const unsigned char* __restrict__ A, const float* __restrict__ sa,
                                                           const unsigned char* __restrict__ W, const float* __restrict__ sw,
                                                           const float* __restrict__ bias, void* __restrict__ out_,
                                                           float* __restrict__ resid, int64_t M, int N, int K, int g_glds_epi_dev) {
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int STAGES = (4 * STAGE_BYTES <= 144 * 1024) ? 4 : 3;
  constexpr int D = STAGES - 1;
  constexpr int NIA = BM / (8 * NW), NIB = BN / (8 * NW);
  constexpr int NI = NIA + NIB;
  constexpr int WMW = NW / 2;
  constexpr int WM = BM / WMW, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int nbx = gridDim.x, nby = gridDim.y;
  const int nblk = nbx * nby;
  int bid = blockIdx.y * nbx + blockIdx.x;
  {  // XCD-aware order over the full-height tiles, tail-row tiles last (as gemm_glds.hip)
    const int nfull = (int)(M / BM) * nbx;
    const int nr = bid < nfull ? nfull : nblk;
    if (bid < nfull || nfull == 0) {
      const int q = nr / 8, r = nr % 8, xcd = bid % 8, idx = bid / 8;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
  }
  const int64_t m0 = (int64_t)(bid / nbx) * BM;
  const int n0 = (bid % nbx) * BN;

  const int prow = lane >> 3;
  const unsigned char* srcA[NIA];
  const unsigned char* srcB[NIB];
#pragma unroll
  for (int p = 0; p < NIA; ++p) {
    int64_t gm = m0 + (p * NW + wave) * 8 + prow;
    gm = gm < M ? gm : M - 1;
    srcA[p] = A + gm * K + ((lane & 7) ^ (((p * NW + wave) * 8 + prow) & 7)) * 16;
  }
#pragma unroll
  for (int p = 0; p < NIB; ++p) {
    int gn = n0 + (p * NW + wave) * 8 + prow;
    gn = gn < N ? gn : N - 1;
    srcB[p] = W + (int64_t)gn * K + ((lane & 7) ^ (((p * NW + wave) * 8 + prow) & 7)) * 16;
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
  // epilogue operands first (ahead of the DMA queue): bias and weight scale of the lane's 4 consecutive columns per
  // n-fragment, activation scale of its row per m-fragment
  g8_f32x4 bias4[FN], sw4[FN];
  float sa1[FM];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = min(n0 + wn0 + j * 16 + fg * 4, N - 4);
    bias4[j] = bias != nullptr ? *reinterpret_cast<const g8_f32x4*>(bias + n) : g8_f32x4{0.f, 0.f, 0.f, 0.f};
    sw4[j] = *reinterpret_cast<const g8_f32x4*>(sw + n);
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int64_t m = m0 + wm0 + i * 16 + fr;
    sa1[i] = sa[m < M ? m : M - 1];
  }

  g8_f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
  int nlive = (int)((M - (m0 + wm0) + 15) / 16);
  nlive = nlive < 0 ? 0 : (nlive > FM ? FM : nlive);

  const int KT = K / 128;
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < KT) issue(s);

  constexpr int UNIT = 0x7f7f7f7f;  // E8M0 2^0 for every 32-k block of both operands
  for (int kt = 0; kt < KT; ++kt) {
    const int younger = KT - 1 - kt;
    if (younger >= D - 1) g8_wait_vm<(D - 1) * NI>();
    else if (D >= 3 && younger == 1) g8_wait_vm<NI>();
    else g8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + D < KT) issue(kt + D);

    const unsigned char* As = smem + (kt % STAGES) * STAGE_BYTES;
    const unsigned char* Bs = As + BM * 128;
    auto frag = [&](const unsigned char* base, int row) -> g8_i32x8 {
      const g8_i32x4 lo = *reinterpret_cast<const g8_i32x4*>(base + row * 128 + ((fg ^ (row & 7)) << 4));
      const g8_i32x4 hi = *reinterpret_cast<const g8_i32x4*>(base + row * 128 + (((4 + fg) ^ (row & 7)) << 4));
      return g8_i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    };
    if (nlive == FM) {
      g8_i32x8 bfr[FN], af[FM];
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = frag(Bs, wn0 + j * 16 + fr);
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = frag(As, wm0 + i * 16 + fr);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)  // W fragment as the A operand: C^T, a lane owns 4 consecutive output columns
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, UNIT, 0, UNIT);
    } else if (nlive > 0) {  // tail tile (rows beyond M): only fragment rows that exist
      g8_i32x8 bfr[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = frag(Bs, wn0 + j * 16 + fr);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (i < nlive) {
          const g8_i32x8 a1 = frag(As, wm0 + i * 16 + fr);
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], a1, acc[i][j], 0, 0, 0, UNIT, 0, UNIT);
        }
      }
    }
  }

  // Tiles inside N: epilogue staged through LDS as a swizzled row-major image and written as whole rows -- gemm_glds.hip's
  // epilogue with the two scale factors applied on the way in (see there for the layout and why).
  if (n0 + BN <= N && g_glds_epi_dev) {
    constexpr bool F32OUT = EPI == EPI_RESID || EPI == EPI_F32;
    constexpr int ES = F32OUT ? 4 : 2, RB = BN * ES, CPR = RB / 16, RPI = 64 / CPR, IT = BM / NW / RPI;
    static_assert(BM * RB <= STAGES * STAGE_BYTES && (BM / NW) % RPI == 0, "epilogue image must fit the LDS ring");
    __syncthreads();
    unsigned char* const E = smem;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wm0 + i * 16 + fr, col = wn0 + j * 16 + fg * 4;
        g8_f32x4 v = acc[i][j] * (sw4[j] * sa1[i]) + bias4[j];  // the two scales are powers of two: exact
        if constexpr (EPI == EPI_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr (F32OUT) {
          *reinterpret_cast<g8_f32x4*>(E + row * RB + (((col >> 2) ^ (fr & 7)) << 4)) = v;
        } else {
          g8_bf16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = (__bf16)v[r];
          *reinterpret_cast<g8_bf16x4*>(E + row * RB + (((col >> 3) ^ (fr & 7)) << 4) + ((((col >> 2) & 1) ^ (fr >> 3)) << 3)) = o4;
        }
      }
    const int l = lane % CPR, rsub = lane / CPR;
    g8_f32x4 old[EPI == EPI_RESID ? IT : 1];
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int64_t m = m0 + wave * (BM / NW) + it * RPI + rsub;
        old[it] = *reinterpret_cast<const g8_f32x4*>(resid + (m < M ? m : M - 1) * N + n0 + l * 4);
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = wave * (BM / NW) + it * RPI + rsub;
      const int64_t m = m0 + r;
      g8_i32x4 v = *reinterpret_cast<const g8_i32x4*>(E + r * RB + ((l ^ (r & 7)) << 4));
      if constexpr (!F32OUT) {
        if ((r >> 3) & 1) v = g8_i32x4{v[2], v[3], v[0], v[1]};
      }
      if (m < M) {
        if constexpr (EPI == EPI_RESID) {
          const g8_f32x4 f = g8_f32x4{__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
          *reinterpret_cast<g8_f32x4*>(resid + m * N + n0 + l * 4) = old[it] + f;
        } else if constexpr (EPI == EPI_F32) {
          *reinterpret_cast<g8_i32x4*>(reinterpret_cast<float*>(out_) + m * N + n0 + l * 4) = v;
        } else {
          *reinterpret_cast<g8_i32x4*>(reinterpret_cast<bf16_t*>(out_) + m * N + n0 + l * 8) = v;
        }
      }
    }
    return;
  }
  const bool full = m0 + BM <= M && n0 + BN <= N;
  if constexpr (EPI == EPI_RESID) {
    g8_f32x4 old[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int64_t m = m0 + wm0 + i * 16 + fr;
        const int n = n0 + wn0 + j * 16 + fg * 4;
        old[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
        if (full || (m < M && n < N)) old[i][j] = *reinterpret_cast<const g8_f32x4*>(resid + m * N + n);
      }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = old[i][j] + (acc[i][j] * (sw4[j] * sa1[i]) + bias4[j]);
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acc[i][j] = acc[i][j] * (sw4[j] * sa1[i]) + bias4[j];  // the two scales are powers of two: exact
        if constexpr (EPI == EPI_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
        }
      }
  }
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
          *reinterpret_cast<g8_f32x4*>(resid + m * N + n) = acc[i][j];
        } else if constexpr (EPI == EPI_F32) {
          *reinterpret_cast<g8_f32x4*>(reinterpret_cast<float*>(out_) + m * N + n) = acc[i][j];
        } else {
          g8_bf16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = (__bf16)acc[i][j][r];
          *reinterpret_cast<g8_bf16x4*>(reinterpret_cast<bf16_t*>(out_) + m * N + n) = o4;
        }
      }
    }
}

template <int BM, int BN, int NW>
static int g8_launch(hipStream_t st, const unsigned char* A, const float* sa, const unsigned char* W, const float* sw, const float* bias,
                     void* out, float* resid, int64_t M, int N, int K, int epi) {
  const dim3 grid((N + BN - 1) / BN, (unsigned)((M + BM - 1) / BM)), block(NW * 64);
#define VLE_G8(E) hipLaunchKernelGGL((gemm_fp8_kernel<BM, BN, E, NW>), grid, block, 0, st, A, sa, W, sw, bias, out, resid, M, N, K, g_glds_epi)
  switch (epi) {
    case EPI_STORE: VLE_G8(EPI_STORE); break;
    case EPI_RELU: VLE_G8(EPI_RELU); break;
    case EPI_RESID: VLE_G8(EPI_RESID); break;
    case EPI_F32: VLE_G8(EPI_F32); break;
    default: return -1;
  }
#undef VLE_G8
  return 0;
}

// returns 0 = launched, 1 = shape not covered (the caller then runs the bf16 kernels on bf16(W'))
int launch_gemm_fp8(hipStream_t st, const void* A8, const float* a_scale, const void* W8, const float* w_scale, const float* bias,
                    void* out, float* resid, int64_t M, int N, int K, int epi) {
  if (K % 128 != 0 || K < 128 || M < 1 || N < 4 || N % 4 != 0 || !a_scale || !w_scale) return 1;
  const unsigned char* a = (const unsigned char*)A8;
  const unsigned char* w = (const unsigned char*)W8;
  const int64_t t256 = (M / 256) * ((N + 127) / 128);
  const int64_t full128 = (M / 128) * ((N + 127) / 128) + ((M % 128) ? ((N + 127) / 128) : 0);
  if (t256 >= 256) return g8_launch<256, 128, 8>(st, a, a_scale, w, w_scale, bias, out, resid, M, N, K, epi);
  if (full128 >= 160) return g8_launch<128, 128, 8>(st, a, a_scale, w, w_scale, bias, out, resid, M, N, K, epi);
  return g8_launch<128, 64, 8>(st, a, a_scale, w, w_scale, bias, out, resid, M, N, K, epi);
}

}  // namespace vle
