// Dense GEMM for the prefill and the 7 NAR passes:  out = epi(A[M x K] @ W[N x K]^T + bias)
//   reference ops: in-proj / out-proj `linear` inside F.multi_head_attention_forward
//   (valle/modules/activation.py:414-421), FFN linear1/linear2 (valle/modules/transformer.py:332-334),
//   nar_predict_layers (valle/models/valle.py:1128).
//
// gfx950 design: both operands are K-contiguous ([rows][K]), which is exactly the MFMA A/B fragment
// shape -- no transposes.  256 threads = 4 wave64 in a 2x2 grid; tile BM x BN (128x128 or 64x64 so
// N = d GEMMs still fill 256 CUs); K advances 128 bytes per stage (64 bf16 / 32 fp32) through a
// double-buffered, XOR-swizzled LDS image read with ds_read_b128:
//   bf16 : v_mfma_f32_16x16x32_bf16      (one 16-byte fragment = 8 bf16 along K per lane)
//   fp32 : v_mfma_f32_16x16x4_f32 x 4    (exact fp32 FMA chain -> token-id-exact mode)
// A lane group g = lane>>4 supplies the same K slice for A and for B, so the K permutation inside a
// fragment is irrelevant to the result; only row = lane&15 and the C/D map (col = lane&15,
// row = 4*(lane>>4) + reg) matter.
// Epilogues fuse bias, ReLU and the residual add (fp32 residual stream, in place).
#include "common.h"
#include "kernels.h"

namespace vle {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ inline void mma_step(const uint4& a, const uint4& b, f32x4_t& c);
template <>
__device__ inline void mma_step<bf16_t>(const uint4& a, const uint4& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ inline void mma_step<float>(const uint4& a, const uint4& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

// One LDS stage: rows x 8 sixteen-byte vectors (128 B of K per row); vector c of row r lives at
// slot c ^ (r & 7) so that the 16 rows a ds_read_b128 lane group touches hit distinct bank groups.
template <typename T, int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                   const float* __restrict__ bias, void* __restrict__ out_,
                                                   float* __restrict__ resid, int64_t M, int N, int K, int64_t lda) {
  constexpr int KE = 128 / sizeof(T);  // K elements per stage
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int AV = BM * 8 / 256, BV = BN * 8 / 256;  // staged vectors per thread
  __shared__ uint4 As[2][BM * 8];
  __shared__ uint4 Bs[2][BN * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // XCD-aware tile order: consecutive block ids land on different XCDs (block b -> XCD b % 8), so
  // give each XCD a contiguous run of tiles that share W panels in its private L2.
  const int nbx = gridDim.x, nby = gridDim.y;
  const int nblk = nbx * nby;
  int bid = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int64_t m0 = (int64_t)(bid / nbx) * BM;
  const int n0 = (bid % nbx) * BN;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int KT = K / KE;
  uint4 ra[AV], rb[BV];
  auto gload = [&](int kt) {
#pragma unroll
    for (int v = 0; v < AV; ++v) {
      const int vi = tid + v * 256, row = vi >> 3, c = vi & 7;
      int64_t gm = m0 + row;
      gm = gm < M ? gm : M - 1;
      ra[v] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(A + gm * lda) + (int64_t)kt * 128 + c * 16);
    }
#pragma unroll
    for (int v = 0; v < BV; ++v) {
      const int vi = tid + v * 256, row = vi >> 3, c = vi & 7;
      int gn = n0 + row;
      gn = gn < N ? gn : N - 1;
      rb[v] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(W + (int64_t)gn * K) + (int64_t)kt * 128 + c * 16);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int v = 0; v < AV; ++v) {
      const int vi = tid + v * 256, row = vi >> 3, c = vi & 7;
      As[buf][row * 8 + (c ^ (row & 7))] = ra[v];
    }
#pragma unroll
    for (int v = 0; v < BV; ++v) {
      const int vi = tid + v * 256, row = vi >> 3, c = vi & 7;
      Bs[buf][row * 8 + (c ^ (row & 7))] = rb[v];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 af[FM], bfr[FN];
      const int c = ks * 4 + fg;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm0 + i * 16 + fr;
        af[i] = As[cur][row * 8 + (c ^ (row & 7))];
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn0 + j * 16 + fr;
        bfr[j] = Bs[cur][row * 8 + (c ^ (row & 7))];
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma_step<T>(af[i], bfr[j], acc[i][j]);
    }
    if (kt + 1 < KT) lstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: C/D map col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn0 + j * 16 + fr;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm0 + i * 16 + fg * 4 + r;
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if constexpr (EPI == EPI_RELU) v = fmaxf(v, 0.f);
        if constexpr (EPI == EPI_RESID) {
          resid[m * N + n] += v;
        } else if constexpr (EPI == EPI_F32) {
          reinterpret_cast<float*>(out_)[m * N + n] = v;
        } else {
          store_elem<T>(reinterpret_cast<T*>(out_) + m * N + n, v);
        }
      }
    }
  }
}

template <typename T, int BM, int BN>
static int gemm_dispatch_epi(hipStream_t st, const T* A, const T* W, const float* bias, void* out, float* resid, int64_t M,
                             int N, int K, int epi, int64_t lda) {
  const dim3 grid((N + BN - 1) / BN, (unsigned)((M + BM - 1) / BM)), block(256);
  switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI_STORE>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, lda); break;
    case EPI_RELU: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI_RELU>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, lda); break;
    case EPI_RESID: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI_RESID>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, lda); break;
    case EPI_F32: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI_F32>), grid, block, 0, st, A, W, bias, out, resid, M, N, K, lda); break;
    default: return -1;
  }
  return 0;
}

template <typename T>
static int gemm_dispatch(hipStream_t st, const void* A, const void* W, const float* bias, void* out, float* resid, int64_t M,
                         int N, int K, int epi, int64_t lda = 0) {
  if (K % (int)(128 / sizeof(T)) != 0) return -1;
  if (lda == 0) lda = K;
  // 128x128 tiles only when they alone fill the chip; otherwise 64x64 (4x the blocks)
  const int64_t big_blocks = ((M + 127) / 128) * ((N + 127) / 128);
  if (big_blocks >= 256)
    return gemm_dispatch_epi<T, 128, 128>(st, (const T*)A, (const T*)W, bias, out, resid, M, N, K, epi, lda);
  return gemm_dispatch_epi<T, 64, 64>(st, (const T*)A, (const T*)W, bias, out, resid, M, N, K, epi, lda);
}

int launch_gemm(hipStream_t st, int dtype, const void* A, const void* W, const float* bias, void* out, float* resid,
                int64_t M, int N, int K, int epi, const GemmLn* ln) {
  if (M <= 0 || N <= 0) return 0;
  if (epi >= EPI_RESID_LNP) {  // LayerNorm folded into the GEMM: gemm_glds.hip / gemm_8ph.hip only (callers ask gemm_ln_supports first)
    if (dtype != DT_BF16 || ln == nullptr) return -1;
    return launch_gemm_glds(st, A, W, bias, out, resid, M, N, K, epi, ln) == 0 ? 0 : -1;
  }
  if (dtype == DT_F32) {  // packed rows: the LDS-DMA ring of gemm_glds.hip on fp32 operands (bit-identical to the kernel above); else this file's
    if (launch_gemm_glds_f32(st, (const float*)A, (const float*)W, bias, out, resid, M, N, K, epi) == 0) return 0;
    return gemm_dispatch<float>(st, A, W, bias, out, resid, M, N, K, epi);
  }
  // bf16: the LDS-DMA pipelined kernel (gemm_glds.hip) when it has the shape
  if (launch_gemm_glds(st, A, W, bias, out, resid, M, N, K, epi) == 0) return 0;
  return gemm_dispatch<bf16_t>(st, A, W, bias, out, resid, M, N, K, epi);
}

// fp32 GEMM whose A rows start `lda` elements apart (lda * 4 bytes a multiple of 16, lda < K allowed: overlapping rows).  With a
// time-major [T][C] signal, row t of "A with lda = C, K = k * C" is the k consecutive frames t .. t+k-1 -- the im2col matrix of
// a 1-D convolution without materialising it (codec.hip).
int launch_gemm_f32_strided(hipStream_t st, const float* A, int64_t lda, const float* W, const float* bias, float* out, float* resid,
                            int64_t M, int N, int K, int epi) {
  if (M <= 0 || N <= 0) return 0;
  if (lda < 4 || lda % 4 != 0) return -1;
  return gemm_dispatch<float>(st, A, W, bias, out, resid, M, N, K, epi, lda);
}

}  // namespace vle
