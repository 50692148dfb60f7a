// EnCodec 24 kHz DECODER (codes -> waveform), the step right after VALLE.inference():
//   reference call site: audio_tokenizer.decode([(codes.transpose(2, 1), None)])  valle/bin/infer.py:261-263,
//   AudioTokenizer.decode -> self.codec.decode(frames)                               valle/data/tokenizer.py:241-242
// The arithmetic lives in the third-party `encodec` package (EncodecModel.encodec_model_24khz, 6 kbps = 8 codebooks;
// not vendored, not installed, weights not fetchable here): this file implements the PUBLISHED architecture restated in
// oracle/encodec_oracle.py (RVQ codebook sum -> SConv1d -> 2-layer LSTM with skip -> 4 x [ELU, causal ConvTranspose1d,
// residual block] -> ELU -> SConv1d) and is checked against that oracle; parity with real weights is UNPINNED.
//
// gfx950 design.  Every signal is kept TIME-MAJOR, fp32: x[t][c], channels contiguous.  Then
//   * a causal k-tap convolution is ONE GEMM with overlapping A rows: row t of "A with lda = C, K = k C" is the k consecutive
//     frames t .. t+k-1 of the left-padded signal -- the im2col matrix never exists (launch_gemm_f32_strided, gemm.hip);
//   * a stride-r transposed convolution with kernel 2r (all four of EnCodec's) is ONE GEMM too: output frame t r + j takes tap j
//     of input t and tap j + r of input t-1, so out[T][r C_out] = [x[t-1] ; x[t]] (K = 2 C_in, lda = C_in, one zero frame in
//     front) x W'[(j, c_out)][...], and the [T][r C_out] result IS the time-major [T r][C_out] signal; the causal right-trim of
//     k - r samples is exactly the frame that is never computed;
//   * ELU, the reflect / zero left padding and the residual add ride on one elementwise pass / the GEMM epilogue;
//   * the LSTM's input projections are GEMMs over all T frames; the recurrence is one small kernel per time step (64 blocks,
//     4 MB of W_hh resident in L2 / MALL), ~1500 launches per utterance (a few ms per 10 s of audio).
// All GEMMs run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains): ~30 GFLOP per 10 s utterance, milliseconds.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "valle_engine.h"

namespace vle {

// ---- elementwise / gather kernels --------------------------------------------------------------------------------------
// out[t][:] = sum_q codebook_q[codes[t][q]][:]   (ResidualVectorQuantization.decode); one wave per frame, dim = 128
__global__ __launch_bounds__(256) void rvq_decode_kernel(const int64_t* __restrict__ codes, const float* const* __restrict__ books,
                                                          float* __restrict__ out, int64_t T, int Q, int dim, int bins) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  for (int c = lane; c < dim; c += 64) {
    float acc = 0.f;
    for (int q = 0; q < Q; ++q) {
      int64_t id = codes[t * Q + q];
      id = id < 0 ? 0 : (id >= bins ? bins - 1 : id);  // the caller range-checks; stay in bounds regardless
      acc += books[q][id * dim + c];
    }
    out[t * dim + c] = acc;
  }
}

// dst[(P + T)][C]: P frames of left padding (mode 0: zeros, mode 1: reflect as encodec.modules.conv.pad1d -- inputs not longer
// than P are zero-extended first), then act(src) with act = ELU(alpha 1) or identity; optional second addend (LSTM skip)
__global__ __launch_bounds__(256) void pad_act_kernel(const float* __restrict__ src, const float* __restrict__ add, float* __restrict__ dst,
                                                       int64_t T, int C, int P, int reflect, int elu) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // float4 index
  const int c4 = C >> 2;
  if (i >= (T + P) * c4) return;
  const int64_t row = i / c4;
  const int col = (int)(i % c4);
  int64_t srow;
  bool zero = false;
  if (row >= P) srow = row - P;
  else if (!reflect) zero = true;
  else {
    srow = P - row;          // reflect without repeating the edge sample
    zero = srow >= T;        // the zero-extension of a too-short input
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!zero) {
    v = reinterpret_cast<const float4*>(src)[srow * c4 + col];
    if (add != nullptr) {
      const float4 a = reinterpret_cast<const float4*>(add)[srow * c4 + col];
      v = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    }
    if (elu) {
      v.x = v.x > 0.f ? v.x : expm1f(v.x);
      v.y = v.y > 0.f ? v.y : expm1f(v.y);
      v.z = v.z > 0.f ? v.z : expm1f(v.z);
      v.w = v.w > 0.f ? v.w : expm1f(v.w);
    }
  }
  reinterpret_cast<float4*>(dst)[i] = v;
}

// One LSTM time step (torch.nn.LSTM, gate order i, f, g, o): block b owns hidden units 8 b .. 8 b + 7; wave g computes gate g's
// 8 pre-activations gx[g H + u] + W_hh[g H + u] . h_prev (each lane 8 of the 512 columns), thread u finishes cell and hidden state
template <int H>
__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ gx, const float* __restrict__ whh, const float* __restrict__ h_prev,
                                                         float* __restrict__ c_state, float* __restrict__ h_out, const float* __restrict__ skip,
                                                         float* __restrict__ y_out) {
  static_assert(H % 256 == 0, "H = 64 lanes x float4 chunks");
  constexpr int NV = H / 256;
  __shared__ float sm[4][8];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 8;
  float4 hv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) hv[i] = reinterpret_cast<const float4*>(h_prev)[i * 64 + lane];
  float4 w[8][NV];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < NV; ++i) w[j][i] = reinterpret_cast<const float4*>(whh + (int64_t)(g * H + u0 + j) * H)[i * 64 + lane];
  float mine = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (w[j][i].x * hv[i].x + w[j][i].y * hv[i].y) + (w[j][i].z * hv[i].z + w[j][i].w * hv[i].w);
    s = wave_sum(s);
    if (lane == j) mine = s;
  }
  if (lane < 8) sm[g][lane] = mine + gx[g * H + u0 + lane];
  __syncthreads();
  if (threadIdx.x < 8) {
    const int u = u0 + threadIdx.x;
    const float ig = 1.f / (1.f + expf(-sm[0][threadIdx.x]));
    const float fg = 1.f / (1.f + expf(-sm[1][threadIdx.x]));
    const float gg = tanhf(sm[2][threadIdx.x]);
    const float og = 1.f / (1.f + expf(-sm[3][threadIdx.x]));
    const float c = fg * c_state[u] + ig * gg;
    const float h = og * tanhf(c);
    c_state[u] = c;
    h_out[u] = h;
    if (y_out != nullptr) y_out[u] = h + skip[u];
  }
}

}  // namespace vle

using namespace vle;

namespace {
constexpr int CD_DIM = 128, CD_NF = 32, CD_BINS = 1024, CD_H = 512, CD_HOP = 320;
const int CD_RATIOS[4] = {8, 5, 4, 2};

struct ConvW {  // GEMM form of one convolution: out[t][n] = bias[n] + sum_k A[t][k] * w[n][k]
  float *w = nullptr, *b = nullptr;
  int N = 0, K = 0, lda = 0, pad = 0;  // pad = frames of left padding the A operand needs (k - 1, or 1 for the transposed conv)
};
}  // namespace

struct vle_codec {
  int device = 0, n_q = 8;
  bool finalized = false;
  std::string err;
  hipStream_t st = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  std::map<std::string, std::vector<float>> host_w;
  std::map<std::string, std::vector<int64_t>> host_shape;
  std::vector<void*> allocs;
  const float** books_dev = nullptr;
  ConvW conv0, fin, tr[4], b1[4], b3[4], sc[4];
  float *wih[2] = {nullptr, nullptr}, *whh[2] = {nullptr, nullptr}, *bsum[2] = {nullptr, nullptr};
  // scratch, grown on demand
  int64_t cap_T = 0;
  float *X = nullptr, *P = nullptr, *Hd = nullptr, *S = nullptr, *GX = nullptr, *Y1 = nullptr, *Y2 = nullptr, *cst = nullptr;
  std::vector<void*> scratch;
  int fail(int code, const std::string& m) {
    err = m;
    return code;
  }
};

namespace {

#define C_HIP(c, expr)                                                                            \
  do {                                                                                            \
    hipError_t _r = (expr);                                                                       \
    if (_r != hipSuccess) return (c)->fail(VLE_EHIP, std::string("HIP error: ") + hipGetErrorString(_r) + " at " #expr); \
  } while (0)
#define C_LAUNCH(c, expr)                                                \
  do {                                                                   \
    if ((expr) != 0) return (c)->fail(VLE_EINVAL, "launch rejected: " #expr); \
  } while (0)

int c_upload(vle_codec* c, float** dst, const std::vector<float>& v) {
  void* p = nullptr;
  C_HIP(c, hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(float)));
  c->allocs.push_back(p);
  C_HIP(c, hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  *dst = (float*)p;
  return 0;
}

const std::vector<float>* c_find(vle_codec* c, const std::string& key, std::initializer_list<int64_t> shape) {
  auto it = c->host_w.find(key);
  if (it == c->host_w.end()) return nullptr;
  const auto& sh = c->host_shape[key];
  if (sh.size() != shape.size() || !std::equal(sh.begin(), sh.end(), shape.begin())) return nullptr;
  return &it->second;
}

// torch.nn.utils.weight_norm (dim 0): w = g * v / ||v||, the norm over every dim but 0; also a plain `weight`
int c_weight(vle_codec* c, const std::string& prefix, int64_t d0, int64_t d1, int64_t d2, std::vector<float>& w) {
  if (const auto* p = c_find(c, prefix + ".weight", {d0, d1, d2})) {
    w = *p;
    return 0;
  }
  const auto* g = c_find(c, prefix + ".weight_g", {d0, 1, 1});
  const auto* v = c_find(c, prefix + ".weight_v", {d0, d1, d2});
  if (!g) g = c_find(c, prefix + ".parametrizations.weight.original0", {d0, 1, 1});
  if (!v) v = c_find(c, prefix + ".parametrizations.weight.original1", {d0, d1, d2});
  if (!g || !v) return c->fail(VLE_EKEY, "missing / mis-shaped weight: " + prefix + ".weight_g / weight_v");
  w.resize((size_t)(d0 * d1 * d2));
  const int64_t inner = d1 * d2;
  for (int64_t i = 0; i < d0; ++i) {
    double n2 = 0.0;
    for (int64_t j = 0; j < inner; ++j) n2 += (double)(*v)[i * inner + j] * (double)(*v)[i * inner + j];
    const float s = (float)((double)(*g)[i] / std::sqrt(n2));
    for (int64_t j = 0; j < inner; ++j) w[i * inner + j] = (*v)[i * inner + j] * s;
  }
  return 0;
}

// Conv1d weight (cout, cin, k) -> GEMM operand [cout][k * cin] (tap-major), K padded with zero columns to a multiple of 32
int c_load_conv(vle_codec* c, const std::string& prefix, int cout, int cin, int k, ConvW& out) {
  std::vector<float> w;
  int r = c_weight(c, prefix + ".conv.conv", cout, cin, k, w);
  if (r) return r;
  const auto* b = c_find(c, prefix + ".conv.conv.bias", {cout});
  if (!b) return c->fail(VLE_EKEY, "missing bias: " + prefix);
  const int K = (k * cin + 31) / 32 * 32;
  std::vector<float> g((size_t)cout * K, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int j = 0; j < k; ++j) g[(size_t)co * K + j * cin + ci] = w[((size_t)co * cin + ci) * k + j];
  out.N = cout; out.K = K; out.lda = cin; out.pad = k - 1;
  if ((r = c_upload(c, &out.w, g))) return r;
  return c_upload(c, &out.b, *b);
}

// ConvTranspose1d weight (cin, cout, 2r), stride r -> GEMM operand [(j, cout)][2 cin]: columns [0, cin) multiply x[t-1] with tap
// j + r, columns [cin, 2 cin) multiply x[t] with tap j
int c_load_convtr(vle_codec* c, const std::string& prefix, int cin, int cout, int rr, ConvW& out) {
  std::vector<float> w;
  int r = c_weight(c, prefix + ".convtr.convtr", cin, cout, 2 * rr, w);
  if (r) return r;
  const auto* b = c_find(c, prefix + ".convtr.convtr.bias", {cout});
  if (!b) return c->fail(VLE_EKEY, "missing bias: " + prefix);
  const int K = 2 * cin, N = rr * cout;
  std::vector<float> g((size_t)N * K, 0.f), bb((size_t)N);
  for (int j = 0; j < rr; ++j)
    for (int co = 0; co < cout; ++co) {
      bb[(size_t)j * cout + co] = (*b)[co];
      for (int ci = 0; ci < cin; ++ci) {
        g[((size_t)j * cout + co) * K + ci] = w[((size_t)ci * cout + co) * 2 * rr + j + rr];
        g[((size_t)j * cout + co) * K + cin + ci] = w[((size_t)ci * cout + co) * 2 * rr + j];
      }
    }
  out.N = N; out.K = K; out.lda = cin; out.pad = 1;
  if ((r = c_upload(c, &out.w, g))) return r;
  return c_upload(c, &out.b, bb);
}

int c_reserve(vle_codec* c, int64_t T) {
  if (T <= c->cap_T) return 0;
  for (void* p : c->scratch) (void)hipFree(p);
  c->scratch.clear();
  c->cap_T = 0;
  const size_t big = (size_t)(T * CD_HOP + 64) * CD_NF + 4096;  // the widest signal: [320 T][32] (= [T][10240]) + slack for K padding
  float** bufs[4] = {&c->X, &c->P, &c->Hd, &c->S};
  for (auto b : bufs) {
    void* p = nullptr;
    C_HIP(c, hipMalloc(&p, big * sizeof(float)));
    C_HIP(c, hipMemset(p, 0, big * sizeof(float)));
    c->scratch.push_back(p);
    *b = (float*)p;
  }
  const size_t ts[4] = {(size_t)T * 4 * CD_H, (size_t)(T + 1) * CD_H, (size_t)(T + 1) * CD_H, (size_t)2 * CD_H};
  float** tb[4] = {&c->GX, &c->Y1, &c->Y2, &c->cst};
  for (int i = 0; i < 4; ++i) {
    void* p = nullptr;
    C_HIP(c, hipMalloc(&p, ts[i] * sizeof(float)));
    C_HIP(c, hipMemset(p, 0, ts[i] * sizeof(float)));
    c->scratch.push_back(p);
    *tb[i] = (float*)p;
  }
  c->cap_T = T;
  return 0;
}

int c_pad(vle_codec* c, const float* src, const float* add, float* dst, int64_t T, int C, int P, int reflect, int elu) {
  const int64_t n4 = (T + P) * (C / 4);
  hipLaunchKernelGGL(pad_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, c->st, src, add, dst, T, C, P, reflect, elu);
  return 0;
}

// out (or resid +=) = conv(A) where A already holds the padded / activated signal
int c_gemm(vle_codec* c, const ConvW& w, const float* A, float* out, float* resid, int64_t M) {
  return launch_gemm_f32_strided(c->st, A, w.lda, w.w, w.b, out, resid, M, w.N, w.K, resid ? EPI_RESID : EPI_F32);
}

}  // namespace

extern "C" int vle_codec_create(int32_t device, int32_t n_q, vle_codec** out) {
  if (!out || n_q < 1 || n_q > 32) return VLE_EINVAL;
  if (hipSetDevice(device) != hipSuccess) {
    set_global_error("vle_codec_create: hipSetDevice failed (no such GPU)");
    return VLE_EHIP;
  }
  vle_codec* c = new vle_codec();
  c->device = device;
  c->n_q = n_q;
  if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
    delete c;
    set_global_error("vle_codec_create: stream / event creation failed");
    return VLE_EHIP;
  }
  *out = c;
  return VLE_OK;
}

extern "C" void vle_codec_destroy(vle_codec* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->st) (void)hipStreamSynchronize(c->st);
  for (void* p : c->allocs) (void)hipFree(p);
  for (void* p : c->scratch) (void)hipFree(p);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

extern "C" const char* vle_codec_last_error(const vle_codec* c) { return c ? c->err.c_str() : vle_last_error(nullptr); }

extern "C" int vle_codec_load_tensor(vle_codec* c, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!c || !key || !data || !shape || ndim < 1 || ndim > 3) return VLE_EINVAL;
  if (c->finalized) return c->fail(VLE_ESTATE, "vle_codec_load_tensor after vle_codec_finalize");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  c->host_w[key].assign(data, data + n);
  c->host_shape[key].assign(shape, shape + ndim);
  return VLE_OK;
}

extern "C" int vle_codec_finalize(vle_codec* c) {
  if (!c) return VLE_EINVAL;
  if (c->finalized) return VLE_OK;
  C_HIP(c, hipSetDevice(c->device));
  int r;
  std::vector<const float*> books((size_t)c->n_q);
  for (int q = 0; q < c->n_q; ++q) {
    const auto* e = c_find(c, "quantizer.vq.layers." + std::to_string(q) + "._codebook.embed", {CD_BINS, CD_DIM});
    if (!e) return c->fail(VLE_EKEY, "missing codebook " + std::to_string(q));
    float* d = nullptr;
    if ((r = c_upload(c, &d, *e))) return r;
    books[q] = d;
  }
  {
    void* p = nullptr;
    C_HIP(c, hipMalloc(&p, books.size() * sizeof(float*)));
    c->allocs.push_back(p);
    C_HIP(c, hipMemcpy(p, books.data(), books.size() * sizeof(float*), hipMemcpyHostToDevice));
    c->books_dev = (const float**)p;
  }
  int ch = CD_NF * 16;  // 512
  if ((r = c_load_conv(c, "decoder.model.0", ch, CD_DIM, 7, c->conv0))) return r;
  for (int l = 0; l < 2; ++l) {
    const std::string p = "decoder.model.1.lstm.";
    const auto* wi = c_find(c, p + "weight_ih_l" + std::to_string(l), {4 * CD_H, CD_H});
    const auto* wh = c_find(c, p + "weight_hh_l" + std::to_string(l), {4 * CD_H, CD_H});
    const auto* bi = c_find(c, p + "bias_ih_l" + std::to_string(l), {4 * CD_H});
    const auto* bh = c_find(c, p + "bias_hh_l" + std::to_string(l), {4 * CD_H});
    if (!wi || !wh || !bi || !bh) return c->fail(VLE_EKEY, "missing LSTM tensors of layer " + std::to_string(l));
    std::vector<float> bs(4 * CD_H);
    for (int i = 0; i < 4 * CD_H; ++i) bs[i] = (*bi)[i] + (*bh)[i];
    if ((r = c_upload(c, &c->wih[l], *wi)) || (r = c_upload(c, &c->whh[l], *wh)) || (r = c_upload(c, &c->bsum[l], bs))) return r;
  }
  int idx = 2;
  for (int s = 0; s < 4; ++s) {
    const int rr = CD_RATIOS[s];
    if ((r = c_load_convtr(c, "decoder.model." + std::to_string(idx + 1), ch, ch / 2, rr, c->tr[s]))) return r;
    const std::string rb = "decoder.model." + std::to_string(idx + 2);
    if ((r = c_load_conv(c, rb + ".block.1", ch / 4, ch / 2, 3, c->b1[s]))) return r;
    if ((r = c_load_conv(c, rb + ".block.3", ch / 2, ch / 4, 1, c->b3[s]))) return r;
    if ((r = c_load_conv(c, rb + ".shortcut", ch / 2, ch / 2, 1, c->sc[s]))) return r;
    ch /= 2;
    idx += 3;
  }
  if ((r = c_load_conv(c, "decoder.model." + std::to_string(idx + 1), 1, CD_NF, 7, c->fin))) return r;
  c->host_w.clear();
  c->host_shape.clear();
  c->finalized = true;
  return VLE_OK;
}

extern "C" int vle_codec_decode(vle_codec* c, void* stream, const int64_t* codes, int64_t T, float* wav) {
  if (!c) return VLE_EINVAL;
  if (!c->finalized) return c->fail(VLE_ESTATE, "weights not finalized");
  if (!codes || !wav || T < 1) return c->fail(VLE_EINVAL, "bad argument");
  C_HIP(c, hipSetDevice(c->device));
  int r;
  C_HIP(c, hipStreamSynchronize(c->st));
  if ((r = c_reserve(c, T))) return r;
  C_HIP(c, hipEventRecord(c->ev_in, (hipStream_t)stream));
  C_HIP(c, hipStreamWaitEvent(c->st, c->ev_in, 0));
  hipStream_t st = c->st;
  // RVQ decode -> X [T][128]
  hipLaunchKernelGGL(rvq_decode_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, codes, c->books_dev, c->X, T, c->n_q, CD_DIM, CD_BINS);
  // model.0: SConv1d(128 -> 512, k 7): reflect pad 6, GEMM -> S [T][512]
  c_pad(c, c->X, nullptr, c->P, T, CD_DIM, c->conv0.pad, 1, 0);
  C_LAUNCH(c, c_gemm(c, c->conv0, c->P, c->S, nullptr, T));
  // model.1: SLSTM.  Layer l: GX = in . W_ih^T + (b_ih + b_hh) over all frames, then T recurrent steps; h_t is row t + 1 of Y
  // (row 0 = the zero initial state); the second layer's steps also write y = h + skip into X [T][512]
  const float* lin = c->S;
  float* Y[2] = {c->Y1, c->Y2};
  for (int l = 0; l < 2; ++l) {
    C_LAUNCH(c, launch_gemm_f32_strided(st, lin, CD_H, c->wih[l], c->bsum[l], c->GX, nullptr, T, 4 * CD_H, CD_H, EPI_F32));
    C_HIP(c, hipMemsetAsync(Y[l], 0, CD_H * sizeof(float), st));
    C_HIP(c, hipMemsetAsync(c->cst + l * CD_H, 0, CD_H * sizeof(float), st));
    for (int64_t t = 0; t < T; ++t)
      hipLaunchKernelGGL((lstm_step_kernel<CD_H>), dim3(CD_H / 8), dim3(256), 0, st, c->GX + t * 4 * CD_H, c->whh[l], Y[l] + t * CD_H,
                         c->cst + l * CD_H, Y[l] + (t + 1) * CD_H, l == 1 ? c->S + t * CD_H : nullptr, l == 1 ? c->X + t * CD_H : nullptr);
    lin = Y[l] + CD_H;
  }
  // four upsampling stages
  int ch = CD_H;
  int64_t len = T;
  float* x = c->X;
  for (int s = 0; s < 4; ++s) {
    const int rr = CD_RATIOS[s];
    c_pad(c, x, nullptr, c->P, len, ch, 1, 0, 1);                               // [0 ; ELU(x)]
    C_LAUNCH(c, c_gemm(c, c->tr[s], c->P, c->Hd, nullptr, len));                  // -> Hd = [len * rr][ch / 2]
    len *= rr;
    ch /= 2;
    C_LAUNCH(c, c_gemm(c, c->sc[s], c->Hd, c->S, nullptr, len));                  // shortcut (k 1, no activation) -> S
    c_pad(c, c->Hd, nullptr, c->P, len, ch, c->b1[s].pad, 1, 1);                   // reflect pad 2, ELU
    C_LAUNCH(c, c_gemm(c, c->b1[s], c->P, c->X, nullptr, len));                   // k 3 -> X [len][ch / 2]
    c_pad(c, c->X, nullptr, c->P, len, ch / 2, 0, 0, 1);                           // ELU
    C_LAUNCH(c, c_gemm(c, c->b3[s], c->P, nullptr, c->S, len));                   // k 1, S += ...
    x = c->S;
    std::swap(c->S, c->X);                                                         // the stage's output is x = (old S)
    x = c->X;
  }
  // model.14, 15: ELU, SConv1d(32 -> 1, k 7) -> wav [len]
  c_pad(c, x, nullptr, c->P, len, ch, c->fin.pad, 1, 1);
  C_LAUNCH(c, launch_gemm_f32_strided(st, c->P, c->fin.lda, c->fin.w, c->fin.b, wav, nullptr, len, 1, c->fin.K, EPI_F32));
  C_HIP(c, hipEventRecord(c->ev_out, st));
  C_HIP(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_out, 0));
  return VLE_OK;
}
