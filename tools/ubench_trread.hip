// Probe: the lane map of gfx950's LDS transpose read `ds_read_b64_tr_b16` (the operand maps of cdna4_isa.md are not shipped in
// this image; the attention V^T pre-pass of attn_mfma2.hip exists because of that).  The LDS is filled with u16[i] = i, every
// lane passes a DISTINCT random 8-byte-aligned address, and the 4 x u16 each lane receives say which (lane's address, element)
// they came from:   result[lane][j] == (addr[src_lane] / 2 + k)   =>   (lane, j) <- (src_lane, k).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_trread tools/ubench_trread.hip && tools/bin/ubench_trread
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* __restrict__ addr, unsigned short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];  // byte offset, multiple of 8
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) unsigned char*)lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

int main() {
  int h_addr[64];
  unsigned short h_out[256];
  int* d_addr;
  unsigned short* d_out;
  CK(hipMalloc(&d_addr, sizeof(h_addr)));
  CK(hipMalloc(&d_out, sizeof(h_out)));
  // distinct 8-byte slots, far enough apart that (slot, element) decodes uniquely: lane l -> slot perm[l] * 8 elements (64 B apart)
  unsigned s = 12345u;
  int perm[64];
  for (int i = 0; i < 64; ++i) perm[i] = i;
  for (int i = 63; i > 0; --i) {
    s = s * 1664525u + 1013904223u;
    const int j = (int)((s >> 8) % (unsigned)(i + 1));
    const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
  for (int l = 0; l < 64; ++l) h_addr[l] = perm[l] * 64;  // bytes
  CK(hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
  printf("{\"note\": \"ds_read_b64_tr_b16: result[lane][j] came from element k of the 4 u16 at src_lane's address\",\n \"map\": [\n");
  for (int l = 0; l < 64; ++l) {
    printf("  {\"lane\": %d, \"src\": [", l);
    for (int j = 0; j < 4; ++j) {
      const int elem = h_out[l * 4 + j];  // u16 index in LDS
      const int byte = elem * 2;
      int src = -1, k = -1;
      for (int m = 0; m < 64; ++m)
        if (byte >= h_addr[m] && byte < h_addr[m] + 8) {
          src = m;
          k = (byte - h_addr[m]) / 2;
        }
      printf("[%d, %d]%s", src, k, j < 3 ? ", " : "");
    }
    printf("]}%s\n", l < 63 ? "," : "");
  }
  printf(" ]\n}\n");
  return 0;
}
