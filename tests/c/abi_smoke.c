/* C consumer of the drop-in boundary: include/valle_engine.h must be plain C, every call must link against
 * libvalle_engine.so, the host-side weight quantiser must work without a GPU, and engine creation must fail with a
 * status code + message (never crash) on a box without one.  Built and run by tests/test_abi_c_cpu.py. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "valle_engine.h"

int main(void) {
  /* 1. VLE_DTYPE_FP8W weight format on host buffers */
  float w[2 * 8] = {0.5f, -1.0f, 0.25f, 3.0f, -7.0f, 0.0f, 1e-4f, 2.5f, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned char q[16];
  float scale[2], deq[16];
  if (vle_quantize_fp8w(w, 2, 8, q, scale, deq) != VLE_OK) return 10;
  if (scale[0] != 0.015625f || scale[1] != 1.0f) return 11; /* 7 / 448 = 2^-6: the smallest admissible power of two */
  if (deq[4] != -7.0f || deq[1] != -1.0f || deq[5] != 0.0f) return 12;
  for (int i = 0; i < 8; ++i)
    if (fabsf(deq[i] - w[i]) > fabsf(w[i]) * 0.0625f + scale[0] * 0.002f) return 13;
  if (vle_quantize_fp8w(NULL, 2, 8, q, scale, deq) != VLE_EINVAL) return 14;

  /* 2. configuration errors are reported, not asserted */
  vle_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.d_model = 64; cfg.nhead = 4; cfg.num_layers = 1; cfg.num_quantizers = 8; cfg.prefix_mode = 1;
  cfg.norm_first = 0; /* post-norm: outside the decode path */
  cfg.dtype_mode = VLE_DTYPE_F32; cfg.max_batch = 1; cfg.max_text = 8; cfg.max_prompt = 8; cfg.use_graph = 1;
  vle_engine* e = NULL;
  int rc = vle_create(&cfg, &e);
  if (rc != VLE_EINVAL || e != NULL) return 20;
  const char* msg = vle_last_error(NULL);
  if (msg == NULL || strlen(msg) == 0) return 21;
  printf("vle_create(norm_first=0) -> %d: %s\n", rc, msg);

  /* 3. a valid configuration: succeeds on a GPU box, fails with VLE_EHIP (and a message) without a device */
  cfg.norm_first = 1;
  rc = vle_create(&cfg, &e);
  if (rc == VLE_OK) {
    if (e == NULL) return 30;
    if (vle_finalize_weights(e) == VLE_OK) return 31; /* no tensors loaded yet: must be an error */
    printf("vle_create ok; finalize without weights -> %s\n", vle_last_error(e));
    vle_destroy(e);
  } else {
    if (rc != VLE_EHIP || e != NULL) return 32;
    printf("vle_create without a GPU -> %d: %s\n", rc, vle_last_error(NULL));
  }
  printf("workspace bytes %lld\n", (long long)vle_op_linear_workspace_bytes());
  puts("abi_smoke ok");
  return 0;
}
