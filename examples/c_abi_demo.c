/* The C ABI without Python or torch: canonicalize a small batch through libeqa_hip.so from plain C.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -Iinclude -I/opt/rocm/include -Lequiadapt_amd/csrc -leqa_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/equiadapt_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 * (the HIP runtime is used only for device memory; -D__HIP_PLATFORM_AMD__ is what hip_runtime_api.h asks of a plain C compiler)
 *
 * Two group elements whose affine rows are exact in fp32: the identity and the rotation by 180 degrees (theta = -I), on a
 * frame edge-padded by ceil(W/2) like the reference's canonicalize (discrete_group.py:204-215).  The identity must return
 * the image, the half turn the image flipped along both axes.  Exit code 0 on success.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "eqa_hip.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "hip error %d at line %d\n", (int)r_, __LINE__); return 2; } } while (0)

int main(void) {
  enum { B = 4, C = 3, H = 8, W = 8, PAD = 4 };
  const size_t n = (size_t)B * C * H * W;
  float* hx = (float*)malloc(n * sizeof(float));
  float* hy = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
  const float theta[2][6] = {{1, 0, 0, 0, 1, 0}, {-1, 0, 0, 0, -1, 0}};
  const int32_t gidx[B] = {0, 1, 1, 0};

  if (eqa_abi_version() != EQA_ABI_VERSION) { fprintf(stderr, "unexpected ABI version %d\n", eqa_abi_version()); return 3; }
  float *dx, *dy, *dth;
  int32_t* dg;
  CHECK_HIP(hipMalloc((void**)&dx, n * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dy, n * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dth, sizeof(theta)));
  CHECK_HIP(hipMalloc((void**)&dg, sizeof(gidx)));
  CHECK_HIP(hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dth, theta, sizeof(theta), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dg, gidx, sizeof(gidx), hipMemcpyHostToDevice));

  const int rc = eqa_canon_transform_fwd(dx, dy, dg, dth, NULL, 2, B, C, H, W, PAD, NULL /* default stream */);
  if (rc != EQA_OK) { fprintf(stderr, "eqa_canon_transform_fwd returned %d\n", rc); return 4; }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(hy, dy, n * sizeof(float), hipMemcpyDeviceToHost));

  double worst = 0.0;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
          const size_t o = (((size_t)b * C + c) * H + i) * W + j;
          const size_t s = gidx[b] ? (((size_t)b * C + c) * H + (H - 1 - i)) * W + (W - 1 - j) : o;
          const double d = hy[o] > hx[s] ? hy[o] - hx[s] : hx[s] - hy[o];
          if (d > worst) worst = d;
        }
  /* argument validation is part of the contract */
  const int bad = eqa_canon_transform_fwd(NULL, dy, dg, dth, NULL, 2, B, C, H, W, PAD, NULL);
  printf("max |error| = %.3g, invalid-argument call returned %d\n", worst, bad);
  hipFree(dx); hipFree(dy); hipFree(dth); hipFree(dg);
  free(hx); free(hy);
  return (worst <= 1e-5 && bad == EQA_ERR_INVALID_ARG) ? 0 : 1;
}
