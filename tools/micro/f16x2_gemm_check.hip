// Can the channel contraction run on TWO fp16 pieces per fp32 operand instead of three bf16 ones?
//   x = h1 + h2 + d,  h1 = rn16(x), h2 = rn16(x - h1): 11 + 11 significant bits and a sign -> |d| <= 2^-23 |x| (one fp32 ulp at worst, a third of one in rms)
//   x y ~ h1 k1 + h1 k2 + h2 k1 [+ h2 k2]   (3 or 4 exact products on v_mfma_f32_32x32x16_f16, fp32 accumulate)
// against the six bf16 piece products of csrc/cgemm3m_bf16.hip and the fp32 instruction, all measured as distance to an fp64 product:
// one wave per 32 x 32 tile, K = 256, A ~ N(0, 1) * scale, B ~ N(0, 1) / (5 sqrt K) (tools/kbench_gemm_error.py's magnitudes).
// Also: are fp16 SUBNORMAL operands of the matrix instruction honoured or flushed?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/f16x2_gemm_check.hip -o tools/micro/_bin/f16x2_gemm_check && tools/micro/_bin/f16x2_gemm_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
constexpr int K = 256, T = 32;

// FORM 0: v_mfma_f32_32x32x2_f32; 1: bf16 x 3 pieces, 6 products (truncation split); 2: fp16 x 2, 3 products; 3: fp16 x 2, 4 products
template <int FORM>
__global__ __launch_bounds__(64) void gemm_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, float sa, float sb) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const float* a = A + (size_t)blockIdx.x * T * K;      // [row][k]
  const float* b = B + (size_t)blockIdx.x * K * T;      // [k][col]
  f32x16 acc = {};
  if (FORM == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * K + k + h], b[(k + h) * T + i], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      float av[8], bv[8];
      for (int j = 0; j < 8; ++j) { av[j] = a[i * K + k0 + 8 * h + j]; bv[j] = b[(k0 + 8 * h + j) * T + i]; }
      if (FORM == 1) {
        b8 ap[3], bp[3];
        for (int j = 0; j < 8; ++j) {
          float x = av[j];
          for (int p = 0; p < 3; ++p) {
            const float t = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
            ap[p][j] = __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, t) >> 16));
            x -= t;
          }
          x = bv[j];
          for (int p = 0; p < 3; ++p) {
            const float t = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
            bp[p][j] = __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, t) >> 16));
            x -= t;
          }
        }
        for (int w = 0; w < 3; ++w)
          for (int pa = 0; pa <= w; ++pa) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa], bp[w - pa], acc, 0, 0, 0);
      } else {
        h8 ap[2], bp[2];
        for (int j = 0; j < 8; ++j) {
          const float x = av[j] * sa, y = bv[j] * sb;
          ap[0][j] = (_Float16)x; ap[1][j] = (_Float16)(x - (float)ap[0][j]);
          bp[0][j] = (_Float16)y; bp[1][j] = (_Float16)(y - (float)bp[0][j]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bp[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bp[0], acc, 0, 0, 0);
        if (FORM == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bp[1], acc, 0, 0, 0);
      }
    }
  }
  const float inv = FORM >= 2 ? 1.0f / (sa * sb) : 1.0f;
  for (int e = 0; e < 16; ++e) C[(size_t)blockIdx.x * T * T + ((e & 3) + 8 * (e >> 2) + 4 * h) * T + i] = acc[e] * inv;
}

__global__ void subnormal_probe(float* out) {
  // A = 2^-20 (an fp16 subnormal: the smallest normal is 2^-14), B = 2^10: 16 products of 2^-10 each = 2^-6 if honoured, 0 if flushed
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.0f; }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
  // and a subnormal produced by the conversion itself
  const float x = 3.0e-6f;
  out[1 + threadIdx.x] = (float)(_Float16)(x * (1.0f + threadIdx.x));
}

int main() {
  const int tiles = 256;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A((size_t)tiles * T * K), B((size_t)tiles * K * T);
  for (auto& v : A) v = nd(rng);
  for (auto& v : B) v = nd(rng) / (5.0f * 16.0f);
  std::vector<double> want((size_t)tiles * T * T);
  for (int t = 0; t < tiles; ++t)
    for (int r = 0; r < T; ++r)
      for (int c = 0; c < T; ++c) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)A[((size_t)t * T + r) * K + k] * (double)B[((size_t)t * K + k) * T + c];
        want[((size_t)t * T + r) * T + c] = s;
      }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, want.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> got(want.size());
  auto report = [&](const char* name) {
    hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, ss = 0;
    for (size_t i = 0; i < got.size(); ++i) { const double d = std::fabs((double)got[i] - want[i]); mx = std::max(mx, d); ss += d * d; }
    printf("%-64s max %.3e  rms %.3e\n", name, mx, std::sqrt(ss / got.size()));
  };
  gemm_tile<0><<<tiles, 64>>>(dA, dB, dC, 1.f, 1.f); report("v_mfma_f32_32x32x2_f32");
  gemm_tile<1><<<tiles, 64>>>(dA, dB, dC, 1.f, 1.f); report("bf16 x 3 pieces, 6 products");
  for (float sa : {1.0f, 1024.0f, 1.0f / 1024.0f}) {
    char nm[128];
    snprintf(nm, sizeof nm, "fp16 x 2 pieces, 3 products (A scaled by %g, B by 64)", sa);
    gemm_tile<2><<<tiles, 64>>>(dA, dB, dC, sa, 64.f); report(nm);
    snprintf(nm, sizeof nm, "fp16 x 2 pieces, 4 products (A scaled by %g, B by 64)", sa);
    gemm_tile<3><<<tiles, 64>>>(dA, dB, dC, sa, 64.f); report(nm);
  }
  float* dP; hipMalloc(&dP, 65 * 4);
  subnormal_probe<<<1, 64>>>(dP);
  float p[65]; hipMemcpy(p, dP, sizeof p, hipMemcpyDeviceToHost);
  printf("subnormal operand probe: 16 x (2^-20 * 2^10) = %.6e (honoured: 1.5625e-02, flushed: 0)\n", p[0]);
  printf("conversion to a subnormal fp16: 3.0e-6 -> %.6e, 6.0e-6 -> %.6e (flushed: 0)\n", p[1], p[2]);
  return 0;
}
