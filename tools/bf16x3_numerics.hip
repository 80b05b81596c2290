// Numerics probe for the bf16x3 (6-term) emulation of an fp32 product on the bf16 matrix cores:
// one wave computes a 32x32 tile of C = A . B^T (A, B: [32][K] fp32) three ways and the host
// compares each with a float64 product.  Development tool (not part of the library).
//   hipcc -O3 --offload-arch=gfx950 tools/bf16x3_numerics.hip -o tools/bf16x3_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, unsigned short &h0, unsigned short &h1, unsigned short &h2) {
  // round-to-nearest-even at every level
  auto rne = [](float v) -> unsigned short {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  h0 = rne(x);
  float r1 = x - __uint_as_float((unsigned)h0 << 16);
  h1 = rne(r1);
  float r2 = r1 - __uint_as_float((unsigned)h1 << 16);
  h2 = rne(r2);
}

// mode 0: exact fp32 MFMA chain (32x32x2), mode 1: 6 terms one accumulator, mode 2: 6 terms,
// a0*b0 in one accumulator and the five correction terms in a second one, mode 3: 3 terms
__global__ void probe(const float *A, const float *B, int K, int mode, float *C) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc = {0}, lo = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[r * K + k + h], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      u16x8 a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 8 * h + j;
        const float av = k < K ? A[r * K + k] : 0.f, bv = k < K ? B[r * K + k] : 0.f;
        unsigned short x0, x1, x2;
        split3(av, x0, x1, x2); a[0][j] = x0; a[1][j] = x1; a[2][j] = x2;
        split3(bv, x0, x1, x2); b[0][j] = x0; b[1][j] = x1; b[2][j] = x2;
      }
      auto mm = [&](int i, int j, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                       __builtin_bit_cast(bf16x8, b[j]), c, 0, 0, 0);
      };
      if (mode == 1) {
        // small terms first
        acc = mm(2, 0, acc); acc = mm(0, 2, acc); acc = mm(1, 1, acc);
        acc = mm(1, 0, acc); acc = mm(0, 1, acc); acc = mm(0, 0, acc);
      } else if (mode == 2) {
        lo = mm(2, 0, lo); lo = mm(0, 2, lo); lo = mm(1, 1, lo); lo = mm(1, 0, lo); lo = mm(0, 1, lo);
        acc = mm(0, 0, acc);
      } else {
        acc = mm(1, 0, acc); acc = mm(0, 1, acc); acc = mm(0, 0, acc);
      }
    }
    if (mode == 2) for (int i = 0; i < 16; ++i) acc[i] += lo[i];
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = acc[i];
  }
}

int main() {
  const int K = 1000;
  std::mt19937 gen(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const char *names[] = {"N(0,1)", "wide dynamic range exp(N(0,3))*sign", "cancelling (x, -x+1e-3 pairs)",
                         "positive (no cancellation)"};
  for (int data = 0; data < 4; ++data) {
    std::vector<float> A(32 * K), B(32 * K);
    for (int i = 0; i < 32 * K; ++i) {
      float a = nd(gen), b = nd(gen);
      if (data == 1) { a = std::exp(3.f * nd(gen)) * (nd(gen) > 0 ? 1 : -1); b = std::exp(3.f * nd(gen)) * (nd(gen) > 0 ? 1 : -1); }
      if (data == 3) { a = std::fabs(a); b = std::fabs(b); }
      A[i] = a; B[i] = b;
    }
    if (data == 2)
      for (int r = 0; r < 32; ++r)
        for (int k = 0; k + 1 < K; k += 2) { A[r * K + k + 1] = A[r * K + k]; B[r * K + k + 1] = -B[r * K + k] + 1e-3f; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(32 * 32), mag(32 * 32);
    double refmax = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0, m = 0;
        for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[j * K + k]; m += std::fabs((double)A[i * K + k] * B[j * K + k]); }
        ref[i * 32 + j] = s; mag[i * 32 + j] = m; refmax = std::max(refmax, std::fabs(s));
      }
    printf("%s\n", names[data]);
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, K, mode, dC);
      std::vector<float> C(32 * 32);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double e1 = 0, e2 = 0, rms = 0;
      for (int i = 0; i < 32 * 32; ++i) {
        const double d = std::fabs(C[i] - ref[i]);
        e1 = std::max(e1, d / refmax); e2 = std::max(e2, d / mag[i]); rms += (d / mag[i]) * (d / mag[i]);
      }
      const char *mn[] = {"exact fp32 mfma       ", "bf16x3 6 terms 1 acc  ", "bf16x3 6 terms hi/lo  ", "bf16x3 3 terms        "};
      printf("  %s max|err|/max|ref| %.3e   max|err|/sum|a||b| %.3e   rms %.3e\n", mn[mode], e1, e2, std::sqrt(rms / 1024));
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
