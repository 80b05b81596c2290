// Development probe: sustained rate of v_mfma_f32_32x32x16_bf16 on this device, one wave per SIMD,
// operands in registers, in the accumulator pattern of the GEMM (five chained on one accumulator,
// one on another, eight accumulator tiles).  Prints cycles per instruction and the in-kernel clock.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool ZERO>
__global__ __launch_bounds__(256) void probe(const float *seed, float *out, unsigned long long *stamps, int iters) {
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j) {
      const float va = ZERO ? 0.f : seed[(threadIdx.x * 8 + j + 64 * p) & 4095];
      const float vb = ZERO ? 0.f : seed[(threadIdx.x * 8 + j + 64 * p + 1000) & 4095];
      a[p][j] = (__bf16)va; b[p][j] = (__bf16)vb;
    }
  f32x16 hi[4], lo[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) { hi[t][r] = 0.f; lo[t][r] = 0.f; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      lo[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo[t], 0, 0, 0);
      hi[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi[t], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += hi[t][r] + lo[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same work on 16x16x32 tiles: 16 accumulator tiles of 4 registers per accumulator set
template <bool ZERO>
__global__ __launch_bounds__(256) void probe16(const float *seed, float *out, unsigned long long *stamps, int iters) {
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j) {
      const float va = ZERO ? 0.f : seed[(threadIdx.x * 8 + j + 64 * p) & 4095];
      const float vb = ZERO ? 0.f : seed[(threadIdx.x * 8 + j + 64 * p + 1000) & 4095];
      a[p][j] = (__bf16)va; b[p][j] = (__bf16)vb;
    }
  f32x4 hi[16], lo[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) { hi[t][r] = 0.f; lo[t][r] = 0.f; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], lo[t], 0, 0, 0);
      lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], lo[t], 0, 0, 0);
      hi[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], hi[t], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += hi[t][r] + lo[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks = 256, iters = 20000;
  std::vector<float> h(4096);
  for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *seed, *out; unsigned long long *st;
  hipMalloc(&seed, 4096 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
  hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int zero = 0; zero < 2; ++zero) {
    for (int rep = 0; rep < 3; ++rep) {
      if (zero) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters);
      else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> s(2 * blocks);
    hipMemcpy(s.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    const double cyc = (double)s[2 * 100] / (iters * 24.0), ghz = (double)s[2 * 100] / s[2 * 100 + 1] / 10.0;
    printf("%s data: %.2f cycles per MFMA, clock %.3f GHz, %.0f TFLOP/s bf16 over 1024 SIMDs\n",
           zero ? "zero  " : "random", cyc, ghz, 1024 * 32768.0 / cyc * ghz * 1e9 / 1e12);
  }
  for (int zero = 0; zero < 2; ++zero) {
    for (int rep = 0; rep < 3; ++rep) {
      if (zero) hipLaunchKernelGGL(probe16<true>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters / 4);
      else hipLaunchKernelGGL(probe16<false>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters / 4);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> s(2 * blocks);
    hipMemcpy(s.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    const double cyc = (double)s[2 * 100] / (iters / 4 * 96.0), ghz = (double)s[2 * 100] / s[2 * 100 + 1] / 10.0;
    printf("16x16x32 %s data: %.2f cycles per MFMA, clock %.3f GHz, %.0f TFLOP/s bf16 over 1024 SIMDs\n",
           zero ? "zero  " : "random", cyc, ghz, 1024 * 16384.0 / cyc * ghz * 1e9 / 1e12);
  }
  return 0;
}
