// Stand-alone lab for the fp32 MFMA GEMM main loop (not part of the library): times ablations
// of the production tile loop to see where matrix-pipe time is lost.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gemm_lab tools/gemm_lab.hip && ./tools/gemm_lab
// C[M,N] = A[M,K] . B[N,K]^T  (the "NT" forward layer shape), all dims multiples of 128/32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32, LS = 132, TF = BK * LS;

// FLAGS: 1 = no global loads / LDS stores in the loop, 2 = no barrier in the loop,
//        4 = no LDS fragment reads (operands stay in registers)
template <int FLAGS, int OCC>
__global__ __launch_bounds__(256, OCC) void lab_kernel(const float *A, const float *B, float *C,
                                                       int M, int N, int K, int tiles_n) {
  __shared__ __align__(16) float smem[OCC == 1 ? 25600 : 4 * TF];  // 100 KB forces one block per CU
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  const int tn = L % tiles_n, tm = L / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4], rb[4];
  const int lx = t >> 3, lk = (t & 7) * 4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      ra[it] = *reinterpret_cast<const float4 *>(A + (size_t)(m0 + lx + 32 * it) * K + k0 + lk);
      rb[it] = *reinterpret_cast<const float4 *>(B + (size_t)(n0 + lx + 32 * it) * K + k0 + lk);
    }
  };
  auto sstore = [&](float *as, float *bs) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int x = lx + 32 * it;
      as[(lk + 0) * LS + x] = ra[it].x; as[(lk + 1) * LS + x] = ra[it].y;
      as[(lk + 2) * LS + x] = ra[it].z; as[(lk + 3) * LS + x] = ra[it].w;
      bs[(lk + 0) * LS + x] = rb[it].x; bs[(lk + 1) * LS + x] = rb[it].y;
      bs[(lk + 2) * LS + x] = rb[it].z; bs[(lk + 3) * LS + x] = rb[it].w;
    }
  };
  float *As = smem, *Bs = smem + 2 * TF;
  gload(0);
  sstore(As, Bs);
  __syncthreads();
  const int kh = lane >> 5, xl = lane & 31;
  const int nt = K / BK;
  for (int it = 0; it < nt; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < nt;
    if (!(FLAGS & 1) && more) gload((it + 1) * BK);
    const float *as = As + cur * TF + wm * 64 + xl + kh * LS;
    const float *bs = Bs + cur * TF + wn * 64 + xl + kh * LS;
    float fa[2][2], fb[2][2];
    fa[0][0] = as[0]; fa[0][1] = as[32]; fb[0][0] = bs[0]; fb[0][1] = bs[32];
    if (FLAGS & 4) { fa[1][0] = fa[0][0]; fa[1][1] = fa[0][1]; fb[1][0] = fb[0][0]; fb[1][1] = fb[0][1]; }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int c = (kk >> 1) & 1, nx = c ^ 1;
      if (!(FLAGS & 4) && kk + 2 < BK) {
        fa[nx][0] = as[(kk + 2) * LS]; fa[nx][1] = as[(kk + 2) * LS + 32];
        fb[nx][0] = bs[(kk + 2) * LS]; fb[nx][1] = bs[(kk + 2) * LS + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
      if (!(FLAGS & 4)) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    if (!(FLAGS & 1) && more) sstore(As + (cur ^ 1) * TF, Bs + (cur ^ 1) * TF);
    if (!(FLAGS & 2)) __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + xl;
      const int rbase = m0 + wm * 64 + i * 32 + 4 * kh;
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * N + col] = acc[i][j][r];
    }
}


// ---- v2: write-after-barrier pipeline, row-major operands staged as they are ([x][k], stride 36,
// ds_write_b128 / ds_read_b128; an MFMA step pairs k = 8*kb+s with 8*kb+4+s), k-major operands
// staged [k][x] (stride 132, ds_read_b32)
constexpr int SX = 36, SK = 132;
constexpr int TFX = 128 * SX, TFK = 32 * SK;
constexpr int TF2 = TFX > TFK ? TFX : TFK;

template <bool KX>
struct Op {
  float r[16];
  const float *p[4];
  __device__ __forceinline__ void init(const float *base, int ld, int x0, int t) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (!KX) p[it] = base + (size_t)(x0 + (t >> 3) + 32 * it) * ld + (t & 7) * 4;
      else     p[it] = base + (size_t)((t >> 5) + 8 * it) * ld + x0 + (t & 31) * 4;
    }
  }
  __device__ __forceinline__ void load(int ld) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float4 v = *reinterpret_cast<const float4 *>(p[it]);
      r[4 * it] = v.x; r[4 * it + 1] = v.y; r[4 * it + 2] = v.z; r[4 * it + 3] = v.w;
      p[it] += KX ? (size_t)BK * ld : (size_t)BK;
    }
  }
  __device__ __forceinline__ void store(float *s, int t) const {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float4 v = make_float4(r[4 * it], r[4 * it + 1], r[4 * it + 2], r[4 * it + 3]);
      if (!KX) *reinterpret_cast<float4 *>(&s[((t >> 3) + 32 * it) * SX + (t & 7) * 4]) = v;
      else     *reinterpret_cast<float4 *>(&s[((t >> 5) + 8 * it) * SK + (t & 31) * 4]) = v;
    }
  }
  // fragments of k-block kb for the two 32-wide sub-tiles of this wave: f[sub][s]
  __device__ __forceinline__ void frag(const float *s, int w64, int lane, int kb, float (&f)[2][4]) const {
    const int xl = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (!KX) {
        const float4 v = *reinterpret_cast<const float4 *>(&s[(w64 + sub * 32 + xl) * SX + 8 * kb + 4 * kh]);
        f[sub][0] = v.x; f[sub][1] = v.y; f[sub][2] = v.z; f[sub][3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[sub][q] = s[(8 * kb + 4 * kh + q) * SK + w64 + sub * 32 + xl];
      }
    }
  }
};

template <bool A_KX, bool B_KX>
__global__ __launch_bounds__(256, 2) void lab2_kernel(const float *A, const float *B, float *C,
                                                      int M, int N, int K, int tiles_n) {
  __shared__ __align__(16) float smem[4 * TF2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  const int tn = L % tiles_n, tm = L / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lda = A_KX ? M : K, ldb = B_KX ? N : K;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  Op<A_KX> oa; Op<B_KX> ob;
  oa.init(A, lda, m0, t); ob.init(B, ldb, n0, t);
  float *As = smem, *Bs = smem + 2 * TF2;
  const int nt = K / BK;
  oa.load(lda); ob.load(ldb);
  oa.store(As, t); ob.store(Bs, t);
  if (nt > 1) { oa.load(lda); ob.load(ldb); }
  for (int it = 0; it < nt; ++it) {
    const int cur = it & 1;
    __syncthreads();
    if (it + 1 < nt) { oa.store(As + (cur ^ 1) * TF2, t); ob.store(Bs + (cur ^ 1) * TF2, t); }
    if (it + 2 < nt) { oa.load(lda); ob.load(ldb); }
    const float *as = As + cur * TF2, *bs = Bs + cur * TF2;
    float fa[2][2][4], fb[2][2][4];
    oa.frag(as, wm * 64, lane, 0, fa[0]); ob.frag(bs, wn * 64, lane, 0, fb[0]);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int c = kb & 1;
      if (kb + 1 < 4) { oa.frag(as, wm * 64, lane, kb + 1, fa[c ^ 1]); ob.frag(bs, wn * 64, lane, kb + 1, fb[c ^ 1]); }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][s2], fb[c][0][s2], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][s2], fb[c][1][s2], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][s2], fb[c][0][s2], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][s2], fb[c][1][s2], acc[1][1], 0, 0, 0);
      }
    }
  }
  const int kh = lane >> 5, xl = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + xl;
      const int rbase = m0 + wm * 64 + i * 32 + 4 * kh;
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * N + col] = acc[i][j][r];
    }
}

template <bool A_KX, bool B_KX>
float run2(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  const int tiles_n = N / BN, nwg = (M / BM) * tiles_n;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab2_kernel<A_KX, B_KX>), dim3(nwg), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lab2_kernel<A_KX, B_KX>), dim3(nwg), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-44s M=%6d N=%5d K=%5d wg=%5d  %8.3f ms  %7.1f TF\n", name, M, N, K, nwg, ms, 2.0 * M * N * K / ms / 1e9);
  return ms;
}

// max |C1 - C2| over a strided sample (both on the device)
__global__ void diff_kernel(const float *a, const float *b, size_t n, float *out) {
  float m = 0.f;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(a[i] - b[i]));
  atomicMax((int *)out, __float_as_int(m));
}

template <int FLAGS, int OCC>
void run(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  const int tiles_n = N / BN, nwg = (M / BM) * tiles_n;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_kernel<FLAGS, OCC>), dim3(nwg), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lab_kernel<FLAGS, OCC>), dim3(nwg), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-44s M=%6d N=%5d K=%5d wg=%5d  %8.3f ms  %7.1f TF\n", name, M, N, K, nwg, ms, 2.0 * M * N * K / ms / 1e9);
}

extern "C" int tv_gemm_f32(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
                           int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc,
                           const float *bias, const float *row_scale, int accumulate, void *workspace,
                           int64_t workspace_bytes, void *stream);
extern "C" int64_t tv_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k);

// the production kernel on the same buffers (NT)
void run_lib(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  const int64_t wsb = tv_gemm_f32_workspace_bytes(M, N, K);
  void *ws = nullptr;
  if (wsb) CHECK(hipMalloc(&ws, wsb));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) tv_gemm_f32(0, 1, M, N, K, A, K, B, K, C, N, nullptr, nullptr, 0, ws, wsb, nullptr);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) tv_gemm_f32(0, 1, M, N, K, A, K, B, K, C, N, nullptr, nullptr, 0, ws, wsb, nullptr);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-44s M=%6d N=%5d K=%5d ws=%4lld MB %8.3f ms  %7.1f TF\n", name, M, N, K, (long long)(wsb >> 20), ms, 2.0 * M * N * K / ms / 1e9);
  if (ws) CHECK(hipFree(ws));
}

int main() {
  const int shapes[][3] = {{51200, 1024, 1024}, {4096, 4096, 4096}, {49152, 1024, 1024}, {1024, 1024, 51200}};
  for (auto &s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    float *A, *B, *C, *C2, *d;
    CHECK(hipMalloc(&A, (size_t)M * K * 4)); CHECK(hipMalloc(&B, (size_t)N * K * 4));
    CHECK(hipMalloc(&C, (size_t)M * N * 4)); CHECK(hipMalloc(&C2, (size_t)M * N * 4)); CHECK(hipMalloc(&d, 4));
    std::vector<float> h((size_t)(M > N ? M : N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CHECK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, h.data() + 17, (size_t)N * K * 4 - 68, hipMemcpyHostToDevice));
    run<0, 2>("v1 baseline NT", A, B, C, M, N, K);
    run2<false, false>("v2 NT (A [M][K], B [N][K])", A, B, C2, M, N, K);
    run_lib("library tv_gemm_f32 NT", A, B, C2, M, N, K);
    CHECK(hipMemset(d, 0, 4));
    hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, C, C2, (size_t)M * N, d);
    float hd; CHECK(hipMemcpy(&hd, d, 4, hipMemcpyDeviceToHost));
    printf("    max |v1 - v2| = %g\n", hd);
    // the other storage combinations reuse the same buffers as differently-shaped operands
    // (timing only; M, N, K are all multiples of 128 so every interpretation is in bounds)
    run2<false, true>("v2 NN (A [M][K], B [K][N])", A, B, C2, M, N, K);
    run2<true, true>("v2 TN (A [K][M], B [K][N])", A, B, C2, M, N, K);
    CHECK(hipFree(A)); CHECK(hipFree(B)); CHECK(hipFree(C)); CHECK(hipFree(C2)); CHECK(hipFree(d));
  }
  return 0;
}
