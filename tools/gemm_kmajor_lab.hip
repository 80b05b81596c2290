// LAB (not part of the library): the weight-gradient product C[M,N] = A^T B with BOTH operands
// k-major ([K][M] and [K][N] fp32 rows: the gradient and the activation of a layer), on the
// structure of gemm_wide_kernel: eight waves that all multiply, 128 x 256 tile, split-K units.
//
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -o tools/gemm_kmajor_lab tools/gemm_kmajor_lab.hip
//   tools/gemm_kmajor_lab [M N K splits reps]
//
// What is different from the row-major case: a thread that loads 16 bytes along x holds ONE k
// of four x, but a matrix-instruction operand wants eight k of one x.  Here the planes stay
// k-major in LDS too ([k][x] bf16 rows of 256 bytes, chunk index XOR-ed with the row as the
// MI355X guide's dual-use image (b)), each thread stores its four bf16 of a plane with one
// ds_write_b64, and the fragments are read with the transposing read of gfx950
// (ds_read_b64_tr_b16: a 4-row x 16-column block per 16 lanes, delivered column-major).
// The library's producer / consumer kernel reads these operands with 4-byte loads instead
// (32 per thread and k-tile) and runs them at 163-167 TF.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef PIECES
#define PIECES 1
#endif
constexpr int BM = 128, BN = 256, BK = 32;
constexpr int PLANE = BK * 256;                  // one bf16 plane of a 128-column image: 32 rows x 256 B
constexpr int IMAGE = 3 * PLANE;                 // 24 KB
constexpr int BUF = 3 * IMAGE;                   // A image | two B images

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned img_off(int row, int ch) {      // byte offset of 16-byte chunk ch of row
  return (unsigned)(256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))));
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// four x of one k -> 8 bytes in each of the three planes
__device__ __forceinline__ void split4_store(const f32x4 &x, char *img, int k, int xcol) {
  const unsigned p01 = pack_bf16(x[0], x[1]), p23 = pack_bf16(x[2], x[3]);
  const float e0 = x[0] - __uint_as_float(p01 << 16), e1 = x[1] - __uint_as_float(p01 & 0xffff0000u);
  const float e2 = x[2] - __uint_as_float(p23 << 16), e3 = x[3] - __uint_as_float(p23 & 0xffff0000u);
  const unsigned q01 = pack_bf16(e0, e1), q23 = pack_bf16(e2, e3);
  const float f0 = e0 - __uint_as_float(q01 << 16), f1 = e1 - __uint_as_float(q01 & 0xffff0000u);
  const float f2 = e2 - __uint_as_float(q23 << 16), f3 = e3 - __uint_as_float(q23 & 0xffff0000u);
  char *dst = img + img_off(k, xcol >> 3) + 2 * (xcol & 7);
  *reinterpret_cast<uint2 *>(dst) = make_uint2(p01, p23);
  *reinterpret_cast<uint2 *>(dst + PLANE) = make_uint2(q01, q23);
  *reinterpret_cast<uint2 *>(dst + 2 * PLANE) = make_uint2(pack_bf16(f0, f1), pack_bf16(f2, f3));
}

// the three planes of a 32-x sub-tile for one 16-deep k-step, transposed on the way out of LDS:
// lane l gets x = xbase + (l & 31), k = 16 step + 8 (l >> 5) .. +7
__device__ __forceinline__ void read_set_tr(const char *img, int xbase, int lane, int step, bf16x8 (&f)[3]) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int c0 = xbase + 16 * (g & 1);
  const int ch = (c0 >> 3) + (p >> 1);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    s16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = 16 * step + 8 * (g >> 1) + 4 * hh + q;
      const char *a = img + pl * PLANE + img_off(row, ch) + 8 * (p & 1);
      h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)a);
    }
    const s16x8 v = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
    f[pl] = __builtin_bit_cast(bf16x8, v);
  }
}
__device__ __forceinline__ void mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 &hi, f32x16 &lo) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
}

// A [K][lda], B [K][ldb] (M, N multiples of the tile here; K / splits a multiple of 32);
// slabs [splits][M][N]
__global__ __launch_bounds__(512) void gemm_kmajor_kernel(const float *A, int64_t lda, const float *B, int64_t ldb,
                                                          float *slabs, int64_t M, int64_t N, int64_t K,
                                                          int tiles_n, int splits) {
  extern __shared__ __align__(16) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles = (int)(M / BM) * tiles_n;
  const int tile = blockIdx.x % tiles, sp = blockIdx.x / tiles;
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t kchunk = K / splits, kbeg = sp * kchunk;
  const int nt = (int)(kchunk / BK);
  // A: thread owns x = 4 (t & 31) .. +3 of rows k = t >> 5 and 16 + (t >> 5)
  // B: thread owns x = 4 (t & 63) .. +3 of rows k = (t >> 6) + 8 j, j = 0..3
  const int ax = 4 * (t & 31), ak = t >> 5;
  const int bx = 4 * (t & 63), bk = t >> 6;
  const char *a_base = reinterpret_cast<const char *>(A + kbeg * lda + m0);
  const char *b_base = reinterpret_cast<const char *>(B + kbeg * ldb + n0);
  const unsigned a_off0 = (unsigned)((ak * lda + ax) * 4), a_off1 = (unsigned)(((ak + 16) * lda + ax) * 4);
  unsigned b_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b_off[j] = (unsigned)(((bk + 8 * j) * ldb + bx) * 4);
  f32x4 ra[2], rb[4];
  auto load = [&](int kt) {
    const char *ta = a_base + (int64_t)kt * BK * lda * 4, *tb = b_base + (int64_t)kt * BK * ldb * 4;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[0]) : "v"(a_off0), "s"(ta));
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[1]) : "v"(a_off1), "s"(ta));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[j]) : "v"(b_off[j]), "s"(tb));
  };
  auto split = [&](int kt) {
    asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
    char *buf = smem + (kt & 1) * BUF;
    split4_store(ra[0], buf, ak, ax);
    split4_store(ra[1], buf, ak + 16, ax);
#pragma unroll
    for (int j = 0; j < 4; ++j) split4_store(rb[j], buf + IMAGE + (bx >> 7) * IMAGE, bk + 8 * j, bx & 127);
  };
  f32x16 hi[2][2], lo[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }
  load(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  split(0);
#if PIECES   // (0: the whole tile split in one block after the fourth block of matrix instructions: 192 TF against 196)
  // one operand register set; float4 j of tile it + 1 is split and stored after the j-th block of
  // matrix instructions of tile it, and reloaded for tile it + 2 straight away (clamped to the
  // last tile so that the count of loads in flight is the same in every iteration: 5 younger
  // ones when piece j waits)
  auto tile_ptr = [&](int kt) { return kt < nt ? kt : nt - 1; };
  load(tile_ptr(1));
  auto piece = [&](int it, int j) {
    char *buf = smem + ((it + 1) & 1) * BUF;
    const int kt = tile_ptr(it + 2);
    const char *ta = a_base + (int64_t)kt * BK * lda * 4, *tb = b_base + (int64_t)kt * BK * ldb * 4;
    if (j < 2) {
      asm volatile("s_waitcnt vmcnt(5)" : "+v"(ra[j]));
      split4_store(ra[j], buf, ak + 16 * j, ax);
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[j]) : "v"(j ? a_off1 : a_off0), "s"(ta));
    } else {
      asm volatile("s_waitcnt vmcnt(5)" : "+v"(rb[j - 2]));
      split4_store(rb[j - 2], buf + IMAGE + (bx >> 7) * IMAGE, bk + 8 * (j - 2), bx & 127);
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[j - 2]) : "v"(b_off[j - 2]), "s"(tb));
    }
  };
  for (int it = 0; it < nt; ++it) {
    __syncthreads();                                   // planes of tile it are in buffer it & 1
    const char *As = smem + (it & 1) * BUF, *Bs = As + IMAGE + (wn >> 1) * IMAGE;
    const int bn0 = (wn & 1) * 64;
    bf16x8 P[3], Q[3], R[3], S[3];
    read_set_tr(As, wm * 64, lane, 0, P);
    read_set_tr(Bs, bn0, lane, 0, Q);
    read_set_tr(Bs, bn0 + 32, lane, 0, R);
    read_set_tr(As, wm * 64 + 32, lane, 0, S);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma6(P, Q, hi[0][0], lo[0][0]);
    piece(it, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma6(P, R, hi[0][1], lo[0][1]);
    piece(it, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr(As, wm * 64, lane, 1, P);
    mma6(S, R, hi[1][1], lo[1][1]);
    piece(it, 2);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr(Bs, bn0 + 32, lane, 1, R);
    mma6(S, Q, hi[1][0], lo[1][0]);
    piece(it, 3);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr(Bs, bn0, lane, 1, Q);
    read_set_tr(As, wm * 64 + 32, lane, 1, S);
    __builtin_amdgcn_s_setprio(0);
    mma6(P, R, hi[0][1], lo[0][1]);
    piece(it, 4);
    __builtin_amdgcn_sched_barrier(0);
    mma6(P, Q, hi[0][0], lo[0][0]);
    piece(it, 5);
    __builtin_amdgcn_sched_barrier(0);
    mma6(S, Q, hi[1][0], lo[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma6(S, R, hi[1][1], lo[1][1]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  if (nt > 1) load(1);
  for (int it = 0; it < nt; ++it) {
    const bool more = it + 1 < nt;
    __syncthreads();                                   // planes of tile it are in buffer it & 1
    const char *As = smem + (it & 1) * BUF, *Bs = As + IMAGE + (wn >> 1) * IMAGE;
    const int bn0 = (wn & 1) * 64;
    bf16x8 P[3], Q[3], R[3], S[3];
    read_set_tr(As, wm * 64, lane, 0, P);
    read_set_tr(Bs, bn0, lane, 0, Q);
    read_set_tr(Bs, bn0 + 32, lane, 0, R);
    read_set_tr(As, wm * 64 + 32, lane, 0, S);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma6(P, Q, hi[0][0], lo[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma6(P, R, hi[0][1], lo[0][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr(As, wm * 64, lane, 1, P);
    mma6(S, R, hi[1][1], lo[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr(Bs, bn0 + 32, lane, 1, R);
    mma6(S, Q, hi[1][0], lo[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr(Bs, bn0, lane, 1, Q);
    read_set_tr(As, wm * 64 + 32, lane, 1, S);
    __builtin_amdgcn_s_setprio(0);
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      split(it + 1);
      if (it + 2 < nt) load(it + 2);
    }
    mma6(P, R, hi[0][1], lo[0][1]);
    __builtin_amdgcn_sched_barrier(0);
    mma6(P, Q, hi[0][0], lo[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma6(S, Q, hi[1][0], lo[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma6(S, R, hi[1][1], lo[1][1]);
  }
#endif
  const int kh = lane >> 5, xl = lane & 31;
  float *out = slabs + (int64_t)sp * M * N + (m0 + wm * 64 + 4 * kh) * N + n0 + wn * 64 + xl;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = hi[i][j][r] + lo[i][j][r];
}

int main(int argc, char **argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 1024, N = argc > 2 ? atoll(argv[2]) : 1024;
  const int64_t K = argc > 3 ? atoll(argv[3]) : 51200;
  const int splits = argc > 4 ? atoi(argv[4]) : 8, reps = argc > 5 ? atoi(argv[5]) : 50;
  if (M % BM || N % BN || K % (splits * BK)) { fprintf(stderr, "sizes must be multiples of the tile / split\n"); return 1; }
  std::vector<float> ha((size_t)K * M), hb((size_t)K * N);
  srand(1);
  for (auto &v : ha) v = (float)(rand() % 20001) / 10000.f - 1.f;
  for (auto &v : hb) v = (float)(rand() % 20001) / 10000.f - 1.f;
  float *dA, *dB, *dS;
  CHECK(hipMalloc(&dA, ha.size() * 4)); CHECK(hipMalloc(&dB, hb.size() * 4));
  CHECK(hipMalloc(&dS, (size_t)splits * M * N * 4));
  CHECK(hipMemcpy(dA, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = 2 * (size_t)BUF;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kmajor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles_n = (int)(N / BN), grid = (int)(M / BM) * tiles_n * splits;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL(gemm_kmajor_kernel, dim3(grid), dim3(512), lds, 0, dA, M, dB, N, dS, M, N, K, tiles_n, splits);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(gemm_kmajor_kernel, dim3(grid), dim3(512), lds, 0, dA, M, dB, N, dS, M, N, K, tiles_n, splits);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("k-major operands, 128x256, 8 waves, %d units: %lld x %lld x %lld  %.3f ms  %.1f TF (six-term work, slabs not reduced)\n",
         grid, (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
  std::vector<float> hs((size_t)splits * M * N);
  CHECK(hipMemcpy(hs.data(), dS, hs.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int s = 0; s < 400; ++s) {
    const int64_t r = (int64_t)rand() % M, c = (int64_t)rand() % N;
    double ref = 0, scale = 0, got = 0;
    for (int64_t k = 0; k < K; ++k) { ref += (double)ha[k * M + r] * hb[k * N + c]; scale += fabs((double)ha[k * M + r] * hb[k * N + c]); }
    for (int sp = 0; sp < splits; ++sp) got += hs[(size_t)sp * M * N + r * N + c];
    worst = fmax(worst, fabs(got - ref) / scale);
  }
  printf("max |err| / sum|a||b| over 400 sampled entries: %.2e\n", worst);
  return worst < 1e-6 ? 0 : 2;
}
