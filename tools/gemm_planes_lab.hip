// LAB (not part of the library): what would the six-term GEMM reach if BOTH operands arrived
// pre-split, as bf16 plane images in memory, and every wave multiplied?
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/gemm_planes_lab tools/gemm_planes_lab.hip
//   tools/gemm_planes_lab [M N K reps]
//
// Structure under test (MI355X guide, "glds vs register staging"): one 256-thread workgroup per
// 128x128 output tile, four waves that ALL multiply (2x2, 64x64 each, the same 48 matrix
// instructions and 24 ds_read_b128 per 32-deep k-tile as csrc/gemm_f32.hip), no producer waves:
// the tile images (three bf16 planes per operand, the LDS layout of csrc/gemm_f32.hip, 24 KB per
// operand and k-tile) are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
// registers, no ds_write, no vector ALU), three buffers, counted vmcnt, one raw barrier per
// k-tile.  The split itself is done ahead by a separate kernel and is NOT in the timed region.
// Output is checked against an fp64 product on a sample of entries.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <type_traits>
#include <utility>

#ifndef BB_EXP
#define BB_EXP 0   // 1: no barrier, 2: no DMA inside the k-loop (timing only, results wrong)
#endif
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE_BYTES = BM * 64;
constexpr int OPERAND_BYTES = 3 * PLANE_BYTES;     // 24 KB

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned plane_offset(int row, int piece) {
  return (unsigned)(row * 64 + ((((piece >> 1) ^ (row >> 2)) & 3) << 4) + ((piece & 1) << 3));
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// X [rows][K] fp32 -> images [row tile][k tile][3 planes], zero outside
__global__ __launch_bounds__(256) void split_kernel(const float *X, int64_t ld, int64_t R, int64_t K,
                                                    int64_t ktiles, char *planes) {
  const int t = threadIdx.x;
  const int64_t tr = blockIdx.x / ktiles, kt = blockIdx.x % ktiles;
  char *image = planes + (int64_t)blockIdx.x * OPERAND_BYTES;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = (t >> 3) + 32 * a, piece = t & 7;
    const int64_t n = tr * BM + row, k0 = kt * BK + 4 * piece;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = (n < R && k0 + j < K) ? X[n * ld + k0 + j] : 0.f;
    const unsigned p01 = pack_bf16(x[0], x[1]), p23 = pack_bf16(x[2], x[3]);
    const float e0 = x[0] - __uint_as_float(p01 << 16), e1 = x[1] - __uint_as_float(p01 & 0xffff0000u);
    const float e2 = x[2] - __uint_as_float(p23 << 16), e3 = x[3] - __uint_as_float(p23 & 0xffff0000u);
    const unsigned q01 = pack_bf16(e0, e1), q23 = pack_bf16(e2, e3);
    const float f0 = e0 - __uint_as_float(q01 << 16), f1 = e1 - __uint_as_float(q01 & 0xffff0000u);
    const float f2 = e2 - __uint_as_float(q23 << 16), f3 = e3 - __uint_as_float(q23 & 0xffff0000u);
    const unsigned s01 = pack_bf16(f0, f1), s23 = pack_bf16(f2, f3);
    char *dst = image + plane_offset(row, piece);
    *reinterpret_cast<uint2 *>(dst) = make_uint2(p01, p23);
    *reinterpret_cast<uint2 *>(dst + PLANE_BYTES) = make_uint2(q01, q23);
    *reinterpret_cast<uint2 *>(dst + 2 * PLANE_BYTES) = make_uint2(s01, s23);
  }
}

__device__ __forceinline__ void read_set(const char *lds, int row0, int lane, int step, bf16x8 (&f)[3]) {
  const int rowi = row0 + (lane & 31);
  const char *p = lds + plane_offset(rowi, 2 * (2 * step + (lane >> 5)));
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8 *>(p + pl * PLANE_BYTES);
}
__device__ __forceinline__ void mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 &hi, f32x16 &lo) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
}

// one k-tile of the operands global -> LDS buffer by LDS-DMA (each instruction moves 64 lanes x 16
// bytes = 1 KB, lane-linear: the images are byte copies).  NIMG images of 24 KB, NWAVES waves.
template <int NIMG, int NWAVES>
__device__ __forceinline__ void stage(const char *a_img, const char *b_img, int64_t b_stride, char *buf,
                                      int wave, int lane) {
  typedef __attribute__((address_space(3))) void lds_void;
  // every wave copies 24 / NWAVES chunks of EACH image: the image of a chunk is known at compile
  // time, the bases stay wave-uniform and the per-lane part is one 32-bit offset
  constexpr int PER = 24 / NWAVES;
  const unsigned voff = (unsigned)(wave * PER * 1024 + lane * 16);
  const unsigned loff = (unsigned)(wave * PER * 1024);
#pragma unroll
  for (int img = 0; img < NIMG; ++img) {
    const char *src = img == 0 ? a_img : b_img + (int64_t)(img - 1) * b_stride;
#pragma unroll
    for (int c = 0; c < PER; ++c)
      __builtin_amdgcn_global_load_lds((const void *)(src + voff + c * 1024),
                                       (lds_void *)(buf + img * OPERAND_BYTES + loff + c * 1024), 16, 0, 0);
  }
}

template <int PERSIST, int NWN>
__global__ __launch_bounds__(128 * NWN) void gemm_planes_kernel(const char *A, const char *B, float *C, int64_t M,
                                                          int64_t N, int64_t ldc, int tiles_m, int tiles_n,
                                                          int ktiles) {
  extern __shared__ __align__(16) char smem[];        // NBUF x (A image | B image)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int NB = NWN / 2;                         // 128-column B images per tile
  constexpr int NBUFS = NWN == 2 ? 3 : 2;             // LDS buffers of (1 + NB) images
  constexpr int BUF_BYTES = (1 + NB) * OPERAND_BYTES;
  const int wm = wave / NWN, wn = wave % NWN;
  tiles_n /= NB;
  const int ntiles = tiles_m * tiles_n;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // XCD-aware order: ids equal mod 8 walk contiguous tiles, n fastest (ntiles % 8 == 0 here)
    const int per = ntiles >> 3;
    const int L = (tile & 7) * per + (tile >> 3);
    const int tn = L % tiles_n, tm = L / tiles_n;
    const char *a_img = A + (int64_t)tm * ktiles * OPERAND_BYTES;
    const char *b_img = B + (int64_t)tn * NB * ktiles * OPERAND_BYTES;
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }
    __builtin_amdgcn_s_barrier();                      // previous tile's last reads are done
    constexpr int NW = 2 * NWN, PER_WAVE = (1 + NB) * 24 / NW;
    const int64_t bs = (int64_t)ktiles * OPERAND_BYTES;
    stage<1 + NB, NW>(a_img, b_img, bs, smem, wave, lane);
    if (NBUFS == 3 && ktiles > 1)
      stage<1 + NB, NW>(a_img + OPERAND_BYTES, b_img + OPERAND_BYTES, bs, smem + BUF_BYTES, wave, lane);
    for (int it = 0; it < ktiles; ++it) {
      // tile it has landed (this wave's 12 DMAs of it); at most tile it+1 stays in flight
      if (NBUFS == 3 && it + 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // ... of every wave; the buffer read last is free
      constexpr int AHEAD = NBUFS - 1;
      if (it + AHEAD < ktiles)
        stage<1 + NB, NW>(a_img + (int64_t)(it + AHEAD) * OPERAND_BYTES, b_img + (int64_t)(it + AHEAD) * OPERAND_BYTES,
                          bs, smem + ((it + AHEAD) % NBUFS) * BUF_BYTES, wave, lane);
      const char *As = smem + (it % NBUFS) * BUF_BYTES, *Bs = As + OPERAND_BYTES + (wn >> 1) * OPERAND_BYTES;
      bf16x8 P[3], Q[3], R[3], S[3];
      read_set(As, wm * 64, lane, 0, P);
      read_set(Bs, (wn & 1) * 64, lane, 0, Q);
      read_set(Bs, (wn & 1) * 64 + 32, lane, 0, R);
      read_set(As, wm * 64 + 32, lane, 0, S);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, Q, hi[0][0], lo[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, R, hi[0][1], lo[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(As, wm * 64, lane, 1, P);
      mma6(S, R, hi[1][1], lo[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(Bs, (wn & 1) * 64 + 32, lane, 1, R);
      mma6(S, Q, hi[1][0], lo[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(Bs, (wn & 1) * 64, lane, 1, Q);
      read_set(As, wm * 64 + 32, lane, 1, S);
      mma6(P, R, hi[0][1], lo[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, Q, hi[0][0], lo[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(S, Q, hi[1][0], lo[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(S, R, hi[1][1], lo[1][1]);
    }
    // plain epilogue (the lab measures the main loop; the product kernel stages and defers this)
    const int kh = lane >> 5, xl = lane & 31;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN * NB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wn * 64 + j * 32 + xl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 64 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (row < M && col < N) C[row * ldc + col] = hi[i][j][r] + lo[i][j][r];
        }
      }
  }
}


// the two B images of a 256-column tile: 48 chunks of 1 KB over 8 waves, 3 per image and wave
__device__ __forceinline__ void stage_b(const char *b_img, int64_t b_stride, char *buf, int wave, int lane) {
  typedef __attribute__((address_space(3))) void lds_void;
  const unsigned voff = (unsigned)(wave * 3 * 1024 + lane * 16), loff = (unsigned)(wave * 3 * 1024);
#pragma unroll
  for (int img = 0; img < 2; ++img)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      __builtin_amdgcn_global_load_lds((const void *)(b_img + (int64_t)img * b_stride + voff + c * 1024),
                                       (lds_void *)(buf + img * OPERAND_BYTES + loff + c * 1024), 16, 0, 0);
}

// Third form: B as ready-made plane images by LDS-DMA, A as fp32 rows split BY THE EIGHT
// MULTIPLYING WAVES THEMSELVES between their matrix instructions (8 elements per thread and
// k-tile: two 16-byte loads, ~44 vector instructions, three ds_write_b128), 128x256 tile, two
// LDS buffers.  This is what the library could be rebuilt into without touching the data layout.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8_store(const f32x4 &u, const f32x4 &v, char *dst) {
  unsigned pl[3][4];
  const f32x4 *src[2] = {&u, &v};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float x0 = (*src[h])[0], x1 = (*src[h])[1], x2 = (*src[h])[2], x3 = (*src[h])[3];
    const unsigned p01 = pack_bf16(x0, x1), p23 = pack_bf16(x2, x3);
    const float e0 = x0 - __uint_as_float(p01 << 16), e1 = x1 - __uint_as_float(p01 & 0xffff0000u);
    const float e2 = x2 - __uint_as_float(p23 << 16), e3 = x3 - __uint_as_float(p23 & 0xffff0000u);
    const unsigned q01 = pack_bf16(e0, e1), q23 = pack_bf16(e2, e3);
    const float f0 = e0 - __uint_as_float(q01 << 16), f1 = e1 - __uint_as_float(q01 & 0xffff0000u);
    const float f2 = e2 - __uint_as_float(q23 << 16), f3 = e3 - __uint_as_float(q23 & 0xffff0000u);
    pl[0][2 * h] = p01; pl[0][2 * h + 1] = p23;
    pl[1][2 * h] = q01; pl[1][2 * h + 1] = q23;
    pl[2][2 * h] = pack_bf16(f0, f1); pl[2][2 * h + 1] = pack_bf16(f2, f3);
  }
  // stores the compiler does not see: beside LDS-DMA in flight it would drain the DMAs
  // (vmcnt(0)) before any ds_write of its own to the same array
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dst;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const u32x4 w = {pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
    if (p == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(w) : "memory");
    if (p == 1) asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(addr), "v"(w) : "memory");
    if (p == 2) asm volatile("ds_write_b128 %0, %1 offset:16384" ::"v"(addr), "v"(w) : "memory");
  }
}

__global__ __launch_bounds__(512) void gemm_split_a_kernel(const float *A, int64_t lda, const char *B, float *C,
                                                           int64_t M, int64_t N, int64_t ldc, int tiles_m,
                                                           int tiles_n, int ktiles) {
  extern __shared__ __align__(16) char smem[];        // 2 x (A image | 2 B images)
  constexpr int BUF_BYTES = 3 * OPERAND_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  tiles_n /= 2;
  const int ntiles = tiles_m * tiles_n;
  // A: thread owns row t>>2, k = 8*(t&3) .. +7 of every k-tile
  const int arow = t >> 2, ap = 2 * (t & 3);
  const unsigned a_lds = plane_offset(arow, ap);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int per = ntiles >> 3;
    const int L = (tile & 7) * per + (tile >> 3);
    const int tn = L % tiles_n, tm = L / tiles_n;
    const char *a_base = reinterpret_cast<const char *>(A + (int64_t)tm * BM * lda);
    const unsigned a_off = (unsigned)(((int64_t)arow * lda + 4 * ap) * 4);
    const char *b_img = B + (int64_t)tn * 2 * ktiles * OPERAND_BYTES;
    const int64_t bs = (int64_t)ktiles * OPERAND_BYTES;
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }
    f32x4 u, v;
#define LAB_LOAD_A(KT)                                                                                   \
  do {                                                                                                   \
    const char *tp = a_base + (int64_t)(KT) * (BK * 4);                                                  \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(u) : "v"(a_off), "s"(tp));                      \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(v) : "v"(a_off), "s"(tp));            \
  } while (0)
#define LAB_STAGE_B(KT, BUF)                                                                             \
  stage_b(b_img + (int64_t)(KT) * OPERAND_BYTES, bs, smem + (BUF) * BUF_BYTES + OPERAND_BYTES, wave, lane)
    __builtin_amdgcn_s_barrier();                      // previous tile's last reads are done
    LAB_STAGE_B(0, 0);
    LAB_LOAD_A(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(u), "+v"(v));
    split8_store(u, v, smem + a_lds);
    if (ktiles > 1) LAB_LOAD_A(1);
    for (int it = 0; it < ktiles; ++it) {
      const bool more = it + 1 < ktiles;
      // B(it) has landed (the two A loads of tile it+1 may still fly); own plane stores are done
      if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (more) LAB_STAGE_B(it + 1, (it + 1) & 1);
      const char *As = smem + (it & 1) * BUF_BYTES, *Bs = As + OPERAND_BYTES + (wn >> 1) * OPERAND_BYTES;
      bf16x8 P[3], Q[3], R[3], S[3];
      read_set(As, wm * 64, lane, 0, P);
      read_set(Bs, (wn & 1) * 64, lane, 0, Q);
      read_set(Bs, (wn & 1) * 64 + 32, lane, 0, R);
      read_set(As, wm * 64 + 32, lane, 0, S);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, Q, hi[0][0], lo[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, R, hi[0][1], lo[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(As, wm * 64, lane, 1, P);
      mma6(S, R, hi[1][1], lo[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(Bs, (wn & 1) * 64 + 32, lane, 1, R);
      mma6(S, Q, hi[1][0], lo[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(Bs, (wn & 1) * 64, lane, 1, Q);
      read_set(As, wm * 64 + 32, lane, 1, S);
      if (more) {
        // A(it+1): landed two half-iterations ago; split it into the other buffer, then reuse
        // its registers for the loads of A(it+2)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("" : "+v"(u), "+v"(v));
        split8_store(u, v, smem + ((it + 1) & 1) * BUF_BYTES + a_lds);
        if (it + 2 < ktiles) LAB_LOAD_A(it + 2);
      }
      mma6(P, R, hi[0][1], lo[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, Q, hi[0][0], lo[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(S, Q, hi[1][0], lo[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(S, R, hi[1][1], lo[1][1]);
    }
#undef LAB_LOAD_A
#undef LAB_STAGE_B
    const int kh = lane >> 5, xl = lane & 31;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wn * 64 + j * 32 + xl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 64 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (row < M && col < N) C[row * ldc + col] = hi[i][j][r] + lo[i][j][r];
        }
      }
  }
}

// Fourth form: ONE multiplying wave per SIMD with the whole register file (256 threads, 512
// registers per lane: 256 accumulators + two sets of fragments), a 128 x 64 block per wave, 256 x 128
// tile, both operands as ready-made plane images by LDS-DMA.  36 ds_read_b128 feed 96 matrix
// instructions per wave and k-tile (0.375 per instruction against 0.5 for the 64 x 64 blocks of the
// forms above), and the workgroup reads 144 KB of LDS per k-tile instead of 192.  There is no second
// wave to cover a wait, so the reads of one k-step are issued under the matrix instructions of the
// other, and the barrier sits between the two k-steps of a tile: by then the DMA of the next tile
// has had a whole k-step to land and this wave's reads of the current one are all issued.
__global__ __launch_bounds__(256) void gemm_big_block_kernel(const char *A, const char *B, float *C, int64_t M,
                                                             int64_t N, int64_t ldc, int tiles_m, int tiles_n,
                                                             int ktiles) {
  extern __shared__ __align__(16) char smem[];        // 2 x (B image | A image 0 | A image 1)
  constexpr int BUF_BYTES = 3 * OPERAND_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = (tiles_m / 2) * tiles_n;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int per = ntiles >> 3;
    const int L = (tile & 7) * per + (tile >> 3);
    const int tn = L % tiles_n, tm = L / tiles_n;
    const char *a_img = A + (int64_t)(2 * tm) * ktiles * OPERAND_BYTES;
    const char *b_img = B + (int64_t)tn * ktiles * OPERAND_BYTES;
    const int64_t as = (int64_t)ktiles * OPERAND_BYTES;
    f32x16 hi[4][2], lo[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }
    bf16x8 X0[4][3], Y0[2][3], X1[4][3], Y1[2][3];
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned dma_voff = (unsigned)(wave * 6 * 1024 + lane * 16), dma_loff = (unsigned)(wave * 6 * 1024);
    const unsigned rd_b[2] = {plane_offset(wn * 64 + (lane & 31), 2 * (lane >> 5)),
                              plane_offset(wn * 64 + 32 + (lane & 31), 2 * (lane >> 5))};
    // read R of a k-step: plane R % 3 of fragment set R / 3 (sets 0, 1: B columns; 2..5: A rows)
    auto one_read = [&](auto rc, const char *buf, int step, bf16x8 (&X)[4][3], bf16x8 (&Y)[2][3]) {
      constexpr int R = decltype(rc)::value, SET = R / 3, PL = R % 3;
      if constexpr (SET < 2) {
        const char *p = buf + plane_offset(wn * 64 + 32 * SET + (lane & 31), 2 * (2 * step + (lane >> 5)));
        Y[SET][PL] = *reinterpret_cast<const bf16x8 *>(p + PL * PLANE_BYTES);
      } else {
        const char *p = buf + (1 + wm) * OPERAND_BYTES +
                        plane_offset(32 * (SET - 2) + (lane & 31), 2 * (2 * step + (lane >> 5)));
        X[SET - 2][PL] = *reinterpret_cast<const bf16x8 *>(p + PL * PLANE_BYTES);
      }
    };
    // matrix instruction K of a k-step: term K % 6 of block K / 6 (blocks walk i = 0..3, j in a snake)
    auto one_mfma = [&](auto kc, bf16x8 (&X)[4][3], bf16x8 (&Y)[2][3]) {
      constexpr int K = decltype(kc)::value, BLK = K / 6, TERM = K % 6, I = BLK >> 1;
      constexpr int J = (I & 1) ? 1 - (BLK & 1) : (BLK & 1);
      constexpr int AP = TERM == 0 ? 2 : (TERM == 1 ? 0 : (TERM == 2 ? 1 : (TERM == 3 ? 1 : 0)));
      constexpr int BP = TERM == 0 ? 0 : (TERM == 1 ? 2 : (TERM == 2 ? 1 : (TERM == 3 ? 0 : (TERM == 4 ? 1 : 0))));
      if constexpr (TERM < 5) lo[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[I][AP], Y[J][BP], lo[I][J], 0, 0, 0);
      else hi[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[I][0], Y[J][0], hi[I][J], 0, 0, 0);
    };
    // DMA Q of a k-tile: chunk Q % 6 of this wave's share of image Q / 6 (B, A0, A1)
    auto one_dma = [&](auto qc, int kt, char *buf) {
      constexpr int Q = decltype(qc)::value, IMG = Q / 6, C = Q % 6;
      const char *src = (IMG == 0 ? b_img : a_img + (int64_t)(IMG - 1) * as) + (int64_t)kt * OPERAND_BYTES;
      __builtin_amdgcn_global_load_lds((const void *)(src + dma_voff + C * 1024),
                                       (lds_void *)(buf + IMG * OPERAND_BYTES + dma_loff + C * 1024), 16, 0, 0);
    };
    (void)rd_b;
    // one k-step: 48 matrix instructions on (Xm, Ym) with the 18 reads of the other k-step's
    // fragments (and, when DMA, the 18 DMAs of the tile after next) dealt out between them
    auto kstep = [&](auto dmac, bf16x8 (&Xm)[4][3], bf16x8 (&Ym)[2][3], const char *rbuf, int rstep,
                     bf16x8 (&Xr)[4][3], bf16x8 (&Yr)[2][3], int kt, char *dbuf) {
      constexpr bool DMA = decltype(dmac)::value;
      auto group = [&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        constexpr int K0 = (Q / 3) * 8 + (Q % 3) * 3;   // 3, 3, 2 matrix instructions per group
        one_read(qc, rbuf, rstep, Xr, Yr);
        one_mfma(std::integral_constant<int, K0>(), Xm, Ym);
        if constexpr (DMA && BB_EXP != 2) one_dma(qc, kt, dbuf);
        one_mfma(std::integral_constant<int, K0 + 1>(), Xm, Ym);
        if constexpr (Q % 3 != 2) one_mfma(std::integral_constant<int, K0 + 2>(), Xm, Ym);
        __builtin_amdgcn_sched_barrier(0);
      };
      group(std::integral_constant<int, 0>()); group(std::integral_constant<int, 1>());
      group(std::integral_constant<int, 2>()); group(std::integral_constant<int, 3>());
      group(std::integral_constant<int, 4>()); group(std::integral_constant<int, 5>());
      group(std::integral_constant<int, 6>()); group(std::integral_constant<int, 7>());
      group(std::integral_constant<int, 8>()); group(std::integral_constant<int, 9>());
      group(std::integral_constant<int, 10>()); group(std::integral_constant<int, 11>());
      group(std::integral_constant<int, 12>()); group(std::integral_constant<int, 13>());
      group(std::integral_constant<int, 14>()); group(std::integral_constant<int, 15>());
      group(std::integral_constant<int, 16>()); group(std::integral_constant<int, 17>());
    };
    auto frags = [&](const char *buf, int step, bf16x8 (&X)[4][3], bf16x8 (&Y)[2][3]) {
      const char *As = buf + (1 + wm) * OPERAND_BYTES;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) read_set(buf, wn * 64 + 32 * j2, lane, step, Y[j2]);
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) read_set(As, 32 * i2, lane, step, X[i2]);
    };
    auto multiply = [&](bf16x8 (&X)[4][3], bf16x8 (&Y)[2][3]) {
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) mma6(X[i2], Y[(i2 & 1) ? 1 - j2 : j2], hi[i2][(i2 & 1) ? 1 - j2 : j2], lo[i2][(i2 & 1) ? 1 - j2 : j2]);
    };
    __builtin_amdgcn_s_barrier();                      // previous tile's last reads are done
    stage<3, 4>(b_img, a_img, as, smem, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ktiles > 1) stage<3, 4>(b_img + OPERAND_BYTES, a_img + OPERAND_BYTES, as, smem + BUF_BYTES, wave, lane);
    frags(smem, 0, X0, Y0);
    for (int it = 0; it + 1 < ktiles; ++it) {          // (no branches inside: one scheduling region per k-step)
      char *cur = smem + (it & 1) * BUF_BYTES, *nxt = smem + ((it + 1) & 1) * BUF_BYTES;
      const int kt = it + 2 < ktiles ? it + 2 : ktiles - 1;   // (the last one is a repeat, never read)
      kstep(std::false_type(), X0, Y0, cur, 1, X1, Y1, 0, nullptr);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile it + 1 has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's reads of tile it are done
#if BB_EXP != 1
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      kstep(std::true_type(), X1, Y1, nxt, 0, X0, Y0, kt, cur);
    }
    {
      char *cur = smem + ((ktiles - 1) & 1) * BUF_BYTES;
      kstep(std::false_type(), X0, Y0, cur, 1, X1, Y1, 0, nullptr);
      multiply(X1, Y1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the repeated DMA, before the buffers are reused)
    }
    const int kh = lane >> 5, xl = lane & 31;
    const int64_t m0 = (int64_t)tm * 2 * BM, n0 = (int64_t)tn * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wn * 64 + j * 32 + xl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 128 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (row < M && col < N) C[row * ldc + col] = hi[i][j][r] + lo[i][j][r];
        }
      }
  }
}

int main(int argc, char **argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 51200, N = argc > 2 ? atoll(argv[2]) : 1024;
  const int64_t K = argc > 3 ? atoll(argv[3]) : 1024;
  const int reps = argc > 4 ? atoi(argv[4]) : 50;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN), ktiles = (int)((K + BK - 1) / BK);
  if ((tiles_m * tiles_n) % 16 || tiles_n % 2) { fprintf(stderr, "tile count must be a multiple of 16 and the column tiles even\n"); return 1; }
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto &v : ha) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : hb) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *dA, *dB, *dC;
  char *pA, *pB;
  CHECK(hipMalloc(&dA, ha.size() * 4)); CHECK(hipMalloc(&dB, hb.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMalloc(&pA, (size_t)tiles_m * ktiles * OPERAND_BYTES)); CHECK(hipMalloc(&pB, (size_t)tiles_n * ktiles * OPERAND_BYTES));
  CHECK(hipMemcpy(dA, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(split_kernel, dim3(tiles_m * ktiles), dim3(256), 0, 0, dA, K, M, K, (int64_t)ktiles, pA);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms_split = 0; CHECK(hipEventElapsedTime(&ms_split, e0, e1));
  hipLaunchKernelGGL(split_kernel, dim3(tiles_n * ktiles), dim3(256), 0, 0, dB, K, N, K, (int64_t)ktiles, pB);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(split_kernel, dim3(tiles_m * ktiles), dim3(256), 0, 0, dA, K, M, K, (int64_t)ktiles, pA);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  CHECK(hipEventElapsedTime(&ms_split, e0, e1));
  auto run = [&](auto kern, const char *name, int nwn, int persist) {
    const int nb = nwn / 2, nbufs = nwn == 2 ? 3 : 2;
    const size_t lds = (size_t)nbufs * (1 + nb) * OPERAND_BYTES;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = persist ? 256 : tiles_m * tiles_n / nb;
    for (int i = 0; i < 3; ++i)
      hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * nwn), lds, 0, pA, pB, dC, M, N, N, tiles_m, tiles_n, ktiles);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
      hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * nwn), lds, 0, pA, pB, dC, M, N, N, tiles_m, tiles_n, ktiles);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-44s %lld x %lld x %lld  %.3f ms  %.1f TF (six-term work)\n", name, (long long)M, (long long)N,
           (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
  };
  printf("split of A ahead of the product (not timed below): %.3f ms\n", ms_split);
  run(gemm_planes_kernel<0, 2>, "128x128, 4 waves, one workgroup per tile", 2, 0);
  run(gemm_planes_kernel<1, 2>, "128x128, 4 waves, persistent", 2, 1);
  run(gemm_planes_kernel<0, 4>, "128x256, 8 waves, one workgroup per tile", 4, 0);
  run(gemm_planes_kernel<1, 4>, "128x256, 8 waves, persistent", 4, 1);
  {
    const size_t lds = 2 * 3 * (size_t)OPERAND_BYTES;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_split_a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int persist = 0; persist < 2; ++persist) {
      const int grid = persist ? 256 : tiles_m * tiles_n / 2;
      CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
      for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(gemm_split_a_kernel, dim3(grid), dim3(512), lds, 0, dA, K, pB, dC, M, N, N, tiles_m, tiles_n, ktiles);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(gemm_split_a_kernel, dim3(grid), dim3(512), lds, 0, dA, K, pB, dC, M, N, N, tiles_m, tiles_n, ktiles);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("%-44s %lld x %lld x %lld  %.3f ms  %.1f TF (six-term work)\n",
             persist ? "128x256, 8 waves split A themselves, persist." : "128x256, 8 waves split A themselves",
             (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
    }
  }
  if (tiles_m % 2 == 0 && ((tiles_m / 2) * tiles_n) % 8 == 0) {
    const size_t lds = 2 * 3 * (size_t)OPERAND_BYTES;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_big_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int persist = 0; persist < 2; ++persist) {
      const int grid = persist ? 256 : (tiles_m / 2) * tiles_n;
      CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
      for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(gemm_big_block_kernel, dim3(grid), dim3(256), lds, 0, pA, pB, dC, M, N, N, tiles_m, tiles_n, ktiles);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(gemm_big_block_kernel, dim3(grid), dim3(256), lds, 0, pA, pB, dC, M, N, N, tiles_m, tiles_n, ktiles);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("%-44s %lld x %lld x %lld  %.3f ms  %.1f TF (six-term work)\n",
             persist ? "256x128, 4 waves of 128x64, persistent" : "256x128, 4 waves of 128x64",
             (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
    }
  }
  // check a sample of entries against float64
  std::vector<float> hc((size_t)M * N);
  CHECK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int s = 0; s < 2000; ++s) {
    const int64_t r = (int64_t)rand() % M, c = (int64_t)rand() % N;
    double ref = 0, scale = 0;
    for (int64_t k = 0; k < K; ++k) { ref += (double)ha[r * K + k] * hb[c * K + k]; scale += fabs((double)ha[r * K + k] * hb[c * K + k]); }
    worst = fmax(worst, fabs(hc[r * N + c] - ref) / scale);
  }
  printf("max |err| / sum|a||b| over 2000 sampled entries: %.2e\n", worst);
  return worst < 1e-6 ? 0 : 2;
}
