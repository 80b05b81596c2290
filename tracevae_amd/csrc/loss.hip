// The scalar end of the training step (train.py:229-270): from the log-probability vectors of
// the chain to (loss, recons, kl), and back.  Arithmetic-wise nothing; as separate tensor ops it
// is ~90 launches of a few microseconds per step, here it is one kernel each way.
//
//   recons = mean_b sum_{x observed} log p(x)                          (train.py:229-236)
//   kl     = mean_b [sum_latent log q - sum_latent log p]              (train.py:237-242)
//   loss   = beta * kl - recons + l2_reg * l2 + sum_t reg_t * unit_ball(z_t)   (:245-270)
//   unit_ball(z) = mean_b m_b^2 + mean_b (v_b - 1)^2, (m_b, v_b) = moments of z[b, :]
//                                                        (tensor_utils.py:117-125)
// One workgroup: B is a mini-batch (hundreds of traces); sums in double, fixed order.
#include "tv_common.h"

namespace {

constexpr int LOSS_THREADS = 256;

struct LossArgs {
  int64_t B;
  int n_vec, n_z;
  const float *vec[TV_LOSS_MAX_VECTORS];
  int kind[TV_LOSS_MAX_VECTORS];      // 0: observed (recons), 1: log q, 2: latent log p
  const float *z[TV_LOSS_MAX_LATENTS];
  int64_t zdim[TV_LOSS_MAX_LATENTS];
  float reg[TV_LOSS_MAX_LATENTS];
  float *dvec[TV_LOSS_MAX_VECTORS];   // backward outputs (nullable)
  float *dz[TV_LOSS_MAX_LATENTS];
  const float *l2;                    // device scalar (nullable)
  float l2_reg, beta;
  const float *dloss;                 // device scalar
  float *out;                         // [3] loss, recons, kl
};

__device__ __forceinline__ void row_moments(const float *row, int64_t d, float &m, float &v) {
  float s = 0.f, s2 = 0.f;
  for (int64_t j = 0; j < d; ++j) {
    const float x = row[j];
    s += x;
    s2 = fmaf(x, x, s2);
  }
  m = s / (float)d;
  v = s2 / (float)d - m * m;
}

__global__ __launch_bounds__(LOSS_THREADS) void train_loss_fwd_kernel(LossArgs a) {
  __shared__ double red[LOSS_THREADS / 64][3 + TV_LOSS_MAX_LATENTS];
  double acc[3 + TV_LOSS_MAX_LATENTS];
#pragma unroll
  for (int i = 0; i < 3 + TV_LOSS_MAX_LATENTS; ++i) acc[i] = 0.0;
  for (int64_t b = threadIdx.x; b < a.B; b += LOSS_THREADS) {
    for (int i = 0; i < a.n_vec; ++i) acc[a.kind[i]] += (double)a.vec[i][b];
    for (int t = 0; t < a.n_z; ++t) {
      float m, v;
      row_moments(a.z[t] + b * a.zdim[t], a.zdim[t], m, v);
      acc[3 + t] += (double)(m * m + (v - 1.f) * (v - 1.f));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 3 + TV_LOSS_MAX_LATENTS; ++i) {
    const double s = tv_wave_sum_d(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[3 + TV_LOSS_MAX_LATENTS];
    for (int i = 0; i < 3 + TV_LOSS_MAX_LATENTS; ++i) {
      tot[i] = 0.0;
      for (int w = 0; w < LOSS_THREADS / 64; ++w) tot[i] += red[w][i];
    }
    const double n = (double)a.B;
    const float recons = (float)(tot[0] / n);
    const float kl = (float)((tot[1] - tot[2]) / n);
    float loss = a.beta * kl - recons;
    if (a.l2) loss += a.l2_reg * a.l2[0];
    for (int t = 0; t < a.n_z; ++t) loss += a.reg[t] * (float)(tot[3 + t] / n);
    a.out[0] = loss;
    a.out[1] = recons;
    a.out[2] = kl;
  }
}

__global__ __launch_bounds__(LOSS_THREADS) void train_loss_bwd_kernel(LossArgs a) {
  const float g = a.dloss[0] / (float)a.B;
  for (int64_t b = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; b < a.B;
       b += (int64_t)gridDim.x * LOSS_THREADS) {
    for (int i = 0; i < a.n_vec; ++i)
      if (a.dvec[i]) a.dvec[i][b] = a.kind[i] == 0 ? -g : (a.kind[i] == 1 ? a.beta * g : -a.beta * g);
    for (int t = 0; t < a.n_z; ++t) {
      if (!a.dz[t]) continue;
      const int64_t d = a.zdim[t];
      const float *row = a.z[t] + b * d;
      float m, v;
      row_moments(row, d, m, v);
      // d/dz_j [m^2 + (v-1)^2] = 2m/d + 2(v-1) * (2 z_j/d - 2m/d)
      const float c0 = a.reg[t] * g / (float)d;
      for (int64_t j = 0; j < d; ++j)
        a.dz[t][b * d + j] = c0 * (2.f * m + 4.f * (v - 1.f) * (row[j] - m));
    }
  }
}

int fill(LossArgs &a, const char *name, int64_t batch, int n_vec, const float *const *vecs,
         const int *kinds, int n_z, const float *const *zs, const int64_t *z_dims,
         const float *z_regs) {
  TV_REQUIRE(batch >= 1, "%s: batch must be positive", name);
  TV_REQUIRE(n_vec >= 1 && n_vec <= TV_LOSS_MAX_VECTORS, "%s: 1..%d log-prob vectors, got %d", name,
             TV_LOSS_MAX_VECTORS, n_vec);
  TV_REQUIRE(n_z >= 0 && n_z <= TV_LOSS_MAX_LATENTS, "%s: at most %d regularised latents", name,
             TV_LOSS_MAX_LATENTS);
  TV_REQUIRE(vecs && kinds && (n_z == 0 || (zs && z_dims && z_regs)), "%s: null table", name);
  a.B = batch; a.n_vec = n_vec; a.n_z = n_z;
  for (int i = 0; i < n_vec; ++i) {
    TV_REQUIRE(vecs[i] && kinds[i] >= 0 && kinds[i] <= 2, "%s: bad vector %d", name, i);
    a.vec[i] = vecs[i];
    a.kind[i] = kinds[i];
  }
  for (int t = 0; t < n_z; ++t) {
    TV_REQUIRE(zs[t] && z_dims[t] >= 1, "%s: bad latent %d", name, t);
    a.z[t] = zs[t];
    a.zdim[t] = z_dims[t];
    a.reg[t] = z_regs[t];
  }
  return TV_OK;
}

}  // namespace

extern "C" int tv_train_loss_fwd(int64_t batch, int n_vec, const float *const *vecs,
                                 const int *kinds, int n_z, const float *const *zs,
                                 const int64_t *z_dims, const float *z_regs, const float *l2_value,
                                 float l2_reg, float beta, float *out3, void *stream) {
  LossArgs a = {};
  if (int rc = fill(a, "tv_train_loss_fwd", batch, n_vec, vecs, kinds, n_z, zs, z_dims, z_regs))
    return rc;
  TV_REQUIRE(out3, "tv_train_loss_fwd: null output");
  a.l2 = l2_value; a.l2_reg = l2_reg; a.beta = beta; a.out = out3;
  hipLaunchKernelGGL(train_loss_fwd_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
  TV_LAUNCH_CHECK("tv_train_loss_fwd");
  return TV_OK;
}

extern "C" int tv_train_loss_bwd(int64_t batch, int n_vec, const float *const *vecs,
                                 const int *kinds, float *const *dvecs, int n_z,
                                 const float *const *zs, const int64_t *z_dims,
                                 const float *z_regs, float *const *dzs, float beta,
                                 const float *dloss, void *stream) {
  LossArgs a = {};
  if (int rc = fill(a, "tv_train_loss_bwd", batch, n_vec, vecs, kinds, n_z, zs, z_dims, z_regs))
    return rc;
  TV_REQUIRE(dvecs && dloss && (n_z == 0 || dzs), "tv_train_loss_bwd: null table");
  for (int i = 0; i < n_vec; ++i) a.dvec[i] = dvecs[i];
  for (int t = 0; t < n_z; ++t) a.dz[t] = dzs[t];
  a.beta = beta; a.dloss = dloss;
  const int64_t blocks = tv_cdiv(batch, LOSS_THREADS);
  hipLaunchKernelGGL(train_loss_bwd_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)),
                     dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
  TV_LAUNCH_CHECK("tv_train_loss_bwd");
  return TV_OK;
}
