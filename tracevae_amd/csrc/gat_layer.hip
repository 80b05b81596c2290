// One GATConvAgg layer of a TRAINING step as one library call each way (reference:
// model/gnn_layers.py:112-157 - GATConv, head concat, BatchNorm, LeakyReLU - in the form this
// library runs it: the BatchNorm + LeakyReLU of the layer BEFORE rides in this layer's products,
// the attention scores are columns of the fc product).
//
// Nothing new is computed here: the two entry points issue exactly the launches that the
// operator-level entry points (tv_bn_finalize_coeffs, tv_gat_fc_score_weight, tv_gemm_f32_ex,
// tv_gat_aggregate_fwd / _bwd, tv_gat_fc_score_weight_bwd, tv_bn_lrelu_bwd) issue when a host
// calls them one by one, in the same order and with the same arguments, so the results are
// bit-equal.  What goes away is the host's share: at the reference's batch size of 64 traces a
// step is ~460 launches of a few microseconds each, and a Python host that crosses its FFI once
// per launch (plus an autograd node and a handful of allocations per operator) needs longer to
// issue them than the GPU needs to run them.  Per layer: 4-5 crossings forward and 6-7 backward
// become one each.
#include "tv_common.h"

extern "C" {
int64_t tv_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k);
int tv_gemm_f32_last_form(void);
int tv_gemm_f32_ex(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
                   int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc,
                   const float *bias, const float *row_scale, int accumulate, int act_operand,
                   const float *act_scale, const float *act_shift, float act_slope, float *a_max,
                   int a_max_state, float *b_max, int b_max_state, void *workspace,
                   int64_t workspace_bytes, void *stream);
int tv_bn_finalize_coeffs(int64_t cols, int64_t nblk, const float *partial, int64_t count,
                          float eps, float momentum, float *mean, float *rstd,
                          float *running_mean, float *running_var, const float *gamma,
                          const float *beta, float *scale, float *shift, float *act_max,
                          void *stream);
int64_t tv_bn_lrelu_bwd_workspace_bytes(int64_t rows, int64_t cols);
int tv_bn_lrelu_bwd(int64_t rows, int64_t cols, const float *x, const float *dy, const float *mean,
                    const float *rstd, const float *gamma, const float *beta, float slope,
                    const float *row_weight, int64_t count, float *dx, float *dgamma,
                    float *dbeta, void *workspace, int64_t workspace_bytes, void *stream);
int tv_gat_fc_score_weight(int64_t heads, int64_t feats, int64_t in_feats, const float *weight,
                           const float *attn_l, const float *attn_r, float *weight_ext,
                           float *weight_ext_max, void *stream);
int tv_gat_fc_score_weight_bwd(int64_t heads, int64_t feats, int64_t in_feats,
                               const float *weight, const float *attn_l, const float *attn_r,
                               float *dweight_ext, float *dattn_l, float *dattn_r, void *stream);
int tv_gat_aggregate_fwd(int64_t rows_per_rep, int64_t n_rep, int64_t heads, int64_t feats,
                         const int64_t *indptr, const int64_t *indices, const float *ft,
                         const float *el, const float *er, const float *bias, float *out,
                         float *alpha, float *stats_partial, const float *stats_row_weight,
                         const float *ep_scale, const float *ep_shift, float ep_slope,
                         int64_t ft_stride, int64_t score_stride, float *out_max, void *stream);
int64_t tv_gat_aggregate_bwd_workspace_bytes(int64_t rows_per_rep, int64_t n_rep, int64_t heads,
                                             int64_t feats, int64_t num_edges);
int tv_gat_aggregate_bwd(int64_t rows_per_rep, int64_t n_rep, int64_t heads, int64_t feats,
                         int64_t num_edges, const int64_t *indptr, const int64_t *indices,
                         const int64_t *rev, const float *ft, const float *el, const float *er,
                         const float *attn_l, const float *attn_r, const float *alpha,
                         const float *dout, float *dft, float *dattn_l, float *dattn_r,
                         float *dbias, int64_t ft_stride, int64_t score_stride,
                         int64_t dft_stride, int score_grad_columns, float *dft_max,
                         void *workspace, int64_t workspace_bytes, void *stream);
}

namespace {
inline int64_t pad4(int64_t v) { return (v + 3) & ~3ll; }
inline int64_t pad256(int64_t v) { return (v + 255) & ~255ll; }
inline int64_t max2(int64_t a, int64_t b) { return a > b ? a : b; }

// workspace of the calls made inside a layer call (they run one after the other and share it)
int64_t inner_bytes(int64_t rows, int64_t in_feats, int64_t width, int64_t heads, int64_t feats,
                    int64_t num_edges) {
  int64_t b = tv_gemm_f32_workspace_bytes(rows, width, in_feats);               // forward product
  b = max2(b, tv_gemm_f32_workspace_bytes(width, in_feats, rows));              // weight gradient
  b = max2(b, tv_gemm_f32_workspace_bytes(rows, in_feats, width));              // input gradient
  b = max2(b, tv_gat_aggregate_bwd_workspace_bytes(rows, 1, heads, feats, num_edges));
  b = max2(b, tv_bn_lrelu_bwd_workspace_bytes(rows, in_feats));
  return pad256(b);
}

// workspace of the backward call: [dft | dy | maxima of dft | inner]; the forward call uses the
// first `inner` bytes of a buffer of the same size
struct Layout {
  int64_t dy, maxima, inner, total;
};
Layout layout(int64_t rows, int64_t in_feats, int64_t heads, int64_t feats, int64_t num_edges) {
  const int64_t width = heads * feats + pad4(2 * heads);
  Layout l;
  l.dy = pad256(rows * width * 4);
  l.maxima = l.dy + pad256(rows * pad4(in_feats) * 4);
  l.inner = l.maxima + 1024;
  l.total = l.inner + inner_bytes(rows, in_feats, width, heads, feats, num_edges);
  return l;
}
}  // namespace

extern "C" int64_t tv_gat_layer_workspace_bytes(int64_t rows, int64_t in_feats, int64_t heads,
                                                int64_t feats, int64_t num_edges) {
  if (rows <= 0 || in_feats <= 0 || heads <= 0 || feats <= 0) return 0;
  return layout(rows, in_feats, heads, feats, num_edges).total;
}

extern "C" int tv_gat_layer_train_fwd(
    int64_t rows, int64_t in_feats, int64_t heads, int64_t feats, const int64_t *indptr,
    const int64_t *indices, const float *x, int64_t x_stride, const float *in_stats,
    int64_t in_nblk, int64_t in_count, float eps, float momentum, float slope,
    const float *in_gamma, const float *in_beta, float *in_running_mean, float *in_running_var,
    float *in_coeffs, const float *weight, const float *attn_l, const float *attn_r,
    const float *bias, const float *stats_row_weight, float *weight_ext, float *ft, float *out,
    float *alpha, float *out_stats, float *x_max, int x_max_state, float *w_max, int w_prepared,
    void *workspace, int64_t workspace_bytes, void *stream) {
  TV_REQUIRE(rows >= 0 && in_feats >= 1 && heads >= 1 && feats >= 1, "tv_gat_layer_train_fwd: bad sizes");
  TV_REQUIRE(w_prepared == 0 || w_prepared == 3 || w_prepared == 4,
             "tv_gat_layer_train_fwd: w_prepared must be 0, 3 or 4 (tv_gemm_f32_ex states of B)");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && weight && attn_l && attn_r && bias && weight_ext && ft && out && x_max && w_max,
             "tv_gat_layer_train_fwd: null pointer");
  const int64_t hf = heads * feats, width = hf + pad4(2 * heads);
  const bool deferred = in_coeffs != nullptr;
  const int64_t cpad = pad4(in_feats);
  const float *scale = nullptr, *shift = nullptr;
  if (deferred) {
    TV_REQUIRE(in_stats && in_gamma && in_beta, "tv_gat_layer_train_fwd: null BatchNorm input");
    float *mean = in_coeffs, *rstd = in_coeffs + cpad, *sc = in_coeffs + 2 * cpad, *sh = in_coeffs + 3 * cpad;
    const bool bound = in_feats <= 2048;
    if (int rc = tv_bn_finalize_coeffs(in_feats, in_nblk, in_stats, in_count, eps, momentum, mean, rstd,
                                       in_running_mean, in_running_var, in_gamma, in_beta, sc, sh,
                                       bound ? x_max : nullptr, stream))
      return rc;
    x_max_state = bound ? 2 : 1;
    scale = sc; shift = sh;
  }
  // the extended weight: written here, or handed over as the plane image that a call for the whole
  // stack prepared (tv_gat_fc_score_weight_batch + tv_gemm_prepare_b_batch)
  if (!w_prepared)
    if (int rc = tv_gat_fc_score_weight(heads, feats, in_feats, weight, attn_l, attn_r, weight_ext, w_max, stream))
      return rc;
  if (int rc = tv_gemm_f32_ex(0, 1, rows, width, in_feats, x, x_stride, weight_ext, in_feats, ft, width,
                              nullptr, nullptr, 0, deferred ? 1 : 0, scale, shift, slope, x_max,
                              x_max_state, w_max, w_prepared ? w_prepared : 2, workspace, workspace_bytes,
                              stream))
    return rc;
  if (int rc = tv_gat_aggregate_fwd(rows, 1, heads, feats, indptr, indices, ft, ft + hf, ft + hf + heads,
                                    bias, out, alpha, out_stats, out_stats ? stats_row_weight : nullptr,
                                    nullptr, nullptr, slope, width, width, nullptr, stream))
    return rc;
  return TV_OK;
}

extern "C" int tv_gat_layer_train_bwd(
    int64_t rows, int64_t in_feats, int64_t heads, int64_t feats, int64_t num_edges,
    const int64_t *indptr, const int64_t *indices, const int64_t *rev, const float *x,
    int64_t x_stride, const float *in_coeffs, const float *in_gamma, const float *in_beta,
    float slope, const float *in_row_weight, int64_t in_count, const float *weight,
    const float *attn_l, const float *attn_r, const float *weight_ext, const float *ft,
    const float *alpha, const float *dout, float *x_max, int x_max_state, float *w_max,
    int w_max_state, int need_dx, int need_dw, float *dweight_ext, float *dattn_l, float *dattn_r,
    float *dbias, float *dx, float *dgamma, float *dbeta, void *workspace,
    int64_t workspace_bytes, void *stream) {
  TV_REQUIRE(rows >= 0 && in_feats >= 1 && heads >= 1 && feats >= 1, "tv_gat_layer_train_bwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && weight && attn_l && attn_r && weight_ext && ft && alpha && dout && dbias && x_max && w_max,
             "tv_gat_layer_train_bwd: null pointer");
  const int64_t hf = heads * feats, width = hf + pad4(2 * heads), cpad = pad4(in_feats);
  const bool deferred = in_coeffs != nullptr;
  TV_REQUIRE(!deferred || x_stride == in_feats, "tv_gat_layer_train_bwd: x must be dense behind a BatchNorm");
  const Layout l = layout(rows, in_feats, heads, feats, num_edges);
  if (!workspace || workspace_bytes < l.total) {
    tv_set_error("tv_gat_layer_train_bwd: workspace of %lld bytes needed, %lld given", (long long)l.total,
                 (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  TV_REQUIRE(((uintptr_t)workspace & 255) == 0, "tv_gat_layer_train_bwd: workspace must be 256-byte aligned");
  char *ws = (char *)workspace;
  float *dft = (float *)ws;                       // [dft | del | der | 0] rows
  float *dy = (float *)(ws + l.dy);               // gradient behind the input's BatchNorm
  float *dft_max = (float *)(ws + l.maxima);      // maxima of the dft rows, for its two products
  char *inner = ws + l.inner;
  const int64_t inner_b = workspace_bytes - l.inner;
  if (int rc = tv_gat_aggregate_bwd(rows, 1, heads, feats, num_edges, indptr, indices, rev, ft, ft + hf,
                                    ft + hf + heads, nullptr, nullptr, alpha, dout, dft, nullptr, nullptr,
                                    dbias, width, width, width, 1, dft_max, inner, inner_b, stream))
    return rc;
  const float *scale = deferred ? in_coeffs + 2 * cpad : nullptr;
  const float *shift = deferred ? in_coeffs + 3 * cpad : nullptr;
  if (need_dw) {
    TV_REQUIRE(dweight_ext && dattn_l && dattn_r, "tv_gat_layer_train_bwd: null weight gradient");
    if (int rc = tv_gemm_f32_ex(1, 0, width, in_feats, rows, dft, width, x, x_stride, dweight_ext, in_feats,
                                nullptr, nullptr, 0, deferred ? 2 : 0, scale, shift, slope, dft_max, 2,
                                x_max, x_max_state, inner, inner_b, stream))
      return rc;
    if (int rc = tv_gat_fc_score_weight_bwd(heads, feats, in_feats, weight, attn_l, attn_r, dweight_ext,
                                            dattn_l, dattn_r, stream))
      return rc;
  }
  if (need_dx) {
    TV_REQUIRE(dx, "tv_gat_layer_train_bwd: null dx");
    float *target = deferred ? dy : dx;
    if (int rc = tv_gemm_f32_ex(0, 0, rows, in_feats, width, dft, width, weight_ext, in_feats, target,
                                in_feats, nullptr, nullptr, 0, 0, nullptr, nullptr, 0.f, dft_max, 2, w_max,
                                w_max_state, inner, inner_b, stream))
      return rc;
    if (deferred) {
      TV_REQUIRE(dgamma && dbeta && in_gamma && in_beta, "tv_gat_layer_train_bwd: null BatchNorm gradient");
      if (int rc = tv_bn_lrelu_bwd(rows, in_feats, x, dy, in_coeffs, in_coeffs + cpad, in_gamma, in_beta, slope,
                                   in_row_weight, in_count, dx, dgamma, dbeta, inner, inner_b, stream))
        return rc;
    }
  }
  return TV_OK;
}
