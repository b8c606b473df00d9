// torch operator surface of the hot path: namespace o2m:: (TORCH_LIBRARY), one op per launcher of
// include/o2m_hip.h.  The reference's operator API is torch.nn.functional;
// this shim is where the build's own operators enter the dispatcher: each op validates its tensors
// (TORCH_CHECK -> Python RuntimeError), takes the CURRENT HIP stream of the tensors' device under a
// device guard, and enqueues the extern "C" launcher.  libo2m_hip.so itself stays torch-free.
//
// Schema convention: tensors a launcher writes are `Tensor(a!)` arguments allocated by the caller
// (PyTorch's caching allocator owns all memory); ops return nothing, so the Meta kernels used for
// FakeTensor / torch.compile tracing are no-ops that only run the shape checks.
//
// Built with g++ (host code only) by one_to_many_gan_amd/build.py into lib/libo2m_torch.so.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <optional>
#include <vector>

#include "../../include/o2m_hip.h"

namespace {

using at::Tensor;
using OptT = std::optional<Tensor>;

struct Launch {  // device guard + the stream PyTorch is enqueueing on for that device
  c10::DeviceGuard guard;
  void* stream;
  explicit Launch(const Tensor& t)
      : guard(t.device()),
        stream(t.is_meta() ? nullptr
                           : static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream())) {}
};

inline bool is_fp8(const Tensor& t) {
  return t.scalar_type() == at::kFloat8_e4m3fn || t.scalar_type() == at::kFloat8_e5m2;
}
inline int dtype_code(const Tensor& t, const char* what) {
  if (t.scalar_type() == at::kBFloat16) return O2M_BF16;
  if (t.scalar_type() == at::kFloat) return O2M_F32;
  TORCH_CHECK(false, what, ": activation dtype must be bfloat16 or float32, got ", t.scalar_type());
}

// every tensor handed to a launcher: dense, on the GPU (or meta while tracing)
inline void chk(const Tensor& t, const char* op, const char* name) {
  TORCH_CHECK(t.is_cuda() || t.is_meta(), op, ": `", name,
              "` is not a GPU tensor (the o2m hot path has no CPU fallback)");
  TORCH_CHECK(t.is_contiguous(), op, ": `", name, "` must be contiguous");
}
inline void chk(const OptT& t, const char* op, const char* name) {
  if (t.has_value()) chk(*t, op, name);
}
inline void chk_f32(const Tensor& t, const char* op, const char* name) {
  chk(t, op, name);
  TORCH_CHECK(t.scalar_type() == at::kFloat, op, ": `", name, "` must be float32");
}
inline void chk_f32(const OptT& t, const char* op, const char* name) {
  if (t.has_value()) chk_f32(*t, op, name);
}
inline void same_dtype(const Tensor& a, const Tensor& b, const char* op, const char* na, const char* nb) {
  TORCH_CHECK(a.scalar_type() == b.scalar_type(), op, ": `", na, "` and `", nb, "` must share one dtype");
}
inline void same_dtype(const Tensor& a, const OptT& b, const char* op, const char* na, const char* nb) {
  if (b.has_value()) same_dtype(a, *b, op, na, nb);
}
template <typename T = void>
inline T* ptr(const Tensor& t) { return t.is_meta() ? nullptr : static_cast<T*>(t.data_ptr()); }
template <typename T = void>
inline T* ptr(const OptT& t) { return t.has_value() ? ptr<T>(*t) : nullptr; }
inline const float* fptr(const OptT& t) { return ptr<float>(t); }
inline int i32(int64_t v, const char* op) {
  TORCH_CHECK(v >= INT32_MIN && v <= INT32_MAX, op, ": dimension ", v, " exceeds int32");
  return static_cast<int>(v);
}
inline void done(int err, const char* op, bool meta) {
  TORCH_CHECK(meta || err == 0, op, " failed with code ", err,
              err == O2M_ERR_BAD_ARG ? " (argument rejected)" : err == O2M_ERR_UNSUPPORTED ? " (unsupported size)" : "");
}

#define O2M_CALL(op, t, expr)                      \
  do {                                             \
    Launch L(t);                                   \
    const bool meta_ = (t).is_meta();              \
    int err_ = 0;                                  \
    if (!meta_) { void* stream = L.stream; err_ = (expr); } \
    done(err_, op, meta_);                         \
  } while (0)

// ------------------------------------------------------------------------------------ conv

void conv2d_fwd(const Tensor& x, const Tensor& w, Tensor& y, const OptT& in_scale, const OptT& out_scale,
                const OptT& bias, const OptT& residual, int64_t pad, int64_t pad_mode, int64_t act,
                bool per_sample_w, int64_t stride, const std::optional<Tensor>& stats, const OptT& deq, const OptT& aux,
                const std::optional<Tensor>& aux_scaled, int64_t fold_pad) {
  const char* op = "o2m::conv2d_fwd";
  chk_f32(stats, op, "stats"); chk_f32(deq, op, "deq"); chk(aux, op, "aux"); chk(aux_scaled, op, "aux_scaled");
  const bool f8 = is_fp8(x);
  chk(x, op, "x"); chk(w, op, "w"); chk(y, op, "y"); chk(residual, op, "residual");
  chk_f32(in_scale, op, "in_scale"); chk_f32(out_scale, op, "out_scale"); chk_f32(bias, op, "bias");
  TORCH_CHECK(x.dim() == 4 && y.dim() == 4 && w.dim() == (per_sample_w ? 5 : 4), op, ": x, y are NHWC; w is ",
              per_sample_w ? "[B][Co][KH][KW][Ci]" : "[Co][KH][KW][Ci]");
  if (f8) {  // config #5: fp8 operands, bf16 result
    TORCH_CHECK(w.scalar_type() == at::kFloat8_e4m3fn && y.scalar_type() == at::kBFloat16, op,
                ": fp8 activations need float8_e4m3fn filters and a bfloat16 output");
    TORCH_CHECK(deq.has_value() && deq->numel() >= 4, op, ": fp8 operands need deq = {1/scale_x, amax_x, 1/scale_w, amax_w}");
    same_dtype(y, residual, op, "y", "residual");
  } else {
    TORCH_CHECK(!deq.has_value(), op, ": deq is for fp8 operands only");
    same_dtype(x, w, op, "x", "w"); same_dtype(x, y, op, "x", "y"); same_dtype(x, residual, op, "x", "residual");
  }
  const int wd = per_sample_w ? 1 : 0;
  const int64_t Co = w.size(wd), KH = w.size(wd + 1), KW = w.size(wd + 2);
  TORCH_CHECK(w.size(wd + 3) == x.size(3), op, ": filter has ", w.size(wd + 3), " input channels, x has ", x.size(3));
  TORCH_CHECK(!per_sample_w || w.size(0) == x.size(0), op, ": per-sample filters need one filter per sample");
  const int64_t s = stride > 1 ? stride : 1;
  const int64_t Ho = (x.size(1) + 2 * pad - KH) / s + 1, Wo = (x.size(2) + 2 * pad - KW) / s + 1;
  TORCH_CHECK(fold_pad >= 0 && 2 * fold_pad < Ho && 2 * fold_pad < Wo, op, ": fold_pad out of range");
  const int64_t Hy = Ho - 2 * fold_pad, Wy = Wo - 2 * fold_pad;  // fold_pad: y is the cropped map (o2m_conv_desc.fold_pad)
  TORCH_CHECK(y.size(0) == x.size(0) && y.size(1) == Hy && y.size(2) == Wy && y.size(3) == Co, op,
              ": y must be [", x.size(0), ", ", Hy, ", ", Wy, ", ", Co, "], got ", y.sizes());
  TORCH_CHECK(!residual.has_value() || residual->sizes() == y.sizes(), op, ": residual must have y's shape");
  TORCH_CHECK(!in_scale.has_value() || (in_scale->size(0) == x.size(0) && in_scale->size(-1) == x.size(3)), op, ": in_scale is [B][Ci]");
  TORCH_CHECK(!out_scale.has_value() || (out_scale->size(0) == x.size(0) && out_scale->size(-1) == Co), op, ": out_scale is [B][Co]");
  TORCH_CHECK(!bias.has_value() || bias->numel() == Co, op, ": bias is [Co] (padded)");
  o2m_conv_desc d{};
  d.x = ptr(x); d.w = ptr(w); d.y = ptr(y);
  d.in_scale = fptr(in_scale); d.out_scale = fptr(out_scale); d.bias = fptr(bias); d.residual = ptr(residual);
  d.B = i32(x.size(0), op); d.H = i32(x.size(1), op); d.W = i32(x.size(2), op); d.Ci = i32(x.size(3), op);
  d.Co = i32(Co, op); d.KH = i32(KH, op); d.KW = i32(KW, op); d.pad = i32(pad, op); d.pad_mode = i32(pad_mode, op);
  d.act = i32(act, op);
  d.dtype = f8 ? (x.scalar_type() == at::kFloat8_e4m3fn ? O2M_FP8_E4M3 : O2M_BF8_E5M2) : dtype_code(x, op);
  d.deq_scale = fptr(deq);
  d.w_batch_stride = per_sample_w ? i32(Co * KH * KW * x.size(3), op) : 0;
  d.stride = i32(stride, op);
  d.fold_pad = i32(fold_pad, op);
  if (stats.has_value()) {
    const int chunks = o2m_conv2d_stats_chunks(&d);
    TORCH_CHECK(chunks > 0, op, ": this conv cannot emit InstanceNorm partials (Ho*Wo is not a multiple of its row block)");
    TORCH_CHECK(stats->numel() >= x.size(0) * chunks * Co * 2, op, ": stats workspace too small");
    d.stats = ptr<float>(stats);
  }
  if (aux.has_value()) {  // style-dot partials instead of InstanceNorm moments (O2M_STATS_DOT)
    TORCH_CHECK(stats.has_value(), op, ": aux needs the stats workspace");
    TORCH_CHECK(aux->sizes() == y.sizes() && aux->scalar_type() == y.scalar_type(), op, ": aux must have y's shape and dtype");
    d.aux = ptr(aux);
    d.stats_mode = O2M_STATS_DOT;
    if (aux_scaled.has_value()) {
      TORCH_CHECK(aux_scaled->sizes() == y.sizes() && aux_scaled->scalar_type() == y.scalar_type() && out_scale.has_value(), op,
                  ": aux_scaled must have y's shape and dtype and needs out_scale");
      d.aux_scaled = ptr(aux_scaled);
    }
  } else {
    TORCH_CHECK(!aux_scaled.has_value(), op, ": aux_scaled needs aux");
  }
  O2M_CALL(op, x, o2m_conv2d_fwd(&d, stream));
}

void conv2d_reflect_border(const Tensor& x, const Tensor& w, Tensor& y) {
  const char* op = "o2m::conv2d_reflect_border";
  chk(x, op, "x"); chk(w, op, "w"); chk(y, op, "y");
  TORCH_CHECK(x.dim() == 4 && y.dim() == 4 && w.dim() == 4, op, ": x, y are NHWC; w is [Co][3][3][Ci]");
  same_dtype(x, w, op, "x", "w"); same_dtype(x, y, op, "x", "y");
  TORCH_CHECK(w.size(1) == 3 && w.size(2) == 3 && w.size(3) == x.size(3), op, ": w must be [Co][3][3][Ci] with x's channel count");
  TORCH_CHECK(y.size(0) == x.size(0) && y.size(1) == x.size(1) && y.size(2) == x.size(2) && y.size(3) == w.size(0), op,
              ": y must be [B][H][W][Co] on x's map");
  o2m_conv_desc d{};
  d.x = ptr(x); d.w = ptr(w); d.y = ptr(y);
  d.B = i32(x.size(0), op); d.H = i32(x.size(1), op); d.W = i32(x.size(2), op); d.Ci = i32(x.size(3), op);
  d.Co = i32(w.size(0), op); d.KH = 3; d.KW = 3; d.pad = 1; d.pad_mode = O2M_PAD_ZERO; d.dtype = dtype_code(x, op);
  O2M_CALL(op, x, o2m_conv2d_reflect_border(&d, stream));
}

void lsgan_fwd(const Tensor& scores, Tensor& out, int64_t n_first, double t0, double t1) {
  const char* op = "o2m::lsgan_fwd";
  chk(scores, op, "scores"); chk_f32(out, op, "out");
  TORCH_CHECK(scores.dim() == 4 && out.numel() >= 4, op, ": scores is internal [N][H][W][C]; out holds 4 floats");
  O2M_CALL(op, scores, o2m_lsgan_fwd(ptr(scores), ptr<float>(out), i32(scores.size(0), op), i32(scores.size(1) * scores.size(2), op),
                                    i32(scores.size(3), op), i32(n_first, op), static_cast<float>(t0), static_cast<float>(t1),
                                    dtype_code(scores, op), stream));
}

void lsgan_bwd(const Tensor& scores, const Tensor& coef, Tensor& g_scores, int64_t n_first, double t0, double t1) {
  const char* op = "o2m::lsgan_bwd";
  chk(scores, op, "scores"); chk_f32(coef, op, "coef"); chk(g_scores, op, "g_scores");
  TORCH_CHECK(scores.dim() == 4 && g_scores.sizes() == scores.sizes() && coef.numel() >= 2, op,
              ": scores / g_scores are internal [N][H][W][C]; coef holds 2 floats");
  same_dtype(scores, g_scores, op, "scores", "g_scores");
  O2M_CALL(op, scores, o2m_lsgan_bwd(ptr(scores), ptr<float>(coef), ptr(g_scores), i32(scores.size(0), op),
                                    i32(scores.size(1) * scores.size(2), op), i32(scores.size(3), op), i32(n_first, op),
                                    static_cast<float>(t0), static_cast<float>(t1), dtype_code(scores, op), stream));
}

void conv2d_dots_finalize(const Tensor& partial, Tensor& dots, int64_t nchunks) {
  const char* op = "o2m::conv2d_dots_finalize";
  chk_f32(partial, op, "partial"); chk_f32(dots, op, "dots");
  TORCH_CHECK(dots.dim() == 2 && nchunks > 0 && partial.numel() >= dots.numel() * nchunks * 2, op,
              ": dots is [B][C]; partial holds [B][nchunks][C][2]");
  O2M_CALL(op, dots, o2m_conv2d_dots_finalize(ptr<float>(partial), ptr<float>(dots), i32(dots.size(0), op), i32(dots.size(1), op),
                                             i32(nchunks, op), stream));
}

int64_t conv2d_stats_rows(const Tensor& x, const Tensor& w, const Tensor& y, int64_t pad, int64_t stride) {
  const char* op = "o2m::conv2d_stats_rows";
  TORCH_CHECK(x.dim() == 4 && w.dim() == 4 && y.dim() == 4, op, ": x, y are NHWC; w is [Co][KH][KW][Ci]");
  o2m_conv_desc d{};
  d.B = i32(x.size(0), op); d.H = i32(x.size(1), op); d.W = i32(x.size(2), op); d.Ci = i32(x.size(3), op);
  d.Co = i32(w.size(0), op); d.KH = i32(w.size(1), op); d.KW = i32(w.size(2), op); d.pad = i32(pad, op);
  d.dtype = is_fp8(x) ? O2M_FP8_E4M3 : dtype_code(x, op); d.stride = i32(stride, op);
  return o2m_conv2d_stats_rows(&d);
}

int64_t conv2d_stats_chunks(const Tensor& x, const Tensor& w, const Tensor& y, int64_t pad, int64_t stride) {
  const char* op = "o2m::conv2d_stats_chunks";
  TORCH_CHECK(x.dim() == 4 && w.dim() == 4 && y.dim() == 4, op, ": x, y are NHWC; w is [Co][KH][KW][Ci]");
  o2m_conv_desc d{};
  d.B = i32(x.size(0), op); d.H = i32(x.size(1), op); d.W = i32(x.size(2), op); d.Ci = i32(x.size(3), op);
  d.Co = i32(w.size(0), op); d.KH = i32(w.size(1), op); d.KW = i32(w.size(2), op); d.pad = i32(pad, op);
  d.dtype = is_fp8(x) ? O2M_FP8_E4M3 : dtype_code(x, op); d.stride = i32(stride, op);
  return o2m_conv2d_stats_chunks(&d);
}

static void fill_wgrad_dims(o2m_wgrad_desc& d, const Tensor& x, const Tensor& dw, int64_t pad, int64_t pad_mode, int64_t splits,
                            int64_t stride, const char* op, int64_t hint = 0) {
  d.kernel_hint = i32(hint, op);
  d.B = i32(x.size(0), op); d.H = i32(x.size(1), op); d.W = i32(x.size(2), op); d.Ci = i32(x.size(3), op);
  d.Co = i32(dw.size(0), op); d.KH = i32(dw.size(1), op); d.KW = i32(dw.size(2), op);
  d.pad = i32(pad, op); d.pad_mode = i32(pad_mode, op); d.dtype = dtype_code(x, op); d.splits = i32(splits, op);
  d.stride = i32(stride, op);
}

int64_t conv2d_wgrad_slab_floats(const Tensor& x, const Tensor& gy, const Tensor& dw, int64_t pad, int64_t pad_mode, int64_t splits,
                                 int64_t stride, int64_t hint) {
  const char* op = "o2m::conv2d_wgrad_slab_floats";
  TORCH_CHECK(x.dim() == 4 && gy.dim() == 4 && dw.dim() == 4, op, ": x, gy are NHWC; dw is [Co][KH][KW][Ci]");
  o2m_wgrad_desc d{};
  fill_wgrad_dims(d, x, dw, pad, pad_mode, splits, stride, op, hint);
  return static_cast<int64_t>(o2m_conv2d_wgrad_slab_floats(&d));
}

void conv2d_wgrad(const Tensor& x, const Tensor& gy, Tensor& dw, const OptT& in_scale, const OptT& gy_scale,
                  int64_t pad, int64_t pad_mode, int64_t splits, int64_t stride, const std::optional<Tensor>& slabs,
                  int64_t hint) {
  const char* op = "o2m::conv2d_wgrad";
  chk_f32(slabs, op, "slabs");
  chk(x, op, "x"); chk(gy, op, "gy"); chk_f32(dw, op, "dw"); chk_f32(in_scale, op, "in_scale"); chk_f32(gy_scale, op, "gy_scale");
  TORCH_CHECK(x.dim() == 4 && gy.dim() == 4 && dw.dim() == 4, op, ": x, gy are NHWC; dw is [Co][KH][KW][Ci]");
  same_dtype(x, gy, op, "x", "gy");
  TORCH_CHECK(dw.size(3) == x.size(3) && dw.size(0) == gy.size(3) && gy.size(0) == x.size(0), op, ": shapes of x / gy / dw disagree");
  o2m_wgrad_desc d{};
  d.x = ptr(x); d.gy = ptr(gy); d.dw = ptr<float>(dw); d.in_scale = fptr(in_scale); d.gy_scale = fptr(gy_scale);
  fill_wgrad_dims(d, x, dw, pad, pad_mode, splits, stride, op, hint);
  if (slabs.has_value()) {
    TORCH_CHECK(static_cast<size_t>(slabs->numel()) >= o2m_conv2d_wgrad_slab_floats(&d), op, ": slab workspace too small");
    d.slabs = ptr<float>(slabs);
  }
  O2M_CALL(op, x, o2m_conv2d_wgrad(&d, stream));
}

void wgrad_finalize(Tensor& acc, const std::optional<Tensor>& gq, const Tensor& w32, Tensor& grad, int64_t co, int64_t ci, double c) {
  const char* op = "o2m::wgrad_finalize";
  chk_f32(acc, op, "acc"); chk_f32(gq, op, "gq"); chk_f32(w32, op, "w32"); chk_f32(grad, op, "grad");
  TORCH_CHECK(acc.dim() == 4 && w32.sizes() == acc.sizes(), op, ": acc and w32 are [Cop][KH][KW][Cip]");
  const int64_t kk = acc.size(1) * acc.size(2);
  TORCH_CHECK(grad.numel() == co * ci * kk, op, ": grad must hold Co*Ci*KH*KW elements");
  TORCH_CHECK(!gq.has_value() || (gq->size(0) == acc.size(0) && gq->size(1) == acc.size(3)), op, ": gq is [Cop][Cip]");
  O2M_CALL(op, acc, o2m_wgrad_finalize(ptr<float>(acc), ptr<float>(gq), ptr<float>(w32), ptr<float>(grad), i32(co, op), i32(ci, op),
                                      i32(kk, op), i32(acc.size(0), op), i32(acc.size(3), op), static_cast<float>(c), stream));
}

void prepare_weights(const Tensor& w, Tensor& full, Tensor& w_f, Tensor& w_d, const std::optional<Tensor>& q,
                     const std::optional<Tensor>& qt, double c) {
  const char* op = "o2m::prepare_weights";
  chk_f32(w, op, "w"); chk_f32(full, op, "full"); chk(w_f, op, "w_f"); chk(w_d, op, "w_d"); chk_f32(q, op, "q"); chk_f32(qt, op, "qt");
  TORCH_CHECK(w.dim() == 4 && full.dim() == 4, op, ": w is [Co][Ci][KH][KW], full is [Cop][KH][KW][Cip]");
  same_dtype(w_f, w_d, op, "w_f", "w_d");
  const int64_t kk = w.size(2) * w.size(3), cop = full.size(0), cip = full.size(3);
  TORCH_CHECK(full.size(1) * full.size(2) == kk && cop >= w.size(0) && cip >= w.size(1), op, ": full does not match w");
  TORCH_CHECK(w_f.numel() == full.numel() && w_d.numel() == full.numel(), op, ": w_f / w_d must have full's element count");
  TORCH_CHECK(q.has_value() == qt.has_value(), op, ": q and qt come together");
  TORCH_CHECK(!q.has_value() || (q->numel() == cop * cip && qt->numel() == cop * cip), op, ": q is [Cop][Cip], qt its transpose");
  const int dt_w_f = dtype_code(w_f, op);
  O2M_CALL(op, w, o2m_prepare_weights(ptr<float>(w), ptr<float>(full), ptr(w_f), ptr(w_d), ptr<float>(q), ptr<float>(qt),
                                     i32(w.size(0), op), i32(w.size(1), op), i32(kk, op), i32(cop, op), i32(cip, op),
                                     static_cast<float>(c), dt_w_f, stream));
}

int64_t wgrad_finalize_blocks(int64_t cop, int64_t kk, int64_t cip) {
  return o2m_wgrad_finalize_blocks(static_cast<int32_t>(cop), static_cast<int32_t>(kk), static_cast<int32_t>(cip));
}

void wgrad_finalize_batched(const Tensor& jobs, int64_t n_jobs, int64_t total_blocks, bool any_gq, at::TensorList touched) {
  const char* op = "o2m::wgrad_finalize_batched";
  chk(jobs, op, "jobs");
  TORCH_CHECK(jobs.scalar_type() == at::kByte && jobs.numel() >= n_jobs * static_cast<int64_t>(sizeof(o2m_wfin_job)), op,
              ": jobs is a uint8 device tensor of n_jobs o2m_wfin_job records");
  for (const Tensor& t : touched) chk_f32(t, op, "touched[]");  // accumulators, dL/dQ tables, gradients the records point at
  O2M_CALL(op, jobs, o2m_wgrad_finalize_batched(static_cast<const o2m_wfin_job*>(ptr(jobs)), i32(n_jobs, op), i32(total_blocks, op),
                                               any_gq ? 1 : 0, stream));
}

void prepare_weights_batched(const Tensor& jobs, int64_t n_jobs, int64_t total_blocks, int64_t dtype, at::TensorList outs) {
  const char* op = "o2m::prepare_weights_batched";
  chk(jobs, op, "jobs");
  TORCH_CHECK(jobs.scalar_type() == at::kByte && jobs.numel() >= n_jobs * static_cast<int64_t>(sizeof(o2m_prep_job)), op,
              ": jobs is a uint8 device tensor of n_jobs o2m_prep_job records");
  for (const Tensor& t : outs) chk(t, op, "outs[]");  // the buffers the records point at (kept alive / declared mutated)
  O2M_CALL(op, jobs, o2m_prepare_weights_batched(static_cast<const o2m_prep_job*>(ptr(jobs)), i32(n_jobs, op), i32(total_blocks, op),
                                                i32(dtype, op), stream));
}

void amax(const Tensor& x, Tensor& out) {
  const char* op = "o2m::amax";
  chk(x, op, "x"); chk_f32(out, op, "amax");
  TORCH_CHECK(out.numel() >= O2M_AMAX_PARTIALS && x.numel() % 8 == 0, op, ": amax workspace holds ", O2M_AMAX_PARTIALS,
              " floats; x.numel() % 8 == 0");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_amax(ptr(x), ptr<float>(out), x.numel(), dt_x, stream));
}

void quantize_fp8(const Tensor& x, const Tensor& amax_t, Tensor& y, Tensor& deq) {
  const char* op = "o2m::quantize_fp8";
  chk(x, op, "x"); chk_f32(amax_t, op, "amax"); chk(y, op, "y"); chk_f32(deq, op, "deq");
  TORCH_CHECK(is_fp8(y) && y.numel() == x.numel() && x.numel() % 8 == 0, op, ": y is an fp8 tensor of x's size (multiple of 8)");
  TORCH_CHECK(amax_t.numel() >= O2M_AMAX_PARTIALS && deq.numel() >= 2, op, ": amax workspace of ", O2M_AMAX_PARTIALS,
              " floats, deq of 2 ({1 / scale, amax})");
  const int dt_x = dtype_code(x, op);
  const int fmt = y.scalar_type() == at::kFloat8_e4m3fn ? O2M_FP8_E4M3 : O2M_BF8_E5M2;
  O2M_CALL(op, x, o2m_quantize_fp8(ptr(x), ptr<float>(amax_t), ptr(y), ptr<float>(deq), x.numel(), dt_x, fmt, stream));
}

void quantize_fp8_delayed(const Tensor& x, const Tensor& amax_prev, Tensor& y, Tensor& deq, Tensor& amax_next) {
  const char* op = "o2m::quantize_fp8_delayed";
  chk(x, op, "x"); chk_f32(amax_prev, op, "amax_prev"); chk(y, op, "y"); chk_f32(deq, op, "deq"); chk_f32(amax_next, op, "amax_next");
  TORCH_CHECK(is_fp8(y) && y.numel() == x.numel() && x.numel() % 8 == 0, op, ": y is an fp8 tensor of x's size (multiple of 8)");
  TORCH_CHECK(amax_prev.numel() >= O2M_AMAX_PARTIALS && amax_next.numel() >= O2M_AMAX_PARTIALS && deq.numel() >= 2 &&
                  amax_prev.data_ptr() != amax_next.data_ptr(),
              op, ": two different amax workspaces of ", O2M_AMAX_PARTIALS, " floats, deq of 2 ({1 / scale, amax})");
  const int dt_x = dtype_code(x, op);
  const int fmt = y.scalar_type() == at::kFloat8_e4m3fn ? O2M_FP8_E4M3 : O2M_BF8_E5M2;
  O2M_CALL(op, x, o2m_quantize_fp8_delayed(ptr(x), ptr<float>(amax_prev), ptr(y), ptr<float>(deq), ptr<float>(amax_next), x.numel(),
                                           dt_x, fmt, stream));
}

void modulate_weights(const Tensor& w32, const Tensor& s, Tensor& out) {
  const char* op = "o2m::modulate_weights";
  chk_f32(w32, op, "w32"); chk_f32(s, op, "s"); chk(out, op, "out");
  TORCH_CHECK(w32.dim() == 4 && s.dim() == 2 && s.size(1) == w32.size(3), op, ": w32 is [Co][KH][KW][Ci], s is [B][Ci]");
  TORCH_CHECK(out.numel() == s.size(0) * w32.numel(), op, ": out is [B][Co][KH][KW][Ci]");
  const int dt_out = dtype_code(out, op);
  O2M_CALL(op, out, o2m_modulate_weights(ptr<float>(w32), ptr<float>(s), ptr(out), i32(s.size(0), op), i32(w32.size(0), op),
                                        i32(w32.size(1) * w32.size(2), op), i32(w32.size(3), op), dt_out, stream));
}

// ------------------------------------------------------------------------------------ style

void style_fwd(const Tensor& w, const Tensor& ws, const Tensor& bs, const OptT& qt, Tensor& s, const std::optional<Tensor>& d,
               int64_t ci, double cs, double eps) {
  const char* op = "o2m::style_fwd";
  chk_f32(w, op, "w"); chk_f32(ws, op, "ws"); chk_f32(bs, op, "bs"); chk_f32(qt, op, "qt"); chk_f32(s, op, "s"); chk_f32(d, op, "d");
  TORCH_CHECK(w.dim() == 2 && s.dim() == 2 && s.size(0) == w.size(0), op, ": w is [B][WD], s is [B][Cip]");
  TORCH_CHECK(ws.numel() == ci * w.size(1) && bs.numel() == ci, op, ": to_style weight is [Ci][WD], bias [Ci]");
  const int64_t cop = d.has_value() ? d->size(1) : 0;
  TORCH_CHECK(!d.has_value() || (qt.has_value() && qt->numel() == s.size(1) * cop), op, ": demodulation needs qt [Cip][Cop]");
  O2M_CALL(op, s, o2m_style_fwd(ptr<float>(w), ptr<float>(ws), ptr<float>(bs), fptr(qt), ptr<float>(s), ptr<float>(d),
                               i32(w.size(0), op), i32(w.size(1), op), i32(ci, op), i32(s.size(1), op), i32(cop, op),
                               static_cast<float>(cs), static_cast<float>(eps), stream));
}

void style_bwd(const OptT& sums, const OptT& bias, const OptT& dots, const Tensor& s, const OptT& d, const OptT& q,
               const Tensor& w, const Tensor& ws, const std::optional<Tensor>& e, Tensor& gs, Tensor& gw, Tensor& gws, Tensor& gbs,
               const std::optional<Tensor>& gq, int64_t ci, double cs, bool accumulate) {
  const char* op = "o2m::style_bwd";
  chk_f32(sums, op, "sums"); chk_f32(bias, op, "bias"); chk_f32(dots, op, "dots"); chk_f32(s, op, "s"); chk_f32(d, op, "d");
  chk_f32(q, op, "q"); chk_f32(w, op, "w"); chk_f32(ws, op, "ws"); chk_f32(e, op, "e"); chk_f32(gs, op, "gs"); chk_f32(gw, op, "gw");
  chk_f32(gws, op, "gws"); chk_f32(gbs, op, "gbs"); chk_f32(gq, op, "gq");
  TORCH_CHECK(w.dim() == 2 && s.dim() == 2 && s.size(0) == w.size(0) && gs.sizes() == s.sizes() && gw.sizes() == w.sizes(), op,
              ": w / gw are [B][WD], s / gs are [B][Cip]");
  TORCH_CHECK(gws.numel() == ci * w.size(1) && gbs.numel() == ci, op, ": gws is [Ci][WD], gbs is [Ci]");
  const int64_t cop = d.has_value() ? d->size(1) : 8;
  O2M_CALL(op, s, o2m_style_bwd(fptr(sums), fptr(bias), fptr(dots), ptr<float>(s), fptr(d), fptr(q), ptr<float>(w), ptr<float>(ws),
                               ptr<float>(e), ptr<float>(gs), ptr<float>(gw), ptr<float>(gws), ptr<float>(gbs), ptr<float>(gq),
                               i32(w.size(0), op), i32(w.size(1), op), i32(ci, op), i32(s.size(1), op), i32(cop, op),
                               static_cast<float>(cs), accumulate ? 1 : 0, stream));
}

// -------------------------------------------------------------------------------- pointwise

void act_bwd_reduce(const Tensor& g, const OptT& y, const OptT& residual, const OptT& out_mul, const std::optional<Tensor>& gu,
                    const std::optional<Tensor>& sums, int64_t act, const std::optional<Tensor>& partials) {
  const char* op = "o2m::act_bwd_reduce";
  chk(g, op, "g"); chk(y, op, "y"); chk(residual, op, "residual"); chk_f32(out_mul, op, "out_mul"); chk(gu, op, "gu"); chk_f32(sums, op, "sums");
  TORCH_CHECK(g.dim() == 4, op, ": g is NHWC");
  same_dtype(g, y, op, "g", "y"); same_dtype(g, residual, op, "g", "residual"); same_dtype(g, gu, op, "g", "gu");
  TORCH_CHECK(!y.has_value() || y->sizes() == g.sizes(), op, ": y must have g's shape");
  TORCH_CHECK(!gu.has_value() || gu->sizes() == g.sizes(), op, ": gu must have g's shape");
  TORCH_CHECK(!sums.has_value() || sums->numel() == g.size(0) * 2 * g.size(3), op, ": sums is [B][2][C]");
  chk_f32(partials, op, "partials");
  TORCH_CHECK(!partials.has_value() ||
                  partials->numel() >= static_cast<int64_t>(o2m_chan_partials_floats(g.size(0), g.size(1) * g.size(2), g.size(3), 2)),
              op, ": partials workspace too small");
  const int dt_g = dtype_code(g, op);
  O2M_CALL(op, g, o2m_act_bwd_reduce(ptr(g), ptr(y), ptr(residual), fptr(out_mul), ptr(gu), ptr<float>(sums), i32(g.size(0), op),
                                    i32(g.size(1) * g.size(2), op), i32(g.size(3), op), i32(act, op), dt_g, ptr<float>(partials), stream));
}

void fold_scale_dot(const Tensor& gpad, const OptT& x, const OptT& scale, Tensor& gx, const std::optional<Tensor>& dots, int64_t pad,
                    const std::optional<Tensor>& xs, const OptT& gres, int64_t act, const OptT& act_mul,
                    const std::optional<Tensor>& act_sums, const std::optional<Tensor>& partials) {
  const char* op = "o2m::fold_scale_dot";
  chk(gpad, op, "gpad"); chk(x, op, "x"); chk_f32(scale, op, "scale"); chk(gx, op, "gx"); chk_f32(dots, op, "dots"); chk(xs, op, "xs");
  chk(gres, op, "gres"); chk_f32(act_mul, op, "act_mul"); chk_f32(act_sums, op, "act_sums");
  TORCH_CHECK(gx.dim() == 4 && gpad.dim() == 4, op, ": gpad, gx are NHWC");
  same_dtype(gx, gpad, op, "gx", "gpad"); same_dtype(gx, x, op, "gx", "x"); same_dtype(gx, xs, op, "gx", "xs");
  same_dtype(gx, gres, op, "gx", "gres");
  TORCH_CHECK(gpad.size(0) == gx.size(0) && gpad.size(1) == gx.size(1) + 2 * pad && gpad.size(2) == gx.size(2) + 2 * pad &&
                  gpad.size(3) == gx.size(3), op, ": gpad must be gx's shape plus the padding margins");
  TORCH_CHECK(!x.has_value() || x->sizes() == gx.sizes(), op, ": x must have gx's shape");
  TORCH_CHECK(!xs.has_value() || xs->sizes() == gx.sizes(), op, ": xs must have gx's shape");
  TORCH_CHECK(!gres.has_value() || gres->sizes() == gx.sizes(), op, ": gres must have gx's shape");
  TORCH_CHECK(!act_sums.has_value() || act_sums->numel() == gx.size(0) * 2 * gx.size(3), op, ": act_sums is [B][2][C]");
  TORCH_CHECK(!act_mul.has_value() || act_mul->numel() == gx.size(0) * gx.size(3), op, ": act_mul is [B][C]");
  chk_f32(partials, op, "partials");
  TORCH_CHECK(!partials.has_value() ||
                  partials->numel() >= static_cast<int64_t>(o2m_chan_partials_floats(gx.size(0), gx.size(1) * gx.size(2), gx.size(3),
                                                                                     act_sums.has_value() ? 3 : 1)),
              op, ": partials workspace too small");
  const int dt_gx = dtype_code(gx, op);
  O2M_CALL(op, gx, o2m_fold_scale_dot(ptr(gpad), ptr(x), fptr(scale), ptr(gx), ptr<float>(dots), ptr(xs), ptr(gres), i32(act, op),
                                     fptr(act_mul), ptr<float>(act_sums), i32(gx.size(0), op), i32(gx.size(1), op),
                                     i32(gx.size(2), op), i32(gx.size(3), op), i32(pad, op), dt_gx, ptr<float>(partials), stream));
}

void instnorm_act_resample2d(const Tensor& x, const Tensor& mean_rstd, Tensor& y, const Tensor& sy, const Tensor& wy, const Tensor& sx,
                             const Tensor& wx, int64_t t, int64_t span_y, int64_t span_x, int64_t act) {
  const char* op = "o2m::instnorm_act_resample2d";
  chk(x, op, "x"); chk_f32(mean_rstd, op, "mean_rstd"); chk(y, op, "y"); chk(sy, op, "sy"); chk_f32(wy, op, "wy"); chk(sx, op, "sx");
  chk_f32(wx, op, "wx");
  TORCH_CHECK(x.dim() == 4 && y.dim() == 4 && x.size(0) == y.size(0) && x.size(3) == y.size(3), op, ": x, y are NHWC of one batch / width");
  same_dtype(x, y, op, "x", "y");
  TORCH_CHECK(sy.scalar_type() == at::kInt && sx.scalar_type() == at::kInt, op, ": tap starts are int32");
  TORCH_CHECK(sy.numel() == y.size(1) && sx.numel() == y.size(2) && wy.numel() == y.size(1) * t && wx.numel() == y.size(2) * t, op,
              ": taps are [Ho] / [Ho][T] and [Wo] / [Wo][T]");
  TORCH_CHECK(mean_rstd.numel() == x.size(0) * x.size(3) * 2, op, ": mean_rstd is [B][C][2]");
  TORCH_CHECK(t == 4 && span_y >= 2 && span_y <= 3 && span_x >= 2 && span_x <= 3, op,
              ": covers the DownSample operators only (4 taps, starts 2 or 3 apart)");
  const int dt = dtype_code(x, op);
  O2M_CALL(op, x, o2m_instnorm_act_resample2d(ptr(x), ptr<float>(mean_rstd), ptr(y), ptr<int32_t>(sy), ptr<float>(wy), ptr<int32_t>(sx),
                                             ptr<float>(wx), i32(x.size(0), op), i32(x.size(1), op), i32(x.size(2), op),
                                             i32(y.size(1), op), i32(y.size(2), op), i32(x.size(3), op), i32(t, op), i32(span_y, op),
                                             i32(span_x, op), i32(act, op), dt, stream));
}

void instnorm_resample_bwd(const Tensor& g_coarse, const Tensor& x, const Tensor& mean_rstd, Tensor& partial, Tensor& gsums, Tensor& gx,
                           const Tensor& sy, const Tensor& wy, const Tensor& sx, const Tensor& wx, int64_t t, int64_t act) {
  const char* op = "o2m::instnorm_resample_bwd";
  chk(g_coarse, op, "g_coarse"); chk(x, op, "x"); chk_f32(mean_rstd, op, "mean_rstd"); chk_f32(partial, op, "partial");
  chk_f32(gsums, op, "gsums"); chk(gx, op, "gx"); chk(sy, op, "sy"); chk_f32(wy, op, "wy"); chk(sx, op, "sx"); chk_f32(wx, op, "wx");
  TORCH_CHECK(x.dim() == 4 && g_coarse.dim() == 4 && gx.sizes() == x.sizes() && g_coarse.size(0) == x.size(0) &&
                  g_coarse.size(3) == x.size(3), op, ": x, gx [B][H][W][C]; g_coarse [B][Hl][Wl][C]");
  same_dtype(x, g_coarse, op, "x", "g_coarse"); same_dtype(x, gx, op, "x", "gx");
  TORCH_CHECK(sy.scalar_type() == at::kInt && sx.scalar_type() == at::kInt, op, ": tap starts are int32");
  TORCH_CHECK(sy.numel() == x.size(1) && sx.numel() == x.size(2) && wy.numel() == x.size(1) * t && wx.numel() == x.size(2) * t, op,
              ": the TRANSPOSED operator's taps are [H] / [H][T] and [W] / [W][T]");
  TORCH_CHECK(partial.numel() >= static_cast<int64_t>(o2m_instnorm_ws_floats(x.size(0), x.size(1) * x.size(2), x.size(3))) &&
                  gsums.numel() == x.size(0) * x.size(3) * 2 && mean_rstd.numel() == gsums.numel(), op, ": workspace sizes");
  const int dt = dtype_code(x, op);
  O2M_CALL(op, x, o2m_instnorm_resample_bwd(ptr(g_coarse), ptr(x), ptr<float>(mean_rstd), ptr<float>(partial), ptr<float>(gsums), ptr(gx),
                                           ptr<int32_t>(sy), ptr<float>(wy), ptr<int32_t>(sx), ptr<float>(wx), i32(x.size(0), op),
                                           i32(x.size(1), op), i32(x.size(2), op), i32(g_coarse.size(1), op), i32(g_coarse.size(2), op),
                                           i32(x.size(3), op), i32(t, op), i32(act, op), dt, stream));
}

int64_t chan_partials_floats(int64_t B, int64_t P, int64_t C, int64_t nv) {
  const char* op = "o2m::chan_partials_floats";
  return static_cast<int64_t>(o2m_chan_partials_floats(i32(B, op), i32(P, op), i32(C, op), i32(nv, op)));
}

int64_t instnorm_ws_floats(int64_t B, int64_t P, int64_t C) {
  return static_cast<int64_t>(o2m_instnorm_ws_floats(i32(B, "o2m::instnorm_ws_floats"), i32(P, "o2m::instnorm_ws_floats"),
                                                     i32(C, "o2m::instnorm_ws_floats")));
}

void instnorm_stats(const Tensor& x, Tensor& partial, Tensor& mean_rstd, double eps) {
  const char* op = "o2m::instnorm_stats";
  chk(x, op, "x"); chk_f32(partial, op, "partial"); chk_f32(mean_rstd, op, "mean_rstd");
  TORCH_CHECK(x.dim() == 4 && mean_rstd.numel() == x.size(0) * x.size(3) * 2, op, ": x is NHWC, mean_rstd is [B][C][2]");
  const int B = i32(x.size(0), op), P = i32(x.size(1) * x.size(2), op), C = i32(x.size(3), op);
  TORCH_CHECK(partial.numel() >= static_cast<int64_t>(o2m_instnorm_ws_floats(B, P, C)), op, ": workspace too small");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_instnorm_stats(ptr(x), ptr<float>(partial), ptr<float>(mean_rstd), B, P, C, static_cast<float>(eps), dt_x, stream));
}

void instnorm_finalize(const Tensor& partial, Tensor& mean_rstd, int64_t P, int64_t nchunks, double eps) {
  const char* op = "o2m::instnorm_finalize";
  chk_f32(partial, op, "partial"); chk_f32(mean_rstd, op, "mean_rstd");
  TORCH_CHECK(mean_rstd.dim() == 3 && mean_rstd.size(2) == 2, op, ": mean_rstd is [B][C][2]");
  const int64_t B = mean_rstd.size(0), C = mean_rstd.size(1);
  TORCH_CHECK(nchunks > 0 && partial.numel() >= B * nchunks * C * 2, op, ": partial is [B][nchunks][C][2]");
  O2M_CALL(op, partial, o2m_instnorm_finalize(ptr<float>(partial), ptr<float>(mean_rstd), i32(B, op), i32(P, op), i32(C, op),
                                           i32(nchunks, op), static_cast<float>(eps), stream));
}

void instnorm_apply(const Tensor& x, const Tensor& mean_rstd, const OptT& residual, Tensor& y, int64_t act) {
  const char* op = "o2m::instnorm_apply";
  chk(x, op, "x"); chk_f32(mean_rstd, op, "mean_rstd"); chk(residual, op, "residual"); chk(y, op, "y");
  TORCH_CHECK(x.dim() == 4 && y.sizes() == x.sizes() && mean_rstd.numel() == x.size(0) * x.size(3) * 2, op, ": shapes disagree");
  same_dtype(x, y, op, "x", "y"); same_dtype(x, residual, op, "x", "residual");
  TORCH_CHECK(!residual.has_value() || residual->sizes() == x.sizes(), op, ": residual must have x's shape");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_instnorm_apply(ptr(x), ptr<float>(mean_rstd), ptr(residual), ptr(y), i32(x.size(0), op), i32(x.size(1) * x.size(2), op),
                                    i32(x.size(3), op), i32(act, op), dt_x, stream));
}

void instnorm_bwd(const Tensor& g, const Tensor& x, const Tensor& mean_rstd, Tensor& partial, Tensor& gsums, Tensor& gx, int64_t act) {
  const char* op = "o2m::instnorm_bwd";
  chk(g, op, "g"); chk(x, op, "x"); chk_f32(mean_rstd, op, "mean_rstd"); chk_f32(partial, op, "partial"); chk_f32(gsums, op, "gsums"); chk(gx, op, "gx");
  TORCH_CHECK(x.dim() == 4 && g.sizes() == x.sizes() && gx.sizes() == x.sizes(), op, ": g, x, gx share one NHWC shape");
  same_dtype(x, g, op, "x", "g"); same_dtype(x, gx, op, "x", "gx");
  const int B = i32(x.size(0), op), P = i32(x.size(1) * x.size(2), op), C = i32(x.size(3), op);
  TORCH_CHECK(mean_rstd.numel() == (int64_t)B * C * 2 && gsums.numel() == (int64_t)B * C * 2, op, ": mean_rstd / gsums are [B][C][2]");
  TORCH_CHECK(partial.numel() >= static_cast<int64_t>(o2m_instnorm_ws_floats(B, P, C)), op, ": workspace too small");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_instnorm_bwd(ptr(g), ptr(x), ptr<float>(mean_rstd), ptr<float>(partial), ptr<float>(gsums), ptr(gx), B, P, C,
                                  i32(act, op), dt_x, stream));
}

void resample2d(const Tensor& x, Tensor& y, const Tensor& sy, const Tensor& wy, const Tensor& sx, const Tensor& wx, int64_t ty,
                int64_t tx, int64_t span_y, int64_t span_x) {
  const char* op = "o2m::resample2d";
  chk(x, op, "x"); chk(y, op, "y"); chk(sy, op, "sy"); chk_f32(wy, op, "wy"); chk(sx, op, "sx"); chk_f32(wx, op, "wx");
  TORCH_CHECK(x.dim() == 4 && y.dim() == 4 && y.size(0) == x.size(0) && y.size(3) == x.size(3), op, ": x, y are NHWC with equal B, C");
  same_dtype(x, y, op, "x", "y");
  TORCH_CHECK(sy.scalar_type() == at::kInt && sx.scalar_type() == at::kInt, op, ": sy / sx must be int32");
  TORCH_CHECK(sy.numel() == y.size(1) && sx.numel() == y.size(2) && wy.numel() == y.size(1) * ty && wx.numel() == y.size(2) * tx, op,
              ": taps do not match the output size");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_resample2d(ptr(x), ptr(y), ptr<int32_t>(sy), ptr<float>(wy), ptr<int32_t>(sx), ptr<float>(wx), i32(x.size(0), op),
                                i32(x.size(1), op), i32(x.size(2), op), i32(y.size(1), op), i32(y.size(2), op), i32(x.size(3), op), i32(ty, op),
                                i32(tx, op), i32(span_y, op), i32(span_x, op), dt_x, stream));
}

// --------------------------------------------------------------------- augmentation / input

void ada_grid_sample(const Tensor& x, const Tensor& theta, Tensor& y) {
  const char* op = "o2m::ada_grid_sample";
  chk(x, op, "x"); chk_f32(theta, op, "theta"); chk(y, op, "y");
  TORCH_CHECK(x.dim() == 4 && y.dim() == 4 && y.size(0) == x.size(0) && y.size(3) == x.size(3) && theta.numel() == x.size(0) * 6, op,
              ": x, y are NHWC with equal B, C; theta is [B][2][3]");
  same_dtype(x, y, op, "x", "y");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_ada_grid_sample(ptr(x), ptr<float>(theta), ptr(y), i32(x.size(0), op), i32(x.size(1), op), i32(x.size(2), op),
                                     i32(y.size(1), op), i32(y.size(2), op), i32(x.size(3), op), dt_x, stream));
}

void ada_grid_sample_bwd(const Tensor& gy, const Tensor& theta, Tensor& gx) {
  const char* op = "o2m::ada_grid_sample_bwd";
  chk(gy, op, "gy"); chk_f32(theta, op, "theta"); chk(gx, op, "gx");
  TORCH_CHECK(gy.dim() == 4 && gx.dim() == 4 && gy.size(0) == gx.size(0) && gy.size(3) == gx.size(3) && theta.numel() == gy.size(0) * 6, op,
              ": gy, gx are NHWC with equal B, C; theta is [B][2][3]");
  same_dtype(gy, gx, op, "gy", "gx");
  const int dt_gy = dtype_code(gy, op);
  O2M_CALL(op, gy, o2m_ada_grid_sample_bwd(ptr(gy), ptr<float>(theta), ptr(gx), i32(gy.size(0), op), i32(gx.size(1), op), i32(gx.size(2), op),
                                          i32(gy.size(1), op), i32(gy.size(2), op), i32(gy.size(3), op), dt_gy, stream));
}

void reflect_fold(const Tensor& gpad, Tensor& gx, int64_t pad_top, int64_t pad_left) {
  const char* op = "o2m::reflect_fold";
  chk(gpad, op, "gpad"); chk(gx, op, "gx");
  TORCH_CHECK(gpad.dim() == 4 && gx.dim() == 4 && gpad.size(0) == gx.size(0) && gpad.size(3) == gx.size(3), op, ": gpad, gx are NHWC with equal B, C");
  const int dt_gpad = dtype_code(gpad, op);
  const int dt_gx = dtype_code(gx, op);
  O2M_CALL(op, gx, o2m_reflect_fold(ptr(gpad), ptr(gx), i32(gx.size(0), op), i32(gx.size(1), op), i32(gx.size(2), op), i32(gpad.size(1), op),
                                   i32(gpad.size(2), op), i32(pad_top, op), i32(pad_left, op), i32(gx.size(3), op), dt_gpad,
                                   dt_gx, stream));
}

void ada_colour(const Tensor& x, const Tensor& m, Tensor& y, int64_t c) {
  const char* op = "o2m::ada_colour";
  chk(x, op, "x"); chk_f32(m, op, "m"); chk(y, op, "y");
  TORCH_CHECK(x.dim() == 4 && y.sizes() == x.sizes() && m.numel() == x.size(0) * 12, op, ": x, y share one NHWC shape; m is [B][3][4]");
  same_dtype(x, y, op, "x", "y");
  const int dt_x = dtype_code(x, op);
  O2M_CALL(op, x, o2m_ada_colour(ptr(x), ptr<float>(m), ptr(y), i32(x.size(0), op), x.size(1) * x.size(2), i32(c, op), i32(x.size(3), op),
                                dt_x, stream));
}

void gather_images(const Tensor& pool, const Tensor& index, const Tensor& flip, Tensor& out) {
  const char* op = "o2m::gather_images";
  chk(pool, op, "pool"); chk(index, op, "index"); chk(flip, op, "flip"); chk(out, op, "out");
  TORCH_CHECK(pool.scalar_type() == at::kByte && index.scalar_type() == at::kInt && flip.scalar_type() == at::kByte, op,
              ": pool uint8, index int32, flip uint8");
  TORCH_CHECK(pool.dim() == 4 && out.dim() == 4 && index.numel() == out.size(0) && flip.numel() == out.size(0) && out.size(1) == pool.size(1) &&
                  out.size(2) == pool.size(2), op, ": shape mismatch");
  const int dt_out = dtype_code(out, op);
  O2M_CALL(op, out, o2m_gather_images(ptr<uint8_t>(pool), ptr<int32_t>(index), ptr<uint8_t>(flip), ptr(out), i32(pool.size(0), op),
                                     i32(out.size(0), op), i32(pool.size(1), op), i32(pool.size(2), op), i32(pool.size(3), op),
                                     i32(out.size(3), op), dt_out, stream));
}

void pack_nchw(const Tensor& src, Tensor& dst) {
  const char* op = "o2m::pack_nchw";
  chk_f32(src, op, "src"); chk(dst, op, "dst");
  TORCH_CHECK(src.dim() == 4 && dst.dim() == 4 && dst.size(0) == src.size(0) && dst.size(1) == src.size(2) && dst.size(2) == src.size(3) &&
                  dst.size(3) >= src.size(1), op, ": src is NCHW fp32, dst its NHWC buffer (padded channels)");
  const int dt_dst = dtype_code(dst, op);
  O2M_CALL(op, dst, o2m_pack_nchw(ptr<float>(src), ptr(dst), i32(src.size(0), op), i32(src.size(1), op), i32(src.size(2), op),
                                 i32(src.size(3), op), i32(dst.size(3), op), dt_dst, stream));
}

void unpack_nhwc(const Tensor& src, Tensor& dst) {
  const char* op = "o2m::unpack_nhwc";
  chk(src, op, "src"); chk_f32(dst, op, "dst");
  TORCH_CHECK(src.dim() == 4 && dst.dim() == 4 && dst.size(0) == src.size(0) && dst.size(2) == src.size(1) && dst.size(3) == src.size(2) &&
                  src.size(3) >= dst.size(1), op, ": src is an NHWC buffer, dst NCHW fp32");
  const int dt_src = dtype_code(src, op);
  O2M_CALL(op, src, o2m_unpack_nhwc(ptr(src), ptr<float>(dst), i32(dst.size(0), op), i32(dst.size(1), op), i32(dst.size(2), op),
                                   i32(dst.size(3), op), i32(src.size(3), op), dt_src, stream));
}

// ---------------------------------------------------------------------- losses / optimiser

int64_t reduce_blocks(int64_t n) { return o2m_reduce_blocks(n); }

void reduce_fwd(const Tensor& a, const OptT& b, const OptT& w, Tensor& partials, int64_t mode) {
  const char* op = "o2m::reduce_fwd";
  chk(a, op, "a"); chk(b, op, "b"); chk_f32(w, op, "w"); chk_f32(partials, op, "partials");
  same_dtype(a, b, op, "a", "b");
  TORCH_CHECK(a.dim() >= 1 && a.size(0) > 0 && (!b.has_value() || b->sizes() == a.sizes()), op, ": a and b share one shape");
  TORCH_CHECK(!w.has_value() || w->numel() == a.size(0), op, ": w is [B]");
  const int64_t nps = a.numel() / a.size(0);
  TORCH_CHECK(partials.numel() >= (mode == O2M_RED_MOM ? 2 : 1) * (int64_t)o2m_reduce_blocks(a.numel()), op, ": partials too small");
  const int dt_a = dtype_code(a, op);
  O2M_CALL(op, a, o2m_reduce_fwd(ptr(a), ptr(b), fptr(w), ptr<float>(partials), i32(a.size(0), op), nps, i32(mode, op), dt_a, stream));
}

void reduce_bwd(const Tensor& a, const OptT& b, const OptT& w, const Tensor& coef, Tensor& ga, int64_t mode) {
  const char* op = "o2m::reduce_bwd";
  chk(a, op, "a"); chk(b, op, "b"); chk_f32(w, op, "w"); chk_f32(coef, op, "coef"); chk(ga, op, "ga");
  same_dtype(a, b, op, "a", "b"); same_dtype(a, ga, op, "a", "ga");
  TORCH_CHECK(ga.sizes() == a.sizes() && (!b.has_value() || b->sizes() == a.sizes()), op, ": a, b, ga share one shape");
  TORCH_CHECK(coef.numel() >= (mode == O2M_RED_MOM ? 2 : 1), op, ": coef too small");
  const int dt_a = dtype_code(a, op);
  O2M_CALL(op, a, o2m_reduce_bwd(ptr(a), ptr(b), fptr(w), ptr<float>(coef), ptr(ga), i32(a.size(0), op), a.numel() / a.size(0), i32(mode, op),
                                dt_a, stream));
}

void pair_grad(const Tensor& a, const Tensor& b, const OptT& w, const Tensor& coef, const OptT& gin_a, const OptT& gin_b, Tensor& ga,
               Tensor& gb) {
  const char* op = "o2m::pair_grad";
  chk(a, op, "a"); chk(b, op, "b"); chk_f32(w, op, "w"); chk_f32(coef, op, "coef"); chk(gin_a, op, "gin_a"); chk(gin_b, op, "gin_b");
  chk(ga, op, "ga"); chk(gb, op, "gb");
  same_dtype(a, b, op, "a", "b"); same_dtype(a, ga, op, "a", "ga"); same_dtype(a, gb, op, "a", "gb");
  same_dtype(a, gin_a, op, "a", "gin_a"); same_dtype(a, gin_b, op, "a", "gin_b");
  TORCH_CHECK(b.sizes() == a.sizes() && ga.sizes() == a.sizes() && gb.sizes() == a.sizes() &&
                  (!gin_a.has_value() || gin_a->sizes() == a.sizes()) && (!gin_b.has_value() || gin_b->sizes() == a.sizes()),
              op, ": a, b, gin_a, gin_b, ga, gb share one shape");
  TORCH_CHECK(coef.numel() >= 1 && (!w.has_value() || w->numel() >= a.size(0)), op, ": coef is one scalar, w one weight per sample");
  O2M_CALL(op, a, o2m_pair_grad(ptr(a), ptr(b), fptr(w), ptr<float>(coef), ptr(gin_a), ptr(gin_b), ptr(ga), ptr(gb), i32(a.size(0), op),
                               a.numel() / a.size(0), dtype_code(a, op), stream));
}

void adam_step(Tensor& p, const Tensor& g, Tensor& m, Tensor& v, const Tensor& step, double lr, double beta1, double beta2, double eps,
               double grad_scale) {
  const char* op = "o2m::adam_step";
  chk_f32(p, op, "p"); chk_f32(g, op, "g"); chk_f32(m, op, "m"); chk_f32(v, op, "v"); chk_f32(step, op, "step");
  TORCH_CHECK(g.numel() == p.numel() && m.numel() == p.numel() && v.numel() == p.numel() && step.numel() == 1, op,
              ": p, g, m, v are flat buckets of one length; step is one device scalar");
  O2M_CALL(op, p, o2m_adam_step(ptr<float>(p), ptr<float>(g), ptr<float>(m), ptr<float>(v), ptr<float>(step), p.numel(), static_cast<float>(lr),
                               static_cast<float>(beta1), static_cast<float>(beta2), static_cast<float>(eps), static_cast<float>(grad_scale), stream));
}

int64_t abi_version() { return o2m_abi_version(); }

}  // namespace

TORCH_LIBRARY(o2m, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("instnorm_ws_floats(int B, int P, int C) -> int", &instnorm_ws_floats);
  m.def("reduce_blocks(int n) -> int", &reduce_blocks);
  m.def("conv2d_fwd(Tensor x, Tensor w, Tensor(a!) y, Tensor? in_scale, Tensor? out_scale, Tensor? bias, Tensor? residual, "
        "int pad, int pad_mode, int act, bool per_sample_w, int stride, Tensor(b!)? stats=None, Tensor? deq=None, Tensor? aux=None, "
        "Tensor(c!)? aux_scaled=None, int fold_pad=0) -> ()");
  m.def("conv2d_reflect_border(Tensor x, Tensor w, Tensor(a!) y) -> ()");
  m.def("lsgan_fwd(Tensor scores, Tensor(a!) out, int n_first, float t0, float t1) -> ()");
  m.def("lsgan_bwd(Tensor scores, Tensor coef, Tensor(a!) g_scores, int n_first, float t0, float t1) -> ()");
  m.def("conv2d_dots_finalize(Tensor partial, Tensor(a!) dots, int nchunks) -> ()");
  m.def("amax(Tensor x, Tensor(a!) amax) -> ()");
  m.def("quantize_fp8(Tensor x, Tensor amax, Tensor(a!) y, Tensor(b!) deq) -> ()");
  m.def("quantize_fp8_delayed(Tensor x, Tensor amax_prev, Tensor(a!) y, Tensor(b!) deq, Tensor(c!) amax_next) -> ()");
  m.def("conv2d_stats_rows(Tensor x, Tensor w, Tensor y, int pad, int stride) -> int");
  m.def("conv2d_stats_chunks(Tensor x, Tensor w, Tensor y, int pad, int stride) -> int");
  m.def("instnorm_finalize(Tensor partial, Tensor(a!) mean_rstd, int P, int nchunks, float eps) -> ()");
  m.def("conv2d_wgrad(Tensor x, Tensor gy, Tensor(a!) dw, Tensor? in_scale, Tensor? gy_scale, int pad, int pad_mode, int splits, "
        "int stride, Tensor(b!)? slabs=None, int kernel_hint=0) -> ()");
  m.def("conv2d_wgrad_slab_floats(Tensor x, Tensor gy, Tensor dw, int pad, int pad_mode, int splits, int stride, "
        "int kernel_hint=0) -> int");
  m.def("wgrad_finalize(Tensor(a!) acc, Tensor(b!)? gq, Tensor w32, Tensor(c!) grad, int co, int ci, float c) -> ()");
  m.def("prepare_weights(Tensor w, Tensor(a!) full, Tensor(b!) w_f, Tensor(c!) w_d, Tensor(d!)? q, Tensor(e!)? qt, float c) -> ()");
  m.def("prepare_weights_batched(Tensor jobs, int n_jobs, int total_blocks, int dtype, Tensor(a!)[] outs) -> ()");
  m.def("wgrad_finalize_blocks(int cop, int kk, int cip) -> int", &wgrad_finalize_blocks);
  m.def("wgrad_finalize_batched(Tensor jobs, int n_jobs, int total_blocks, bool any_gq, Tensor(a!)[] touched) -> ()");
  m.def("modulate_weights(Tensor w32, Tensor s, Tensor(a!) out) -> ()");
  m.def("style_fwd(Tensor w, Tensor ws, Tensor bs, Tensor? qt, Tensor(a!) s, Tensor(b!)? d, int ci, float cs, float eps) -> ()");
  m.def("style_bwd(Tensor? sums, Tensor? bias, Tensor? dots, Tensor s, Tensor? d, Tensor? q, Tensor w, Tensor ws, Tensor(a!)? e, "
        "Tensor(b!) gs, Tensor(c!) gw, Tensor(d!) gws, Tensor(e!) gbs, Tensor(f!)? gq, int ci, float cs, bool accumulate) -> ()");
  m.def("act_bwd_reduce(Tensor g, Tensor? y, Tensor? residual, Tensor? out_mul, Tensor(a!)? gu, Tensor(b!)? sums, int act, Tensor(c!)? partials=None) -> ()");
  m.def("fold_scale_dot(Tensor gpad, Tensor? x, Tensor? scale, Tensor(a!) gx, Tensor(b!)? dots, int pad, Tensor(c!)? xs, Tensor? gres=None, "
        "int act=0, Tensor? act_mul=None, Tensor(d!)? act_sums=None, Tensor(e!)? partials=None) -> ()");
  m.def("chan_partials_floats(int B, int P, int C, int nv) -> int", &chan_partials_floats);
  m.def("instnorm_stats(Tensor x, Tensor(a!) partial, Tensor(b!) mean_rstd, float eps) -> ()");
  m.def("instnorm_apply(Tensor x, Tensor mean_rstd, Tensor? residual, Tensor(a!) y, int act) -> ()");
  m.def("instnorm_bwd(Tensor g, Tensor x, Tensor mean_rstd, Tensor(a!) partial, Tensor(b!) gsums, Tensor(c!) gx, int act) -> ()");
  m.def("instnorm_act_resample2d(Tensor x, Tensor mean_rstd, Tensor(a!) y, Tensor sy, Tensor wy, Tensor sx, Tensor wx, int t, int span_y, "
        "int span_x, int act) -> ()");
  m.def("instnorm_resample_bwd(Tensor g_coarse, Tensor x, Tensor mean_rstd, Tensor(a!) partial, Tensor(b!) gsums, Tensor(c!) gx, "
        "Tensor sy, Tensor wy, Tensor sx, Tensor wx, int t, int act) -> ()");
  m.def("resample2d(Tensor x, Tensor(a!) y, Tensor sy, Tensor wy, Tensor sx, Tensor wx, int ty, int tx, int span_y, int span_x) -> ()");
  m.def("ada_grid_sample(Tensor x, Tensor theta, Tensor(a!) y) -> ()");
  m.def("ada_grid_sample_bwd(Tensor gy, Tensor theta, Tensor(a!) gx) -> ()");
  m.def("reflect_fold(Tensor gpad, Tensor(a!) gx, int pad_top, int pad_left) -> ()");
  m.def("ada_colour(Tensor x, Tensor m, Tensor(a!) y, int c) -> ()");
  m.def("gather_images(Tensor pool, Tensor index, Tensor flip, Tensor(a!) out) -> ()");
  m.def("pack_nchw(Tensor src, Tensor(a!) dst) -> ()");
  m.def("unpack_nhwc(Tensor src, Tensor(a!) dst) -> ()");
  m.def("reduce_fwd(Tensor a, Tensor? b, Tensor? w, Tensor(a!) partials, int mode) -> ()");
  m.def("reduce_bwd(Tensor a, Tensor? b, Tensor? w, Tensor coef, Tensor(a!) ga, int mode) -> ()");
  m.def("pair_grad(Tensor a, Tensor b, Tensor? w, Tensor coef, Tensor? gin_a, Tensor? gin_b, Tensor(a!) ga, Tensor(b!) gb) -> ()");
  m.def("adam_step(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, Tensor step, float lr, float beta1, float beta2, float eps, "
        "float grad_scale) -> ()");
}

// One registration serves the GPU key and the Meta key: on meta tensors every op runs its argument
// checks and returns (outputs are caller-allocated), which is all FakeTensor tracing needs.
#define O2M_IMPLS(m)                              \
  m.impl("conv2d_fwd", &conv2d_fwd);              \
  m.impl("conv2d_wgrad", &conv2d_wgrad);          \
  m.impl("conv2d_wgrad_slab_floats", &conv2d_wgrad_slab_floats); \
  m.impl("wgrad_finalize", &wgrad_finalize);      \
  m.impl("prepare_weights", &prepare_weights);    \
  m.impl("prepare_weights_batched", &prepare_weights_batched); \
  m.impl("wgrad_finalize_batched", &wgrad_finalize_batched); \
  m.impl("modulate_weights", &modulate_weights);  \
  m.impl("amax", &amax);                          \
  m.impl("quantize_fp8", &quantize_fp8);          \
  m.impl("quantize_fp8_delayed", &quantize_fp8_delayed); \
  m.impl("style_fwd", &style_fwd);                \
  m.impl("style_bwd", &style_bwd);                \
  m.impl("act_bwd_reduce", &act_bwd_reduce);      \
  m.impl("fold_scale_dot", &fold_scale_dot);      \
  m.impl("instnorm_stats", &instnorm_stats);      \
  m.impl("instnorm_finalize", &instnorm_finalize); \
  m.impl("conv2d_stats_rows", &conv2d_stats_rows); \
  m.impl("conv2d_stats_chunks", &conv2d_stats_chunks); \
  m.impl("conv2d_dots_finalize", &conv2d_dots_finalize); \
  m.impl("lsgan_fwd", &lsgan_fwd);                \
  m.impl("lsgan_bwd", &lsgan_bwd);                \
  m.impl("conv2d_reflect_border", &conv2d_reflect_border); \
  m.impl("instnorm_apply", &instnorm_apply);      \
  m.impl("instnorm_bwd", &instnorm_bwd);          \
  m.impl("instnorm_act_resample2d", &instnorm_act_resample2d); \
  m.impl("instnorm_resample_bwd", &instnorm_resample_bwd); \
  m.impl("resample2d", &resample2d);              \
  m.impl("ada_grid_sample", &ada_grid_sample);    \
  m.impl("ada_grid_sample_bwd", &ada_grid_sample_bwd); \
  m.impl("reflect_fold", &reflect_fold);          \
  m.impl("ada_colour", &ada_colour);              \
  m.impl("gather_images", &gather_images);        \
  m.impl("pack_nchw", &pack_nchw);                \
  m.impl("unpack_nhwc", &unpack_nhwc);            \
  m.impl("reduce_fwd", &reduce_fwd);              \
  m.impl("reduce_bwd", &reduce_bwd);              \
  m.impl("pair_grad", &pair_grad);                \
  m.impl("adam_step", &adam_step);

TORCH_LIBRARY_IMPL(o2m, CUDA, m) { O2M_IMPLS(m) }
TORCH_LIBRARY_IMPL(o2m, Meta, m) { O2M_IMPLS(m) }
// CPU tensors reach the same functions and are rejected there with the "no CPU fallback" message
TORCH_LIBRARY_IMPL(o2m, CPU, m) { O2M_IMPLS(m) }
