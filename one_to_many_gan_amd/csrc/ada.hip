// Adaptive discriminator augmentation: the kernels that are not
// banded resampling.  The low-pass 2x up / down-sampling and the reflection padding of the
// published pipe run through o2m_resample2d with operators built on the host
// (one_to_many_gan_amd/ada.py); here are the bilinear affine resampling between them (and its
// adjoint), the adjoint of an asymmetric reflection padding, and the per-sample colour affine.
// Images are NHWC with the channels padded to 8 (Cp); only the first C (1 or 3) are real.
#include "common.h"

namespace {

constexpr int NT = 256;

// F.affine_grid(theta, size, align_corners=False) + F.grid_sample(bilinear, zeros, False):
// output pixel (ox, oy) -> normalised (xn, yn) -> source = theta @ (xn, yn, 1) -> pixel coords.
__device__ __forceinline__ void source_coords(const float* th, int ox, int oy, int Wo, int Ho, int Ws, int Hs,
                                              float& ix, float& iy) {
  const float xn = (2.f * ox + 1.f) / Wo - 1.f, yn = (2.f * oy + 1.f) / Ho - 1.f;
  const float xs = th[0] * xn + th[1] * yn + th[2];
  const float ys = th[3] * xn + th[4] * yn + th[5];
  ix = ((xs + 1.f) * Ws - 1.f) * 0.5f;
  iy = ((ys + 1.f) * Hs - 1.f) * 0.5f;
}

template <typename T>
__global__ __launch_bounds__(NT) void grid_sample_fwd_kernel(const T* __restrict__ x, const float* __restrict__ theta,
                                                             T* __restrict__ y, int Hs, int Ws, int Ho, int Wo,
                                                             int Cp) {
  const int b = blockIdx.z, oy = blockIdx.y, ox = blockIdx.x * NT + threadIdx.x;
  if (ox >= Wo) return;
  float ix, iy;
  source_coords(theta + b * 6, ox, oy, Wo, Ho, Ws, Hs, ix, iy);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float ax = ix - fx, ay = iy - fy;
  const float wgt[4] = {(1.f - ax) * (1.f - ay), ax * (1.f - ay), (1.f - ax) * ay, ax * ay};
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
      if ((unsigned)xx < (unsigned)Ws && (unsigned)yy < (unsigned)Hs) {
        float v[8];
        load8(x + (((size_t)b * Hs + yy) * Ws + xx) * Cp + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += wgt[k] * v[i];
      }
    }
    store8(y + (((size_t)b * Ho + oy) * Wo + ox) * Cp + c0, acc);
  }
}

// Adjoint as a GATHER.  Bilinear sampling with zero padding is y[o] = sum_s hat(ix(o) - sx) *
// hat(iy(o) - sy) * x[s] over the integer source pixels s, hat(t) = max(0, 1 - |t|), and the
// output -> source map is affine in pixel units: (ix, iy) = A (ox, oy) + t.  So
// gx[s] = sum_o hat * hat * gy[o], and the outputs that can reach s lie in the box
// |o - A^-1 (s - t)| <= (|A^-1| 1) -- a handful for the pipe's scales.  One thread per source
// pixel: no atomics (the scatter form cost 2.1 ms per call), no fp32 staging buffer, no zero
// fill; source pixels outside the sampled region write zeros.
template <typename T>
__global__ __launch_bounds__(NT) void grid_sample_bwd_kernel(const T* __restrict__ gy, const float* __restrict__ theta,
                                                             T* __restrict__ gx, int Hs, int Ws, int Ho, int Wo,
                                                             int Cp) {
  const int b = blockIdx.z, sy = blockIdx.y, sx = blockIdx.x * NT + threadIdx.x;
  if (sx >= Ws) return;
  const float* th = theta + b * 6;
  // pixel-space affine of source_coords(): ix = a00 ox + a01 oy + t0, iy = a10 ox + a11 oy + t1
  const float a00 = th[0] * Ws / Wo, a01 = th[1] * Ws / Ho, a10 = th[3] * Hs / Wo, a11 = th[4] * Hs / Ho;
  const float t0 = ((th[0] * (1.f / Wo - 1.f) + th[1] * (1.f / Ho - 1.f) + th[2] + 1.f) * Ws - 1.f) * 0.5f;
  const float t1 = ((th[3] * (1.f / Wo - 1.f) + th[4] * (1.f / Ho - 1.f) + th[5] + 1.f) * Hs - 1.f) * 0.5f;
  const float det = a00 * a11 - a01 * a10;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (fabsf(det) > 1e-12f) {
    const float i00 = a11 / det, i01 = -a01 / det, i10 = -a10 / det, i11 = a00 / det;
    const float dx = sx - t0, dy = sy - t1;
    const float cx = i00 * dx + i01 * dy, cy = i10 * dx + i11 * dy;
    const float rx = fabsf(i00) + fabsf(i01) + 1e-3f, ry = fabsf(i10) + fabsf(i11) + 1e-3f;
    const int ox0 = max(0, (int)ceilf(cx - rx)), ox1 = min(Wo - 1, (int)floorf(cx + rx));
    const int oy0 = max(0, (int)ceilf(cy - ry)), oy1 = min(Ho - 1, (int)floorf(cy + ry));
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) {
        float ix, iy;
        source_coords(th, ox, oy, Wo, Ho, Ws, Hs, ix, iy);  // the forward's own arithmetic
        // the forward splits at floor(): weight of integer pixel sx is 1 - |ix - sx| when it is one of
        // the two neighbours floor(ix), floor(ix) + 1
        const float fx = floorf(ix), fy = floorf(iy);
        const float wx = sx == (int)fx ? 1.f - (ix - fx) : (sx == (int)fx + 1 ? ix - fx : 0.f);
        const float wy = sy == (int)fy ? 1.f - (iy - fy) : (sy == (int)fy + 1 ? iy - fy : 0.f);
        const float w = wx * wy;
        if (w != 0.f) {
          float g[8];
          load8(gy + (((size_t)b * Ho + oy) * Wo + ox) * Cp, g);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += w * g[i];
        }
      }
  }
  store8(gx + (((size_t)b * Hs + sy) * Ws + sx) * Cp, acc);
  for (int c0 = 8; c0 < Cp; c0 += 8) {
    float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    store8(gx + (((size_t)b * Hs + sy) * Ws + sx) * Cp + c0, z);
  }
}

// adjoint of F.pad(mode="reflect") with margins (left m0x, top m0y; the right / bottom ones follow
// from the padded size): every source pixel gathers its centre image and up to one mirror image
// per side and axis (margins < size, so an index is mirrored at most once per side).
template <typename TI, typename TO>
__global__ __launch_bounds__(NT) void reflect_fold_kernel(const TI* __restrict__ gp, TO* __restrict__ gx, int H,
                                                          int W, int Hp, int Wp, int m0y, int m0x, int Cp) {
  const int b = blockIdx.z, yy = blockIdx.y, xx = blockIdx.x * NT + threadIdx.x;
  if (xx >= W) return;
  const int m1y = Hp - H - m0y, m1x = Wp - W - m0x;
  int ys[3], xs[3], ny = 0, nx = 0;
  ys[ny++] = yy + m0y;
  if (yy >= 1 && yy <= m0y) ys[ny++] = m0y - yy;
  if (yy >= H - 1 - m1y && yy <= H - 2) ys[ny++] = 2 * (H - 1) + m0y - yy;
  xs[nx++] = xx + m0x;
  if (xx >= 1 && xx <= m0x) xs[nx++] = m0x - xx;
  if (xx >= W - 1 - m1x && xx <= W - 2) xs[nx++] = 2 * (W - 1) + m0x - xx;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int c = 0; c < nx; ++c) {
        float v[8];
        load8(gp + (((size_t)b * Hp + ys[a]) * Wp + xs[c]) * Cp + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
      }
    store8(gx + (((size_t)b * H + yy) * W + xx) * Cp + c0, acc);
  }
}

// y[c] = sum_k M[b][c][k] x[k] + M[b][c][3]   (c, k < C; M is fp32 [B][3][4]); padding channels stay 0
template <typename T>
__global__ __launch_bounds__(NT) void colour_kernel(const T* __restrict__ x, const float* __restrict__ m,
                                                    T* __restrict__ y, long P, int C, int Cp) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * NT + threadIdx.x;
  if (p >= P) return;
  const float* mb = m + b * 12;
  float v[8], o[8];
  const size_t off = ((size_t)b * P + p) * Cp;
  load8(x + off, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  for (int c = 0; c < C; ++c) {
    float a = mb[c * 4 + 3];
    for (int k = 0; k < C; ++k) a += mb[c * 4 + k] * v[k];
    o[c] = a;
  }
  store8(y + off, o);
  for (int c0 = 8; c0 < Cp; c0 += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    store8(y + off + c0, o);
  }
}

#define ADA_DISPATCH_T(dtype, ...)                                       \
  if ((dtype) == O2M_BF16) { using T = unsigned short; __VA_ARGS__; }    \
  else if ((dtype) == O2M_F32) { using T = float; __VA_ARGS__; }         \
  else return O2M_ERR_BAD_ARG;

}  // namespace

extern "C" {

int o2m_ada_grid_sample(const void* x, const float* theta, void* y, int32_t B, int32_t Hs, int32_t Ws,
                        int32_t Ho, int32_t Wo, int32_t Cp, int32_t dtype, void* stream) {
  if (!x || !theta || !y || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || Cp <= 0 || (Cp & 7))
    return O2M_ERR_BAD_ARG;
  if (B > 65535 || Ho > 65535) return O2M_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((Wo + NT - 1) / NT), (unsigned)Ho, (unsigned)B);
  ADA_DISPATCH_T(dtype, hipLaunchKernelGGL(grid_sample_fwd_kernel<T>, grid, dim3(NT), 0, s, (const T*)x, theta,
                                           (T*)y, Hs, Ws, Ho, Wo, Cp));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_ada_grid_sample_bwd(const void* gy, const float* theta, void* gx, int32_t B, int32_t Hs, int32_t Ws,
                            int32_t Ho, int32_t Wo, int32_t Cp, int32_t dtype, void* stream) {
  if (!gy || !theta || !gx || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return O2M_ERR_BAD_ARG;
  if (Cp <= 0 || (Cp & 7)) return O2M_ERR_BAD_ARG;
  if (B > 65535 || Hs > 65535) return O2M_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((Ws + NT - 1) / NT), (unsigned)Hs, (unsigned)B);
  ADA_DISPATCH_T(dtype, hipLaunchKernelGGL(grid_sample_bwd_kernel<T>, grid, dim3(NT), 0, s, (const T*)gy, theta,
                                           (T*)gx, Hs, Ws, Ho, Wo, Cp));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_reflect_fold(const void* gpad, void* gx, int32_t B, int32_t H, int32_t W, int32_t Hp, int32_t Wp,
                     int32_t pad_top, int32_t pad_left, int32_t Cp, int32_t in_dtype, int32_t out_dtype,
                     void* stream) {
  if (!gpad || !gx || B <= 0 || H <= 0 || W <= 0 || Cp <= 0 || (Cp & 7)) return O2M_ERR_BAD_ARG;
  const int m1y = Hp - H - pad_top, m1x = Wp - W - pad_left;
  if (pad_top < 0 || pad_left < 0 || m1y < 0 || m1x < 0) return O2M_ERR_BAD_ARG;
  if (pad_top >= H || m1y >= H || pad_left >= W || m1x >= W) return O2M_ERR_BAD_ARG;  // single reflection only
  if (B > 65535 || H > 65535) return O2M_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((W + NT - 1) / NT), (unsigned)H, (unsigned)B);
  if (in_dtype == O2M_F32) {
    ADA_DISPATCH_T(out_dtype, hipLaunchKernelGGL((reflect_fold_kernel<float, T>), grid, dim3(NT), 0, s,
                                                 (const float*)gpad, (T*)gx, H, W, Hp, Wp, pad_top, pad_left, Cp));
  } else if (in_dtype == O2M_BF16) {
    ADA_DISPATCH_T(out_dtype, hipLaunchKernelGGL((reflect_fold_kernel<unsigned short, T>), grid, dim3(NT), 0, s,
                                                 (const unsigned short*)gpad, (T*)gx, H, W, Hp, Wp, pad_top,
                                                 pad_left, Cp));
  } else {
    return O2M_ERR_BAD_ARG;
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_ada_colour(const void* x, const float* m, void* y, int32_t B, int64_t P, int32_t C, int32_t Cp,
                   int32_t dtype, void* stream) {
  if (!x || !m || !y || B <= 0 || P <= 0 || (C != 1 && C != 3) || Cp < C || (Cp & 7)) return O2M_ERR_BAD_ARG;
  if (B > 65535) return O2M_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((P + NT - 1) / NT), (unsigned)B);
  ADA_DISPATCH_T(dtype, hipLaunchKernelGGL(colour_kernel<T>, grid, dim3(NT), 0, s, (const T*)x, m, (T*)y, (long)P, C,
                                           Cp));
  O2M_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
