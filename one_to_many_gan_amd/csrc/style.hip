// Style / demodulation algebra of the modulated conv (reference layers.py:145-161) as three
// small kernels instead of ~20 eager ops per call.  Everything here is B x C sized.
//
//   forward :  s[b,i] = cs * sum_j w[b,j] Ws[i,j] + bs[i]            (to_style, layers.py:138-140,148)
//              d[b,o] = rsqrt( sum_i Q[o,i] s[b,i]^2 + eps )          (demodulation, layers.py:156-161)
//   backward:  e[b,o]  = -(1/2) d^3 * dL/dd = -(1/2) d[b,o]^2 * (S1[b,o] - bias[o] S0[b,o])
//              gs[b,i] = dots[b,i] + 2 s[b,i] sum_o e[b,o] Q[o,i]
//              gw[b,j] = cs sum_i gs[b,i] Ws[i,j]
//              gWs[i,j] = cs sum_b gs[b,i] w[b,j] ;  gbs[i] = sum_b gs[b,i]
//              gq[o,i] += sum_b e[b,o] s[b,i]^2        (-> dL/dQ, folded into the weight gradient
//                                                         by o2m_wgrad_finalize)
// with S0 = sum_p gu, S1 = sum_p gu*(y-residual) from o2m_act_bwd_reduce and
// dots = sum_p gxm*x from o2m_fold_scale_dot.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXWD = 16;

__global__ __launch_bounds__(NT) void style_fwd_kernel(const float* w, const float* Ws, const float* bs,
                                                       const float* Qt, float* s, float* d, int WD,
                                                       int Ci, int Cip, int Cop, float cs, float eps) {
  extern __shared__ float s2[];  // [Cip]
  const int b = blockIdx.x;
  float wv[MAXWD];
#pragma unroll
  for (int j = 0; j < MAXWD; ++j) wv[j] = j < WD ? w[(size_t)b * WD + j] : 0.f;
  for (int i = threadIdx.x; i < Cip; i += NT) {
    float v = 0.f;
    if (i < Ci) {
      for (int j = 0; j < WD; ++j) v += wv[j] * Ws[(size_t)i * WD + j];
      v = v * cs + bs[i];
    }
    s[(size_t)b * Cip + i] = v;
    s2[i] = v * v;
  }
  if (!d) return;
  __syncthreads();
  for (int o = threadIdx.x; o < Cop; o += NT) {
    float a = eps;
    for (int i = 0; i < Cip; ++i) a += Qt[(size_t)i * Cop + o] * s2[i];  // lanes over o: coalesced
    d[(size_t)b * Cop + o] = rsqrtf(a);
  }
}

// per sample: e, gs, gw
__global__ __launch_bounds__(NT) void style_bwd_sample_kernel(
    const float* sums, const float* bias, const float* dots, const float* s, const float* d, const float* Q,
    const float* Ws, float* e, float* gs, float* gw, int WD, int Ci, int Cip, int Cop, float cs) {
  extern __shared__ float sm[];  // e[Cop] then per-wave partials
  float* es = sm;
  float* part = sm + Cop;  // [NT/64][MAXWD]
  const int b = blockIdx.x;
  if (d) {
    for (int o = threadIdx.x; o < Cop; o += NT) {
      const float s0 = sums[((size_t)b * 2) * Cop + o], s1 = sums[((size_t)b * 2 + 1) * Cop + o];
      const float dd = d[(size_t)b * Cop + o];
      const float ev = -0.5f * dd * dd * (s1 - (bias ? bias[o] * s0 : 0.f));
      es[o] = ev;
      e[(size_t)b * Cop + o] = ev;
    }
  }
  __syncthreads();
  float acc[MAXWD];
#pragma unroll
  for (int j = 0; j < MAXWD; ++j) acc[j] = 0.f;
  for (int i = threadIdx.x; i < Cip; i += NT) {
    float g = dots[(size_t)b * Cip + i];
    if (d) {
      float t = 0.f;
      for (int o = 0; o < Cop; ++o) t += es[o] * Q[(size_t)o * Cip + i];  // lanes over i: coalesced
      g += 2.f * s[(size_t)b * Cip + i] * t;
    }
    if (i >= Ci) g = 0.f;
    gs[(size_t)b * Cip + i] = g;
    if (i < Ci)
      for (int j = 0; j < WD; ++j) acc[j] += g * Ws[(size_t)i * WD + j];
  }
#pragma unroll
  for (int j = 0; j < MAXWD; ++j) acc[j] = wave_sum(acc[j]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int j = 0; j < MAXWD; ++j) part[(threadIdx.x >> 6) * MAXWD + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < WD) {
    float t = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) t += part[wv * MAXWD + threadIdx.x];
    gw[(size_t)b * WD + threadIdx.x] = t * cs;
  }
}

// over the batch: gWs, gbs (block 0 .. ) and gq rows
__global__ __launch_bounds__(NT) void style_bwd_batch_kernel(const float* gs, const float* w, const float* e,
                                                             const float* s, float* gWs, float* gbs, float* gq,
                                                             int B, int WD, int Ci, int Cip, int Cop, float cs,
                                                             int param_blocks) {
  if ((int)blockIdx.x < param_blocks) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= Ci) return;
    float acc[MAXWD], sb = 0.f;
#pragma unroll
    for (int j = 0; j < MAXWD; ++j) acc[j] = 0.f;
    for (int b = 0; b < B; ++b) {
      const float g = gs[(size_t)b * Cip + i];
      sb += g;
      for (int j = 0; j < WD; ++j) acc[j] += g * w[(size_t)b * WD + j];
    }
    gbs[i] = sb;
    for (int j = 0; j < WD; ++j) gWs[(size_t)i * WD + j] = acc[j] * cs;
    return;
  }
  if (!gq) return;
  const int o = blockIdx.x - param_blocks;  // one block per output channel row of gq
  for (int i = threadIdx.x; i < Cip; i += NT) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) {
      const float sv = s[(size_t)b * Cip + i];
      a += e[(size_t)b * Cop + o] * sv * sv;
    }
    gq[(size_t)o * Cip + i] += a;  // accumulates over the uses of the layer in one backward
  }
}

}  // namespace

extern "C" {

int o2m_style_fwd(const float* w, const float* Ws, const float* bs, const float* Qt, float* s, float* d,
                  int32_t B, int32_t WD, int32_t Ci, int32_t Cip, int32_t Cop, float cs, float eps,
                  void* stream) {
  if (!w || !Ws || !bs || !s || B <= 0 || WD <= 0 || WD > MAXWD || Ci <= 0 || Cip < Ci) return O2M_ERR_BAD_ARG;
  if (d && (!Qt || Cop <= 0)) return O2M_ERR_BAD_ARG;
  hipLaunchKernelGGL(style_fwd_kernel, dim3(B), dim3(NT), Cip * sizeof(float), static_cast<hipStream_t>(stream),
                     w, Ws, bs, Qt, s, d, WD, Ci, Cip, Cop, cs, eps);
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_style_bwd(const float* sums, const float* bias, const float* dots, const float* s, const float* d,
                  const float* Q, const float* w, const float* Ws, float* e, float* gs, float* gw,
                  float* gWs, float* gbs, float* gq, int32_t B, int32_t WD, int32_t Ci, int32_t Cip,
                  int32_t Cop, float cs, void* stream) {
  if (!dots || !s || !w || !Ws || !gs || !gw || !gWs || !gbs) return O2M_ERR_BAD_ARG;
  if (B <= 0 || WD <= 0 || WD > MAXWD || Ci <= 0 || Cip < Ci || Cop <= 0) return O2M_ERR_BAD_ARG;
  if (d && (!sums || !Q || !e || !gq)) return O2M_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = (Cop + (NT / 64) * MAXWD) * sizeof(float);
  hipLaunchKernelGGL(style_bwd_sample_kernel, dim3(B), dim3(NT), lds, st, sums, bias, dots, s, d, Q, Ws, e, gs,
                     gw, WD, Ci, Cip, Cop, cs);
  O2M_LAUNCH_CHECK();
  const int pb = (Ci + NT - 1) / NT;
  hipLaunchKernelGGL(style_bwd_batch_kernel, dim3(pb + (d ? Cop : 0)), dim3(NT), 0, st, gs, w, e, s, gWs, gbs,
                     d ? gq : nullptr, B, WD, Ci, Cip, Cop, cs, pb);
  O2M_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
