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

// grid (B, ceil(Cop/64)): every block recomputes the (cheap) style vector of its sample into
// LDS; its 256 threads then cover 64 output channels x 4 slices of the reduction over i.
__global__ __launch_bounds__(NT) void style_fwd_kernel(const float* w, const float* Ws, const float* bs,
                                                       const float* Qt, float* s, float* d, int WD,
                                                       int Ci, int Cip, int Cop, float cs, float eps) {
  extern __shared__ float sm[];  // s2[Cip] then part[4][64]
  float* s2 = sm;
  float* part = sm + Cip;
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
    if (blockIdx.y == 0) s[(size_t)b * Cip + i] = v;
    s2[i] = v * v;
  }
  if (!d) return;
  __syncthreads();
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int o = blockIdx.y * 64 + ol;
  const int per = (Cip + 3) / 4, i0 = sl * per, i1 = min(Cip, i0 + per);
  float a = 0.f;
  if (o < Cop) {
#pragma unroll 16
    for (int i = i0; i < i1; ++i) a += Qt[(size_t)i * Cop + o] * s2[i];  // lanes over o: coalesced
  }
  part[sl * 64 + ol] = a;
  __syncthreads();
  if (sl == 0 && o < Cop)
    d[(size_t)b * Cop + o] = rsqrtf(eps + part[ol] + part[64 + ol] + part[128 + ol] + part[192 + ol]);
}

// grid (B, ceil(Cip/64)): e (recomputed per block into LDS), then 64 input channels x 4 slices
// of the reduction over o.  (gw, the gradient of the latent, is formed by style_bwd_batch_kernel from the
// finished gs rows: one block per sample, fixed summation order, no atomics.)
__global__ __launch_bounds__(NT) void style_bwd_sample_kernel(
    const float* sums, const float* bias, const float* dots, const float* s, const float* d, const float* Q,
    float* e, float* gs, int Ci, int Cip, int Cop) {
  extern __shared__ float sm[];  // es[Cop], part[4][64]
  float* es = sm;
  float* part = sm + Cop;
  const int b = blockIdx.x;
  if (d) {
    for (int o = threadIdx.x; o < Cop; o += NT) {
      const float s0 = sums[((size_t)b * 2) * Cop + o], s1 = sums[((size_t)b * 2 + 1) * Cop + o];
      const float dd = d[(size_t)b * Cop + o];
      const float ev = -0.5f * dd * dd * (s1 - (bias ? bias[o] * s0 : 0.f));
      es[o] = ev;
      if (blockIdx.y == 0) e[(size_t)b * Cop + o] = ev;
    }
  }
  __syncthreads();
  const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.y * 64 + il;
  float t = 0.f;
  if (d && i < Cip) {
    const int per = (Cop + 3) / 4, o0 = sl * per, o1 = min(Cop, o0 + per);
#pragma unroll 16
    for (int o = o0; o < o1; ++o) t += es[o] * Q[(size_t)o * Cip + i];  // lanes over i: coalesced
  }
  part[sl * 64 + il] = t;
  __syncthreads();
  if (sl == 0) {
    float g = 0.f;
    if (i < Ci) {
      g = dots[(size_t)b * Cip + i];
      if (d) g += 2.f * s[(size_t)b * Cip + i] * (part[il] + part[64 + il] + part[128 + il] + part[192 + il]);
    }
    if (i < Cip) gs[(size_t)b * Cip + i] = g;
  }
}

// blocks [0, param_blocks): gWs, gbs (over the batch); then B blocks: gw[b][:] = cs * gs[b][:] Ws; then the gq rows
__global__ __launch_bounds__(NT) void style_bwd_batch_kernel(const float* gs, const float* w, const float* e,
                                                             const float* s, const float* Ws, float* gWs, float* gbs,
                                                             float* gw, float* gq, int B, int WD, int Ci, int Cip,
                                                             int Cop, float cs, int param_blocks, int accumulate) {
  if ((int)blockIdx.x >= param_blocks && (int)blockIdx.x < param_blocks + B) {
    __shared__ float red[NT][MAXWD];
    const int b = blockIdx.x - param_blocks;
    float a[MAXWD];
#pragma unroll
    for (int j = 0; j < MAXWD; ++j) a[j] = 0.f;
    for (int i = threadIdx.x; i < Ci; i += NT) {
      const float g = gs[(size_t)b * Cip + i];
#pragma unroll
      for (int j = 0; j < MAXWD; ++j)
        if (j < WD) a[j] += g * Ws[(size_t)i * WD + j];
    }
#pragma unroll
    for (int j = 0; j < MAXWD; ++j) red[threadIdx.x][j] = a[j];
    __syncthreads();
    if ((int)threadIdx.x < WD) {
      float t = 0.f;
      for (int q = 0; q < NT; ++q) t += red[q][threadIdx.x];
      gw[(size_t)b * WD + threadIdx.x] = t * cs;
    }
    return;
  }
  if ((int)blockIdx.x < param_blocks) {
    // gWs[i][:] = cs * sum_b gs[b][i] * w[b][:], gbs[i] = sum_b gs[b][i]: 32 input channels x 8 slices of the batch per
    // block, the slice sums added in slice order.  (One thread per channel walking the whole batch with a run-time
    // inner loop over WD was a chain of B x WD dependent scalar loads: ~30 us on the launch's critical path.)
    __shared__ float pr[8][32][MAXWD + 1];
    const int il = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + il;
    float acc[MAXWD], sb = 0.f;
#pragma unroll
    for (int j = 0; j < MAXWD; ++j) acc[j] = 0.f;
    if (i < Ci) {
      for (int b = sl; b < B; b += 8) {
        const float gv = gs[(size_t)b * Cip + i];
        sb += gv;
#pragma unroll
        for (int j = 0; j < MAXWD; ++j)
          if (j < WD) acc[j] += gv * w[(size_t)b * WD + j];
      }
    }
#pragma unroll
    for (int j = 0; j < MAXWD; ++j) pr[sl][il][j] = acc[j];
    pr[sl][il][MAXWD] = sb;
    __syncthreads();
    // thread (il, j = sl .. step 8): one output element each
    if (i < Ci) {
      for (int j = sl; j <= WD; j += 8) {
        const int jj = j == WD ? MAXWD : j;  // slot MAXWD = the bias sum
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += pr[k][il][jj];
        // accumulate = 1 adds into the caller's buffers (the parameters' .grad).  One writer per element WITHIN a
        // launch; fp32 atomics because two launches for the same layer may run at the same time on two streams (the
        // decode and the extraction group of generator_step, core/training.py)
        float* dst = j == WD ? gbs + i : gWs + (size_t)i * WD + j;
        const float v = j == WD ? t : t * cs;
        if (accumulate) atomicAdd(dst, v);
        else *dst = v;
      }
    }
    return;
  }
  if (!gq) return;
  // dL/dQ[o][i] += sum_b e[b][o] * s[b][i]^2: a block takes EIGHT rows o (their e values in LDS), so a thread squares
  // s[b][i] once per sample for eight accumulators -- the one-row-per-block form re-read the whole s table per row
  // (2 B loads per thread and row in six dependent batches: most of this launch's 26 us)
  __shared__ float es[8][64];
  const int o0 = (blockIdx.x - param_blocks - B) * 8;
  float a[2][8];  // (Cip <= 2 NT: checked by the launcher)
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 8; ++r) a[k][r] = 0.f;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = min(64, B - b0);
    __syncthreads();
    for (int t = threadIdx.x; t < 8 * nb; t += NT) {
      const int r = t / nb, bb = t - r * nb;
      es[r][bb] = o0 + r < Cop ? e[(size_t)(b0 + bb) * Cop + o0 + r] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = threadIdx.x + k * NT;
      if (i >= Cip) continue;
#pragma unroll 4
      for (int bb = 0; bb < nb; ++bb) {
        const float sv = s[(size_t)(b0 + bb) * Cip + i], s2 = sv * sv;
#pragma unroll
        for (int r = 0; r < 8; ++r) a[k][r] += es[r][bb] * s2;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i >= Cip) continue;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (o0 + r < Cop) atomicAdd(gq + (size_t)(o0 + r) * Cip + i, a[k][r]);  // accumulates over the uses of the layer (see above)
  }
}

}  // namespace

extern "C" {

int o2m_style_fwd(const float* w, const float* Ws, const float* bs, const float* Qt, float* s, float* d,
                  int32_t B, int32_t WD, int32_t Ci, int32_t Cip, int32_t Cop, float cs, float eps,
                  void* stream) {
  if (!w || !Ws || !bs || !s || B <= 0 || WD <= 0 || WD > MAXWD || Ci <= 0 || Cip < Ci) return O2M_ERR_BAD_ARG;
  if (d && (!Qt || Cop <= 0)) return O2M_ERR_BAD_ARG;
  hipLaunchKernelGGL(style_fwd_kernel, dim3(B, d ? (Cop + 63) / 64 : 1), dim3(NT), (Cip + 256) * sizeof(float),
                     static_cast<hipStream_t>(stream), w, Ws, bs, Qt, s, d, WD, Ci, Cip, Cop, cs, eps);
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_style_bwd(const float* sums, const float* bias, const float* dots, const float* s, const float* d,
                  const float* Q, const float* w, const float* Ws, float* e, float* gs, float* gw,
                  float* gWs, float* gbs, float* gq, int32_t B, int32_t WD, int32_t Ci, int32_t Cip,
                  int32_t Cop, float cs, int32_t accumulate, void* stream) {
  if (!dots || !s || !w || !Ws || !gs || !gw || !gWs || !gbs) return O2M_ERR_BAD_ARG;
  if (B <= 0 || WD <= 0 || WD > MAXWD || Ci <= 0 || Cip < Ci || Cop <= 0) return O2M_ERR_BAD_ARG;
  if (d && (!sums || !Q || !e || !gq)) return O2M_ERR_BAD_ARG;
  if (d && Cip > 2 * NT) return O2M_ERR_UNSUPPORTED;  // (the dL/dQ rows: two input channels per thread)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = (Cop + 256) * sizeof(float);
  hipLaunchKernelGGL(style_bwd_sample_kernel, dim3(B, (Cip + 63) / 64), dim3(NT), lds, st, sums, bias, dots, s, d,
                     Q, e, gs, Ci, Cip, Cop);
  O2M_LAUNCH_CHECK();
  const int pb = (Ci + 31) / 32;
  hipLaunchKernelGGL(style_bwd_batch_kernel, dim3(pb + B + (d ? (Cop + 7) / 8 : 0)), dim3(NT), 0, st, gs, w, e, s, Ws, gWs, gbs,
                     gw, d ? gq : nullptr, B, WD, Ci, Cip, Cop, cs, pb, accumulate);
  O2M_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
