// HBM-bound kernels of the training step (gfx950): every one streams NHWC tensors with 16-B
// (bf16) / 32-B (fp32) per-lane accesses, lanes running over channel vectors so that a wave
// touches whole contiguous pixel rows; per-(sample,channel) statistics are reduced
// in registers over a pixel chunk, then across the block's pixel lanes through LDS.
#include "common.h"

namespace {

constexpr int NT = 256;

// ---- geometry shared by the per-(b,c) reductions ----------------------------------------
// A block covers one pixel chunk of one sample.  Thread t handles channel vector t % CV and
// pixel lane t / CV (PL = NT / CV lanes); pixels are strided by PL inside the chunk.
struct ChanGeom {
  int CV, PL, chunk, nchunks;
};
__host__ __device__ inline ChanGeom chan_geom(int B, int P, int C) {
  ChanGeom g;
  g.CV = C / 8;
  g.PL = g.CV >= NT ? 1 : NT / g.CV;
  // aim for ~1024 blocks overall (4 per CU), at least 8 pixels per pixel lane
  int want = (1024 + B - 1) / B;
  int maxc = (P + g.PL * 8 - 1) / (g.PL * 8);
  g.nchunks = want < maxc ? want : maxc;
  if (g.nchunks < 1) g.nchunks = 1;
  g.chunk = (P + g.nchunks - 1) / g.nchunks;
  g.nchunks = (P + g.chunk - 1) / g.chunk;
  return g;
}

// reduce NV values per thread across the block's pixel lanes; result valid on pl == 0
template <int NV>
__device__ __forceinline__ void lanes_reduce(float (&v)[NV], int cv, int pl, int CV, int PL,
                                             float* sm) {
  // sm: [PL][CV][NV]
  if (cv < CV && pl < PL) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[(pl * CV + cv) * NV + i] = v[i];
  }
  __syncthreads();
  if (pl == 0 && cv < CV) {
    for (int q = 1; q < PL; ++q)
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += sm[(q * CV + cv) * NV + i];
  }
}

// ---- act backward + per-(b,c) sums --------------------------------------------------------
// HAS_Y / HAS_RES: which operands exist is a property of the launch; as run-time pointer tests inside the
// 8-element loop they compiled into two uniform branches per ELEMENT.
template <typename T, bool HAS_Y, bool HAS_RES>
__global__ __launch_bounds__(NT) void act_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                            const T* __restrict__ res,
                                                            const float* __restrict__ out_mul,
                                                            T* __restrict__ gu, float* __restrict__ sums,
                                                            float* __restrict__ partials, int P, int C, int act,
                                                            ChanGeom gm) {
  extern __shared__ float sm[];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int cv = threadIdx.x % gm.CV, pl = threadIdx.x / gm.CV;
  const int pe = min(P, (ch + 1) * gm.chunk);
  const bool big = (size_t)gridDim.y * P * C * sizeof(T) >= kStreamBytes;  // touched once: stream (common.h)
  float acc[16], mul[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) mul[i] = out_mul ? out_mul[(size_t)b * C + cv * 8 + i] : 1.f;
  if (pl < gm.PL) {
#pragma unroll 4
    for (int p = ch * gm.chunk + pl; p < pe; p += gm.PL) {
      const size_t o = ((size_t)b * P + p) * C + cv * 8;
      float gv[8], yv[8], rv[8], ov[8];
      load8x(g + o, gv, big);
      if constexpr (HAS_Y) load8x(y + o, yv, big);
      if constexpr (HAS_RES) load8x(res + o, rv, big);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // no y: reduce-only call (identity activation), sums[.][1] stays 0
        const float u = HAS_Y ? (HAS_RES ? yv[i] - rv[i] : yv[i]) : 0.f;
        const float d = HAS_Y ? gv[i] * act_bwd_from_out(u, act) : gv[i];
        ov[i] = d * mul[i];
        acc[i] += d;
        acc[8 + i] += d * u;
      }
      if (gu) store8x(gu + o, ov, big);
    }
  }
  lanes_reduce<16>(acc, cv, pl, gm.CV, gm.PL, sm);
  // coalesce: [2][C] in LDS, then consecutive lanes add consecutive floats (256-B atomics)
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[cv * 8 + i] = acc[i]; sm[C + cv * 8 + i] = acc[8 + i]; }
  }
  __syncthreads();
  // deterministic mode: this chunk's row goes to the workspace (one writer per element) and
  // chunk_sum_kernel adds the rows in chunk order; else fp32 atomics straight into sums
  if (partials) {
    for (int t = threadIdx.x; t < 2 * C; t += NT) partials[((size_t)b * gridDim.x + ch) * 2 * C + t] = sm[t];
  } else {
    for (int t = threadIdx.x; t < 2 * C; t += NT) atomicAdd(sums + (size_t)b * 2 * C + t, sm[t]);
  }
}

// out[b][t] += sum over the chunks, in chunk order (fixed summation order: bitwise reproducible)
__global__ __launch_bounds__(NT) void chunk_sum_kernel(float* __restrict__ out, const float* __restrict__ partials,
                                                       int nchunks, int n) {
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < n; t += NT) {
    float a = out[(size_t)b * n + t];
    for (int c = 0; c < nchunks; ++c) a += partials[((size_t)b * nchunks + c) * n + t];
    out[(size_t)b * n + t] = a;
  }
}

// ---- reflect-pad backward (fold) + style scale + style dot (+ residual gradient, + next activation) --------
// FUSE_ACT: the tensor this kernel would store is the gradient of an activation output y == x (the layer below
// applied act and a demodulation d, and its output IS this layer's input x): instead of storing gx and having
// o2m_act_bwd_reduce read it and x again, the kernel stores gu = gx * act'(x) * act_mul[b,c] and adds that layer's
// sums {sum gx act'(x), sum gx act'(x) x} -- the ModulatedResnetBlock's conv2 -> ReLU -> conv1 backward chain in
// one pass (4 tensor passes instead of 7).
template <typename T, bool FUSE_ACT>
__global__ __launch_bounds__(NT) void fold_scale_dot_kernel(const T* __restrict__ gpad, const T* __restrict__ x,
                                                            const float* __restrict__ scale,
                                                            T* __restrict__ gx, float* __restrict__ dots,
                                                            T* __restrict__ xmod, const T* __restrict__ gres,
                                                            const float* __restrict__ act_mul,
                                                            float* __restrict__ act_sums, float* __restrict__ partials,
                                                            int H, int W, int C, int pad, int act, ChanGeom gm) {
  extern __shared__ float sm[];
  constexpr int NV = FUSE_ACT ? 24 : 8;  // accumulators per thread: dots (+ the two activation sums)
  const int b = blockIdx.y, ch = blockIdx.x;
  const int cv = threadIdx.x % gm.CV, pl = threadIdx.x / gm.CV;
  const int P = H * W, Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int pe = min(P, (ch + 1) * gm.chunk);
  const bool big = (size_t)gridDim.y * P * C * sizeof(T) >= kStreamBytes;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  float sc[8], am[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale ? scale[(size_t)b * C + cv * 8 + i] : 1.f;
    am[i] = (FUSE_ACT && act_mul) ? act_mul[(size_t)b * C + cv * 8 + i] : 1.f;
  }
  if (pl < gm.PL) {
#pragma unroll 2
    for (int p = ch * gm.chunk + pl; p < pe; p += gm.PL) {
      const int yy = p / W, xx = p - yy * W;
      int ys[3], xs[3], ny = 0, nx = 0;
      ys[ny++] = yy + pad;
      if (yy >= 1 && yy <= pad) ys[ny++] = pad - yy;
      if (yy >= H - 1 - pad && yy <= H - 2) ys[ny++] = 2 * H - 2 + pad - yy;
      xs[nx++] = xx + pad;
      if (xx >= 1 && xx <= pad) xs[nx++] = pad - xx;
      if (xx >= W - 1 - pad && xx <= W - 2) xs[nx++] = 2 * W - 2 + pad - xx;
      float f[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = 0.f;
      for (int a = 0; a < ny; ++a)
        for (int c = 0; c < nx; ++c) {
          float t[8];
          load8x(gpad + (((size_t)b * Hp + ys[a]) * Wp + xs[c]) * C + cv * 8, t, big);
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += t[i];
        }
      const size_t o = ((size_t)b * P + p) * C + cv * 8;
      float xv[8];
      if (dots || FUSE_ACT) load8x(x + o, xv, big);
      if (dots) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += f[i] * xv[i];
        if (xmod) {  // by-product for the weight gradient: the modulated input x * s
          float xm[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) xm[i] = xv[i] * sc[i];
          store8x(xmod + o, xm, big);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] *= sc[i];
      if (gres) {  // gradient arriving through a residual connection around the layer (added, not scaled)
        float rv[8];
        load8x(gres + o, rv, big);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += rv[i];
      }
      if constexpr (FUSE_ACT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dl = f[i] * act_bwd_from_out(xv[i], act);
          acc[8 + i] += dl;
          acc[16 + i] += dl * xv[i];
          f[i] = dl * am[i];
        }
      }
      store8x(gx + o, f, big);
    }
  }
  if (dots || FUSE_ACT) {
    lanes_reduce<NV>(acc, cv, pl, gm.CV, gm.PL, sm);
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sm[cv * 8 + i] = acc[i];
        if constexpr (FUSE_ACT) { sm[C + cv * 8 + i] = acc[8 + i]; sm[2 * C + cv * 8 + i] = acc[16 + i]; }
      }
    }
    __syncthreads();
    constexpr int ROWS = FUSE_ACT ? 3 : 1;  // [dots | sums0 | sums1]
    if (partials) {
      for (int t = threadIdx.x; t < ROWS * C; t += NT) partials[((size_t)b * gridDim.x + ch) * ROWS * C + t] = sm[t];
    } else {
      for (int t = threadIdx.x; t < ROWS * C; t += NT) {
        if (t < C) { if (dots) atomicAdd(dots + (size_t)b * C + t, sm[t]); }
        else atomicAdd(act_sums + (size_t)b * 2 * C + (t - C), sm[t]);
      }
    }
  }
}

// deterministic mode of the fused form: partial rows [dots | sums0 | sums1] -> dots [B][C], sums [B][2][C]
__global__ __launch_bounds__(NT) void chunk_sum3_kernel(float* __restrict__ dots, float* __restrict__ sums,
                                                        const float* __restrict__ partials, int nchunks, int C) {
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < 3 * C; t += NT) {
    float* dst = t < C ? (dots ? dots + (size_t)b * C + t : nullptr) : sums + (size_t)b * 2 * C + (t - C);
    if (!dst) continue;
    float a = *dst;
    for (int c = 0; c < nchunks; ++c) a += partials[((size_t)b * nchunks + c) * 3 * C + t];
    *dst = a;
  }
}

// ---- instance norm ---------------------------------------------------------------------------
// MODE 0: {sum x, sum x^2};  MODE 1 (backward): {sum gh, sum gh*xh}
// Banded operator applied to the incoming gradient on the fly (InstanceNorm backward behind a DownSample: the
// gradient of the normalised map is the TRANSPOSED down-sampling of the low-resolution gradient, two taps per axis;
// it is gathered per pixel instead of being stored and read twice).
struct GatherTaps {
  const int* sy;
  const float* wy;
  const int* sx;
  const float* wx;
  int W, Hl, Wl;  // width of the fine map; size of the coarse gradient
};
// The four tap tables of a launch in LDS (12 KB): a pixel's gather is then table reads from LDS -> four independent
// coarse loads, instead of a dependent global table load -> address -> data chain per pixel (the gathering passes ran
// at 1.5-1.9 TB/s of their tensors, profiles/r03_pointwise_bw.txt).  Maps beyond 512 pixels a side keep the global tables.
constexpr int kTapCap = 512;
struct TapTables {
  int sy[kTapCap], sx[kTapCap];
  float wy[2 * kTapCap], wx[2 * kTapCap];
};
__device__ __forceinline__ GatherTaps stage_taps(const GatherTaps& tp, TapTables& t, int H) {
  if (H > kTapCap || tp.W > kTapCap) return tp;  // (uniform)
  for (int i = threadIdx.x; i < H; i += blockDim.x) { t.sy[i] = tp.sy[i]; t.wy[2 * i] = tp.wy[2 * i]; t.wy[2 * i + 1] = tp.wy[2 * i + 1]; }
  for (int i = threadIdx.x; i < tp.W; i += blockDim.x) { t.sx[i] = tp.sx[i]; t.wx[2 * i] = tp.wx[2 * i]; t.wx[2 * i + 1] = tp.wx[2 * i + 1]; }
  __syncthreads();
  GatherTaps l = tp;
  l.sy = t.sy; l.wy = t.wy; l.sx = t.sx; l.wx = t.wx;
  return l;
}
template <typename T, int TG>
__device__ __forceinline__ void gather_grad(const T* __restrict__ gl, const GatherTaps& tp, int b, int yy, int xx, int C,
                                            int cv, float (&gv)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) gv[i] = 0.f;
  const int y0 = tp.sy[yy], x0 = tp.sx[xx];
#pragma unroll
  for (int ty = 0; ty < TG; ++ty) {
    const float a = tp.wy[yy * TG + ty];
#pragma unroll
    for (int tx = 0; tx < TG; ++tx) {
      const float w = a * tp.wx[xx * TG + tx];
      float v[8];
      load8(gl + (((size_t)b * tp.Hl + min(y0 + ty, tp.Hl - 1)) * tp.Wl + min(x0 + tx, tp.Wl - 1)) * C + cv * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) gv[i] += w * v[i];
    }
  }
}

template <typename T, int MODE, int TG = 0>
__global__ __launch_bounds__(NT) void in_partial_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                        const float* __restrict__ mean_rstd,
                                                        float* __restrict__ partial,
                                                        int P, int C, int act, ChanGeom gm, GatherTaps tp_in = GatherTaps{}) {
  extern __shared__ float sm[];
  __shared__ TapTables taps_lds[1];
  const GatherTaps tp = TG > 0 ? stage_taps(tp_in, taps_lds[0], P / tp_in.W) : tp_in;
  const int b = blockIdx.y, ch = blockIdx.x;
  const int cv = threadIdx.x % gm.CV, pl = threadIdx.x / gm.CV;
  const int pe = min(P, (ch + 1) * gm.chunk);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float mu[8], rs[8];
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mu[i] = mean_rstd[((size_t)b * C + cv * 8 + i) * 2];
      rs[i] = mean_rstd[((size_t)b * C + cv * 8 + i) * 2 + 1];
    }
  }
  if (pl < gm.PL) {
    int yy = 0, xx = 0;
    if constexpr (TG > 0) {
      const int p0 = ch * gm.chunk + pl;
      yy = p0 / tp.W;
      xx = p0 - yy * tp.W;
    }
#pragma unroll 4
    for (int p = ch * gm.chunk + pl; p < pe; p += gm.PL) {
      const size_t o = ((size_t)b * P + p) * C + cv * 8;
      float xv[8];
      load8(x + o, xv);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] += xv[i]; acc[8 + i] += xv[i] * xv[i]; }
      } else {
        float gv[8];
        if constexpr (TG > 0) {
          gather_grad<T, TG>(g, tp, b, yy, xx, C, cv, gv);
          xx += gm.PL;
          while (xx >= tp.W) { xx -= tp.W; ++yy; }
        } else {
          load8(g + o, gv);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float xh = (xv[i] - mu[i]) * rs[i];
          float gh = gv[i] * act_bwd_from_out(xh, act);  // relu/lrelu: sign(xh) == sign(out)
          acc[i] += gh;
          acc[8 + i] += gh * xh;
        }
      }
    }
  }
  lanes_reduce<16>(acc, cv, pl, gm.CV, gm.PL, sm);
  if (pl == 0) {
    float* dst = partial + (((size_t)b * gm.nchunks + ch) * C + cv * 8) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dst[2 * i] = acc[i]; dst[2 * i + 1] = acc[8 + i]; }
  }
}

// MODE 0: -> {mean, rstd};  MODE 1: -> {mean gh, mean gh*xh}
// 256 threads = 32 (sample, channel) pairs x 8 slices of the chunk list: a thread adds every 8th chunk's pair (8-byte
// loads, 32 consecutive channels = 256 B per slice and chunk), the 8 slice sums meet in LDS in slice order -- a fixed
// summation order as before, with an 8 times shorter dependent chain (the one-thread-per-pair form walked up to 1024
// chunks serially: 13 us per launch, 51 launches per step).
template <int MODE>
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* partial, float* out, int B, int P, int C,
                                                          int nchunks, float eps) {
  __shared__ float red[8][32][2];
  const int pi = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + pi;
  float s0 = 0.f, s1 = 0.f;
  if (idx < B * C) {
    const int b = idx / C, c = idx - b * C;
    const float* p = partial + (((size_t)b * nchunks + sl) * C + c) * 2;
    const size_t step = (size_t)8 * C * 2;
#pragma unroll 4
    for (int ch = sl; ch < nchunks; ch += 8, p += step) {
      const float2 v = *reinterpret_cast<const float2*>(p);
      s0 += v.x;
      s1 += v.y;
    }
  }
  red[sl][pi][0] = s0;
  red[sl][pi][1] = s1;
  __syncthreads();
  if (sl != 0 || idx >= B * C) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) { s0 += red[k][pi][0]; s1 += red[k][pi][1]; }
  const float inv = 1.f / (float)P;
  if (MODE == 0) {
    float mean = s0 * inv;
    float var = fmaxf(s1 * inv - mean * mean, 0.f);
    out[(size_t)idx * 2] = mean;
    out[(size_t)idx * 2 + 1] = rsqrtf(var + eps);
  } else {
    out[(size_t)idx * 2] = s0 * inv;
    out[(size_t)idx * 2 + 1] = s1 * inv;
  }
}

// MODE 0: y = act((x-mean)*rstd) + res ;  MODE 1: gx = rstd*(gh - m1 - xh*m2)
// Same block geometry as the reductions (one pixel chunk of one sample per block, a thread keeps
// its 8 channels): the per-(b,c) statistics sit in registers and the pixel walk needs no integer
// division; four pixels are in flight per thread.
template <typename T, int MODE, int TG = 0>
__global__ __launch_bounds__(NT) void in_apply_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                      const float* __restrict__ mean_rstd,
                                                      const float* __restrict__ gsums,
                                                      const T* __restrict__ res, T* __restrict__ out,
                                                      int P, int C, int act, ChanGeom gm, GatherTaps tp_in = GatherTaps{}) {
  __shared__ TapTables taps_lds[1];
  const GatherTaps tp = TG > 0 ? stage_taps(tp_in, taps_lds[0], P / tp_in.W) : tp_in;
  const int b = blockIdx.y, ch = blockIdx.x;
  const int cv = threadIdx.x % gm.CV, pl = threadIdx.x / gm.CV;
  if (pl >= gm.PL) return;
  const int pe = min(P, (ch + 1) * gm.chunk);
  float mean[8], rstd[8], m1[8], m2[8];
  {
    const float* mr = mean_rstd + ((size_t)b * C + cv * 8) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { mean[i] = mr[2 * i]; rstd[i] = mr[2 * i + 1]; m1[i] = m2[i] = 0.f; }
    if (MODE == 1) {
      const float* gs = gsums + ((size_t)b * C + cv * 8) * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) { m1[i] = gs[2 * i]; m2[i] = gs[2 * i + 1]; }
    }
  }
  constexpr int U = 4;
  for (int p0 = ch * gm.chunk + pl; p0 < pe; p0 += U * gm.PL) {
    float xv[U][8], sv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * gm.PL;
      if (p < pe) {
        const size_t o = ((size_t)b * P + p) * C + cv * 8;
        load8(x + o, xv[u]);
        if (MODE == 1) {
          if constexpr (TG > 0) {
            const int yy = p / tp.W;
            gather_grad<T, TG>(g, tp, b, yy, p - yy * tp.W, C, cv, sv[u]);
          } else {
            load8(g + o, sv[u]);
          }
        } else if (res) {
          load8(res + o, sv[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * gm.PL;
      if (p >= pe) continue;
      float ov[8], xh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) xh[i] = (xv[u][i] - mean[i]) * rstd[i];
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = xh[i];
        act_fwd8(ov, act);  // one dispatch per vector (common.h)
        if (res) {
#pragma unroll
          for (int i = 0; i < 8; ++i) ov[i] += sv[u][i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float gh = sv[u][i] * act_bwd_from_out(xh[i], act);
          ov[i] = rstd[i] * (gh - m1[i] - xh[i] * m2[i]);
        }
      }
      store8(out + ((size_t)b * P + p) * C + cv * 8, ov);
    }
  }
}

// ---- separable banded resample -----------------------------------------------------------
// Tap count is a template parameter: the 2*TN weights and the TN*TN input vectors of one output
// are all loaded up front (independent loads in flight) instead of a dependent
// weight -> index -> data chain per tap.  Source indices are clamped (taps whose weight is
// zero may point past the edge).
template <typename T, int TY, int TX>
__global__ __launch_bounds__(NT) void resample_kernel(const T* x, T* y, const int* sy,
                                                      const float* wy, const int* sx,
                                                      const float* wx, int H, int W, int Ho,
                                                      int Wo, int C, long nvec) {
  const int CV = C / 8;
  for (long v = (long)blockIdx.x * NT + threadIdx.x; v < nvec; v += (long)gridDim.x * NT) {
    const int cv = (int)(v % CV);
    long r = v / CV;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int y0 = sy[oy], x0 = sx[ox];
    float a[TY], c[TX];
#pragma unroll
    for (int t = 0; t < TY; ++t) a[t] = wy[oy * TY + t];
#pragma unroll
    for (int t = 0; t < TX; ++t) c[t] = wx[ox * TX + t];
    const T* base = x + ((size_t)b * H * W) * C + cv * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int ty = 0; ty < TY; ++ty) {
      const int iy = min(y0 + ty, H - 1);
      float row[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) row[i] = 0.f;
#pragma unroll
      for (int tx = 0; tx < TX; ++tx) {
        const int ix = min(x0 + tx, W - 1);
        float t[8];
        load8(base + ((size_t)iy * W + ix) * C, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] += c[tx] * t[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += a[ty] * row[i];
    }
    store8(y + (size_t)v * 8, acc);
  }
}

template <typename T> struct RawVec;  // 8 consecutive channels as loaded (unpacked to floats on use)
template <> struct RawVec<unsigned short> {
  u32x4 v;
  __device__ __forceinline__ void load(const unsigned short* p) { v = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void unpack(float (&f)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
      f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
  }
};
template <> struct RawVec<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void unpack(float (&f)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
};

// 2x2 outputs per thread from ONE (TY+SY) x (TX+SX) input patch: neighbouring outputs of a banded
// operator start at most SPAN source pixels apart (host-checked), so their taps overlap and the
// per-output TN^2 loads (L1-bandwidth-bound: 36 x 16 B per output for the transposed upsample)
// drop to a quarter of the patch.  Each patch row is reduced horizontally for both output columns as it
// arrives, then scattered into the two output rows with the vertical weights.  Grid (x, oy/2, b):
// no 64-bit index arithmetic.
// NORM: the operator is applied to act(InstanceNorm(x)) formed on the fly from mean_rstd ([B][C][2]) -- the
// conv -> InstanceNorm -> (Leaky)ReLU -> DownSample chains of the encoder and of the discriminator / style extractor
// (builder.py:170-173,272-282) without the normalised map ever being written: it is not needed by the backward
// pass either (InstanceNorm differentiates through x and the statistics, the activation mask is the sign of xh).
template <typename T, int TY, int TX, int SY, int SX, bool NORM = false>
__global__ __launch_bounds__(NT) void resample2x2_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         const int* __restrict__ sy,
                                                         const float* __restrict__ wy,
                                                         const int* __restrict__ sx,
                                                         const float* __restrict__ wx, int H, int W,
                                                         int Ho, int Wo, int C,
                                                         const float* __restrict__ mean_rstd = nullptr, int act = 0) {
  constexpr int PY = TY + SY, PX = TX + SX;
  const int CV = C / 8;
  const int t = blockIdx.x * NT + threadIdx.x;
  const int cv = t % CV, ox0 = (t / CV) * 2;
  const int oy0 = blockIdx.y * 2, b = blockIdx.z;
  if (ox0 >= Wo) return;
  float mu[NORM ? 8 : 1], rs[NORM ? 8 : 1];
  const bool relu = act == O2M_ACT_RELU;
  const float neg = act == O2M_ACT_LRELU ? 0.2f : 1.f;
  if constexpr (NORM) {
    const float* mr = mean_rstd + ((size_t)b * C + cv * 8) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = mr[2 * i]; rs[i] = mr[2 * i + 1]; }
  }
  const bool has_x1 = ox0 + 1 < Wo, has_y1 = oy0 + 1 < Ho;
  const int ox1 = has_x1 ? ox0 + 1 : ox0, oy1 = has_y1 ? oy0 + 1 : oy0;
  const int x0 = sx[ox0], y0 = sy[oy0];
  const int dx1 = sx[ox1] - x0, dy1 = sy[oy1] - y0;  // 0 .. SPAN
  // weights of the two outputs re-indexed on the patch
  float cx[2][PX], cy[2][PY];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    cx[0][k] = k < TX ? wx[ox0 * TX + k] : 0.f;
    const int kx = k - dx1;
    cx[1][k] = (kx >= 0 && kx < TX) ? wx[ox1 * TX + kx] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < PY; ++k) {
    cy[0][k] = k < TY ? wy[oy0 * TY + k] : 0.f;
    const int ky = k - dy1;
    cy[1][k] = (ky >= 0 && ky < TY) ? wy[oy1 * TY + ky] : 0.f;
  }
  const T* base = x + ((size_t)b * H * W) * C + cv * 8;
  float acc[2][2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[a][c][i] = 0.f;
#pragma unroll
  for (int r = 0; r < PY; ++r) {
    const int iy = min(y0 + r, H - 1);
    float h0[8], h1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h0[i] = h1[i] = 0.f;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      const int ix = min(x0 + k, W - 1);
      float v[8];
      load8(base + ((size_t)iy * W + ix) * C, v);
      if constexpr (NORM) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_fwd_piecewise((v[i] - mu[i]) * rs[i], relu, neg);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { h0[i] += cx[0][k] * v[i]; h1[i] += cx[1][k] * v[i]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0][0][i] += cy[0][r] * h0[i];
      acc[0][1][i] += cy[0][r] * h1[i];
      acc[1][0][i] += cy[1][r] * h0[i];
      acc[1][1][i] += cy[1][r] * h1[i];
    }
  }
  T* o00 = y + (((size_t)b * Ho + oy0) * Wo + ox0) * C + cv * 8;
  store8(o00, acc[0][0]);
  if (has_x1) store8(o00 + C, acc[0][1]);
  if (has_y1) {
    store8(o00 + (size_t)Wo * C, acc[1][0]);
    if (has_x1) store8(o00 + (size_t)Wo * C + C, acc[1][1]);
  }
}

// Tile form for WIDE operators (the transposed 6-tap upsample: 36 taps per output).  One block owns an
// OH x OW output tile of CVB channel vectors and runs the separable operator in two phases through LDS:
//   A: thread (patch column, channel vector) loads the PH = SY (OH - 1) + TY input rows of its column ONCE
//      (coalesced: a wave covers 4 consecutive pixels x 256 B) and forms the OH vertically combined rows
//      -> LDS [OH][PW][2][CVB] float4 (two planes, so a lane group's 16-B reads are contiguous);
//   B: thread (output pixel, channel vector) combines its TX horizontal taps from LDS and stores 16 B.
// Every input element is fetched once (plus the tile halo, served by L2) and every output written once: the
// two 1-D passes this replaces moved the intermediate map through HBM and ran their row walk at ~2 TB/s.
// The vertical weights are block-uniform: re-indexed on the patch rows (zero outside an output's band) they
// sit in SGPRs and the row walk has only static register indices.
// NORM: the operator is applied to act(InstanceNorm(x)) formed as the rows are unpacked (as resample2x2_kernel's NORM
// form, whose per-thread 6 x 6 patch walk ran at 2.1 TB/s: 260 VGPRs -- one wave per SIMD -- and every patch element
// normalised 2.25 times; here each element is fetched and normalised once per tile, all rows of a column in flight).
template <typename T, int TY, int TX, int SY, int SX, int OH, int OW, bool NORM = false>
__global__ __launch_bounds__(NT) void resample_tile_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           const int* __restrict__ sy, const float* __restrict__ wy,
                                                           const int* __restrict__ sx, const float* __restrict__ wx,
                                                           int H, int W, int Ho, int Wo, int C, int CVB, int tiles_x,
                                                           const float* __restrict__ mean_rstd = nullptr, int act = 0) {
  constexpr int PH = SY * (OH - 1) + TY, PW = SX * (OW - 1) + TX;
  extern __shared__ float sm[];  // [OH][PW][2][CVB][4]
  const int chunk = blockIdx.x / tiles_x, tx_ = blockIdx.x - chunk * tiles_x;
  const int ox0 = tx_ * OW, oy0 = blockIdx.y * OH, b = blockIdx.z;
  const int y_start = sy[oy0], x_start = sx[ox0];
  const T* base = x + ((size_t)b * H * W) * C + (size_t)chunk * CVB * 8;
  f32x4* sm4 = reinterpret_cast<f32x4*>(sm);
  // ---- phase A -------------------------------------------------------------------------------------
  // Regular tiles (everything but the image's first / last tile rows): output row o starts exactly SY * o patch
  // rows below the tile's first, so the vertical taps index the preloaded rows STATICALLY and the TY weights per
  // output row are block-uniform scalars.  (Re-indexing the weights on the patch rows for every tile -- the first
  // version -- needs OH x PH uniform values: past the scalar file, 239 VGPRs, and twice the multiply-adds.)
  bool regular = true;
#pragma unroll
  for (int o = 0; o < OH; ++o) regular = regular && (oy0 + o < Ho) && (sy[min(oy0 + o, Ho - 1)] - y_start == SY * o);
  for (int item = threadIdx.x; item < PW * CVB; item += NT) {
    const int pc = item / CVB, cv = item - pc * CVB;
    const int ix = min(x_start + pc, W - 1);
    const T* col = base + (size_t)ix * C + cv * 8;
    float mu[NORM ? 8 : 1], rs[NORM ? 8 : 1];
    const bool relu = act == O2M_ACT_RELU;
    const float neg = act == O2M_ACT_LRELU ? 0.2f : 1.f;
    if constexpr (NORM) {
      const float* mr = mean_rstd + ((size_t)b * C + (size_t)chunk * CVB * 8 + cv * 8) * 2;
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(mr + 2 * i);
        mu[i] = m4[0]; rs[i] = m4[1]; mu[i + 1] = m4[2]; rs[i + 1] = m4[3];
      }
    }
    if (regular) {
      RawVec<T> raw[PH];  // all PH row loads in flight as raw 16-B (32-B) vectors, unpacked one row at a time
#pragma unroll
      for (int r = 0; r < PH; ++r) raw[r].load(col + (size_t)min(y_start + r, H - 1) * W * C);
      float acc[OH][8];
#pragma unroll
      for (int o = 0; o < OH; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[o][i] = 0.f;
#pragma unroll
      for (int r = 0; r < PH; ++r) {
        float v[8];
        raw[r].unpack(v);
        if constexpr (NORM) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = act_fwd_piecewise((v[i] - mu[i]) * rs[i], relu, neg);
        }
#pragma unroll
        for (int o = 0; o < OH; ++o) {
          const int t = r - SY * o;  // compile-time
          if (t >= 0 && t < TY) {
            const float w = wy[(oy0 + o) * TY + t];  // block-uniform: a scalar load
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[o][i] += w * v[i];
          }
        }
      }
#pragma unroll
      for (int o = 0; o < OH; ++o) {
        f32x4* dst = sm4 + ((size_t)(o * PW + pc) * 2) * CVB + cv;
        dst[0] = f32x4{acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
        dst[CVB] = f32x4{acc[o][4], acc[o][5], acc[o][6], acc[o][7]};
      }
    } else {
      // edge tiles (clamped starts, rows past Ho): one output row at a time, its TY source rows fetched directly
#pragma unroll 1
      for (int o = 0; o < OH; ++o) {
        const int oy = min(oy0 + o, Ho - 1);
        const int y0r = sy[oy];
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          float v[8];
          load8(col + (size_t)min(y0r + t, H - 1) * W * C, v);
          if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = act_fwd_piecewise((v[i] - mu[i]) * rs[i], relu, neg);
          }
          const float w = wy[oy * TY + t];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += w * v[i];
        }
        f32x4* dst = sm4 + ((size_t)(o * PW + pc) * 2) * CVB + cv;
        dst[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
        dst[CVB] = f32x4{acc[4], acc[5], acc[6], acc[7]};
      }
    }
  }
  __syncthreads();
  // ---- phase B -------------------------------------------------------------------------------------
  for (int item = threadIdx.x; item < OH * OW * CVB; item += NT) {
    const int cv = item % CVB, r_ = item / CVB;
    const int oxl = r_ % OW, o = r_ / OW;
    const int gx = ox0 + oxl, gy = oy0 + o;
    if (gx >= Wo || gy >= Ho) continue;
    const int offx = sx[gx] - x_start;  // 0 .. SX * oxl
    float out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = 0.f;
#pragma unroll
    for (int t = 0; t < TX; ++t) {
      const float w = wx[gx * TX + t];
      const f32x4* src = sm4 + ((size_t)(o * PW + offx + t) * 2) * CVB + cv;
      const f32x4 lo = src[0], hi = src[CVB];
#pragma unroll
      for (int i = 0; i < 4; ++i) { out[i] += w * lo[i]; out[4 + i] += w * hi[i]; }
    }
    store8(y + (((size_t)b * Ho + gy) * Wo + gx) * C + (size_t)chunk * CVB * 8 + cv * 8, out);
  }
}

// ---- device-resident image pool -> training batch ----------------------------------------------
// out[b][y][x][c] = ((pool[idx[b]][y][flip[b] ? W-1-x : x][c] / 255) - 0.5) / 0.5   (c < C; else 0)
// i.e. ToTensor + Normalize(0.5, 0.5) + RandomHorizontalFlip of the reference's input pipeline
// (train.py:120-126, datasets.py:44-50) in fp32, in the order the reference applies them, on uint8
// images that never leave HBM.  One thread per output pixel (all Cp channels: C is 1 or 3).
template <typename T>
__global__ __launch_bounds__(NT) void gather_images_kernel(const unsigned char* __restrict__ pool,
                                                           const int* __restrict__ idx,
                                                           const unsigned char* __restrict__ flip,
                                                           T* __restrict__ out, int H, int W, int C, int Cp) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * NT + threadIdx.x;
  if (x >= W) return;
  const int sx = flip[b] ? W - 1 - x : x;
  const unsigned char* src = pool + (((size_t)idx[b] * H + y) * W + sx) * C;
  T* dst = out + (((size_t)b * H + y) * W + x) * Cp;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      v[k] = (c0 + k) < C ? ((float)src[c0 + k] / 255.f - 0.5f) / 0.5f : 0.f;
    store8(dst + c0, v);
  }
}

// ---- NCHW fp32 <-> NHWC (channel padded) ----------------------------------------------------
template <typename T>
__global__ void pack_kernel(const float* src, T* dst, int C, int HW, int Cp, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (c0 + k) < C ? src[((size_t)b * C + c0 + k) * HW + p] : 0.f;
      store8(dst + (size_t)i * Cp + c0, v);
    }
  }
}
template <typename T>
__global__ void unpack_kernel(const T* src, float* dst, int C, int HW, int Cp, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    for (int c0 = 0; c0 < C; c0 += 8) {
      float v[8];
      load8(src + (size_t)i * Cp + c0, v);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c0 + k < C) dst[((size_t)b * C + c0 + k) * HW + p] = v[k];
    }
  }
}

// ---- loss reductions -------------------------------------------------------------------------
constexpr int RED_VEC_PER_BLOCK = NT * 8;  // 8 vectors (64 elements) per thread

template <typename T>
__global__ __launch_bounds__(NT) void reduce_fwd_kernel(const T* a, const T* b, const float* w,
                                                        float* partials, long nps_vec, long nvec,
                                                        int mode, int nblocks) {
  __shared__ float sm[2][NT / 64];
  float s0 = 0.f, s1 = 0.f;
  const long base = (long)blockIdx.x * RED_VEC_PER_BLOCK;
  for (int it = 0; it < 8; ++it) {
    const long v = base + it * NT + threadIdx.x;
    if (v >= nvec) break;
    float av[8], bv[8];
    load8(a + (size_t)v * 8, av);
    if (b) load8(b + (size_t)v * 8, bv);
    const float ws = (w && mode == O2M_RED_SQ) ? w[v / nps_vec] : 1.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float dlt = b ? av[i] - bv[i] : av[i];
      if (mode == O2M_RED_L1) s0 += fabsf(dlt);
      else if (mode == O2M_RED_SQ) s0 += ws * dlt * dlt;
      else { s0 += av[i]; s1 += av[i] * av[i]; }
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[0][wv] = s0; sm[1][wv] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int i = 0; i < NT / 64; ++i) { t0 += sm[0][i]; t1 += sm[1][i]; }
    partials[blockIdx.x] = t0;
    if (mode == O2M_RED_MOM) partials[nblocks + blockIdx.x] = t1;
  }
}

// ---- adversarial (LSGAN) loss on the discriminator's patch map -----------------------------------------------
// scores: internal [N][P][C] (C a multiple of 8, channel 0 = the one logical score).  One block: the map is 13-27 k
// elements.  out[0] / out[1] = sum (s - t)^2 over samples [0, n_first) with target t0 / [n_first, N) with t1;
// out[2] / out[3] = sum sign(2 s - 1) over the same halves (the reference's confidence, training.py:117-118).
// Fixed order: a thread's strided walk, then an LDS tree.
template <typename T>
__global__ __launch_bounds__(1024) void lsgan_fwd_kernel(const T* __restrict__ s, float* __restrict__ out, int N, int P, int C,
                                                         int n_first, float t0, float t1) {
  __shared__ float red[4][1024];
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  const long first = (long)n_first * P, total = (long)N * P;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const float v = Elem<T>::ld(s + i * C);
    const int h = i >= first;
    const float dlt = v - (h ? t1 : t0), u = 2.f * v - 1.f;
    a[h] += dlt * dlt;
    a[2 + h] += u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) out[threadIdx.x] = red[threadIdx.x][0];
}

// gs[n][p][0] = coef[half] * 2 (s - t_half), the padded channels 0
template <typename T>
__global__ __launch_bounds__(NT) void lsgan_bwd_kernel(const T* __restrict__ s, const float* __restrict__ coef, T* __restrict__ gs,
                                                       long total, long first, int C, float t0, float t1) {
  const long i = (long)blockIdx.x * NT + threadIdx.x;
  if (i >= total) return;
  const int h = i >= first;
  const float v = Elem<T>::ld(s + i * C);
  float o[8] = {coef[h] * 2.f * (v - (h ? t1 : t0)), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  store8(gs + i * C, o);
  const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = 8; c < C; c += 8) store8(gs + i * C + c, z);
}

template <typename T>
__global__ __launch_bounds__(NT) void reduce_bwd_kernel(const T* a, const T* b, const float* w,
                                                        const float* coef, T* ga, long nps_vec,
                                                        long nvec, int mode) {
  const float c0 = coef[0];
  const float c1 = mode == O2M_RED_MOM ? coef[1] : 0.f;
  for (long v = (long)blockIdx.x * NT + threadIdx.x; v < nvec; v += (long)gridDim.x * NT) {
    float av[8], bv[8], ov[8];
    load8(a + (size_t)v * 8, av);
    if (b) load8(b + (size_t)v * 8, bv);
    const float ws = (w && mode == O2M_RED_SQ) ? w[v / nps_vec] : 1.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float dlt = b ? av[i] - bv[i] : av[i];
      if (mode == O2M_RED_L1) ov[i] = dlt > 0.f ? c0 : (dlt < 0.f ? -c0 : 0.f);
      else if (mode == O2M_RED_SQ) ov[i] = c0 * ws * dlt;
      else ov[i] = c0 + c1 * av[i];
    }
    store8(ga + (size_t)v * 8, ov);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void pair_grad_kernel(const T* a, const T* b, const float* w, const float* coef,
                                                       const T* gin_a, const T* gin_b, T* ga, T* gb, long nps_vec,
                                                       long nvec) {
  const float c0 = coef[0];
  for (long v = (long)blockIdx.x * NT + threadIdx.x; v < nvec; v += (long)gridDim.x * NT) {
    float av[8], bv[8], ia[8], ib[8];
    load8(a + (size_t)v * 8, av);
    load8(b + (size_t)v * 8, bv);
#pragma unroll
    for (int i = 0; i < 8; ++i) ia[i] = ib[i] = 0.f;
    if (gin_a) load8(gin_a + (size_t)v * 8, ia);
    if (gin_b) load8(gin_b + (size_t)v * 8, ib);
    const float ws = c0 * (w ? w[v / nps_vec] : 1.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = ws * (av[i] - bv[i]);
      ia[i] += t;
      ib[i] -= t;
    }
    store8(ga + (size_t)v * 8, ia);
    store8(gb + (size_t)v * 8, ib);
  }
}

// ---- per-sample modulated filters ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void modulate_weights_kernel(const float* w32, const float* s, T* out,
                                                              long per, int Ci) {
  const int b = blockIdx.y;
  const int CV = Ci / 8;
  for (long v = (long)blockIdx.x * NT + threadIdx.x; v < per; v += (long)gridDim.x * NT) {
    const int cv = (int)(v % CV);
    float w[8], sv[8];
    load8(w32 + (size_t)v * 8, w);
    load8(s + (size_t)b * Ci + cv * 8, sv);
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] *= sv[i];
    store8(out + ((size_t)b * per + v) * 8, w);
  }
}

// ---- kernel-side forms of an equalised-LR filter ---------------------------------------------
// One thread per padded (o, i): walks the KK taps of the parameter (contiguous in [Co][Ci][KK]) and
// writes W*c in the GEMM layouts: full / w_f [Cop][KK][Cip], w_d [Cip][KK reversed][Cop], and the
// demodulation table q[o][i] = sum_kk (c W)^2 with its transpose.
template <typename T>
__global__ __launch_bounds__(NT) void prepare_weights_kernel(const float* w, float* full, T* w_f, T* w_d,
                                                             float* q, float* qt, int Co, int Ci, int KK,
                                                             int Cop, int Cip, float c) {
  const int t = blockIdx.x * NT + threadIdx.x;
  if (t >= Cop * Cip) return;
  const int i = t % Cip, o = t / Cip;
  const bool valid = o < Co && i < Ci;
  const float* src = w + ((size_t)o * Ci + i) * KK;
  float acc = 0.f;
  for (int kk = 0; kk < KK; ++kk) {
    const float v = valid ? src[kk] * c : 0.f;
    const size_t f = ((size_t)o * KK + kk) * Cip + i;
    full[f] = v;
    Elem<T>::st(w_f + f, v);
    Elem<T>::st(w_d + ((size_t)i * KK + (KK - 1 - kk)) * Cop + o, v);
    acc += v * v;
  }
  if (q) {
    q[(size_t)o * Cip + i] = acc;
    qt[(size_t)i * Cop + o] = acc;
  }
}

// Every filter of a network in ONE launch (o2m_prepare_weights_batched): block -> job by a scan of the (short)
// table, then prepare_weights_kernel's per-(o, i) walk.
template <typename T>
__global__ __launch_bounds__(NT) void prepare_weights_batched_kernel(const o2m_prep_job* __restrict__ jobs, int n_jobs) {
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;  // uniform
  const o2m_prep_job jb = jobs[j];
  const int t = ((int)blockIdx.x - jb.first_block) * NT + threadIdx.x;
  if (t >= jb.Cop * jb.Cip) return;
  const int i = t % jb.Cip, o = t / jb.Cip;
  const bool valid = o < jb.Co && i < jb.Ci;
  const float* src = jb.w + ((size_t)o * jb.Ci + i) * jb.KK;
  T* w_f = static_cast<T*>(jb.w_f);
  T* w_d = static_cast<T*>(jb.w_d);
  float acc = 0.f;
  for (int kk = 0; kk < jb.KK; ++kk) {
    const float v = valid ? src[kk] * jb.c : 0.f;
    const size_t f = ((size_t)o * jb.KK + kk) * jb.Cip + i;
    jb.full[f] = v;
    Elem<T>::st(w_f + f, v);
    Elem<T>::st(w_d + ((size_t)i * jb.KK + (jb.KK - 1 - kk)) * jb.Cop + o, v);
    acc += v * v;
  }
  if (jb.q) {
    jb.q[(size_t)o * jb.Cip + i] = acc;
    jb.qt[(size_t)i * jb.Cop + o] = acc;
  }
}

// ---- weight-gradient finalisation ----------------------------------------------------------
// grad[o][i][kh][kw] += c * ( acc[o][kh][kw][i] + 2 * gq[o][i] * w32[o][kh][kw][i] ), then the
// accumulators are cleared for the next backward pass.  acc is the kernel-layout fp32 buffer
// o2m_conv2d_wgrad adds into; the gq term is dL/dQ of the demodulation (Q = sum_k (c W)^2).
__global__ __launch_bounds__(NT) void wgrad_finalize_kernel(float* acc, float* gq, const float* w32,
                                                            float* grad, int Co, int Ci, int KK, int Cip,
                                                            float c, long n) {
  for (long t = (long)blockIdx.x * NT + threadIdx.x; t < n; t += (long)gridDim.x * NT) {
    // t indexes the kernel layout [o][kk][i] (reads coalesced; writes scattered but tiny)
    const int i = (int)(t % Cip);
    const long r = t / Cip;
    const int kk = (int)(r % KK);
    const int o = (int)(r / KK);
    float v = acc[t];
    acc[t] = 0.f;
    if (o < Co && i < Ci) {
      if (gq) v += 2.f * gq[(size_t)o * Cip + i] * w32[t];
      grad[((size_t)o * Ci + i) * KK + kk] += c * v;
    }
  }
}
// every pending layer of a backward pass in one launch (+ one for the dL/dQ tables): blockIdx -> job through the
// first_block table, FIN_PER_BLOCK consecutive kernel-layout elements per block
constexpr int FIN_PER_BLOCK = NT * 8;
__global__ __launch_bounds__(NT) void wgrad_finalize_batched_kernel(const o2m_wfin_job* __restrict__ jobs, int n_jobs) {
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;  // uniform
  const o2m_wfin_job jb = jobs[j];
  const long n = (long)jb.Cop * jb.KK * jb.Cip;
  const long t0 = (long)((int)blockIdx.x - jb.first_block) * FIN_PER_BLOCK + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long t = t0 + (long)k * NT;
    if (t >= n) break;
    const int i = (int)(t % jb.Cip);
    const long r = t / jb.Cip;
    const int kk = (int)(r % jb.KK);
    const int o = (int)(r / jb.KK);
    float v = jb.acc[t];
    jb.acc[t] = 0.f;
    if (o < jb.Co && i < jb.Ci) {
      if (jb.gq) v += 2.f * jb.gq[(size_t)o * jb.Cip + i] * jb.w32[t];
      jb.grad[((size_t)o * jb.Ci + i) * jb.KK + kk] += jb.c * v;
    }
  }
}
__global__ __launch_bounds__(NT) void wgrad_clear_gq_batched_kernel(const o2m_wfin_job* __restrict__ jobs, int n_jobs) {
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
  const o2m_wfin_job jb = jobs[j];
  if (!jb.gq) return;
  const long n = (long)jb.Cop * jb.Cip;  // (<= the job's element count: its blocks cover it)
  const long t0 = (long)((int)blockIdx.x - jb.first_block) * FIN_PER_BLOCK + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long t = t0 + (long)k * NT;
    if (t < n) jb.gq[t] = 0.f;
  }
}
__global__ void clear_kernel(float* p, long n) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) p[t] = 0.f;
}

// ---- per-tensor fp8 quantisation (config #5) ---------------------------------------------------
// amax: O2M_AMAX_PARTIALS blocks, one plain store each (8192 atomicMax onto one word cost ~100 us: a
// same-address atomic retires every ~12 ns chip-wide); the quantising kernel reduces the partials itself.
template <typename T>
__global__ __launch_bounds__(NT) void amax_kernel(const T* __restrict__ x, float* __restrict__ partial, long nvec) {
  float m = 0.f;
  for (long v = (long)blockIdx.x * NT + threadIdx.x; v < nvec; v += (long)gridDim.x * NT) {
    float f[8];
    load8(x + v * 8, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(f[i]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float red[NT / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) m = fmaxf(m, red[i]);
    partial[blockIdx.x] = m;
  }
}

// y = fp8(x * scale), 8 values (one 8-byte store) per thread; thread 0 also publishes 1 / scale
// TRACK (delayed scaling): `amax` holds the partial maxima of the tensor this site quantised LAST time; the maxima of THIS
// tensor go to `next` (one per block, the rest zeroed by block 0) for the next call -- one pass over x instead of two.
template <typename T, bool E5M2, bool TRACK = false>
__global__ __launch_bounds__(NT) void quantize_fp8_kernel(const T* __restrict__ x, const float* __restrict__ amax,
                                                          unsigned long long* __restrict__ y, float* __restrict__ deq,
                                                          long nvec, float* __restrict__ next = nullptr) {
  const float fmt_max = E5M2 ? 57344.f : 448.f;
  // every block reduces the O2M_AMAX_PARTIALS partial maxima (4 KB out of L2) to the tensor's amax
  __shared__ float red[NT / 64];
  float m = 0.f;
  for (int i = threadIdx.x; i < O2M_AMAX_PARTIALS; i += NT) m = fmaxf(m, amax[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) m = fmaxf(m, red[i]);
  const float scale = fmt_max / fmaxf(m, 1e-12f);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    deq[0] = 1.f / scale;
    deq[1] = m;  // the tensor's amax, for the caller's records (delayed scaling, tests)
  }
  float seen = 0.f;
  for (long v = (long)blockIdx.x * NT + threadIdx.x; v < nvec; v += (long)gridDim.x * NT) {
    float f[8];
    load8(x + v * 8, f);
    if (TRACK) {
#pragma unroll
      for (int i = 0; i < 8; ++i) seen = fmaxf(seen, fabsf(f[i]));
    }
    // v_cvt_pk_{fp8,bf8}_f32: two floats -> two bytes (RNE), into the low / high half of a dword
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = fminf(fmaxf(f[i] * scale, -fmt_max), fmt_max);
    unsigned int w[2] = {0u, 0u};
    if (E5M2) {
      w[0] = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], w[0], false);
      w[0] = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], w[0], true);
      w[1] = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], w[1], false);
      w[1] = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], w[1], true);
    } else {
      w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], w[0], false);
      w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w[0], true);
      w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], w[1], false);
      w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], w[1], true);
    }
    y[v] = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
  }
  if (TRACK) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) seen = fmaxf(seen, __shfl_xor(seen, off, 64));
    __syncthreads();  // (red is read above by every thread)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = seen;
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = red[0];
#pragma unroll
      for (int i = 1; i < NT / 64; ++i) mm = fmaxf(mm, red[i]);
      next[blockIdx.x] = mm;
    }
    if (blockIdx.x == 0)
      for (int i = gridDim.x + threadIdx.x; i < O2M_AMAX_PARTIALS; i += NT) next[i] = 0.f;
  }
}

// ---- fused Adam ------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void adam_kernel(float* p, const float* g, float* m, float* v,
                                                  const float* step, long n, float lr, float b1,
                                                  float b2, float eps, float gscale) {
  const float t = step[0];
  const float bc1 = 1.f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

inline unsigned grid_for(long n, int per_block = NT) {
  long g = (n + per_block - 1) / per_block;
  if (g > 8192) g = 8192;  // grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

#define DISPATCH_T(dtype, ...)                                             \
  if ((dtype) == O2M_BF16) { using T = unsigned short; __VA_ARGS__; }      \
  else if ((dtype) == O2M_F32) { using T = float; __VA_ARGS__; }           \
  else return O2M_ERR_BAD_ARG;

}  // namespace

// ---------------------------------------------------------------------------------- launch timing
#include <mutex>
#include <string>
#include <vector>

namespace o2m_timing {
bool g_on = false;
namespace {
struct Rec {
  std::string name;
  double flops;
  hipEvent_t e0, e1;
};
std::mutex g_mu;  // forward runs on the caller's thread, backward on autograd's device thread
std::vector<Rec> g_recs;
}  // namespace

int open(const char* name, double flops, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_mu);
  Rec r{name, flops, nullptr, nullptr};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void close(int slot, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (slot >= 0 && slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].e1, s);
}
}  // namespace o2m_timing

extern "C" {

int o2m_abi_version(void) { return 22; }

int32_t o2m_launch_timing(int32_t enable) {
  std::lock_guard<std::mutex> lock(o2m_timing::g_mu);
  const int32_t was = o2m_timing::g_on;
  o2m_timing::g_on = enable != 0;
  return was;
}

int32_t o2m_launch_timing_read(o2m_launch_stat* out, int32_t capacity) {
  using namespace o2m_timing;
  if (capacity < 0 || (capacity > 0 && !out)) return -O2M_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  int32_t n = 0, err = 0;
  for (Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) err = 1;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    int32_t i = 0;
    while (i < n && r.name != out[i].kernel) ++i;
    if (i == n) {
      if (n == capacity) continue;  // table full: the remaining kernels are dropped, the count says so
      snprintf(out[i].kernel, sizeof(out[i].kernel), "%s", r.name.c_str());
      out[i].launches = 0;
      out[i].ms = 0.f;
      out[i].flops = 0.0;
      ++n;
    }
    out[i].launches += 1;
    out[i].ms += ms;
    out[i].flops += r.flops;
  }
  g_recs.clear();
  return err ? -1 : n;
}

int o2m_amax(const void* x, float* amax, int64_t n, int32_t dtype, void* stream) {
  if (!x || !amax || n <= 0 || (n & 7)) return O2M_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long nvec = n / 8;
  DISPATCH_T(dtype, hipLaunchKernelGGL(amax_kernel<T>, dim3(O2M_AMAX_PARTIALS), dim3(NT), 0, st, (const T*)x, amax, nvec));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_quantize_fp8(const void* x, const float* amax, void* y, float* deq, int64_t n, int32_t dtype, int32_t fmt,
                     void* stream) {
  if (!x || !amax || !y || !deq || n <= 0 || (n & 7)) return O2M_ERR_BAD_ARG;
  if (fmt != O2M_FP8_E4M3 && fmt != O2M_BF8_E5M2) return O2M_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long nvec = n / 8;
  if (fmt == O2M_FP8_E4M3) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((quantize_fp8_kernel<T, false>), dim3(grid_for(nvec)), dim3(NT), 0, st,
                                         (const T*)x, amax, (unsigned long long*)y, deq, nvec));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((quantize_fp8_kernel<T, true>), dim3(grid_for(nvec)), dim3(NT), 0, st,
                                         (const T*)x, amax, (unsigned long long*)y, deq, nvec));
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_quantize_fp8_delayed(const void* x, const float* amax_prev, void* y, float* deq, float* amax_next, int64_t n,
                             int32_t dtype, int32_t fmt, void* stream) {
  if (!x || !amax_prev || !y || !deq || !amax_next || amax_prev == amax_next || n <= 0 || (n & 7)) return O2M_ERR_BAD_ARG;
  if (fmt != O2M_FP8_E4M3 && fmt != O2M_BF8_E5M2) return O2M_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long nvec = n / 8;
  const unsigned grid = (unsigned)std::min<long>(grid_for(nvec), O2M_AMAX_PARTIALS);  // one partial per block
  if (fmt == O2M_FP8_E4M3) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((quantize_fp8_kernel<T, false, true>), dim3(grid), dim3(NT), 0, st, (const T*)x,
                                         amax_prev, (unsigned long long*)y, deq, nvec, amax_next));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((quantize_fp8_kernel<T, true, true>), dim3(grid), dim3(NT), 0, st, (const T*)x,
                                         amax_prev, (unsigned long long*)y, deq, nvec, amax_next));
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_modulate_weights(const float* w32, const float* s, void* out, int32_t B, int32_t Co,
                         int32_t KK, int32_t Ci, int32_t dtype, void* stream) {
  if (!w32 || !s || !out || B <= 0 || Co <= 0 || KK <= 0 || Ci <= 0 || (Ci & 7)) return O2M_ERR_BAD_ARG;
  const long per = (long)Co * KK * (Ci / 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(modulate_weights_kernel<T>, dim3(grid_for(per), B), dim3(NT), 0, st,
                                       w32, s, (T*)out, per, Ci));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_prepare_weights(const float* w, float* full, void* w_f, void* w_d, float* q, float* qt,
                        int32_t Co, int32_t Ci, int32_t KK, int32_t Cop, int32_t Cip, float c,
                        int32_t dtype, void* stream) {
  if (!w || !full || !w_f || !w_d || (q == nullptr) != (qt == nullptr)) return O2M_ERR_BAD_ARG;
  if (Co <= 0 || Ci <= 0 || KK <= 0 || Cop < Co || Cip < Ci || (Cop & 7) || (Cip & 7)) return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(prepare_weights_kernel<T>, dim3(grid_for((long)Cop * Cip)), dim3(NT),
                                       0, s, w, full, (T*)w_f, (T*)w_d, q, qt, Co, Ci, KK, Cop, Cip, c));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_prepare_weights_batched(const o2m_prep_job* jobs, int32_t n_jobs, int32_t total_blocks, int32_t dtype,
                                void* stream) {
  if (!jobs || n_jobs <= 0 || total_blocks <= 0) return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(prepare_weights_batched_kernel<T>, dim3((unsigned)total_blocks), dim3(NT), 0, s,
                                       jobs, n_jobs));
  O2M_LAUNCH_CHECK();
  return 0;
}

size_t o2m_chan_partials_floats(int32_t B, int32_t P, int32_t C, int32_t nv) {
  if (B <= 0 || P <= 0 || C <= 0 || nv <= 0) return 0;
  return (size_t)B * chan_geom(B, P, C).nchunks * nv * C;
}

int o2m_act_bwd_reduce(const void* g, const void* y, const void* residual, const float* out_mul,
                       void* gu, float* sums, int32_t B, int32_t P, int32_t C, int32_t act,
                       int32_t dtype, float* partials, void* stream) {
  if (!g || !sums || B <= 0 || P <= 0 || C <= 0 || (C & 7) || C > 8 * NT) return O2M_ERR_BAD_ARG;
  // y may be omitted only for the identity activation without residual (reduce-only: gu optional)
  if (!y && (act != O2M_ACT_NONE || residual)) return O2M_ERR_BAD_ARG;
  if (y && !gu) return O2M_ERR_BAD_ARG;
  ChanGeom gm = chan_geom(B, P, C);
  const size_t lds = (size_t)gm.PL * gm.CV * 16 * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define O2M_ABR(HY, HR)                                                                                   \
  DISPATCH_T(dtype, hipLaunchKernelGGL((act_bwd_reduce_kernel<T, HY, HR>), dim3(gm.nchunks, B), dim3(NT), lds, s, \
                                       (const T*)g, (const T*)y, (const T*)residual, out_mul, (T*)gu, sums,     \
                                       partials, P, C, act, gm))
  if (!y) { O2M_ABR(false, false); }
  else if (!residual) { O2M_ABR(true, false); }
  else { O2M_ABR(true, true); }
#undef O2M_ABR
  O2M_LAUNCH_CHECK();
  if (partials) {
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(B), dim3(NT), 0, s, sums, partials, gm.nchunks, 2 * C);
    O2M_LAUNCH_CHECK();
  }
  return 0;
}

int o2m_fold_scale_dot(const void* gpad, const void* x, const float* scale, void* gx, float* dots,
                       void* xs, const void* gres, int32_t act, const float* act_mul, float* act_sums,
                       int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t dtype,
                       float* partials, void* stream) {
  if (!gpad || !gx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || C > 8 * NT || pad < 0)
    return O2M_ERR_BAD_ARG;
  if (pad >= H || pad >= W || (dots && !x) || (xs && !dots)) return O2M_ERR_BAD_ARG;
  if (act_sums && (!x || act == O2M_ACT_TANH)) return O2M_ERR_BAD_ARG;  // the fused activation reads its output x
  if (!act_sums && (act_mul || act != O2M_ACT_NONE)) return O2M_ERR_BAD_ARG;
  ChanGeom gm = chan_geom(B, H * W, C);
  const size_t lds = (size_t)gm.PL * gm.CV * (act_sums ? 24 : 8) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = (dots || act_sums) ? partials : nullptr;
  if (act_sums) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((fold_scale_dot_kernel<T, true>), dim3(gm.nchunks, B), dim3(NT), lds, s,
                                         (const T*)gpad, (const T*)x, scale, (T*)gx, dots, (T*)xs, (const T*)gres,
                                         act_mul, act_sums, part, H, W, C, pad, act, gm));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((fold_scale_dot_kernel<T, false>), dim3(gm.nchunks, B), dim3(NT), lds, s,
                                         (const T*)gpad, (const T*)x, scale, (T*)gx, dots, (T*)xs, (const T*)gres,
                                         act_mul, act_sums, part, H, W, C, pad, act, gm));
  }
  O2M_LAUNCH_CHECK();
  if (part && act_sums) {
    hipLaunchKernelGGL(chunk_sum3_kernel, dim3(B), dim3(NT), 0, s, dots, act_sums, part, gm.nchunks, C);
    O2M_LAUNCH_CHECK();
  } else if (part) {
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(B), dim3(NT), 0, s, dots, part, gm.nchunks, C);
    O2M_LAUNCH_CHECK();
  }
  return 0;
}

size_t o2m_instnorm_ws_floats(int32_t B, int32_t P, int32_t C) {
  if (B <= 0 || P <= 0 || C <= 0 || (C & 7)) return 0;
  ChanGeom gm = chan_geom(B, P, C);
  return (size_t)B * gm.nchunks * C * 2;
}

int o2m_instnorm_stats(const void* x, float* partial, float* mean_rstd, int32_t B, int32_t P,
                       int32_t C, float eps, int32_t dtype, void* stream) {
  if (!x || !partial || !mean_rstd || B <= 0 || P <= 0 || C <= 0 || (C & 7) || C > 8 * NT)
    return O2M_ERR_BAD_ARG;
  ChanGeom gm = chan_geom(B, P, C);
  const size_t lds = (size_t)gm.PL * gm.CV * 16 * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL((in_partial_kernel<T, 0>), dim3(gm.nchunks, B), dim3(NT), lds, s,
                                       (const T*)x, (const T*)nullptr, (const float*)nullptr, partial,
                                       P, C, 0, gm));
  O2M_LAUNCH_CHECK();
  hipLaunchKernelGGL(in_finalize_kernel<0>, dim3((B * C + 31) / 32), dim3(256), 0, s, partial,
                     mean_rstd, B, P, C, gm.nchunks, eps);
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_instnorm_finalize(const float* partial, float* mean_rstd, int32_t B, int32_t P, int32_t C,
                          int32_t nchunks, float eps, void* stream) {
  if (!partial || !mean_rstd || B <= 0 || P <= 0 || C <= 0 || nchunks <= 0) return O2M_ERR_BAD_ARG;
  hipLaunchKernelGGL(in_finalize_kernel<0>, dim3((B * C + 31) / 32), dim3(256), 0,
                     static_cast<hipStream_t>(stream), partial, mean_rstd, B, P, C, nchunks, eps);
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_instnorm_apply(const void* x, const float* mean_rstd, const void* residual, void* y,
                       int32_t B, int32_t P, int32_t C, int32_t act, int32_t dtype, void* stream) {
  if (!x || !mean_rstd || !y || B <= 0 || P <= 0 || C <= 0 || (C & 7)) return O2M_ERR_BAD_ARG;
  if (C > 8 * NT) return O2M_ERR_BAD_ARG;
  ChanGeom gm = chan_geom(B, P, C);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL((in_apply_kernel<T, 0>), dim3(gm.nchunks, B), dim3(NT), 0, s,
                                       (const T*)x, (const T*)nullptr, mean_rstd,
                                       (const float*)nullptr, (const T*)residual, (T*)y, P, C, act, gm));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_instnorm_bwd(const void* g, const void* x, const float* mean_rstd, float* partial,
                     float* gsums, void* gx, int32_t B, int32_t P, int32_t C, int32_t act,
                     int32_t dtype, void* stream) {
  if (!g || !x || !mean_rstd || !partial || !gsums || !gx) return O2M_ERR_BAD_ARG;
  if (B <= 0 || P <= 0 || C <= 0 || (C & 7) || C > 8 * NT || act == O2M_ACT_TANH)
    return O2M_ERR_BAD_ARG;
  ChanGeom gm = chan_geom(B, P, C);
  const size_t lds = (size_t)gm.PL * gm.CV * 16 * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL((in_partial_kernel<T, 1>), dim3(gm.nchunks, B), dim3(NT), lds, s, (const T*)x,
                       (const T*)g, mean_rstd, partial, P, C, act, gm);
    hipLaunchKernelGGL(in_finalize_kernel<1>, dim3((B * C + 31) / 32), dim3(256), 0, s, partial,
                       gsums, B, P, C, gm.nchunks, 0.f);
    hipLaunchKernelGGL((in_apply_kernel<T, 1>), dim3(gm.nchunks, B), dim3(NT), 0, s, (const T*)x,
                       (const T*)g, mean_rstd, gsums, (const T*)nullptr, (T*)gx, P, C, act, gm);
  });
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_instnorm_act_resample2d(const void* x, const float* mean_rstd, void* y, const int32_t* sy, const float* wy,
                                const int32_t* sx, const float* wx, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t C, int32_t T_, int32_t span_y, int32_t span_x, int32_t act, int32_t dtype,
                                void* stream) {
  if (!x || !mean_rstd || !y || !sy || !wy || !sx || !wx) return O2M_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 7) || act == O2M_ACT_TANH) return O2M_ERR_BAD_ARG;
  if (T_ != 4 || span_y < 2 || span_y > 3 || span_x < 2 || span_x > 3 || B > 65535 || (Ho + 1) / 2 > 65535)
    return O2M_ERR_UNSUPPORTED;  // the DownSample operators (4 taps, starts 2 or 3 apart); the caller falls back
  hipStream_t s = static_cast<hipStream_t>(stream);
  // even maps (every start exactly two apart): the tile form -- 8 x 7 outputs of 16 channel vectors per block, a
  // 18 x 16 patch whose 256 (column, vector) items are one per thread
  static const int tile_on = [] { const char* e = getenv("O2M_NORM_DOWN_TILE"); return e ? atoi(e) : 1; }();
  if (tile_on && span_y == 2 && span_x == 2 && (C / 8) % 16 == 0 && (Ho + 7) / 8 <= 65535) {
    constexpr int OH = 8, OW = 7, PW = 2 * (OW - 1) + 4, CVB = 16;
    const int tiles_x = (Wo + OW - 1) / OW, tiles_y = (Ho + OH - 1) / OH;
    const size_t lds = (size_t)OH * PW * 2 * CVB * 16;
    const dim3 tgrid((unsigned)(tiles_x * (C / 8 / CVB)), (unsigned)tiles_y, (unsigned)B);
    DISPATCH_T(dtype, {
      auto kern = resample_tile_kernel<T, 4, 4, 2, 2, OH, OW, true>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, tgrid, dim3(NT), lds, s, (const T*)x, (T*)y, sy, wy, sx, wx, H, W, Ho, Wo, C, CVB, tiles_x,
                         mean_rstd, act);
    });
    O2M_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid((unsigned)((((long)(Wo + 1) / 2) * (C / 8) + NT - 1) / NT), (unsigned)((Ho + 1) / 2), (unsigned)B);
#define O2M_IN_DOWN(SY, SX)                                                                                      \
  if (span_y == SY && span_x == SX) {                                                                            \
    DISPATCH_T(dtype, hipLaunchKernelGGL((resample2x2_kernel<T, 4, 4, SY, SX, true>), grid, dim3(NT), 0, s,      \
                                         (const T*)x, (T*)y, sy, wy, sx, wx, H, W, Ho, Wo, C, mean_rstd, act));  \
    O2M_LAUNCH_CHECK();                                                                                          \
    return 0;                                                                                                    \
  }
  O2M_IN_DOWN(2, 2) O2M_IN_DOWN(3, 3) O2M_IN_DOWN(2, 3) O2M_IN_DOWN(3, 2)
#undef O2M_IN_DOWN
  return O2M_ERR_UNSUPPORTED;
}

int o2m_instnorm_resample_bwd(const void* g_coarse, const void* x, const float* mean_rstd, float* partial, float* gsums,
                              void* gx, const int32_t* sy, const float* wy, const int32_t* sx, const float* wx,
                              int32_t B, int32_t H, int32_t W, int32_t Hl, int32_t Wl, int32_t C, int32_t T_, int32_t act,
                              int32_t dtype, void* stream) {
  if (!g_coarse || !x || !mean_rstd || !partial || !gsums || !gx || !sy || !wy || !sx || !wx) return O2M_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || Hl <= 0 || Wl <= 0 || C <= 0 || (C & 7) || C > 8 * NT || act == O2M_ACT_TANH)
    return O2M_ERR_BAD_ARG;
  if (T_ != 2) return O2M_ERR_UNSUPPORTED;  // the transposed DownSample: two taps per axis
  const int P = H * W;
  ChanGeom gm = chan_geom(B, P, C);
  const size_t lds = (size_t)gm.PL * gm.CV * 16 * sizeof(float);
  const GatherTaps tp{sy, wy, sx, wx, W, Hl, Wl};
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL((in_partial_kernel<T, 1, 2>), dim3(gm.nchunks, B), dim3(NT), lds, s, (const T*)x, (const T*)g_coarse,
                       mean_rstd, partial, P, C, act, gm, tp);
    hipLaunchKernelGGL(in_finalize_kernel<1>, dim3((B * C + 31) / 32), dim3(256), 0, s, partial, gsums, B, P, C,
                       gm.nchunks, 0.f);
    hipLaunchKernelGGL((in_apply_kernel<T, 1, 2>), dim3(gm.nchunks, B), dim3(NT), 0, s, (const T*)x, (const T*)g_coarse,
                       mean_rstd, gsums, (const T*)nullptr, (T*)gx, P, C, act, gm, tp);
  });
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_resample2d(const void* x, void* y, const int32_t* sy, const float* wy, const int32_t* sx,
                   const float* wx, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                   int32_t C, int32_t Ty, int32_t Tx, int32_t span_y, int32_t span_x, int32_t dtype,
                   void* stream) {
  if (!x || !y || !sy || !wy || !sx || !wx) return O2M_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 7) || Ty <= 0 || Tx <= 0)
    return O2M_ERR_BAD_ARG;
  const long nvec = (long)B * Ho * Wo * (C / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // tile kernel (LDS, separable in one launch) for the 6-tap transposed upsample
  if (Ty == 6 && Tx == 6 && span_y >= 1 && span_y <= 2 && span_x >= 1 && span_x <= 2 && B <= 65535) {
    constexpr int OH = 4, OW = 6, PW = 2 * (OW - 1) + 6;
    const int CV = C / 8;
    int CVB = CV < 16 ? CV : 16;
    while (CV % CVB) --CVB;
    const int tiles_x = (Wo + OW - 1) / OW, tiles_y = (Ho + OH - 1) / OH;
    if (tiles_y <= 65535) {
      const size_t lds = (size_t)OH * PW * 2 * CVB * 16;
      const dim3 grid((unsigned)(tiles_x * (CV / CVB)), (unsigned)tiles_y, (unsigned)B);
      DISPATCH_T(dtype, hipLaunchKernelGGL((resample_tile_kernel<T, 6, 6, 2, 2, OH, OW>), grid, dim3(NT), lds, s,
                                           (const T*)x, (T*)y, sy, wy, sx, wx, H, W, Ho, Wo, C, CVB, tiles_x));
      O2M_LAUNCH_CHECK();
      return 0;
    }
  }
  // 2x2-block kernel for the operators the model uses; anything else takes the per-output kernel
  if (span_y >= 1 && span_x >= 1 && B <= 65535 && (Ho + 1) / 2 <= 65535) {
    const dim3 grid((unsigned)((((long)(Wo + 1) / 2) * (C / 8) + NT - 1) / NT), (unsigned)((Ho + 1) / 2), (unsigned)B);
#define O2M_RESAMPLE_2X2(TY, TX, SY, SX)                                                           \
    if (Ty == TY && Tx == TX && span_y == SY && span_x == SX) {                                    \
      DISPATCH_T(dtype, hipLaunchKernelGGL((resample2x2_kernel<T, TY, TX, SY, SX>), grid, dim3(NT), 0, s, \
                                           (const T*)x, (T*)y, sy, wy, sx, wx, H, W, Ho, Wo, C));  \
      O2M_LAUNCH_CHECK();                                                                          \
      return 0;                                                                                    \
    }
    O2M_RESAMPLE_2X2(2, 2, 1, 1) O2M_RESAMPLE_2X2(3, 3, 1, 1) O2M_RESAMPLE_2X2(4, 4, 2, 2)
    O2M_RESAMPLE_2X2(4, 4, 3, 3) O2M_RESAMPLE_2X2(4, 4, 2, 3) O2M_RESAMPLE_2X2(4, 4, 3, 2)
    // one axis at a time (the transposed upsample runs as a vertical and a horizontal pass)
    O2M_RESAMPLE_2X2(6, 1, 2, 1) O2M_RESAMPLE_2X2(1, 6, 1, 2)
    // the 12-tap low-pass filters of the augmentation pipe (ada.py) and their transposes
    O2M_RESAMPLE_2X2(12, 1, 2, 1) O2M_RESAMPLE_2X2(1, 12, 1, 2) O2M_RESAMPLE_2X2(6, 1, 1, 1)
    O2M_RESAMPLE_2X2(1, 6, 1, 1)
#undef O2M_RESAMPLE_2X2
  }
#define O2M_RESAMPLE_CASE(TY, TX)                                                                \
  if (Ty == TY && Tx == TX) {                                                                    \
    DISPATCH_T(dtype, hipLaunchKernelGGL((resample_kernel<T, TY, TX>), dim3(grid_for(nvec)), dim3(NT), 0, s, \
                                         (const T*)x, (T*)y, sy, wy, sx, wx, H, W, Ho, Wo, C, nvec)); \
    O2M_LAUNCH_CHECK();                                                                          \
    return 0;                                                                                    \
  }
  O2M_RESAMPLE_CASE(1, 1) O2M_RESAMPLE_CASE(2, 2) O2M_RESAMPLE_CASE(3, 3) O2M_RESAMPLE_CASE(4, 4)
  O2M_RESAMPLE_CASE(5, 5) O2M_RESAMPLE_CASE(6, 6) O2M_RESAMPLE_CASE(7, 7) O2M_RESAMPLE_CASE(8, 8)
  // 1-D passes with arbitrary (non-monotonic) starts: reflection padding folded into an upsampling
  O2M_RESAMPLE_CASE(6, 1) O2M_RESAMPLE_CASE(1, 6) O2M_RESAMPLE_CASE(7, 1) O2M_RESAMPLE_CASE(1, 7)
  O2M_RESAMPLE_CASE(8, 1) O2M_RESAMPLE_CASE(1, 8)
  return O2M_ERR_UNSUPPORTED;
#undef O2M_RESAMPLE_CASE
}

int o2m_gather_images(const uint8_t* pool, const int32_t* index, const uint8_t* flip, void* out,
                      int32_t N, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t dtype,
                      void* stream) {
  if (!pool || !index || !flip || !out) return O2M_ERR_BAD_ARG;
  if (N <= 0 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cp < C || (Cp & 7)) return O2M_ERR_BAD_ARG;
  if (B > 65535 || H > 65535) return O2M_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((W + NT - 1) / NT), (unsigned)H, (unsigned)B);
  DISPATCH_T(dtype, hipLaunchKernelGGL(gather_images_kernel<T>, grid, dim3(NT), 0, s, pool, index, flip,
                                       (T*)out, H, W, C, Cp));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_pack_nchw(const float* src, void* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                  int32_t Cp, int32_t dtype, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C || (Cp & 7)) return O2M_ERR_BAD_ARG;
  const long npix = (long)B * H * W;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(pack_kernel<T>, dim3(grid_for(npix)), dim3(NT), 0, s, src,
                                       (T*)dst, C, H * W, Cp, npix));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_unpack_nhwc(const void* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t Cp, int32_t dtype, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C || (Cp & 7)) return O2M_ERR_BAD_ARG;
  const long npix = (long)B * H * W;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(unpack_kernel<T>, dim3(grid_for(npix)), dim3(NT), 0, s,
                                       (const T*)src, dst, C, H * W, Cp, npix));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_wgrad_finalize(float* acc, float* gq, const float* w32, float* grad, int32_t Co, int32_t Ci,
                       int32_t KK, int32_t Cop, int32_t Cip, float c, void* stream) {
  if (!acc || !grad || Co <= 0 || Ci <= 0 || KK <= 0 || Cop < Co || Cip < Ci || (gq && !w32))
    return O2M_ERR_BAD_ARG;
  const long n = (long)Cop * KK * Cip;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(grid_for(n)), dim3(NT), 0, s, acc, gq, w32, grad, Co, Ci, KK,
                     Cip, c, n);
  O2M_LAUNCH_CHECK();
  if (gq) {
    hipLaunchKernelGGL(clear_kernel, dim3(grid_for((long)Cop * Cip)), dim3(NT), 0, s, gq, (long)Cop * Cip);
    O2M_LAUNCH_CHECK();
  }
  return 0;
}

int32_t o2m_wgrad_finalize_blocks(int32_t Cop, int32_t KK, int32_t Cip) {
  if (Cop <= 0 || KK <= 0 || Cip <= 0) return 0;
  return (int32_t)(((long)Cop * KK * Cip + FIN_PER_BLOCK - 1) / FIN_PER_BLOCK);
}

int o2m_wgrad_finalize_batched(const o2m_wfin_job* jobs, int32_t n_jobs, int32_t total_blocks, int32_t any_gq, void* stream) {
  if (!jobs || n_jobs <= 0 || total_blocks <= 0) return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(wgrad_finalize_batched_kernel, dim3((unsigned)total_blocks), dim3(NT), 0, s, jobs, n_jobs);
  O2M_LAUNCH_CHECK();
  if (any_gq) {
    hipLaunchKernelGGL(wgrad_clear_gq_batched_kernel, dim3((unsigned)total_blocks), dim3(NT), 0, s, jobs, n_jobs);
    O2M_LAUNCH_CHECK();
  }
  return 0;
}

int32_t o2m_reduce_blocks(int64_t n) {
  if (n <= 0 || (n & 7)) return 0;
  const long nvec = n / 8;
  return (int32_t)((nvec + RED_VEC_PER_BLOCK - 1) / RED_VEC_PER_BLOCK);
}

int o2m_reduce_fwd(const void* a, const void* b, const float* w, float* partials, int32_t B,
                   int64_t n_per_sample, int32_t mode, int32_t dtype, void* stream) {
  if (!a || !partials || B <= 0 || n_per_sample <= 0 || (n_per_sample & 7)) return O2M_ERR_BAD_ARG;
  if (mode < O2M_RED_L1 || mode > O2M_RED_MOM) return O2M_ERR_BAD_ARG;
  const long nvec = (long)B * n_per_sample / 8;
  const int nblocks = o2m_reduce_blocks((int64_t)B * n_per_sample);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(reduce_fwd_kernel<T>, dim3(nblocks), dim3(NT), 0, s, (const T*)a,
                                       (const T*)b, w, partials, (long)(n_per_sample / 8), nvec, mode,
                                       nblocks));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_lsgan_fwd(const void* scores, float* out, int32_t N, int32_t P, int32_t C, int32_t n_first, float t0, float t1,
                  int32_t dtype, void* stream) {
  if (!scores || !out || N <= 0 || P <= 0 || C <= 0 || (C & 7) || n_first < 0 || n_first > N) return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(lsgan_fwd_kernel<T>, dim3(1), dim3(1024), 0, s, (const T*)scores, out, N, P, C, n_first, t0, t1));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_lsgan_bwd(const void* scores, const float* coef, void* g_scores, int32_t N, int32_t P, int32_t C, int32_t n_first,
                  float t0, float t1, int32_t dtype, void* stream) {
  if (!scores || !coef || !g_scores || N <= 0 || P <= 0 || C <= 0 || (C & 7) || n_first < 0 || n_first > N) return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long total = (long)N * P;
  DISPATCH_T(dtype, hipLaunchKernelGGL(lsgan_bwd_kernel<T>, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, s, (const T*)scores,
                                       coef, (T*)g_scores, total, (long)n_first * P, C, t0, t1));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_reduce_bwd(const void* a, const void* b, const float* w, const float* coef, void* ga,
                   int32_t B, int64_t n_per_sample, int32_t mode, int32_t dtype, void* stream) {
  if (!a || !coef || !ga || B <= 0 || n_per_sample <= 0 || (n_per_sample & 7)) return O2M_ERR_BAD_ARG;
  if (mode < O2M_RED_L1 || mode > O2M_RED_MOM) return O2M_ERR_BAD_ARG;
  const long nvec = (long)B * n_per_sample / 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(reduce_bwd_kernel<T>, dim3(grid_for(nvec)), dim3(NT), 0, s,
                                       (const T*)a, (const T*)b, w, coef, (T*)ga,
                                       (long)(n_per_sample / 8), nvec, mode));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_pair_grad(const void* a, const void* b, const float* w, const float* coef, const void* gin_a, const void* gin_b,
                  void* ga, void* gb, int32_t B, int64_t n_per_sample, int32_t dtype, void* stream) {
  if (!a || !b || !coef || !ga || !gb || B <= 0 || n_per_sample <= 0 || (n_per_sample & 7)) return O2M_ERR_BAD_ARG;
  const long nvec = (long)B * n_per_sample / 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(pair_grad_kernel<T>, dim3(grid_for(nvec)), dim3(NT), 0, s, (const T*)a, (const T*)b, w,
                                       coef, (const T*)gin_a, (const T*)gin_b, (T*)ga, (T*)gb, (long)(n_per_sample / 8), nvec));
  O2M_LAUNCH_CHECK();
  return 0;
}

int o2m_adam_step(float* p, const float* g, float* m, float* v, const float* step, int64_t n,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !step || n <= 0) return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(NT), 0, s, p, g, m, v, step, (long)n, lr,
                     beta1, beta2, eps, grad_scale);
  O2M_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
