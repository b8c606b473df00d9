// Convolution weight gradient on MFMA for gfx950.
//
//   dW[co][k] += sum_m G[m][co] * A[m][k],   m = (b,oy,ox) output pixel, k = (kh,kw,ci)
//
// The reduction index m is the SLOW index of both operands in memory (NHWC: channels are
// contiguous), while the MFMA wants 8 consecutive reduction elements per lane.  Both tiles
// are therefore staged row-major exactly as they sit in HBM ([32 pixels][channels], 16-B
// coalesced loads, no transposition on the way in) and the fragments are fetched with the
// CDNA4 transposing LDS read ds_read_b64_tr_b16: two reads give one lane its 8 consecutive
// pixels of one channel.  Row stride 2*cols+64 B keeps the four rows of one tr-read on
// disjoint 64-B bank windows.
//
// Grid = co-tiles x k-tiles x splits; each block reduces its slice of the pixels into a
// 32x32-tiled fp32 accumulator and adds it to dW with global_atomic_add_f32 (128-B row
// segments per half-wave: the full-rate atomic shape).  fp32 mode: bf16x3 split as in
// conv_igemm.hip.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int BMR = 32;  // pixels per stage

template <typename T, bool SCALED>
__device__ __forceinline__ void put8(const T* src, bool ok, const float* sc, char* hi, char* lo,
                                     int off) {
  constexpr bool F32 = sizeof(T) == 4;
  if (!ok) {
    *reinterpret_cast<u32x4*>(hi + off) = u32x4{0, 0, 0, 0};
    if constexpr (F32) *reinterpret_cast<u32x4*>(lo + off) = u32x4{0, 0, 0, 0};
    return;
  }
  if constexpr (!F32 && !SCALED) {
    *reinterpret_cast<u32x4*>(hi + off) = *reinterpret_cast<const u32x4*>(src);
  } else {
    float f[8];
    load8(src, f);
    if constexpr (SCALED) {
      f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[i] *= s0[i]; f[4 + i] *= s1[i]; }
    }
    u32x4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    *reinterpret_cast<u32x4*>(hi + off) = h;
    if constexpr (F32) {
      u32x4 l;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        l[i] = pack_bf2(f[2 * i] - __builtin_bit_cast(float, h[i] << 16),
                        f[2 * i + 1] - __builtin_bit_cast(float, h[i] & 0xffff0000u));
      *reinterpret_cast<u32x4*>(lo + off) = l;
    }
  }
}

// fragment: column `col` of a row-major [32][cols] bf16 image, rows 8*lh .. 8*lh+7
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int stride, int colbase, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int row = 8 * (g >> 1) + q;
  const int col = colbase + 16 * (g & 1) + 4 * p;
  const char* a = img + row * stride + col * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * stride));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <typename T, int BCO, int BKO, int WAVES_CO, int WAVES_K, bool XS, bool GS>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_kernel(const o2m_wgrad_desc d, int tiles_co,
                                                           int tiles_k, int rows_per_split) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int NPLANE = F32 ? 2 : 1;
  constexpr int GSTR = BCO * 2 + 64, XSTR = BKO * 2 + 64;  // row strides (bytes)
  constexpr int G_BYTES = BMR * GSTR, X_BYTES = BMR * XSTR;
  constexpr int STAGE_BYTES = NPLANE * (G_BYTES + X_BYTES);
  constexpr int WCO = BCO / WAVES_CO, WK = BKO / WAVES_K;
  constexpr int TM = WCO / 32, TN = WK / 32;
  constexpr int GCH = BCO / 8;                        // 16-B chunks per G row
  constexpr int GLD = (BMR * GCH + NT - 1) / NT;      // G loads per thread per stage
  constexpr int XLD = BMR * (BKO / 8) / NT;           // X loads per thread per stage (=2)
  static_assert(WAVES_CO * WAVES_K == 4 && BKO == 128, "layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const T* __restrict__ X = static_cast<const T*>(d.x);
  const T* __restrict__ G = static_cast<const T*>(d.gy);
  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, KH = d.KH, KW = d.KW, pad = d.pad;
  const int Ho = H + 2 * pad - KH + 1, Wo = W + 2 * pad - KW + 1;
  const int HoWo = Ho * Wo;
  const int M = d.B * HoWo;
  const int K = KH * KW * Ci;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;

  int bid = xcd_tile_order(blockIdx.x, gridDim.x);  // a split's tiles share x / gy: one XCD
  const int tk = bid % tiles_k; bid /= tiles_k;
  const int tco = bid % tiles_co; bid /= tiles_co;
  const int split = bid;
  const int co0 = tco * BCO, k0 = tk * BKO;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;  // uniform per block

  const int tid = threadIdx.x;
  // X gather: this thread always fetches the same 8 reduction columns (tap, ci0)
  const int xc = tid & 15, xr0 = tid >> 4;  // rows xr0 and xr0+16
  const int kx = k0 + xc * 8;
  const bool kxv = kx < K;
  const int tap = kx / Ci, ci0 = kx - tap * Ci;
  const int kh = tap / KW, kw = tap - kh * KW;

  // pixel coordinates of this thread's two X rows, advanced incrementally per stage
  int pb[XLD], py[XLD], px[XLD];
#pragma unroll
  for (int j = 0; j < XLD; ++j) {
    int m = m_begin + xr0 + 16 * j;
    pb[j] = m / HoWo;
    int rem = m - pb[j] * HoWo;
    py[j] = rem / Wo;
    px[j] = rem - py[j] * Wo;
  }

  const int wave = tid >> 6, lane = tid & 63;
  const int wco = (wave / WAVES_K) * WCO, wk = (wave % WAVES_K) * WK;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  char* g_hi = smem;
  char* g_lo = smem + G_BYTES;
  char* x_hi = smem + NPLANE * G_BYTES;
  char* x_lo = x_hi + X_BYTES;

  for (int ms = m_begin; ms < m_end; ms += BMR) {
    // ---- stage: global -> LDS (row-major, as in HBM) ---------------------------------
#pragma unroll
    for (int j = 0; j < GLD; ++j) {
      const int idx = tid + NT * j;
      if (idx < BMR * GCH) {
        const int row = idx / GCH, c = idx - row * GCH;
        const int m = ms + row, co = co0 + c * 8;
        const bool ok = m < m_end && co < Co;
        const float* sc = nullptr;
        if constexpr (GS) sc = d.gy_scale + (size_t)(ok ? m / HoWo : 0) * Co + (ok ? co : 0);
        put8<T, GS>(G + (size_t)m * Co + co, ok, sc, g_hi, g_lo, row * GSTR + c * 16);
      }
    }
#pragma unroll
    for (int j = 0; j < XLD; ++j) {
      const int row = xr0 + 16 * j;
      const int m = ms + row;
      int iy = py[j] + kh - pad, ix = px[j] + kw - pad;
      bool ok = kxv && m < m_end;
      if (reflect) {
        iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
      } else {
        ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      }
      const size_t off = ok ? ((size_t)(pb[j] * H + iy) * W + ix) * Ci + ci0 : 0;
      const float* sc = nullptr;
      if constexpr (XS) sc = d.in_scale + (size_t)(ok ? pb[j] : 0) * Ci + ci0;
      put8<T, XS>(X + off, ok, sc, x_hi, x_lo, row * XSTR + xc * 16);
      // advance this row by one stage (BMR pixels)
      px[j] += BMR;
      while (px[j] >= Wo) { px[j] -= Wo; ++py[j]; }
      while (py[j] >= Ho) { py[j] -= Ho; ++pb[j]; }
    }
    __syncthreads();

    // ---- MFMA: D[co][k] += G^T . X over this stage's 32 pixels ------------------------
#pragma unroll
    for (int ks = 0; ks < BMR / 16; ++ks) {
      const int roff = ks * 16;
      bf16x8 ah[TM], bh[TN], al[F32 ? TM : 1], bl[F32 ? TN : 1];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = tr_frag(g_hi + roff * GSTR, GSTR, wco + i * 32, lane);
        if constexpr (F32) al[i] = tr_frag(g_lo + roff * GSTR, GSTR, wco + i * 32, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = tr_frag(x_hi + roff * XSTR, XSTR, wk + j * 32, lane);
        if constexpr (F32) bl[j] = tr_frag(x_lo + roff * XSTR, XSTR, wk + j * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (F32) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns one k column; fp32 atomics, 128-B segments per half-wave ----
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = k0 + wk + j * 32 + lr;
    if (k >= K) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < Co) atomicAdd(d.dw + (size_t)co * K + k, acc[i][j][r]);
      }
  }
}

template <typename T, int BCO, int WAVES_CO, int WAVES_K>
int launch_cfg(const o2m_wgrad_desc& d, hipStream_t s) {
  constexpr int BKO = 128;
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int lds = (F32 ? 2 : 1) * BMR * ((BCO * 2 + 64) + (BKO * 2 + 64));
  const int Ho = d.H + 2 * d.pad - d.KH + 1, Wo = d.W + 2 * d.pad - d.KW + 1;
  const long M = (long)d.B * Ho * Wo;
  const int K = d.KH * d.KW * d.Ci;
  const int tiles_co = (d.Co + BCO - 1) / BCO, tiles_k = (K + BKO - 1) / BKO;
  long splits = d.splits;
  if (splits <= 0) {
    splits = 2048 / ((long)tiles_co * tiles_k);       // ~8 blocks per CU in flight
    const long max_splits = (M + 4 * BMR - 1) / (4 * BMR);  // >= 4 stages per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  long rows = (M + splits - 1) / splits;
  rows = (rows + BMR - 1) / BMR * BMR;
  splits = (M + rows - 1) / rows;
  const long blocks = splits * tiles_co * tiles_k;
  if (blocks > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, s, d, tiles_co, tiles_k, (int)rows);
  };
  const bool xs = d.in_scale != nullptr, gs = d.gy_scale != nullptr;
  if (xs && gs) go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, true, true>);
  else if (xs) go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, true, false>);
  else if (gs) go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, false, true>);
  else go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, false, false>);
  O2M_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int launch_dtype(const o2m_wgrad_desc& d, hipStream_t s) {
  if (d.Co > 64) return launch_cfg<T, 128, 2, 2>(d, s);
  if (d.Co > 32) return launch_cfg<T, 64, 2, 2>(d, s);
  return launch_cfg<T, 32, 1, 4>(d, s);
}

}  // namespace

extern "C" int o2m_conv2d_wgrad(const o2m_wgrad_desc* d, void* stream) {
  if (!d || !d->x || !d->gy || !d->dw) return O2M_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ci <= 0 || d->Co <= 0) return O2M_ERR_BAD_ARG;
  if ((d->Ci & 7) || (d->Co & 7) || d->KH <= 0 || d->KW <= 0 || d->pad < 0) return O2M_ERR_BAD_ARG;
  if (d->H + 2 * d->pad < d->KH || d->W + 2 * d->pad < d->KW) return O2M_ERR_BAD_ARG;
  if (d->pad_mode == O2M_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W)) return O2M_ERR_BAD_ARG;
  if (d->pad_mode != O2M_PAD_ZERO && d->pad_mode != O2M_PAD_REFLECT) return O2M_ERR_BAD_ARG;
  if ((long)d->B * d->H * d->W * (long)d->Ci > 0x7fffffffL) return O2M_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == O2M_BF16) return launch_dtype<unsigned short>(*d, s);
  if (d->dtype == O2M_F32) return launch_dtype<float>(*d, s);
  return O2M_ERR_BAD_ARG;
}
