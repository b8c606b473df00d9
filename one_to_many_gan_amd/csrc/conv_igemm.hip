// Implicit-GEMM convolution forward / data-gradient on MFMA for gfx950.
//
//   GEMM view:  M = B*Ho*Wo (output pixels), N = Co, K = KH*KW*Ci (ci fastest).
//   A[m][k] is gathered on the fly from the NHWC input (im2col never materialised);
//   zero or reflect padding is index arithmetic in the loader; the per-(b,ci) style scale
//   of the modulated conv is applied while the tile sits in registers, before it is
//   written to LDS.  B[n][k] is the filter, stored [Co][KH][KW][Ci] so both operands are
//   K-contiguous and share one fragment path.
//
//   Block = 256 threads = 4 waves; tile BM x BN x 64; each wave owns a grid of 32x32 MFMA
//   tiles (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Global -> registers -> LDS staging
//   (16 B per lane, issue-early / write-late, two LDS stages), XOR-swizzled 16-B slots so
//   the ds_read_b128 fragment reads are bank-conflict free.
//
//   dtype O2M_F32 ("parity mode"): operands are split hi + lo into two bf16 tiles while
//   staging and every product runs as hi*hi + hi*lo + lo*hi on the bf16 MFMA: ~2^-16
//   relative error per product at 3/16 the cost of the fp32 MFMA.
#include "common.h"

namespace {

constexpr int BK = 64;  // reduction elements per stage (one 128-B LDS row of bf16)
constexpr int NT = 256;

// Byte offset of 16-B slot (row, chunk) in a [rows][64] bf16 tile.  Two 128-B rows share a
// 256-B bank row; XOR with (row>>1)&15 spreads each ds_read_b128 lane group (same chunk,
// 16 different rows) over all 16 slots of the bank row.
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return ((row >> 1) << 8) + (((((row & 1) << 3) | chunk) ^ ((row >> 1) & 15)) << 4);
}

template <typename T> struct Stg;  // staged registers for 8 consecutive reduction elements
template <> struct Stg<unsigned short> { u32x4 v; };
template <> struct Stg<float> { f32x4 a, b; };

__device__ __forceinline__ void stg_zero(Stg<unsigned short>& s) { s.v = u32x4{0, 0, 0, 0}; }
__device__ __forceinline__ void stg_zero(Stg<float>& s) {
  s.a = f32x4{0.f, 0.f, 0.f, 0.f};
  s.b = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void stg_load(Stg<unsigned short>& s, const unsigned short* p) {
  s.v = *reinterpret_cast<const u32x4*>(p);
}
__device__ __forceinline__ void stg_load(Stg<float>& s, const float* p) {
  s.a = *reinterpret_cast<const f32x4*>(p);
  s.b = *reinterpret_cast<const f32x4*>(p + 4);
}
__device__ __forceinline__ void stg_unpack(const Stg<unsigned short>& s, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __builtin_bit_cast(float, s.v[i] << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, s.v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void stg_unpack(const Stg<float>& s, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[i] = s.a[i];
    f[4 + i] = s.b[i];
  }
}

// write 8 values to the tile(s): bf16 -> one tile; fp32 -> hi tile and lo tile
template <typename T, bool SCALED>
__device__ __forceinline__ void stg_to_lds(const Stg<T>& s, const f32x4 (&sc)[2], char* hi,
                                           char* lo, int off) {
  if constexpr (sizeof(T) == 2 && !SCALED) {
    *reinterpret_cast<u32x4*>(hi + off) = s.v;
  } else {
    float f[8];
    stg_unpack(s, f);
    if constexpr (SCALED) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[i] *= sc[0][i]; f[4 + i] *= sc[1][i]; }
    }
    u32x4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    *reinterpret_cast<u32x4*>(hi + off) = h;
    if constexpr (sizeof(T) == 4) {
      u32x4 l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float r0 = f[2 * i] - __builtin_bit_cast(float, h[i] << 16);
        float r1 = f[2 * i + 1] - __builtin_bit_cast(float, h[i] & 0xffff0000u);
        l[i] = pack_bf2(r0, r1);
      }
      *reinterpret_cast<u32x4*>(lo + off) = l;
    }
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool IN_SCALE>
__global__ __launch_bounds__(NT, 2) void conv_igemm_kernel(const o2m_conv_desc d) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int NPLANE = F32 ? 2 : 1;  // hi (+ lo)
  constexpr int NSTAGE = F32 ? 1 : 2;
  constexpr int RA = BM / 32, RB = BN / 32;  // 16-B loads per thread per stage
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = NPLANE * (A_BYTES + B_BYTES);
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const T* __restrict__ X = static_cast<const T*>(d.x);
  const T* __restrict__ Wt = static_cast<const T*>(d.w);
  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, KH = d.KH, KW = d.KW, pad = d.pad;
  const int Ho = H + 2 * pad - KH + 1, Wo = W + 2 * pad - KW + 1;
  const int HoWo = Ho * Wo;
  const int M = d.B * HoWo;
  const int K = KH * KW * Ci;
  const int nk = (K + BK - 1) / BK;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;

  const int tiles_n = (Co + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM;
  const int n0 = (blockIdx.x % tiles_n) * BN;

  const int tid = threadIdx.x;
  const int cc = tid & 7;   // 16-B chunk (8 reduction elements) inside the 64-wide stage
  const int r0 = tid >> 3;  // 0..31

  // ---- per-thread row bookkeeping for the A gather ---------------------------------
  int rb[RA], roy[RA], rox[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    int m = m0 + r0 + 32 * j;
    if (m < M) {
      int b = m / HoWo, rem = m - b * HoWo;
      rb[j] = b;
      roy[j] = rem / Wo;
      rox[j] = rem - roy[j] * Wo;
    } else {
      rb[j] = -1; roy[j] = 0; rox[j] = 0;
    }
  }

  Stg<T> sa[RA], sb[RB];
  f32x4 sca[IN_SCALE ? RA : 1][2];

  auto load_tiles = [&](int kt) {
    const int k = kt * BK + cc * 8;
    const bool kv = k < K;
    const int tap = k / Ci, ci0 = k - tap * Ci;
    const int kh = tap / KW, kw = tap - kh * KW;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      int iy = roy[j] + kh - pad, ix = rox[j] + kw - pad;
      bool ok = kv && rb[j] >= 0;
      if (reflect) {
        iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
      } else {
        ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      }
      if (ok) {
        size_t off = ((size_t)(rb[j] * H + iy) * W + ix) * Ci + ci0;
        stg_load(sa[j], X + off);
      } else {
        stg_zero(sa[j]);
      }
      if constexpr (IN_SCALE) {
        if (ok) {
          const float* sp = d.in_scale + (size_t)rb[j] * Ci + ci0;
          sca[j][0] = *reinterpret_cast<const f32x4*>(sp);
          sca[j][1] = *reinterpret_cast<const f32x4*>(sp + 4);
        } else {
          sca[j][0] = sca[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      int n = n0 + r0 + 32 * j;
      if (kv && n < Co) stg_load(sb[j], Wt + (size_t)n * K + k);
      else stg_zero(sb[j]);
    }
  };

  auto store_tiles = [&](int stage) {
    char* base = smem + stage * STAGE_BYTES;
    char* a_hi = base;
    char* a_lo = base + A_BYTES;                  // only used when F32
    char* b_hi = base + NPLANE * A_BYTES;
    char* b_lo = b_hi + B_BYTES;
    const f32x4 none[2] = {};
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      if constexpr (IN_SCALE)
        stg_to_lds<T, true>(sa[j], sca[j], a_hi, a_lo, tile_off(r0 + 32 * j, cc));
      else
        stg_to_lds<T, false>(sa[j], none, a_hi, a_lo, tile_off(r0 + 32 * j, cc));
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) stg_to_lds<T, false>(sb[j], none, b_hi, b_lo, tile_off(r0 + 32 * j, cc));
  };

  // ---- accumulators ------------------------------------------------------------------
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const char* base = smem + stage * STAGE_BYTES;
    const char* a_hi = base;
    const char* a_lo = base + A_BYTES;
    const char* b_hi = base + NPLANE * A_BYTES;
    const char* b_lo = b_hi + B_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int chunk = ks * 2 + lh;
      bf16x8 ah[TM], bh[TN], al[F32 ? TM : 1], bl[F32 ? TN : 1];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int off = tile_off(wm + i * 32 + lr, chunk);
        ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + off);
        if constexpr (F32) al[i] = *reinterpret_cast<const bf16x8*>(a_lo + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int off = tile_off(wn + j * 32 + lr, chunk);
        bh[j] = *reinterpret_cast<const bf16x8*>(b_hi + off);
        if constexpr (F32) bl[j] = *reinterpret_cast<const bf16x8*>(b_lo + off);
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
  };

  // ---- main loop: issue-early / write-late register staging --------------------------
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = NSTAGE == 2 ? (kt & 1) : 0;
    const bool more = kt + 1 < nk;
    if (more) load_tiles(kt + 1);
    compute(cur);
    if constexpr (NSTAGE == 1) __syncthreads();
    if (more) store_tiles(NSTAGE == 2 ? (cur ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue: C[row = pixel][col = channel]; lane owns one channel, 16 pixels -----
  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  const int act = d.act;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn + j * 32 + lr;
    if (n >= Co) continue;
    const float bias = d.bias ? d.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (d.out_scale) v *= d.out_scale[(size_t)(m / HoWo) * Co + n];
        v = act_fwd(v + bias, act);
        const size_t o = (size_t)m * Co + n;
        if (R) v += Elem<T>::ld(R + o);
        Elem<T>::st(Y + o, v);
      }
    }
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const o2m_conv_desc& d, hipStream_t s) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int lds = (F32 ? 2 : 1) * (F32 ? 1 : 2) * (BM + BN) * BK * 2;
  const int Ho = d.H + 2 * d.pad - d.KH + 1, Wo = d.W + 2 * d.pad - d.KW + 1;
  const long M = (long)d.B * Ho * Wo;
  const long tiles = ((M + BM - 1) / BM) * ((d.Co + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), lds, s, d);
  };
  if (d.in_scale) go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, true>);
  else go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false>);
  O2M_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int launch_dtype(const o2m_conv_desc& d, hipStream_t s) {
  if (d.Co > 64) return launch_cfg<T, 128, 128, 2, 2>(d, s);
  if (d.Co > 32) return launch_cfg<T, 128, 64, 2, 2>(d, s);
  return launch_cfg<T, 256, 32, 4, 1>(d, s);
}

}  // namespace

extern "C" int o2m_conv2d_fwd(const o2m_conv_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y) return O2M_ERR_BAD_ARG;
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
