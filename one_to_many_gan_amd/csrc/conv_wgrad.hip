// Convolution weight gradient on MFMA for gfx950.
//
//   dW[co][k] += sum_m G[m][co] * A[m][k],   m = (b,oy,ox) output pixel, k = (kh,kw,ci)
//
// The reduction index m is the SLOW index of both operands in memory (NHWC: channels are
// contiguous), while the MFMA wants 8 consecutive reduction elements per lane.  Both tiles
// are therefore staged row-major exactly as they sit in HBM ([32 pixels][channels], 16-B
// coalesced buffer loads, no transposition on the way in) and the fragments are fetched with
// the CDNA4 transposing LDS read ds_read_b64_tr_b16: two reads give one lane its 8
// consecutive pixels of one channel.  Row stride 2*cols+64 B keeps the four rows of one
// tr-read on disjoint 64-B bank windows (SQ_LDS_BANK_CONFLICT = 0).
//
// Grid = co-tiles x k-tiles x splits, XCD-ordered so that the tiles of one split (which
// share x and gy) run on one L2.  Each block reduces its slice of the pixels, two LDS stages
// of 32 pixels (issue-early / write-late), into 32x32-tiled fp32 accumulators and adds them
// to dW with global_atomic_add_f32 (128-B row segments per half-wave: the full-rate atomic
// shape).  When a 32-pixel stage lies inside one image row (Wo % 32 == 0: every generator
// layer) the sample / row part of the gather address is scalar.  fp32 mode: bf16x3 split as
// in conv_igemm.hip.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int BMR = 32;  // pixels per stage
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned OOB_OFF = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <typename T> struct Stg;
template <> struct Stg<unsigned short> { u32x4 v; };
template <> struct Stg<float> { f32x4 a, b; };
__device__ __forceinline__ void stg_load(Stg<unsigned short>& s, rsrc_t r, unsigned off) {
  s.v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ void stg_load(Stg<float>& s, rsrc_t r, unsigned off) {
  s.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
  s.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off + 16, 0, 0));
}

// 8 staged values -> LDS image(s), optionally scaled by 8 per-channel factors
template <typename T, bool SCALED>
__device__ __forceinline__ void put8(const Stg<T>& s, const float* sc, char* hi, char* lo, int off) {
  constexpr bool F32 = sizeof(T) == 4;
  if constexpr (!F32 && !SCALED) {
    *reinterpret_cast<u32x4*>(hi + off) = s.v;
  } else {
    float f[8];
    if constexpr (F32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[i] = s.a[i]; f[4 + i] = s.b[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, s.v[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, s.v[i] & 0xffff0000u);
      }
    }
    if constexpr (SCALED) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
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

// fragment = 8 consecutive rows (pixels) of one column of a row-major bf16 image.
// `a` already points at this lane's (row 8*lh + q, column 16*(g&1) + 4p) of the 32x32 block.
template <int STRIDE>
__device__ __forceinline__ bf16x8 tr_frag(const char* a) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * STRIDE));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// ds_read_b64_tr_b16 as INLINE ASM.  Through the builtin, hipcc (ROCm 7.2) puts an `s_waitcnt vmcnt(0)` in front of every
// group of transposing reads that follows an LDS-DMA fill (the builtin's memory operand carries no alias information, so
// SIInsertWaitcnts assumes it may read what the DMA writes): the counted-vmcnt prefetch of the phase-pipelined kernels
// below was drained at every phase that reads fragments.  The asm form is invisible to that pass -- and to its lgkmcnt
// bookkeeping: the caller waits (`lds_reads_done`) before the first MFMA that consumes the registers.
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(size_t)(lds_char*)p; }
template <int OFF>
__device__ __forceinline__ s16x4 ds_tr(unsigned a) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void lds_reads_done() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // MFMAs touch no memory: the clobber alone does not keep them behind the wait
}

template <typename T, int BCO, int BKO, int WAVES_CO, int WAVES_K, bool XS, bool GS, bool ALIGNED>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_kernel(const o2m_wgrad_desc d, int tiles_co,
                                                           int tiles_k, int rows_per_split) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int ES = sizeof(T);
  constexpr int NPLANE = F32 ? 2 : 1;
  constexpr int NSTAGE = 2;
  constexpr int GSTR = BCO * 2 + 64, XSTR = BKO * 2 + 64;  // row strides (bytes)
  constexpr int G_BYTES = BMR * GSTR, X_BYTES = BMR * XSTR;
  constexpr int STAGE_BYTES = NPLANE * (G_BYTES + X_BYTES);
  constexpr int WCO = BCO / WAVES_CO, WK = BKO / WAVES_K;
  constexpr int TM = WCO / 32, TN = WK / 32;
  constexpr int GCH = BCO / 8;                    // 16-B chunks per G row
  constexpr int GLD = (BMR * GCH + NT - 1) / NT;  // G loads per thread per stage
  constexpr int GRS = NT / GCH;                   // G rows covered per pass
  constexpr int XCH = BKO / 8;                    // 16-B chunks per X row
  constexpr int XLD = BMR * XCH / NT;             // X loads per thread per stage
  constexpr int XRS = NT / XCH;                   // X rows covered per pass
  static_assert(WAVES_CO * WAVES_K == 4 && NT % GCH == 0 && NT % XCH == 0 && XLD * XRS == BMR, "layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, KH = d.KH, KW = d.KW, pad = d.pad;
  const int st = d.stride > 0 ? d.stride : 1;
  const int Ho = (H + 2 * pad - KH) / st + 1, Wo = (W + 2 * pad - KW) / st + 1;
  const int HoWo = Ho * Wo;
  const int M = d.B * HoWo;
  const int K = KH * KW * Ci;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;
  // ALIGNED (Wo % BMR == 0): a stage never leaves its image row (nor its segment).  A template
  // parameter, not a branch: with both gathers in one kernel their load destinations alias and the
  // compiler serialises the prefetch with conservative vmcnt waits.

  int bid = xcd_tile_order(blockIdx.x, gridDim.x);  // a split's tiles share x / gy: one XCD
  const int tk = bid % tiles_k; bid /= tiles_k;
  const int tco = bid % tiles_co; bid /= tiles_co;
  const int split = bid;
  const int co0 = tco * BCO, k0 = tk * BKO;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;  // uniform per block

  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * ES));
  const rsrc_t gr = make_rsrc(d.gy, (unsigned)((size_t)M * Co * ES));

  const int tid = threadIdx.x;
  // ---- G (upstream gradient) loader: rows grow0 + GRS*j, 8 channels at co0 + gc*8 --------
  const int gc = tid % GCH, grow0 = tid / GCH;
  const bool gcol_ok = co0 + gc * 8 < Co;
  const unsigned gbase = (unsigned)(co0 + gc * 8) * ES;  // + m*Co*ES per row

  // ---- X gather: this thread always fetches the same 8 reduction columns (tap, ci0) -----
  const int xc = tid % XCH, xr0 = tid / XCH;  // rows xr0 + XRS*j
  const int kx = k0 + xc * 8;
  const bool kxv = kx < K;
  const int tap = kx / Ci, ci0 = kx - tap * Ci;
  const int dy = tap / KW - pad, dx = tap - (tap / KW) * KW - pad;

  // generic path: per-row pixel coordinates, advanced incrementally per stage
  int pb[XLD], py[XLD], px[XLD];
#pragma unroll
  for (int j = 0; j < XLD; ++j) {
    const int m = m_begin + xr0 + XRS * j;
    pb[j] = m / HoWo;
    const int rem = m - pb[j] * HoWo;
    py[j] = rem / Wo;
    px[j] = rem - py[j] * Wo;
  }
  // aligned path: (sample, row) of the whole stage are scalar
  int sb = m_begin / HoWo, sy = (m_begin - sb * HoWo) / Wo, sx = m_begin - sb * HoWo - sy * Wo;

  // two register sets: stage i+2 is in flight while stage i+1 waits to be written to LDS
  Stg<T> sg[2][GLD], sx_[2][XLD];
  // XS: the style scale of each staged X row's sample, fetched WITH the row (two more 16-B loads per row in the
  // same always-issued batch, so the loop's waits stay counted; fetched at LDS-write time they made every stage
  // wait for the youngest load, i.e. drained the prefetch)
  f32x4 xsc[2][XS ? XLD : 1][2];

  // Always issued, also past m_end (every lane then reads the out-of-range offset = zeros, no
  // memory traffic): the number of loads in flight stays static, so the waits stay counted.
  const rsrc_t scr = make_rsrc(d.in_scale, XS ? (unsigned)((size_t)d.B * Ci * 4) : 0u);
  auto load_scale = [&](const int slot, const int j, const int samp) {
    const unsigned off = (kxv && samp < d.B) ? (unsigned)(samp * Ci + ci0) * 4u : OOB_OFF;  // past the end: zeros
    xsc[slot][XS ? j : 0][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(scr, (int)off, 0, 0));
    xsc[slot][XS ? j : 0][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(scr, (int)off + 16, 0, 0));
  };
  auto load_stage = [&](const int slot, int ms) {
    const int mloc0 = ms;
#pragma unroll
    for (int j = 0; j < GLD; ++j) {
      const int m = ms + grow0 + GRS * j;
      const bool ok = gcol_ok && (GLD * GRS == BMR || grow0 + GRS * j < BMR) && m < m_end;
      stg_load(sg[slot][j], gr, ok ? gbase + (unsigned)(mloc0 + grow0 + GRS * j) * (unsigned)(Co * ES) : OOB_OFF);
    }
    if constexpr (ALIGNED) {
      int iy = sy * st + dy;  // per-thread only through dy (constant): cheap
      bool rowok = kxv;
      if (reflect) iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
      else rowok = rowok && (unsigned)iy < (unsigned)H;
      const int rowbase = (sb * H + iy) * W;
#pragma unroll
      for (int j = 0; j < XLD; ++j) {
        int ix = (sx + xr0 + XRS * j) * st + dx;
        bool ok = rowok && ms + xr0 + XRS * j < m_end;
        if (reflect) ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
        else ok = ok && (unsigned)ix < (unsigned)W;
        stg_load(sx_[slot][j], xr, ok ? (unsigned)((rowbase + ix) * Ci + ci0) * ES : OOB_OFF);
        if constexpr (XS) load_scale(slot, j, sb);
      }
      sx += BMR;
      if (sx >= Wo) { sx = 0; if (++sy >= Ho) { sy = 0; ++sb; } }
    } else {
#pragma unroll
      for (int j = 0; j < XLD; ++j) {
        int iy = py[j] * st + dy, ix = px[j] * st + dx;
        bool ok = kxv && ms + xr0 + XRS * j < m_end;
        if (reflect) {
          iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
        } else {
          ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        }
        stg_load(sx_[slot][j], xr, ok ? (unsigned)(((pb[j] * H + iy) * W + ix) * Ci + ci0) * ES : OOB_OFF);
        if constexpr (XS) load_scale(slot, j, pb[j]);
        px[j] += BMR;
        while (px[j] >= Wo) { px[j] -= Wo; ++py[j]; }
        while (py[j] >= Ho) { py[j] -= Ho; ++pb[j]; }
      }
    }
  };

  auto store_stage = [&](const int slot, int stage, int ms) {
    char* base = smem + stage * STAGE_BYTES;
    char* g_hi = base;
    char* g_lo = base + G_BYTES;
    char* x_hi = base + NPLANE * G_BYTES;
    char* x_lo = x_hi + X_BYTES;
#pragma unroll
    for (int j = 0; j < GLD; ++j) {
      if (GLD * GRS != BMR && grow0 + GRS * j >= BMR) continue;
      const float* sc = nullptr;
      if constexpr (GS) {
        const int m = min(ms + grow0 + GRS * j, M - 1);
        sc = d.gy_scale + (size_t)(m / HoWo) * Co + min(co0 + gc * 8, Co - 8);
      }
      put8<T, GS>(sg[slot][j], sc, g_hi, g_lo, (grow0 + GRS * j) * GSTR + gc * 16);
    }
#pragma unroll
    for (int j = 0; j < XLD; ++j) {
      float sc[8] = {};
      if constexpr (XS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[q] = xsc[slot][j][0][q]; sc[4 + q] = xsc[slot][j][1][q]; }
      }
      put8<T, XS>(sx_[slot][j], sc, x_hi, x_lo, (xr0 + XRS * j) * XSTR + xc * 16);
    }
  };

  // ---- accumulators and hoisted fragment addresses ----------------------------------------
  const int wave = tid >> 6, lane = tid & 63;
  const int wco = (wave / WAVES_K) * WCO, wk = (wave % WAVES_K) * WK;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  const int frow = 8 * (fg >> 1) + fq, fcol = 16 * (fg & 1) + 4 * fp;
  const int gfrag = frow * GSTR + (wco + fcol) * 2;  // + i*64 + ks*16*GSTR   (immediates)
  const int xfrag = frow * XSTR + (wk + fcol) * 2;   // + j*64 + ks*16*XSTR

  auto compute = [&](int stage) {
    const char* base = smem + stage * STAGE_BYTES;
    const char* g_hi = base + gfrag;
    const char* g_lo = g_hi + G_BYTES;
    const char* x_hi = base + NPLANE * G_BYTES + xfrag;
    const char* x_lo = x_hi + X_BYTES;
#pragma unroll
    for (int ks = 0; ks < BMR / 16; ++ks) {
      bf16x8 ah[TM], bh[TN], al[F32 ? TM : 1], bl[F32 ? TN : 1];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = tr_frag<GSTR>(g_hi + ks * 16 * GSTR + i * 64);
        if constexpr (F32) al[i] = tr_frag<GSTR>(g_lo + ks * 16 * GSTR + i * 64);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = tr_frag<XSTR>(x_hi + ks * 16 * XSTR + j * 64);
        if constexpr (F32) bl[j] = tr_frag<XSTR>(x_lo + ks * 16 * XSTR + j * 64);
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

  // ---- main loop ------------------------------------------------------------------------------
  // prefetch distance 2: global loads of stage i+2 are issued before the MFMAs of stage i, the
  // registers of stage i+1 (issued one iteration earlier) go to LDS after them
  load_stage(0, m_begin);
  load_stage(1, m_begin + BMR);
  store_stage(0, 0, m_begin);
  __syncthreads();
  // No break inside the loop: an exit edge from the middle would reach the loop header with the
  // first half's loads pending and make the compiler wait for them there.
  const int nst = (m_end - m_begin + BMR - 1) / BMR;
  int ms = m_begin;
  for (int pr = nst >> 1; pr > 0; --pr, ms += 2 * BMR) {
    load_stage(0, ms + 2 * BMR);
    compute(0);
    store_stage(1, 1, ms + BMR);
    __syncthreads();
    load_stage(1, ms + 3 * BMR);
    compute(1);
    store_stage(0, 0, ms + 2 * BMR);
    __syncthreads();
  }
  if (nst & 1) compute(0);  // odd stage count: the last stage sits in buffer 0

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
        if (co < Co) {
          // slab mode: this slice's partial is stored (one writer per element), summed in slice order
          // by wgrad_reduce_kernel; else fp32 atomics straight into dw
          if (d.slabs) d.slabs[((size_t)split * Co + co) * K + k] = acc[i][j][r];
          else atomicAdd(d.dw + (size_t)co * K + k, acc[i][j][r]);
        }
      }
  }
}

// dw[i] += sum over the slices, in slice order (fixed summation order: bitwise reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slabs,
                                                           int splits, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + i;
  f32x4 acc = reinterpret_cast<const f32x4*>(dw)[i];
#pragma unroll 4
  for (int s = 0; s < splits; ++s) {
    const f32x4 v = src[(size_t)s * n4];
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  reinterpret_cast<f32x4*>(dw)[i] = acc;
}

// =============================================================================================
// Phase-pipelined weight gradient ("wgrad p8") for the 256 -> 256 3 x 3 layers on 64-pixel-wide maps (every
// ResNet / modulated block of the generator: ~55 % of the step's weight-gradient time).  Same skeleton as
// conv_igemm_p8_kernel -- 256 x 256 tile, 8 waves = 2 x 4, v_mfma_f32_16x16x32_bf16, four phases per K-tile, LDS-DMA
// fills in four regions issued five phases ahead behind ONE counted vmcnt(8), the two wave rows one barrier apart --
// with the GEMM turned on its side:
//   C[co][kcol] += sum_px G[px][co] * X[px (+) tap][kcol],   rows = 256 output channels, columns = the 256 input
//   channels of ONE filter tap (blockIdx -> tap), reduction = pixels, one image row (64 px) per K-tile.
// Both operands sit in HBM with the REDUCTION index slow (NHWC), so the tiles are staged as they are ([64 px][256 B]
// region images) and the fragments come out of LDS with ds_read_b64_tr_b16 (the CDNA4 transposing read): lane i of a
// 16-lane group gets column i of 4 consecutive pixel rows, two reads = the 8 consecutive reduction elements the MFMA
// wants.  The 16-B chunks of a row are XOR-swizzled (source side of the DMA) with ((row & 3) << 1 | (row >> 3 & 1) << 3)
// so the eight pixel rows one half-wave read touches land in eight different 32-B bank windows.
// Every lane's source offset is a constant of the block: a K-tile is one whole image row, so the pixel column, its
// mirror / padding test for the tap's dx and the channel chunk never change; only the row base moves (an SGPR).
// Output: the block's 256 x 256 fp32 tile goes to its slice's slab (o2m_wgrad_desc.slabs); wgrad_reduce_kernel adds
// the slices in order as before (bitwise reproducible).
// =============================================================================================
__global__ __launch_bounds__(512, 2) void conv_wgrad_p8_kernel(const o2m_wgrad_desc d, const int rows_per_split) {
  constexpr int REG = 16384;   // one region image: 64 pixel rows x 256 B
  // LDS image: region rho = 2 * operand + half (A0, A1, B0, B1), K-tile parity p: byte (2 rho + p) * REG -- every
  // fragment read and fill destination is then a compile-time offset from one base register once the K loop is
  // unrolled by two (as in conv_igemm_p8_kernel: no VALU and no parity arithmetic in the loader half of a phase)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, pad = d.pad;
  const int K = 9 * Ci;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;
  const int bid = xcd_tile_order(blockIdx.x, gridDim.x);  // the nine taps of a slice share G and x: one XCD
  const int tap = bid % 9, split = bid / 9;
  const int dy = tap / 3 - pad, dx = tap % 3 - pad;
  const int r_total = d.B * H;  // image rows = K-tiles of the whole problem
  const int r_begin = split * rows_per_split;
  const int nk = min(r_total, r_begin + rows_per_split) - r_begin;
  if (nk <= 0) return;  // uniform
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t gr = make_rsrc(d.gy, (unsigned)((size_t)d.B * H * W * Co * 2));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wrow = wave >> 2, wcol = wave & 3;

  // ---- fills: region fill q (0..15) = pixel rows 4 q .. 4 q + 3; the wave owns fills 2 w, 2 w + 1 of every region.
  // Lane l owns slot (l & 15) of row 4 q + (l >> 4) and fetches the chunk the swizzle maps there.
  unsigned goff[2][2], xoff[2][2];  // [region half][fill of the wave]
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int row = 4 * (2 * wave + jj) + (lane >> 4);
    const int chunk = (lane & 15) ^ (((row & 3) << 1) | (((row >> 3) & 1) << 3));
    // A regions: chunks 0-7 = channels 0-63 (+64 for A1) of wave row 0, chunks 8-15 = the same of wave row 1
    const int co0 = (chunk >> 3) * 128 + (chunk & 7) * 8;
    // B regions: chunk c = columns 64 (c >> 2) + 8 (c & 3) (+32 for B1): the first / second 32 columns of each wave column
    const int kc0 = (chunk >> 2) * 64 + (chunk & 3) * 8;
    int ix = row + dx;  // the pixel column of this row under the tap
    bool ok = true;
    if (reflect) ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
    else ok = (unsigned)ix < (unsigned)W;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      goff[hf][jj] = (unsigned)(row * Co + co0 + 64 * hf) * 2u;
      xoff[hf][jj] = ok ? (unsigned)(ix * Ci + kc0 + 32 * hf) * 2u : OOB_OFF;
    }
  }
  // per-region stream state (wave-uniform): the next K-tile's image row, clamped to the slice's last one -- past the end
  // of the reduction a region re-fetches its last K-tile into the buffer nobody reads any more
  const unsigned g_row = (unsigned)(W * Co * 2), x_row = (unsigned)(W * Ci * 2);
  int a_kt[2] = {0, 0}, b_kt[2] = {0, 0};
  int b_b[2], b_oy[2];  // sample / image row of the B regions' next K-tile
  b_b[0] = b_b[1] = r_begin / H;
  b_oy[0] = b_oy[1] = r_begin - b_b[0] * H;
  const int fill0 = __builtin_amdgcn_readfirstlane(2 * wave * 1024);

  auto issue_a = [&](int r, int buf) {  // (r, buf: literals at every call site)
    const unsigned soff = (unsigned)(r_begin + a_kt[r]) * g_row;
    char* dst = smem + (2 * r + buf) * REG + fill0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_void*)(dst + jj * 1024), 16, (int)goff[r][jj],
                                               (int)__builtin_amdgcn_readfirstlane(soff), 0, 0);
    a_kt[r] = min(a_kt[r] + 1, nk - 1);
  };
  auto issue_b = [&](int r, int buf) {
    int iy = b_oy[r] + dy;
    bool row_ok = true;  // (wave-uniform)
    if (reflect) iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
    else row_ok = (unsigned)iy < (unsigned)H;
    char* dst = smem + 4 * REG + (2 * r + buf) * REG + fill0;
    const unsigned soff = row_ok ? (unsigned)((b_b[r] * H + iy) * W) * (unsigned)(Ci * 2) : 0u;
    if (row_ok) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(dst + jj * 1024), 16, (int)xoff[r][jj],
                                                 (int)__builtin_amdgcn_readfirstlane(soff), 0, 0);
    } else {  // a row of the zero padding: hardware zero fill
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(dst + jj * 1024), 16, (int)OOB_OFF, 0, 0, 0);
    }
    if (b_kt[r] < nk - 1) {
      ++b_kt[r];
      if (++b_oy[r] == H) { b_oy[r] = 0; ++b_b[r]; }
    }
  };

  // ---- fragments (transposing reads) --------------------------------------------------------------------
  // lane = 16 g + 4 q + p: k-group g (pixel rows 8 g .. 8 g + 7 of a 32-row k-step), block row q, column quad p
  const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
  const int fsw = (q << 1) | ((g & 1) << 3);
  const int fa0 = (8 * g + q) * 256 + (((8 * wrow + (pq >> 1)) ^ fsw) << 4) + 8 * (pq & 1);
  const int fb0 = (8 * g + q) * 256 + (((4 * wcol + (pq >> 1)) ^ fsw) << 4) + 8 * (pq & 1) + 4 * REG;
  // (inline-asm transposing reads: through the builtin the compiler drained the counted fill prefetch -- `s_waitcnt
  // vmcnt(0)` -- in front of every fragment read; see ds_tr above.  The waits for the reads themselves are in
  // WP8_WAIT_AND_SYNC.)
  const unsigned lb = lds_addr(smem);
  auto tr8 = [&](const unsigned a, auto off) {  // 8 consecutive pixel rows of this lane's column: two transposing reads
    constexpr int O = decltype(off)::value;
    const s16x4 lo = ds_tr<O>(a), hi = ds_tr<O + 1024>(a);
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  };
  using std::integral_constant;
  bf16x8 af[4][2], b0f[2][2], b1f[2][2];
  auto read_a = [&](int buf, int mh) {
    const unsigned base = lb + (unsigned)((2 * mh + buf) * REG);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned a = base + (unsigned)(fa0 ^ (i << 5));
      af[i][0] = tr8(a, integral_constant<int, 0>{});
      af[i][1] = tr8(a, integral_constant<int, 8192>{});
    }
  };
  auto read_b = [&](bf16x8 (&bf)[2][2], int buf, int nh) {
    const unsigned base = lb + (unsigned)((2 * nh + buf) * REG);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned a = base + (unsigned)(fb0 ^ (j << 5));
      bf[j][0] = tr8(a, integral_constant<int, 0>{});
      bf[j][1] = tr8(a, integral_constant<int, 8192>{});
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const bf16x8 (&bf)[2][2], int mh, int nh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf[j][ks], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define WP8_WAIT_AND_SYNC()                                         \
  do {                                                              \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)

  // ---- prologue / main loop: the schedule of conv_igemm_p8_kernel, region for region ----------------------
  issue_b(0, 0);  // B0(0)
  issue_a(0, 0);  // A0(0)
  issue_b(1, 0);  // B1(0)
  issue_a(1, 0);  // A1(0)
  issue_b(0, 1);  // B0(1)
  issue_a(0, 1);  // A0(1)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // B0(0), A0(0) have landed for every wave
  __builtin_amdgcn_s_barrier();
  if (wrow == 1) __builtin_amdgcn_s_barrier();  // this wave row runs one barrier behind the other
  read_b(b0f, 0, 0);
  auto ktile = [&](int cur) {  // one K-tile out of buffer `cur` (a literal at both call sites)
    read_a(cur, 0);        // p1: A0 x B0
    issue_b(1, cur ^ 1);   // B1(t+1)
    WP8_WAIT_AND_SYNC();
    multiply(b0f, 0, 0);
    __builtin_amdgcn_s_barrier();
    read_b(b1f, cur, 1);   // p2: A0 x B1
    issue_a(1, cur ^ 1);   // A1(t+1)
    WP8_WAIT_AND_SYNC();
    multiply(b1f, 0, 1);
    __builtin_amdgcn_s_barrier();
    read_a(cur, 1);        // p3: A1 x B1
    issue_b(0, cur);       // B0(t+2)
    WP8_WAIT_AND_SYNC();
    multiply(b1f, 1, 1);
    __builtin_amdgcn_s_barrier();
    issue_a(0, cur);       // p4: A1 x B0, then the next K-tile's B0 fragments (landed: waited for at the end of p3)
    WP8_WAIT_AND_SYNC();
    multiply(b0f, 1, 0);
    read_b(b0f, cur ^ 1, 0);
    __builtin_amdgcn_s_barrier();
  };
#pragma unroll 1
  for (int t = 0; t < nk; t += 2) {
    ktile(0);
    if (t + 1 < nk) ktile(1);
  }
#undef WP8_WAIT_AND_SYNC
  if (wrow == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the fills issued past the end of the reduction (and the
                                                               // last prefetched B0 fragments: asm reads, see ds_tr)
  __syncthreads();

  // ---- epilogue: fp32 tile through LDS -> the slice's slab, whole 16-B vectors of a filter row ---------------
  constexpr int CSTR = 256 + 4;
  float* csm = reinterpret_cast<float*>(smem);
  float* __restrict__ slab = d.slabs + ((size_t)split * Co) * K + (size_t)tap * Ci;
  const int ec4 = tid & 63, erow = tid >> 6;  // float4 column, row within the 8 rows of an iteration
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == wrow) {
      // C/D of 16x16x32: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            csm[(i * 16 + 4 * (lane >> 4) + r) * CSTR + wcol * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = erow + it * 8;
      const f32x4 v = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec4 * 4);
      *reinterpret_cast<f32x4*>(slab + (size_t)(pass * 128 + row) * K + ec4 * 4) = v;
    }
    if (pass == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// =============================================================================================
// Halo-tile weight gradient ("wgrad halo") for the 3 x 3 and 4 x 4, zero-padded (pad 1), stride-1 layers with 64-channel
// multiples on both sides that the phase-pipelined kernel above does not take: the 64 <-> 128 layers at 256 x 256 and the
// 128 <-> 256 layers at 128 x 128 of the generator (~40 % of the step's weight-gradient time) and the 4 x 4 trunk of the
// discriminator / style extractor on its odd-sized maps (builder.py:269-283).  The register-staged tiles at the top of
// this file reduce ONE filter tap per block: the input row block is fetched once per tap -- nine (sixteen) L2 -> LDS
// trips per element, 43-64 FLOP per ingested byte -- and they sit on the L2 -> LDS ingest limit at 0.24-0.33 of the MFMA
// peak (profiles/r02_c_pmc_wgrad_co64.json).  Here a block owns a TH x 32 tile of OUTPUT pixels of one sample at a time
// and a (64 output channels) x (64 input channels) block of the filter for ALL taps:
//   C[co][tap][ci] += sum_{px in tile} G[px][co] * X[px (+) tap][ci]
// with the G tile (TH x 32 px x 128 B) and the (TH + KS - 1) x (32 + KS - 1) input patch resident in LDS: every element
// is ingested once per tile (x: + the halo), 250 FLOP per ingested byte.  Tiles are clipped at the map's edge: pixels
// outside G's map or outside x are out-of-range DMA offsets, i.e. hardware zero fill (no divisibility conditions).
// A k-step = one 32-pixel tile row under one tap; the reduction index (pixels) is the SLOW index of both LDS images
// (NHWC as in HBM, LDS-DMA fills), so the fragments come out with ds_read_b64_tr_b16 as in the kernels above.  The 8
// k-values of a lane are pixels 4 g + 0..3 and 16 + 4 g + 0..3 of the row (the MFMA does not care which pixel is which k
// as long as both operands agree): a half-wave read then touches 8 CONSECUTIVE pixel rows, and the 16-B chunks of pixel
// row pp are XOR-swizzled with ((pp >> 1) & 3) << 1 (source side of the DMA) so that those 8 rows' 32-B windows cover
// all eight 32-B windows of the 256-B bank row, at ANY tap shift.
//  * 8 waves = 2 (co tile pairs) x 4 (ci tiles of 16): a wave holds 2 x KS^2 accumulator tiles (72 / 128 registers) --
//    its two G fragments of a tile row serve every tap, one X fragment per tap;
//  * fully unrolled over the tile's rows and taps: every fragment address is a lane-constant register (one per tap
//    column and residue of the patch row: the patch pitch PW is 34 / 36 pixels so that there are only 4 / 2 residues)
//    plus an immediate -- no address arithmetic in the loop.  The reads are INLINE ASM (ds_tr above: no vmcnt(0) drain
//    behind the fills) in units of one tile row (3 x 3: 9 taps) or half a row (4 x 4: 8 taps), software-pipelined by
//    one unit: a unit's 20-22 reads are in flight under the previous unit's 16-18 MFMAs;
//  * ONE block per CU with TWO tile buffers (2 x 75 / 2 x 48 KB of LDS): the fills of tile t + 1 are issued before the
//    MFMAs of tile t and waited for after them -- one barrier per tile;
//  * a block walks its slice of the tiles, then stores its 64 x KS^2 x 64 fp32 partial to its slice's slab;
//    wgrad_reduce_kernel adds the slices in order (bitwise reproducible);
//  * XS (o2m_wgrad_desc.in_scale, the modulated convs; 3 x 3 only): dW = sum_b s[b, ci] * (per-sample partial).  A lane's
//    accumulators all belong to ONE input channel (the MFMA's output column = lane & 15), so the style is a per-lane
//    scalar: the tile loop accumulates the current sample's partial and folds it into the total with 72 FMAs when the
//    sample changes -- the scaled input x * s is never materialised (the epilogue that wrote it for this kernel's
//    predecessor moved 0.8 GB per launch at 256 x 256).
// Measured (round 4, in-step): 1220 TFLOP/s on the 3 x 3 layers (register-staged tiles: 600-820).
// =============================================================================================
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int KS, int TH, int PW, bool XS>
__global__ __launch_bounds__(512, 2) void conv_wgrad_halo_kernel(const o2m_wgrad_desc d, const int tiles_per_slice) {
  static_assert(!XS || KS == 3, "the in-kernel style scale doubles the accumulators: 3 x 3 only");
  constexpr int NTAP = KS * KS, PCOLS = 32 + KS - 1;
  constexpr int NPIX = (TH + KS - 1) * PW, PFILLS = (NPIX + 7) / 8;  // patch pixels, fills of 8
  constexpr int GFILLS = TH * 4;                                      // G tile: TH rows x 32 pixels
  constexpr int PATCH_B = PFILLS * 1024, G_B = GFILLS * 1024;
  constexpr int TILE_B = PATCH_B + G_B;                               // one tile buffer
  constexpr int KYU = KS == 3 ? 3 : 2, UPR = KS / KYU, TPU = KYU * KS;  // filter rows / units per tile row, taps per unit
  constexpr int NU = TH * UPR;                                        // units per tile
  constexpr int NV = PW % 8 == 4 ? 2 : (PW % 4 == 2 ? 4 : 8);         // residues of (patch row * PW) & 7
  static_assert(PW >= PCOLS && NU % 2 == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, pad = d.pad;
  const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
  const int K = NTAP * Ci;
  const int nci = Ci >> 6, pairs = (Co >> 6) * nci;
  const int bid = xcd_tile_order(blockIdx.x, gridDim.x);  // the filter blocks of one slice share G and x: one XCD
  const int pair = bid % pairs, slice = bid / pairs;
  const int co0 = (pair / nci) << 6, ci0 = (pair % nci) << 6;
  const int tiles_x = (Wo + 31) >> 5, tpi = tiles_x * ((Ho + TH - 1) / TH), total = d.B * tpi;
  const int t_begin = slice * tiles_per_slice, t_end = min(total, t_begin + tiles_per_slice);
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t gr = make_rsrc(d.gy, (unsigned)((size_t)d.B * Ho * Wo * Co * 2));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

  // ---- fills: a wave-instruction fills 8 pixels x 128 B; lane l owns slot (l & 7) of pixel 8 f + (l >> 3) and fetches
  // the chunk the swizzle maps there.  Offsets are recomputed per tile.
  auto issue_tile = [&](int t, int buf) {
    const int b = t / tpi, tis = t - b * tpi;
    const int ty0 = (tis / tiles_x) * TH, tx0 = (tis % tiles_x) << 5;
    char* patch = smem + buf * TILE_B;
    char* gbuf = patch + PATCH_B;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // keeps the offset arithmetic inside the tile loop
#pragma unroll
    for (int j = 0; j < (PFILLS + 7) / 8; ++j) {
      const int f = 8 * j + wave;
      if (f >= PFILLS) continue;  // wave-uniform
      const int pp = 8 * f + (ln >> 3);
      const int c = (ln & 7) ^ (((pp >> 1) & 3) << 1);
      const int py = pp / PW, px = pp - py * PW;
      const int gy = ty0 + py - pad, gx = tx0 + px - pad;
      const bool ok = pp < NPIX && px < PCOLS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((b * H + gy) * W + gx) * Ci + ci0 + c * 8) * 2u : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(patch + f * 1024), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < (GFILLS + 7) / 8; ++j) {
      const int f = 8 * j + wave;  // tile pixels 8 f .. 8 f + 7: row f >> 2, columns 8 (f & 3) ..
      if (f >= GFILLS) continue;
      const int pp = 8 * f + (ln >> 3);
      const int c = (ln & 7) ^ (((pp >> 1) & 3) << 1);
      const int oy = ty0 + (pp >> 5), ox = tx0 + (pp & 31);
      const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)(((b * Ho + oy) * Wo + ox) * Co + co0 + c * 8) * 2u : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_void*)(gbuf + f * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  // ---- fragment addresses (transposing reads): lane = 16 g + 4 q + p ---------------------------------------------
  // k-group g of a 32-pixel row: pixels 4 g + 0..3 (first read) and 16 + 4 g + 0..3 (second read: + 16 x 128 B);
  // q: pixel inside the group of four; p: column quad (4 channels = 8 B) of the 16-channel tile
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int ct0 = 2 * (wave >> 2), cit = wave & 3;  // this wave's two co tiles (16 each) and its ci tile
  const int pg = 4 * g + q;                          // pixel of the first read inside the row
  // G image: pixel row index 32 r + pg (+ 16): swizzle ((pix >> 1) & 3) is a lane constant
  const int ga0 = PATCH_B + pg * 128 + (((2 * ct0 + (p >> 1)) ^ (((pg >> 1) & 3) << 1)) << 4) + 8 * (p & 1);
  const int ga1 = ga0 ^ 32;  // co tile ct0 + 1: the next 32-B window (ct0 is even)
  // patch image: pixel pr * PW + kx + pg (+ 16), pr = r + ky;  (pp & 7) = ((pr * PW) & 7 + kx + pg) & 7, and
  // (pr * PW) & 7 takes NV values, selected by pr % NV
  int xa[KS][NV];
#pragma unroll
  for (int kx = 0; kx < KS; ++kx)
#pragma unroll
    for (int v = 0; v < NV; ++v)
      xa[kx][v] = (kx + pg) * 128 + (((2 * cit + (p >> 1)) ^ ((((((v * PW) & 7) + kx + pg) >> 1) & 3) << 1)) << 4) + 8 * (p & 1);

  f32x4 acc[2][NTAP];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // XS: `cur` holds the running sample's partial, `acc` the scaled total
  f32x4 cur[XS ? 2 : 1][XS ? NTAP : 1];
  if constexpr (XS) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < NTAP; ++t) cur[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // fragments of one unit: the two G fragments of its tile row and its TPU shifted X fragments (inline-asm reads)
  struct UnitFrags { bf16x8 a0, a1, x[TPU]; };
  auto join = [](s16x4 lo, s16x4 hi) {
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  };
  const unsigned lbase = lds_addr(smem);
  auto read_unit = [&](UnitFrags& f, const unsigned tb, auto uc) {  // uc: std::integral_constant unit index
    constexpr int U = decltype(uc)::value, r = U / UPR, hf = U % UPR;
    f.a0 = join(ds_tr<r * 4096>(tb + ga0), ds_tr<r * 4096 + 2048>(tb + ga0));
    f.a1 = join(ds_tr<r * 4096>(tb + ga1), ds_tr<r * 4096 + 2048>(tb + ga1));
    static_for<TPU>([&](auto jc) {
      constexpr int jj = decltype(jc)::value, pr = r + hf * KYU + jj / KS, kx = jj % KS;
      const unsigned a = tb + xa[kx][pr % NV];
      f.x[jj] = join(ds_tr<pr * PW * 128>(a), ds_tr<pr * PW * 128 + 2048>(a));
    });
  };
  auto multiply = [&](const UnitFrags& f, auto uc) {
    constexpr int hf = decltype(uc)::value % UPR;
#pragma unroll
    for (int jj = 0; jj < TPU; ++jj) {
      const int t = hf * TPU + jj;  // tap = (hf * KYU + jj / KS) * KS + jj % KS
      if constexpr (XS) {
        cur[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a0, f.x[jj], cur[0][t], 0, 0, 0);
        cur[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, f.x[jj], cur[1][t], 0, 0, 0);
      } else {
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a0, f.x[jj], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, f.x[jj], acc[1][t], 0, 0, 0);
      }
    }
  };
  auto fold_sample = [&](int b) {  // acc += s[b, this lane's input channel] * cur;  cur = 0
    if constexpr (XS) {
      const float sv = d.in_scale[(size_t)b * Ci + ci0 + 16 * cit + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            acc[i][t][qq] += sv * cur[i][t][qq];
            cur[i][t][qq] = 0.f;
          }
    }
  };

  if (t_begin < t_end) issue_tile(t_begin, 0);
  int cur_b = t_begin / tpi;
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    if constexpr (XS) {
      const int tb_ = t / tpi;  // (uniform) the tiles of a slice are consecutive: a sample's tiles are contiguous
      if (tb_ != cur_b) { fold_sample(cur_b); cur_b = tb_; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's fills of tile t have landed ...
    __syncthreads();                                   // ... everybody's have, and everybody is done with tile t - 1
    if (t + 1 < t_end) issue_tile(t + 1, buf ^ 1);     // into the buffer tile t - 1 was read from
    const unsigned tb = lbase + (unsigned)(buf * TILE_B);
    // units software-pipelined by one: the reads of unit u + 1 are in flight under the MFMAs of unit u
    UnitFrags fa, fb;
    read_unit(fa, tb, std::integral_constant<int, 0>{});
    lds_reads_done();
    static_for<NU / 2>([&](auto hc) {
      constexpr int u = 2 * decltype(hc)::value;
      read_unit(fb, tb, std::integral_constant<int, u + 1>{});
      multiply(fa, std::integral_constant<int, u>{});
      lds_reads_done();
      if constexpr (u + 2 < NU) read_unit(fa, tb, std::integral_constant<int, u + 2>{});
      multiply(fb, std::integral_constant<int, u + 1>{});
      if constexpr (u + 2 < NU) lds_reads_done();
    });
  }
  if (t_begin < t_end) fold_sample(cur_b);

  // ---- epilogue: C/D of 16x16x32: column = lane & 15 (ci), row = 4 (lane >> 4) + register (co) ---------------------
  float* __restrict__ slab = d.slabs + (size_t)slice * Co * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        slab[(size_t)(co0 + 16 * (ct0 + i) + 4 * g + rr) * K + t * Ci + ci0 + 16 * cit + (lane & 15)] = acc[i][t][rr];
}

// the layers the halo-tile weight gradient takes; *slices: its slicing of the pixel tiles
inline bool wgrad_halo_ok(const o2m_wgrad_desc& d, long* slices, long* tiles_per_slice) {
  static const int on = [] { const char* e = getenv("O2M_WGRAD_HALO"); return e ? atoi(e) : 1; }();
  if (!on || d.dtype != O2M_BF16 || d.KH != d.KW || (d.KH != 3 && d.KH != 4) || d.pad != 1 || d.pad_mode != O2M_PAD_ZERO ||
      d.stride > 1 || d.gy_scale || d.splits > 0 || d.Co % 64 || d.Ci % 64 || (d.in_scale && d.KH != 3))
    return false;
  const int Ho = d.H + 2 - d.KH + 1, Wo = d.W + 2 - d.KW + 1, th = d.KH == 3 ? 8 : 4;
  if (Ho < 1 || Wo < 1) return false;
  const long pairs = (long)(d.Co / 64) * (d.Ci / 64), total = (long)d.B * ((Ho + th - 1) / th) * ((Wo + 31) / 32);
  if (pairs > 64) return false;
  if (total < 64) return false;  // (a few tiles: the register-staged tiles' splitting serves those)
  long s = 256 / pairs;  // one resident block per CU
  if (s < 1) s = 1;
  long per = (total + s - 1) / s;
  if (per < 8) per = 8;  // at least 8 tiles per block, to amortise its 64 x KS^2 x 64 fp32 partial
  *tiles_per_slice = per;
  *slices = (total + per - 1) / per;
  return true;
}

// the layers the phase-pipelined weight-gradient kernel takes; *splits / *rows: its slicing of the image rows
inline bool wgrad_p8_ok(const o2m_wgrad_desc& d, long* splits, long* rows) {
  if (d.kernel_hint != O2M_WGRAD_HINT_P8 || d.dtype != O2M_BF16 || d.Co != 256 || d.Ci != 256 || d.KH != 3 || d.KW != 3 || d.pad != 1 || d.stride > 1 ||
      d.W != 64 || d.in_scale || d.gy_scale || d.splits > 0)
    return false;
  const long r_total = (long)d.B * d.H;
  if (r_total < 28 * 8) return false;  // >= 8 K-tiles per slice
  const long per = (r_total + 27) / 28;  // 9 taps x 28 slices = 252 blocks: one per CU
  *rows = per;
  *splits = (r_total + per - 1) / per;
  return true;
}

// slab_floats != nullptr: only report the workspace the launch would need
template <typename T, int BCO, int BKO, int WAVES_CO, int WAVES_K>
int launch_cfg(const o2m_wgrad_desc& d, hipStream_t s, size_t* slab_floats = nullptr) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int lds = 2 * (F32 ? 2 : 1) * BMR * ((BCO * 2 + 64) + (BKO * 2 + 64));
  const int st = d.stride > 0 ? d.stride : 1;
  const int Ho = (d.H + 2 * d.pad - d.KH) / st + 1, Wo = (d.W + 2 * d.pad - d.KW) / st + 1;
  const long M = (long)d.B * Ho * Wo;
  const int K = d.KH * d.KW * d.Ci;
  const int tiles_co = (d.Co + BCO - 1) / BCO, tiles_k = (K + BKO - 1) / BKO;
  long splits = d.splits;
  if (splits <= 0) {
    // Whole waves of co-resident blocks (tools/sweep_wgrad.py: 768 = 3 blocks x 256 CUs is the
    // optimum for the 128-wide tile, 1024 = 1 1/3 waves consistently ~25% slower); more waves only
    // when one block would reduce more than ~12288 pixels.  Every split adds a Co*K slab of atomics.
    const long tiles = (long)tiles_co * tiles_k;
    // resident blocks per CU: 128 accumulator registers -> 2; else LDS- (3) or VGPR-limited (4)
    const long per_cu = BCO * BKO >= 256 * 128 ? 2 : (lds > 36 * 1024 ? 3 : 4);
    const long wave = 256 * per_cu;
    const long rows_one_wave = (M * tiles + wave - 1) / wave;
    const long waves = (rows_one_wave + 12287) / 12288;
    splits = wave * waves / tiles;  // round DOWN: two blocks over the wave cost a whole extra round
    const long max_splits = (M + 8 * BMR - 1) / (8 * BMR);  // >= 8 stages per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  long rows = (M + splits - 1) / splits;
  rows = (rows + BMR - 1) / BMR * BMR;
  splits = (M + rows - 1) / rows;
  const long blocks = splits * tiles_co * tiles_k;
  if (blocks > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  if (slab_floats) {
    *slab_floats = (size_t)splits * d.Co * K;
    return 0;
  }
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, s, d, tiles_co, tiles_k, (int)rows);
  };
  const bool xs = d.in_scale != nullptr, gs = d.gy_scale != nullptr;
  auto pick = [&](auto al) {
    constexpr bool AL = decltype(al)::value;
    if (xs && gs) go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, true, true, AL>);
    else if (xs) go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, true, false, AL>);
    else if (gs) go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, false, true, AL>);
    else go(conv_wgrad_kernel<T, BCO, BKO, WAVES_CO, WAVES_K, false, false, AL>);
  };
  {
    LaunchScope timed(s, 2.0 * M * d.Co * K, "conv_wgrad<%s,co%dxk%d>", sizeof(T) == 2 ? "bf16" : "f32x3", BCO, BKO);
    if (Wo % BMR == 0) pick(std::true_type{});
    else pick(std::false_type{});
  }
  O2M_LAUNCH_CHECK();
  if (d.slabs) {
    const long n4 = (long)d.Co * K / 4;  // Ci % 8 == 0, so K % 8 == 0
    LaunchScope timed(s, 0.0, "%s", "wgrad_reduce");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d.dw, d.slabs,
                       (int)splits, n4);
    O2M_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
int launch_dtype_r2(const o2m_wgrad_desc& d, hipStream_t s, size_t* slab_floats = nullptr);

int launch_wgrad_p8(const o2m_wgrad_desc& d, hipStream_t s, long splits, long rows, size_t* slab_floats) {
  const int K = 9 * d.Ci;
  if (slab_floats) {
    *slab_floats = (size_t)splits * d.Co * K;
    return 0;
  }
  constexpr int lds = 128 * (256 + 4) * 4;  // >= the two 64 KB K-tile buffers
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_p8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * d.B * d.H * d.W * d.Co * K, "%s", "conv_wgrad_p8<bf16,256x256>");
    hipLaunchKernelGGL(conv_wgrad_p8_kernel, dim3((unsigned)(9 * splits)), dim3(512), lds, s, d, (int)rows);
  }
  O2M_LAUNCH_CHECK();
  const long n4 = (long)d.Co * K / 4;
  LaunchScope timed(s, 0.0, "%s", "wgrad_reduce");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d.dw, d.slabs, (int)splits, n4);
  O2M_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_halo(const o2m_wgrad_desc& d, hipStream_t s, long slices, long per, size_t* slab_floats) {
  const int K = d.KH * d.KW * d.Ci;
  if (slab_floats) {
    *slab_floats = (size_t)slices * d.Co * K;
    return 0;
  }
  const long pairs = (long)(d.Co / 64) * (d.Ci / 64);
  const int Ho = d.H + 2 - d.KH + 1, Wo = d.W + 2 - d.KW + 1;
  auto go = [&](auto kern, int lds, const char* name) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    LaunchScope timed(s, 2.0 * d.B * Ho * Wo * d.Co * K, "%s", name);
    hipLaunchKernelGGL(kern, dim3((unsigned)(slices * pairs)), dim3(512), lds, s, d, (int)per);
  };
  // two tile buffers: patch fills (8 pixels each) + G tile
  constexpr int lds3 = 2 * (((10 * 34 + 7) / 8) * 1024 + 8 * 4 * 1024), lds4 = 2 * (((7 * 36 + 7) / 8) * 1024 + 4 * 4 * 1024);
  if (d.KH == 3) {
    if (d.in_scale) go(conv_wgrad_halo_kernel<3, 8, 34, true>, lds3, "conv_wgrad_halo<bf16,64x9x64>");
    else go(conv_wgrad_halo_kernel<3, 8, 34, false>, lds3, "conv_wgrad_halo<bf16,64x9x64>");
  } else {
    go(conv_wgrad_halo_kernel<4, 4, 36, false>, lds4, "conv_wgrad_halo<bf16,64x16x64>");
  }
  O2M_LAUNCH_CHECK();
  const long n4 = (long)d.Co * K / 4;
  LaunchScope timed(s, 0.0, "%s", "wgrad_reduce");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d.dw, d.slabs, (int)slices, n4);
  O2M_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int launch_dtype(const o2m_wgrad_desc& d, hipStream_t s, size_t* slab_floats = nullptr) {
  if constexpr (sizeof(T) == 2) {
    long splits = 0, rows = 0;
    // (slab mode only: the atomic form, O2M_WGRAD_ATOMICS=1, keeps the round-2 kernels; a size query cannot know
    // the mode, and both kernels fit the larger of the two workspaces)
    if (wgrad_p8_ok(d, &splits, &rows)) {
      if (slab_floats) {
        size_t a = 0, b = 0;
        (void)launch_wgrad_p8(d, s, splits, rows, &a);
        (void)launch_dtype_r2<T>(d, s, &b);
        *slab_floats = a > b ? a : b;
        return 0;
      }
      if (d.slabs) return launch_wgrad_p8(d, s, splits, rows, nullptr);
    }
    long slices = 0, per = 0;
    if (wgrad_halo_ok(d, &slices, &per)) {  // (slab mode only, like the p8 form; a size query reports the larger workspace)
      if (slab_floats) {
        size_t a = 0, b = 0;
        (void)launch_wgrad_halo(d, s, slices, per, &a);
        (void)launch_dtype_r2<T>(d, s, &b);
        *slab_floats = a > b ? a : b;
        return 0;
      }
      if (d.slabs) return launch_wgrad_halo(d, s, slices, per, nullptr);
    }
  }
  return launch_dtype_r2<T>(d, s, slab_floats);
}

template <typename T>
int launch_dtype_r2(const o2m_wgrad_desc& d, hipStream_t s, size_t* slab_floats) {
  // O2M_WGRAD_TILES=small keeps the 128-wide tiles (A/B measurements)
  static const bool small = [] { const char* e = getenv("O2M_WGRAD_TILES"); return e && e[0] == 's'; }();
  const int K = d.KH * d.KW * d.Ci;
  // fp32 (bf16x3 split) mode: the 256 x 128 / 128 x 256 register tiles spill 63-199 VGPRs (172-544 B of scratch per lane,
  // -Rpass-analysis=kernel-resource-usage); the 128 x 128 tile does not.  O2M_F32_BIG_TILES=1 keeps the old selection (A/B).
  static const bool f32_big = [] { const char* e = getenv("O2M_F32_BIG_TILES"); return e && e[0] == '1'; }();
  if (!small && (sizeof(T) == 2 || f32_big)) {
    // 128x64 / 64x128 wave tiles (1.5x fewer LDS fragment bytes per MFMA) where they measured
    // faster (tools/sweep_wgrad.py): long reductions for wide layers, K a multiple of 256
    const int st = d.stride > 0 ? d.stride : 1;
    const long M = (long)d.B * ((d.H + 2 * d.pad - d.KH) / st + 1) *
                   ((d.W + 2 * d.pad - d.KW) / st + 1);
    if (d.Co > 256 || (d.Co > 128 && M >= 100000)) return launch_cfg<T, 256, 128, 2, 2>(d, s, slab_floats);
    if (d.Co > 64 && d.Co <= 128 && K % 256 == 0 && sizeof(T) == 2)  // fp32 split: would spill
      return launch_cfg<T, 128, 256, 2, 2>(d, s, slab_floats);
  }
  if (d.Co > 64) return launch_cfg<T, 128, 128, 2, 2>(d, s, slab_floats);
  if (d.Co > 32) return launch_cfg<T, 64, 128, 2, 2>(d, s, slab_floats);
  return launch_cfg<T, 32, 128, 1, 4>(d, s, slab_floats);
}

}  // namespace

static int wgrad_run(const o2m_wgrad_desc* d, void* stream, size_t* slab_floats);

extern "C" size_t o2m_conv2d_wgrad_slab_floats(const o2m_wgrad_desc* d) {
  size_t n = 0;
  return wgrad_run(d, nullptr, &n) == 0 ? n : 0;
}

extern "C" int o2m_conv2d_wgrad(const o2m_wgrad_desc* d, void* stream) { return wgrad_run(d, stream, nullptr); }

static int wgrad_run(const o2m_wgrad_desc* d, void* stream, size_t* slab_floats) {
  if (!d || ((!d->x || !d->gy || !d->dw) && !slab_floats)) return O2M_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ci <= 0 || d->Co <= 0) return O2M_ERR_BAD_ARG;
  if ((d->Ci & 7) || (d->Co & 7) || d->KH <= 0 || d->KW <= 0 || d->pad < 0) return O2M_ERR_BAD_ARG;
  if (d->H + 2 * d->pad < d->KH || d->W + 2 * d->pad < d->KW) return O2M_ERR_BAD_ARG;
  if (d->pad_mode == O2M_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W)) return O2M_ERR_BAD_ARG;
  if (d->pad_mode != O2M_PAD_ZERO && d->pad_mode != O2M_PAD_REFLECT) return O2M_ERR_BAD_ARG;
  const long esz = d->dtype == O2M_F32 ? 4 : 2;
  if (d->stride < 0 || d->stride > 8) return O2M_ERR_BAD_ARG;
  const int st = d->stride > 0 ? d->stride : 1;
  const long howo = (long)((d->H + 2 * d->pad - d->KH) / st + 1) * ((d->W + 2 * d->pad - d->KW) / st + 1);
  // Buffer descriptors address < 2 GiB per tensor: larger operands (fp32 parity mode at 512 x 512) are reduced as
  // several launches over slices of the batch, each adding its slab sum into dw (fixed order).
  const long x_sample = (long)d->H * d->W * (long)d->Ci * esz, g_sample = howo * (long)d->Co * esz;
  if (x_sample > 0x7fffffffL || g_sample > 0x7fffffffL) return O2M_ERR_UNSUPPORTED;
  if ((long)d->B * x_sample > 0x7fffffffL || (long)d->B * g_sample > 0x7fffffffL) {
    const long big = x_sample > g_sample ? x_sample : g_sample;
    const int per = (int)(0x7fffffffL / big);
    size_t most = 0;
    for (int b0 = 0; b0 < d->B; b0 += per) {
      o2m_wgrad_desc c = *d;
      c.B = d->B - b0 < per ? d->B - b0 : per;
      c.x = static_cast<const char*>(d->x) + (size_t)b0 * x_sample;
      c.gy = static_cast<const char*>(d->gy) + (size_t)b0 * g_sample;
      if (d->in_scale) c.in_scale = d->in_scale + (size_t)b0 * d->Ci;
      if (d->gy_scale) c.gy_scale = d->gy_scale + (size_t)b0 * d->Co;
      size_t n = 0;
      const int rc = wgrad_run(&c, stream, slab_floats ? &n : nullptr);
      if (rc) return rc;
      if (n > most) most = n;
    }
    if (slab_floats) *slab_floats = most;
    return 0;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == O2M_BF16) return launch_dtype<unsigned short>(*d, s, slab_floats);
  if (d->dtype == O2M_F32) return launch_dtype<float>(*d, s, slab_floats);
  return O2M_ERR_BAD_ARG;
}
