// Implicit-GEMM convolution forward / data-gradient on MFMA for gfx950.
//
//   GEMM view:  M = B*Ho*Wo (output pixels), N = Co, K = KH*KW*Ci (ci fastest).
//   A[m][k] is gathered on the fly from the NHWC input (im2col never materialised);
//   zero or reflect padding is index arithmetic in the loader; the per-(b,ci) style scale
//   of the modulated conv (fallback path) is applied while the tile sits in registers.
//   B[n][k] is the filter, stored [Co][KH][KW][Ci] so both operands are K-contiguous and
//   share one fragment path.
//
//   Tile BM x BN x 64, 4 or 8 waves; each wave owns a grid of 32x32 MFMA tiles
//   (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Global -> registers -> LDS staging (16 B per
//   lane, issue-early / write-late, two LDS stages), XOR-swizzled 16-B slots: the
//   ds_read_b128 fragment reads are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
//
//   The reduction is walked tap-outer / channel-inner: the gather address of a pixel row
//   (bounds test or mirror) is computed once per filter tap and then advanced by 64
//   channels with ONE add per stage -- the first version recomputed it every stage and was
//   VALU-issue-bound at 16 VALU per MFMA (profiles/r01_b_pmc_igemm_before.txt).
//
//   dtype O2M_F32 ("parity mode"): operands are split hi + lo into two bf16 tiles while
//   staging and every product runs as hi*hi + hi*lo + lo*hi on the bf16 MFMA: ~2^-16
//   relative error per product at 3/16 the cost of the fp32 MFMA.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_direct.h"

namespace {

constexpr int BK = 64;  // reduction elements per stage (one 128-B LDS row of bf16)
constexpr long kFillBlocksDefault = 256;  // one block per CU
long g_fill_blocks = kFillBlocksDefault;   // o2m_debug_fill_blocks (test hook): routes small parity cases to the big tiles
#define kFillBlocks g_fill_blocks

// Byte offset of 16-B slot (row, chunk) in a [rows][64] bf16 tile.  Two 128-B rows share a
// 256-B bank row; XOR with (row>>1)&15 spreads each ds_read_b128 lane group (same chunk,
// 16 different rows) over all 16 slots of the bank row.
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return ((row >> 1) << 8) + (((((row & 1) << 3) | chunk) ^ ((row >> 1) & 15)) << 4);
}

template <typename T> struct Stg;  // staged registers for 8 consecutive reduction elements
template <> struct Stg<unsigned short> { u32x4 v; };
template <> struct Stg<float> { f32x4 a, b; };

// Raw buffer descriptor over a whole tensor: a load whose byte offset is >= the size
// returns zeros, which is how padding taps / rows outside the problem are zero-filled
// without branches (invalid rows carry the offset OOB_OFF).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned OOB_OFF = 0x80000000u;  // tensors are limited to < 2 GiB (checked on the host)
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void stg_load(Stg<unsigned short>& s, rsrc_t r, unsigned byte_off) {
  s.v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void stg_load(Stg<float>& s, rsrc_t r, unsigned byte_off) {
  s.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
  s.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off + 16, 0, 0));
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

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
constexpr int lds_bytes() {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int lds_main = (F32 ? 2 : 1) * (F32 ? 1 : 2) * (BM + BN) * BK * 2;
  constexpr int lds_epi = (BM / WAVES_M) * (BN + 4) * 4;
  return lds_main > lds_epi ? lds_main : lds_epi;
}

// Barrier that orders LDS traffic only.  __syncthreads() also waits for this wave's outstanding GLOBAL stores
// (vmcnt(0)); between the epilogue passes that would park every wave until the previous pass's output rows
// have reached memory, although the next pass only reuses the LDS staging tile.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// InstanceNorm partial sums from the epilogue (o2m_conv_desc.stats).  Every thread of the read-out
// loop keeps ONE 8-channel vector (NT % (BN/8) == 0) and walks rows, so it sums its rows in
// registers; the NT / VPR threads that share a vector are then added through LDS (the staged tile
// is dead by then) and one thread per (channel, statistic) writes the partial: no atomics, a fixed
// summation order.
template <int NT, int VPR>
__device__ __forceinline__ void stats_block_reduce(const float (&st)[16], float* red, float* __restrict__ stats,
                                                   long part, int n0, int Co, int tid) {
  lds_barrier();  // every thread has finished reading the staged tile
#pragma unroll
  for (int q = 0; q < 16; ++q) red[q * NT + tid] = st[q];
  lds_barrier();
  if (tid < VPR * 16) {
    const int c8 = tid % VPR, q = tid / VPR;
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < NT / VPR; ++k) s += red[q * NT + c8 + VPR * k];
    const int n = n0 + c8 * 8 + (q & 7);
    if (n < Co) stats[((size_t)part * Co + n) * 2 + (q >> 3)] = s;
  }
}

// Division of an output-row index by a block-uniform divisor (pixels per sample, pixels per row) without
// the ~35-instruction integer-division expansion per row: float reciprocal + one correction step, exact
// for 0 <= m < 2^22 (the quotient estimate is then off by at most one); larger problems divide normally.
struct RowDiv {
  int d;
  float r;
  bool fast;
  __device__ RowDiv(int d_, int limit) : d(d_), r(1.0f / (float)d_), fast(limit < (1 << 22)) {}
  __device__ __forceinline__ int div(int m) const {
    if (!fast) return m / d;
    int q = (int)((float)m * r);
    const int rem = m - q * d;
    q += (rem >= d) - (rem < 0);
    return q;
  }
};

// o2m_conv_desc.fold_pad = f > 0: output pixel m of the PADDED domain [B][Ho][Wo] lands on its mirror image in the
// cropped map [B][Ho - 2f][Wo - 2f] (the adjoint of ReflectionPad2d(f)); see the header.
struct FoldMap {
  int f, Hc, Wc, Wo, HoWo;
  RowDiv by_howo, by_wo;
  __device__ FoldMap(int f_, int Ho, int Wo_, int M) : f(f_), Hc(Ho - 2 * f_), Wc(Wo_ - 2 * f_), Wo(Wo_), HoWo(Ho * Wo_),
                                                       by_howo(Ho * Wo_, M), by_wo(Wo_, M) {}
  // pixel index in the cropped map; interior: m is not a pad pixel; atomic: the target takes several contributions
  __device__ __forceinline__ long pixel(int m, bool& interior, bool& atomic) const {
    const int b = by_howo.div(m), rem = m - b * HoWo;
    const int oy = by_wo.div(rem), ox = rem - oy * Wo;
    const int qy = oy - f, qx = ox - f;
    interior = (unsigned)qy < (unsigned)Hc && (unsigned)qx < (unsigned)Wc;
    const int py = qy < 0 ? -qy : (qy >= Hc ? 2 * Hc - 2 - qy : qy);
    const int px = qx < 0 ? -qx : (qx >= Wc ? 2 * Wc - 2 - qx : qx);
    const bool target = (py >= 1 && py <= f) || (py >= Hc - 1 - f && py <= Hc - 2) || (px >= 1 && px <= f) ||
                        (px >= Wc - 1 - f && px <= Wc - 2);
    atomic = !interior || target;
    return ((long)b * Hc + py) * Wc + px;
  }
};

// WIDE: Ci % 64 == 0, so one 64-element stage lies inside ONE filter tap (tap-outer walk).
// !WIDE: small Ci (image stems, Ci = 8..32): every 16-B chunk decodes its own tap.
// FOLD: o2m_conv_desc.fold_pad handled by the epilogue.  A separate instantiation (only the 128 x 128 tile, which takes
// the p8 kernel's tail rows and the layers p8 does not cover): compiled into every tile it cost 14-19 VGPRs -- the
// 256 x 64 / 128 x 64 / 256 x 32 tiles went from 120 to 134-139, i.e. from two blocks per CU to one (2.9 -> 4.1 ms per step).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool IN_SCALE, bool WIDE, bool FOLD = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_igemm_kernel(const o2m_conv_desc d, const int m_begin,
                                                                                const int m_end) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int NPLANE = F32 ? 2 : 1;  // hi (+ lo)
  constexpr int NSTAGE = F32 ? 1 : 2;
  constexpr int RSTEP = NT / 8;  // rows covered by one pass of the block's threads
  constexpr int RA = BM / RSTEP, RB = (BN + RSTEP - 1) / RSTEP;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = NPLANE * (A_BYTES + B_BYTES);
  static_assert(BM % RSTEP == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
  // LDS-DMA staging (global_load_lds_dwordx4): the tile goes L2 -> LDS without staging VGPRs
  // and without ds_write.  One wave-instruction fills 64 consecutive 16-B slots (8 rows x 8
  // chunks); the XOR swizzle is applied on the SOURCE side: lane l of fill q owns linear slot
  // 64q + l and fetches the (row, chunk) that tile_off maps there.  Needs untransformed data:
  // plain bf16, no in_scale, stage inside one tap.
  constexpr bool DMA = !F32 && !IN_SCALE && WIDE && (BN % RSTEP == 0);
  // Every wave issues its share of the fills.  (Tried and measured slower on the 256->256
  // layer, 877 TF/s as is: fills dealt out between the k-steps' MFMAs 803; fills issued by
  // half the waves while their SIMD partners multiply 837; s_setprio around the MFMAs 806.
  // s_memtime stamps of a stage: ~1020 cycles issuing 8 fills, ~1520 multiplying, ~1200 at the
  // barrier behind the SIMD partner.)
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int LW = NW;                       // loader waves
  constexpr int FA = DMA ? RA * NW / LW : RA;  // A fills per loader wave
  constexpr int FB = DMA ? RB * NW / LW : RB;  // B fills per loader wave

  extern __shared__ __attribute__((aligned(16))) char smem[];

  constexpr int ES = sizeof(T);  // element size in bytes
  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, KH = d.KH, KW = d.KW, pad = d.pad;
  const int S = d.stride > 1 ? d.stride : 1;
  const int Ho = (H + 2 * pad - KH) / S + 1, Wo = (W + 2 * pad - KW) / S + 1;
  const int HoWo = Ho * Wo;
  const int M = m_end;  // rows [m_begin, m_end) of the B * Ho * Wo output pixels (a launch may cover a slice)
  const int K = KH * KW * Ci;
  const int nk = (K + BK - 1) / BK;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;

  const int tiles_n = (Co + BN - 1) / BN;
  const int tile = xcd_tile_order(blockIdx.x, gridDim.x);
  const int m0 = m_begin + (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  // sample of the tile's first pixel; uniform over the tile when it does not straddle samples
  const int b_first = m0 / HoWo;
  const bool b_uniform = (min(m0 + BM, M) - 1) / HoWo == b_first;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * ES));
  // per-sample filters (host guarantees the tile does not straddle samples)
  const rsrc_t wr = make_rsrc(static_cast<const char*>(d.w) + (size_t)b_first * d.w_batch_stride * ES,
                              (unsigned)((size_t)Co * K * ES));

  const int tid = threadIdx.x;
  const int cc = tid & 7;   // 16-B chunk (8 reduction elements) inside the 64-wide stage
  const int r0 = tid >> 3;  // 0 .. RSTEP-1

  // ---- per-thread row bookkeeping for the A gather ---------------------------------
  // pix[j] = linear index (b*H + oy*S)*W + ox*S of the input pixel under output pixel (b,oy,ox)
  // (-1: row outside M); ryx[j] = oy*S << 16 | ox*S.
  // DMA mode: fill q covers tile rows 8q .. 8q+7; lane l owns linear slot 64q + l
  const int dwave = tid >> 6, dlane = tid & 63;
  auto dma_row = [&](int q) {  // tile row of this lane's slot in fill q
    const int pr = 4 * q + (dlane >> 4);
    return 2 * pr + ((((dlane & 15) ^ (pr & 15))) >> 3);
  };
  auto dma_chk = [&](int q) {
    const int pr = 4 * q + (dlane >> 4);
    return ((dlane & 15) ^ (pr & 15)) & 7;
  };
  const bool loader = !DMA || dwave < LW;  // wave-uniform
  const RowDiv by_howo(HoWo, M), by_wo(Wo, M);
  int pix[FA], ryx[FA];
#pragma unroll
  for (int j = 0; j < FA; ++j) {
    const int m = m0 + (DMA ? dma_row((dwave % LW) * FA + j) : r0 + RSTEP * j);
    if (m < M) {
      const int b = by_howo.div(m), rem = m - b * HoWo;
      const int q = by_wo.div(rem);
      const int oy = q * S, ox = (rem - q * Wo) * S;
      pix[j] = (b * H + oy) * W + ox;
      ryx[j] = (oy << 16) | ox;
    } else {
      pix[j] = -1;
      ryx[j] = 0;
    }
  }
  // LDS write offset of this thread's first row; row r0 + RSTEP*j adds j*RSTEP*128 bytes
  // (RSTEP is a multiple of 32, so the XOR term of tile_off is unchanged)
  const int woff0 = tile_off(r0, cc);
  // filter rows of this thread: element offset of row 0 and a validity bit per row
  const unsigned wbase0 = (unsigned)((n0 + r0) * K + cc * 8) * ES;  // bytes
  unsigned wvalid = 0;
#pragma unroll
  for (int j = 0; j < RB; ++j)
    if (r0 + RSTEP * j < BN && n0 + r0 + RSTEP * j < Co) wvalid |= 1u << j;

  Stg<T> sa[RA], sb[RB];
  f32x4 sca[IN_SCALE ? RA : 1][2];

  // tap-outer walk state (WIDE): element offset of (row, tap), or -1 when the tap falls in
  // the zero padding / the row is outside M
  unsigned aoff[FA];  // byte offsets
  int cur_tap = -1;   // uniform
  // Interior fills (DMA): every row of the fill has its whole KH x KW window inside the image.  Their
  // lane offset points at the window's first tap and never changes; the tap moves only the fill's
  // SGPR offset (tap_soff), so the walk costs such a fill no vector instructions at all.  Border fills
  // (and rows past M) keep the per-tap offset with its padding test.
  unsigned inner = 0;  // bit j: fill j is interior (wave-uniform)
  unsigned tap_soff = 0;
  if constexpr (DMA) {
#pragma unroll
    for (int j = 0; j < FA; ++j) {
      const int oy = ryx[j] >> 16, ox = ryx[j] & 0xffff;
      const bool in = pix[j] >= 0 && oy >= pad && oy - pad + KH <= H && ox >= pad && ox - pad + KW <= W;
      if (__all(in)) {
        inner |= 1u << j;
        aoff[j] = (unsigned)((pix[j] - pad * W - pad) * Ci + dma_chk((dwave % LW) * FA + j) * 8) * ES;
      }
    }
    inner = __builtin_amdgcn_readfirstlane(inner);  // tell the compiler: an SGPR (no waterfall around the fills)
  }

  auto set_tap = [&](int tap) {
    const int dy = tap / KW - pad, dx = tap - (tap / KW) * KW - pad;  // scalar
    if constexpr (DMA) tap_soff = (unsigned)(((dy + pad) * W + dx + pad) * Ci) * ES;
#pragma unroll
    for (int j = 0; j < FA; ++j) {
      if (DMA && (inner >> j & 1)) continue;
      const int oy = ryx[j] >> 16, ox = ryx[j] & 0xffff;
      int iy = oy + dy, ix = ox + dx;
      bool ok = pix[j] >= 0;
      if (reflect) {
        iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
      } else {
        ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      }
      const int chk = DMA ? dma_chk((dwave % LW) * FA + j) : cc;
      aoff[j] = ok ? (unsigned)((pix[j] + (iy - oy) * W + (ix - ox)) * Ci + chk * 8) * ES : OOB_OFF;
    }
  };

  // DMA mode: byte offsets of this lane's filter rows (or the out-of-range mark)
  unsigned dwoff[FB];
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    const int q = (dwave % LW) * FB + j;
    const int n = n0 + dma_row(q);
    dwoff[j] = (DMA && n < Co) ? (unsigned)(n * K + dma_chk(q) * 8) * 2u : OOB_OFF;
  }
  typedef __attribute__((address_space(3))) void lds_void;

  auto dma_tiles = [&](int kt, int stage) {
    if constexpr (LW < NW) {
      if (!loader) return;
    }
    const int k0 = kt * BK;
    const int tap = k0 / Ci;
    const int cbase = k0 - tap * Ci;
    if (tap != cur_tap) { set_tap(tap); cur_tap = tap; }
    char* a_dst = smem + stage * STAGE_BYTES + dwave * FA * 1024;
    char* b_dst = smem + stage * STAGE_BYTES + A_BYTES + dwave * FB * 1024;
    // buffer_load_dwordx4 ... lds: descriptor + this lane's byte offset (computed once per tap) +
    // the stage's channel offset in an SGPR.  No 64-bit pointer arithmetic and no select per fill,
    // and an out-of-range offset (padding tap, row outside the problem) zero-fills its LDS slot
    // (checked on hardware: tools/probe_buffer_lds.py), so the zero page is not needed here.
#pragma unroll
    for (int j = 0; j < FA; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          xr, (lds_void*)(a_dst + j * 1024), 16, (int)aoff[j],
          __builtin_amdgcn_readfirstlane(cbase * 2 + ((inner >> j & 1) ? tap_soff : 0u)), 0, 0);
#pragma unroll
    for (int j = 0; j < FB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(b_dst + j * 1024), 16, (int)dwoff[j], k0 * 2, 0, 0);
  };

  auto load_tiles = [&](int kt) {
    if constexpr (WIDE) {
      // stage order: tap OUTER, channel chunk INNER (consecutive stages stream each pixel's
      // Ci*2 bytes).  The opposite order (chunk outer: smaller L2 working set per XCD) measured
      // SLOWER, 742 vs 846 TF/s on the 256->256 layer: one 128-B line per pixel per stage.
      const int k0 = kt * BK;           // scalar
      const int tap = k0 / Ci;          // scalar
      const int cbase = k0 - tap * Ci;  // scalar
      if (tap != cur_tap) { set_tap(tap); cur_tap = tap; }
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        stg_load(sa[j], xr, aoff[j] + (unsigned)cbase * ES);
        if constexpr (IN_SCALE) {
          if (aoff[j] != OOB_OFF) {
            const float* sp = d.in_scale + (size_t)((m0 + r0 + RSTEP * j) / HoWo) * Ci + cbase + cc * 8;
            sca[j][0] = *reinterpret_cast<const f32x4*>(sp);
            sca[j][1] = *reinterpret_cast<const f32x4*>(sp + 4);
          } else {
            sca[j][0] = sca[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < RB; ++j)
        stg_load(sb[j], wr, (wvalid >> j & 1) ? wbase0 + (unsigned)(j * RSTEP * K + k0) * ES : OOB_OFF);
    } else {
      const int k = kt * BK + cc * 8;
      const bool kv = k < K;
      const int tap = k / Ci, ci0 = k - tap * Ci;
      const int dy = tap / KW - pad, dx = tap - (tap / KW) * KW - pad;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int oy = ryx[j] >> 16, ox = ryx[j] & 0xffff;
        int iy = oy + dy, ix = ox + dx;
        bool ok = kv && pix[j] >= 0;
        if (reflect) {
          iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
        } else {
          ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        }
        stg_load(sa[j], xr, ok ? (unsigned)((pix[j] + (iy - oy) * W + (ix - ox)) * Ci + ci0) * ES : OOB_OFF);
        if constexpr (IN_SCALE) {
          if (ok) {
            const float* sp = d.in_scale + (size_t)((m0 + r0 + RSTEP * j) / HoWo) * Ci + ci0;
            sca[j][0] = *reinterpret_cast<const f32x4*>(sp);
            sca[j][1] = *reinterpret_cast<const f32x4*>(sp + 4);
          } else {
            sca[j][0] = sca[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < RB; ++j)
        stg_load(sb[j], wr, (kv && (wvalid >> j & 1)) ? wbase0 + (unsigned)(j * RSTEP * K + kt * BK) * ES : OOB_OFF);
    }
  };

  auto store_tiles = [&](int stage) {
    char* base = smem + stage * STAGE_BYTES;
    char* a_hi = base;
    char* a_lo = base + A_BYTES;  // only used when F32
    char* b_hi = base + NPLANE * A_BYTES;
    char* b_lo = b_hi + B_BYTES;
    const f32x4 none[2] = {};
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      if constexpr (IN_SCALE) stg_to_lds<T, true>(sa[j], sca[j], a_hi, a_lo, woff0 + j * RSTEP * 128);
      else stg_to_lds<T, false>(sa[j], none, a_hi, a_lo, woff0 + j * RSTEP * 128);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      if (r0 + RSTEP * j < BN) stg_to_lds<T, false>(sb[j], none, b_hi, b_lo, woff0 + j * RSTEP * 128);
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

  // fragment read offsets: tile_off(row, ks*2+lh) = tile_off(row, lh) ^ (ks << 5): the chunk
  // index only enters the XOR-ed 16-B slot field
  int fa[TM], fb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i] = tile_off(wm + i * 32 + lr, lh);
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[j] = tile_off(wn + j * 32 + lr, lh);

  auto compute = [&](int stage) {
    const char* base = smem + stage * STAGE_BYTES;
    const char* a_hi = base;
    const char* a_lo = base + A_BYTES;
    const char* b_hi = base + NPLANE * A_BYTES;
    const char* b_lo = b_hi + B_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 ah[TM], bh[TN], al[F32 ? TM : 1], bl[F32 ? TN : 1];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int off = fa[i] ^ (ks << 5);
        ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + off);
        if constexpr (F32) al[i] = *reinterpret_cast<const bf16x8*>(a_lo + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int off = fb[j] ^ (ks << 5);
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
  if constexpr (DMA) {
    // Two stages, two blocks per CU: the partner block's MFMAs cover this block's fill latency, prologue
    // and epilogue.  (Measured: a 3-stage ring with counted vmcnt needs > 80 KB, i.e. ONE block per CU, and
    // lost 30-40 % on every short-reduction layer -- 128->64 3x3 at 256x256: 437 vs 615 TF/s.)
    dma_tiles(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) dma_tiles(kt + 1, cur ^ 1);  // lands while this stage is multiplied
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
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
  }

  // ---- epilogue -----------------------------------------------------------------------
  // The MFMA leaves each lane with ONE channel and 16 pixels: stored directly that is 64
  // two-byte stores per lane.  Instead the raw fp32 tile goes through LDS (conflict-free
  // ds_write_b32: 32 lanes = 32 consecutive channels of one pixel) and leaves as whole
  // 16-B / 32-B channel vectors of a pixel row, coalesced; demodulation scale, bias,
  // activation and residual are applied on the way out.  One pass per wave-row.
  constexpr int CSTR = BN + 4;  // floats per staged row (+16 B: rows start on different banks)
  float* csm = reinterpret_cast<float*>(smem);
  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  constexpr int VPR = BN / 8;  // 8-channel vectors per row
  const int act = d.act;
  // read-out: a thread keeps ONE 8-channel vector (NT % VPR == 0) and walks rows NT / VPR apart; its bias and
  // (tile inside one sample) demodulation scale are loaded once, ahead of the passes, instead of per row
  static_assert(NT % VPR == 0 && (WM * VPR) % NT == 0, "one channel vector per thread");
  constexpr int RPI = NT / VPR;  // rows per iteration
  const int ec8 = tid % VPR, erow = tid / VPR, en = n0 + ec8 * 8;
  const bool ecol_ok = en < Co;
  float esc[8], ebias[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { esc[q] = 1.f; ebias[q] = 0.f; }
  if (ecol_ok) {
    if (d.out_scale && b_uniform) {
      const float* sp = d.out_scale + (size_t)b_first * Co + en;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { esc[q] = s0[q]; esc[4 + q] = s1[q]; }
    }
    if (d.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + en), b1 = *reinterpret_cast<const f32x4*>(d.bias + en + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ebias[q] = b0[q]; ebias[4 + q] = b1[q]; }
    }
  }
  // Passes: as many wave rows at a time as the (now idle) stage buffers hold -- all of a 256x64 / 128x64 /
  // 256x32 tile in ONE pass, two wave rows of a 256x128 tile -- so the staging writes of several wave rows run
  // in parallel and there are fewer barriers.  With InstanceNorm partials (one per wave row of pixels, the
  // granularity o2m_conv2d_stats_rows reports) it stays one wave row per pass.
  constexpr int LDS_TOTAL = lds_bytes<T, BM, BN, WAVES_M, WAVES_N>();
  constexpr int WPP_MAX = (LDS_TOTAL / (CSTR * 4) / WM) >= WAVES_M ? WAVES_M
                          : ((LDS_TOTAL / (CSTR * 4) / WM) >= 2 && WAVES_M % 2 == 0 ? 2 : 1);
  // Outputs of >= 64 MiB leave through non-temporal stores: written normally, the 134-268 MB of rows of the
  // 256 x 256 layers evict the input lines the filter taps re-read through L2 (kernel alone: 241 -> 210 us on
  // 128 -> 64 at 256 x 256; whole step -0.4 ms in interleaved A/B runs; smaller outputs are better left in
  // the caches for their consumer).
  const bool stream_out = (size_t)M * Co * ES >= ((size_t)64 << 20);
  const bool dot_mode = d.stats && d.stats_mode == O2M_STATS_DOT;
  const T* __restrict__ AUX = static_cast<const T*>(d.aux);
  T* __restrict__ AUXS = static_cast<T*>(d.aux_scaled);
  const int wpp = d.stats ? 1 : WPP_MAX;  // wave rows per pass
  const int npass = WAVES_M / wpp;
  const int wrow_ = wave / WAVES_N;
  const int foldp = FOLD ? d.fold_pad : 0;
  const FoldMap fmap(foldp, Ho, Wo, M);
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    if (wrow_ / wpp == pass) {
      float* dst = csm + (wrow_ % wpp) * WM * CSTR;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dst[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CSTR + wn + j * 32 + lr] = acc[i][j][r];
    }
    lds_barrier();
    const int mbase = m0 + pass * wpp * WM;
    float st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) st[q] = 0.f;
    const int iters = wpp * (WM / RPI);
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
      const int row = erow + it * RPI;
      const int m = mbase + row;
      if (m >= M || !ecol_ok) continue;
      const f32x4 a = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec8 * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec8 * 8 + 4);
      float o[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      bool f_interior = true, f_atomic = false;
      size_t off = (size_t)m * Co + en;
      if constexpr (FOLD) off = (size_t)fmap.pixel(m, f_interior, f_atomic) * Co + en;
      float xv[8];
      if (dot_mode) {  // style dot of the data gradient: sum over pixels of (unscaled result) * aux
        load8x(AUX + off, xv, stream_out);
#pragma unroll
        for (int q = 0; q < 8; ++q) st[q] += o[q] * xv[q];
      }
      if (d.out_scale && !b_uniform) {
        const float* sp = d.out_scale + (size_t)(m / HoWo) * Co + en;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[q] *= s0[q]; o[4 + q] *= s1[q]; }
        if (dot_mode && AUXS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { xv[q] *= s0[q]; xv[4 + q] *= s1[q]; }
        }
      } else if (d.out_scale) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] *= esc[q];
        if (dot_mode && AUXS) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] *= esc[q];
        }
      }
      if (dot_mode && AUXS) store8x(AUXS + off, xv, stream_out);  // aux * out_scale: the modulated input x * s
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] += ebias[q];
      if (d.stats && !dot_mode) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st[q] += o[q]; st[8 + q] += o[q] * o[q]; }
      }
      act_fwd8(o, act);
      if (R && (!FOLD || f_interior)) {
        float rv[8];
        load8(R + off, rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += rv[q];
      }
      if (FOLD && f_atomic) atomic_add8(Y + off, o);
      else if (stream_out && !O2M_NO_STREAMING) store8_stream(Y + off, o);
      else store8(Y + off, o);
    }
    if (d.stats && mbase < M) {  // (a tile's trailing passes can lie past the problem: nothing to report)
      if constexpr (NT % VPR == 0 && VPR * 16 <= NT)
        stats_block_reduce<NT, VPR>(st, csm, d.stats, (long)(mbase / WM), n0, Co, tid);
    }
    if (pass + 1 < npass) lds_barrier();
  }
}

inline long out_rows(const o2m_conv_desc& d) {
  const int S = d.stride > 1 ? d.stride : 1;
  return (long)d.B * ((d.H + 2 * d.pad - d.KH) / S + 1) * ((d.W + 2 * d.pad - d.KW) / S + 1);
}
template <int BM, int BN>
long tiles_rows(const o2m_conv_desc& d, long rows) { return ((rows + BM - 1) / BM) * ((d.Co + BN - 1) / BN); }

template <int BM, int BN>
long tiles_for(const o2m_conv_desc& d) {
  const int S = d.stride > 1 ? d.stride : 1;
  const int Ho = (d.H + 2 * d.pad - d.KH) / S + 1, Wo = (d.W + 2 * d.pad - d.KW) / S + 1;
  const long M = (long)d.B * Ho * Wo;
  return ((M + BM - 1) / BM) * ((d.Co + BN - 1) / BN);
}

// =============================================================================================
// Phase-pipelined variant ("p8") of the 256 x 256 x 64 tile: bf16, Ci % 64 == 0, no in_scale,
// stride 1.  The symmetric kernel above ends every K-stage in `s_waitcnt vmcnt(0)` + barrier with
// all eight waves filling, then all eight multiplying: the matrix pipe idles while the fills are
// issued and the waves park at the drain (profiles/r01_k: 39 % of wave cycles waiting).  Here
//
//  * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64, v_mfma_f32_16x16x32_bf16 (128 accumulator VGPRs);
//  * a K-tile is computed in FOUR phases, one 64 x 32 quadrant of the wave tile each:
//      p1 A0 x B0, p2 A0 x B1, p3 A1 x B1, p4 A1 x B0   (A0/A1 = the wave's pixel rows 0-63 / 64-127,
//      B0/B1 = its filter rows 0-31 / 32-63), so a phase reads at most one new operand half from LDS
//      (8 or 4 ds_read_b128) and issues 16 MFMAs;
//  * a phase is   { LDS reads for its MFMAs ; issue ONE region of LDS-DMA fills ; s_waitcnt vmcnt(8) }
//                 s_barrier  { 16 MFMAs }  s_barrier
//    and the two wave rows run ONE BARRIER APART (the M = 1 row does an extra s_barrier up front): on
//    every SIMD one wave multiplies while its partner reads fragments and issues fills, so the matrix
//    pipe alternates between the two instead of idling;
//  * the fills are cut into four REGIONS per K-tile by the phase that reads them (A0: tile rows
//    0-63 + 128-191, A1: the other A rows, B0 / B1 likewise over the four 64-row filter blocks), each
//    16 fills = 2 per wave.  Region X needed by phase n is issued in phase n - 5, so four regions
//    (8 fills per wave) stay in flight ACROSS the barriers and the only wait in the loop is the counted
//    `vmcnt(8)` = "everything but the four youngest regions has landed"; nothing ever drains to 0.
//    Two 64 KB LDS buffers (K-tile parity); a region is overwritten >= 3 phases after its last read.
//  * past the end of the reduction the fills are issued with out-of-range offsets (hardware zero
//    fill, no memory traffic) so the instruction counts behind vmcnt(8) stay uniform.
//
// Synchronisation rules used (MI355X guide, "Read a staged buffer one phase AFTER the wait that
// retires it"): a region is read only after every wave has passed a vmcnt that covers its fills and
// then a barrier; it is refilled only after every reader has passed the lgkmcnt of its last read and
// then a barrier.  Per phase n the wait leaves regions n+2 .. n+5 in flight, i.e. guarantees n+1.
// =============================================================================================
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#ifndef O2M_P8_CHUNK_MAJOR
#define O2M_P8_CHUNK_MAJOR 0  // (1: chunk-major reduction order -- measured slower, see issue_a; A/B builds: tools/build_variant.sh)
#endif

// FMT: element format of the MFMA operands.  0 = bf16 (x, w, y bf16: v_mfma_f32_16x16x32_bf16).
// 1 / 2 = BASELINE config #5, the fp8 path: x is OCP e4m3 (1) or e5m2 (2, gradients), w is e4m3, y / residual are
// bf16, products on the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (unit E8M0 block scales: the scaling is per
// tensor) with fp32 accumulation and one dequantisation factor (d.deq_scale[0] * d.deq_scale[2]: the {1/scale, amax}
// pairs of x and w sit at [0..1] and [2..3]) applied to the accumulator.  A 128-B LDS row holds 128 reduction elements
// and ONE such MFMA (32 cycles, twice the bf16 16x16x32) consumes all of them: the same fills, the same LDS reads and
// the same MFMA cycles per K-tile as the bf16 form for twice the reduction depth.  (The non-scaled fp8 16x16x32 runs
// at the bf16 rate on gfx950 -- rounds 2-3 used it and config #5 gained nothing.)  Operand layout, probed on the
// device (build/probe/mfma_scale_probe.hip): a lane holds 32 bytes of row (l & 15); which reduction elements they are
// is free as long as both operands agree -- here the two 16-B slots 2 (l >> 4), 2 (l >> 4) + 1 of the row.
template <int FMT>
__global__ __launch_bounds__(512, 2) void conv_igemm_p8_kernel(const o2m_conv_desc d, const int m_begin, const int m_end) {
  using T = unsigned short;               // y / residual element (bf16 in every format)
  constexpr int ES = FMT ? 1 : 2;         // operand element size
  constexpr int KT = 128 / ES;            // reduction elements per K-tile (one 128-B row)
  constexpr int KS = FMT ? 1 : 2;         // MFMA k-steps per K-tile (bf16: 32 elements each; fp8: all 128 at once)
  constexpr int BM = 256, BN = 256, NT = 512;
  constexpr int OPB = 32768;   // one operand of one K-tile: 256 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, KH = d.KH, KW = d.KW, pad = d.pad;
  const int Ho = H + 2 * pad - KH + 1, Wo = W + 2 * pad - KW + 1;
  const int HoWo = Ho * Wo;
  const int M = m_end;  // rows [m_begin, m_end) of the B * Ho * Wo output pixels (a launch may cover a slice)
  const int K = KH * KW * Ci;
  const int nk = K / KT;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;

  const int tiles_n = (Co + BN - 1) / BN;
  const int tile = xcd_tile_order(blockIdx.x, gridDim.x);
  const int m0 = m_begin + (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;
  const int b_first = m0 / HoWo;
  const bool b_uniform = (min(m0 + BM, M) - 1) / HoWo == b_first;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * ES));
  const rsrc_t wr = make_rsrc(static_cast<const char*>(d.w) + (size_t)b_first * d.w_batch_stride * ES,
                              (unsigned)((size_t)Co * K * ES));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wrow = wave >> 2, wcol = wave & 3;
  typedef __attribute__((address_space(3))) void lds_void;

  // ---- fills: slot j = 2 * region + (0 | 1); the wave owns entries 2w, 2w+1 of each region's list ----
  // A regions: A0 = 8-row groups {0..7, 16..23}, A1 = {8..15, 24..31};  B: B0 = {8c + 0..3}, B1 = {8c + 4..7}
  auto a_group = [&](int j) { const int i = 2 * wave + (j & 1); return (i < 8 ? i : i + 8) + 8 * (j >> 1); };
  auto b_group = [&](int j) { const int i = 2 * wave + (j & 1); return 8 * (i >> 2) + 4 * (j >> 1) + (i & 3); };
  // lane l of the fill of group q owns linear 16-B slot 64 q + l = (row, chunk) under tile_off's swizzle
  auto slot_row = [&](int q) { const int pr = 4 * q + (lane >> 4); return 2 * pr + (((lane & 15) ^ (pr & 15)) >> 3); };
  auto slot_chk = [&](int q) { const int pr = 4 * q + (lane >> 4); return ((lane & 15) ^ (pr & 15)) & 7; };

  // Per fill (8 pixels; this lane's pixel = its slot row): cbase = byte offset of the pixel's own (centre) element +
  // this lane's 16-B chunk, msk = which taps stay inside the image: bit ky (rows), bit 8 + kx (columns), bit 16 = the
  // output row exists.  A tap's address is then  cbase + ty + tx  with ty = +-(ky - pad) W Ci, tx = +-(kx - pad) Ci
  // bytes: a cleared bit means zero fill (ZERO padding: out-of-range offset) or the mirrored element (REFLECT, pad 1:
  // the sign of that component flips) -- a handful of VALU per border fill and K-tile instead of the full gather.
  const RowDiv by_howo(HoWo, M), by_wo(Wo, M);
  unsigned cbase[4], msk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + slot_row(a_group(j));
    cbase[j] = 0;
    msk[j] = 0;
    if (m < M) {
      const int b = by_howo.div(m), rem = m - b * HoWo;
      const int oy = by_wo.div(rem), ox = rem - oy * Wo;
      cbase[j] = (unsigned)(((b * H + oy) * W + ox) * Ci * ES + slot_chk(a_group(j)) * 16);
      unsigned mk = 1u << 16;
      for (int ky = 0; ky < KH; ++ky) mk |= (unsigned)((unsigned)(oy + ky - pad) < (unsigned)H) << ky;
      for (int kx = 0; kx < KW; ++kx) mk |= (unsigned)((unsigned)(ox + kx - pad) < (unsigned)W) << (8 + kx);
      msk[j] = mk;
    }
  }
  unsigned dwoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = b_group(j);
    // LDS filter row rho = 64 wcol + 16 j + i holds output channel 64 wcol + 32 (j >> 1) + 8 (i >> 2) + 4 (j & 1) +
    // (i & 3), so that the accumulators of one lane are 8 CONSECUTIVE channels (see the epilogue)
    const int rho = slot_row(q);
    const int n = n0 + (rho & ~63) + 32 * ((rho >> 5) & 1) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3);
    dwoff[j] = n < Co ? (unsigned)(n * K * ES + slot_chk(q) * 16) : OOB_OFF;
  }
  unsigned aoff[4];
  // Interior fills: all 8 pixels of the fill have their whole KH x KW window inside the image: their tap address is
  // cbase + one wave-uniform byte offset (one v_add per tap).  Border fills combine cbase / msk per tap.
  unsigned inner = 0;  // bit j (wave-uniform)
  const unsigned full = (1u << 16) | (((1u << KW) - 1) << 8) | ((1u << KH) - 1);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (__all(msk[j] == full)) inner |= 1u << j;
  inner = __builtin_amdgcn_readfirstlane(inner);

  // The loader half of a phase runs beside the partner wave's 16 MFMAs, which hold the issue priority: every VALU
  // and every branch here is paid several times over (MI355X guide, "Two waves per SIMD"; the round-3 loop spent
  // ~35 SALU + 6 VALU + 2 taken branches per phase on buffer parities, tap state and dead-fill selects and its
  // loader half, not the MFMA half, set the phase length).  So:
  //  * the K loop is unrolled by two and the LDS image is [A(0) | A(1) | B(0) | B(1)] (K-tile parity): every fragment
  //    read and every fill destination is a compile-time offset from one base register (ds_read offset:imm, m0 = imm);
  //  * a region's stream state is two scalars (channel-chunk byte offset, remaining advances) plus the tap; the
  //    per-lane tap offsets are recomputed only when the chunk wraps (every Ci / 64 K-tiles);
  //  * past the end of the reduction a region simply re-fetches its LAST K-tile into the buffer nobody reads any
  //    more (2-3 extra L2 hits per tile) instead of selecting out-of-range offsets per fill.
  const int CiB = Ci * ES;  // bytes of one pixel's channels = distance between taps kx, kx + 1
  int a_ky[2] = {0, 0}, a_kx[2] = {0, 0}, a_cbB[2] = {0, 0}, a_left[2] = {nk - 1, nk - 1};
  int b_koff[2] = {0, 0};
  int b_tap[2] = {0, 0};
  const int b_klast = (nk - 1) * 128;
  int a_dst[4], b_dst[4];  // LDS byte offset of fill j inside its operand tile (wave-uniform)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a_dst[j] = __builtin_amdgcn_readfirstlane(a_group(j) * 1024);
    b_dst[j] = __builtin_amdgcn_readfirstlane(b_group(j) * 1024 + 2 * OPB);
  }
  auto new_tap = [&](int r) {  // lane offsets of region r's two fills for the tap (a_ky[r], a_kx[r])
    const unsigned sdy = (unsigned)((a_ky[r] - pad) * W * CiB), sdx = (unsigned)((a_kx[r] - pad) * CiB);
    const unsigned need = (1u << 16) | (reflect ? 0u : (1u << a_ky[r]) | (1u << (8 + a_kx[r])));
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * r + jj;
      if (inner >> j & 1) {
        aoff[j] = cbase[j] + (sdy + sdx);
      } else {
        const unsigned ty = (msk[j] >> a_ky[r] & 1) ? sdy : 0u - sdy;  // (only REFLECT gets here with a cleared bit)
        const unsigned tx = (msk[j] >> (8 + a_kx[r]) & 1) ? sdx : 0u - sdx;
        aoff[j] = (msk[j] & need) == need ? cbase[j] + ty + tx : OOB_OFF;
      }
    }
  };
  new_tap(0);
  new_tap(1);
  auto issue_a = [&](int r, int buf) {  // (r, buf: literals at every call site)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * r + jj;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(smem + buf * OPB + a_dst[j]), 16, (int)aoff[j],
                                               __builtin_amdgcn_readfirstlane(a_cbB[r]), 0, 0);
    }
    if (a_left[r] > 0) {
      --a_left[r];
#if O2M_P8_CHUNK_MAJOR
      // EXPERIMENT (off): 64-channel chunk outermost, the KH x KW taps inside it, so that a CU cycles through one chunk of
      // its input patch for all taps while the lines the shifted taps share are still in the XCD's L2 (tap-major: 0.77 hit
      // rate, 3.8 x the input fetched from the fabric, profiles/r04_z_pmc_igemm_p8.json).  Measured SLOWER: 3 x 3 256 -> 256
      // at 64 x 64, B = 48 / 32 / 16: 222 / 138 / 70 us against 200 / 128 / 65 us tap-major (profiles/r04_t_p8_order.txt) --
      // the per-K-tile tap switch (new_tap: offsets of every fill recomputed in the loader half) costs more than the
      // L2 misses, which the five-phase prefetch distance already hides.
      if (++a_kx[r] == KW) {
        a_kx[r] = 0;
        if (++a_ky[r] == KH) { a_ky[r] = 0; a_cbB[r] += 128; }
      }
      new_tap(r);
#else
      a_cbB[r] += 128;
      if (a_cbB[r] == CiB) {
        a_cbB[r] = 0;
        if (++a_kx[r] == KW) { a_kx[r] = 0; ++a_ky[r]; }
        new_tap(r);
      }
#endif
    }
  };
  auto issue_b = [&](int r, int buf) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * r + jj;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(smem + buf * OPB + b_dst[j]), 16, (int)dwoff[j],
                                               __builtin_amdgcn_readfirstlane(b_koff[r]), 0, 0);
    }
#if O2M_P8_CHUNK_MAJOR
    if (b_koff[r] != b_klast) {  // (the filter's K index is tap * Ci + channel: one tap further = + CiB bytes)
      b_koff[r] += CiB;
      if (++b_tap[r] == KH * KW) { b_tap[r] = 0; b_koff[r] += 128 - KH * KW * CiB; }
    }
#else
    b_koff[r] = min(b_koff[r] + 128, b_klast);
#endif
  };

  // ---- fragments ------------------------------------------------------------------------------------
  // lane l holds row l & 15 of a 16-row tile.
  //   bf16 (16x16x32): reduction elements 8 (l >> 4) + 32 ks .. + 7 = 16-B chunk (l >> 4) + 4 ks    (ds_read_b128)
  //   fp8 (16x16x128): the 32 bytes of chunks 2 (l >> 4), 2 (l >> 4) + 1, in that order              (2 ds_read_b128;
  //          the swizzle puts the second at the first's address ^ 16)
  // tile_off(R0 + 16 i + r, c ^ x) = (tile_off(R0 + r, c) ^ ((i & 1) << 7 | x << 4)) + i * 2048  for R0 % 32 == 0
  typedef __attribute__((ext_vector_type(4))) int i32x4_t;
  typedef __attribute__((ext_vector_type(8))) int i32x8_t;
  using frag_t = std::conditional_t<FMT == 0, bf16x8, i32x8_t>;
  const int c0 = FMT ? 2 * (lane >> 4) : (lane >> 4);
  const int fa0 = tile_off(128 * wrow + (lane & 15), c0);
  const int fb0 = tile_off(64 * wcol + (lane & 15), c0) + 2 * OPB;
  auto read_frag = [&](const char* base, int off) -> frag_t {
    if constexpr (FMT == 0) {
      return *reinterpret_cast<const bf16x8*>(base + off);
    } else {
      const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(base + off), hi = *reinterpret_cast<const i32x4_t*>(base + (off ^ 16));
      return i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  frag_t af[4][KS], b0f[2][KS], b1f[2][KS];
  auto read_a = [&](int buf, int mh) {
    const char* base = smem + buf * OPB + mh * (64 * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) af[i][ks] = read_frag(base, (fa0 ^ (((i & 1) << 7) | (ks << 6))) + i * 2048);
  };
  auto read_b = [&](frag_t (&bf)[2][KS], int buf, int nh) {
    const char* base = smem + buf * OPB + nh * (32 * 128);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bf[j][ks] = read_frag(base, (fb0 ^ (((j & 1) << 7) | (ks << 6))) + j * 2048);
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](const frag_t (&bf)[2][KS], int mh, int nh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4_t& c = acc[mh * 4 + i][nh * 2 + j];
          // filter rows as the A operand: a lane's 4 result registers are 4 CHANNELS of one pixel
          // (cbsz / blgp = element format of the A / B operand: 0 = e4m3, 1 = e5m2; scale bytes 0x7f = 2^0)
          if constexpr (FMT == 0) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks], af[i][ks], c, 0, 0, 0);
          else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j][ks], af[i][ks], c, 0, FMT == 2 ? 1 : 0, 0, 0x7f7f7f7f, 0,
                                                                    0x7f7f7f7f);
        }
    __builtin_amdgcn_s_setprio(0);
  };
#define P8_WAIT_AND_SYNC()                                  \
  do {                                                      \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                           \
  } while (0)
#define P8_CLOSE() __builtin_amdgcn_s_barrier()

  // ---- prologue: regions needed by phases -1 .. 4 -------------------------------------------------------
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
    // p1: A0 x B0
    read_a(cur, 0);
    issue_b(1, cur ^ 1);  // B1(t+1)
    P8_WAIT_AND_SYNC();
    multiply(b0f, 0, 0);
    P8_CLOSE();
    // p2: A0 x B1
    read_b(b1f, cur, 1);
    issue_a(1, cur ^ 1);  // A1(t+1)
    P8_WAIT_AND_SYNC();
    multiply(b1f, 0, 1);
    P8_CLOSE();
    // p3: A1 x B1
    read_a(cur, 1);
    issue_b(0, cur);  // B0(t+2)
    P8_WAIT_AND_SYNC();
    multiply(b1f, 1, 1);
    P8_CLOSE();
    // p4: A1 x B0, then the next K-tile's B0 fragments (landed: waited for at the end of p3)
    issue_a(0, cur);  // A0(t+2)
    P8_WAIT_AND_SYNC();
    multiply(b0f, 1, 0);
    read_b(b0f, cur ^ 1, 0);
    P8_CLOSE();
  };
#pragma unroll 1
  for (int t = 0; t < nk; t += 2) {
    ktile(0);
    if (t + 1 < nk) ktile(1);
  }
#undef P8_WAIT_AND_SYNC
#undef P8_CLOSE
  if (wrow == 0) __builtin_amdgcn_s_barrier();

  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  const int act = d.act;
  const float deq = (FMT != 0 && d.deq_scale) ? d.deq_scale[0] * d.deq_scale[2] : 1.f;  // {1/scale, amax} of x, of w
  // outputs of >= 64 MiB leave through non-temporal stores, as in conv_igemm_kernel (no cache level holds them for
  // the consumer; written normally they evict the input lines the taps re-read)
#ifndef O2M_P8_STREAM_OUT
#define O2M_P8_STREAM_OUT 1  // (0: A/B builds, tools/build_variant.sh)
#endif
  const bool stream_out = O2M_P8_STREAM_OUT && (size_t)M * Co * sizeof(T) >= ((size_t)64 << 20);
  const bool dot_mode = d.stats && d.stats_mode == O2M_STATS_DOT;
  const T* __restrict__ AUX = static_cast<const T*>(d.aux);
  T* __restrict__ AUXS = static_cast<T*>(d.aux_scaled);

  // ---- epilogue straight from the accumulators -------------------------------------------------------
  // The filter rows were the A operand and sit permuted in LDS (dwoff above), so lane l of acc[i][2h], acc[i][2h+1]
  // holds, for pixel 16 i + (l & 15) of the wave's 128, the EIGHT consecutive channels 32 h + 8 (l >> 4) .. + 7 of
  // the wave's 64: one 16-B vector of the output row, 64 B per pixel per store instruction.  No LDS staging, no
  // barrier: a wave leaves as soon as its own 16 vectors per lane are out (the staged form cost two passes of
  // 128 ds_write_b32 per lane + 4 barriers: profiles/r02, 11 of the 71 us of a tile).
  const int g = lane >> 4, pl = lane & 15;
  const int mwave = m0 + 128 * wrow;
  const int foldp = d.fold_pad;
  const FoldMap fmap(foldp, Ho, Wo, M);
  float esc[2][8], ebias[2][8];
  int en[2];
  bool col_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    en[h] = n0 + 64 * wcol + 32 * h + 8 * g;
    col_ok[h] = en[h] < Co;
#pragma unroll
    for (int q = 0; q < 8; ++q) { esc[h][q] = deq; ebias[h][q] = 0.f; }
    if (col_ok[h]) {
      if (d.out_scale && b_uniform) {
        const float* sp = d.out_scale + (size_t)b_first * Co + en[h];
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { esc[h][q] *= s0[q]; esc[h][4 + q] *= s1[q]; }
      }
      if (d.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + en[h]);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(d.bias + en[h] + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { ebias[h][q] = b0[q]; ebias[h][4 + q] = b1[q]; }
      }
    }
  }
  float st[2][16];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 16; ++q) st[h][q] = 0.f;
  const float rdeq = 1.f / deq;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = mwave + 16 * i + pl;
    bool f_interior = true, f_atomic = false;
    size_t pix_fold = 0;
    if (foldp && m < M) pix_fold = (size_t)fmap.pixel(m, f_interior, f_atomic);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (m >= M || !col_ok[h]) continue;
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { o[r] = acc[i][2 * h][r]; o[4 + r] = acc[i][2 * h + 1][r]; }
      const size_t off = (foldp ? pix_fold : (size_t)m) * Co + en[h];
      float xv[8];
      if (dot_mode) {  // style dot of the data gradient: sum over pixels of (unscaled result) * aux
        load8x(AUX + off, xv, stream_out);
#pragma unroll
        for (int q = 0; q < 8; ++q) st[h][q] += o[q] * deq * xv[q];
      }
      if (d.out_scale && !b_uniform) {
        const float* sp = d.out_scale + (size_t)(m / HoWo) * Co + en[h];
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[q] *= s0[q] * deq; o[4 + q] *= s1[q] * deq; }
        if (dot_mode && AUXS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { xv[q] *= s0[q]; xv[4 + q] *= s1[q]; }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] *= esc[h][q];
        if (dot_mode && AUXS && d.out_scale) {  // (a one-sample tile: esc = dequantisation x out_scale)
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] *= esc[h][q] * rdeq;
        }
      }
      if (dot_mode && AUXS) store8x(AUXS + off, xv, stream_out);  // aux * out_scale: the modulated input x * s
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] += ebias[h][q];
      if (d.stats && !dot_mode) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st[h][q] += o[q]; st[h][8 + q] += o[q] * o[q]; }
      }
      act_fwd8(o, act);
      if (R && f_interior) {
        float rv[8];
        load8(R + off, rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += rv[q];
      }
      if (f_atomic) atomic_add8(Y + off, o);
      else store8x(Y + off, o, stream_out);
    }
  }
  if (d.stats && mwave < M) {
    // the wave's 128 pixels x 64 channels ARE one row of the partial table: sum over the 16 lanes that share a
    // channel vector (quad swaps, then rotations inside the row of 16), fixed order, no LDS, no atomics
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = st[h][q];
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));  // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));  // row_ror:8
        st[h][q] = v;
      }
    if (pl == 0) {
      const size_t part = (size_t)(mwave / 128);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (!col_ok[h]) continue;
        float* sp = d.stats + (part * Co + en[h]) * 2;  // [part][channel][sum | sum of squares]
#pragma unroll
        for (int q = 0; q < 8; q += 2)
          *reinterpret_cast<f32x4*>(sp + 2 * q) = f32x4{st[h][q], st[h][8 + q], st[h][q + 1], st[h][8 + q + 1]};
      }
    }
  }
}

// what conv_igemm_p8_kernel's fills assume (beyond the dtype and the tile count, which the callers check)
inline bool p8_geometry_ok(const o2m_conv_desc& d) {
  return !d.in_scale && d.stride <= 1 && d.Ci % BK == 0 && d.KH <= 8 && d.KW <= 8 &&
         (d.pad_mode != O2M_PAD_REFLECT || d.pad <= 1);  // mirrored taps as a sign flip: pad 1 only
}

int launch_p8(const o2m_conv_desc& d, hipStream_t s, long m_begin, long m_end) {
  constexpr int lds_main = 2 * 65536, lds_epi = 128 * (256 + 4) * 4;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  const long tiles = tiles_rows<256, 256>(d, m_end - m_begin);
  if (tiles <= 0 || tiles > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  auto kern = d.dtype == O2M_FP8_E4M3 ? conv_igemm_p8_kernel<1>
                                      : (d.dtype == O2M_BF8_E5M2 ? conv_igemm_p8_kernel<2> : conv_igemm_p8_kernel<0>);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * (m_end - m_begin) * d.Co * d.KH * d.KW * d.Ci, "conv_igemm_p8<%s,256x256>",
                      d.dtype == O2M_BF16 ? "bf16" : "fp8");
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, s, d, (int)m_begin, (int)m_end);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

// =============================================================================================
// Halo-tile kernel ("h3") for the NARROW 3 x 3 layers at large maps: zero padding 1, stride 1, bf16,
// Ci % 64 == 0, Co = 64 or 128, W % 32 == 0, H % 8 == 0 (the 64 <-> 128-channel layers at 256 x 256 / 128 x 128
// and their data gradients).  The implicit-GEMM tiles above fetch the A operand once PER TAP: nine L2 -> LDS
// trips per input element, 52 FLOP per ingested byte at N = 64 -- they sit on the L2 -> LDS ingest limit
// (~60 GB/s per CU) at 0.2 of the MFMA peak (profiles/r02_c_pmc_igemm_256x64.json).  Here a block owns an
// 8 x 32 tile of output pixels of ONE sample and keeps the 10 x 34 input patch of a 64-channel chunk resident
// in LDS (43 KB): all nine taps read their A fragments from it at shifted pixel offsets, so an input element
// is ingested once per chunk (plus the 1.33x halo), 3.4x less than before; only the filter taps (8 / 16 KB
// each) stream, double-buffered, one tap ahead.
//  * 8 waves = 4 (M: two image rows of the tile each) x 2 (N), v_mfma_f32_16x16x32_bf16, 32 / 64 accumulator
//    registers: TWO blocks per CU (60 / 77 KB of LDS each), which is what hides a block's patch load and
//    epilogue -- no phase choreography needed.
//  * patch image: pixel pp = py * 34 + px at byte pp * 128, 16-B chunk c at slot c ^ ((pp >> 1) & 7): 16
//    consecutive pixels (one fragment read, at ANY tap shift) cover all 16 slots of their bank rows.  Filled by
//    LDS-DMA with the swizzle on the source side; out-of-image pixels (the zero padding) are out-of-range
//    offsets, i.e. hardware zero fill.
// Epilogue as in conv_igemm_kernel (fp32 tile through LDS, whole channel vectors out; demodulation, bias,
// activation, residual, InstanceNorm / style-dot partials), one wave row = 64 pixels = two image-row segments
// per pass.
// =============================================================================================
template <int CO>
__global__ __launch_bounds__(512, 4) void conv3x3_halo_kernel(const o2m_conv_desc d) {
  using T = unsigned short;
  constexpr int NT = 512, TH = 8, TW = 32, PW = TW + 2, NPIX = (TH + 2) * PW;  // 340 patch pixels
  constexpr int PFILLS = (NPIX + 7) / 8;                                       // 43 fills of 8 pixels
  constexpr int PATCH_B = PFILLS * 1024;
  constexpr int WB = CO * 128;     // one filter tap of one chunk: CO rows x 64 channels
  constexpr int WFILLS = CO / 8;   // 8 rows per fill
  constexpr int WPW = WFILLS / 8;  // filter fills per wave (1 or 2)
  constexpr int NJ = CO / 32;      // 16-column MFMA tiles per wave (wave N = CO / 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wbuf = smem + PATCH_B;
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co;
  const int K = 9 * Ci;
  const int tiles_x = W / TW, tiles_y = H / TH, tps = tiles_x * tiles_y;
  const int tile = xcd_tile_order(blockIdx.x, gridDim.x);
  const int b = tile / tps, tis = tile - b * tps;
  const int ty0 = (tis / tiles_x) * TH, tx0 = (tis % tiles_x) * TW;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t wr = make_rsrc(static_cast<const char*>(d.w) + (size_t)b * d.w_batch_stride * 2, (unsigned)((size_t)Co * K * 2));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- patch fills: fill f = 8 j + wave covers patch pixels 8 f .. 8 f + 7; lane l owns slot (l & 7) of pixel
  // 8 f + (l >> 3).  The six source offsets are recomputed per chunk (once or twice per block) rather than kept
  // in registers across the tap loop: the kernel must fit 128 VGPRs for two blocks per CU.
  auto issue_patch = [&](int cb) {
    int ln = lane;
    asm volatile("" : "+v"(ln));  // keeps the offset arithmetic below inside the chunk loop
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int f = 8 * j + wave;
      if (f >= PFILLS) continue;  // wave-uniform
      const int pp = 8 * f + (ln >> 3);
      const int c = (ln & 7) ^ ((pp >> 1) & 7);
      const int py = pp / PW, px = pp - py * PW;
      const int gy = ty0 + py - 1, gx = tx0 + px - 1;
      const bool ok = pp < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((b * H + gy) * W + gx) * Ci + c * 8) * 2u : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(patch + f * 1024), 16, (int)off, cb * 2, 0, 0);
    }
  };
  // ---- filter fills: tile_off's swizzle, lane l of the fill of 8-row group q owns linear slot 64 q + l
  unsigned woff[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int q = WPW * wave + j;
    const int pr = 4 * q + (lane >> 4);
    const int row = 2 * pr + (((lane & 15) ^ (pr & 15)) >> 3), chk = ((lane & 15) ^ (pr & 15)) & 7;
    woff[j] = (unsigned)(row * K + chk * 8) * 2u;
  }
  auto issue_w = [&](int tap, int cb, int buf) {
#pragma unroll
    for (int j = 0; j < WPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(wbuf + buf * WB + (WPW * wave + j) * 1024), 16, (int)woff[j],
                                               (tap * Ci + cb) * 2, 0, 0);
  };

  // ---- fragments -------------------------------------------------------------------------------------
  // A: lane l holds pixel row (l & 15) of a 16-row tile and reduction elements 8 (l >> 4) + 32 ks .. + 7
  const int c0 = lane >> 4;
  // patch pixel of this lane's row at tap (0, 0) in 16-row tile 0; tile i adds (i >> 1) image rows and 16 (i & 1) pixels
  const int ppb0 = 2 * wm * PW + (lane & 15);
  // tile_off(R0 + 16 j + r, c) = (tile_off(R0 + r, c) ^ ((j & 1) << 7)) + j * 2048 for R0 % 32 == 0 (as in the p8 kernel)
  const int fb0 = tile_off(wn * (CO / 2) + (lane & 15), c0);

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const int toff, const int buf) {
    int aoff[4];
    int pb = ppb0;
    asm volatile("" : "+v"(pb));  // recompute the four offsets per tap: hoisted, the 36 of them spill to scratch
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = pb + (toff + (i >> 1) * PW + (i & 1) * 16);
      aoff[i] = (pp << 7) | ((c0 ^ ((pp >> 1) & 7)) << 4);
    }
    const char* wb = wbuf + buf * WB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], bw[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(patch + (aoff[i] ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bw[j] = *reinterpret_cast<const bf16x8*>(wb + ((fb0 ^ (((j & 1) << 7) | (ks << 6))) + j * 2048));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bw[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: per 64-channel chunk, the patch once and the nine taps from it ------------------------
  for (int cb = 0; cb < Ci; cb += 64) {
    issue_patch(cb);
    issue_w(0, cb, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) issue_w(tap + 1, cb, (tap + 1) & 1);  // lands while this tap is multiplied
      compute((tap / 3) * PW + tap % 3, tap & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // next tap's filter has landed; every wave is done with this tap's (and, at tap 8, the patch)
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------------
  constexpr int CSTR = CO + 4;
  constexpr int VPR = CO / 8, RPI = NT / VPR;  // 8-channel vectors per row; rows per read-out iteration
  float* csm = reinterpret_cast<float*>(smem);
  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  const T* __restrict__ AUX = static_cast<const T*>(d.aux);
  T* __restrict__ AUXS = static_cast<T*>(d.aux_scaled);
  const int act = d.act;
  const bool dot_mode = d.stats && d.stats_mode == O2M_STATS_DOT;
  const bool stream_out = (size_t)d.B * H * W * Co * 2 >= ((size_t)64 << 20);
  const int ec8 = tid % VPR, erow = tid / VPR, en = ec8 * 8;
  float esc[8], ebias[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { esc[q] = 1.f; ebias[q] = 0.f; }
  if (d.out_scale) {
    const float* sp = d.out_scale + (size_t)b * Co + en;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { esc[q] = s0[q]; esc[4 + q] = s1[q]; }
  }
  if (d.bias) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + en), b1 = *reinterpret_cast<const f32x4*>(d.bias + en + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { ebias[q] = b0[q]; ebias[4 + q] = b1[q]; }
  }
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {  // wave row `pass`: tile image rows 2 pass, 2 pass + 1
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            csm[(i * 16 + 4 * (lane >> 4) + r) * CSTR + wn * (CO / 2) + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    lds_barrier();
    float st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) st[q] = 0.f;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int row = erow + it * RPI;  // 0 .. 63
      const int gy = ty0 + 2 * pass + (row >> 5), gx = tx0 + (row & 31);
      const size_t off = ((size_t)(b * H + gy) * W + gx) * Co + en;
      const f32x4 va = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec8 * 8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec8 * 8 + 4);
      float o[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      if (dot_mode) {
        float xv[8];
        load8x(AUX + off, xv, stream_out);
#pragma unroll
        for (int q = 0; q < 8; ++q) st[q] += o[q] * xv[q];
        if (AUXS) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] *= esc[q];
          store8x(AUXS + off, xv, stream_out);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = o[q] * esc[q] + ebias[q];
      if (d.stats && !dot_mode) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st[q] += o[q]; st[8 + q] += o[q] * o[q]; }
      }
      act_fwd8(o, act);
      if (R) {
        float rv[8];
        load8(R + off, rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += rv[q];
      }
      store8x(Y + off, o, stream_out);
    }
    // one partial per (tile, wave row): 64 pixels of the sample; o2m_instnorm_finalize / o2m_conv2d_dots_finalize
    // add a sample's H W / 64 partials whatever pixels each covers
    if (d.stats) stats_block_reduce<NT, VPR>(st, csm, d.stats, ((long)b * tps + tis) * 4 + pass, 0, Co, tid);
    lds_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Four-wave form of the halo tile for Co = 128 ("h3w").  conv3x3_halo_kernel<128> needs its 64 accumulators + fragments +
// epilogue state in 128 VGPRs (two 8-wave blocks per CU): it spills 25 registers and its epilogue takes the 8 x 32 x 128
// fp32 tile through LDS in four barrier-separated passes -- at Ci = 64 (K = 576: the 64 <-> 128-channel layers at
// 256 x 256) that epilogue is as long as the nine taps and the kernel ran at 0.26-0.32 of peak (profiles/README.md).
// Here a block is FOUR waves (one per SIMD, 256 VGPRs each, still two blocks per CU = two independent waves per SIMD):
//  * wave w owns image rows 2 w, 2 w + 1 of the tile (64 pixels) x all 128 channels: 128 accumulator VGPRs, per tap and
//    k-step 4 pixel + 8 filter fragment reads for 32 MFMAs (0.375 ds_read_b128 per MFMA, 0.5 before);
//  * the filter is the A operand and its rows sit permuted in LDS (as in conv_igemm_p8_kernel), so a lane's accumulators
//    are 8 CONSECUTIVE channels of one pixel: the epilogue goes straight from registers to 16-B global stores -- no LDS
//    staging, no barrier, a wave leaves when its own stores are out, and the partner block's MFMAs fill the SIMD;
//  * one barrier per tap (4 waves): { wait for this tap's filter ; barrier ; issue the next tap's fills ; 64 MFMAs }.
// Same patch / filter images, preconditions and partial-statistics rows (one per (tile, wave) = 64 pixels) as the 8-wave form.
__global__ __launch_bounds__(256, 2) void conv3x3_halo_w4_kernel(const o2m_conv_desc d) {
  using T = unsigned short;
  constexpr int CO = 128, TH = 8, TW = 32, PW = TW + 2, NPIX = (TH + 2) * PW;  // 340 patch pixels
  constexpr int PFILLS = (NPIX + 7) / 8;                                       // 43 fills of 8 pixels
  constexpr int PATCH_B = PFILLS * 1024;
  constexpr int WB = CO * 128;  // one filter tap of one chunk: 128 rows x 64 channels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wbuf = smem + PATCH_B;
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co;
  const int K = 9 * Ci;
  const int tiles_x = W / TW, tiles_y = H / TH, tps = tiles_x * tiles_y;
  const int tile = xcd_tile_order(blockIdx.x, gridDim.x);
  const int b = tile / tps, tis = tile - b * tps;
  const int ty0 = (tis / tiles_x) * TH, tx0 = (tis % tiles_x) * TW;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t wr = make_rsrc(static_cast<const char*>(d.w) + (size_t)b * d.w_batch_stride * 2, (unsigned)((size_t)Co * K * 2));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

  // ---- patch fills: fill f = 4 j + wave covers patch pixels 8 f .. 8 f + 7; lane l owns slot (l & 7) of pixel 8 f + (l >> 3)
  unsigned poff[11];
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const int f = 4 * j + wave;
    const int pp = 8 * f + (lane >> 3);
    const int c = (lane & 7) ^ ((pp >> 1) & 7);
    const int py = pp / PW, px = pp - py * PW;
    const int gy = ty0 + py - 1, gx = tx0 + px - 1;
    const bool ok = f < PFILLS && pp < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[j] = ok ? (unsigned)(((b * H + gy) * W + gx) * Ci + c * 8) * 2u : OOB_OFF;
  }
  auto issue_patch = [&](int cb) {
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int f = 4 * j + wave;
      if (f >= PFILLS) continue;  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(patch + f * 1024), 16, (int)poff[j], cb * 2, 0, 0);
    }
  };
  // ---- filter fills: tile_off's swizzle, lane l of the fill of 8-row group q owns linear slot 64 q + l; LDS row rho holds
  // output channel 32 (rho >> 5) + 8 ((rho >> 2) & 3) + 4 ((rho >> 4) & 1) + (rho & 3) (see the epilogue)
  unsigned woff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = 4 * wave + j;
    const int pr = 4 * q + (lane >> 4);
    const int rho = 2 * pr + (((lane & 15) ^ (pr & 15)) >> 3), chk = ((lane & 15) ^ (pr & 15)) & 7;
    const int n = 32 * (rho >> 5) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3);
    woff[j] = (unsigned)(n * K + chk * 8) * 2u;
  }
  auto issue_w = [&](int tap, int cb, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(wbuf + buf * WB + (4 * wave + j) * 1024), 16, (int)woff[j],
                                               (tap * Ci + cb) * 2, 0, 0);
  };

  // ---- fragments: lane l holds row (l & 15) of a 16-row tile and reduction elements 8 (l >> 4) + 32 ks .. + 7
  const int c0 = lane >> 4;
  const int ppb0 = 2 * wave * PW + (lane & 15);  // patch pixel of this lane at tap (0, 0), pixel tile 0
  const int fb0 = tile_off(lane & 15, c0);

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const int toff, const int buf) {
    int aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // pixel tile i: image row (i >> 1) of the wave's two, pixels 16 (i & 1) .. + 15
      const int pp = ppb0 + (toff + (i >> 1) * PW + (i & 1) * 16);
      aoff[i] = (pp << 7) | ((c0 ^ ((pp >> 1) & 7)) << 4);
    }
    const char* wb = wbuf + buf * WB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 px[4], wf[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) px[i] = *reinterpret_cast<const bf16x8*>(patch + (aoff[i] ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wb + ((fb0 ^ (((j & 1) << 7) | (ks << 6))) + j * 2048));
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], px[i], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: per 64-channel chunk, the patch once and the nine taps from it ------------------------
  for (int cb = 0; cb < Ci; cb += 64) {
    if (cb) __syncthreads();  // every wave is done with the previous chunk's patch and filter buffers
    issue_patch(cb);
    issue_w(0, cb, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // this tap's filter (at tap 0: the patch) has landed for every wave; all are done with tap - 1
      if (tap < 8) issue_w(tap + 1, cb, (tap + 1) & 1);  // lands while this tap is multiplied
      compute((tap / 3) * PW + tap % 3, tap & 1);
    }
  }

  // ---- epilogue straight from the accumulators ----------------------------------------------------------
  // lane l of acc[i][2 h], acc[i][2 h + 1] holds, for pixel 16 (i & 1) + (l & 15) of image row 2 w + (i >> 1), the eight
  // consecutive channels 32 h + 8 (l >> 4) .. + 7: one 16-B vector of the output row
  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  const T* __restrict__ AUX = static_cast<const T*>(d.aux);
  T* __restrict__ AUXS = static_cast<T*>(d.aux_scaled);
  const int act = d.act;
  const bool dot_mode = d.stats && d.stats_mode == O2M_STATS_DOT;
  const bool stream_out = (size_t)d.B * H * W * Co * 2 >= ((size_t)64 << 20);
  const int g = lane >> 4, pl = lane & 15;
  const size_t part = ((size_t)b * tps + tis) * 4 + wave;  // one partial per (tile, wave): 64 pixels of the sample
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int en = 32 * h + 8 * g;
    float esc[8], ebias[8], st[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) { esc[q] = 1.f; ebias[q] = 0.f; }
#pragma unroll
    for (int q = 0; q < 16; ++q) st[q] = 0.f;
    if (d.out_scale) {
      const float* sp = d.out_scale + (size_t)b * Co + en;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { esc[q] = s0[q]; esc[4 + q] = s1[q]; }
    }
    if (d.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + en), b1 = *reinterpret_cast<const f32x4*>(d.bias + en + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ebias[q] = b0[q]; ebias[4 + q] = b1[q]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = ty0 + 2 * wave + (i >> 1), gx = tx0 + 16 * (i & 1) + pl;
      const size_t off = ((size_t)(b * H + gy) * W + gx) * Co + en;
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { o[r] = acc[i][2 * h][r]; o[4 + r] = acc[i][2 * h + 1][r]; }
      if (dot_mode) {
        float xv[8];
        load8x(AUX + off, xv, stream_out);
#pragma unroll
        for (int q = 0; q < 8; ++q) st[q] += o[q] * xv[q];
        if (AUXS) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] *= esc[q];
          store8x(AUXS + off, xv, stream_out);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = o[q] * esc[q] + ebias[q];
      if (d.stats && !dot_mode) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st[q] += o[q]; st[8 + q] += o[q] * o[q]; }
      }
      act_fwd8(o, act);
      if (R) {
        float rv[8];
        load8(R + off, rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += rv[q];
      }
      store8x(Y + off, o, stream_out);
    }
    if (d.stats) {
      // sum over the 16 lanes that share a channel vector (quad swaps, then rotations inside the row of 16): fixed order
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = st[q];
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));  // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));  // row_ror:8
        st[q] = v;
      }
      if (pl == 0) {
        float* sp = d.stats + (part * Co + en) * 2;  // [part][channel][sum | sum of squares] (dot mode: [dot | 0])
#pragma unroll
        for (int q = 0; q < 8; q += 2)
          *reinterpret_cast<f32x4*>(sp + 2 * q) = f32x4{st[q], st[8 + q], st[q + 1], st[8 + q + 1]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The four-wave form for 64 output channels per block, KS x KS taps (3 or 4), any zero padding < KS and ANY map size
// (clipped tiles; blockIdx.y = the 64-channel block): replaces conv3x3_halo_kernel<64> and conv_halo_any_kernel<KS> (8-wave
// blocks, wave tile 64 pixels x 32 channels: 0.75 fragment reads per MFMA, fp32 tile through LDS in four passes).  Wave w
// owns image rows 2 w, 2 w + 1 of the tile x all 64 channels (64 accumulator VGPRs, 0.5 reads per MFMA); filter rows
// permuted in LDS as in conv3x3_halo_w4_kernel, epilogue straight from the accumulators with stores and statistics
// masked at the clipped edge.  59 / 65 KB of LDS (3 x 3 / 4 x 4): two blocks per CU.
template <int KS>
__global__ __launch_bounds__(256, 2) void conv_halo_w4c_kernel(const o2m_conv_desc d) {
  using T = unsigned short;
  constexpr int CO = 64, TH = 8, TW = 32, PW = TW + KS - 1, NPIX = (TH + KS - 1) * PW;
  constexpr int PFILLS = (NPIX + 7) / 8, PPW = (PFILLS + 3) / 4;  // fills of 8 pixels; fills per wave
  constexpr int PATCH_B = PFILLS * 1024;
  constexpr int WB = CO * 128;  // one filter tap of one chunk: 64 rows x 64 channels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wbuf = smem + PATCH_B;
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, pad = d.pad;
  const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
  const int K = KS * KS * Ci;
  const int n0 = blockIdx.y * CO;
  const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH, tps = tiles_x * tiles_y;
  const int tile = xcd_tile_order(blockIdx.x, gridDim.x);
  const int b = tile / tps, tis = tile - b * tps;
  const int ty0 = (tis / tiles_x) * TH, tx0 = (tis % tiles_x) * TW;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t wr = make_rsrc(static_cast<const char*>(d.w) + (size_t)b * d.w_batch_stride * 2, (unsigned)((size_t)Co * K * 2));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

  unsigned poff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int f = 4 * j + wave;
    const int pp = 8 * f + (lane >> 3);
    const int c = (lane & 7) ^ ((pp >> 1) & 7);
    const int py = pp / PW, px = pp - py * PW;
    const int gy = ty0 + py - pad, gx = tx0 + px - pad;
    const bool ok = f < PFILLS && pp < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[j] = ok ? (unsigned)(((b * H + gy) * W + gx) * Ci + c * 8) * 2u : OOB_OFF;
  }
  auto issue_patch = [&](int cb) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int f = 4 * j + wave;
      if (f >= PFILLS) continue;  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(patch + f * 1024), 16, (int)poff[j], cb * 2, 0, 0);
    }
  };
  unsigned woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = 2 * wave + j;
    const int pr = 4 * q + (lane >> 4);
    const int rho = 2 * pr + (((lane & 15) ^ (pr & 15)) >> 3), chk = ((lane & 15) ^ (pr & 15)) & 7;
    const int n = n0 + 32 * (rho >> 5) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3);
    woff[j] = (unsigned)(n * K + chk * 8) * 2u;
  }
  auto issue_w = [&](int tap, int cb, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(wbuf + buf * WB + (2 * wave + j) * 1024), 16, (int)woff[j],
                                               (tap * Ci + cb) * 2, 0, 0);
  };

  const int c0 = lane >> 4;
  const int ppb0 = 2 * wave * PW + (lane & 15);
  const int fb0 = tile_off(lane & 15, c0);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const int toff, const int buf) {
    int aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = ppb0 + (toff + (i >> 1) * PW + (i & 1) * 16);
      aoff[i] = (pp << 7) | ((c0 ^ ((pp >> 1) & 7)) << 4);
    }
    const char* wb = wbuf + buf * WB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 px[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) px[i] = *reinterpret_cast<const bf16x8*>(patch + (aoff[i] ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wb + ((fb0 ^ (((j & 1) << 7) | (ks << 6))) + j * 2048));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], px[i], acc[i][j], 0, 0, 0);
    }
  };

  for (int cb = 0; cb < Ci; cb += 64) {
    if (cb) __syncthreads();  // every wave is done with the previous chunk's patch and filter buffers
    issue_patch(cb);
    issue_w(0, cb, 0);
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // this tap's filter (at tap 0: the patch) has landed for every wave; all are done with tap - 1
      if (tap + 1 < KS * KS) issue_w(tap + 1, cb, (tap + 1) & 1);
      compute((tap / KS) * PW + tap % KS, tap & 1);
    }
  }

  // ---- epilogue straight from the accumulators (see conv3x3_halo_w4_kernel), masked at the clipped edge -----
  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  const T* __restrict__ AUX = static_cast<const T*>(d.aux);
  T* __restrict__ AUXS = static_cast<T*>(d.aux_scaled);
  const int act = d.act;
  const bool dot_mode = d.stats && d.stats_mode == O2M_STATS_DOT;
  const bool stream_out = (size_t)d.B * Ho * Wo * Co * 2 >= ((size_t)64 << 20);
  const int g = lane >> 4, pl = lane & 15;
  const size_t part = ((size_t)b * tps + tis) * 4 + wave;  // one partial per (tile, wave): whatever of its 64 pixels exist
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int en = n0 + 32 * h + 8 * g;
    float esc[8], ebias[8], st[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) { esc[q] = 1.f; ebias[q] = 0.f; }
#pragma unroll
    for (int q = 0; q < 16; ++q) st[q] = 0.f;
    if (d.out_scale) {
      const float* sp = d.out_scale + (size_t)b * Co + en;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { esc[q] = s0[q]; esc[4 + q] = s1[q]; }
    }
    if (d.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + en), b1 = *reinterpret_cast<const f32x4*>(d.bias + en + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ebias[q] = b0[q]; ebias[4 + q] = b1[q]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = ty0 + 2 * wave + (i >> 1), gx = tx0 + 16 * (i & 1) + pl;
      if (gy >= Ho || gx >= Wo) continue;  // clipped tile: nothing to store, nothing to count
      const size_t off = ((size_t)(b * Ho + gy) * Wo + gx) * Co + en;
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { o[r] = acc[i][2 * h][r]; o[4 + r] = acc[i][2 * h + 1][r]; }
      if (dot_mode) {
        float xv[8];
        load8x(AUX + off, xv, stream_out);
#pragma unroll
        for (int q = 0; q < 8; ++q) st[q] += o[q] * xv[q];
        if (AUXS) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] *= esc[q];
          store8x(AUXS + off, xv, stream_out);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = o[q] * esc[q] + ebias[q];
      if (d.stats && !dot_mode) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st[q] += o[q]; st[8 + q] += o[q] * o[q]; }
      }
      act_fwd8(o, act);
      if (R) {
        float rv[8];
        load8(R + off, rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += rv[q];
      }
      store8x(Y + off, o, stream_out);
    }
    if (d.stats) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = st[q];
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));  // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));  // row_ror:8
        st[q] = v;
      }
      if (pl == 0) {
        float* sp = d.stats + (part * Co + en) * 2;
#pragma unroll
        for (int q = 0; q < 8; q += 2)
          *reinterpret_cast<f32x4*>(sp + 2 * q) = f32x4{st[q], st[8 + q], st[q + 1], st[8 + q + 1]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same scheme for KS x KS taps (3 or 4), any zero padding < KS, ANY map size (tiles clipped at the edge: loads outside
// the image are out-of-range DMA offsets, stores and statistics are masked) and any Co % 64 == 0 (64 output channels per
// block, blockIdx.y = the channel block): the 4 x 4 trunk of the discriminator / style extractor on its odd-sized maps
// (255 -> 127 -> 126 -> 63 -> 62 -> 31 -> 30, builder.py:269-283) and its data gradients (pad 2), which ran on the generic
// 256 x 64 / 128 x 128 tiles at 0.2-0.3 of peak and -- Ho * Wo not being a multiple of any row block -- with a separate
// InstanceNorm statistics pass.  Here the partial of (tile, wave row) covers whatever pixels of the tile exist:
// o2m_conv2d_stats_chunks = 4 x tiles per sample rows per sample, summed by o2m_instnorm_finalize as before.
template <int KS>
__global__ __launch_bounds__(512, 4) void conv_halo_any_kernel(const o2m_conv_desc d) {
  using T = unsigned short;
  constexpr int CO = 64;
  constexpr int NT = 512, TH = 8, TW = 32, PW = TW + KS - 1, NPIX = (TH + KS - 1) * PW;  // patch pixels
  constexpr int PFILLS = (NPIX + 7) / 8;                                       // fills of 8 pixels
  constexpr int PATCH_B = PFILLS * 1024;
  constexpr int WB = CO * 128;     // one filter tap of one chunk: CO rows x 64 channels
  constexpr int WFILLS = CO / 8;   // 8 rows per fill
  constexpr int WPW = WFILLS / 8;  // filter fills per wave (1 or 2)
  constexpr int NJ = CO / 32;      // 16-column MFMA tiles per wave (wave N = CO / 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wbuf = smem + PATCH_B;
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co, pad = d.pad;
  const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
  const int K = KS * KS * Ci;
  const int n0 = blockIdx.y * CO;  // this block's 64 output channels
  const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH, tps = tiles_x * tiles_y;
  const int tile = xcd_tile_order(blockIdx.x, gridDim.x);
  const int b = tile / tps, tis = tile - b * tps;
  const int ty0 = (tis / tiles_x) * TH, tx0 = (tis % tiles_x) * TW;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t wr = make_rsrc(static_cast<const char*>(d.w) + (size_t)b * d.w_batch_stride * 2, (unsigned)((size_t)Co * K * 2));

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- patch fills: fill f = 8 j + wave covers patch pixels 8 f .. 8 f + 7; lane l owns slot (l & 7) of pixel
  // 8 f + (l >> 3).  The six source offsets are recomputed per chunk (once or twice per block) rather than kept
  // in registers across the tap loop: the kernel must fit 128 VGPRs for two blocks per CU.
  auto issue_patch = [&](int cb) {
    int ln = lane;
    asm volatile("" : "+v"(ln));  // keeps the offset arithmetic below inside the chunk loop
#pragma unroll
    for (int j = 0; j < (PFILLS + 7) / 8; ++j) {
      const int f = 8 * j + wave;
      if (f >= PFILLS) continue;  // wave-uniform
      const int pp = 8 * f + (ln >> 3);
      const int c = (ln & 7) ^ ((pp >> 1) & 7);
      const int py = pp / PW, px = pp - py * PW;
      const int gy = ty0 + py - pad, gx = tx0 + px - pad;
      const bool ok = pp < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((b * H + gy) * W + gx) * Ci + c * 8) * 2u : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(patch + f * 1024), 16, (int)off, cb * 2, 0, 0);
    }
  };
  // ---- filter fills: tile_off's swizzle, lane l of the fill of 8-row group q owns linear slot 64 q + l
  unsigned woff[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int q = WPW * wave + j;
    const int pr = 4 * q + (lane >> 4);
    const int row = 2 * pr + (((lane & 15) ^ (pr & 15)) >> 3), chk = ((lane & 15) ^ (pr & 15)) & 7;
    woff[j] = (unsigned)((n0 + row) * K + chk * 8) * 2u;
  }
  auto issue_w = [&](int tap, int cb, int buf) {
#pragma unroll
    for (int j = 0; j < WPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(wbuf + buf * WB + (WPW * wave + j) * 1024), 16, (int)woff[j],
                                               (tap * Ci + cb) * 2, 0, 0);
  };

  // ---- fragments -------------------------------------------------------------------------------------
  // A: lane l holds pixel row (l & 15) of a 16-row tile and reduction elements 8 (l >> 4) + 32 ks .. + 7
  const int c0 = lane >> 4;
  // patch pixel of this lane's row at tap (0, 0) in 16-row tile 0; tile i adds (i >> 1) image rows and 16 (i & 1) pixels
  const int ppb0 = 2 * wm * PW + (lane & 15);
  // tile_off(R0 + 16 j + r, c) = (tile_off(R0 + r, c) ^ ((j & 1) << 7)) + j * 2048 for R0 % 32 == 0 (as in the p8 kernel)
  const int fb0 = tile_off(wn * (CO / 2) + (lane & 15), c0);

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const int toff, const int buf) {
    int aoff[4];
    int pb = ppb0;
    asm volatile("" : "+v"(pb));  // recompute the four offsets per tap: hoisted, the 36 of them spill to scratch
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = pb + (toff + (i >> 1) * PW + (i & 1) * 16);
      aoff[i] = (pp << 7) | ((c0 ^ ((pp >> 1) & 7)) << 4);
    }
    const char* wb = wbuf + buf * WB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], bw[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(patch + (aoff[i] ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bw[j] = *reinterpret_cast<const bf16x8*>(wb + ((fb0 ^ (((j & 1) << 7) | (ks << 6))) + j * 2048));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bw[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: per 64-channel chunk, the patch once and the nine taps from it ------------------------
  for (int cb = 0; cb < Ci; cb += 64) {
    issue_patch(cb);
    issue_w(0, cb, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      if (tap + 1 < KS * KS) issue_w(tap + 1, cb, (tap + 1) & 1);  // lands while this tap is multiplied
      compute((tap / KS) * PW + tap % KS, tap & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // next tap's filter has landed; every wave is done with this tap's (and, at the last tap, the patch)
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------------
  constexpr int CSTR = CO + 4;
  constexpr int VPR = CO / 8, RPI = NT / VPR;  // 8-channel vectors per row; rows per read-out iteration
  float* csm = reinterpret_cast<float*>(smem);
  T* __restrict__ Y = static_cast<T*>(d.y);
  const T* __restrict__ R = static_cast<const T*>(d.residual);
  const T* __restrict__ AUX = static_cast<const T*>(d.aux);
  T* __restrict__ AUXS = static_cast<T*>(d.aux_scaled);
  const bool dot_mode = d.stats && d.stats_mode == O2M_STATS_DOT;
  const int act = d.act;
  const bool stream_out = (size_t)d.B * Ho * Wo * Co * 2 >= ((size_t)64 << 20);
  const int ec8 = tid % VPR, erow = tid / VPR, en = n0 + ec8 * 8;
  float esc[8], ebias[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { esc[q] = 1.f; ebias[q] = 0.f; }
  if (d.out_scale) {
    const float* sp = d.out_scale + (size_t)b * Co + en;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { esc[q] = s0[q]; esc[4 + q] = s1[q]; }
  }
  if (d.bias) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + en), b1 = *reinterpret_cast<const f32x4*>(d.bias + en + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { ebias[q] = b0[q]; ebias[4 + q] = b1[q]; }
  }
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {  // wave row `pass`: tile image rows 2 pass, 2 pass + 1
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            csm[(i * 16 + 4 * (lane >> 4) + r) * CSTR + wn * (CO / 2) + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    lds_barrier();
    float st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) st[q] = 0.f;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int row = erow + it * RPI;  // 0 .. 63
      const int gy = ty0 + 2 * pass + (row >> 5), gx = tx0 + (row & 31);
      if (gy >= Ho || gx >= Wo) continue;  // clipped tile: nothing to store, nothing to count
      const size_t off = ((size_t)(b * Ho + gy) * Wo + gx) * Co + en;
      const f32x4 va = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec8 * 8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(csm + row * CSTR + ec8 * 8 + 4);
      float o[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      if (dot_mode) {  // style-dot partials sum acc * aux (and aux * out_scale on its way past): see conv3x3_halo_kernel
        float xv[8];
        load8x(AUX + off, xv, stream_out);
#pragma unroll
        for (int q = 0; q < 8; ++q) st[q] += o[q] * xv[q];
        if (AUXS) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] *= esc[q];
          store8x(AUXS + off, xv, stream_out);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = o[q] * esc[q] + ebias[q];
      if (d.stats && !dot_mode) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st[q] += o[q]; st[8 + q] += o[q] * o[q]; }
      }
      act_fwd8(o, act);
      if (R) {
        float rv[8];
        load8(R + off, rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += rv[q];
      }
      store8x(Y + off, o, stream_out);
    }
    // one partial per (tile, wave row): 64 pixels of the sample; o2m_instnorm_finalize / o2m_conv2d_dots_finalize
    // add a sample's H W / 64 partials whatever pixels each covers
    if (d.stats) stats_block_reduce<NT, VPR>(st, csm, d.stats, ((long)b * tps + tis) * 4 + pass, n0, Co, tid);
    lds_barrier();
  }
}

template <int CO>
int launch_halo(const o2m_conv_desc& d, hipStream_t s) {
  constexpr int lds_main = 43 * 1024 + 2 * CO * 128, lds_epi = 64 * (CO + 4) * 4, lds_red = 16 * 512 * 4;
  constexpr int lds = lds_main > lds_epi ? (lds_main > lds_red ? lds_main : lds_red) : (lds_epi > lds_red ? lds_epi : lds_red);
  const long tiles = (long)d.B * (d.H / 8) * (d.W / 32);
  if (tiles <= 0 || tiles > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<CO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * d.B * d.H * d.W * d.Co * 9.0 * d.Ci, "conv3x3_halo<bf16,8x32x%d>", CO);
    hipLaunchKernelGGL(conv3x3_halo_kernel<CO>, dim3((unsigned)tiles), dim3(512), lds, s, d);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

int launch_halo_w4(const o2m_conv_desc& d, hipStream_t s) {
  constexpr int lds = 43 * 1024 + 2 * 128 * 128;  // 75 KB: two blocks per CU
  const long tiles = (long)d.B * (d.H / 8) * (d.W / 32);
  if (tiles <= 0 || tiles > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * d.B * d.H * d.W * d.Co * 9.0 * d.Ci, "conv3x3_halo<bf16,8x32x%d>", 128);
    hipLaunchKernelGGL(conv3x3_halo_w4_kernel, dim3((unsigned)tiles), dim3(256), lds, s, d);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

// O2M_HALO_W4C (A/B): bit 0 = the 4 x 4 trunk layers, bit 1 = the 3 x 3 Co = 64 layers on the four-wave kernel (default 3);
// 0 = the 8-wave forms (conv_halo_any_kernel, conv3x3_halo_kernel<64>)
inline bool halo_w4c_on(int bit) {
  static const int on = [] { const char* e = getenv("O2M_HALO_W4C"); return e ? atoi(e) : 3; }();
  return (on >> bit) & 1;
}
template <int KS>
int launch_halo_w4c(const o2m_conv_desc& d, hipStream_t s, const char* name) {
  constexpr int npix = (8 + KS - 1) * (32 + KS - 1);
  constexpr int lds = ((npix + 7) / 8) * 1024 + 2 * 64 * 128;
  const int Ho = d.H + 2 * d.pad - KS + 1, Wo = d.W + 2 * d.pad - KS + 1;
  const long tiles = (long)d.B * ((Ho + 7) / 8) * ((Wo + 31) / 32);
  if (tiles <= 0 || tiles > 0x7fffffffL || d.Co % 64 || d.Co / 64 > 65535) return O2M_ERR_BAD_ARG;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_w4c_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * d.B * Ho * Wo * d.Co * (double)(KS * KS) * d.Ci, "%s", name);
    hipLaunchKernelGGL(conv_halo_w4c_kernel<KS>, dim3((unsigned)tiles, (unsigned)(d.Co / 64)), dim3(256), lds, s, d);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

template <int KS>
int launch_halo_any(const o2m_conv_desc& d, hipStream_t s) {
  constexpr int npix = (8 + KS - 1) * (32 + KS - 1);
  constexpr int lds_main = ((npix + 7) / 8) * 1024 + 2 * 64 * 128, lds_red = 16 * 512 * 4;
  constexpr int lds = lds_main > lds_red ? lds_main : lds_red;
  const int Ho = d.H + 2 * d.pad - KS + 1, Wo = d.W + 2 * d.pad - KS + 1;
  const long tiles = (long)d.B * ((Ho + 7) / 8) * ((Wo + 31) / 32);
  if (tiles <= 0 || tiles > 0x7fffffffL || d.Co / 64 > 65535) return O2M_ERR_BAD_ARG;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_any_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * d.B * Ho * Wo * d.Co * (double)(KS * KS) * d.Ci, "conv_halo<bf16,%dx%d,8x32x64>", KS, KS);
    hipLaunchKernelGGL(conv_halo_any_kernel<KS>, dim3((unsigned)tiles, (unsigned)(d.Co / 64)), dim3(512), lds, s, d);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

// the layers the clipped halo-tile kernel takes: the 4 x 4 trunk convolutions and their data gradients (64-channel
// multiples on both sides, zero padding, plain epilogue incl. InstanceNorm partials)
inline bool halo_any_ok(const o2m_conv_desc& d) {
  static const int on = [] { const char* e = getenv("O2M_CONV_HALO4"); return e ? atoi(e) : 1; }();
  if (!on || d.fold_pad || d.dtype != O2M_BF16 || d.KH != 4 || d.KW != 4 || d.pad < 1 || d.pad > 3 || d.pad_mode != O2M_PAD_ZERO ||
      d.stride > 1 || d.in_scale || d.aux || d.aux_scaled || d.Ci % 64 || d.Co % 64 || d.w_batch_stride)
    return false;
  if (d.stats && d.stats_mode != O2M_STATS_MOMENTS) return false;
  const int Ho = d.H + 2 * d.pad - 4 + 1, Wo = d.W + 2 * d.pad - 4 + 1;
  // at least one block per CU (two are resident): the 30 x 30 maps at B = 16 give 64 tiles x 8 channel blocks
  return Ho > 0 && Wo > 0 && (long)d.B * ((Ho + 7) / 8) * ((Wo + 31) / 32) * (d.Co / 64) >= kFillBlocks;
}

// the layers the halo-tile kernel takes (host side of its preconditions)
inline bool halo_ok(const o2m_conv_desc& d) {
  static const int on = [] { const char* e = getenv("O2M_CONV_HALO"); return e ? atoi(e) : 1; }();
  return on && !d.fold_pad && d.dtype == O2M_BF16 && d.KH == 3 && d.KW == 3 && d.pad == 1 && d.pad_mode == O2M_PAD_ZERO && d.stride <= 1 &&
         !d.in_scale && d.Ci % 64 == 0 && (d.Co == 64 || d.Co == 128) && d.W % 32 == 0 && d.H % 8 == 0 &&
         (long)d.B * (d.H / 8) * (d.W / 32) >= 2 * kFillBlocks;
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const o2m_conv_desc& d, hipStream_t s, long m_begin = 0, long m_end = -1) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int lds = lds_bytes<T, BM, BN, WAVES_M, WAVES_N>();
  if (m_end < 0) m_end = out_rows(d);
  const long tiles = tiles_rows<BM, BN>(d, m_end - m_begin);
  if (tiles <= 0 || tiles > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), lds, s, d, (int)m_begin, (int)m_end);
  };
  const bool wide = d.Ci % BK == 0;
  LaunchScope timed(s, 2.0 * (m_end - m_begin) * d.Co * d.KH * d.KW * d.Ci, "conv_igemm<%s,%dx%d,in_scale=%d>",
                    sizeof(T) == 2 ? "bf16" : "f32x3", BM, BN, d.in_scale ? 1 : 0);
  if (d.fold_pad) {  // (validated: bf16, no in_scale; launch_dtype routes every fold launch to the 128 x 128 tile)
    if constexpr (sizeof(T) == 2 && BM == 128 && BN == 128) {
      if (wide) go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false, true, true>);
      else go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false, false, true>);
    } else {
      return O2M_ERR_UNSUPPORTED;
    }
  } else if (d.in_scale) {
    if (wide) go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, true, true>);
    else go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, true, false>);
  } else {
    if (wide) go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false, true>);
    else go(conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false, false>);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}


inline bool f32_big_tiles() {
  static const bool on = [] { const char* e = getenv("O2M_F32_BIG_TILES"); return e && e[0] == '1'; }();
  return on;
}

template <typename T>
int launch_dtype(const o2m_conv_desc& d, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (o2m_direct::stem8_ok(d)) return o2m_direct::launch_stem8(d, s);
    if (o2m_direct::fewout_ok(d)) return o2m_direct::launch_fewout(d, s);
    if (halo_ok(d)) {
      // O2M_HALO128_SPLIT=1 (A/B): the Co = 128 layers as two 64-channel blocks of the clipped kernel (no spills, the
      // patch ingested twice) instead of conv3x3_halo_kernel<128> (25 VGPRs spilled)
      static const int split128 = [] { const char* e = getenv("O2M_HALO128_SPLIT"); return e ? atoi(e) : 0; }();
      if (d.Co == 128 && split128) return launch_halo_any<3>(d, s);
      // O2M_HALO128_W4=0 (A/B): the 8-wave form of the Co = 128 tile (LDS-staged epilogue, 25 VGPRs spilled)
      static const int w4 = [] { const char* e = getenv("O2M_HALO128_W4"); return e ? atoi(e) : 1; }();
      if (d.Co == 128 && w4) return launch_halo_w4(d, s);
      if (d.Co == 64 && halo_w4c_on(1)) return launch_halo_w4c<3>(d, s, "conv3x3_halo<bf16,8x32x64>");
      return d.Co == 64 ? launch_halo<64>(d, s) : launch_halo<128>(d, s);
    }
    if (halo_any_ok(d)) return halo_w4c_on(0) ? launch_halo_w4c<4>(d, s, "conv_halo<bf16,4x4,8x32x64>") : launch_halo_any<4>(d, s);
  }
  // big 8-wave tiles when they still give every CU a block; otherwise the 4-wave 128-wide
  // tiles (small-M layers of the discriminator) so the chip stays filled
  if (d.Co > 128) {
    if constexpr (sizeof(T) == 2) {
      // O2M_IGEMM_P8=0: the symmetric two-stage kernel instead of the phase-pipelined one (A/B runs)
      static const int p8 = [] { const char* e = getenv("O2M_IGEMM_P8"); return e ? atoi(e) : 1; }();
      if (p8 && p8_geometry_ok(d) && tiles_for<256, 256>(d) >= kFillBlocks) {
        // One block per CU and round: a few tiles past a whole number of rounds would cost a whole
        // extra round (the data gradient of the reflect-padded 64x64 layers is 273 tiles = 2 rounds for
        // 1.07 rounds of work).  Such a tail (<= 1/4 round) runs as a second launch of 128x128 tiles
        // over the last rows instead: 256 + 17 tiles take 1 round + ~0.3 instead of 2.
        const long rows = out_rows(d), tn = (d.Co + 255) / 256, tiles = tiles_for<256, 256>(d);
        const long tail = tiles % kFillBlocks;
        static const int split_tail = [] { const char* e = getenv("O2M_IGEMM_TAIL_SPLIT"); return e ? atoi(e) : 1; }();
        if (split_tail && tiles > kFillBlocks && tail > 0 && tail <= kFillBlocks / 4 && (tiles - tail) % tn == 0 &&
            !d.stats) {
          const long m_split = (tiles - tail) / tn * 256;
          const int rc = launch_p8(d, s, 0, m_split);
          if (rc) return rc;
          // (128 x 64 tail tiles measured 102.4 vs 104.4 us for the whole call: not worth a second configuration)
          return launch_cfg<T, 128, 128, 2, 2>(d, s, m_split, rows);
        }
        return launch_p8(d, s, 0, rows);
      }
    }
    if (d.fold_pad) return launch_cfg<T, 128, 128, 2, 2>(d, s);
    if (tiles_for<256, 256>(d) >= kFillBlocks) {
      // fp32 (bf16x3 split) mode: the 256 x 256 tile spills 8-103 VGPRs (up to 272 B of scratch per lane); 256 x 128 fits
      // (176-180 VGPRs).  O2M_F32_BIG_TILES=1 keeps the old tile (A/B).
      if constexpr (sizeof(T) == 4) {
        if (!f32_big_tiles()) return launch_cfg<T, 256, 128, 4, 2>(d, s);
      }
      return launch_cfg<T, 256, 256, 2, 4>(d, s);
    }
    return launch_cfg<T, 128, 128, 2, 2>(d, s);
  }
  if (d.fold_pad) return launch_cfg<T, 128, 128, 2, 2>(d, s);  // (the one tile compiled with the fold epilogue)
  if (d.Co > 64) {
    // short reductions (<= 18 stages) are dominated by per-block prologue / epilogue: 256x64 tiles
    // need 80 KB of LDS and ~110 VGPRs, so TWO blocks share a CU and overlap each other
    if (d.KH * d.KW * d.Ci <= 1152 && tiles_for<256, 64>(d) >= 2 * kFillBlocks)
      return launch_cfg<T, 256, 64, 4, 2>(d, s);
    if (tiles_for<256, 128>(d) >= kFillBlocks) return launch_cfg<T, 256, 128, 4, 2>(d, s);
    return launch_cfg<T, 128, 128, 2, 2>(d, s);
  }
  if (d.Co > 32) {
    if (tiles_for<256, 64>(d) >= kFillBlocks) return launch_cfg<T, 256, 64, 4, 2>(d, s);
    return launch_cfg<T, 128, 64, 2, 2>(d, s);
  }
  return launch_cfg<T, 256, 32, 4, 1>(d, s);
}

}  // namespace

// rows per InstanceNorm partial = the row block one epilogue pass of the selected configuration covers
// (mirrors launch_dtype below)
static int stats_rows_for(const o2m_conv_desc& d) {
  if (d.stride > 1 || d.in_scale) return 0;
  if (d.dtype == O2M_BF16 && o2m_direct::stem8_ok(d)) return o2m_direct::stem8_stats_rows(d);  // whole output rows of a block
  if (d.dtype == O2M_BF16 && halo_ok(d)) return 64;  // one partial per wave row of an 8 x 32 tile
  if (d.Co > 128) {
    if (tiles_for<256, 256>(d) >= kFillBlocks)               // p8 and the symmetric 256x256 kernel alike: 128-row wave rows;
      return (d.dtype == O2M_F32 && !f32_big_tiles()) ? 64 : 128;  // fp32 mode's 256x128 tile (4 wave rows): 64
    return 64;                                               // 128x128, 2x2 waves
  }
  return 64;  // 256x64 / 256x128 (4 wave rows), 128x128 / 128x64 (2 wave rows), 256x32
}

namespace {
// dots[b][c] = sum over the row blocks of a sample of the epilogue's O2M_STATS_DOT partials, in block order
__global__ __launch_bounds__(256) void dots_finalize_kernel(const float* __restrict__ partial, float* __restrict__ dots,
                                                            int BC, int C, int nchunks) {
  // 32 (sample, channel) pairs x 8 slices of the chunk list, slice sums added in slice order (as in_finalize_kernel)
  __shared__ float red[8][32];
  const int pi = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + pi;
  float s = 0.f;
  if (idx < BC) {
    const int b = idx / C, c = idx - b * C;
#pragma unroll 4
    for (int ch = sl; ch < nchunks; ch += 8) s += partial[(((size_t)b * nchunks + ch) * C + c) * 2];
  }
  red[sl][pi] = s;
  __syncthreads();
  if (sl != 0 || idx >= BC) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) s += red[k][pi];
  dots[idx] = s;
}
}  // namespace

extern "C" int o2m_conv2d_dots_finalize(const float* partial, float* dots, int32_t B, int32_t C, int32_t nchunks,
                                        void* stream) {
  if (!partial || !dots || B <= 0 || C <= 0 || nchunks <= 0) return O2M_ERR_BAD_ARG;
  hipLaunchKernelGGL(dots_finalize_kernel, dim3((B * C + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream),
                     partial, dots, B * C, C, nchunks);
  O2M_LAUNCH_CHECK();
  return 0;
}

// partial rows PER SAMPLE the epilogue of the selected kernel writes (0: it cannot emit them)
extern "C" int32_t o2m_conv2d_stats_chunks(const o2m_conv_desc* d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->KH <= 0 || d->KW <= 0 || d->pad < 0) return 0;
  if (d->dtype == O2M_BF16 && !o2m_direct::stem8_ok(*d) && !halo_ok(*d) && halo_any_ok(*d)) {
    const int Ho = d->H + 2 * d->pad - d->KH + 1, Wo = d->W + 2 * d->pad - d->KW + 1;
    return 4 * ((Ho + 7) / 8) * ((Wo + 31) / 32);  // one per (clipped 8 x 32 tile, wave row)
  }
  const int r = o2m_conv2d_stats_rows(d);
  if (r <= 0) return 0;
  const int S = d->stride > 1 ? d->stride : 1;
  const long howo = (long)((d->H + 2 * d->pad - d->KH) / S + 1) * ((d->W + 2 * d->pad - d->KW) / S + 1);
  return (int32_t)(howo / r);
}

extern "C" int32_t o2m_debug_fill_blocks(int32_t n) {
  const long prev = g_fill_blocks;
  g_fill_blocks = n > 0 ? n : kFillBlocksDefault;
  return (int32_t)prev;
}

extern "C" int32_t o2m_conv2d_stats_rows(const o2m_conv_desc* d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->KH <= 0 || d->KW <= 0 || d->pad < 0) return 0;
  const bool f8 = d->dtype == O2M_FP8_E4M3 || d->dtype == O2M_BF8_E5M2;
  if (d->dtype != O2M_BF16 && d->dtype != O2M_F32 && !f8) return 0;
  const int S = d->stride > 1 ? d->stride : 1;
  const long howo = (long)((d->H + 2 * d->pad - d->KH) / S + 1) * ((d->W + 2 * d->pad - d->KW) / S + 1);
  const int r = f8 ? (d->stride > 1 || d->in_scale ? 0 : 128) : stats_rows_for(*d);  // fp8: always the p8 kernel
  return (r > 0 && howo > 0 && howo % r == 0) ? r : 0;
}

namespace {
// fold_pad: the pixels of the cropped map that take more than one contribution (rows / columns 1..f and Hc-1-f..Hc-2)
// start from zero; one thread per (strip pixel, 16-B channel vector).  Strip pixels are enumerated as 2f whole rows +
// 2f columns of the remaining rows (the overlap is written twice with the same zero).
__global__ __launch_bounds__(256) void fold_zero_kernel(unsigned short* __restrict__ y, int B, int Hc, int Wc, int C, int f) {
  const int CV = C / 8;
  const long per_img = (long)2 * f * Wc + (long)2 * f * Hc;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)B * per_img * CV) return;
  const int cv = (int)(t % CV);
  const long pi = t / CV;
  const int b = (int)(pi / per_img);
  int k = (int)(pi - (long)b * per_img), py, px;
  if (k < 2 * f * Wc) {
    const int r = k / Wc;
    px = k - r * Wc;
    py = r < f ? 1 + r : Hc - 1 - f + (r - f);
  } else {
    k -= 2 * f * Wc;
    const int c = k / Hc;
    py = k - c * Hc;
    px = c < f ? 1 + c : Wc - 1 - f + (c - f);
  }
  *reinterpret_cast<u32x4*>(y + (((size_t)b * Hc + py) * Wc + px) * C + cv * 8) = u32x4{0u, 0u, 0u, 0u};
}
}  // namespace

extern "C" int o2m_conv2d_fwd(const o2m_conv_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y) return O2M_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ci <= 0 || d->Co <= 0) return O2M_ERR_BAD_ARG;
  if ((d->Ci & 7) || (d->Co & 7) || d->KH <= 0 || d->KW <= 0 || d->pad < 0) return O2M_ERR_BAD_ARG;
  if (d->H + 2 * d->pad < d->KH || d->W + 2 * d->pad < d->KW) return O2M_ERR_BAD_ARG;
  if (d->pad_mode == O2M_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W)) return O2M_ERR_BAD_ARG;
  if (d->pad_mode != O2M_PAD_ZERO && d->pad_mode != O2M_PAD_REFLECT) return O2M_ERR_BAD_ARG;
  const bool f8 = d->dtype == O2M_FP8_E4M3 || d->dtype == O2M_BF8_E5M2;
  const long esz = d->dtype == O2M_F32 ? 4 : (f8 ? 1 : 2);
  if (d->w_batch_stride < 0) return O2M_ERR_BAD_ARG;
  // Buffer descriptors address < 2 GiB per tensor (offset 0x80000000 marks "out of range"): a larger input (fp32
  // parity mode at BASELINE config #4: 16 x 512 x 512 x 128 floats = 2 GiB) is run as several launches over
  // slices of the batch -- every sample's rows are independent, so the result is that of one launch.
  const long x_sample = (long)d->H * d->W * (long)d->Ci * esz;
  if (x_sample > 0x7fffffffL) return O2M_ERR_UNSUPPORTED;
  if (d->fold_pad < 0 || d->reserved1 != 0) return O2M_ERR_BAD_ARG;
  if (d->fold_pad > 0) {  // data gradient of a reflection-padded conv, folded in the epilogue (header)
    const int f = d->fold_pad, Ho = d->H + 2 * d->pad - d->KH + 1, Wo = d->W + 2 * d->pad - d->KW + 1;
    if (d->dtype != O2M_BF16 || d->act != O2M_ACT_NONE || d->stats || d->aux || d->stride > 1 || d->w_batch_stride > 0 ||
        d->in_scale)
      return O2M_ERR_BAD_ARG;
    if (Ho - 2 * f < 2 * f + 2 || Wo - 2 * f < 2 * f + 2) return O2M_ERR_BAD_ARG;
    if ((long)d->B * x_sample > 0x7fffffffL) return O2M_ERR_UNSUPPORTED;  // (no batch slicing in this form)
    const int Hc = Ho - 2 * f, Wc = Wo - 2 * f;
    const long n = (long)d->B * (2L * f * Wc + 2L * f * Hc) * (d->Co / 8);
    hipLaunchKernelGGL(fold_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<unsigned short*>(d->y), d->B, Hc, Wc, d->Co, f);
    O2M_LAUNCH_CHECK();
  }
  if ((long)d->B * x_sample > 0x7fffffffL) {
    const int S_ = d->stride > 1 ? d->stride : 1;
    const long howo = (long)((d->H + 2 * d->pad - d->KH) / S_ + 1) * ((d->W + 2 * d->pad - d->KW) / S_ + 1);
    const long ysz = d->dtype == O2M_F32 ? 4 : 2;  // (fp8 operands produce bf16)
    const int per = (int)(0x7fffffffL / x_sample);
    const int rows = d->stats ? o2m_conv2d_stats_rows(d) : 0;
    for (int b0 = 0; b0 < d->B; b0 += per) {
      o2m_conv_desc c = *d;
      c.B = d->B - b0 < per ? d->B - b0 : per;
      c.x = static_cast<const char*>(d->x) + (size_t)b0 * x_sample;
      c.y = static_cast<char*>(d->y) + (size_t)b0 * howo * d->Co * ysz;
      if (d->residual) c.residual = static_cast<const char*>(d->residual) + (size_t)b0 * howo * d->Co * ysz;
      if (d->aux) c.aux = static_cast<const char*>(d->aux) + (size_t)b0 * howo * d->Co * ysz;
      if (d->aux_scaled) c.aux_scaled = static_cast<char*>(d->aux_scaled) + (size_t)b0 * howo * d->Co * ysz;
      if (d->in_scale) c.in_scale = d->in_scale + (size_t)b0 * d->Ci;
      if (d->out_scale) c.out_scale = d->out_scale + (size_t)b0 * d->Co;
      if (d->w_batch_stride > 0) c.w = static_cast<const char*>(d->w) + (size_t)b0 * d->w_batch_stride * esz;
      if (d->stats && rows > 0) c.stats = d->stats + (size_t)b0 * (howo / rows) * d->Co * 2;
      const int rc = o2m_conv2d_fwd(&c, stream);
      if (rc) return rc;
    }
    return 0;
  }
  if ((long)d->Co * d->KH * d->KW * (long)d->Ci * esz > 0x7fffffffL) return O2M_ERR_UNSUPPORTED;
  if (d->stride < 0 || d->stride > 16) return O2M_ERR_BAD_ARG;
  if (d->w_batch_stride > 0) {
    if (d->stride > 1) return O2M_ERR_BAD_ARG;
    const long howo = (long)(d->H + 2 * d->pad - d->KH + 1) * (d->W + 2 * d->pad - d->KW + 1);
    if (howo % 256 != 0 || d->in_scale) return O2M_ERR_BAD_ARG;
  }
  if (d->stats_mode != O2M_STATS_MOMENTS && d->stats_mode != O2M_STATS_DOT) return O2M_ERR_BAD_ARG;
  if (d->stats && d->stats_mode == O2M_STATS_MOMENTS &&
      (d->act != O2M_ACT_NONE || d->residual || d->out_scale || o2m_conv2d_stats_chunks(d) == 0))
    return O2M_ERR_BAD_ARG;
  if (d->stats && d->stats_mode == O2M_STATS_DOT &&
      (!d->aux || d->act != O2M_ACT_NONE || d->residual || d->bias || o2m_conv2d_stats_rows(d) == 0))
    return O2M_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (f8) {
    // fp8 operands run on the phase-pipelined kernel only: a K-tile is 128 elements of one filter tap
    if (d->in_scale || d->stride > 1 || !d->deq_scale) return O2M_ERR_BAD_ARG;
    if (d->Ci % 128 != 0 || !p8_geometry_ok(*d)) return O2M_ERR_UNSUPPORTED;
    return launch_p8(*d, s, 0, out_rows(*d));
  }
  if (d->dtype == O2M_BF16) return launch_dtype<unsigned short>(*d, s);
  if (d->dtype == O2M_F32) return launch_dtype<float>(*d, s);
  return O2M_ERR_BAD_ARG;
}
