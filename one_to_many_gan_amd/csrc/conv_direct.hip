// Direct (LDS-free operand path) convolution kernels for the layers whose GEMM view is too thin for the tiled
// implicit-GEMM kernels of conv_igemm.hip -- they are bound by HBM / L1 traffic, not by the matrix pipe:
//
//   * conv_stem8_kernel   Ci = 8 (the 3- or 1-channel image padded to one 16-B vector per pixel), Co = 64:
//                         the 4 x 4 stem of the discriminator / style extractor (builder.py:269,300), the 7 x 7
//                         stem of the generator's encoder behind ReflectionPad2d(3) (builder.py:162-165) and the
//                         data gradient of the generator's 7 x 7 image head (builder.py:202-204).  K = KH*KW*8 is
//                         128 / 392: on the 256 x 64 igemm tile a block ran 2-7 K-stages between a prologue and an
//                         epilogue through LDS and reached 0.56-1.15 TB/s of output (profiles/r03_conv_shapes.txt).
//   * conv_fewout_kernel  Co = 8 (<= 8 real outputs), Ci % 64 == 0: the data gradient of that 4 x 4 stem (64 -> 3)
//                         and the discriminator's 512 -> 1 head (builder.py:284).  N = 8 filled 1/4 of the narrowest
//                         igemm tile (15-96 TFLOP/s).
//
// Both use v_mfma_f32_16x16x32_bf16 with the FILTER as the A operand (rows = output channels) and PIXELS as the B
// operand (columns = 16 consecutive pixels of one output row).  stem8: a lane's B fragment -- the 8 channels of one
// pixel under one filter tap, four taps per k-step -- is ONE 16-byte global load, so the activation never passes
// through LDS, needs no im2col and no staging pass; neighbouring taps re-read the same lines from L1.  fewout: the
// pixels come from a halo patch in LDS (see there).  The result tile D[co][pixel] leaves each lane with consecutive
// channels of ONE pixel (stem8: the filter rows are permuted for that), i.e. whole 16-B (stem8) / 8-B (fewout)
// channel vectors per lane: no LDS transpose in the epilogue.
#include <type_traits>
#include <utility>

#include "common.h"
#include "conv_direct.h"

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned OOB_OFF = 0x80000000u;  // tensors are < 2 GiB (checked on the host): this offset reads zeros
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// ---------------------------------------------------------------------------------------------------------
// stem8: y[b, oy, ox, 0..63] = act(sum_{ky,kx,c<8} w[co][ky][kx][c] * xpad[b, oy+ky-pad, ox+kx-pad, c] + bias)
// A block owns `rows` consecutive output rows of one sample; each of its 4 waves takes 64-pixel strips of those rows
// (4 pixel tiles of 16 = the MFMA's columns).  k-step (ky, half) = the four taps kx = 4 half .. 4 half + 3 of filter
// row ky (taps past KS multiply zero filter entries and read zeros).  The filter sits in LDS in fragment order (one
// linear ds_read_b128 per 16-row tile and k-step, conflict-free), rows permuted so that lane group g = lane >> 4 ends
// up with channels 16 g .. 16 g + 15 of its pixel.
// STATS: the InstanceNorm partial sums of the block's rows (o2m_conv_desc.stats; one partial per block).
// ---------------------------------------------------------------------------------------------------------
template <int KS, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_stem8_kernel(const o2m_conv_desc d, const int rows) {
  constexpr int NH = (KS + 3) / 4, NSTEP = KS * NH, PT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* wl = reinterpret_cast<u32x4*>(smem);                      // [NSTEP][4 tiles][64 lanes]
  float* red = reinterpret_cast<float*>(smem + NSTEP * 4 * 1024);  // [4 waves][64 ch][2]

  const int H = d.H, W = d.W, pad = d.pad;
  const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
  const bool reflect = d.pad_mode == O2M_PAD_REFLECT;
  const int nrb = (Ho + rows - 1) / rows;
  const int b = blockIdx.x / nrb, rb = blockIdx.x - b * nrb;
  const int y0 = rb * rows, nrows = min(rows, Ho - y0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  // ---- filter -> LDS, fragment order ------------------------------------------------------------------
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(d.w);  // [64][KS][KS] vectors of 8 channels
    for (int idx = tid; idx < NSTEP * 256; idx += 256) {
      const int l = idx & 63, t = (idx >> 6) & 3, step = idx >> 8;
      const int ky = step / NH, kx = 4 * (step - ky * NH) + (l >> 4);
      const int i = l & 15, co = 16 * (i >> 2) + 4 * t + (i & 3);
      wl[idx] = kx < KS ? wsrc[(co * KS + ky) * KS + kx] : u32x4{0u, 0u, 0u, 0u};
    }
  }
  float bias[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bias[q] = d.bias ? d.bias[16 * g + q] : 0.f;
  float ssum[STATS ? 16 : 1], ssq[STATS ? 16 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int q = 0; q < 16; ++q) ssum[q] = ssq[q] = 0.f;
  }
  __syncthreads();

  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * 16));
  unsigned short* __restrict__ Y = static_cast<unsigned short*>(d.y);
  const int act = d.act;
  const int strips = (Wo + 63) >> 6;
  const int ntasks = nrows * strips;

#pragma unroll 1
  for (int task = wave; task < ntasks; task += 4) {
    const int r = task / strips, x0 = (task - r * strips) << 6;
    const int oy = y0 + r;
    // per-lane byte offset of the pixel under tap column 4 half + g, inside an input row
    unsigned voff[PT][NH];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        const int px = x0 + 16 * pt + li, kx = 4 * hf + g;
        int c = px + kx - pad;
        bool ok = px < Wo && kx < KS;
        if (reflect) c = c < 0 ? -c : (c >= W ? 2 * W - 2 - c : c);
        else ok = ok && (unsigned)c < (unsigned)W;
        voff[pt][hf] = ok ? (unsigned)c * 16u : OOB_OFF;
      }
    f32x4 acc[PT][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // two k-steps per iteration (NSTEP is even: 4 or 14), their pixel fragments double-buffered one step ahead; the
    // loop is NOT unrolled further: fully unrolled the compiler hoists all 56 filter-fragment reads (256 VGPRs + 180
    // AGPRs, one wave per SIMD)
    u32x4 bq0[PT], bq1[PT];
    auto load_step = [&](int ky, int hf, u32x4 (&dst)[PT]) {  // (hf: a literal at every call site)
      int iy = oy + ky - pad;
      bool row_ok = ky < KS;  // wave-uniform (ky == KS: the prefetch past the last step)
      if (reflect) iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
      else row_ok = row_ok && (unsigned)iy < (unsigned)H;
      const unsigned soff = row_ok ? (unsigned)((b * H + iy) * W) * 16u : 0u;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        dst[pt] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(row_ok ? voff[pt][hf] : OOB_OFF),
                                                        (int)__builtin_amdgcn_readfirstlane(soff), 0);
    };
    auto multiply = [&](const u32x4* wf, const u32x4 (&bq)[PT]) {
      bf16x8 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = as_frag(wf[t * 64]);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[pt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], as_frag(bq[pt]), acc[pt][t], 0, 0, 0);
    };
    constexpr int KYB = NH == 2 ? 0 : 1, HFB = NH == 2 ? 1 : 0;  // the second step of a pair: (ky + KYB, HFB)
    constexpr int KYN = NH == 2 ? 1 : 2;                          // filter rows per pair
    load_step(0, 0, bq0);
#pragma unroll 1
    for (int ky = 0; ky < KS; ky += KYN) {
      const u32x4* wf = wl + (ky * NH) * 256 + lane;
      load_step(ky + KYB, HFB, bq1);
      multiply(wf, bq0);
      load_step(ky + KYN, 0, bq0);  // (past the end: zero fills nobody multiplies)
      multiply(wf + 256, bq1);
    }

    // ---- epilogue: lane (g, li) holds channels 16 g .. 16 g + 15 of pixel x0 + 16 pt + li ------------------
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int px = x0 + 16 * pt + li;
      if (px < Wo) {
        float o[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) o[4 * t + q] = acc[pt][t][q] + bias[4 * t + q];
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 16; ++q) { ssum[q] += o[q]; ssq[q] += o[q] * o[q]; }
        }
        float lo[8], hi[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { lo[q] = o[q]; hi[q] = o[8 + q]; }
        act_fwd8(lo, act);
        act_fwd8(hi, act);
        unsigned short* dst = Y + ((size_t)(b * Ho + oy) * Wo + px) * 64 + 16 * g;
        store8(dst, lo);
        store8(dst + 8, hi);
      }
    }
  }

  if constexpr (STATS) {
    // the 16 lanes of a group hold different pixels of the same 16 channels: butterfly over them, lane li == 0 reports
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        ssum[q] += __shfl_xor(ssum[q], m, 64);
        ssq[q] += __shfl_xor(ssq[q], m, 64);
      }
    }
    if (li == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        red[(wave * 64 + 16 * g + q) * 2] = ssum[q];
        red[(wave * 64 + 16 * g + q) * 2 + 1] = ssq[q];
      }
    }
    __syncthreads();
    if (tid < 128) {  // (channel, moment): the four waves in wave order -- one fixed summation order
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) s += red[wv * 128 + tid];
      d.stats[(size_t)blockIdx.x * 128 + tid] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// fewout: y[b, oy, ox, 0..7] = act(sum_{ky,kx,c} w[co][ky][kx][c] * xpad0[b, oy+ky-pad, ox+kx-pad, c] + bias), zero
// padding, Ci % 64 == 0.  A first form took both operands straight from global memory (one 16-B load per lane, tap and
// 32 channels, a wave per filter row): every input element crossed L1 sixteen times and the 64 -> 3 layer at 256 x 256
// ran at 262 us (B = 16), slower than the igemm tile it replaced.  This one is the halo-tile scheme of
// conv3x3_halo_kernel with the roles of the MFMA operands swapped: a block owns an 8 x 32 tile of output pixels (clipped
// at the map's edge) and keeps the (8 + KS - 1) x (32 + KS - 1) input patch of a 64-channel chunk in LDS (LDS-DMA,
// swizzle on the source side, out-of-image pixels = hardware zero fill) together with ALL KS x KS taps of the 8 filter
// rows for that chunk (16 KB); the pixel fragments of every tap are ds_read_b128 at shifted patch addresses, so an input
// element is ingested once per tile (x the halo).  8 waves x 2 pixel tiles of 16; the 16 rows of the A tile are the 8
// filter rows twice (rows 8..15 are never stored).  Two blocks per CU (65 KB of LDS each).
// ---------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(512, 4) void conv_fewout_kernel(const o2m_conv_desc d) {
  constexpr int TH = 8, TW = 32, PWD = TW + KS - 1, NPIX = (TH + KS - 1) * PWD, PFILLS = (NPIX + 7) / 8;
  constexpr int PATCH_B = PFILLS * 1024, NTAP = KS * KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wbuf = smem + PATCH_B;  // [tap][8 filter rows][128 B]
  typedef __attribute__((address_space(3))) void lds_void;

  const int H = d.H, W = d.W, Ci = d.Ci, pad = d.pad;
  const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
  const int tiles_x = (Wo + TW - 1) / TW, tps = tiles_x * ((Ho + TH - 1) / TH);
  const int b = blockIdx.x / tps, tis = blockIdx.x - b * tps;
  const int ty0 = (tis / tiles_x) * TH, tx0 = (tis % tiles_x) * TW;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t wr = make_rsrc(d.w, (unsigned)((size_t)8 * NTAP * Ci * 2));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  // ---- fills (per 64-channel chunk): lane l owns slot (l & 7) of patch pixel / filter row 8 f + (l >> 3) -------------
  auto issue_chunk = [&](int cb) {
    int ln = lane;
    asm volatile("" : "+v"(ln));  // offsets recomputed per chunk rather than kept across the tap loop
#pragma unroll
    for (int j = 0; j < (PFILLS + 7) / 8; ++j) {
      const int f = 8 * j + wave;
      if (f >= PFILLS) continue;  // wave-uniform
      const int pp = 8 * f + (ln >> 3);
      const int c = (ln & 7) ^ ((pp >> 1) & 7);
      const int py = pp / PWD, px = pp - py * PWD;
      const int gy = ty0 + py - pad, gx = tx0 + px - pad;
      const bool ok = pp < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((b * H + gy) * W + gx) * Ci + c * 8) * 2u : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(patch + f * 1024), 16, (int)off, cb * 2, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < (NTAP + 7) / 8; ++j) {
      const int tap = 8 * j + wave;
      if (tap >= NTAP) continue;
      const int co = ln >> 3, c = (ln & 7) ^ (co & 6);
      const unsigned off = (unsigned)((co * NTAP + tap) * Ci + c * 8) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(wbuf + tap * 1024), 16, (int)off, cb * 2, 0, 0);
    }
  };

  // ---- fragments -----------------------------------------------------------------------------------------------
  // A (filter): row li & 7, chunk g + 4 ks at slot chunk ^ (row & 6): ks = 1 is the address ^ 64
  const int co_a = li & 7;
  const int wa0 = PATCH_B + co_a * 128 + ((g ^ (co_a & 6)) << 4);
  // B (pixels): this wave's two pixel tiles = tile row `wave`, columns 0..15 / 16..31; patch pixel at tap (0, 0)
  const int ppb = wave * PWD + li;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

#pragma unroll 1
  for (int cb = 0; cb < Ci; cb += 64) {
    issue_chunk(cb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      int pb = ppb;
      asm volatile("" : "+v"(pb));  // per-tap address arithmetic stays here (hoisted, 2 x 16 offsets would spill)
      int boff[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pp = pb + (tap / KS) * PWD + (tap % KS) + 16 * j;
        boff[j] = (pp << 7) | ((g ^ ((pp >> 1) & 7)) << 4);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + ((wa0 ^ (ks << 6)) + tap * 1024));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 xb = *reinterpret_cast<const bf16x8*>(patch + (boff[j] ^ (ks << 6)));
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // every wave is done with the chunk before the next fills land
  }

  // ---- epilogue: D rows = filter rows (4 g + register), columns = pixels: lane groups 0 / 1 hold channels 0..3 / 4..7
  const int oy = ty0 + wave;
  if (g < 2 && oy < Ho) {
    float bs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bs[q] = d.bias ? d.bias[4 * g + q] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ox = tx0 + 16 * j + li;
      if (ox < Wo) {
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = act_fwd(acc[j][q] + bs[q], d.act);
        unsigned short* dst = static_cast<unsigned short*>(d.y) + ((size_t)(b * Ho + oy) * Wo + ox) * 8 + 4 * g;
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// reflect border (o2m_conv2d_reflect_border): the part of the data gradient of a 3 x 3 conv behind ReflectionPad2d(1)
// that the zero-padded data-gradient conv on the CROPPED H x W domain leaves out.  With full[oy][ox] the full
// correlation on the padded (H + 2) x (W + 2) domain, the cropped conv is its interior; the ring oy in {0, H + 1} or
// ox in {0, W + 1} lands on y[R(oy - 1)][R(ox - 1)], R(-1) = 1, R(H) = H - 2 (the adjoint of the pad).  A ring pixel
// sees at most three filter taps (one filter row for the top / bottom strips, one filter column for the left / right
// ones), so the ring is 2 (W + 2) + 2 H pixels x K = 3 Ci per sample: 2 % of the layer's work, against the 6 % more
// rows + a 128 x 128 tail launch + a zero-fill launch + bf16 atomics on 12 % of the map of the padded-domain form.
// A block = 64 ring pixels of one strip of one sample (4 pixel tiles = the MFMA's columns) x 64 output channels (filter
// rows permuted as in stem8 so that a lane ends up with 16 consecutive channels of its pixel); its four waves split the
// reduction (one k-step = 32 channels under one tap) and combine through LDS as [pixel][channel] rows, added to y as
// channel PAIRS (packed bf16 atomics in their full-rate shape: two contiguous 128-B segments per wave-instruction).
// Both operands come straight from global memory.
// MEASURED (round 4, gpurun_out/r04g, r04h): 65 us per launch at B = 48 (a wave per 64 channels walking all 24 k-steps)
// -> 54 us (reduction split over the waves) -> 36 us (atomics in the contiguous shape).  What is left is operand
// traffic: 64 x 64 tiles without LDS sharing move 196 KB per block, 300 MB per launch through L2 for 5 GFLOP.  22
// launches = 0.8 ms against the 0.9 ms the cropped-domain conv saves the 256 x 256-tile kernel, and the step is 0.5 ms
// SLOWER with it (three same-box pairs): the step keeps the padded-domain form (ops._BORDER_DGRAD = off).  Making it
// pay needs the ring as rows of an LDS-tiled GEMM (128 x 128 tiles: 4x less traffic), i.e. a border row mapping in
// conv_igemm_kernel's loader and fold epilogue -- not built.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_reflect_border_kernel(const o2m_conv_desc d, const int segs) {
  constexpr int PT = 4;
  __shared__ float part[4][16][64 + 4];  // [wave][pixel][channel]: one pixel tile's partials at a time
  const int H = d.H, W = d.W, Ci = d.Ci, Co = d.Co;
  const int seg = blockIdx.x % segs, st = (blockIdx.x / segs) & 3, b = blockIdx.x / (4 * segs);
  const int L = st < 2 ? W + 2 : H, x0 = seg << 6;
  if (x0 >= L) return;  // (block-uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = blockIdx.y * 64;  // the block's 64 output channels; its four waves split the REDUCTION (k-steps w, w + 4, ..)
  const int g = lane >> 4, li = lane & 15;
  const rsrc_t xr = make_rsrc(d.x, (unsigned)((size_t)d.B * H * W * Ci * 2));
  const rsrc_t wr = make_rsrc(d.w, (unsigned)((size_t)Co * 9 * Ci * 2));

  // ring pixel -> position on the padded domain, its (<= 3) source pixels, its target.  The three sources are
  // consecutive pixels (row strips) / rows (column strips): byte offset vbase + t * vstep where bit t of vmask is set
  // (kept as base + mask, not as a table: indexed by the run-time tap a table goes to scratch)
  int oy[PT], ox[PT];
  unsigned vbase[PT], vmask[PT];
  const unsigned vstep = (unsigned)((st < 2 ? 1 : W) * Ci) * 2u;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int i = x0 + 16 * pt + li;
    oy[pt] = st == 0 ? 0 : (st == 1 ? H + 1 : 1 + i);
    ox[pt] = st < 2 ? i : (st == 2 ? 0 : W + 1);
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int ky = st == 0 ? 2 : (st == 1 ? 0 : t), kx = st < 2 ? t : (st == 2 ? 2 : 0);
      const int iy = oy[pt] + ky - 2, ix = ox[pt] + kx - 2;
      if (i < L && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) m |= 1u << t;
    }
    const int ky0 = st == 0 ? 2 : 0, kx0 = st < 2 ? 0 : (st == 2 ? 2 : 0);
    vbase[pt] = (unsigned)(((b * H + oy[pt] + ky0 - 2) * W + ox[pt] + kx0 - 2) * Ci + 8 * g) * 2u;  // (tap 0; may lie outside: masked)
    vmask[pt] = m;
  }
  unsigned woff[4];  // filter row of A tile mt, lane row li: channel cw + 16 (li >> 2) + 4 mt + (li & 3)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) woff[mt] = (unsigned)((cw + 16 * (li >> 2) + 4 * mt + (li & 3)) * 9 * Ci + 8 * g) * 2u;

  f32x4 acc[PT][4];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[pt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // k-step gs (0 .. 3 nks - 1) = 32 channels ks = gs % nks under tap t = gs / nks; this wave takes gs = wave + 4 j, two
  // steps in flight.  A step past the end loads through the out-of-range offset (zeros, no traffic).
  const int nks = Ci >> 5, nstep = 3 * nks;
  u32x4 a0[4], a1[4], q0[PT], q1[PT];
  auto load_step = [&](int gs_, u32x4 (&a)[4], u32x4 (&q)[PT]) {
    const int gs = __builtin_amdgcn_readfirstlane(gs_);
    const bool live = gs < nstep;
    const int t = gs / nks, ks = gs - t * nks;
    const int ky = st == 0 ? 2 : (st == 1 ? 0 : t), kx = st < 2 ? t : (st == 2 ? 2 : 0);
    const int wso = __builtin_amdgcn_readfirstlane(live ? ((ky * 3 + kx) * Ci + 32 * ks) * 2 : 0);
    const int xso = __builtin_amdgcn_readfirstlane(live ? 64 * ks : 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a[mt] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(live ? woff[mt] : OOB_OFF), wso, 0);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const bool ok = live && ((vmask[pt] >> t) & 1u);
      q[pt] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(ok ? vbase[pt] + (unsigned)t * vstep : OOB_OFF), xso, 0);
    }
  };
  auto multiply = [&](const u32x4 (&a)[4], const u32x4 (&q)[PT]) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[pt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a[mt]), as_frag(q[pt]), acc[pt][mt], 0, 0, 0);
  };
  load_step(wave, a0, q0);
  load_step(wave + 4, a1, q1);
#pragma unroll 1
  for (int gs = wave; gs < nstep; gs += 8) {
    multiply(a0, q0);
    load_step(gs + 8, a0, q0);
    multiply(a1, q1);  // (gs + 4 >= nstep: zero fills)
    load_step(gs + 12, a1, q1);
  }

  // ---- the four waves' partials meet in LDS, one pixel tile per round, as [pixel][channel] rows; the 256 threads then
  // add CHANNEL PAIRS: a wave-instruction covers two pixels x 32 pairs = two contiguous 128-B segments, the full-rate
  // shape of the memory-side atomics (the lane-per-pixel shape of the accumulators -- 64 scattered dwords per
  // instruction -- ran the launch at 54 us, MI355X_MICROARCH.md "Global float atomics")
  unsigned short* __restrict__ Y = static_cast<unsigned short*>(d.y);
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) s16x2_t gs16x2_t;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const f32x4 v = acc[pt][mt];
      part[wave][li][16 * g + 4 * mt + 0] = v[0];
      part[wave][li][16 * g + 4 * mt + 1] = v[1];
      part[wave][li][16 * g + 4 * mt + 2] = v[2];
      part[wave][li][16 * g + 4 * mt + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pair = tid + 256 * k, px = pair >> 5, cp = pair & 31;
      const int i = x0 + 16 * pt + px;
      if (i < L) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { s0 += part[wv][px][2 * cp]; s1 += part[wv][px][2 * cp + 1]; }
        int ty = (st == 0 ? 0 : (st == 1 ? H + 1 : 1 + i)) - 1, tx = (st < 2 ? i : (st == 2 ? 0 : W + 1)) - 1;
        ty = ty < 0 ? 1 : (ty >= H ? H - 2 : ty);
        tx = tx < 0 ? 1 : (tx >= W ? W - 2 : tx);
        unsigned short* dst = Y + ((size_t)(b * H + ty) * W + tx) * Co + cw + 2 * cp;
        __builtin_amdgcn_global_atomic_fadd_v2bf16((gs16x2_t*)dst, __builtin_bit_cast(s16x2_t, pack_bf2(s0, s1)));
      }
    }
    __syncthreads();
  }
}

bool plain(const o2m_conv_desc& d) {
  return d.dtype == O2M_BF16 && d.stride <= 1 && !d.in_scale && !d.out_scale && !d.residual && !d.aux && !d.aux_scaled &&
         !d.fold_pad && d.w_batch_stride == 0 && !d.deq_scale && d.KH == d.KW;
}

}  // namespace

namespace o2m_direct {

static int enabled() {
  static const int on = [] { const char* e = getenv("O2M_CONV_DIRECT"); return e ? atoi(e) : 1; }();
  return on;
}

bool stem8_ok(const o2m_conv_desc& d) {
  if (!enabled() || !plain(d) || d.Ci != 8 || d.Co != 64 || (d.KH != 4 && d.KH != 7)) return false;
  if (d.stats && d.stats_mode != O2M_STATS_MOMENTS) return false;
  if (d.pad_mode == O2M_PAD_REFLECT && (d.pad >= d.H || d.pad >= d.W)) return false;
  return (long)d.B * d.H * d.W * 16 < 0x7fffffffL;
}

int stem8_rows(const o2m_conv_desc& d) {  // output rows per block; a stats partial covers rows * Wo consecutive pixels
  // 7 x 7: the 56 KB filter image is rebuilt per block -- 8 rows amortise it; 4 x 4 (16 KB): 2 rows, four times the
  // blocks (B = 16 at 255 rows: 512 blocks of 8 rows left the CUs two resident blocks = 8 waves each, latency-bound)
  const int Ho = d.H + 2 * d.pad - d.KH + 1;
  if (d.KH == 4) return 2;
  return Ho % 8 == 0 ? 8 : (Ho % 4 == 0 ? 4 : 8);
}

int stem8_stats_rows(const o2m_conv_desc& d) {
  const int Ho = d.H + 2 * d.pad - d.KH + 1, Wo = d.W + 2 * d.pad - d.KW + 1, rows = stem8_rows(d);
  return Ho % rows == 0 ? rows * Wo : 0;
}

int launch_stem8(const o2m_conv_desc& d, hipStream_t s) {
  const int Ho = d.H + 2 * d.pad - d.KH + 1, Wo = d.W + 2 * d.pad - d.KW + 1, rows = stem8_rows(d);
  if (d.stats && Ho % rows != 0) return O2M_ERR_BAD_ARG;
  const long blocks = (long)d.B * ((Ho + rows - 1) / rows);
  if (blocks <= 0 || blocks > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  const int nstep = d.KH * ((d.KH + 3) / 4);
  const int lds = nstep * 4096 + 2048;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, d, rows);
  };
  {
    LaunchScope timed(s, 2.0 * d.B * Ho * Wo * 64.0 * d.KH * d.KW * 8.0, "conv_stem8<bf16,%dx%d>", d.KH, d.KW);
    if (d.KH == 4) { if (d.stats) go(conv_stem8_kernel<4, true>); else go(conv_stem8_kernel<4, false>); }
    else           { if (d.stats) go(conv_stem8_kernel<7, true>); else go(conv_stem8_kernel<7, false>); }
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

bool fewout_ok(const o2m_conv_desc& d) {
  if (!enabled() || !plain(d) || d.Co != 8 || d.Ci % 64 != 0 || d.KH != 4 || d.pad_mode != O2M_PAD_ZERO || d.stats) return false;
  return (long)d.B * d.H * d.W * d.Ci * 2 < 0x7fffffffL;
}

int launch_fewout(const o2m_conv_desc& d, hipStream_t s) {
  const int Ho = d.H + 2 * d.pad - d.KH + 1, Wo = d.W + 2 * d.pad - d.KW + 1;
  const long blocks = (long)d.B * ((Ho + 7) / 8) * ((Wo + 31) / 32);
  if (blocks <= 0 || blocks > 0x7fffffffL) return O2M_ERR_BAD_ARG;
  constexpr int lds = ((11 * 35 + 7) / 8) * 1024 + 16 * 1024;  // patch (49 fills) + 16 taps x 8 rows x 128 B
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fewout_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  {
    LaunchScope timed(s, 2.0 * d.B * Ho * Wo * 8.0 * d.KH * d.KW * d.Ci, "conv_fewout<bf16,%dx%d>", d.KH, d.KW);
    hipLaunchKernelGGL(conv_fewout_kernel<4>, dim3((unsigned)blocks), dim3(512), lds, s, d);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}

}  // namespace o2m_direct

extern "C" int o2m_conv2d_reflect_border(const o2m_conv_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y) return O2M_ERR_BAD_ARG;
  if (d->dtype != O2M_BF16 || d->KH != 3 || d->KW != 3 || d->B <= 0 || d->H < 4 || d->W < 4) return O2M_ERR_BAD_ARG;
  if (d->Ci <= 0 || d->Co <= 0 || (d->Ci % 64 && d->Ci != 32) || d->Co % 64) return O2M_ERR_BAD_ARG;
  if (d->in_scale || d->out_scale || d->bias || d->residual || d->stats || d->aux || d->aux_scaled || d->w_batch_stride ||
      d->stride > 1 || d->fold_pad || d->deq_scale)
    return O2M_ERR_BAD_ARG;
  if ((long)d->B * d->H * d->W * d->Ci * 2 > 0x7fffffffL || (long)d->Co * 9 * d->Ci * 2 > 0x7fffffffL) return O2M_ERR_UNSUPPORTED;
  const int longest = (d->W + 2 > d->H ? d->W + 2 : d->H), segs = (longest + 63) / 64;
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    LaunchScope timed(s, 2.0 * d->B * (2.0 * (d->W + 2) + 2.0 * d->H) * d->Co * 3.0 * d->Ci, "%s", "conv_reflect_border<bf16,3x3>");
    hipLaunchKernelGGL(conv_reflect_border_kernel, dim3((unsigned)(d->B * 4 * segs), (unsigned)(d->Co / 64)), dim3(256), 0, s,
                       *d, segs);
  }
  O2M_LAUNCH_CHECK();
  return 0;
}
