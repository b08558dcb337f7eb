// 3x3 convolution on the bf16 / fp16 matrix pipe, round 6: sliding-window implicit GEMM on 2-byte activation planes, every operand
// by LDS-DMA (models/VGG16/mnc_5stage/test.prototxt:41-412; BASELINE.json configs[2] "bf16 convs via MFMA").  Replaces the
// round-1/2 kernel (conv3x3_x3_kernel: register staging, 2x2 register tile, 12-dword pixel pitch, separate tail-reduction launch).
//
// Arithmetic.  M = output channels (MFMA A operand), N = 32 pixels of one image row (B operand), K = 16 input channels x one tap
// per v_mfma_f32_32x32x16_{f16,bf16}; fp32 accumulation.  Three math modes share the kernel:
//   f16  (1): operands rounded to fp16 (weights at load, activations by their producer);
//   bf16 (2): the same with bf16 (BASELINE configs[2] as written: one product per term);
//   x3   (0): "bf16x3" split precision (x3_split.h): every fp32 operand is hi + lo (both bf16) and a product is
//             w_hi x_hi + w_hi x_lo + w_lo x_hi.  A 16-channel block is multiplied in three PASSES that each look like one fp16
//             block: pass 0 = [w_hi(c 0-7) | w_hi(c 0-7)] x [x_hi(c 0-7) | x_lo(c 0-7)], pass 1 the same for channels 8-15,
//             pass 2 = [w_lo(c 0-7) | w_lo(c 8-15)] x [x_hi(c 0-7) | x_hi(c 8-15)]: 27 MFMAs per block and tile, no padding slot
//             (the old kernel: 30).
//
// Register tile and LDS traffic.  A wave owns PR = 5 consecutive pixel rows x 32 columns x 32 output channels (80 accumulator
// registers).  Per 16-channel block it reads its nine weight fragments once (9 x ds_read_b128) and every halo row's three column
// shifts once (7 rows x 3 = 21 reads); a halo fragment feeds the up to three output rows it is a tap of (45 MFMAs).  30 fragment
// reads per 45 MFMAs = 0.67 per MFMA, what a 3x3 register tile gets with 144 accumulators, and the LDS carries no transposed or
// padded image: the weight panel [tap][k half][32 channels] x 16 B and the halo planes [k half][row][34 columns] x 16 B are both
// read as 512 contiguous bytes per 32 lanes -- conflict-free without padding, which is what lets LDS-DMA fill them
// (buffer_load_dwordx4 ... lds writes 64 lanes x 16 B linearly; the gather is on the global side, out-of-image pixels carry an
// out-of-range offset and arrive as zeros).  No staging registers, no ds_write, no conversion arithmetic in the loop.
//
// Work decomposition.  Workgroup = KW K-ranges x RG row groups x CG channel tiles (waves).  600x1000 VGG-16 maps are 600 / 300 /
// 150 / 75 / 38 rows: five-row wave tiles divide all but the last exactly.  (RG, CG, KW) = (2, 2, 1): 10 rows x 32 columns x 64
// channels per 256-thread workgroup, 64 KB of LDS, two per CU -- 1920 / 960 / 480 workgroups on conv1_2 / conv2_x / conv3_x = 3.75 /
// 1.875 / 0.94 rounds of the chip's 512 slots.  conv4_x would be 256 such workgroups (one per CU, one wave per SIMD): (2, 2, 2) puts
// two K ranges of the same tile into one 512-thread workgroup instead (own LDS buffers per range, one shared barrier), summed through
// LDS at the end -- no partial sums in HBM, no second launch.  conv5_x / rpn (38 x 63) run (1, 1, 4): 256 workgroups of four K
// ranges.  The old kernel's K-split tail with its x3_tail_reduce launch is gone.
//
// Pipeline.  One barrier per pass: wait for my copies of pass v, barrier (everybody's copies landed, everybody finished reading the
// other buffer), issue the copies of pass v + 1 into the other buffer, multiply pass v.
#include <atomic>
#include <type_traits>

#include "mnc_internal.h"
#include "x3_split.h"

// Tuning ablations (never set in a product build; wrong results): 1 = no copies inside the loop, 2 = no fragment reads (MFMAs on
// register constants), 4 = no output stores, 8 = the copies inside the loop carry only out-of-range lanes,
// 16 / 32 = every pass copies the first chunk's weights / halo again (cache hits), 64 = no wait and no barrier between the passes, 128 = both buffers filled with the first pass's (real) data before the loop (with 1: MFMAs on real operands, no copies)
#ifndef MNC_SW_ABL
#define MNC_SW_ABL 0
#endif
#ifndef MNC_SW_ISSUE
#define MNC_SW_ISSUE 0
#endif

namespace mnc {

typedef float sw_f32x16 __attribute__((ext_vector_type(16)));
typedef int sw_i32x4 __attribute__((ext_vector_type(4)));

constexpr int kSwCols = 32, kSwHC = kSwCols + 2;
constexpr int kSwOOB = 0x7FFFFFF0;                 // per-lane offset no buffer reaches: the copy answers with zeros
constexpr int kSwTapB = 32 * 16;                   // one (plane, tap) of a 32-channel tile: 32 x 16 B
constexpr int kSwPlaneB = 9 * kSwTapB;

// compile-time loop: f(std::integral_constant<int, 0>()) ... f(<N - 1>)
template <int I, int N, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    x3_static_for<I + 1, N>(f);
  }
}

__host__ __device__ constexpr int sw_planes(int mode) { return mode == 0 ? 4 : 2; }
__host__ __device__ constexpr int sw_pixb(int mode) { return mode == 0 ? 32 : 16; }

// Geometry of one instantiation (host and device)
template <int PR, int RG, int CG, int KW>
struct SwGeom {
  static constexpr int NWG = RG * CG;                        // waves of one K range
  static constexpr int NT = 64 * NWG * KW;
  static constexpr int ROWS = RG * PR;
  static constexpr int HR = ROWS + 2;
  static constexpr int UPL = HR * kSwHC;                     // 16-byte units of one halo plane
  static constexpr int PP = (UPL + 63) / 64;                 // 1 KB copy pieces per halo plane
  static constexpr int PLANEB = PP * 1024;
  static constexpr int AB = CG * 9 * 1024;                   // weight panel: [channel tile][tap][k half][32] x 16 B
  static constexpr int BUFB = AB + 2 * PLANEB;
  static constexpr int NAS = (9 * CG + NWG - 1) / NWG;       // copy slots per wave and pass
  static constexpr int NBS = (2 * PP + NWG - 1) / NWG;
  static constexpr int DUMMY = KW * 2 * BUFB;                // 1 KB that surplus slots fill with zeros
  static constexpr int RED = (KW - 1) * NWG * PR * 4096;     // K ranges' accumulators on their way to range 0
  static constexpr int LDS = (DUMMY > RED ? DUMMY : RED) + 1024;
};

template <int MODE, int PR, int RG, int CG, int KW>
__global__ __launch_bounds__(64 * RG * CG * KW) void conv3x3_sw_kernel(const void* __restrict__ in_, const void* __restrict__ wpk_,
                                                                       const float* __restrict__ bias, void* __restrict__ out_pk,
                                                                       float* __restrict__ out_f32, int H, int W, int Cin, int Cout,
                                                                       int relu, int pool) {
  typedef SwGeom<PR, RG, CG, KW> G;
  constexpr int NPL = sw_planes(MODE), NPASS = MODE == 0 ? 3 : 1, PIXB = sw_pixb(MODE);
  extern __shared__ __attribute__((aligned(16))) unsigned char s_sw[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kw = wave / G::NWG, wl = wave - kw * G::NWG;
  const int rg = wl / CG, cg = wl - rg * CG;
  const int j = lane & 31, kb = lane >> 5;

  // ---- tile of this workgroup: channel group fastest, then column tile, then row tile; every XCD (block b runs on XCD b % 8,
  // used for speed only) takes a contiguous range of that order, so the channel groups of a spatial tile share one L2
  const int tiles_x = (W + kSwCols - 1) / kSwCols, ncog = Cout / (32 * CG), ncot = Cout >> 5;
  int logical;
  {
    const int total = (int)gridDim.x, q = total >> 3, r = total & 7;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cog = logical % ncog, sp = logical / ncog;
  const int tx = sp % tiles_x, ty = sp / tiles_x;
  const int w0 = tx * kSwCols, h0 = ty * G::ROWS, cot0 = cog * CG;
  const int nblk = Cin >> 3, nchunks = (nblk + 1) >> 1;
  const int c_begin = kw * nchunks / KW, c_end = (kw + 1) * nchunks / KW;      // (the launcher makes nchunks a multiple of KW)

  // ---- copy assignment, fixed per thread.  Weight slot i of wave wl = piece wl + NWG i of the panel (channel tile pa / 9, tap
  // pa % 9): lanes 0-31 fetch the tap's k-half-0 plane, lanes 32-63 its k-half-1 plane.  Halo slot i = piece pb of the two planes:
  // unit 64 (pb % PP) + lane = (row, column) of the halo.  Slots past the last piece fill the dummy kilobyte with zeros.
  auto make_rsrc = [](const void* base, long bytes) {
    const unsigned long a = (unsigned long)base;
    sw_i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFFL ? bytes : 0x7FFFFFFFL));
    r.w = 0x00020000;
    return r;
  };
  const int blk_bytes = H * W * PIXB;                              // one 8-channel block of the input
  const sw_i32x4 in_rsrc = make_rsrc(in_, (long)nblk * blk_bytes);
  const sw_i32x4 w_rsrc = make_rsrc(wpk_, (long)nchunks * ncot * NPL * kSwPlaneB);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)s_sw;
  const int kwbase = kw * 2 * G::BUFB;

  int vb[G::NBS];
#pragma unroll
  for (int i = 0; i < G::NBS; ++i) {
    const int pb = wl + G::NWG * i;
    const int unit = (pb % G::PP) * 64 + lane;
    const int row = unit / kSwHC, col = unit - row * kSwHC;
    const int y = h0 - 1 + row, x = w0 - 1 + col;
    const bool ok = pb < 2 * G::PP && unit < G::UPL && y >= 0 && y < H && x >= 0 && x < W;
    vb[i] = ok ? (y * W + x) * PIXB : kSwOOB;
  }
  const int va0 = j * 16, va1 = j * 16 + kb * kSwPlaneB;

  // Copy slot k (of NS per wave and pass) of pass `pass` of chunk c -> buffer `buf` of this K range.  live == false (behind the
  // last pass): every lane out of range -- zeros into the free buffer, no memory traffic, the loop body stays one basic block.
  constexpr int NS = G::NAS + G::NBS;
  auto dma_slot = [&](int c, auto pass_, int buf, int k, bool more, bool in_loop = true) {
    constexpr int pass = decltype(pass_)::value;
    if ((MNC_SW_ABL & 1) && in_loop) return;
    if (MNC_SW_ABL & 8) more = false;                          // copies issued, every lane out of range: issue cost without traffic
    const sw_i32x4 wr = w_rsrc, ir = in_rsrc;                  // (named here: a generic lambda does not capture what only an asm operand uses)
    const unsigned base = lds0 + (unsigned)(kwbase + buf * G::BUFB);
    if (k < G::NAS) {
      // weights: plane pair of the pass = (h0, h0) | (h1, h1) | (l0, l1) in x3, (k half 0, k half 1) otherwise
      constexpr int a_plane0 = MODE == 0 ? (pass == 2 ? 2 : pass) : 0;
      constexpr bool a_two = MODE != 0 || pass == 2;
      const int pa = wl + G::NWG * k;
      const bool live = pa < 9 * CG;
      const int cgi = pa / 9, tap = pa - cgi * 9;
      const int ca = (MNC_SW_ABL & 16) ? c_begin : c;          // ablation: the same weight panel every pass (L2 hits)
      const int so = __builtin_amdgcn_readfirstlane(((ca * ncot + cot0 + cgi) * NPL + a_plane0) * kSwPlaneB + tap * kSwTapB);
      const unsigned l = __builtin_amdgcn_readfirstlane(live ? base + (unsigned)pa * 1024u : lds0 + (unsigned)G::DUMMY);
      const int vo = (live && more) ? (a_two ? va1 : va0) : kSwOOB;
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(wr), "s"(so), "s"(l) : "memory");
    } else {
      // halo: block and 16-byte half of the pixel record per k half.  x3: pass 0 (2c: hi | lo), pass 1 (2c + 1: hi | lo),
      // pass 2 (hi of 2c | hi of 2c + 1); otherwise (2c | 2c + 1).  A block past the input's last (odd block count) reads as zeros.
      const int i = k - G::NAS;
      const int pb = wl + G::NWG * i;
      const bool live = pb < 2 * G::PP;
      const int kbp = pb / G::PP;
      int blk, half;
      const int ch = (MNC_SW_ABL & 32) ? c_begin : c;          // ablation: the same halo every pass
      if (MODE == 0 && pass < 2) { blk = 2 * ch + pass; half = kbp; }
      else { blk = 2 * ch + kbp; half = 0; }
      const int so = __builtin_amdgcn_readfirstlane(min(blk, nblk - 1) * blk_bytes + half * 16);
      const unsigned l = __builtin_amdgcn_readfirstlane(live ? base + (unsigned)(G::AB + pb * 1024) : lds0 + (unsigned)G::DUMMY);
      const int vo = (live && more && blk < nblk) ? vb[i] : kSwOOB;
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(ir), "s"(so), "s"(l) : "memory");
    }
  };
#define MNC_SW_SYNC() asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  sw_f32x16 acc[PR];
#pragma unroll
  for (int r = 0; r < PR; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

  auto mfma = [](const uint4 a, const uint4 b, const sw_f32x16 c) {
    if constexpr (MODE == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_as_f16x8(a), x3_as_f16x8(b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(a), x3_as_bf16x8(b), c, 0, 0, 0);
  };
  auto opaque = [](uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
  // One pass out of buffer `buf`: halo row hr, column shift dx is tap (dy, dx) of output row hr - dy.  The NS copies of the NEXT
  // pass (`issue(k)`) are spread over the 3 (PR + 2) fragment groups, so that the matrix pipe works off queued MFMAs while the
  // wave issues a copy.
  auto compute = [&](int buf, auto&& issue) {
    const unsigned char* pb_ = s_sw + kwbase + buf * G::BUFB;
    const uint4* Ap = reinterpret_cast<const uint4*>(pb_ + cg * 9 * 1024) + lane;
    const uint4* Bp = reinterpret_cast<const uint4*>(pb_ + G::AB + kb * G::PLANEB) + (rg * PR * kSwHC + j);
    constexpr int NGRP = 3 * (PR + 2);
    // copy slot k goes behind fragment group k * kIssueDen / kIssueNum: spread over all groups of the pass -- except in the one-wave
    // K ranges of the split-bf16 mode (the 38x63 layers: 17 copies per wave and pass, nobody else on the SIMD to cover a late one),
    // which issue four per group from the start of the pass: conv5_1 42.9 -> 37.9 us (profiles/r06_conv_sw.txt section 8; every
    // other plan and mode is within run-to-run noise of the spread form, fp16's one-wave plan 4 % slower).  MNC_SW_ISSUE = n: n per 2 groups.
    constexpr bool kFront = MNC_SW_ISSUE != 0 || (G::NWG == 1 && MODE == 0);
    constexpr int kIssueNum = MNC_SW_ISSUE ? MNC_SW_ISSUE : kFront ? 8 : NS, kIssueDen = kFront ? 2 : NGRP;
    uint4 a[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MNC_SW_ABL & 2) { a[t] = make_uint4(0x3c003c00u + t, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); opaque(a[t]); }
      else a[t] = Ap[t * 64];
    }
    int k = 0;
#pragma unroll
    for (int hr = 0; hr < PR + 2; ++hr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        uint4 b;
        if (MNC_SW_ABL & 2) { b = make_uint4(0x3c003c00u + hr, 0x3c003c00u + dx, 0x3c003c00u, 0x3c003c00u); opaque(b); }
        else b = Bp[hr * kSwHC + dx];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int r = hr - dy;
          if (r >= 0 && r < PR) acc[r] = mfma(a[dy * 3 + dx], b, acc[r]);
        }
        const int g = hr * 3 + dx;
#pragma unroll
        for (int q = 0; q < NS; ++q)
          if (q == k && k * kIssueDen < (g + 1) * kIssueNum) { issue(k); ++k; }
      }
  };

  // ---- main loop over the passes of this K range, two buffers: the copies of pass v + 1 are issued while pass v is multiplied.
  // (Three buffers -- two passes ahead, the newest pass's copies left in flight across the barrier -- were built and measured on the
  // 8-wave plans whose LDS holds them: no gain, profiles/r06_conv_sw.txt; removed.)
#pragma unroll
  for (int k = 0; k < NS; ++k) dma_slot(c_begin, std::integral_constant<int, 0>(), 0, k, true, false);
  if (MNC_SW_ABL & 128) {                                    // ablation 128 (+ 1): both buffers hold real data, no copies in the loop
#pragma unroll
    for (int k = 0; k < NS; ++k) dma_slot(c_begin, std::integral_constant<int, 0>(), 1, k, true, false);
  }
  int bc = 0;                                                // buffer of the pass being multiplied
  for (int c = c_begin; c < c_end; ++c) {
    x3_static_for<0, NPASS>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      const int ct = c + (p + 1) / NPASS;
      const bool more = ct < c_end;
      const int ctc = more ? ct : c_end - 1;
      // (a K range of ONE wave -- the 38x63 plan -- reads only what it copied itself: its own wait is the whole synchronisation, the
      // four ranges of the workgroup meet only at the final sum)
      if constexpr (G::NWG == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else if (!(MNC_SW_ABL & 64)) MNC_SW_SYNC();            // (ablation 64: no wait, no barrier -- racy, timing only)
      compute(bc, [&](int k) { dma_slot(ctc, std::integral_constant<int, (p + 1) % NPASS>(), bc ^ 1, k, more); });
      bc ^= 1;
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the last pass's all-out-of-range copies)

  // ---- K ranges of the workgroup: ranges 1.. hand their accumulators to range 0 through LDS (summed in range order)
  if constexpr (KW > 1) {
    MNC_SW_SYNC();
    float4* red = reinterpret_cast<float4*>(s_sw);
    if (kw > 0) {
#pragma unroll
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          red[((((kw - 1) * G::NWG + wl) * PR + r) * 4 + q) * 64 + lane] =
              make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
    }
    __syncthreads();
    if (kw > 0 && !pool) return;                             // (with the pooling epilogue: two more barriers to attend)
    if (kw == 0)
#pragma unroll
    for (int k = 1; k < KW; ++k)
#pragma unroll
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 p = red[((((k - 1) * G::NWG + wl) * PR + r) * 4 + q) * 64 + lane];
          acc[r][4 * q] += p.x; acc[r][4 * q + 1] += p.y; acc[r][4 * q + 2] += p.z; acc[r][4 * q + 3] += p.w;
        }
  }

  // ---- epilogue with the following Pooling MAX 2x2/2 (Caffe's ceil output size) folded in (test.prototxt:81-92, 137-148, 221-232,
  // 305-316; row groups even: a workgroup's rows start at an even image row).  A wave's five rows are two whole window rows and half
  // of a third: the odd row group hands its first row to the even one below it through LDS.  Columns: lanes 2m, 2m + 1 (DPP
  // quad_perm) -- even lanes keep the window's maximum.  max commutes with + bias and ReLU, so this is the packed form of the pooled
  // fp32 tensor, and (rounding is monotonic, x -> (hi, lo) too) what mnc_maxpool2_c8_* makes of the unpooled packed output.
  if constexpr (RG % 2 == 0) {
    if (pool) {
      MNC_SW_SYNC();                                         // (KW == 1: the other waves' last fragment reads)
      float4* xch = reinterpret_cast<float4*>(s_sw + G::RED);
      const int slot = (rg >> 1) * CG + cg;
      if (kw == 0 && (rg & 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xch[(slot * 4 + q) * 64 + lane] = make_float4(acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]);
      }
      __syncthreads();
      if (kw > 0) return;
      sw_f32x16 recv = acc[0];
      if (!(rg & 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 p = xch[(slot * 4 + q) * 64 + lane];
          recv[4 * q] = p.x; recv[4 * q + 1] = p.y; recv[4 * q + 2] = p.z; recv[4 * q + 3] = p.w;
        }
      }
      const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
      const int ow = w0 + j, co0 = (cot0 + cg) * 32;
      const bool partner = (ow | 1) < W;                     // the window's second column exists
      const bool active = !(j & 1) && ow < W && (!(MNC_SW_ABL & 4) || relu == 0x7fff);
      const long gstride = (long)OH * OW;
      auto emit = [&](const sw_f32x16& top, const sw_f32x16& bot, int rtop) {      // rows rtop (even), rtop + 1 of the image
        if (rtop >= H) return;
        const bool two = rtop + 1 < H;
        float4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b = *reinterpret_cast<const float4*>(bias + co0 + g * 8 + kb * 4);
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float m = two ? fmaxf(top[4 * g + e], bot[4 * g + e]) : top[4 * g + e];
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true));
            m = partner ? fmaxf(m, o) : m;
            x[e] = m;
          }
          v[g] = make_float4(x[0] + b.x, x[1] + b.y, x[2] + b.z, x[3] + b.w);
          if (relu) { v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f); }
        }
        if (!active) return;                                 // (lanes j and j + 32 agree)
        const long pix0 = ((long)(co0 >> 3) * OH + (rtop >> 1)) * OW + (ow >> 1);
        if constexpr (MODE != 0) {
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const uint2 A = MODE == 2 ? x3_bf16x4(v[g]) : x3_f16x4(v[g]), B = MODE == 2 ? x3_bf16x4(v[g + 1]) : x3_f16x4(v[g + 1]);
            const auto sx = __builtin_amdgcn_permlane32_swap(A.x, B.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(A.y, B.y, false, false);
            reinterpret_cast<uint4*>(out_pk)[pix0 + (g + kb) * gstride] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hi, lo;
            x3_split4(v[g], hi, lo);
            const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
            reinterpret_cast<uint4*>(out_pk)[(pix0 + g * gstride) * 2 + kb] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
      };
      const int rbase = h0 + rg * PR;
      static_assert(PR == 5, "pooling epilogue: five rows per wave");
      if (rg & 1) {
        emit(acc[1], acc[2], rbase + 1);
        emit(acc[3], acc[4], rbase + 3);
      } else {
        emit(acc[0], acc[1], rbase);
        emit(acc[2], acc[3], rbase + 2);
        emit(acc[4], recv, rbase + 4);
      }
      return;
    }
  }

  // ---- epilogue: D[row = channel (e & 3) + 8 (e >> 2) + 4 kb][column = pixel j]; bias, ReLU, the requested output forms
  const int ow = w0 + j, co0 = (cot0 + cg) * 32;
  const long gstride = (long)H * W;
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    const int oh = h0 + rg * PR + r;
    if (oh < H && ow < W && (!(MNC_SW_ABL & 4) || relu == 0x7fff)) {             // lanes j and j + 32 (the two channel halves of a pixel) agree
      float4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bias + co0 + g * 8 + kb * 4);
        v[g] = make_float4(acc[r][4 * g + 0] + b.x, acc[r][4 * g + 1] + b.y, acc[r][4 * g + 2] + b.z, acc[r][4 * g + 3] + b.w);
        if (relu) { v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f); }
      }
      const long pix0 = ((long)(co0 >> 3) * H + oh) * W + ow;   // + g * H * W
      if (out_f32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(out_f32 + (pix0 + g * gstride) * 8 + kb * 4) = v[g];
      }
      if (out_pk) {
        if constexpr (MODE != 0) {
          // 16-byte stores: v_permlane32_swap hands lane j the whole pixel of channel block g and lane j + 32 that of g + 1
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const uint2 A = MODE == 2 ? x3_bf16x4(v[g]) : x3_f16x4(v[g]), B = MODE == 2 ? x3_bf16x4(v[g + 1]) : x3_f16x4(v[g + 1]);
            const auto sx = __builtin_amdgcn_permlane32_swap(A.x, B.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(A.y, B.y, false, false);
            reinterpret_cast<uint4*>(out_pk)[pix0 + (g + kb) * gstride] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        } else {
          // lane j stores the pixel's hi x8 (its own four channels and lane j + 32's), lane j + 32 the lo x8
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hi, lo;
            x3_split4(v[g], hi, lo);
            const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
            reinterpret_cast<uint4*>(out_pk)[(pix0 + g * gstride) * 2 + kb] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
      }
    }
  }
#undef MNC_SW_SYNC
}

// OIHW fp32 -> [ceil(Cin/16)][Cout/32][planes][9 taps][32 channels] x 16 B (8 two-byte values).  f16 / bf16: plane p = channels
// 8p .. 8p + 7 of the 16-channel block, rounded to nearest even.  x3: planes (hi of channels 0-7, hi of 8-15, lo of 0-7, lo of
// 8-15), hi = rne(w), lo = rne(w - hi).  Channels past Cin are zero.
template <int MODE>
__global__ void pack_conv_sw_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin) {
  constexpr int NPL = sw_planes(MODE);
  const int nchunks = (Cin + 15) / 16, ncot = Cout >> 5;
  const long total = (long)nchunks * ncot * NPL * 9 * 32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long q = i;
    const int col = (int)(q & 31); q >>= 5;
    const int tap = (int)(q % 9); q /= 9;
    const int plane = (int)(q % NPL); q /= NPL;
    const int cot = (int)(q % ncot), chunk = (int)(q / ncot);
    const int co = cot * 32 + col, c0 = chunk * 16 + (plane & 1) * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = c0 + e < Cin ? w[((long)co * Cin + c0 + e) * 9 + tap] : 0.f;
    uint4 v;
    if (MODE == 1) {
      const f16x8 h = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3], (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
      v = __builtin_bit_cast(uint4, h);
    } else if (MODE == 2) {
      bf16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (__bf16)x[e];
      v = __builtin_bit_cast(uint4, h);
    } else {
      uint4 hi, lo;
      x3_split8_rne(x, hi, lo);
      v = plane < 2 ? hi : lo;
    }
    out[i] = v;
  }
}

template <int F16>
__global__ __launch_bounds__(256) void maxpool2_packed_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int CB, int H,
                                                              int W, int OH, int OW);

static int sw_grid_for(long total) {
  const long g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

template <int MODE, int PR, int RG, int CG, int KW>
static int launch_sw(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out_pk, float* d_out_f32, int H,
                     int W, int Cin, int Cout, int relu, int pool) {
  typedef SwGeom<PR, RG, CG, KW> G;
  static_assert(G::LDS <= 160 * 1024, "conv3x3_sw: LDS budget");
  auto kern = conv3x3_sw_kernel<MODE, PR, RG, CG, KW>;
  static std::atomic<unsigned long long> attr_set{0};          // one bit per device: function attributes are per device
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const int blocks = cdiv(W, kSwCols) * cdiv(H, G::ROWS) * (Cout / (32 * CG));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(G::NT), G::LDS, ctx->stream, d_in, d_wpk, d_bias, d_out_pk, d_out_f32, H, W, Cin, Cout,
                     relu, RG % 2 == 0 ? pool : 0);
  return MNC_OK;
}

// Which (RG, CG, KW) a shape runs on -- see the header comment.  Plan 0: (2, 2, 1); 1: (2, 2, 2); 2: (1, 1, 4); 3: (1, 1, 1), the
// form every shape fits (Cout % 32 == 0).  CONVX3_TILE = 100 + plan overrides the choice where the plan fits the shape.  (8-wave
// workgroups of 20 rows x 64 channels / 10 rows x 128 channels, with two and three staging buffers, were measured on conv1_2 ..
// conv3_3 in round 6: within 2 % of plan 0 everywhere, never better -- profiles/r06_conv_sw.txt -- and removed.)
static int sw_plan(const mnc_ctx* ctx, int H, int W, int Cin, int Cout) {
  const int nchunks = (Cin / 8 + 1) / 2;
  const bool fits[4] = {Cout % 64 == 0, Cout % 64 == 0 && nchunks % 2 == 0, nchunks % 4 == 0, true};
  if (tune_set(ctx, T_CONVX3_TILE)) {
    const int p = tune(ctx, T_CONVX3_TILE, 0) - 100;
    if (p >= 0 && p < 4 && fits[p]) return p;
  }
  const long wg0 = (long)cdiv(W, kSwCols) * cdiv(H, 10) * (Cout / 64);
  // Round 6 (profiles/r06_fc_ranges.txt): with several images in flight the plan that costs the least CU time wins, and that is plan 0
  // -- four-wave workgroups, each wave its own K loop, no sums through LDS -- down to 64 workgroups (conv5_x / rpn_conv on a quarter of
  // the chip): f16 918 -> 978 images/s, mixed 578 -> 611 against the chip-filling plans 1 / 2 (conv4_x in two K ranges per workgroup,
  // conv5_x in four), which remain the latency plan (PLAN=1: one image at a time f16 572 vs 549, mixed 425 vs 377).
  const bool lat = plan_latency(ctx);
  const int p0min = tune(ctx, T_CONVX3_P0MIN, lat ? 384 : 64), p1min = tune(ctx, T_CONVX3_P1MIN, lat ? 128 : 64);
  if (fits[0] && wg0 >= p0min) return 0;
  if (fits[1] && wg0 >= p1min) return 1;
  if (fits[2] && (long)cdiv(W, kSwCols) * cdiv(H, 5) * (Cout / 32) <= 640) return 2;
  if (fits[0] && wg0 >= 128) return 0;
  return fits[0] && wg0 >= 64 ? 0 : 3;
}

template <int MODE>
static int conv3x3_sw(mnc_ctx* ctx, const char* name, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out_pk,
                      float* d_out_f32, int H, int W, int Cin, int Cout, int relu, int pool = 0) {
  MNC_REQUIRE(ctx && d_in && d_wpk && d_bias && (d_out_pk || d_out_f32), "%s: null pointer", name);
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "%s: unsupported shape H=%d W=%d Cin=%d Cout=%d (need Cin%%8==0, Cout%%32==0)", name, H, W, Cin, Cout);
  MNC_REQUIRE((double)H * W * Cin * (sw_pixb(MODE) / 8) < 2.0e9 && (double)H * W * Cout * 4 < 8.0e9,
              "%s: the input tensor must stay below 2 GB (32-bit copy offsets)", name);
  const double flops = 2.0 * H * W * 9.0 * Cin * Cout;
  const double ib = MODE == 0 ? 4.0 : 2.0;
  MNC_REQUIRE(!pool || (d_out_pk && !d_out_f32), "%s: the pooling epilogue writes the packed form only", name);
  // The plan is the unpooled convolution's (same K ranges, same bits).  Plans with one row group per workgroup (tiles that start on odd
  // image rows) have no pooling epilogue: the convolution goes to the scratch arena and the pooling kernel follows.
  const int plan = sw_plan(ctx, H, W, Cin, Cout);
  void* d_pooled = nullptr;
  if (pool && (plan == 2 || plan == 3)) {
    int rc = ensure_scratch(ctx, (size_t)H * W * Cout * (MODE == 0 ? 4 : 2));
    if (rc) return rc;
    d_pooled = d_out_pk;
    d_out_pk = ctx->scratch;
    pool = 0;
  }
  const double opix = pool ? (double)((H + 1) / 2) * ((W + 1) / 2) : (double)H * W;
  const double bytes = (double)H * W * Cin * ib + opix * Cout * ((d_out_pk ? ib : 0.0) + (d_out_f32 ? 4.0 : 0.0)) + 4.0 * 9.0 * Cin * Cout;
  LaunchScope ls(ctx, name, flops, bytes);
  int rc;
  switch (plan) {
    case 0: rc = launch_sw<MODE, 5, 2, 2, 1>(ctx, d_in, d_wpk, d_bias, d_out_pk, d_out_f32, H, W, Cin, Cout, relu, pool); break;
    case 1: rc = launch_sw<MODE, 5, 2, 2, 2>(ctx, d_in, d_wpk, d_bias, d_out_pk, d_out_f32, H, W, Cin, Cout, relu, pool); break;
    case 2: rc = launch_sw<MODE, 5, 1, 1, 4>(ctx, d_in, d_wpk, d_bias, d_out_pk, d_out_f32, H, W, Cin, Cout, relu, pool); break;
    default: rc = launch_sw<MODE, 5, 1, 1, 1>(ctx, d_in, d_wpk, d_bias, d_out_pk, d_out_f32, H, W, Cin, Cout, relu, pool); break;
  }
  if (rc) return rc;
  if (d_pooled)
    hipLaunchKernelGGL(maxpool2_packed_kernel<MODE>, dim3(sw_grid_for((long)(Cout / 8) * ((H + 1) / 2) * ((W + 1) / 2))), dim3(256), 0,
                       ctx->stream, (const uint4*)d_out_pk, (uint4*)d_pooled, Cout / 8, H, W, (H + 1) / 2, (W + 1) / 2);
  return ls.finish("conv3x3_sw_kernel");
}

// Pooling MAX 2x2/2 (Caffe's ceil output size) on packed activations.  f16 / bf16: the maximum of the rounded values is the rounded
// maximum (rounding is monotonic).  bf16x3: the window's largest (hi, lo) pair in lexicographic order is copied; x -> (hi, lo)
// is monotonic for that order (hi truncates toward zero, lo rounds the remainder), so this is the split of the fp32 maximum.
template <int F16>
__global__ __launch_bounds__(256) void maxpool2_packed_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int CB, int H,
                                                              int W, int OH, int OW) {
  const long total = (long)CB * OH * OW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long p = idx;
    const int ow = (int)(p % OW);
    p /= OW;
    const int oh = (int)(p % OH);
    const int cb = (int)(p / OH);
    const int h0 = oh * 2, x0 = ow * 2;
    const int h1 = min(h0 + 1, H - 1), x1 = min(x0 + 1, W - 1);      // a clipped window repeats its last row / column
    const long base = (long)cb * H * W;
    const long q[4] = {base + (long)h0 * W + x0, base + (long)h0 * W + x1, base + (long)h1 * W + x0, base + (long)h1 * W + x1};
    if (F16 == 2) {                                        // bf16: compare as fp32 (exact widening), keep the upper halves
      uint4 m = in[q[0]];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const uint4 v = in[q[k]];
        auto mx = [](unsigned a, unsigned b) {
          const float lo = fmaxf(__uint_as_float(a << 16), __uint_as_float(b << 16));
          const float hi = fmaxf(__uint_as_float(a & 0xFFFF0000u), __uint_as_float(b & 0xFFFF0000u));
          return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
        };
        m = make_uint4(mx(m.x, v.x), mx(m.y, v.y), mx(m.z, v.z), mx(m.w, v.w));
      }
      out[idx] = m;
    } else if (F16) {
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      h8 m = __builtin_bit_cast(h8, in[q[0]]);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const h8 v = __builtin_bit_cast(h8, in[q[k]]);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
      out[idx] = __builtin_bit_cast(uint4, m);
    } else {
      unsigned short mh[8], ml[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 hi = in[q[k] * 2], lo = in[q[k] * 2 + 1];
        const unsigned hw_[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned short h = (unsigned short)(hw_[e >> 1] >> ((e & 1) * 16)), l = (unsigned short)(lw[e >> 1] >> ((e & 1) * 16));
          const float fh = __uint_as_float((unsigned)h << 16), fl = __uint_as_float((unsigned)l << 16);
          const float gh = __uint_as_float((unsigned)mh[e] << 16), gl = __uint_as_float((unsigned)ml[e] << 16);
          if (k == 0 || fh > gh || (fh == gh && fl > gl)) { mh[e] = h; ml[e] = l; }
        }
      }
      out[idx * 2] = make_uint4(mh[0] | ((unsigned)mh[1] << 16), mh[2] | ((unsigned)mh[3] << 16), mh[4] | ((unsigned)mh[5] << 16),
                                mh[6] | ((unsigned)mh[7] << 16));
      out[idx * 2 + 1] = make_uint4(ml[0] | ((unsigned)ml[1] << 16), ml[2] | ((unsigned)ml[3] << 16),
                                    ml[4] | ((unsigned)ml[5] << 16), ml[6] | ((unsigned)ml[7] << 16));
    }
  }
}

// fp32 c8 <-> packed (tests, and the boundaries of a packed chain that no producer epilogue covers)
template <int F16>
__global__ void pack_act_kernel(const float4* __restrict__ in, void* __restrict__ out, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x)
    x3_store8<F16, true>(out, i, in[2 * i], in[2 * i + 1]);
}
template <int F16>
__global__ void unpack_act_kernel(const unsigned* __restrict__ in, float4* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long pix = i >> 1;
    const int kb = (int)(i & 1);
    float4 v;
    if (F16 == 2) {
      const uint2 h = *reinterpret_cast<const uint2*>(in + pix * 4 + kb * 2);
      v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xFFFF0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xFFFF0000u));
    } else if (F16) {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const f16x4 h = __builtin_bit_cast(f16x4, *reinterpret_cast<const uint2*>(in + pix * 4 + kb * 2));
      v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
      const uint2 hi = *reinterpret_cast<const uint2*>(in + pix * 8 + kb * 2), lo = *reinterpret_cast<const uint2*>(in + pix * 8 + 4 + kb * 2);
      auto f = [](unsigned w, int k) { return __uint_as_float(k ? (w & 0xFFFF0000u) : (w << 16)); };
      v = make_float4(f(hi.x, 0) + f(lo.x, 0), f(hi.x, 1) + f(lo.x, 1), f(hi.y, 0) + f(lo.y, 0), f(hi.y, 1) + f(lo.y, 1));
    }
    out[i] = v;
  }
}


// fp32 c8 -> the mode's packed form in the context's scratch arena (an input no producer wrote packed: tests, the fp32-tensor
// entry points, the Python engine's bf16x3 graph)
static int sw_pack_input(mnc_ctx* ctx, int mode, const float* d_c8, size_t n, const void** packed) {
  const size_t bytes = n * (mode == 0 ? 4 : 2);
  (void)hipSetDevice(ctx->device);
  int rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  const long npix = (long)(n / 8);
  auto kern = mode == 0 ? pack_act_kernel<0> : mode == 1 ? pack_act_kernel<1> : pack_act_kernel<2>;
  hipLaunchKernelGGL(kern, dim3(sw_grid_for(npix)), dim3(256), 0, ctx->stream, (const float4*)d_c8, ctx->scratch, npix);
  *packed = ctx->scratch;
  return MNC_OK;
}

template <int MODE>
static int conv3x3_sw_any(mnc_ctx* ctx, const char* name, const void* d_in, int in_packed, const void* d_wpk, const float* d_bias,
                          void* d_out, int out_packed, int H, int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(ctx && d_in && d_out, "%s: null pointer", name);
  if (!in_packed) {
    MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0, "%s: unsupported shape", name);
    int rc = sw_pack_input(ctx, MODE, (const float*)d_in, (size_t)Cin * H * W, &d_in);
    if (rc) return rc;
  }
  return conv3x3_sw<MODE>(ctx, name, d_in, d_wpk, d_bias, out_packed ? d_out : nullptr, out_packed ? nullptr : (float*)d_out, H, W,
                          Cin, Cout, relu);
}

template <int F16>
static int maxpool2_packed(mnc_ctx* ctx, const char* name, const void* d_in, void* d_out, int C, int H, int W) {
  MNC_REQUIRE(ctx && d_in && d_out && C > 0 && C % 8 == 0 && H > 0 && W > 0, "%s: bad argument", name);
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const double b = (F16 ? 2.0 : 4.0) * C;
  LaunchScope ls(ctx, name, 0.0, b * ((double)H * W + (double)OH * OW));
  hipLaunchKernelGGL(maxpool2_packed_kernel<F16>, dim3(sw_grid_for((long)(C / 8) * OH * OW)), dim3(256), 0, ctx->stream,
                     (const uint4*)d_in, (uint4*)d_out, C / 8, H, W, OH, OW);
  return ls.finish("maxpool2_packed_kernel");
}

}  // namespace mnc

using namespace mnc;

extern "C" {

size_t mnc_conv3x3_lowp_weight_bytes(int mode, int Cout, int Cin) {
  if (mode < 0 || mode > 2 || Cout <= 0 || Cin <= 0) return 0;
  return (size_t)((Cin + 15) / 16) * (size_t)((Cout + 31) / 32) * sw_planes(mode) * kSwPlaneB;
}

int mnc_pack_conv3x3_lowp(mnc_ctx* ctx, int mode, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  MNC_REQUIRE(ctx && d_oihw && d_packed && mode >= 0 && mode <= 2 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_pack_conv3x3_lowp: bad argument");
  LaunchScope ls(ctx, "pack_conv3x3_lowp");
  const long total = (long)((Cin + 15) / 16) * (Cout / 32) * sw_planes(mode) * 9 * 32;
  auto kern = mode == 0 ? pack_conv_sw_kernel<0> : mode == 1 ? pack_conv_sw_kernel<1> : pack_conv_sw_kernel<2>;
  hipLaunchKernelGGL(kern, dim3(sw_grid_for(total)), dim3(256), 0, ctx->stream, d_oihw, (uint4*)d_packed, Cout, Cin);
  return ls.finish("pack_conv_sw_kernel");
}

int mnc_conv3x3_lowp(mnc_ctx* ctx, int mode, const void* d_in_packed, const void* d_w_packed, const float* d_bias, void* d_out_packed,
                     float* d_out_c8, int H, int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(mode >= 0 && mode <= 2, "mnc_conv3x3_lowp: mode must be 0 (bf16x3), 1 (f16) or 2 (bf16)");
  if (mode == 0) return conv3x3_sw<0>(ctx, "conv3x3_bf16x3", d_in_packed, d_w_packed, d_bias, d_out_packed, d_out_c8, H, W, Cin, Cout, relu);
  if (mode == 1) return conv3x3_sw<1>(ctx, "conv3x3_f16", d_in_packed, d_w_packed, d_bias, d_out_packed, d_out_c8, H, W, Cin, Cout, relu);
  return conv3x3_sw<2>(ctx, "conv3x3_bf16", d_in_packed, d_w_packed, d_bias, d_out_packed, d_out_c8, H, W, Cin, Cout, relu);
}

int mnc_conv3x3_lowp_pool(mnc_ctx* ctx, int mode, const void* d_in_packed, const void* d_w_packed, const float* d_bias,
                          void* d_out_pooled_packed, int H, int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(mode >= 0 && mode <= 2, "mnc_conv3x3_lowp_pool: mode must be 0 (bf16x3), 1 (f16) or 2 (bf16)");
  if (mode == 0) return conv3x3_sw<0>(ctx, "conv3x3_bf16x3", d_in_packed, d_w_packed, d_bias, d_out_pooled_packed, nullptr, H, W, Cin, Cout, relu, 1);
  if (mode == 1) return conv3x3_sw<1>(ctx, "conv3x3_f16", d_in_packed, d_w_packed, d_bias, d_out_pooled_packed, nullptr, H, W, Cin, Cout, relu, 1);
  return conv3x3_sw<2>(ctx, "conv3x3_bf16", d_in_packed, d_w_packed, d_bias, d_out_pooled_packed, nullptr, H, W, Cin, Cout, relu, 1);
}

// ---- the entry points of rounds 1-5, on the round-6 kernel (same names and argument meaning; the packed weight layout and its size
// are mnc_pack_conv3x3_lowp's).  fp32 inputs are packed into the context's scratch arena first -- the producer's own rounding /
// split, so the fp32-tensor route and the packed route give the same bits.
int mnc_pack_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return mnc_pack_conv3x3_lowp(ctx, 0, d_oihw, d_packed, Cout, Cin);
}
int mnc_pack_conv3x3_f16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return mnc_pack_conv3x3_lowp(ctx, 1, d_oihw, d_packed, Cout, Cin);
}
int mnc_pack_conv3x3_bf16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return mnc_pack_conv3x3_lowp(ctx, 2, d_oihw, d_packed, Cout, Cin);
}

int mnc_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W,
                       int Cin, int Cout, int relu) {
  return conv3x3_sw_any<0>(ctx, "conv3x3_bf16x3", d_in, 0, d_wpk, d_bias, d_out, 0, H, W, Cin, Cout, relu);
}
int mnc_conv3x3_f16(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                    int Cout, int relu) {
  return conv3x3_sw_any<1>(ctx, "conv3x3_f16", d_in, 0, d_wpk, d_bias, d_out, 0, H, W, Cin, Cout, relu);
}
int mnc_conv3x3_bf16(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                     int Cout, int relu) {
  return conv3x3_sw_any<2>(ctx, "conv3x3_bf16", d_in, 0, d_wpk, d_bias, d_out, 0, H, W, Cin, Cout, relu);
}
int mnc_conv3x3_bf16x3_pk(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out, int H, int W,
                          int Cin, int Cout, int relu, int in_packed, int out_packed) {
  return conv3x3_sw_any<0>(ctx, "conv3x3_bf16x3", d_in, in_packed, d_wpk, d_bias, d_out, out_packed, H, W, Cin, Cout, relu);
}
int mnc_conv3x3_f16_pk(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out, int H, int W, int Cin,
                       int Cout, int relu, int in_packed, int out_packed) {
  return conv3x3_sw_any<1>(ctx, "conv3x3_f16", d_in, in_packed, d_wpk, d_bias, d_out, out_packed, H, W, Cin, Cout, relu);
}
int mnc_conv3x3_bf16_pk(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out, int H, int W, int Cin,
                        int Cout, int relu, int in_packed, int out_packed) {
  return conv3x3_sw_any<2>(ctx, "conv3x3_bf16", d_in, in_packed, d_wpk, d_bias, d_out, out_packed, H, W, Cin, Cout, relu);
}

int mnc_maxpool2_c8_bf16x3(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W) {
  return maxpool2_packed<0>(ctx, "maxpool2_c8_bf16x3", d_in, d_out, C, H, W);
}
int mnc_maxpool2_c8_f16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W) {
  return maxpool2_packed<1>(ctx, "maxpool2_c8_f16", d_in, d_out, C, H, W);
}
int mnc_maxpool2_c8_bf16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W) {
  return maxpool2_packed<2>(ctx, "maxpool2_c8_bf16", d_in, d_out, C, H, W);
}

int mnc_act_pack(mnc_ctx* ctx, const float* d_c8, void* d_packed, size_t n, int f16) {
  MNC_REQUIRE(ctx && d_c8 && d_packed && n > 0 && n % 8 == 0 && f16 >= 0 && f16 <= 2, "mnc_act_pack: bad argument");
  LaunchScope ls(ctx, "act_pack", 0.0, (f16 ? 6.0 : 8.0) * n);
  auto kern = f16 == 0 ? pack_act_kernel<0> : f16 == 1 ? pack_act_kernel<1> : pack_act_kernel<2>;
  hipLaunchKernelGGL(kern, dim3(sw_grid_for((long)(n / 8))), dim3(256), 0, ctx->stream, (const float4*)d_c8, d_packed, (long)(n / 8));
  return ls.finish("pack_act_kernel");
}

int mnc_act_unpack(mnc_ctx* ctx, const void* d_packed, float* d_c8, size_t n, int f16) {
  MNC_REQUIRE(ctx && d_c8 && d_packed && n > 0 && n % 8 == 0 && f16 >= 0 && f16 <= 2, "mnc_act_unpack: bad argument");
  LaunchScope ls(ctx, "act_unpack", 0.0, (f16 ? 6.0 : 8.0) * n);
  auto kern = f16 == 0 ? unpack_act_kernel<0> : f16 == 1 ? unpack_act_kernel<1> : unpack_act_kernel<2>;
  hipLaunchKernelGGL(kern, dim3(sw_grid_for((long)(n / 4))), dim3(256), 0, ctx->stream, (const unsigned*)d_packed, (float4*)d_c8, (long)(n / 4));
  return ls.finish("unpack_act_kernel");
}

}  // extern "C"
