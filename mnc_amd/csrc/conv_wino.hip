// 3x3 convolution (pad 1, stride 1) by Winograd's minimal filtering F(2x2, 3x3) on the fp32 matrix pipe, for gfx950
// (models/VGG16/mnc_5stage/test.prototxt:41-412: 12 of the 13 trunk convolutions + rpn_conv_3x3).
//
// Why: the direct implicit GEMM (conv.hip) runs at 76 % of the fp32-matrix peak (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) -- the
// contraction itself is what bounds the fp32 mode.  F(2x2, 3x3) computes every 2x2 output tile from a 4x4 input tile with 16
// multiplies per (input channel, output channel) instead of 36: 2.25x fewer matrix-pipe cycles for the same result.
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          (Lavin & Gray 2015; the transforms below are the standard ones)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// All transform coefficients are 0, +-1, +-1/2: the input and output transforms are exact additions in fp32, the filter
// transform is evaluated in double and rounded once.  The products differ from the direct form's (different summation
// order, 16 partial products per tile recombined with signs), so the result agrees with the direct kernel to fp32 rounding
// (measured in tests/test_gpu_ops.py), not bit for bit.
//
// Mapping.  For each of the 16 transform positions xi the contraction over input channels is a GEMM
//     M_xi[co][tile] += U_xi[co][ci] * V_xi[ci][tile]
// run as MFMA 32x32x2: A operand = U (lane: co = lane % 32, k = lane / 32), B operand = V (lane: tile = lane % 32, k = lane / 32).
//   * a wave owns 32 output channels x 32 Winograd tiles (2 tile rows x 16 tile columns = 4 pixel rows x 32 pixel columns)
//     x all 16 positions: 16 accumulator tiles = 256 accumulator registers -> one wave per SIMD, by design; every MFMA of a
//     channel block goes to a different accumulator than its predecessor (no dependent issue);
//   * a workgroup is ROWS waves stacked vertically (4*ROWS pixel rows x 32 columns) sharing the weight panel;
//   * K is walked in 8-channel blocks as in conv.hip: per block the (4*ROWS + 2) x 34 halo and the 32 x 8 x 16 transformed
//     weights are staged global -> registers -> LDS (double-buffered, one barrier per block); each lane reads the 4x4 input
//     tile of ITS Winograd tile (its 4 channels, two at a time: 2 x 16 ds_read_b64), transforms it in registers (32 additions per
//     channel) and feeds 64 MFMAs (16 positions x 4 channel pairs), reading a position's weight fragment with ds_read_b64;
//     channel pair m of a position multiplies channels m and 4+m (lane half k supplies channel 4k+m), the packed weights use
//     the same pairing;
//   * epilogue: the output transform in registers (24 additions per output channel), + bias + ReLU, 2x2 pixels x 16 channels per
//     lane as float4 stores into the c8 layout; K-split partial sums go through the same transform (it is linear).
#include <atomic>
#include <cstdlib>

#include "mnc_internal.h"
#include "wino_common.h"

namespace mnc {

// VAR: scheduling variant (tuning; MNC_WINO_VAR): bit 0 = pinned software pipeline of the two halves of a block (below).
template <int ROWS, int VAR>
__global__ __launch_bounds__(64 * ROWS) void conv3x3_wino_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int H, int W, int Cin, int Cout, int relu, int ksplit,
                                                                  float* __restrict__ part) {
  constexpr int NT = 64 * ROWS;
  constexpr int kHaloRows = 4 * ROWS + 2;
  constexpr int kHaloFloats = kHaloRows * kWHaloCols * kWPixPitch;
  constexpr int kHaloVec = kHaloRows * kWHaloCols * 2;            // float4 items per halo
  constexpr int kHPer = (kHaloVec + NT - 1) / NT;
  constexpr int kWVec = kWPanel / 4;                              // float4 items per weight panel
  constexpr int kWPer = (kWVec + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];   // halo[2] then weights[2]
  float* const s_halo = s_mem;
  float* const s_w = s_mem + 2 * kHaloFloats;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kk = lane >> 5;
  const int ty = j >> 4, tx = j & 15;                            // this lane's Winograd tile inside the wave's 2 x 16 tiles
  const int ncot = Cout >> 5;
  const int split = blockIdx.z / ncot;
  const int cot = blockIdx.z - split * ncot;
  const int w0 = blockIdx.x * kWCols, h0 = blockIdx.y * (4 * ROWS), co0 = cot * 32;
  const int nchunks = (Cin >> 3) / ksplit;
  const int chunk0 = split * nchunks;

  // ---- staging assignment (fixed per thread), branch-free as in conv.hip ----
  int h_off[kHPer];
  int h_src[kHPer];
  unsigned h_keep[kHPer];
#pragma unroll
  for (int u = 0; u < kHPer; ++u) {
    const int q = min(tid + u * NT, kHaloVec - 1);
    const int pix = q >> 1, half = q & 1;
    const int r = pix / kWHaloCols, c = pix - r * kWHaloCols;
    const int gh = h0 - 1 + r, gw = w0 - 1 + c;
    h_off[u] = pix * kWPixPitch + half * 4;
    h_src[u] = (min(max(gh, 0), H - 1) * W + min(max(gw, 0), W - 1)) * 8 + half * 4;        // < 2^31 floats per plane
    h_keep[u] = (gh >= 0 && gh < H && gw >= 0 && gw < W) ? 0xFFFFFFFFu : 0u;
  }
  int w_idx[kWPer];
#pragma unroll
  for (int u = 0; u < kWPer; ++u) w_idx[u] = min(tid + u * NT, kWVec - 1);
  const long plane = (long)H * W * 8;

  struct Regs {
    float4 h[kHPer];
    float4 w[kWPer];
  };
  Regs G;
#pragma unroll
  for (int u = 0; u < kHPer; ++u) G.h[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < kWPer; ++u) G.w[u] = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load_chunk = [&](int c) {
    c = chunk0 + min(c, nchunks - 1);
    const float* src = in + (long)c * plane;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) G.h[u] = *reinterpret_cast<const float4*>(src + h_src[u]);
    const float4* wsrc = reinterpret_cast<const float4*>(wpk + ((long)c * ncot + cot) * kWPanel);
#pragma unroll
    for (int u = 0; u < kWPer; ++u) G.w[u] = wsrc[w_idx[u]];
  };
  auto store_chunk = [&](int buf) {
    float* hdst = s_halo + buf * kHaloFloats;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) {
      float4 v = G.h[u];
      v.x = __uint_as_float(__float_as_uint(v.x) & h_keep[u]);
      v.y = __uint_as_float(__float_as_uint(v.y) & h_keep[u]);
      v.z = __uint_as_float(__float_as_uint(v.z) & h_keep[u]);
      v.w = __uint_as_float(__float_as_uint(v.w) & h_keep[u]);
      *reinterpret_cast<float4*>(hdst + h_off[u]) = v;
    }
    float4* wdst = reinterpret_cast<float4*>(s_w + buf * kWPanel);
#pragma unroll
    for (int u = 0; u < kWPer; ++u) wdst[w_idx[u]] = G.w[u];
  };

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

  // lane-constant LDS offsets: the 4x4 input window of this lane's tile starts at halo pixel (4*wave + 2*ty, 2*tx)
  const int d_base = ((4 * wave + 2 * ty) * kWHaloCols + 2 * tx) * kWPixPitch + kk * 4;
  const int u_base = (kk * 32 + j) * kWRowPitch;

  // A block is multiplied in two channel-pair halves (the lane's channels 4*kk + {0,1}, then {2,3}; MFMA pair m multiplies
  // channels m and 4 + m across the two lane halves), software-pipelined against the single wave per SIMD:
  //   read half 0 (d, u) -> transform half 0 -> issue the reads of half 1 ->
  //   32 MFMAs of half 0 with the 64 additions of half 1's transform interleaved (1 MFMA : 2 VALU) -> 32 MFMAs of half 1.
  // hipcc on its own places every ds_read right before the MFMA that consumes it, and with one wave per SIMD nobody covers
  // that latency; the scheduling fences pin the phases, sched_group_barrier pins the interleave.
  // ablation builds (tuning only, wrong results): VAR & 16 no staging inside the loop, & 32 no barrier either, & 64 no LDS
  // reads (operands are opaque register constants), & 128 no input transform
  auto read_d = [&](const float* sh, float2 (&d)[4][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (VAR & 64) {
          d[r][c] = make_float2(1.f, 2.f);
          asm volatile("" : "+v"(d[r][c].x), "+v"(d[r][c].y));
        } else {
          d[r][c] = *reinterpret_cast<const float2*>(sh + (r * kWHaloCols + c) * kWPixPitch);
        }
      }
  };
  auto read_u = [&](const float* sw, float2 (&u)[16]) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      if (VAR & 64) {
        u[p] = make_float2(3.f, 4.f);
        asm volatile("" : "+v"(u[p].x), "+v"(u[p].y));
      } else {
        u[p] = *reinterpret_cast<const float2*>(sw + p * 4);
      }
    }
  };
  auto transform = [&](const float2 (&d)[4][4], float2 (&v)[16]) {
    if (VAR & 128) {
#pragma unroll
      for (int p = 0; p < 16; ++p) v[p] = d[p >> 2][p & 3];
      return;
    }
    float2 t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {                                 // t = B^T d (rows)
      t[0][c] = make_float2(d[0][c].x - d[2][c].x, d[0][c].y - d[2][c].y);
      t[1][c] = make_float2(d[1][c].x + d[2][c].x, d[1][c].y + d[2][c].y);
      t[2][c] = make_float2(d[2][c].x - d[1][c].x, d[2][c].y - d[1][c].y);
      t[3][c] = make_float2(d[1][c].x - d[3][c].x, d[1][c].y - d[3][c].y);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                 // V = t B (columns)
      const float2 a = t[r][0], b = t[r][1], c = t[r][2], e = t[r][3];
      v[r * 4 + 0] = make_float2(a.x - c.x, a.y - c.y);
      v[r * 4 + 1] = make_float2(b.x + c.x, b.y + c.y);
      v[r * 4 + 2] = make_float2(c.x - b.x, c.y - b.y);
      v[r * 4 + 3] = make_float2(b.x - e.x, b.y - e.y);
    }
  };
  auto mfma_half = [&](const float2 (&u)[16], const float2 (&v)[16]) {
    // positions in pairs: consecutive MFMAs alternate between the two positions' accumulators
#pragma unroll
    for (int p = 0; p < 16; p += 2) {
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[p].x, v[p].x, acc[p], 0, 0, 0);
      acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[p + 1].x, v[p + 1].x, acc[p + 1], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[p].y, v[p].y, acc[p], 0, 0, 0);
      acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[p + 1].y, v[p + 1].y, acc[p + 1], 0, 0, 0);
    }
  };
  auto multiply = [&](int buf) {
    const float* sh = s_halo + buf * kHaloFloats + d_base;
    const float* sw = s_w + buf * kWPanel + u_base;
    float2 d0[4][4], d1[4][4], u0[16], u1[16], v0[16], v1[16];
    read_d(sh, d0);
    read_u(sw, u0);
    transform(d0, v0);
    if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
    read_d(sh + 2, d1);
    read_u(sw + 2, u1);
    if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
    mfma_half(u0, v0);
    transform(d1, v1);
    if (VAR & 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);        // 2 VALU (half 1's transform)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_half(u1, v1);
  };

  // One staging register set, T14-style: the registers always hold the block AFTER the one in LDS.  Per iteration: barrier
  // (block c complete in LDS[buf], LDS[buf^1] free) -> write the registers (block c+1) to LDS[buf^1] -> re-issue the loads for
  // block c+2 at once -> multiply block c.  The loads have a whole multiply (>= 4096 matrix-pipe cycles) to land before the
  // next iteration's write; the scheduling fence keeps hipcc from sinking them below the MFMAs (it does, to shorten live
  // ranges -- and then every block waits for HBM/L2 at its end).
  load_chunk(0);
  store_chunk(0);
  load_chunk(1);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (!(VAR & 32)) __syncthreads();
    if (!(VAR & 16)) {
      store_chunk(buf ^ 1);         // block c+1 (behind the last block: a clamped duplicate into the idle buffer, harmless)
      load_chunk(c + 2);            // unconditional (clamped): no load under a branch
    }
    __builtin_amdgcn_sched_barrier(0);
    multiply(buf);
  }

  // ---- epilogue: Y = A^T M A per output channel; acc[p][e]: p = 4*xi_row + xi_col, e -> channel (e&3) + 8*(e>>2) + 4*kk ----
  const int oy = h0 + 4 * wave + 2 * ty, ox = w0 + 2 * tx;
  float* dst = out;
  if (ksplit > 1) dst = part + (long)split * Cout * H * W;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int co = co0 + g * 8 + kk * 4;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ksplit == 1) b = *reinterpret_cast<const float4*>(bias + co);
    float y[2][2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = 4 * g + q;
      float s0[4], s1[4];                         // s = A^T M: rows of the 2 x 4 intermediate
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        s0[cc] = acc[0 + cc][e] + acc[4 + cc][e] + acc[8 + cc][e];
        s1[cc] = acc[4 + cc][e] - acc[8 + cc][e] - acc[12 + cc][e];
      }
      y[0][0][q] = s0[0] + s0[1] + s0[2];
      y[0][1][q] = s0[1] - s0[2] - s0[3];
      y[1][0][q] = s1[0] + s1[1] + s1[2];
      y[1][1][q] = s1[1] - s1[2] - s1[3];
    }
    const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = oy + dy, xx = ox + dx;
        if (yy < H && xx < W) {
          float4 o = make_float4(y[dy][dx][0] + bb[0], y[dy][dx][1] + bb[1], y[dy][dx][2] + bb[2], y[dy][dx][3] + bb[3]);
          if (relu && ksplit == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * H + yy) * W + xx) * 8 + kk * 4) = o;
        }
      }
  }
}

// ---- v2: the 16 transform positions of a tile block split over a PAIR of waves ------------------------------------------
// v1 spends 256 registers on accumulators, so one wave per SIMD and one workgroup per CU: nothing covers a workgroup's
// prologue / epilogue, its staging, its LDS reads or its input transform (ablations on conv2_2: 315 us -> 217 us with all of
// them removed; ideal matrix-pipe time 156 us).  Here wave (rg, hf) owns the 32 tiles of row group rg and transform rows
// 2*hf, 2*hf + 1 (8 of the 16 positions): 128 accumulator registers, two waves per SIMD, two workgroups per CU -- one wave's
// LDS reads / transform / staging / epilogue run under its SIMD partner's MFMAs.
//   * input transform per wave: only the two rows of t = B^T d it needs (hf = 0: d0-d2, d1+d2; hf = 1: d2-d1, d1-d3) from three
//     of the four input rows, then the column pass: 64 additions for the lane's 4 channels instead of 128;
//   * per block: 12 + 8 ds_read_b128, 64 VALU, 32 MFMAs per wave;
//   * epilogue: the output transform is linear, so each wave applies it to its own rows (hf = 0: s0 = M0 + M1, s1 = M1;
//     hf = 1: s0 = M2, s1 = -M2 - M3) and the pair's two partial 2x2 outputs are added through LDS (the staging buffers are
//     free by then): hf = 1 writes 64 floats per lane, hf = 0 adds bias / ReLU and stores.
//   * DMA != 0: the weight panel of a block (17 KB, stored in global memory exactly as it sits in LDS) is copied by LDS-DMA
//     (global_load_lds_dwordx4: 1 KB per wave instruction, no staging registers, no ds_write pass); only the halo -- which
//     needs the out-of-image mask and the padded pixel pitch -- goes through registers.
template <int RG, int ABL = 0, int DMA = 0, int XCD = 1>
__global__ __launch_bounds__(128 * RG, RG <= 2 ? 2 : 1) void conv3x3_wino2_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int H, int W, int Cin, int Cout, int relu, int ksplit_a,
                                                                     float* __restrict__ part, int tiles_x, int pool_a,
                                                                     int pix_a, int ksplit_b) {
  constexpr int NT = 128 * RG;
  constexpr int kHaloRows = 4 * RG + 2;
  constexpr int kHaloFloats = kHaloRows * kWHaloCols * kWPixPitch;
  constexpr int kHaloVec = kHaloRows * kWHaloCols * 2;
  constexpr int kHPer = (kHaloVec + NT - 1) / NT;
  constexpr int kWVec = kWPanel / 4;
  constexpr int kWPer = (kWVec + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];   // halo[2] then weights[2]; the epilogue reuses it
  float* const s_halo = s_mem;
  float* const s_w = s_mem + 2 * kHaloFloats;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;     // scalar: the hf branches below are uniform
  const int rg = wave % RG, hf = wave / RG;                      // waves w and w + RG are a pair (same tiles, other positions)
  const int j = lane & 31, kk = lane >> 5;
  const int ty = j >> 4, tx = j & 15;
  const int ncot = Cout >> 5;
  // XCD-aware block order (1-D grid).  The dispatcher puts block b on XCD b % 8 (used for speed only).  Blocks are re-numbered
  // so that every XCD gets a CONTIGUOUS range of the logical order (output-channel tile fastest, then K split, then pixel tile):
  // the Cout/32 workgroups that read the same input halo then run on ONE XCD at about the same time and the halo is fetched
  // into that XCD's L2 once instead of once per channel tile from Infinity Cache / HBM.  Bijective for any block count.
  // Two SECTIONS of the grid (the launcher's tail plan, wino_impl): blocks [0, pix_a * ncot * ksplit_a) are the pixel tiles
  // [0, pix_a) with ksplit_a K ranges each, the blocks behind them the remaining pixel tiles with ksplit_b ranges each -- the tiles
  // of a last, partly filled round of workgroups are cut into shorter pieces.  A tile with one range writes the finished output
  // (bias, ReLU, pooling); with several, each range writes raw partial sums to its plane of `part` (wino_section_reduce_kernel
  // finishes them).  K ranges may be uneven: range k of s covers blocks [k * nb / s, (k + 1) * nb / s).
  int bz, bx, by, ksplit;
  {
    const int n_a = pix_a * ncot * ksplit_a;
    int b = blockIdx.x, total = n_a, pix0 = 0;
    ksplit = ksplit_a;
    if (b >= n_a) { b -= n_a; total = gridDim.x - n_a; pix0 = pix_a; ksplit = ksplit_b; }
    const int q = total >> 3, r = total & 7, xcd = b & 7, idx = b >> 3;
    const int logical = XCD ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : b;
    const int nz = ncot * ksplit;
    bz = logical % nz;
    const int rest = pix0 + logical / nz;
    bx = rest % tiles_x;
    by = rest / tiles_x;
  }
  const int pool = ksplit == 1 ? pool_a : 0;
  const int split = bz / ncot;
  const int cot = bz - split * ncot;
  const int w0 = bx * kWCols, h0 = by * (4 * RG), co0 = cot * 32;
  const int chunk0 = split * (Cin >> 3) / ksplit;
  const int nchunks = (split + 1) * (Cin >> 3) / ksplit - chunk0;

  int h_off[kHPer];
  int h_src[kHPer];
  unsigned h_keep[kHPer];
#pragma unroll
  for (int u = 0; u < kHPer; ++u) {
    const int q = min(tid + u * NT, kHaloVec - 1);
    const int pix = q >> 1, half = q & 1;
    const int r = pix / kWHaloCols, c = pix - r * kWHaloCols;
    const int gh = h0 - 1 + r, gw = w0 - 1 + c;
    // ABL & 1: the two 16-byte halves of a pixel are swapped in halo rows 2, 3, 6, 7, ... ((row >> 1) odd).  A halo read is a
    // ds_read_b128 whose 16-lane groups hold tile columns of BOTH tile rows of the wave (rows R and R + 2); with a pixel pitch
    // of 12 dwords a tile column step is 24 dwords, so the bank of every lane starts on a multiple of 8 and each group's 16
    // lanes x 4 banks fold onto 32 of the 64 banks: 2-way conflicts on every read (SQ_LDS_BANK_CONFLICT = 39 % of the LDS cycles).
    // With the swap the lanes of row R + 2 start 4 banks off those of row R: conflict-free.
    h_off[u] = pix * kWPixPitch + (((ABL & 1) ? half ^ ((r >> 1) & 1) : half)) * 4;
    h_src[u] = (min(max(gh, 0), H - 1) * W + min(max(gw, 0), W - 1)) * 8 + half * 4;
    h_keep[u] = (gh >= 0 && gh < H && gw >= 0 && gw < W) ? 0xFFFFFFFFu : 0u;
  }
  int w_idx[kWPer];
#pragma unroll
  for (int u = 0; u < kWPer; ++u) w_idx[u] = min(tid + u * NT, kWVec - 1);
  const long plane = (long)H * W * 8;

  struct Regs {
    float4 h[kHPer];
    float4 w[kWPer];
  };
  Regs G;
#pragma unroll
  for (int u = 0; u < kHPer; ++u) G.h[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < kWPer; ++u) G.w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  // ABL & 1: buffer loads.  One descriptor per operand (wave-uniform kernel arguments), the block's offset in the scalar
  // soffset, the lane's part in a 32-bit voffset: no 64-bit address arithmetic, and a halo element outside the image carries an
  // out-of-range voffset, which the hardware answers with zeros -- no keep masks (3 registers, 12 v_and per block).
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)min((long)Cin * H * W * 4, 0x7FFFFFFFL), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, (int)min((long)(Cin >> 3) * ncot * kWPanel * 4, 0x7FFFFFFFL), 0x00020000);
  int hb_off[kHPer];                                              // byte voffsets of the halo pieces (out of range: 0x7FFFFFF0)
#pragma unroll
  for (int u = 0; u < kHPer; ++u) hb_off[u] = h_keep[u] ? h_src[u] * 4 : 0x7FFFFFF0;
  const int wb_off = tid * 16, wb_last = min(tid, kWVec - 1 - (kWPer - 1) * NT) * 16;
  auto load_chunk = [&](int c) {
    c = chunk0 + min(c, nchunks - 1);
    if (ABL & 1) {
      const int hs = __builtin_amdgcn_readfirstlane(c * (int)(plane * 4));
      const int ws = __builtin_amdgcn_readfirstlane((c * ncot + cot) * (kWPanel * 4));
#pragma unroll
      for (int u = 0; u < kHPer; ++u) {
        const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, hb_off[u], hs, 0);
        G.h[u] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
      }
      if (!DMA) {
#pragma unroll
        for (int u = 0; u < kWPer; ++u) {
          const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, u == kWPer - 1 ? wb_last : wb_off, ws + u * NT * 16, 0);
          G.w[u] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
        }
      }
      return;
    }
    const float* src = in + (long)c * plane;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) G.h[u] = *reinterpret_cast<const float4*>(src + h_src[u]);
    if (!DMA) {
      const float4* wsrc = reinterpret_cast<const float4*>(wpk + ((long)c * ncot + cot) * kWPanel);
#pragma unroll
      for (int u = 0; u < kWPer; ++u) G.w[u] = wsrc[w_idx[u]];
    }
  };
  // weight panel of block c -> s_w[buf] by LDS-DMA: kWPanel floats = 17 pieces of 1 KB, piece i issued by wave i % waves
  auto dma_panel = [&](int c, int buf) {
    c = chunk0 + min(c, nchunks - 1);
    const float* src = wpk + ((long)c * ncot + cot) * kWPanel;
    float* dstw = s_w + buf * kWPanel;
#pragma unroll
    for (int i = 0; i < (kWPanel / 256 + 2 * RG - 1) / (2 * RG); ++i) {
      const int piece = min(wave + i * 2 * RG, kWPanel / 256 - 1);   // branch-free: the waves without a last piece repeat piece 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dstw + piece * 256), 16, 0, 0);
    }
  };
  auto store_chunk = [&](int buf) {
    float* hdst = s_halo + buf * kHaloFloats;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) {
      float4 v = G.h[u];
      if (ABL & 1) {
        *reinterpret_cast<float4*>(hdst + h_off[u]) = v;
        continue;
      }
      v.x = __uint_as_float(__float_as_uint(v.x) & h_keep[u]);
      v.y = __uint_as_float(__float_as_uint(v.y) & h_keep[u]);
      v.z = __uint_as_float(__float_as_uint(v.z) & h_keep[u]);
      v.w = __uint_as_float(__float_as_uint(v.w) & h_keep[u]);
      *reinterpret_cast<float4*>(hdst + h_off[u]) = v;
    }
    if (!DMA) {
      float4* wdst = reinterpret_cast<float4*>(s_w + buf * kWPanel);
      if (ABL & 1) {
        char* wb = reinterpret_cast<char*>(wdst);
#pragma unroll
        for (int u = 0; u < kWPer; ++u)
          *reinterpret_cast<float4*>(wb + (u == kWPer - 1 ? wb_last : wb_off) + u * NT * 16) = G.w[u];
        return;
      }
#pragma unroll
      for (int u = 0; u < kWPer; ++u) wdst[w_idx[u]] = G.w[u];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

  // input rows this wave reads: hf = 0 -> d0, d1, d2; hf = 1 -> d1, d2, d3 (three consecutive halo rows from row hf)
  const int d_base = ((4 * rg + 2 * ty + hf) * kWHaloCols + 2 * tx) * kWPixPitch + kk * 4;
  const int u_base = (kk * 32 + j) * kWRowPitch + hf * 32;       // positions 8*hf .. 8*hf + 7

  auto multiply = [&](int buf) {
    const float* sh = s_halo + buf * kHaloFloats + d_base;
    const float* sw = s_w + buf * kWPanel + u_base;
    float4 t0[4], t1[4];                                          // the wave's two rows of t = B^T d
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a, b, e;
      if (ABL & 64) {                                             // ablation: operands are opaque register constants
        a = make_float4(1.f, 2.f, 3.f, 4.f);
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
        b = a; e = a;
      } else {
        a = *reinterpret_cast<const float4*>(sh + (0 * kWHaloCols + c) * kWPixPitch);
        b = *reinterpret_cast<const float4*>(sh + (1 * kWHaloCols + c) * kWPixPitch);
        e = *reinterpret_cast<const float4*>(sh + (2 * kWHaloCols + c) * kWPixPitch);
      }
      // hf = 0: (a, b, e) = (d0, d1, d2): t0 = d0 - d2, t1 = d1 + d2;   hf = 1: (a, b, e) = (d1, d2, d3): t2 = d2 - d1, t3 = d1 - d3
      if (hf == 0) {
        t0[c] = make_float4(a.x - e.x, a.y - e.y, a.z - e.z, a.w - e.w);
        t1[c] = make_float4(b.x + e.x, b.y + e.y, b.z + e.z, b.w + e.w);
      } else {
        t0[c] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, b.w - a.w);
        t1[c] = make_float4(a.x - e.x, a.y - e.y, a.z - e.z, a.w - e.w);
      }
    }
    float4 v[8];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float4 a = r ? t1[0] : t0[0], b = r ? t1[1] : t0[1], c = r ? t1[2] : t0[2], e = r ? t1[3] : t0[3];
      v[r * 4 + 0] = make_float4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w);
      v[r * 4 + 1] = make_float4(b.x + c.x, b.y + c.y, b.z + c.z, b.w + c.w);
      v[r * 4 + 2] = make_float4(c.x - b.x, c.y - b.y, c.z - b.z, c.w - b.w);
      v[r * 4 + 3] = make_float4(b.x - e.x, b.y - e.y, b.z - e.z, b.w - e.w);
    }
#pragma unroll
    for (int p = 0; p < 8; p += 2) {
      float4 u0, u1;
      if (ABL & 64) {
        u0 = make_float4(1.f, 2.f, 3.f, 4.f);
        asm volatile("" : "+v"(u0.x), "+v"(u0.y), "+v"(u0.z), "+v"(u0.w));
        u1 = u0;
      } else {
        u0 = *reinterpret_cast<const float4*>(sw + p * 4);
        u1 = *reinterpret_cast<const float4*>(sw + p * 4 + 4);
      }
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.x, v[p].x, acc[p], 0, 0, 0);
      acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.x, v[p + 1].x, acc[p + 1], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.y, v[p].y, acc[p], 0, 0, 0);
      acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.y, v[p + 1].y, acc[p + 1], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.z, v[p].z, acc[p], 0, 0, 0);
      acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.z, v[p + 1].z, acc[p + 1], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.w, v[p].w, acc[p], 0, 0, 0);
      acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.w, v[p + 1].w, acc[p + 1], 0, 0, 0);
    }
  };

  // ---- ABL & 1: the same block as ONE basic block with its LDS reads up front.  multiply() above branches on hf inside the
  // column loop: four basic blocks, each opening with three ds_reads and a full LDS round trip before its eight additions, and
  // the MFMAs wait behind all four.  Here the wave's three halo rows are named by ROLE instead of by position --
  //   hf = 0: (A, B, C) = (d0, d1, d2): t0 = A - C = d0 - d2, t1 = C + B = d1 + d2
  //   hf = 1: (A, B, C) = (d2, d3, d1): t0 = A - C = d2 - d1, t1 = C - B = d1 - d3
  // -- i.e. t0 = A - C and t1 = fma(sgn, B, C) with a scalar sgn = +-1 (one rounding, the same value as the add / subtract) and
  // per-wave scalar row offsets: no branch.  All 12 halo reads and the first pair of weight fragments are issued right behind
  // the barrier, before anything else of the block.
  const int rowA = hf ? 2 : 0, rowB = hf ? 3 : 1, rowC = hf ? 1 : 2;
  const float sgn = hf ? -1.f : 1.f;
  const int d_row0 = ((4 * rg + 2 * ty) * kWHaloCols + 2 * tx) * kWPixPitch;
  auto row_off = [&](int row) {     // halo row 4 rg + 2 ty + row, this lane's channel half (swapped where (halo row >> 1) is odd)
    return d_row0 + row * kWHaloCols * kWPixPitch + (kk ^ ((ty + (row >> 1)) & 1)) * 4;
  };
  const int offA = row_off(rowA), offB = row_off(rowB), offC = row_off(rowC);
  // (ext_vector_type operands: hipcc lowers their arithmetic to v_pk_add_f32 / v_pk_fma_f32 on the register pairs the LDS reads
  // delivered -- 32 VALU instructions per block; on float4 structs its SLP pass pairs elements of different reads and pays
  // for every packed operation with register moves)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 sgn4 = {sgn, sgn, sgn, sgn};
  f32x4 fA[4], fB[4], fC[4], fu0, fu1;
  auto read_block = [&](int buf) {
    const float* sh = s_halo + buf * kHaloFloats;
    const float* sw = s_w + buf * kWPanel + u_base;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fA[c] = *reinterpret_cast<const f32x4*>(sh + offA + c * kWPixPitch);
      fC[c] = *reinterpret_cast<const f32x4*>(sh + offC + c * kWPixPitch);
    }
    fu0 = *reinterpret_cast<const f32x4*>(sw);
    fu1 = *reinterpret_cast<const f32x4*>(sw + 4);
  };
  // One block in five pinned segments (sched_barrier between them, sched_group_barrier inside): (1) t0 = A - C and the first
  // transform row -- all that stands between the LDS reads and the first MFMA; (2) row B is requested, MFMAs of positions 0-1
  // with the second row's arithmetic under the later ones; (3) positions 2-3 with the LDS stores of block c + 1 (loaded an
  // iteration ago) and then the loads of block c + 2 into the same registers; (4) positions 4-5; (5) positions 6-7.  The weight
  // fragments of the next segment are requested at the head of each.
  auto mfma_pair = [&](int p, const f32x4& u0, const f32x4& u1, const f32x4 (&v)[8]) {
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.x, v[p].x, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.x, v[p + 1].x, acc[p + 1], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.y, v[p].y, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.y, v[p + 1].y, acc[p + 1], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.z, v[p].z, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.z, v[p + 1].z, acc[p + 1], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.w, v[p].w, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.w, v[p + 1].w, acc[p + 1], 0, 0, 0);
  };
  auto transform_row = [&](const f32x4 (&t)[4], f32x4* v) {
    v[0] = t[0] - t[2];
    v[1] = t[1] + t[2];
    v[2] = t[2] - t[1];
    v[3] = t[1] - t[3];
  };
  auto multiply_flat = [&](int buf, int c_next) {
    const float* sw = s_w + buf * kWPanel + u_base;
    f32x4 t0[4], t1[4], v[8];
    // (1)
#pragma unroll
    for (int c = 0; c < 4; ++c) t0[c] = fA[c] - fC[c];
    transform_row(t0, v);
    __builtin_amdgcn_sched_barrier(0);
    // (2)  (row B arrives in the registers row A has just left: with all three rows in flight at once the wave is 14 registers
    // short of its 128 + 128)
    {
      const float* sh = s_halo + buf * kHaloFloats;
#pragma unroll
      for (int c = 0; c < 4; ++c) fB[c] = *reinterpret_cast<const f32x4*>(sh + offB + c * kWPixPitch);
    }
    f32x4 n0 = *reinterpret_cast<const f32x4*>(sw + 8), n1 = *reinterpret_cast<const f32x4*>(sw + 12);
#pragma unroll
    for (int c = 0; c < 4; ++c) t1[c] = __builtin_elementwise_fma(sgn4, fB[c], fC[c]);
    transform_row(t1, v + 4);
    mfma_pair(0, fu0, fu1, v);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (3)
    f32x4 m0 = *reinterpret_cast<const f32x4*>(sw + 16), m1 = *reinterpret_cast<const f32x4*>(sw + 20);
    if (!(ABL & 16)) {
      store_chunk(buf ^ 1);
      load_chunk(c_next);
    }
    mfma_pair(2, n0, n1, v);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (4)
    n0 = *reinterpret_cast<const f32x4*>(sw + 24);
    n1 = *reinterpret_cast<const f32x4*>(sw + 28);
    mfma_pair(4, m0, m1, v);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    // (5)
    mfma_pair(6, n0, n1, v);
  };

  // ---- ABL & 2 (with ABL & 1 and DMA): the loop rotated so that the barrier sits in the MIDDLE of a block's MFMAs.  Every LDS read
  // of block c is issued before the barrier of iteration c (the fragments of positions 4-7 wait in registers), so behind it
  // buffer c % 2 is free and block c + 1 is complete in the other: the wave requests block c + 2 (halo into registers, weight
  // panel by LDS-DMA straight into buffer c % 2), reads block c + 1's rows A and C and builds its first transform row UNDER the
  // MFMAs of positions 4-7.  The next iteration opens with MFMAs: nothing but the barrier itself is exposed.  The DMA that made
  // the plain loop slower (hipcc drains it with vmcnt(0) in front of every barrier, right after it was issued) is issued right
  // BEHIND a barrier here and has 24 MFMAs to land; the halo registers are the only staging registers left.
  if constexpr ((ABL & 2) != 0) {
    static_assert(DMA == 1 && (ABL & 1), "rotated Winograd loop: LDS-DMA weight panel, flat block");
    auto read_AC = [&](int buf) {
      const float* sh = s_halo + buf * kHaloFloats;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        fA[c] = *reinterpret_cast<const f32x4*>(sh + offA + c * kWPixPitch);
        fC[c] = *reinterpret_cast<const f32x4*>(sh + offC + c * kWPixPitch);
      }
      fu0 = *reinterpret_cast<const f32x4*>(s_w + buf * kWPanel + u_base);
      fu1 = *reinterpret_cast<const f32x4*>(s_w + buf * kWPanel + u_base + 4);
    };
    f32x4 v[8], t0[4], t1[4];
    // prologue: blocks 0 and 1 requested together -- ONE global round trip in front of the first MFMA, not two (the halo of
    // block 1 waits in a second register set that only lives here)
    load_chunk(0);
    dma_panel(0, 0);
    const Regs G0 = G;
    load_chunk(1);
    dma_panel(1, 1);
    const Regs G1 = G;
    G = G0;
    store_chunk(0);
    G = G1;
    __syncthreads();
    read_AC(0);
    if constexpr ((ABL & 4) != 0) {
      // ABL & 4: ALL of block c + 1's halo rows are read behind the barrier (the LDS-DMA freed 20 staging registers), both
      // rows of t = B^T d are built under the MFMAs of positions 6-7 and the two rows of the column pass under positions 6-7 and
      // 0-1: an iteration opens with MFMAs whose operands are in registers, not with an LDS round trip for row B.
#pragma unroll
      for (int q = 0; q < 4; ++q) fB[q] = *reinterpret_cast<const f32x4*>(s_halo + offB + q * kWPixPitch);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        t0[q] = fA[q] - fC[q];
        t1[q] = __builtin_elementwise_fma(sgn4, fB[q], fC[q]);
      }
      transform_row(t0, v);
      for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const float* sw = s_w + buf * kWPanel + u_base;
        __builtin_amdgcn_sched_barrier(0);
        // positions 0-1, the second row of the column pass under them
        f32x4 n0 = *reinterpret_cast<const f32x4*>(sw + 8), n1 = *reinterpret_cast<const f32x4*>(sw + 12);
        transform_row(t1, v + 4);
        mfma_pair(0, fu0, fu1, v);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // (first the MFMA: a read in front of it would make the
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        //  loop-carried fragments wait for an LDS round trip)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // positions 2-3; the last fragments of block c in, block c + 1's halo out
        f32x4 m0 = *reinterpret_cast<const f32x4*>(sw + 16), m1 = *reinterpret_cast<const f32x4*>(sw + 20);
        f32x4 q0 = *reinterpret_cast<const f32x4*>(sw + 24), q1 = *reinterpret_cast<const f32x4*>(sw + 28);
        store_chunk(buf ^ 1);
        mfma_pair(2, n0, n1, v);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // positions 4-5: block c + 1's rows and first fragments requested behind the first MFMA, then block c + 2 from memory
        read_AC(buf ^ 1);
        {
          const float* sh = s_halo + (buf ^ 1) * kHaloFloats;
#pragma unroll
          for (int q = 0; q < 4; ++q) fB[q] = *reinterpret_cast<const f32x4*>(sh + offB + q * kWPixPitch);
        }
        load_chunk(c + 2);
        dma_panel(c + 2, buf);
        mfma_pair(4, m0, m1, v);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        // positions 6-7, block c + 1's transform rows and the first row of its column pass under them
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          t0[q] = fA[q] - fC[q];
          t1[q] = __builtin_elementwise_fma(sgn4, fB[q], fC[q]);
        }
        transform_row(t0, v);               // (v[0..3] were last read by the MFMAs of positions 2-3)
        mfma_pair(6, q0, q1, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
      }
      __syncthreads();
    } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) t0[c] = fA[c] - fC[c];
    transform_row(t0, v);
    for (int c = 0; c < nchunks; ++c) {
      const int buf = c & 1;
      const float* sh = s_halo + buf * kHaloFloats;
      const float* sw = s_w + buf * kWPanel + u_base;
      __builtin_amdgcn_sched_barrier(0);
      // positions 0-1; row B in, second transform row under the MFMAs
#pragma unroll
      for (int q = 0; q < 4; ++q) fB[q] = *reinterpret_cast<const f32x4*>(sh + offB + q * kWPixPitch);
      f32x4 n0 = *reinterpret_cast<const f32x4*>(sw + 8), n1 = *reinterpret_cast<const f32x4*>(sw + 12);
#pragma unroll
      for (int q = 0; q < 4; ++q) t1[q] = __builtin_elementwise_fma(sgn4, fB[q], fC[q]);
      transform_row(t1, v + 4);
      mfma_pair(0, fu0, fu1, v);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // positions 2-3; the last fragments of block c in, block c + 1's halo out
      f32x4 m0 = *reinterpret_cast<const f32x4*>(sw + 16), m1 = *reinterpret_cast<const f32x4*>(sw + 20);
      f32x4 q0 = *reinterpret_cast<const f32x4*>(sw + 24), q1 = *reinterpret_cast<const f32x4*>(sw + 28);
      store_chunk(buf ^ 1);
      mfma_pair(2, n0, n1, v);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // positions 4-5; block c + 2 requested, block c + 1's rows A, C in, its first transform row under the MFMAs
      load_chunk(c + 2);
      dma_panel(c + 2, buf);
      read_AC(buf ^ 1);
      mfma_pair(4, m0, m1, v);
      __builtin_amdgcn_sched_group_barrier(0x020, kHPer + (kWPanel / 256 + 2 * RG - 1) / (2 * RG), 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      // positions 6-7
#pragma unroll
      for (int q = 0; q < 4; ++q) t0[q] = fA[q] - fC[q];
      mfma_pair(6, q0, q1, v);
      __builtin_amdgcn_sched_barrier(0);
      transform_row(t0, v);                 // (v[0..3] were last read by the MFMAs of positions 2-3)
    }
    __syncthreads();
    }
  } else {
  load_chunk(0);
  if (DMA) dma_panel(0, 0);
  store_chunk(0);
  load_chunk(1);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (!(ABL & 32)) __syncthreads();      // with a DMA in flight hipcc puts s_waitcnt vmcnt(0) in front: block c's panel has landed
    if (ABL & 1) {
      read_block(buf);
      __builtin_amdgcn_sched_barrier(0);
      multiply_flat(buf, c + 2);
      continue;
    }
    if (!(ABL & 16)) {
      if (DMA) dma_panel(c + 1, buf ^ 1);  // s_w[buf^1] is free: everybody is past the barrier, i.e. done with block c-1
      store_chunk(buf ^ 1);
      load_chunk(c + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    multiply(buf);
  }
  __syncthreads();                                               // every wave is done with the staging buffers
  }

  // ---- epilogue.  acc[p][e], p = 4*r + c with r the wave's local row (global row 2*hf + r), e -> channel (e&3) + 8*(e>>2) + 4*kk.
  // Partial output transform of this wave's rows:  hf = 0: s0 = M0 + M1, s1 = M1;   hf = 1: s0 = M2, s1 = -(M2 + M3).
  float* xch = s_mem + ((long)rg * 64 * 64);                      // [value 0..63][lane] floats per row group
  const int oy = h0 + 4 * rg + 2 * ty, ox = w0 + 2 * tx;
  float* dst = out;
  if (ksplit > 1) dst = part + (long)split * Cout * H * W;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int co = co0 + g * 8 + kk * 4;
    float y[2][2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = 4 * g + q;
      float s0[4], s1[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float m0 = acc[cc][e], m1 = acc[4 + cc][e];
        s0[cc] = hf ? m0 : m0 + m1;
        s1[cc] = hf ? -(m0 + m1) : m1;
      }
      y[0][0][q] = s0[0] + s0[1] + s0[2];
      y[0][1][q] = s0[1] - s0[2] - s0[3];
      y[1][0][q] = s1[0] + s1[1] + s1[2];
      y[1][1][q] = s1[1] - s1[2] - s1[3];
    }
    if (hf == 1) {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
          for (int q = 0; q < 4; ++q) xch[(((g * 2 + dy) * 2 + dx) * 4 + q) * 64 + lane] = y[dy][dx][q];
    }
    __syncthreads();                                             // (uniform: every wave runs all four g iterations)
    if (hf == 0) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ksplit == 1) b = *reinterpret_cast<const float4*>(bias + co);
      const float bb[4] = {b.x, b.y, b.z, b.w};
      // pool != 0 (ksplit == 1 only): the following Pooling MAX 2x2 stride 2 (test.prototxt:69-79, ...) is applied here -- a
      // Winograd tile IS a pooling window (tile origins are even), so the pooled value is the maximum of the lane's own 2x2
      // outputs that lie inside the image (Caffe's ceil rule: the last window of an odd-sized map is clipped), written to
      // [C/8][OH][OW][8]; the full-resolution tensor never reaches HBM.
      float pmax[4] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (y[dy][dx][q] + xch[(((g * 2 + dy) * 2 + dx) * 4 + q) * 64 + lane]) + bb[q];
          const int yy = oy + dy, xx = ox + dx;
          if (yy < H && xx < W) {
            float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            if (relu && ksplit == 1) { ov.x = fmaxf(ov.x, 0.f); ov.y = fmaxf(ov.y, 0.f); ov.z = fmaxf(ov.z, 0.f); ov.w = fmaxf(ov.w, 0.f); }
            if (pool) {
              pmax[0] = fmaxf(pmax[0], ov.x); pmax[1] = fmaxf(pmax[1], ov.y);
              pmax[2] = fmaxf(pmax[2], ov.z); pmax[3] = fmaxf(pmax[3], ov.w);
            } else {
              *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * H + yy) * W + xx) * 8 + kk * 4) = ov;
            }
          }
        }
      if (pool && oy < H && ox < W) {
        const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;             // ceil((n - 2) / 2) + 1 for n >= 2
        *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * OH + (oy >> 1)) * OW + (ox >> 1)) * 8 + kk * 4) =
            make_float4(pmax[0], pmax[1], pmax[2], pmax[3]);
      }
    }
  }
}

// OIHW fp32 [Cout][Cin][3][3] -> [Cin/8][Cout/32][2 (k half)][32 (co)][68]: element (cb, ct, kh, j, xi*4 + m) =
// (G g G^T)[xi] of filter (co = ct*32 + j, ci = cb*8 + kh*4 + m), evaluated in double and rounded once; the 4 pad floats are 0.
__global__ void pack_conv3x3_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
  const long rows = (long)(Cin >> 3) * (Cout >> 5) * 64;          // (cb, ct, kh, j)
  for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < rows * 4; row += (long)gridDim.x * blockDim.x) {
    const int m = (int)(row & 3);
    const long r = row >> 2;
    const int jj = (int)(r & 31), kh = (int)((r >> 5) & 1);
    const long t = r >> 6;
    const int ct = (int)(t % (Cout >> 5)), cb = (int)(t / (Cout >> 5));
    const int co = ct * 32 + jj, ci = cb * 8 + kh * 4 + m;
    const float* g = w + ((long)co * Cin + ci) * 9;
    double gg[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) gg[a][b] = (double)g[a * 3 + b];
    double tmp[4][3];                                             // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      tmp[0][b] = gg[0][b];
      tmp[1][b] = 0.5 * (gg[0][b] + gg[1][b] + gg[2][b]);
      tmp[2][b] = 0.5 * (gg[0][b] - gg[1][b] + gg[2][b]);
      tmp[3][b] = gg[2][b];
    }
    float* dst = out + r * kWRowPitch;
#pragma unroll
    for (int a = 0; a < 4; ++a) {                                 // (G g) G^T
      dst[(a * 4 + 0) * 4 + m] = (float)tmp[a][0];
      dst[(a * 4 + 1) * 4 + m] = (float)(0.5 * (tmp[a][0] + tmp[a][1] + tmp[a][2]));
      dst[(a * 4 + 2) * 4 + m] = (float)(0.5 * (tmp[a][0] - tmp[a][1] + tmp[a][2]));
      dst[(a * 4 + 3) * 4 + m] = (float)tmp[a][2];
    }
    if (m == 0) { dst[64] = 0.f; dst[65] = 0.f; dst[66] = 0.f; dst[67] = 0.f; }
  }
}

// Finishes the pixel tiles [pix0, pix0 + npix) of a section whose tiles were cut into `s` K ranges: out = act(sum_k part[k] + bias)
// over the tile's 4*RG x 32 pixels (clipped to the image), all channels; POOL: followed by the Pooling MAX 2x2/2 that the
// unsplit tiles apply in their epilogue (same order: ReLU, then the maximum over the window's in-image pixels), written to
// the pooled tensor.  One thread per 4 channels of a pixel (of a pooling window).
template <int POOL>
__global__ __launch_bounds__(256) void wino_section_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                  float* __restrict__ out, int H, int W, int Cout, int s, int relu,
                                                                  int pix0, int npix, int tiles_x, int tile_rows) {
  const int rows = POOL ? tile_rows >> 1 : tile_rows, cols = POOL ? kWCols >> 1 : kWCols;
  const int per_tile = (Cout >> 3) * rows * cols * 2;
  const long plane = (long)Cout * H * W;
  const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)npix * per_tile; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i / per_tile);
    int r = (int)(i - (long)t * per_tile);
    const int half = r & 1; r >>= 1;
    const int col = r % cols; r /= cols;
    const int row = r % rows;
    const int cb = r / rows;
    const int pix = pix0 + t, bx = pix % tiles_x, by = pix / tiles_x;
    const float4 b = *reinterpret_cast<const float4*>(bias + cb * 8 + half * 4);
    const int y0 = by * tile_rows + (POOL ? 2 * row : row), x0 = bx * kWCols + (POOL ? 2 * col : col);
    if (y0 >= H || x0 >= W) continue;
    float4 best = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
    for (int dy = 0; dy < (POOL ? 2 : 1); ++dy)
#pragma unroll
      for (int dx = 0; dx < (POOL ? 2 : 1); ++dx) {
        const int y = y0 + dy, x = x0 + dx;
        if (y >= H || x >= W) continue;
        const long e = (((long)cb * H + y) * W + x) * 8 + half * 4;
        float4 v = *reinterpret_cast<const float4*>(part + e);
        for (int k = 1; k < s; ++k) {
          const float4 q = *reinterpret_cast<const float4*>(part + k * plane + e);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (POOL) {
          best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
        } else {
          *reinterpret_cast<float4*>(out + e) = v;
        }
      }
    if (POOL) *reinterpret_cast<float4*>(out + (((long)cb * OH + (y0 >> 1)) * OW + (x0 >> 1)) * 8 + half * 4) = best;
  }
}

template <int ROWS, int VAR>
static int launch_wino(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                       int Cin, int Cout, int relu, int ksplit, float* part) {
  constexpr size_t lds = 2 * 4 * ((size_t)(4 * ROWS + 2) * kWHaloCols * kWPixPitch + (size_t)kWPanel);
  static_assert(lds <= 160 * 1024, "conv3x3_wino: LDS budget");
  auto kern = conv3x3_wino_kernel<ROWS, VAR>;
  static std::atomic<unsigned long long> attr_set{0};            // one bit per device: function attributes are per device
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  dim3 grid(cdiv(W, kWCols), cdiv(H, 4 * ROWS), (Cout >> 5) * ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(64 * ROWS), lds, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit,
                     part);
  return MNC_OK;
}

template <int RG, int ABL, int DMA, int XCD = 1>
static int launch_wino2(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                        int Cin, int Cout, int relu, int ksplit, float* part, int pool, int pix_a, int ksplit_b) {
  constexpr size_t lds_stage = 2 * 4 * ((size_t)(4 * RG + 2) * kWHaloCols * kWPixPitch + (size_t)kWPanel);
  constexpr size_t lds_xch = (size_t)RG * 64 * 64 * 4;
  constexpr size_t lds = lds_stage > lds_xch ? lds_stage : lds_xch;
  static_assert(lds <= (RG <= 2 ? 80 : 160) * 1024, "conv3x3_wino2: LDS budget (two workgroups per CU up to RG = 2)");
  auto kern = conv3x3_wino2_kernel<RG, ABL, DMA, XCD>;
  static std::atomic<unsigned long long> attr_set{0};
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const int tiles_x = cdiv(W, kWCols);
  const int pix = tiles_x * cdiv(H, 4 * RG);
  if (pix_a < 0 || pix_a > pix) pix_a = pix;                     // no second section
  dim3 grid((pix_a * ksplit + (pix - pix_a) * ksplit_b) * (Cout >> 5));
  hipLaunchKernelGGL(kern, grid, dim3(128 * RG), lds, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit,
                     part, tiles_x, pool, pix_a, ksplit_b);
  return MNC_OK;
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_pack_conv3x3_wino(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin) {
  MNC_REQUIRE(ctx && d_oihw && d_packed && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_pack_conv3x3_wino: bad argument (Cin%%8==0, Cout%%32==0)");
  LaunchScope ls(ctx, "pack_conv3x3_wino");
  const long items = (long)(Cin >> 3) * (Cout >> 5) * 64 * 4;
  long g = (items + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_conv3x3_wino_kernel, dim3((int)g), dim3(256), 0, ctx->stream, d_oihw, d_packed, Cout, Cin);
  return ls.finish("pack_conv3x3_wino_kernel");
}

}  // extern "C"

// pool != 0: d_out is the pooled tensor [Cout/8][ceil(H/2)][ceil(W/2)][8] (the following Pooling MAX 2x2/2 applied in the epilogue)
static int wino_impl(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                     int Cin, int Cout, int relu, int pool) {
  MNC_REQUIRE(ctx && d_in && d_wpk && d_bias && d_out, "mnc_conv3x3_wino: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_conv3x3_wino: unsupported shape H=%d W=%d Cin=%d Cout=%d (need Cin%%8==0, Cout%%32==0)", H, W, Cin, Cout);
  // Workgroup = ROWS waves (4*ROWS pixel rows x 32 columns x 32 channels), one workgroup per CU (256 accumulator registers per
  // wave).  Pick the tallest workgroup that still gives every CU >= 3 workgroups; small maps additionally split K.
  const int ncot = Cout >> 5;
  // measured (tools/kernel_bench.py convwino, MNC_WINO_ROWS): 4-wave workgroups win on every trunk shape from 600x1000 down
  // to 75x125 (one workgroup per CU either way; taller workgroups amortise the weight panel), 2 waves on maps under 64 rows
  int rows = H >= 64 ? 4 : 2;
  if (tune_set(ctx, T_WINO_ROWS)) {
    const int v = tune(ctx, T_WINO_ROWS, 0);
    if (v == 1 || v == 2 || v == 4) rows = v;
  }
  // K splits: the chip holds 512 workgroups at a time (two per CU); with fewer than ~1000 the last, partly filled round costs
  // more than the extra pass over the partial sums (measured: conv4_x 640 workgroups 304 -> 279 us with 2 splits, conv5_x 160
  // workgroups 118 -> 93 us with 4)
  int ksplit = 1;
  {
    const long wgs = (long)cdiv(W, kWCols) * cdiv(H, 4 * (rows >= 2 ? 2 : 1)) * ncot;
    const int blocks = Cin / 8;
    while (ksplit < 4 && wgs * ksplit < 1024 && blocks % (2 * ksplit) == 0 && blocks / (2 * ksplit) >= 4) ksplit *= 2;
    if (tune_set(ctx, T_CONV_KSPLIT)) {
      const int v = tune(ctx, T_CONV_KSPLIT, 0);
      if (v >= 1 && v <= 8 && blocks % v == 0) ksplit = v;
    }
  }
  // Tail plan (two-row-group kernel, no MNC_CONV_KSPLIT override; MNC_WINO_TAIL=0 keeps the uniform rule above).  The chip holds
  // `slots` workgroups (two per CU).  The pixel tiles of the full rounds run unsplit; the tiles of the last, partly filled
  // round are cut into sB >= 3 K ranges -- as many as fill that round once, with at least 8 blocks each; on small maps whose
  // last round leaves three quarters of the chip idle also 2 ranges, of at least 4 blocks -- and finished by
  // wino_section_reduce_kernel.  600x1000: conv4_x 512 + 128 x 4 workgroups instead of 640 x 2 (2.5 rounds): 256 -> 240 us;
  // conv5_x / rpn_conv 160 x 3 instead of 160 x 4 (1.25 rounds): 84.5 -> 73.4 us.  A tail workgroup that has its CU to itself
  // runs faster than one of a pair, so a lightly filled last round costs less than its length suggests: cutting conv3_x's 192
  // tail tiles in two (the reduce pass included) LOST 11 us, and those layers stay whole.
  int pix_a = -1, ksplit_b = 1;
  {
    const bool tail_on = tune(ctx, T_WINO_TAIL, 1) != 0;
    const int ver_env = tune(ctx, T_WINO_V, 2);
    if (tail_on && rows >= 2 && ver_env == 2 && !tune_set(ctx, T_CONV_KSPLIT)) {
      const int pix = cdiv(W, kWCols) * cdiv(H, 8), blocks = Cin / 8, slots = 512;      // MI355X: 256 CUs x two workgroups
      const int full_pix = (int)((long)pix * ncot / slots) * slots / ncot;      // pixel tiles of the full rounds
      const int rest = (pix - full_pix) * ncot;
      ksplit = 1;
      if (rest > 0) {
        const bool light = rest <= slots / 4;                    // a last round that leaves three quarters of the chip idle
        const int min_blocks = light ? 4 : 8;                    // (small maps: ranges of 4 blocks, as the uniform rule had)
        int sb = slots / rest;
        if (sb > blocks / min_blocks) sb = blocks / min_blocks;
        if (sb > 8) sb = 8;
        // two ranges do not pay on a well filled round (conv3_x, 192 tail tiles: 237 -> 248 us)
        if (sb >= 3 || (sb == 2 && light)) { pix_a = full_pix; ksplit_b = sb; }
      }
    }
  }
  float* part = nullptr;
  float* full = nullptr;            // pooled output of a K-split layer: the reduction writes full resolution here first
  if (ksplit > 1 || ksplit_b > 1) {
    int rc = ensure_scratch(ctx, (size_t)((ksplit > ksplit_b ? ksplit : ksplit_b) + (pool && ksplit > 1 ? 1 : 0)) * Cout * H * W * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
    if (pool && ksplit > 1) full = part + (size_t)ksplit * Cout * H * W;
  }
  const int kpool = (pool && ksplit == 1) ? 1 : 0;               // pooling inside the kernel's epilogue
  const double flops = 2.0 * H * W * 9.0 * Cin * Cout;           // ALGORITHMIC work of the convolution (direct form)
  // algorithmic bytes: input + weights + what this launch must write -- the pooled tensor when the Pooling is fused (the
  // full-resolution output then never exists)
  const double out_px = pool ? (double)((H + 1) / 2) * ((W + 1) / 2) : (double)H * W;
  const double bytes = 4.0 * ((double)H * W * Cin + out_px * Cout + 9.0 * Cin * Cout);
  LaunchScope ls(ctx, "conv3x3_wino_mfma", flops, bytes);
  // WINO_VAR: 7 (the product build) rotated loop, all halo rows read behind the barrier, LDS-DMA weight panel.  -DMNC_TUNING builds
  // also carry 3 (row B read at the head of the iteration), 1 (flat block schedule, register staging), 0 (the round-2 v2 loop)
  // (13-layer trunk, kernel_bench convwino: 2.19 / 2.22 / 2.28 / 2.54 ms), the ablations 16 / 48 / 112 and WINO_V = 1 (v1 kernel).
  // One-row-group workgroups (maps under 64 rows) and inputs beyond 32-bit byte offsets run the v2 loop (var 0) in every build.
  int var = tune(ctx, T_WINO_VAR, 7), ver = tune(ctx, T_WINO_V, 2);
#ifndef MNC_TUNING
  var = 7; ver = 2;
#endif
  if ((var == 1 || var == 3 || var == 7) && (rows < 2 || ver != 2 || (double)Cin * H * W * 4.0 >= 2147483648.0 || tune_set(ctx, T_WINO_DMA))) var = 0;
  MNC_REQUIRE(!pool || (ver == 2 && (var == 0 || var == 1 || var == 3 || var == 7)), "mnc_conv3x3_wino_pool: only the default kernel build fuses the pooling");
  int rc = MNC_ERR_INVALID;
  // XCD-aware order (see the kernel) where the channel tiles' shared input dominates the traffic; measured FETCH_SIZE per launch
  // plain -> XCD-aware: conv1_2 197 -> 93 MB, conv2_x 153 -> 41, conv3_x 126 -> 60; for the 512-channel layers the 17.8 MB of
  // transformed weights dominate and every XCD would stream all of them (conv4_x 84 -> 248 MB, conv5_x 38 -> 93): plain order
  const bool plain_order = tune_set(ctx, T_WINO_XCD) ? tune(ctx, T_WINO_XCD, 1) == 0 : Cout > 256;
#ifdef MNC_TUNING
  // WINO_STREAM = k > 0: the layer as one stream of (tile, block) units in 512 k equal ranges (conv_wino_stream.hip: measured, slower)
  const int stream_k = (ver == 2 && var == 7) ? tune(ctx, T_WINO_STREAM, 0) : 0;
  if (stream_k > 0) {
    rc = wino_stream_launch(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, pool, stream_k, plain_order ? 0 : 1);
    if (rc == MNC_OK) return ls.finish("conv3x3_wino_stream_kernel");
    if (rc != MNC_ERR_INVALID) return rc;
  }
#endif
#ifdef MNC_TUNING
  if (ver == 2 && var == 7 && tune(ctx, T_WINO_MFMA16, 0) != 0)    // the same kernel on 16 x 16 x 4 fragments (conv_wino16.hip: measured, equal)
    rc = wino16_launch(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b, plain_order ? 0 : 1);
  else
#endif
  if (ver == 2 && var == 7)
    rc = plain_order ? launch_wino2<2, 7, 1, 0>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b)
                     : launch_wino2<2, 7, 1, 1>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
  else if (ver == 2 && var == 0 && !tune_set(ctx, T_WINO_DMA) && !(rows == 4 && tune_set(ctx, T_WINO_ROWS)))
#ifdef MNC_TUNING            // (the one-row-group build spills ten registers at its 256-register budget: measurement builds only, round 6)
    rc = rows >= 2 ? launch_wino2<2, 0, 0, 1>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b)
                   : launch_wino2<1, 0, 0, 1>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
#else
    rc = launch_wino2<2, 0, 0, 1>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
#endif
#ifdef MNC_TUNING
  else if (ver == 2) {              // wave pairs, 128 accumulators, two workgroups per CU (rows = row groups per workgroup: 1 | 2)
    // measured (kernel_bench convwino, 13-layer trunk): register staging 2.526 ms, LDS-DMA weight panel 2.564 ms -- the DMA saves
    // 20 registers and the ds_write pass but hipcc drains it with vmcnt(0) in front of every barrier; kept selectable
    int dma = tune(ctx, T_WINO_DMA, 0) != 0;
    if (plain_order && rows >= 2 && var == 0 && dma == 0)
      rc = launch_wino2<2, 0, 0, 0>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
    if (plain_order && rows >= 2 && var == 1 && dma == 0)
      rc = launch_wino2<2, 1, 0, 0>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
    if (var == 3) dma = 1;
    if (plain_order && rows >= 2 && var == 3)
      rc = launch_wino2<2, 3, 1, 0>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
#define MNC_WINO2_CASE(R, A, D) if (rc == MNC_ERR_INVALID && (rows >= 2 ? 2 : 1) == R && var == A && dma == D) rc = launch_wino2<R, A, D>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
    if (rc == MNC_ERR_INVALID && rows == 4 && tune_set(ctx, T_WINO_ROWS) && var == 0 && dma == 0)          // 8-wave workgroups
      rc = launch_wino2<4, 0, 0>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part, kpool, pix_a, ksplit_b);
    MNC_WINO2_CASE(2, 0, 1) MNC_WINO2_CASE(1, 0, 1) MNC_WINO2_CASE(2, 1, 0) MNC_WINO2_CASE(2, 3, 1)
    MNC_WINO2_CASE(2, 16, 0) MNC_WINO2_CASE(2, 48, 0) MNC_WINO2_CASE(2, 112, 0)                  // ablations
#undef MNC_WINO2_CASE
  } else {
#define MNC_WINO_CASE(R, V) if (rows == R && var == V) rc = launch_wino<R, V>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part);
    MNC_WINO_CASE(4, 0) MNC_WINO_CASE(4, 1) MNC_WINO_CASE(2, 0) MNC_WINO_CASE(2, 1) MNC_WINO_CASE(1, 0) MNC_WINO_CASE(1, 1)
    MNC_WINO_CASE(4, 16) MNC_WINO_CASE(4, 48) MNC_WINO_CASE(4, 112) MNC_WINO_CASE(4, 240)      // ablations
#undef MNC_WINO_CASE
  }
#endif
  MNC_REQUIRE(rc != MNC_ERR_INVALID, "mnc_conv3x3_wino: no kernel for rows=%d WINO_VAR=%d WINO_V=%d (ablation / superseded builds need -DMNC_TUNING)", rows, var, ver);
  if (rc) return rc;
  if (ksplit > 1) conv_splitk_reduce_launch(ctx->stream, part, d_bias, pool ? full : d_out, H, W, Cout, ksplit, relu);
  if (ksplit_b > 1) {
    const int tiles_x = cdiv(W, kWCols), npix = tiles_x * cdiv(H, 8) - pix_a;
    const long items = (long)npix * (Cout >> 3) * 8 * kWCols * 2 / (pool ? 4 : 1);
    auto grid_for = [](long n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); };
    if (pool)
      hipLaunchKernelGGL(wino_section_reduce_kernel<1>, dim3(grid_for(items)), dim3(256), 0, ctx->stream, part, d_bias, d_out, H, W,
                         Cout, ksplit_b, relu, pix_a, npix, tiles_x, 8);
    else
      hipLaunchKernelGGL(wino_section_reduce_kernel<0>, dim3(grid_for(items)), dim3(256), 0, ctx->stream, part, d_bias, d_out, H, W,
                         Cout, ksplit_b, relu, pix_a, npix, tiles_x, 8);
  }
  rc = ls.finish("conv3x3_wino_kernel");
  if (rc) return rc;
  if (pool && ksplit > 1) return mnc_maxpool2_c8(ctx, full, d_out, Cout, H, W);     // K-split layer: pooled after the reduction
  return MNC_OK;
}

extern "C" {

int mnc_conv3x3_wino(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                     int Cin, int Cout, int relu) {
  return wino_impl(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, 0);
}

int mnc_conv3x3_wino_pool(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out_pooled, int H,
                          int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(H >= 2 && W >= 2, "mnc_conv3x3_wino_pool: map %dx%d too small for MAX 2x2/2", H, W);
  return wino_impl(ctx, d_in, d_wpk, d_bias, d_out_pooled, H, W, Cin, Cout, relu, 1);
}

}  // extern "C"
