// Winograd F(2x2, 3x3) convolution as ONE STREAM of (tile, channel block) units, for gfx950 -- the trunk's 3x3 layers
// (models/VGG16/mnc_5stage/test.prototxt:41-412) on the wave-pair kernel of conv_wino.hip, with a different outer structure.
//
// Why.  conv3x3_wino2_kernel gives every 8 x 32-pixel x 32-channel tile its own workgroup: the chip holds 512 of them (two per
// CU), so a layer costs ceil(tiles / 512) rounds (conv3_x: 1216 tiles = 2.375 rounds, the third 37 % full) and every workgroup
// pays a prologue (two global round trips before its first MFMA) and an epilogue next to 2.85 us per channel block -- a quarter of
// the time of the 64- and 128-channel layers.  Here the layer is the sequence of units
//     u = tile * (Cin / 8) + block,      tiles in the order (pixel tile, output-channel tile fastest)
// cut into G equal ranges [start(l), start(l + 1)), start(l) = 2 * floor(l * U / 2 / G): workgroup l walks ITS range block by
// block with the same double-buffered loop -- the prefetch simply runs on into the next tile, so only the first block of a
// workgroup waits for memory -- and runs the epilogue whenever a tile's last block (or the range's) is done.  Every workgroup
// gets the same number of blocks (+-2): no partly filled round, whatever the tile count.
//   * a tile that lies inside one range is finished in the epilogue as before (output transform, bias, ReLU, optional MAX 2x2/2);
//   * a tile cut by a range boundary leaves its pieces (the transformed, pair-reduced 2x2 outputs, no bias) in `part`
//     [workgroup][slot 0: the range's first tile | slot 1: its last][8192 floats]; wino_stream_fix_kernel adds the pieces of such
//     a tile in range order and finishes it.  The partition depends on the shape and G only: results are deterministic (they
//     differ from the tiled kernel's in the last bits where a tile is cut: other summation order).
//   * the epilogue's wave-pair exchange uses the halo buffer of the block that was just finished (free behind its barrier), in two
//     halves of 16 KB: the staging buffers of the NEXT tile's first blocks stay intact under it.
//   * XCD order: workgroup b runs on XCD b % 8; ranges are numbered so that every XCD walks one contiguous eighth of the
//     stream (neighbouring tiles, shared halos and the channel tiles of one pixel tile in one L2).
//
// STATUS: a measurement build (-DMNC_TUNING, MNC_WINO_STREAM=k or abl,k), not the product path.  It is correct (tests/test_gpu_ops.py
// runs it against torch and the direct kernel) and SLOWER than the tiled kernel on every trunk layer (13-layer trunk 2.29 vs
// 2.17 ms): the premise above was wrong.  What its per-range stamps (ABL & 32: s_memtime cycles, wall_clock64, HW_ID) showed --
// DESIGN.md section 9, profiles/r03_wino_stream_stamps.txt:
//   * the two workgroups of a CU share the matrix pipe, so a partly filled round runs its workgroups faster, not the chip emptier:
//     slot quantisation costs the tiled kernel ~5 %, not 21 %; and the hardware dispatcher re-balances continuously, which a
//     static partition cannot -- of a CU's two resident workgroups the one in wave slot 0 is served first (4300 vs 5150 cycles
//     per block) and the other sets the kernel's time (s_setprio alternation makes both 5065: the CU's throughput is the same);
//   * the shader clock under this loop is 2.0 GHz, not 2.4: the fp32 matrix pipe is POWER limited (tools/probes/
//     mfma_f32_mix_probe.hip: a bare MFMA stream issues every 48 cycles at 1.55 GHz = 132-150 TFLOP/s; with the loop's 20
//     ds_read_b128 + 64 VALU per 32 MFMAs 4830 cycles per block and wave at 2.36 GHz = 128 TFLOP/s -- the loop here: 4850 at 2.0);
//   * ablations in cycles (per block, both workgroups of a CU averaged, 4850): LDS reads -570, input transform -400, global
//     loads + LDS-DMA -200, barrier -150, halo stores -110: every register-file write beside the MFMAs' own 4 KB per
//     instruction costs pipe time.
#ifdef MNC_TUNING
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <vector>

#include "wino_common.h"

namespace mnc {

namespace {

constexpr int kSNT = 256;                                         // four waves: two row groups x the two halves of the positions
constexpr int kSHaloRows = 10;                                    // 8 pixel rows + halo
constexpr int kSHaloVec = kSHaloRows * kWHaloCols * 2;            // float4 pieces per halo (680)
constexpr int kSHaloStride = 4096;                                // floats per halo buffer: 4080 used, 16 KB = one exchange half
constexpr int kSHPer = (kSHaloVec + kSNT - 1) / kSNT;             // halo pieces per thread (3)
constexpr int kSTileFloats = 32 * 8 * kWCols;                     // a tile's outputs (8192)
constexpr int kSDma = (kWPanel / 256 + 3) / 4;                    // LDS-DMA instructions per wave and weight panel (17 KB / 4 waves)
static_assert(kSHaloRows * kWHaloCols * kWPixPitch <= kSHaloStride, "halo buffer");

__device__ __host__ __forceinline__ int stream_start(int l, int u2, int g) { return 2 * (int)((long)l * u2 / g); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ABL (tuning builds): timing ablations of the block loop, results are wrong -- 1: no global loads / LDS-DMA, 2: no halo stores,
// 4: no barrier, 8: no LDS reads, 16: no input transform.
template <int XCD, int ABL = 0>
__global__ __launch_bounds__(kSNT, 2) void conv3x3_wino_stream_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                                      int H, int W, int Cin, int Cout, int relu, int pool,
                                                                      int tiles_x, int u2, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];   // halo[2] (4096 floats each) then weights[2]
  float* const s_halo = s_mem;
  float* const s_w = s_mem + 2 * kSHaloStride;
  unsigned long long stamp_k = 0;
  if (ABL & 32) stamp_k = wall_clock64();

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int rg = wave & 1, hf = wave >> 1;                        // waves w and w + 2 are a pair (same tiles, other positions)
  const int j = lane & 31, kk = lane >> 5;
  const int ty = j >> 4, tx = j & 15;
  const int ncot = Cout >> 5, nch = Cin >> 3;
  const int G = gridDim.x;
  int l;
  {
    const int b = blockIdx.x, q = G >> 3, r = G & 7, xcd = b & 7, idx = b >> 3;
    l = XCD ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : b;
  }
  const int u0 = stream_start(l, u2, G), u1 = stream_start(l + 1, u2, G);

  // ---- staging: halo through registers (buffer loads: a piece outside the image carries an out-of-range offset and reads
  // zeros), weight panel by LDS-DMA; conv_wino.hip explains the layouts
  int h_off[kSHPer];
#pragma unroll
  for (int u = 0; u < kSHPer; ++u) {
    const int q = min(tid + u * kSNT, kSHaloVec - 1);
    const int pix = q >> 1, half = q & 1, r = pix / kWHaloCols;
    h_off[u] = pix * kWPixPitch + (half ^ ((r >> 1) & 1)) * 4;
  }
  const long plane = (long)H * W * 8;
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)min((long)Cin * H * W * 4, 0x7FFFFFFFL), 0x00020000);
  int hb_off[kSHPer];                                             // byte offsets of the halo pieces of the PREFETCH tile
  int pf_u, pf_c, pf_t, pf_cot;                                   // prefetch cursor: unit, its block, its tile, the tile's channel tile
  auto set_tile = [&](int t) {
    const int p = t / ncot, bx = p % tiles_x, by = p / tiles_x;
    pf_cot = t - p * ncot;
#pragma unroll
    for (int u = 0; u < kSHPer; ++u) {
      const int q = min(tid + u * kSNT, kSHaloVec - 1);
      const int pix = q >> 1, half = q & 1, r = pix / kWHaloCols, c = pix - r * kWHaloCols;
      const int gh = by * 8 - 1 + r, gw = bx * kWCols - 1 + c;
      hb_off[u] = (gh >= 0 && gh < H && gw >= 0 && gw < W) ? ((gh * W + gw) * 8 + half * 4) * 4 : 0x7FFFFFF0;
    }
  };
  float4 Gh[kSHPer];
  auto issue = [&](int buf) {                                     // the prefetch unit: halo -> registers, panel -> s_w[buf]
    const int hs = __builtin_amdgcn_readfirstlane(pf_c * (int)(plane * 4));
#pragma unroll
    for (int u = 0; u < kSHPer; ++u) {
      const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, hb_off[u], hs, 0);
      Gh[u] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
    }
    const float* src = wpk + ((long)pf_c * ncot + pf_cot) * kWPanel;
    float* dstw = s_w + buf * kWPanel;
#pragma unroll
    for (int i = 0; i < kSDma; ++i) {
      const int piece = min(wave + i * 4, kWPanel / 256 - 1);     // branch-free: the waves without a last piece repeat piece 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dstw + piece * 256), 16, 0, 0);
    }
  };
  auto advance = [&]() {                                          // past the range's end the last unit is requested again (unused)
    if (pf_u + 1 < u1) {
      ++pf_u;
      if (++pf_c == nch) {
        pf_c = 0;
        set_tile(++pf_t);
      }
    }
  };
  auto store_halo = [&](int buf) {
    float* hdst = s_halo + buf * kSHaloStride;
#pragma unroll
    for (int u = 0; u < kSHPer; ++u) *reinterpret_cast<float4*>(hdst + h_off[u]) = Gh[u];
  };

  f32x16 acc[8];

  // ---- the block loop of conv3x3_wino2_kernel<2, 7, 1, *> (rotated: barrier in the middle of a block's MFMAs; see there)
  const int rowA = hf ? 2 : 0, rowB = hf ? 3 : 1, rowC = hf ? 1 : 2;
  const float sgn = hf ? -1.f : 1.f;
  const int d_row0 = ((4 * rg + 2 * ty) * kWHaloCols + 2 * tx) * kWPixPitch;
  auto row_off = [&](int row) { return d_row0 + row * kWHaloCols * kWPixPitch + (kk ^ ((ty + (row >> 1)) & 1)) * 4; };
  const int offA = row_off(rowA), offB = row_off(rowB), offC = row_off(rowC);
  const int u_base = (kk * 32 + j) * kWRowPitch + hf * 32;        // positions 8*hf .. 8*hf + 7
  const f32x4 sgn4 = {sgn, sgn, sgn, sgn};
  f32x4 fA[4], fB[4], fC[4], fu0, fu1;
  f32x4 v[8], t0[4], t1[4];
  auto mfma_pair = [&](int p, const f32x4& a0, const f32x4& a1) {
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, v[p].x, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, v[p + 1].x, acc[p + 1], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, v[p].y, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, v[p + 1].y, acc[p + 1], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, v[p].z, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, v[p + 1].z, acc[p + 1], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, v[p].w, acc[p], 0, 0, 0);
    acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, v[p + 1].w, acc[p + 1], 0, 0, 0);
  };
  auto transform_row = [&](const f32x4 (&t)[4], f32x4* o) {
    o[0] = t[0] - t[2];
    o[1] = t[1] + t[2];
    o[2] = t[2] - t[1];
    o[3] = t[1] - t[3];
  };
  auto read_rows = [&](int buf) {                                 // a block's three halo rows and its first weight fragments
    const float* sh = s_halo + buf * kSHaloStride;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fA[c] = *reinterpret_cast<const f32x4*>(sh + offA + c * kWPixPitch);
      fC[c] = *reinterpret_cast<const f32x4*>(sh + offC + c * kWPixPitch);
    }
    fu0 = *reinterpret_cast<const f32x4*>(s_w + buf * kWPanel + u_base);
    fu1 = *reinterpret_cast<const f32x4*>(s_w + buf * kWPanel + u_base + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) fB[c] = *reinterpret_cast<const f32x4*>(sh + offB + c * kWPixPitch);
  };

  // ---- prologue: the range's first two units requested together (one global round trip in front of the first MFMA)
  pf_u = u0;
  pf_t = u0 / nch;
  pf_c = u0 - pf_t * nch;
  set_tile(pf_t);
  {
    issue(0);
    advance();
    float4 G0[kSHPer];
#pragma unroll
    for (int u = 0; u < kSHPer; ++u) G0[u] = Gh[u];
    issue(1);                                                     // (the loop advances the cursor at the head of every block)
    float* hdst = s_halo;
#pragma unroll
    for (int u = 0; u < kSHPer; ++u) *reinterpret_cast<float4*>(hdst + h_off[u]) = G0[u];
  }
  __syncthreads();

  int hw_slot = 0;
  if (ABL & (64 | 128)) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    hw_slot = hw & 1;                                             // wave slot on the SIMD: 0 = the older workgroup of the CU's two
    if ((ABL & 128) && hw_slot) __builtin_amdgcn_s_setprio(1);    // tuning: the younger workgroup always first
  }
  unsigned long long stamp_c = 0, stamp_w = 0;
  if (ABL & 32) { stamp_c = __builtin_readcyclecounter(); stamp_w = wall_clock64(); }
  int par = 0;                                                    // staging buffer of the unit about to be multiplied
  int t = u0 / nch;
  for (int u = u0; u < u1; ++t) {
    const int seg_end = min((t + 1) * nch, u1);
    const int nseg = seg_end - u;
    const int slot = u == u0 ? 0 : 1;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    // the loop-carried registers of the segment's first block (behind a tile's epilogue: read once more -- the block is in LDS)
    read_rows(par);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      t0[q] = fA[q] - fC[q];
      t1[q] = __builtin_elementwise_fma(sgn4, fB[q], fC[q]);
    }
    transform_row(t0, v);
    for (int i = 0; i < nseg; ++i) {
      const int buf = par;
      // the cursor moves HERE, not behind the MFMAs: a branch at the end of the body lets hipcc sink the next block's transform
      // arithmetic behind it, out from under the MFMAs of positions 6-7
      advance();
      if (ABL & 64) {                                             // tuning: issue priority alternates block by block between the CU's two workgroups
        if ((par ^ hw_slot) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      }
      const float* sw = s_w + buf * kWPanel + u_base;
      __builtin_amdgcn_sched_barrier(0);
      // positions 0-1, the second row of the column pass under them
      f32x4 n0 = fu0, n1 = fu1;
      if (!(ABL & 8)) {
        n0 = *reinterpret_cast<const f32x4*>(sw + 8);
        n1 = *reinterpret_cast<const f32x4*>(sw + 12);
      }
      if (!(ABL & 16)) transform_row(t1, v + 4);
      mfma_pair(0, fu0, fu1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // positions 2-3; the last fragments of this block in, the next block's halo out
      f32x4 m0 = fu0, m1 = fu1, q0 = fu0, q1 = fu1;
      if (!(ABL & 8)) {
        m0 = *reinterpret_cast<const f32x4*>(sw + 16);
        m1 = *reinterpret_cast<const f32x4*>(sw + 20);
        q0 = *reinterpret_cast<const f32x4*>(sw + 24);
        q1 = *reinterpret_cast<const f32x4*>(sw + 28);
      }
      if (!(ABL & 2)) store_halo(buf ^ 1);
      mfma_pair(2, n0, n1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // positions 4-5: the next block's rows and first fragments requested behind the first MFMA, then the unit two ahead
      if (!(ABL & 8)) read_rows(buf ^ 1);
      if (!(ABL & 1)) issue(buf);
      mfma_pair(4, m0, m1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      // positions 6-7, the next block's transform rows and the first row of its column pass under them
      if (!(ABL & 16)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          t0[q] = fA[q] - fC[q];
          t1[q] = __builtin_elementwise_fma(sgn4, fB[q], fC[q]);
        }
        transform_row(t0, v);
      }
      mfma_pair(6, q0, q1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      par ^= 1;
    }
    u = seg_end;

    // ---- epilogue of tile t (its blocks [.., seg_end) are in acc).  acc[p][e], p = 4*r + c with r the wave's local row (global
    // row 2*hf + r), e -> channel (e&3) + 8*(e>>2) + 4*kk.  Partial output transform of this wave's rows (hf = 0: s0 = M0 + M1,
    // s1 = M1; hf = 1: s0 = M2, s1 = -(M2 + M3)); the pair's sums meet in the halo buffer of the block just finished.
    const bool whole = nseg == nch;
    const int p_ = t / ncot, bx = p_ % tiles_x, by = p_ / tiles_x, cot = t - p_ * ncot;
    const int oy = by * 8 + 4 * rg + 2 * ty, ox = bx * kWCols + 2 * tx;
    float* xch = s_halo + (par ^ 1) * kSHaloStride + rg * (32 * 64);
    float* ptile = part + ((long)l * 2 + slot) * kSTileFloats;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float y[2][2][2][4];
#pragma unroll
      for (int gl = 0; gl < 2; ++gl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = 4 * (2 * s + gl) + q;
          float s0[4], s1[4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float m0 = acc[cc][e], m1 = acc[4 + cc][e];
            s0[cc] = hf ? m0 : m0 + m1;
            s1[cc] = hf ? -(m0 + m1) : m1;
          }
          y[gl][0][0][q] = s0[0] + s0[1] + s0[2];
          y[gl][0][1][q] = s0[1] - s0[2] - s0[3];
          y[gl][1][0][q] = s1[0] + s1[1] + s1[2];
          y[gl][1][1][q] = s1[1] - s1[2] - s1[3];
        }
      if (hf == 1) {
#pragma unroll
        for (int gl = 0; gl < 2; ++gl)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
              for (int q = 0; q < 4; ++q) xch[((((gl * 2 + dy) * 2 + dx) * 4 + q) * 64) + lane] = y[gl][dy][dx][q];
      }
      __syncthreads();
      if (hf == 0) {
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
          const int g = 2 * s + gl;
          const int co = cot * 32 + g * 8 + kk * 4;
          if (!whole) {
            // a piece of a cut tile: raw sums, [g][dy][dx][row group][lane] float4 (wino_stream_fix_kernel reads the same order)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
              for (int dx = 0; dx < 2; ++dx) {
                float4 o;
                o.x = y[gl][dy][dx][0] + xch[((((gl * 2 + dy) * 2 + dx) * 4 + 0) * 64) + lane];
                o.y = y[gl][dy][dx][1] + xch[((((gl * 2 + dy) * 2 + dx) * 4 + 1) * 64) + lane];
                o.z = y[gl][dy][dx][2] + xch[((((gl * 2 + dy) * 2 + dx) * 4 + 2) * 64) + lane];
                o.w = y[gl][dy][dx][3] + xch[((((gl * 2 + dy) * 2 + dx) * 4 + 3) * 64) + lane];
                *reinterpret_cast<float4*>(ptile + ((((g * 2 + dy) * 2 + dx) * 2 + rg) * 64 + lane) * 4) = o;
              }
            continue;
          }
          const float4 b = *reinterpret_cast<const float4*>(bias + co);
          const float bb[4] = {b.x, b.y, b.z, b.w};
          // pool != 0: the following Pooling MAX 2x2 stride 2 applied here (a Winograd tile IS a pooling window; Caffe's ceil
          // rule clips the last window of an odd-sized map): conv_wino.hip
          float pmax[4] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              float o[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (y[gl][dy][dx][q] + xch[((((gl * 2 + dy) * 2 + dx) * 4 + q) * 64) + lane]) + bb[q];
              const int yy = oy + dy, xx = ox + dx;
              if (yy < H && xx < W) {
                float4 ov = make_float4(o[0], o[1], o[2], o[3]);
                if (relu) { ov.x = fmaxf(ov.x, 0.f); ov.y = fmaxf(ov.y, 0.f); ov.z = fmaxf(ov.z, 0.f); ov.w = fmaxf(ov.w, 0.f); }
                if (pool) {
                  pmax[0] = fmaxf(pmax[0], ov.x); pmax[1] = fmaxf(pmax[1], ov.y);
                  pmax[2] = fmaxf(pmax[2], ov.z); pmax[3] = fmaxf(pmax[3], ov.w);
                } else {
                  *reinterpret_cast<float4*>(out + (((long)(co >> 3) * H + yy) * W + xx) * 8 + kk * 4) = ov;
                }
              }
            }
          if (pool && oy < H && ox < W) {
            const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
            *reinterpret_cast<float4*>(out + (((long)(co >> 3) * OH + (oy >> 1)) * OW + (ox >> 1)) * 8 + kk * 4) =
                make_float4(pmax[0], pmax[1], pmax[2], pmax[3]);
          }
        }
      }
      // the exchange half is rewritten by s = 1, the buffer by the next block's halo: both behind every reader
      if (s == 0 || u < u1) __syncthreads();
    }
  }
  if ((ABL & 32) && tid == 0) {                                   // tuning: shader cycles and wall time (100 MHz) of the range
    unsigned long long* st = reinterpret_cast<unsigned long long*>(part + (long)G * 2 * kSTileFloats) + (long)l * 8;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    st[0] = stamp_k; st[1] = stamp_w; st[2] = wall_clock64(); st[3] = __builtin_readcyclecounter() - stamp_c; st[4] = u1 - u0;
    st[5] = hw; st[6] = blockIdx.x;
  }
}

// Finishes the tiles that a range boundary cut: block l owns the tile that BEGINS inside range l and ends outside it (if any),
// adds its pieces in range order (range l's last-tile slot, then the first-tile slots of the ranges that follow up to the
// tile's end) and applies bias / ReLU / pooling exactly as the kernel's epilogue does.  Thread = (row group, lane) of the
// kernel's hf = 0 waves: same pixels, same channels.
template <int POOL>
__global__ __launch_bounds__(128) void wino_stream_fix_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              float* __restrict__ out, int H, int W, int Cin, int Cout, int relu,
                                                              int tiles_x, int u2, int G) {
  const int l = blockIdx.x, nch = Cin >> 3, ncot = Cout >> 5;
  const int s_l = stream_start(l, u2, G), e_l = stream_start(l + 1, u2, G);
  if (e_l % nch == 0) return;                                     // the range ends with a tile
  const int t = e_l / nch, tb = t * nch, te = tb + nch;
  if (tb < s_l) return;                                           // the tile began in an earlier range: that block owns it
  const int tid = threadIdx.x, rg = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kk = lane >> 5, ty = j >> 4, tx = j & 15;
  const int p_ = t / ncot, bx = p_ % tiles_x, by = p_ / tiles_x, cot = t - p_ * ncot;
  const int oy = by * 8 + 4 * rg + 2 * ty, ox = bx * kWCols + 2 * tx;
  float4 acc[16];
  {
    const float* src = part + ((long)l * 2 + (s_l / nch == t ? 0 : 1)) * kSTileFloats;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = *reinterpret_cast<const float4*>(src + ((i * 2 + rg) * 64 + lane) * 4);
  }
  for (int l2 = l + 1; l2 < G && stream_start(l2, u2, G) < te; ++l2) {
    const float* src = part + (long)l2 * 2 * kSTileFloats;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 q = *reinterpret_cast<const float4*>(src + ((i * 2 + rg) * 64 + lane) * 4);
      acc[i].x += q.x; acc[i].y += q.y; acc[i].z += q.z; acc[i].w += q.w;
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int co = cot * 32 + g * 8 + kk * 4;
    const float4 b = *reinterpret_cast<const float4*>(bias + co);
    float4 best = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float4 v = acc[(g * 2 + dy) * 2 + dx];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        const int yy = oy + dy, xx = ox + dx;
        if (yy < H && xx < W) {
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (POOL) {
            best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
          } else {
            *reinterpret_cast<float4*>(out + (((long)(co >> 3) * H + yy) * W + xx) * 8 + kk * 4) = v;
          }
        }
      }
    if (POOL && oy < H && ox < W) {
      const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
      *reinterpret_cast<float4*>(out + (((long)(co >> 3) * OH + (oy >> 1)) * OW + (ox >> 1)) * 8 + kk * 4) = best;
    }
  }
}

template <int XCD, int ABL = 0>
int launch_stream(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                  int Cout, int relu, int pool, int tiles_x, int u2, int G, float* part) {
  constexpr size_t lds = 4 * (2 * (size_t)kSHaloStride + 2 * (size_t)kWPanel);
  static_assert(lds <= 80 * 1024, "conv3x3_wino_stream: two workgroups per CU");
  auto kern = conv3x3_wino_stream_kernel<XCD, ABL>;
  static std::atomic<unsigned long long> attr_set{0};            // one bit per device: function attributes are per device
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(G), dim3(kSNT), lds, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, pool, tiles_x,
                     u2, part);
  return MNC_OK;
}

}  // namespace

int wino_stream_launch(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                       int Cin, int Cout, int relu, int pool, int wgs_per_slot, int xcd_order) {
  const int abl = wgs_per_slot / 1000;                            // MNC_WINO_STREAM=abl,k (tuning builds)
  wgs_per_slot %= 1000;
  const int nch = Cin >> 3, ncot = Cout >> 5;
  if (nch % 2 || (double)Cin * H * W * 4.0 >= 2147483648.0 || wgs_per_slot < 1) return MNC_ERR_INVALID;
  const int tiles_x = cdiv(W, kWCols);
  const long tiles = (long)tiles_x * cdiv(H, 8) * ncot;
  const long u2 = tiles * nch / 2;
  if (u2 >= (1L << 30)) return MNC_ERR_INVALID;
  long G = 512L * wgs_per_slot;                                   // MI355X: 256 CUs x two resident workgroups
  if (G > u2) G = u2;
  int rc = ensure_scratch(ctx, (size_t)G * 2 * kSTileFloats * 4 + (size_t)G * 64);
  if (rc) return rc;
  float* part = (float*)ctx->scratch;
#ifdef MNC_TUNING
#define MNC_STREAM_ABL(A) if (abl == A) rc = launch_stream<1, A>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, pool, tiles_x, (int)u2, (int)G, part); else
  MNC_STREAM_ABL(1) MNC_STREAM_ABL(3) MNC_STREAM_ABL(4) MNC_STREAM_ABL(7) MNC_STREAM_ABL(8) MNC_STREAM_ABL(15) MNC_STREAM_ABL(16)
  MNC_STREAM_ABL(24) MNC_STREAM_ABL(31) MNC_STREAM_ABL(23) MNC_STREAM_ABL(32) MNC_STREAM_ABL(63) MNC_STREAM_ABL(47) MNC_STREAM_ABL(33) MNC_STREAM_ABL(34) MNC_STREAM_ABL(36) MNC_STREAM_ABL(40) MNC_STREAM_ABL(48) MNC_STREAM_ABL(35) MNC_STREAM_ABL(39) MNC_STREAM_ABL(56) MNC_STREAM_ABL(55) MNC_STREAM_ABL(59) MNC_STREAM_ABL(62)
#undef MNC_STREAM_ABL
#else
  if (abl) return MNC_ERR_INVALID;
#endif
  rc = xcd_order ? launch_stream<1>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, pool, tiles_x, (int)u2, (int)G, part)
                 : launch_stream<0>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, pool, tiles_x, (int)u2, (int)G, part);
  if (rc) return rc;
#ifdef MNC_TUNING
  if (abl & 32) {                                                 // per-range stamps -> a summary on stderr (synchronises: timing aid only)
    std::vector<unsigned long long> st((size_t)G * 8);
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    MNC_HIP_TRY(hipMemcpy(st.data(), part + (size_t)G * 2 * kSTileFloats, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> start, pro, loop, end, cpb, clk;
    unsigned long long t0 = ~0ull;
    for (long i = 0; i < G; ++i) t0 = std::min(t0, st[i * 8]);
    for (long i = 0; i < G; ++i) {
      const unsigned long long* r = &st[i * 8];
      start.push_back((r[0] - t0) * 0.01); pro.push_back((r[1] - r[0]) * 0.01); loop.push_back((r[2] - r[1]) * 0.01);
      end.push_back((r[2] - t0) * 0.01); cpb.push_back((double)r[3] / r[4]); clk.push_back(r[3] / ((r[2] - r[1]) * 10.0));
    }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    fprintf(stderr, "stream %dx%d %d->%d G=%ld: start p50 %.1f p90 %.1f max %.1f | prologue med %.1f max %.1f | loop min %.1f med %.1f p90 %.1f max %.1f | "
            "end min %.1f med %.1f max %.1f us | cycles/block med %.0f | clock med %.2f GHz\n", H, W, Cin, Cout, G, pct(start, .5), pct(start, .9),
            pct(start, 1), pct(pro, .5), pct(pro, 1), pct(loop, 0), pct(loop, .5), pct(loop, .9), pct(loop, 1), pct(end, 0), pct(end, .5),
            pct(end, 1), pct(cpb, .5), pct(clk, .5));
    fprintf(stderr, "   cycles/block min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f | clock min %.2f p10 %.2f p90 %.2f max %.2f\n", pct(cpb, 0), pct(cpb, .1),
            pct(cpb, .5), pct(cpb, .9), pct(cpb, 1), pct(clk, 0), pct(clk, .1), pct(clk, .9), pct(clk, 1));
    for (int x = 0; x < 8; ++x) {
      std::vector<double> lx, cx, kx;
      for (long i = 0; i < G; ++i)
        if ((int)(st[i * 8 + 6] & 7) == x) { lx.push_back(loop[i]); cx.push_back(cpb[i]); kx.push_back(clk[i]); }
      if (!lx.empty())
        fprintf(stderr, "   xcd %d: loop med %.1f max %.1f us, cycles/block med %.0f max %.0f, clock med %.2f\n", x, pct(lx, .5), pct(lx, 1), pct(cx, .5),
                pct(cx, 1), pct(kx, .5));
    }
  }
#endif
  if (pool)
    hipLaunchKernelGGL(wino_stream_fix_kernel<1>, dim3((int)G), dim3(128), 0, ctx->stream, part, d_bias, d_out, H, W, Cin, Cout, relu,
                       tiles_x, (int)u2, (int)G);
  else
    hipLaunchKernelGGL(wino_stream_fix_kernel<0>, dim3((int)G), dim3(128), 0, ctx->stream, part, d_bias, d_out, H, W, Cin, Cout, relu,
                       tiles_x, (int)u2, (int)G);
  return MNC_OK;
}

}  // namespace mnc
#endif  // MNC_TUNING
