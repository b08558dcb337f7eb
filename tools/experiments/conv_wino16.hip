// Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32 fragments, for gfx950: conv3x3_wino2_kernel<2, 7, 1, *> (conv_wino.hip -- the wave
// pairs, the rotated block loop, the LDS-DMA weight panel, the sections / K ranges of the launcher's tail plan, the fused pooling)
// with the contraction re-tiled.  Why: an fp32 MFMA's result write-back (32x32x2: 16 registers per 4096 FLOP) shares the register
// file's write ports with everything else a block does -- 20 LDS reads, 64 transform results, the staged halo -- and the pipe is
// power limited (tools/probes/mfma_f32_mix_probe.hip: the same FLOPs with this block's LDS reads and VALU
// beside them run 12 % fewer cycles as 16x16x4, 4 registers per 2048 FLOP; the InnerProduct gained 6-9 % from the same change).
//
// What changes against conv3x3_wino2_kernel (everything else -- global layouts, packed weights, staging, barriers -- is the same):
//   * instruction operands: lane l = (r = l % 16, g = l / 16) supplies A[row r][k = g] and B[col r][k = g]; D register i is row
//     4 g + i, column r.  A channel block's 8 input channels are two MFMAs (h = 0, 1) with k <-> channel 2 g + h: a lane works on
//     the channel PAIR (2 g, 2 g + 1) -- one ds_read_b64 per halo pixel and per (position, output-channel half) weight fragment.
//   * a wave's 32 Winograd tiles x 32 output channels x 8 positions are 8 x 2 x 2 accumulators [position][channel half sr]
//     [tile row tg] of 16 x 16: lane r holds tile column tx = r of BOTH tile rows and output channels 16 sr + 4 g .. + 3.
//   * halo pixels keep their 12-float pitch; the two 16-byte channel halves of a pixel are swapped where (column >> 4) is odd
//     (instead of by row): the 32 lanes of a ds_read_b64 service group -- tile columns 0-15 x channel pairs {0, 1} or {2, 3} -- start
//     on banks 24 tx + 4 (tx >> 3) + 2 (g & 1) (mod 64), all different: conflict-free.  Weight rows (68 floats) put output channel
//     j on bank 4 j: conflict-free as they are.
//   * per block and wave: 24 + 16 ds_read_b64 (same bytes as 12 + 8 ds_read_b128), 64 transform results, 64 MFMAs of 32 cycles.
//
// STATUS: a measurement build (-DMNC_TUNING, MNC_WINO_MFMA16=1), not the product path.  Correct (tests/test_gpu_ops.py runs it against
// torch and the direct kernel, fused pooling and K ranges included) and EXACTLY as fast as the 32x32x2 kernel (13-layer trunk 2.114
// vs 2.123 ms, kernel_bench convwino): what the lighter write-back gives, the instruction stream takes -- ~245 instructions per
// block and wave (64 MFMAs, 40 LDS reads, ~75 VALU, ~45 scalar) against ~150, about one issue slot per 8 cycles of a SIMD that two
// waves share.  (With hipcc's default merging of neighbouring reads into ds_read2_b64 -- half rate, 32-dword banking -- it was 6 %
// slower: hence the volatile reads.)  The InnerProduct, whose loop carries 12 LDS reads and no VALU per 80 MFMAs, gained 6-9 % from
// the same re-tiling (gemm.hip: fc_mfma_dma16_kernel).
#ifdef MNC_TUNING
#include <atomic>

#include "wino_common.h"

namespace mnc {

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kNT = 256;
constexpr int kHaloRows = 10;
constexpr int kHaloFloats = kHaloRows * kWHaloCols * kWPixPitch;
constexpr int kHaloVec = kHaloRows * kWHaloCols * 2;
constexpr int kHPer = (kHaloVec + kNT - 1) / kNT;
constexpr int kDma = (kWPanel / 256 + 3) / 4;

template <int XCD>
__global__ __launch_bounds__(kNT, 2) void conv3x3_wino16_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                                const float* __restrict__ bias, float* __restrict__ out, int H,
                                                                int W, int Cin, int Cout, int relu, int ksplit_a,
                                                                float* __restrict__ part, int tiles_x, int pool_a, int pix_a,
                                                                int ksplit_b) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];   // halo[2] then weights[2]; the epilogue reuses it
  float* const s_halo = s_mem;
  float* const s_w = s_mem + 2 * kHaloFloats;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int rg = wave & 1, hf = wave >> 1;                        // waves w and w + 2 are a pair (same tiles, other positions)
  const int r16 = lane & 15, g4 = lane >> 4;                      // tile column / output-channel row; channel pair (2 g4, 2 g4 + 1)
  const int ncot = Cout >> 5;
  // block -> (section, K range, channel tile, pixel tile): conv3x3_wino2_kernel's decode
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
  const int w0 = bx * kWCols, h0 = by * 8, co0 = cot * 32;
  const int chunk0 = split * (Cin >> 3) / ksplit;
  const int nchunks = (split + 1) * (Cin >> 3) / ksplit - chunk0;

  // ---- staging: halo through registers (buffer loads; a piece outside the image reads zeros), weight panel by LDS-DMA
  int h_off[kHPer], hb_off[kHPer];
#pragma unroll
  for (int u = 0; u < kHPer; ++u) {
    const int q = min(tid + u * kNT, kHaloVec - 1);
    const int pix = q >> 1, half = q & 1;
    const int r = pix / kWHaloCols, c = pix - r * kWHaloCols;
    const int gh = h0 - 1 + r, gw = w0 - 1 + c;
    h_off[u] = pix * kWPixPitch + (half ^ ((c >> 4) & 1)) * 4;
    hb_off[u] = (gh >= 0 && gh < H && gw >= 0 && gw < W) ? ((gh * W + gw) * 8 + half * 4) * 4 : 0x7FFFFFF0;
  }
  const long plane = (long)H * W * 8;
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)min((long)Cin * H * W * 4, 0x7FFFFFFFL), 0x00020000);
  float4 Gh[kHPer];
  auto load_chunk = [&](int c) {
    c = chunk0 + min(c, nchunks - 1);
    const int hs = __builtin_amdgcn_readfirstlane(c * (int)(plane * 4));
#pragma unroll
    for (int u = 0; u < kHPer; ++u) {
      const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, hb_off[u], hs, 0);
      Gh[u] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
    }
  };
  auto dma_panel = [&](int c, int buf) {
    c = chunk0 + min(c, nchunks - 1);
    const float* src = wpk + ((long)c * ncot + cot) * kWPanel;
    float* dstw = s_w + buf * kWPanel;
#pragma unroll
    for (int i = 0; i < kDma; ++i) {
      const int piece = min(wave + i * 4, kWPanel / 256 - 1);     // branch-free: the waves without a last piece repeat piece 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dstw + piece * 256), 16, 0, 0);
    }
  };
  auto store_chunk = [&](int buf) {
    float* hdst = s_halo + buf * kHaloFloats;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) *reinterpret_cast<float4*>(hdst + h_off[u]) = Gh[u];
  };

  f32x4v acc[8][2][2];                                            // [position][output-channel half][tile row]
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[p][a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // halo rows by ROLE (conv_wino.hip): hf = 0: (A, B, C) = (d0, d1, d2): t0 = A - C, t1 = C + B;  hf = 1: (d2, d3, d1): t0 = A - C,
  // t1 = C - B.  Tile row tg of the lane: halo rows 4 rg + 2 tg + role; columns 2 tx + cc.  The swap of a pixel's halves follows
  // (column >> 4) & 1, which is the same for cc = 0, 1 and for cc = 2, 3: two bases per role.
  const int rowA = hf ? 2 : 0, rowB = hf ? 3 : 1, rowC = hf ? 1 : 2;
  const float sgn = hf ? -1.f : 1.f;
  const f32x2v sgn2 = {sgn, sgn};
  auto col_off = [&](int cc) { return (2 * r16 + cc) * kWPixPitch + ((g4 >> 1) ^ (((2 * r16 + cc) >> 4) & 1)) * 4 + 2 * (g4 & 1); };
  auto row_base = [&](int row) { return (4 * rg + row) * kWHaloCols * kWPixPitch; };
  const int c01 = col_off(0), c23 = col_off(2);                   // (cc = 1, 3: + kWPixPitch)
  const int offA0 = row_base(rowA) + c01, offA2 = row_base(rowA) + c23;
  const int offB0 = row_base(rowB) + c01, offB2 = row_base(rowB) + c23;
  const int offC0 = row_base(rowC) + c01, offC2 = row_base(rowC) + c23;
  constexpr int kTg = 2 * kWHaloCols * kWPixPitch;                // tile row 1: two halo rows down
  // weights: row (k half g4 >> 1, output channel r16 (+ 16 sr)), element 4 p + 2 (g4 & 1) + h of the wave's positions 8 hf + p
  const int u_base = ((g4 >> 1) * 32 + r16) * kWRowPitch + hf * 32 + 2 * (g4 & 1);
  constexpr int kSr = 16 * kWRowPitch;

  f32x2v fA[2][4], fB[2][4], fC[2][4];                            // [tile row][column]
  f32x2v ua[2][2];                                                // weights of the position pair about to be multiplied [p & 1][sr]
  f32x2v v[2][8], t0[2][4], t1[2][4];
  // volatile: hipcc otherwise merges neighbouring reads into ds_read2_b64, which runs at half the rate of two ds_read_b64 and banks
  // by 32 dwords (MI355X_MICROARCH.md, LDS table) -- the layouts above are conflict-free for ds_read_b64
  typedef __attribute__((address_space(3))) const volatile f32x2v lds_f32x2v;
  auto rd2 = [&](const float* p) -> f32x2v { return *(lds_f32x2v*)(p); };
  auto read_AC = [&](int buf) {
    const float* sh = s_halo + buf * kHaloFloats;
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      fA[tg][0] = rd2(sh + offA0 + tg * kTg);
      fA[tg][1] = rd2(sh + offA0 + tg * kTg + kWPixPitch);
      fA[tg][2] = rd2(sh + offA2 + tg * kTg);
      fA[tg][3] = rd2(sh + offA2 + tg * kTg + kWPixPitch);
      fC[tg][0] = rd2(sh + offC0 + tg * kTg);
      fC[tg][1] = rd2(sh + offC0 + tg * kTg + kWPixPitch);
      fC[tg][2] = rd2(sh + offC2 + tg * kTg);
      fC[tg][3] = rd2(sh + offC2 + tg * kTg + kWPixPitch);
    }
    const float* sw = s_w + buf * kWPanel + u_base;
    ua[0][0] = rd2(sw);
    ua[0][1] = rd2(sw + kSr);
    ua[1][0] = rd2(sw + 4);
    ua[1][1] = rd2(sw + 4 + kSr);
  };
  auto read_B = [&](int buf) {
    const float* sh = s_halo + buf * kHaloFloats;
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      fB[tg][0] = rd2(sh + offB0 + tg * kTg);
      fB[tg][1] = rd2(sh + offB0 + tg * kTg + kWPixPitch);
      fB[tg][2] = rd2(sh + offB2 + tg * kTg);
      fB[tg][3] = rd2(sh + offB2 + tg * kTg + kWPixPitch);
    }
  };
  auto rows_t = [&]() {                                           // the wave's two rows of t = B^T d, both tile rows
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        t0[tg][q] = fA[tg][q] - fC[tg][q];
        t1[tg][q] = __builtin_elementwise_fma(sgn2, fB[tg][q], fC[tg][q]);
      }
  };
  auto transform_row = [&](const f32x2v (&t)[2][4], int o) {      // column pass: positions o .. o + 3
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      v[tg][o + 0] = t[tg][0] - t[tg][2];
      v[tg][o + 1] = t[tg][1] + t[tg][2];
      v[tg][o + 2] = t[tg][2] - t[tg][1];
      v[tg][o + 3] = t[tg][1] - t[tg][3];
    }
  };
  // positions p, p + 1: 2 positions x 2 channel halves x 2 tile rows x (h = 0, 1) = 16 MFMAs on 8 accumulators
  auto mfma_pair = [&](int p, const f32x2v (&u)[2][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
          for (int tg = 0; tg < 2; ++tg)
            acc[p + pp][sr][tg] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pp][sr][h], v[tg][p + pp][h], acc[p + pp][sr][tg], 0, 0, 0);
  };
  auto read_u = [&](const float* sw, int p, f32x2v (&u)[2][2]) {
    u[0][0] = rd2(sw + p * 4);
    u[0][1] = rd2(sw + p * 4 + kSr);
    u[1][0] = rd2(sw + p * 4 + 4);
    u[1][1] = rd2(sw + p * 4 + 4 + kSr);
  };

  // ---- prologue: blocks 0 and 1 requested together (one global round trip in front of the first MFMA)
  {
    load_chunk(0);
    dma_panel(0, 0);
    float4 G0[kHPer];
#pragma unroll
    for (int u = 0; u < kHPer; ++u) G0[u] = Gh[u];
    load_chunk(1);
    dma_panel(1, 1);
#pragma unroll
    for (int u = 0; u < kHPer; ++u) *reinterpret_cast<float4*>(s_halo + h_off[u]) = G0[u];
  }
  __syncthreads();
  read_AC(0);
  read_B(0);
  rows_t();
  transform_row(t0, 0);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const float* sw = s_w + buf * kWPanel + u_base;
    __builtin_amdgcn_sched_barrier(0);
    // positions 0-1, the second row of the column pass under them
    f32x2v un[2][2];
    read_u(sw, 2, un);
    transform_row(t1, 4);
    mfma_pair(0, ua);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // (first an MFMA: the loop-carried fragments are in registers)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // positions 2-3; the last fragments of block c in, block c + 1's halo out
    f32x2v um[2][2], uq[2][2];
    read_u(sw, 4, um);
    read_u(sw, 6, uq);
    store_chunk(buf ^ 1);
    mfma_pair(2, un);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i % 4 == 1 && i / 4 < kHPer) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // positions 4-5: block c + 1's rows and first fragments requested behind the first MFMA, then block c + 2 from memory
    read_AC(buf ^ 1);
    read_B(buf ^ 1);
    load_chunk(c + 2);
    dma_panel(c + 2, buf);
    mfma_pair(4, um);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 28, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    // positions 6-7, block c + 1's transform rows and the first row of its column pass under them
    rows_t();
    transform_row(t0, 0);               // (v[.][0..3] were last read by the MFMAs of positions 2-3)
    mfma_pair(6, uq);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
  }
  __syncthreads();                                                // every wave is done with the staging buffers

  // ---- epilogue.  acc[p][sr][tg][i], p = 4 * r + c with r the wave's local transform row (global row 2 hf + r): output channel
  // co0 + 16 sr + 4 g4 + i of tile (tile row tg, tile column r16).  Partial output transform of this wave's rows (hf = 0: s0 = M0 + M1,
  // s1 = M1; hf = 1: s0 = M2, s1 = -(M2 + M3)), the pair's sums added through LDS: conv_wino.hip.
  float* xch = s_mem + ((long)rg * 64 * 64);                      // [value 0..63][lane] floats per row group
  float* dst = out;
  if (ksplit > 1) dst = part + (long)split * Cout * H * W;
#pragma unroll
  for (int un_ = 0; un_ < 4; ++un_) {                             // unit = (channel half sr, tile row tg)
    const int sr = un_ >> 1, tg = un_ & 1;
    const int co = co0 + 16 * sr + 4 * g4;
    const int oy = h0 + 4 * rg + 2 * tg, ox = w0 + 2 * r16;
    float y[2][2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s0[4], s1[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float m0 = acc[cc][sr][tg][q], m1 = acc[4 + cc][sr][tg][q];
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
          for (int q = 0; q < 4; ++q) xch[(((un_ * 2 + dy) * 2 + dx) * 4 + q) * 64 + lane] = y[dy][dx][q];
    }
    __syncthreads();                                             // (uniform: every wave runs all four iterations)
    if (hf == 0) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ksplit == 1) b = *reinterpret_cast<const float4*>(bias + co);
      const float bb[4] = {b.x, b.y, b.z, b.w};
      // pool != 0 (ksplit == 1 only): the following Pooling MAX 2x2 stride 2 applied here -- a Winograd tile IS a pooling window
      float pmax[4] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (y[dy][dx][q] + xch[(((un_ * 2 + dy) * 2 + dx) * 4 + q) * 64 + lane]) + bb[q];
          const int yy = oy + dy, xx = ox + dx;
          if (yy < H && xx < W) {
            float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            if (relu && ksplit == 1) { ov.x = fmaxf(ov.x, 0.f); ov.y = fmaxf(ov.y, 0.f); ov.z = fmaxf(ov.z, 0.f); ov.w = fmaxf(ov.w, 0.f); }
            if (pool) {
              pmax[0] = fmaxf(pmax[0], ov.x); pmax[1] = fmaxf(pmax[1], ov.y);
              pmax[2] = fmaxf(pmax[2], ov.z); pmax[3] = fmaxf(pmax[3], ov.w);
            } else {
              *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * H + yy) * W + xx) * 8 + (co & 7)) = ov;
            }
          }
        }
      if (pool && oy < H && ox < W) {
        const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;             // ceil((n - 2) / 2) + 1 for n >= 2
        *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * OH + (oy >> 1)) * OW + (ox >> 1)) * 8 + (co & 7)) =
            make_float4(pmax[0], pmax[1], pmax[2], pmax[3]);
      }
    }
  }
}

}  // namespace

int wino16_launch(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                  int Cout, int relu, int ksplit, float* part, int pool, int pix_a, int ksplit_b, int xcd_order) {
  constexpr size_t lds_stage = 2 * 4 * ((size_t)kHaloFloats + (size_t)kWPanel);
  constexpr size_t lds_xch = (size_t)2 * 64 * 64 * 4;
  constexpr size_t lds = lds_stage > lds_xch ? lds_stage : lds_xch;
  static_assert(lds <= 80 * 1024, "conv3x3_wino16: two workgroups per CU");
  static std::atomic<unsigned long long> attr_set{0};            // one bit per device: function attributes are per device
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const int tiles_x = cdiv(W, kWCols);
  const int pix = tiles_x * cdiv(H, 8);
  if (pix_a < 0 || pix_a > pix) pix_a = pix;                     // no second section
  dim3 grid((pix_a * ksplit + (pix - pix_a) * ksplit_b) * (Cout >> 5));
  if (xcd_order)
    hipLaunchKernelGGL(conv3x3_wino16_kernel<1>, grid, dim3(kNT), lds, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu,
                       ksplit, part, tiles_x, pool, pix_a, ksplit_b);
  else
    hipLaunchKernelGGL(conv3x3_wino16_kernel<0>, grid, dim3(kNT), lds, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu,
                       ksplit, part, tiles_x, pool, pix_a, ksplit_b);
  return MNC_OK;
}

}  // namespace mnc
#endif  // MNC_TUNING
