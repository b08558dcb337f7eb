// 3x3 convolution (pad 1, stride 1) by Winograd's minimal filtering F(4x4, 3x3) on the fp32 matrix pipe, fused, for gfx950
// (models/VGG16/mnc_5stage/test.prototxt:41-412: the trunk's 3x3 layers behind conv1_1 and rpn_conv_3x3; round 4).
//
// Why.  F(2x2, 3x3) (conv_wino.hip) spends 16 multiplies per 2x2 outputs = 4 per output and its loop sits at what the fp32 pipe
// sustains beside its operand traffic.  F(4x4, 3x3) computes a 4x4 output tile from a 6x6 input tile with 36
// multiplies per (input channel, output channel) = 2.25 per output: 0.5625x the matrix-pipe work for the same result.
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          (Lavin & Gray 2015, interpolation points 0, +-1, +-2)
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// The filter transform is evaluated in double and rounded once; input and output transforms are fp32 fma chains.  Rounding error
// measured 5e-6 .. 1.5e-5 of the output range for 64 .. 512 input channels (tools/studies/winograd_f4_error.py; F(2x2): 3e-7 ..
// 7e-7, the direct fp32 sum 1e-6 .. 2e-6) -- under the kernels' 1e-4 bar and two orders under the path's 1e-3.
//
// Mapping:
//   * the contraction over input channels of each of the 36 transform positions is a GEMM  M_p[co][tile] += U_p[co][ci] V_p[ci][tile]
//     on v_mfma_f32_16x16x4_f32: A = U (lane: co = l & 15, k = l >> 4), B = V (lane: tile = l & 15, k = l >> 4), D: lane holds
//     co = 4 (l >> 4) + e of tile l & 15.
//   * workgroup = 4 waves = 2 tile rows x 2 POSITION HALVES = 32 output channels x 32 tiles (8 x 64 pixels).  The two waves of a tile
//     row (4 pixel rows x 64 columns = 16 tiles) both work on all 32 output channels; wave hp owns transform rows 3 hp .. 3 hp + 2
//     (18 of the 36 positions): 18 positions x 2 channel groups = 36 accumulators of 4 registers (144, architectural), 36 MFMAs per
//     channel, and HALF of the input transform each (see the note on the transform below: on this part the transform arithmetic,
//     not the MFMAs' operand traffic, is what a Winograd loop pays for).  The output transform is linear: each wave applies it to its
//     rows and the partner waves exchange partial sums through LDS once, in the epilogue.
//   * K is walked in 8-channel blocks (lane group k: channels 2k, 2k + 1).  A block is six steps of twelve MFMAs (one of the wave's
//     three transform rows x one channel of the pair); meanwhile the wave reads the windows of the NEXT block (30 ds_read_b64: five
//     window rows x six columns, both channels of the pair per read) and transforms them, both channels per instruction (72 packed
//     VALU instructions per block, in runs behind the MFMA runs), in the registers the current block frees row by row.
//   * 80 KB of LDS, TWO workgroups per CU (one per-CU workgroup of eight waves with 2 x 77 KB was measured first: 1.85 ms for the
//     trunk against 1.75 -- an 8-block workgroup spends a quarter of its life in a prologue burst and an epilogue nothing overlapped).
//     LDS = the two HALF weight panels [g][position half][lane][36 floats] (lane pitch 144 B = 9 x 16 B: every 16-lane group of a
//     ds_read_b128 covers all 64 banks; stored in global memory exactly as they sit in LDS) + two halo images (10 x 66 pixels,
//     22 KB): exactly half a CU's 160 KB.  Half panel g is read by pass (s, g) only and refilled for block s + 1 while the other
//     pass runs; halo s is read by passes (s - 1, 1) and (s, 0) and its buffer refilled for block s + 2 during pass (s, 1).  All
//     copies are buffer_load_dwordx4 ... lds (no staging registers, no ds_write pass); two barriers per block.
//   * halo image in LDS: dense 32-byte pixels (the DMA writes 16-byte pieces back to back), pixel column xh of a row sits in slot
//     (xh & 3) * 17 + (xh >> 2) and its two 16-byte channel halves are swapped where bit 5 of xh is set.  Window column c of tile t
//     is pixel 4 t + c: for a fixed c the 16 tiles of a wave read slots 17 (c & 3) + t (+ 1) -- consecutive 32-byte pixels -- and
//     tiles t, t + 8 (whose pixels are 256 bytes = all 64 banks apart) read opposite halves: each 32-lane group of a
//     ds_read_b64 covers the 64 banks exactly once.  Out-of-image pixels, pad slots and rows past the halo are buffer loads with an
//     out-of-range offset: the hardware writes zeros.  (tests/test_wino4_index_math.py emulates all of this on the CPU.)
//   * epilogue: z = M A along the wave's three rows, the rows' share of A^T z, exchange (12 KB per wave), the finished 4x4 tile per
//     lane, + bias, ReLU, optionally the following Pooling MAX 2x2/2 (a 4x4 tile holds four whole pooling windows), stores into the
//     c8 layout through the wave's own 16 KB of LDS (whole cache lines); K ranges write raw partial outputs (the transform is
//     linear) that wino4_section_reduce_kernel finishes.
// What it costs (kernel_bench convwino4 ablations, 13-layer trunk unpooled, MI355X; tuning builds, MNC_WINO_F4, operands that cost no
// instruction in place of the reads): all 1.43 ms; without the window reads 1.29, without the weight reads 1.35, without both 1.21;
// without the copies 1.28 (weight copies and halo copies about half each); without the output stores 1.33; without waits and
// barriers 1.42; without the transform arithmetic 1.40 (it was 0.35 of 1.66 with the scalar transform, 0.23 of 1.60 packed, before
// the position split).  On exact tilings the loop alone reaches 88 TFLOP/s of MFMA work, 149 with everything but the MFMAs removed
// (the pipe's peak: profiles/r04_wino4_ablations.txt).  F(2x2,3x3) (conv_wino.hip): 2.13 ms.
// Built and measured on the way (git history, profiles/r04_wino4_variants.txt): one 8-wave workgroup per CU with whole-block
// buffers, transform before the MFMAs (1.85 ms) and half-block pipelined (1.94); one wave per SIMD owning 32 channels x 16 tiles
// = 288 accumulators (hipcc shuttles them between the AGPR and VGPR halves of the file: 2.09); waves splitting the output channels,
// every wave transforming all 36 positions with scalar fma (1.66) and with packed fp32 (1.60); the position split with register
// pairs of window COLUMNS, one channel per pass and ds_read_b32 windows (1.46).
#include <atomic>
#include <type_traits>

#include "mnc_internal.h"

namespace mnc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kF4Rows = 8;                           // output rows per workgroup (2 tile rows)
constexpr int kF4Cols = 64;                          // output columns per workgroup (16 tiles)
constexpr int kF4HaloRows = kF4Rows + 2;
constexpr int kF4RowSlots = 68;                      // pixel slots per halo row: 4 residues x 17 (66 used)
constexpr int kF4RowBytes = kF4RowSlots * 32;        // 2176
constexpr int kF4HaloPieces = 22;                    // 10 x 2176 = 21760 bytes -> 22 DMA pieces of 1 KB
constexpr int kF4HaloBytes = kF4HaloPieces * 1024;   // 22528
constexpr int kF4LanePitch = 36;                     // floats per lane of a HALF panel: the 36 positions of one channel (144 B = 9 x 16 B)
constexpr int kF4HalfFloats = 2 * 64 * kF4LanePitch;             // half panel [channel group][lane][36]: 4608 floats
constexpr int kF4HalfBytes = kF4HalfFloats * 4;                  // 18432
constexpr int kF4HalfPieces = kF4HalfBytes / 1024;               // 18
constexpr int kF4PanelFloats = 2 * kF4HalfFloats;                // per (channel block, 32-channel tile): [g][cg][lane][36] = 9216 floats
constexpr int kF4PanelBytes = kF4PanelFloats * 4;                // 36864
constexpr int kF4LdsBytes = 2 * kF4HalfBytes + 2 * kF4HaloBytes; // the two half panels + two halo images: 81920 = half a CU's LDS
static_assert(kF4HalfPieces * 1024 == kF4HalfBytes, "whole DMA pieces per half panel");
static_assert(kF4HaloBytes >= kF4HaloRows * kF4RowBytes, "halo image fits its pieces");
static_assert(2 * kF4LdsBytes <= 160 * 1024, "conv3x3_wino4: two workgroups per CU");

// The INPUT TRANSFORM is what the loop pays for next to its MFMAs: the fp32 MFMA and the VALU share the SIMD's multipliers
// (tools/probes/pk_f32_under_mfma_probe.hip: beside an fp32 MFMA stream EVERY VALU instruction costs the stream 4-6 cycles, packed
// or not, dependent or not; nothing overlaps -- 4 v_fma_f32 per MFMA: 97 TFLOP/s of 155, 2 v_pk_fma_f32: 102, 1: 113, and a run of
// VALU behind a run of MFMAs is cheaper than alternating them), so the instruction COUNT is what matters.  History of this kernel
// (kernel_bench convwino4, trunk + rpn): scalar transform, 144 VALU per 36 MFMAs, 1.68 ms (without the transform arithmetic:
// 1.32); packed fp32, 72: 1.60; packed + the transform SPLIT between the two waves of a tile row (below), 36: see DESIGN.md.
//   * Packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes of arithmetic per instruction and register pair).  A pair holds the two
//     channels 2k, 2k + 1 of one position -- what one ds_read_b64 of the halo image delivers.
//   * The two waves that share 16 tiles do not split the 32 output channels (each would transform all 36 positions of every
//     channel: the same arithmetic twice) but the 36 POSITIONS: wave hp owns transform rows 3 hp .. 3 hp + 2 (all six columns) of
//     all 32 output channels -- 18 positions x 2 channel groups = the same 36 accumulators and 36 MFMAs per pass, half the
//     transform.  V = B^T d B: first dimension DOWN the window columns, only the wave's three rows of B^T d (f4_bt_col<hp>: 6
//     instructions per column), second dimension ALONG those three rows (f4_bt2: 12 per row): 72 instructions per channel pair =
//     36 per 36 MFMAs.  The price: the output transform needs all 36
//     positions of an (output channel, tile) -- each wave applies A^T . A to its half (it is linear) and the two exchange partial
//     sums through LDS once per workgroup, in the epilogue.
// Inline assembly, not vector C++: hipcc's pre-emit peephole UNPACKS a v_pk_*_f32 it finds behind an MFMA into two scalar
// instructions (two thirds of a vector-typed version of the transform came out scalar, no faster than the scalar code).
struct F4K {                                         // the constant pair (-5, -5) in scalar registers (-5 is not an inline constant)
  unsigned long long m5;
};
__device__ __forceinline__ F4K f4_consts() {
  F4K k;
  k.m5 = 0xC0A00000C0A00000ull;
  asm volatile("" : "+s"(k.m5));                     // (opaque: stays in one scalar register pair)
  return k;
}
// First dimension, one column pair: rows r0 .. r5 of the window (pairs over the two columns) -> rows 3 HP .. 3 HP + 2 of B^T d.
//   HP = 0: x0 = 4 d0 - 5 d2 + d4, x1 = (d4 - 4 d2) + (d3 - 4 d1), x2 = (d4 - 4 d2) - (d3 - 4 d1)          (d5 unused)
//   HP = 1: x3 = (d4 - d2) + 2 (d3 - d1), x4 = (d4 - d2) - 2 (d3 - d1), x5 = 4 d1 - 5 d3 + d5                (d0 unused)
template <int HP>
__device__ __forceinline__ void f4_bt_col(const F4K& K, const f32x2 (&r)[6], f32x2& y0, f32x2& y1, f32x2& y2) {
  f32x2 u, a, b;
  if (HP == 0) {
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u) : "v"(r[2]), "s"(K.m5), "v"(r[4]));
    asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(y0) : "v"(r[0]), "v"(u));
    asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(r[2]), "v"(r[4]));
    asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(b) : "v"(r[1]), "v"(r[3]));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(y1) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(y2) : "v"(a), "v"(b));
  } else {
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a) : "v"(r[4]), "v"(r[2]));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(b) : "v"(r[3]), "v"(r[1]));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u) : "v"(r[3]), "s"(K.m5), "v"(r[5]));
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]" : "=v"(y0) : "v"(b), "v"(a));
    asm("v_pk_fma_f32 %0, %1, -2.0, %2 op_sel_hi:[1,0,1]" : "=v"(y1) : "v"(b), "v"(a));
    asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(y2) : "v"(r[1]), "v"(u));
  }
}
// One whole dimension on pairs, in place (12 instructions): the scalar chain on both halves.
__device__ __forceinline__ void f4_bt2(const F4K& K, f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5) {
  f32x2 a, b, c, e, t0, t5, x0, x1, x2, x3, x4, x5;
  asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(d2), "v"(d4));
  asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(b) : "v"(d1), "v"(d3));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(c) : "v"(d4), "v"(d2));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e) : "v"(d3), "v"(d1));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(d2), "s"(K.m5), "v"(d4));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t5) : "v"(d3), "s"(K.m5), "v"(d5));
  asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(x0) : "v"(d0), "v"(t0));
  asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(x5) : "v"(d1), "v"(t5));
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(x1) : "v"(a), "v"(b));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x2) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]" : "=v"(x3) : "v"(e), "v"(c));
  asm("v_pk_fma_f32 %0, %1, -2.0, %2 op_sel_hi:[1,0,1]" : "=v"(x4) : "v"(e), "v"(c));
  d0 = x0; d1 = x1; d2 = x2; d3 = x3; d4 = x4; d5 = x5;
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// One dimension of the output transform: y = A^T m (6 values -> 4), on pairs (two of a lane's four output channels).
__device__ __forceinline__ void f4_at2(f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2& y0, f32x2& y1, f32x2& y2,
                                       f32x2& y3) {
  const f32x2 p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4;
  y0 = (m0 + p) + r;
  y1 = fma2(f32x2{2.f, 2.f}, s, q);
  y2 = fma2(f32x2{4.f, 4.f}, r, p);
  y3 = fma2(f32x2{8.f, 8.f}, s, q) + m5;
}

// Block order.  The dispatcher puts block b on XCD b % 8 (used for speed only): blocks are re-numbered so that every XCD gets a
// contiguous range of the logical order (output-channel tile fastest, then K range, then pixel tile) -- the Cout / 32 workgroups
// that read one halo run on one XCD at about the same time.  Two SECTIONS (the launcher's tail plan): blocks [0, n_a) are the pixel
// tiles [0, pix_a) with ksplit_a K ranges each, the blocks behind them the remaining tiles with ksplit_b ranges.  A tile with one
// range writes the finished output; with several, each range writes raw partial outputs to its plane of `part`.
// ABL (tuning builds only, wrong results): 1 no DMA inside the loop, 2 no halo reads (opaque register constants), 4 no input
// transform, 8 no weight-fragment reads, 16 no wait / barrier, 32 no output stores -- what each part of a block costs (kernel_bench convwino4, MNC_WINO_F4).
template <int XCD, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_wino4_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            float* __restrict__ part, int H, int W, int Cin, int Cout, int relu,
                                                            int pool_a, int tiles_x, int pix_a, int ksplit_a, int ksplit_b,
                                                            unsigned* __restrict__ tickets) {
  extern __shared__ __attribute__((aligned(1024))) char s_f4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hp = wave & 1, tg = wave >> 1;                        // 4 waves: position half (transform rows 3 hp .. 3 hp + 2) x tile row
  const int t = lane & 15, k = lane >> 4;
  const int ncot = Cout >> 5;
  int bx, by, cot, split, ksplit, tile;            // tile: index of the (pixel tile, channel tile) inside its section
  {
    // section a = the first pix_a pixel tiles, section b (the tail of a plan) the rest.  (The tail's ranges FIRST in the block
    // order was measured neutral in round 5 -- profiles/r05_ab_switches_and_plans.txt -- and removed in round 6.)
    const int n_a = pix_a * ncot * ksplit_a, n_b = (int)gridDim.x - n_a;
    int b = blockIdx.x, total = n_a, pix0 = 0;
    ksplit = ksplit_a;
    if (b >= n_a) { b -= n_a; total = n_b; pix0 = pix_a; ksplit = ksplit_b; }
    const int q = total >> 3, r = total & 7, xcd = b & 7, idx = b >> 3;
    const int logical = XCD ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : b;
    const int nz = ncot * ksplit;
    const int bz = logical % nz;
    const int pixt = pix0 + logical / nz;
    bx = pixt % tiles_x;
    by = pixt / tiles_x;
    split = bz / ncot;
    cot = bz - split * ncot;
    tile = (logical / nz) * ncot + cot;
  }
  // K ranges: with `tickets` the tile's last arriver finishes it inside the launch (epilogue), so it pools like an unsplit tile
  const int pool = (ksplit == 1 || tickets) ? pool_a : 0;
  const int w0 = bx * kF4Cols, h0 = by * kF4Rows;
  const int nblk = Cin >> 3;
  const int chunk0 = split * nblk / ksplit;
  const int nchunks = (split + 1) * nblk / ksplit - chunk0;
  const long plane = (long)H * W * 8;                              // floats per 8-channel block of the input

  // ---- DMA assignment (fixed per thread).  Halo piece p = wave + 4 i covers LDS bytes [1024 p, 1024 p + 1024) of the halo image:
  // lane L fills 16-byte slot 64 p + L = pixel slot q = 32 p + (L >> 1) (row q / 68, slot q % 68 = a * 17 + b <-> pixel column
  // xh = 4 b + a), half position L & 1, i.e. it fetches channel half (L & 1) ^ ((xh >> 5) & 1) of that pixel.
  // Out-of-image pixels, pad slots and rows past the halo carry an out-of-range offset: the buffer load answers them with zeros.
  auto halo_off = [&](int i) {
    const int q = 32 * min(wave + 4 * i, kF4HaloPieces - 1) + (lane >> 1);       // (the waves without a sixth piece repeat piece 21)
    const int row = q / kF4RowSlots, sl = q - row * kF4RowSlots;
    const int a = sl / 17, b = sl - a * 17;
    const int xh = 4 * b + a;
    const int half = (lane & 1) ^ ((xh >> 5) & 1);
    const int y = h0 - 1 + row, x = w0 - 1 + xh;
    const bool ok = row < kF4HaloRows && xh < kF4Cols + 2 && y >= 0 && y < H && x >= 0 && x < W;
    return ok ? ((y * W + x) * 8 + half * 4) * 4 : 0x7FFFFFF0;             // byte offset inside an 8-channel block of the input
  };
  int h_off[6];                                      // (six registers through the loop: rebuilding them costs 70 VALU per block)
#pragma unroll
  for (int i = 0; i < 6; ++i) h_off[i] = halo_off(i);
  // One buffer descriptor per operand (wave-uniform), the block's offset in the scalar soffset, the lane's part in a 32-bit voffset:
  // no 64-bit address registers.  (Inline assembly as in gemm.hip: behind the DMA builtins hipcc makes every later ds_read wait for
  // vmcnt(0).)  M0 (the LDS destination of a DMA) is written and consumed inside ONE asm statement each time: hipcc reserves m0,
  // does not keep values in it across statements it cannot see into, and rejects it as a clobber ("reserved register ... undefined
  // behaviour" -- ADVICE r4 asked for the clobber); cdna_hip_programming.md 5.7 prescribes exactly this form.
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto make_rsrc = [](const float* base, long bytes) {
    const unsigned long a = (unsigned long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFFL ? bytes : 0x7FFFFFFFL));
    r.w = 0x00020000;
    return r;
  };
  const i32x4 in_rsrc = make_rsrc(in, (long)Cin * H * W * 4);
  const i32x4 w_rsrc = make_rsrc(wpk, (long)nblk * ncot * kF4PanelBytes);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_f4;
  const int w_voff = lane * 16;
  // LDS: half panel of channel g at g * kF4HalfBytes, then the two halo images
  constexpr int kHalo0 = 2 * kF4HalfBytes;
  // A copy for a block past the end of this workgroup's K range is still ISSUED, with every lane out of range (no memory traffic,
  // zeros into the free buffer): the loop body stays one basic block with a fixed copy count per barrier -- with branches around
  // the copies hipcc sinks the transform arithmetic of a pass into the blocks behind them, where no MFMA covers it.
  auto dma_u_piece = [&](int c, int g, int i) {                    // piece i (of 5 per wave) of half panel g of block c -> its buffer
    const int voff = c < nchunks ? w_voff : 0x7FFFFFF0;
    const int cb = chunk0 + min(c, nchunks - 1);
    const int wsoff = __builtin_amdgcn_readfirstlane((cb * ncot + cot) * kF4PanelBytes + g * kF4HalfBytes);
    // branch-free: the waves without a fifth piece copy pieces 16 / 17 a second time (same bytes, same place)
    const int p = min(wave + 4 * i, kF4HalfPieces - 1);
    const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(g * kF4HalfBytes) + (unsigned)p * 1024u);
    const int so = __builtin_amdgcn_readfirstlane(wsoff + p * 1024);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(w_rsrc), "s"(so), "s"(l) : "memory");
  };
  auto dma_h_piece = [&](int c, int hbuf, int i) {                 // piece i (of 6 per wave) of the halo of block c -> halo buffer `hbuf`
    const bool live = c < nchunks;
    const int cb = chunk0 + min(c, nchunks - 1);
    const int hsoff = __builtin_amdgcn_readfirstlane(cb * (int)(plane * 4));
    const int p = min(wave + 4 * i, kF4HaloPieces - 1);
    const int ho = live ? h_off[i] : 0x7FFFFFF0;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kHalo0 + hbuf * kF4HaloBytes) + (unsigned)p * 1024u);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(ho), "s"(in_rsrc), "s"(hsoff), "s"(l) : "memory");
  };
  auto dma_u = [&](int c, int g) {
#pragma unroll
    for (int i = 0; i < 5; ++i) dma_u_piece(c, g, i);
  };
  auto dma_h = [&](int c, int hbuf) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_h_piece(c, hbuf, i);
  };
  // wait until at most `left` of this wave's copies are in flight (they complete in issue order), then the workgroup barrier; the
  // "memory" clobber keeps hipcc from moving LDS accesses across it (the copies are invisible to its own wait-count pass)
#define MNC_F4_SYNC(left) asm volatile("s_waitcnt vmcnt(" #left ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // (ablations 256 / 512 -- no weight copies / no halo copies -- run with 16, no waits: the counts above assume every copy is issued)

  f32x4 acc[36];
#pragma unroll
  for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The accumulators stay in architectural registers: two 256-thread workgroups per CU leave a wave 256 registers, and once a
  // function touches AGPRs hipcc splits that budget 128 / 128 (gemm_x3.hip).
  auto pin_acc = [&]() {
#pragma unroll
    for (int p = 0; p < 36; ++p) asm volatile("" : "+v"(acc[p]));
  };

  // lane-constant LDS byte offsets
  typedef __attribute__((address_space(3))) const char* lds_cp;   // (explicit LDS address space: a volatile access through a generic pointer is a flat_load)
  const lds_cp lds = (lds_cp)(__attribute__((address_space(3))) char*)s_f4;
  const int f0 = (t >> 3) & 1, f1 = ((t + 1) >> 3) & 1;
  // Lane group k multiplies channel 2k + g of the block in pass g; a window read is ONE ds_read_b64 of the pixel's channel pair
  // (2k, 2k + 1) -- the register pair the packed transform works on (half = channel = pass).  16-byte half (k >> 1) of the pixel,
  // stored at half position (k >> 1) ^ f (f = the pixel's swap bit): each 32-lane group of a ds_read_b64 covers the 64 banks once.
  const int hb = kHalo0 + 4 * tg * kF4RowBytes + (k & 1) * 8;
  const int base0 = hb + t * 32 + ((k >> 1) ^ f0) * 16;          // window columns 0..3: slot 17 c + t
  const int base1 = hb + (t + 1) * 32 + ((k >> 1) ^ f1) * 16;    // window columns 4, 5: slot 17 (c - 4) + t + 1
  const int ub = (hp * 64 + lane) * (kF4LanePitch * 4);

  // ---- the loop.  Block s = 8 input channels (lane group k: channels 2k, 2k + 1) = six steps of twelve MFMAs; step i multiplies
  // row i / 2 (of the wave's three transform rows 3 hp .. 3 hp + 2) of channel 2k + (i & 1): six transform columns x the two
  // 16-channel groups, from half i & 1 of that row's six register pairs and twelve of the lane's weights (order of use n = 12 (i %
  // 3) + 2 j + cg in half panel i / 3).  A register pair holds the two channels (2k, 2k + 1) of one position: one ds_read_b64 fills
  // it and both transform dimensions run on it as it is (v_pk_*: both channels per instruction).  WHILE block s is multiplied, the
  // wave builds the operand of block s + 1: step i = five reads of window column i (the five window rows the wave's transform rows
  // use) -> the twelve MFMAs -> one run of VALU: the first dimension of the column just read (6 instructions: three of the six
  // rows of B^T d) and, in steps 0 and 2, the second dimension of the CURRENT operand's next row (12, in place; row 0 was done
  // behind the last MFMAs of the block before).  A row of the current operand dies after its two steps, where two columns of the
  // next one have been added: the operand registers stay at 36-48.  72 VALU instructions and 30 window reads per 72 MFMAs.
  // (The kernel before this one read the two channels with separate ds_read_b32 into the halves of a pair of window COLUMNS: the
  // same arithmetic, twice the read instructions -- and an LDS read costs the loop by the instruction: kernel_bench convwino4,
  // MNC_WINO_F4 = 2 with operands that cost no instruction: 21 % of the loop were the 60 window reads, 15 % the 18 weight reads.)
  // Buffers: half panel h (steps 3h .. 3h + 2) is refilled (block s + 1) while the other half is in use; halo s + 1 is read
  // during block s, its buffer refilled (block s + 3) from the start of block s + 1 -- more than a block to land.
  // Operand registers: x[6 ii + c] = row ii (of the wave's three) of B^T d, window column c; after the second dimension of row ii:
  // transform column c.
  f32x2 va[18], vb[18];
  const F4K K = f4_consts();
  f32x2 abl_pair = {1.f, 2.f};                       // ablations 2 / 8 (tuning builds): operands that cost no instruction
  f32x4 abl_quad = {1.f, 2.f, 3.f, 4.f};
  if (ABL & 10) asm volatile("" : "+v"(abl_pair), "+v"(abl_quad));
  auto run = [&](auto hp_tag) {
    constexpr int HP = decltype(hp_tag)::value;
    auto read_col = [&](int hbuf, int c, f32x2 (&raw)[6]) {
      const lds_cp sh = lds + hbuf * kF4HaloBytes + (c < 4 ? base0 + c * 544 : base1 + (c - 4) * 544);
#pragma unroll
      for (int r = HP; r < 5 + HP; ++r) {            // (window row 5 is not used by transform rows 0..2, row 0 not by rows 3..5)
        if (ABL & 2) raw[r] = abl_pair;
        // (volatile: hipcc otherwise merges neighbouring reads into ds_read2_b64 -- half rate, 32-dword banking)
        else raw[r] = *(__attribute__((address_space(3))) const volatile f32x2*)(sh + r * kF4RowBytes);
      }
    };
    auto first_dim = [&](int c, const f32x2 (&raw)[6], f32x2 (&x)[18]) {
      if (ABL & 4) { x[c] = raw[1]; x[6 + c] = raw[2]; x[12 + c] = raw[3]; }
      else f4_bt_col<HP>(K, raw, x[c], x[6 + c], x[12 + c]);
    };
    auto second_dim = [&](f32x2 (&x)[18], int ii) {
      if (!(ABL & 4)) f4_bt2(K, x[6 * ii], x[6 * ii + 1], x[6 * ii + 2], x[6 * ii + 3], x[6 * ii + 4], x[6 * ii + 5]);
    };
    // the lane's twelve weights of step i: one ds_read_b128 per two transform columns, [column][channel group]
    auto load_u = [&](int i, f32x4 (&uq)[3]) {
      const lds_cp su = lds + (i / 3) * kF4HalfBytes + ub;
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) {
        if (ABL & 8) uq[pp] = abl_quad;
        else uq[pp] = *(__attribute__((address_space(3))) const f32x4*)(su + ((i % 3) * 3 + pp) * 16);
      }
    };
    auto mfma_step = [&](int i, const f32x4 (&uq)[3], const f32x2 (&x)[18]) {
      const int ii = i >> 1, g = i & 1;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f32x2 pr = x[6 * ii + j];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int a = (ii * 6 + j) * 2 + c2;
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(uq[j >> 1][(j & 1) * 2 + c2], g ? pr.y : pr.x, acc[a], 0, 0, 0);
        }
      }
    };
    // The copies a block owes -- first half (h = 0): the halo of block s + 2 (six pieces per wave), then half panel 1 of block s
    // (five); second half: half panel 0 of block s + 1 (five) -- are issued one or two at a time in front of and behind the MFMA runs
    // instead of in a burst behind the barrier: a copy costs its wave 60-180 issue cycles, paid where the other wave of the SIMD
    // has MFMAs to issue.  Both barriers of a block wait for everything in flight.
    auto copies = [&](int g, int slot, int s) {      // slot 0..5: in front of / behind the MFMAs of step slot / 2
      if (ABL & 1) return;
      auto cu = [&](int c, int gg, int i) { if (!(ABL & 256)) dma_u_piece(c, gg, i); };
      auto ch = [&](int c, int hbuf, int i) { if (!(ABL & 512)) dma_h_piece(c, hbuf, i); };
      if (g == 0) {
        if (slot < 3) { ch(s + 2, s & 1, 2 * slot); ch(s + 2, s & 1, 2 * slot + 1); }
        else if (slot < 5) { cu(s, 1, 2 * slot - 6); cu(s, 1, 2 * slot - 5); }
        else cu(s, 1, 4);
      } else if (slot < 5) {
        cu(s + 1, 0, slot);
      }
    };
    // LDS answers in order: a step's MFMAs must not have the window reads in front of the weights they wait for -- the weights
    // of row m + 1 are requested at the end of step m (those of row 0 first thing in the pass: the half panel is only known to
    // have landed behind the barrier), the window reads behind them, and nobody waits for the window reads before the MFMAs are
    // issued.
    auto half = [&](int h, f32x2 (&cur)[18], int hbuf_next, f32x2 (&nxt)[18], int s) {
      f32x4 uq[3];
      load_u(3 * h, uq);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int i = 3 * h + m;
        f32x2 raw[6];
        read_col(hbuf_next, i, raw);
        copies(h, 2 * m, s);
        mfma_step(i, uq, cur);
        // The step's MFMAs back to back, then the transform arithmetic in one run: every change between the two costs about ten
        // cycles.  Nothing moves across a step either: left alone hipcc sinks the arithmetic behind the copies at the end of the half.
        if (!(ABL & 128)) __builtin_amdgcn_sched_barrier(0);
        if (m < 2) load_u(i + 1, uq);
        if (i == 0 || i == 2) second_dim(cur, (i >> 1) + 1);
        first_dim(i, raw, nxt);
        if (i == 5) second_dim(nxt, 0);
        copies(h, 2 * m + 1, s);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    dma_u(0, 0);
    dma_h(0, 0);
    dma_h(1, 1);
    // half panel 0 and halo 0 have landed; halo 1 (this wave's last six copies; they complete in issue order) may still be in
    // flight under the prologue transform -- the barrier behind it waits for everything (round 5)
    if (ABL & 768) { MNC_F4_SYNC(0); } else { MNC_F4_SYNC(6); }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      f32x2 raw[6];
      read_col(0, c, raw);
      first_dim(c, raw, va);
    }
    second_dim(va, 0);
    // every wave has read halo 0 before anybody's first copies (block 0 refills that buffer with the halo of block 2 from its first
    // step on; inside the loop the barrier at the end of block s - 1 stands between the reads of halo s and its refill)
    if (!(ABL & 16)) MNC_F4_SYNC(0);
    // (the loop body names its operand arrays: two blocks per trip, the odd one peeled when the count is odd)
    auto block = [&](f32x2 (&cur)[18], f32x2 (&nxt)[18], int s) {
      half(0, cur, (s + 1) & 1, nxt, s);
      pin_acc();
      // middle of block s: every wave is done with half panel 0; half panel 1 of this block has landed (and the halo of block s + 2)
      if (!(ABL & 16)) MNC_F4_SYNC(0);
      half(1, cur, (s + 1) & 1, nxt, s);
      pin_acc();
      // end of block s: every wave is done with half panel 1 and halo s + 1; half panel 0 of block s + 1 has landed
      if (!(ABL & 16)) MNC_F4_SYNC(0);
    };
    int s = 0;
    for (; s + 2 <= nchunks; s += 2) {
      block(va, vb, s);
      block(vb, va, s + 1);
    }
    if (s < nchunks) block(va, vb, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the dead copies behind the last block write the LDS too: none may outlive the wave)
  };
  if (nchunks > 0) {
    if (hp) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
  }
#undef MNC_F4_SYNC

  // ---- epilogue: Y = A^T M A per (output channel, tile).  The wave holds rows 3 hp .. 3 hp + 2 of M for both channel groups: it
  // applies A along its rows (z = M A, 3 x 4 per channel), then the part of A^T that its rows feed -- hp = 0: (m0 + p, p, q) with
  // p = z1 + z2, q = z1 - z2; hp = 1: (r, s, z5) with r = z3 + z4, s = z3 - z4 -- keeps the triple of channel group hp, hands the
  // other group's to its partner (wave ^ 1: same tile row, same lanes) through LDS and finishes
  //   Y0 = (m0 + p) + r, Y1 = 2 s + q, Y2 = 4 r + p, Y3 = (8 s + q) + z5                    (the operations of the one-wave form)
  // for channel group cg = hp.  lane: tile t of tile row tg, channels cbase .. cbase + 3.
  const int cg = hp;
  const int oy = h0 + 4 * tg, ox = w0 + 4 * t;
  const int cbase = cot * 32 + cg * 16 + 4 * k;
  f32x2 tri[2][3][4][2];                             // [channel group][kind][output column][channel pair]
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x2 z[3][4];
#pragma unroll
      for (int ii = 0; ii < 3; ++ii) {
        f32x2 m[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f32x4 v = acc[(ii * 6 + j) * 2 + c2];
          m[j] = e ? f32x2{v.z, v.w} : f32x2{v.x, v.y};
        }
        f4_at2(m[0], m[1], m[2], m[3], m[4], m[5], z[ii][0], z[ii][1], z[ii][2], z[ii][3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (hp == 0) {
          const f32x2 pq = z[1][j] + z[2][j];
          tri[c2][0][j][e] = z[0][j] + pq;
          tri[c2][1][j][e] = pq;
          tri[c2][2][j][e] = z[1][j] - z[2][j];
        } else {
          tri[c2][0][j][e] = z[0][j] + z[1][j];
          tri[c2][1][j][e] = z[0][j] - z[1][j];
          tri[c2][2][j][e] = z[2][j];
        }
      }
    }
  // exchange: 12 KB per wave, slot (kind, column) x 64 lanes x 16 bytes (lane-contiguous: conflict-free both ways)
  typedef __attribute__((address_space(3))) char* lds_p;
  __syncthreads();                                   // (all waves' copies -- the dead ones behind the last block too -- have landed)
  {
    const lds_p mine = (lds_p)(__attribute__((address_space(3))) char*)s_f4 + wave * 12288 + lane * 16;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 a0 = hp ? tri[0][kd][j][0] : tri[1][kd][j][0], a1 = hp ? tri[0][kd][j][1] : tri[1][kd][j][1];
        *(__attribute__((address_space(3))) f32x4*)(mine + (kd * 4 + j) * 1024) = f32x4{a0.x, a0.y, a1.x, a1.y};
      }
  }
  __syncthreads();
  f32x2 y[4][4][2];                                  // [row][column][channel pair]
  {
    const lds_p theirs = (lds_p)(__attribute__((address_space(3))) char*)s_f4 + (wave ^ 1) * 12288 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 got[3][2];
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const f32x4 v = *(__attribute__((address_space(3))) const f32x4*)(theirs + (kd * 4 + j) * 1024);
        got[kd][0] = f32x2{v.x, v.y};
        got[kd][1] = f32x2{v.z, v.w};
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x2 m0p = hp ? got[0][e] : tri[0][0][j][e], pq = hp ? got[1][e] : tri[0][1][j][e], q = hp ? got[2][e] : tri[0][2][j][e];
        const f32x2 r = hp ? tri[1][0][j][e] : got[0][e], sd = hp ? tri[1][1][j][e] : got[1][e], z5 = hp ? tri[1][2][j][e] : got[2][e];
        y[0][j][e] = m0p + r;
        y[1][j][e] = fma2(f32x2{2.f, 2.f}, sd, q);
        y[2][j][e] = fma2(f32x2{4.f, 4.f}, r, pq);
        y[3][j][e] = fma2(f32x2{8.f, 8.f}, sd, q) + z5;
      }
    }
  }
  if ((ABL & 32) && relu != 12345) return;           // ablation: no stores (the condition keeps the transform arithmetic alive)
  bool fin = ksplit == 1;
  if (!fin && tickets) {
    // K ranges finished inside the launch (mnc_internal.h, slab_last_arriver): the output transform is linear, so a range's slab
    // is its partial 4x4 tiles as they sit in registers -- [wave][tile position (i, j)][lane] x 16 bytes (the lane's four channels),
    // 64 KB per workgroup, a kilobyte per wave instruction, no trip through the LDS transposition.  The last arriver adds the
    // slabs in range order starting from range 0 -- wino4_section_reduce_kernel's order: the same bits -- with the next range's
    // sixteen loads in flight behind the current range's additions, then finishes the tile as an unsplit workgroup would.
    // (only the multi-range section of a launch has slabs: either the whole launch -- uniform cut -- or its tail)
    const __amdgpu_buffer_rsrc_t rs = slab_rsrc(part + (size_t)tile * ksplit * 16384);
    const int lane_off = (wave * 16 * 64 + lane) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        slab_store(rs, lane_off + (i * 4 + j) * 1024, split * 65536, f32x4{y[i][j][0].x, y[i][j][0].y, y[i][j][1].x, y[i][j][1].y});
    if (!slab_last_arriver(tickets + tile, ksplit, reinterpret_cast<volatile unsigned*>(s_f4))) return;
    f32x4 cur[16], nxt[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) cur[q] = slab_load(rs, lane_off + q * 1024, 0);
    f32x4 sum[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) sum[q] = cur[q];
    for (int kq = 1; kq < ksplit; ++kq) {
#pragma unroll
      for (int q = 0; q < 16; ++q) nxt[q] = slab_load(rs, lane_off + q * 1024, kq * 65536);
      if (kq > 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sum[q] += cur[q];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
    }
    if (ksplit > 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) sum[q] += cur[q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = sum[i * 4 + j];
        y[i][j][0] = f32x2{v.x, v.y};
        y[i][j][1] = f32x2{v.z, v.w};
      }
    fin = true;
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fin) bv = *reinterpret_cast<const float4*>(bias + cbase);
  float* dst = fin ? out : part + (long)split * Cout * H * W;
  const long cplane = (long)(cbase >> 3);
  const int chalf = ((cbase >> 2) & 1) * 4;
  if (!pool) {
    // Full-resolution output (or a K range's partial outputs).  A lane holds 4 channels of a 4x4 pixel tile -- 16 bytes of every
    // second 32-byte pixel segment, 128 bytes from its neighbour tile's: stored from there a wave instruction touches 32 cache
    // lines with 32 bytes each (kernel_bench convwino4: the stores cost 0.13 of the trunk's 1.75 ms).  The wave's outputs are 8 runs
    // of 2 KB in memory -- (8-channel block, pixel row) x 64 pixels x 32 B -- so they go through the wave's own 16 KB of LDS (the
    // loop's buffers are free: every copy has landed, every wave is past the barrier): written as 16-byte chunks in memory order,
    // chunk C = 8 t + 2 j + half at position C ^ (t & 7) (the eight lanes of a ds_write_b128 group then hit eight different
    // 16-byte bank groups), read back a kilobyte per instruction (lane L: chunk 64 m + L, conflict-free under the same XOR) and
    // stored as eight whole cache lines.
    __syncthreads();                                 // (every wave has read its partner's partial sums: the exchange area is free)
    const lds_p reg = (lds_p)(__attribute__((address_space(3))) char*)s_f4 + wave * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 o = make_float4(y[i][j][0].x + bv.x, y[i][j][0].y + bv.y, y[i][j][1].x + bv.z, y[i][j][1].y + bv.w);
        if (relu && fin) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        const int C = 8 * t + 2 * j + (k & 1);
        *(__attribute__((address_space(3))) f32x4*)(reg + ((k >> 1) * 4 + i) * 2048 + (C ^ (t & 7)) * 16) = f32x4{o.x, o.y, o.z, o.w};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes are complete (LDS operations of a wave run in order)
    const long cpl0 = (long)((cot * 32 + cg * 16) >> 3);
#pragma unroll
    for (int sg = 0; sg < 8; ++sg)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int C = 64 * m + lane;
        const f32x4 v = *(__attribute__((address_space(3))) const volatile f32x4*)(reg + sg * 2048 + (C ^ ((C >> 3) & 7)) * 16);
        const int yy = oy + (sg & 3), xx = w0 + (C >> 1);
        if (yy < H && xx < W)
          *reinterpret_cast<f32x4*>(dst + (((cpl0 + (sg >> 2)) * H + yy) * W + xx) * 8 + (C & 1) * 4) = v;
      }
  } else {
    // the following Pooling MAX 2x2 stride 2 (test.prototxt:69-79, ...; Caffe's ceil rule: the last window of an odd-sized map
    // is clipped): tile origins are multiples of 4, so a tile is four whole pooling windows
    const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        const int py = oy + 2 * pi, px = ox + 2 * pj;
        if (py >= H || px >= W) continue;
        float4 best = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
          for (int dj = 0; dj < 2; ++dj) {
            if (py + di < H && px + dj < W) {
              const int i = 2 * pi + di, j = 2 * pj + dj;
              float4 o = make_float4(y[i][j][0].x + bv.x, y[i][j][0].y + bv.y, y[i][j][1].x + bv.z, y[i][j][1].y + bv.w);
              if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
              best.x = fmaxf(best.x, o.x); best.y = fmaxf(best.y, o.y); best.z = fmaxf(best.z, o.z); best.w = fmaxf(best.w, o.w);
            }
          }
        *reinterpret_cast<float4*>(dst + ((cplane * OH + (py >> 1)) * OW + (px >> 1)) * 8 + chalf) = best;
      }
  }
}

// OIHW fp32 [Cout][Cin][3][3] -> [Cin/8][Cout/32][2 (half panel h)][2 (position half hp)][64 (lane)][36]: element (cb, ct, h, hp,
// lane = kk * 16 + i, n = 12 m + 2 j + cg) belongs to step 3 h + m of a block: = (G f G^T)[row 3 hp + (3 h + m) / 2][column j] of
// filter (co = ct * 32 + cg * 16 + i, ci = cb * 8 + 2 * kk + ((3 h + m) & 1)) -- the order in which wave hp multiplies (one
// ds_read_b128 = two transform columns x both channel groups), evaluated in double and rounded once.  Cin * Cout * 36 floats, no
// padding: a lane's 144 bytes are 9 x 16, and an odd multiple of 16 bytes as lane pitch is what keeps ds_read_b128 free of bank
// conflicts.
__global__ void pack_conv3x3_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
  const double G[6][3] = {{0.25, 0.0, 0.0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  const int ncot = Cout >> 5;
  const long items = (long)(Cin >> 3) * ncot * 2 * 2 * 64;        // (cb, ct, h, hp, lane)
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < items; r += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(r & 63), hp = (int)((r >> 6) & 1), h = (int)((r >> 7) & 1);
    const long tt = r >> 8;
    const int ct = (int)(tt % ncot), cb = (int)(tt / ncot);
    const int kk = lane >> 4, i = lane & 15;
    float* dst = out + r * kF4LanePitch;
    for (int m = 0; m < 3; ++m) {
      const int step = 3 * h + m, row = 3 * hp + (step >> 1), ci = cb * 8 + 2 * kk + (step & 1);
      for (int cg = 0; cg < 2; ++cg) {
        const int co = ct * 32 + cg * 16 + i;
        const float* f = w + ((long)co * Cin + ci) * 9;
        double tmp[3];                                            // row `row` of G f
#pragma unroll
        for (int b = 0; b < 3; ++b) tmp[b] = G[row][0] * (double)f[b] + G[row][1] * (double)f[3 + b] + G[row][2] * (double)f[6 + b];
#pragma unroll
        for (int b = 0; b < 6; ++b)                               // (G f) G^T, column b
          dst[12 * m + 2 * b + cg] = (float)(tmp[0] * G[b][0] + tmp[1] * G[b][1] + tmp[2] * G[b][2]);
      }
    }
  }
}

// Finishes the pixel tiles [pix0, pix0 + npix) whose K was cut into `s` ranges: out = act(sum_k part[k] + bias) over the tile's
// 16 x 64 pixels (clipped to the image), all channels; POOL: followed by the Pooling MAX 2x2/2 the unsplit tiles apply in their
// epilogue (ReLU first, then the maximum over the window's in-image pixels).  One thread per 4 channels of a pixel (window).
template <int POOL>
__global__ __launch_bounds__(256) void wino4_section_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                   float* __restrict__ out, int H, int W, int Cout, int s, int relu,
                                                                   int pix0, int npix, int tiles_x) {
  constexpr int rows = POOL ? kF4Rows / 2 : kF4Rows, cols = POOL ? kF4Cols / 2 : kF4Cols;
  const int per_tile = (Cout >> 3) * rows * cols * 2;
  const long plane = (long)Cout * H * W;
  const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)npix * per_tile; i += (long)gridDim.x * blockDim.x) {
    const int tl = (int)(i / per_tile);
    int r = (int)(i - (long)tl * per_tile);
    const int half = r & 1; r >>= 1;
    const int col = r % cols; r /= cols;
    const int row = r % rows;
    const int cb = r / rows;
    const int pix = pix0 + tl, bx = pix % tiles_x, by = pix / tiles_x;
    const int y0 = by * kF4Rows + (POOL ? 2 * row : row), x0 = bx * kF4Cols + (POOL ? 2 * col : col);
    if (y0 >= H || x0 >= W) continue;
    const float4 b = *reinterpret_cast<const float4*>(bias + cb * 8 + half * 4);
    float4 best = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
    for (int dy = 0; dy < (POOL ? 2 : 1); ++dy)
#pragma unroll
      for (int dx = 0; dx < (POOL ? 2 : 1); ++dx) {
        const int yy = y0 + dy, xx = x0 + dx;
        if (yy >= H || xx >= W) continue;
        const long e = (((long)cb * H + yy) * W + xx) * 8 + half * 4;
        float4 v = *reinterpret_cast<const float4*>(part + e);
        for (int kq = 1; kq < s; ++kq) {
          const float4 q = *reinterpret_cast<const float4*>(part + kq * plane + e);
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

}  // namespace mnc

using namespace mnc;

// Plan of one layer: how many pixel tiles run whole (section A, ksplit_a ranges each) and into how many K ranges the others are cut.
// The chip holds `slots` = 512 workgroups at a time (two per CU: 80 KB of LDS and 4 waves of 256 registers each).  Full rounds run unsplit; the tiles of a last,
// partly filled round are cut into as many K ranges (of >= 4 blocks) as fill that round once.  A layer that does not fill one
// round at all (conv4_x: 160 workgroups, conv5_x / rpn_conv: 48) is cut uniformly.
static void wino4_plan(int pix, int ncot, int blocks, int* pix_a, int* ksplit_a, int* ksplit_b, int fill = 512) {
  const int slots = 512, min_blocks = 4;
  *pix_a = pix; *ksplit_a = 1; *ksplit_b = 1;
  const long wgs = (long)pix * ncot;
  const int smax = blocks / min_blocks > 1 ? blocks / min_blocks : 1;
  if (wgs <= slots) {
    // uniform cut: the range count whose rounds x range length is smallest; a range of `per` blocks costs `per`, every extra range a
    // pass over its partial outputs (~1.5 blocks' worth at these sizes)
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= smax && s <= 8; ++s) {
      const int per = (blocks + s - 1) / s;
      const double cost = (double)((wgs * s + fill - 1) / fill) * per + (s > 1 ? 1.5 * s : 0.0);
      if (cost < best_cost) { best_cost = cost; best = s; }
    }
    *ksplit_a = best;
    return;
  }
  const int full_pix = (int)(wgs / slots) * slots / ncot;        // pixel tiles of the full rounds
  const int rest = (pix - full_pix) * ncot;
  if (rest > 0 && rest <= slots * 3 / 4) {
    int sb = slots / rest;
    if (sb > smax) sb = smax;
    if (sb > 8) sb = 8;
    // (two ranges do not pay: conv2_x, 192 tail workgroups, 113 / 194 us cut in two against 106 / 187 us whole; conv3_x, 96 tail
    // workgroups in four or five ranges, 105 / 181 us against 115 / 211 us whole -- kernel_bench convwino4, MNC_WINO_TAIL)
    if (sb >= 3) { *pix_a = full_pix; *ksplit_b = sb; }
  }
}

static int wino4_impl(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                      int Cout, int relu, int pool) {
  MNC_REQUIRE(ctx && d_in && d_wpk && d_bias && d_out, "mnc_conv3x3_wino4: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_conv3x3_wino4: unsupported shape H=%d W=%d Cin=%d Cout=%d (need Cin%%8==0, Cout%%32==0)", H, W, Cin, Cout);
  // the operands are addressed through buffer descriptors: 32-bit byte offsets into the whole input tensor and the packed weights
  MNC_REQUIRE((double)Cin * H * W * 4.0 < 2147483648.0 && (double)Cin * Cout * 36.0 * 4.0 < 2147483648.0,
              "mnc_conv3x3_wino4: input %d x %dx%d or weights %d x %d beyond 2 GB (32-bit buffer offsets); use mnc_conv3x3_wino", Cin, H, W,
              Cin, Cout);
  const int ncot = Cout >> 5, blocks = Cin >> 3;
  const int tiles_x = cdiv(W, kF4Cols), pix = tiles_x * cdiv(H, kF4Rows);
  int pix_a, ksplit_a, ksplit_b;
  // WINO_FILL: the workgroup slots a layer smaller than one round (conv4_x, conv5_x, rpn_conv) is cut to fill.  512 = the chip, the
  // shortest launch for an image that has the chip to itself (rounds 4-5); 384 = three quarters (round 6): conv4_x in 2 ranges instead
  // of 3, conv5_x / rpn_conv in 4 instead of 6 -- fewer prologues, slabs and reducers, less CU time, and with four images in flight the
  // CUs left free run the other images' kernels: 265.9 -> 271.0 images/s (three runs each; 320: 271.4, 448: 264.7, 256: 266.2), one
  // image at a time 236.4 -> 230.6 (profiles/r06_fc_ranges.txt)
  wino4_plan(pix, ncot, blocks, &pix_a, &ksplit_a, &ksplit_b, tune(ctx, T_WINO_FILL, plan_latency(ctx) ? 512 : 384));
  if (tune_set(ctx, T_CONV_KSPLIT)) {                            // uniform K ranges (tests, A/B)
    const int v = tune(ctx, T_CONV_KSPLIT, 1);
    if (v >= 1 && v <= 8 && v <= blocks) { pix_a = pix; ksplit_a = v; ksplit_b = 1; }
  } else if (tune(ctx, T_WINO_TAIL, 1) == 0) {
    // (WINO_TAIL=0, round 6: no cut of the tail round of a layer larger than one round -- with four images in flight the tail's free
    // CUs are not idle: 271.4 -> 273.6 images/s, two runs each.  NOT the default: uncut, conv3_x runs the matrix pipe 44 % of a solo
    // launch's cycles instead of 51 %, and ">= 50 % MFMA utilisation on conv3_x" is a target of BASELINE.json measured per launch.)
    if (pix_a < pix) { pix_a = pix; ksplit_b = 1; }
  }
  float* part = nullptr;
  const int smax = ksplit_a > ksplit_b ? ksplit_a : ksplit_b;
  // K ranges finished inside the launch by each tile's last arriver (the kernel's epilogue; FC_REDUCE bit 1, on by default; 0: by
  // the separate wino4_section_reduce_kernel -- the same bits).  Only one section of a plan is ever cut; its tiles index slabs and
  // tickets.  Time-neutral on the trunk (kernel_bench convwino4 1.491 vs 1.492 ms, profiles/r05_inlaunch_reduce.txt: a reducer
  // reads <= 6 x 64 KB, two reducers per CU) and ten launches per image fewer.
  const long cut_tiles = ksplit_a > 1 ? (long)pix_a * ncot : (long)(pix - pix_a) * ncot;
  const bool inkernel = smax > 1 && !(ksplit_a > 1 && ksplit_b > 1 && pix_a < pix) && cut_tiles <= kTickets &&
                        (tune(ctx, T_FC_REDUCE, 2) & 2) != 0;
  if (smax > 1) {
    int rc = ensure_scratch(ctx, inkernel ? (size_t)cut_tiles * smax * 65536 : (size_t)smax * Cout * H * W * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  const double flops = 2.0 * H * W * 9.0 * Cin * Cout;           // ALGORITHMIC work of the convolution (direct form)
  const double out_px = pool ? (double)((H + 1) / 2) * ((W + 1) / 2) : (double)H * W;
  const double bytes = 4.0 * ((double)H * W * Cin + out_px * Cout + 9.0 * Cin * Cout);
  LaunchScope ls(ctx, "conv3x3_wino4_mfma", flops, bytes);
  // XCD-aware order where the channel tiles' shared halo dominates the traffic; for the 512-channel layers the transformed weights
  // (Cin x Cout x 38 floats) dominate and every XCD would stream all of them: plain order (as conv_wino.hip measured)
  const bool plain = tune_set(ctx, T_WINO_XCD) ? tune(ctx, T_WINO_XCD, 1) == 0 : Cout > 256;
  auto kern = plain ? conv3x3_wino4_kernel<0> : conv3x3_wino4_kernel<1>;
#ifdef MNC_TUNING
  switch (tune(ctx, T_WINO_F4, 0)) {
#define MNC_F4_ABL(A) case A: kern = conv3x3_wino4_kernel<1, A>; break;
    MNC_F4_ABL(1) MNC_F4_ABL(2) MNC_F4_ABL(4) MNC_F4_ABL(6) MNC_F4_ABL(8) MNC_F4_ABL(10) MNC_F4_ABL(14) MNC_F4_ABL(15) MNC_F4_ABL(16) MNC_F4_ABL(31) MNC_F4_ABL(32) MNC_F4_ABL(63) MNC_F4_ABL(128) MNC_F4_ABL(272) MNC_F4_ABL(528)
#undef MNC_F4_ABL
    default: break;
  }
#endif
  static std::atomic<unsigned long long> attr_set[2] = {{0}, {0}};      // one bit per device: function attributes are per device
  const unsigned long long bit = 1ull << (ctx->device & 63);
#ifdef MNC_TUNING
  MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kF4LdsBytes));
#else
  if (!(attr_set[plain ? 0 : 1].load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kF4LdsBytes));
    attr_set[plain ? 0 : 1].fetch_or(bit, std::memory_order_relaxed);
  }
#endif
  const long nblocks = ((long)pix_a * ksplit_a + (long)(pix - pix_a) * ksplit_b) * ncot;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), kF4LdsBytes, ctx->stream, d_in, d_wpk, d_bias, d_out, part, H, W,
                     Cin, Cout, relu, pool, tiles_x, pix_a, ksplit_a, ksplit_b, inkernel ? ctx->tickets : nullptr);
  auto grid_for = [](long n) { return (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384); };
  auto reduce = [&](int s, int pix0, int npix) {
    const long items = (long)npix * (Cout >> 3) * kF4Rows * kF4Cols * 2 / (pool ? 4 : 1);
    if (pool)
      hipLaunchKernelGGL(wino4_section_reduce_kernel<1>, dim3(grid_for(items)), dim3(256), 0, ctx->stream, part, d_bias, d_out, H, W,
                         Cout, s, relu, pix0, npix, tiles_x);
    else
      hipLaunchKernelGGL(wino4_section_reduce_kernel<0>, dim3(grid_for(items)), dim3(256), 0, ctx->stream, part, d_bias, d_out, H, W,
                         Cout, s, relu, pix0, npix, tiles_x);
  };
  if (!inkernel && ksplit_a > 1 && pix_a > 0) reduce(ksplit_a, 0, pix_a);
  if (!inkernel && ksplit_b > 1 && pix_a < pix) reduce(ksplit_b, pix_a, pix - pix_a);
  return ls.finish("conv3x3_wino4_kernel");
}

extern "C" {

int mnc_pack_conv3x3_wino4(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin) {
  MNC_REQUIRE(ctx && d_oihw && d_packed && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_pack_conv3x3_wino4: bad argument (Cin%%8==0, Cout%%32==0)");
  LaunchScope ls(ctx, "pack_conv3x3_wino4");
  const long items = (long)(Cin >> 3) * (Cout >> 5) * 2 * 2 * 64;
  long g = (items + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_conv3x3_wino4_kernel, dim3((int)g), dim3(256), 0, ctx->stream, d_oihw, d_packed, Cout, Cin);
  return ls.finish("pack_conv3x3_wino4_kernel");
}

int mnc_conv3x3_wino4(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                      int Cout, int relu) {
  return wino4_impl(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, 0);
}

int mnc_conv3x3_wino4_pool(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out_pooled, int H,
                           int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(H >= 2 && W >= 2, "mnc_conv3x3_wino4_pool: map %dx%d too small for MAX 2x2/2", H, W);
  return wino4_impl(ctx, d_in, d_wpk, d_bias, d_out_pooled, H, W, Cin, Cout, relu, 1);
}

}  // extern "C"
