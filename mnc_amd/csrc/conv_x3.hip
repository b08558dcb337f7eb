// 3x3 convolution on the bf16 matrix pipe with fp32-class accuracy ("bf16x3" split precision, x3_split.h) for gfx950
// (models/VGG16/mnc_5stage/test.prototxt:41-412; BASELINE.json configs[2] "bf16 convs via MFMA").
//
// Same implicit GEMM as conv.hip (M = output channels = MFMA A operand, N = pixels = B operand, c8 feature maps in and
// out, fp32 in HBM), but the contraction runs on v_mfma_f32_32x32x16_bf16: three MFMAs (a_lo*b_hi, a_hi*b_lo, a_hi*b_hi)
// cover 16 K-values in 3 x 32 cycles where the fp32 pipe needs 8 x 64.
//
//   * K is walked in 8-channel blocks, as in conv.hip.  One MFMA K-step (16 values) = 8 channels x TWO taps: lanes 0-31
//     (k 0-7) take tap 2s, lanes 32-63 (k 8-15) take tap 2s+1, s = 0..4.  The tenth tap slot has zero weights
//     (1/10 of the MFMAs is padding; the matrix pipe is not what bounds this kernel, LDS bandwidth is).
//   * weights are split ONCE (mnc_pack_conv3x3_bf16x3): [Cin/8][Cout][84 dwords] = 10 tap slots x (hi x8 | lo x8) bf16
//     + 16 B pad -- a workgroup's panel per channel block is one linear copy into LDS, row pitch 84 dwords
//     (84/4 odd -> conflict-free ds_read_b128);
//   * activations stay fp32 in HBM and are split while the halo is staged into LDS: pixel = hi x8 (16 B) | lo x8 (16 B)
//     | 16 B pad, pitch 12 dwords as in conv.hip;
//   * a wave owns PR pixel rows x 32 columns x 32*CT output channels: per K-step it reads 2*(PR + CT) 16-byte fragments
//     for 3*PR*CT MFMAs -- the register tile is what keeps LDS traffic under the matrix pipe's appetite;
//   * staging loads are unconditional (clamped addresses, out-of-image pixels masked to zero) and issued for block c+1
//     before the MFMAs of block c, stored to the other LDS buffer after them: one barrier per block.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kX3Cols = 32;
constexpr int kX3HaloCols = kX3Cols + 2;
constexpr int kX3PixPitch = 12;        // dwords per halo pixel in LDS: hi x8 | lo x8 | pad
constexpr int kX3WPitch = 84;          // bf16x3: dwords per weight row: 10 slots x 8 + 4 pad (global packed layout and LDS)
constexpr int kF16WPitch = 76;         // f16: 9 taps x 16 channels x 2 B = 72 dwords + 4 pad
// compile-time loop: f(std::integral_constant<int, 0>()) ... f(<N - 1>)
template <int I, int N, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    x3_static_for<I + 1, N>(f);
  }
}

// Activation formats ("fmt" below; bit 0 = input, bit 1 = output):
//   fp32 c8   [C/8][H][W][8] float                                            (32 B per pixel and channel block)
//   packed    bf16x3: [C/8][H][W][hi x8 | lo x8] bf16, the x3_split of the fp32 value (32 B: two 2-byte planes per pixel);
//             f16:    [C/8][H][W][8] fp16, round-to-nearest-even of the fp32 value    (16 B)
// Between two MFMA layers the trunk keeps the packed form: the producer's epilogue applies exactly the split / rounding the
// consumer's staging would apply to the fp32 value, so both routes give the same bits and the consumer's staging becomes a
// plain 16-byte copy into LDS (tests/test_gpu_ops.py::test_conv3x3_packed_activations).
constexpr int kFmtInPacked = 1, kFmtOutPacked = 2;

// Tuning ablations of the pipelined loop (never set in a product build): 1 = no global loads, 2 = no LDS stores,
// 4 = no fragment reads (profiles/r02_x3_ablation.txt)
#ifndef MNC_X3_ABL
#define MNC_X3_ABL 0
#endif

// sched_group_barrier pins for one K-step: NM MFMAs in slots of two; the NR fragment reads of the NEXT K-step front-loaded
// (so that the last of them has a slot of MFMAs behind it), NW LDS stores and NG global loads spread over the slots.
template <int I, int S, int NM, int NR, int NW, int NG>
struct X3Pin {
  static __device__ __forceinline__ void run() {
    constexpr int mf = (I == S - 1) ? NM - 2 * (S - 1) : 2;
    constexpr int RS = S > 1 ? S - 1 : 1;                     // slots that carry reads
    constexpr int rper = (NR + RS - 1) / RS;
    constexpr int r = (I * rper >= NR) ? 0 : ((I + 1) * rper > NR ? NR - I * rper : rper);
    constexpr int w = NW * (I + 1) / S - NW * I / S;
    constexpr int g = NG * (I + 1) / S - NG * I / S;
    __builtin_amdgcn_sched_group_barrier(0x008, mf, 0);
    if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
    if constexpr (g > 0) __builtin_amdgcn_sched_group_barrier(0x020, g, 0);
    if constexpr (w > 0) __builtin_amdgcn_sched_group_barrier(0x200, w, 0);
    if constexpr (I + 1 < S) X3Pin<I + 1, S, NM, NR, NW, NG>::run();
  }
};

// F16 != 0: the "f16" math mode (BASELINE configs[4]) -- one product per term on v_mfma_f32_32x32x16_f16, activations and
// weights rounded to fp16 (nearest even).  Its own K layout (round 2; the first version reused the split layout with the lo
// halves empty and spent its time staging them -- 25 bytes/clk/CU of L2 -> LDS traffic for 20 % of the matrix pipe): a block
// is SIXTEEN channels (two c8 planes), one MFMA K-step = 16 channels x ONE tap (lane half kb = channels 8 kb .. 8 kb + 7), nine
// K-steps per block and no padding slot; a pixel's LDS slot holds its 16 channels (32 B + 16 B pad, the same 12-dword pitch),
// a weight row 9 taps x 16 channels (mnc_pack_conv3x3_f16: [ceil(Cin/16)][Cout][76 dwords], missing channels zero).  Per
// 16 channels a workgroup stages 30 KB instead of 54 KB and passes one barrier instead of two.
//
// Schedule (round 2; profiles/r02_pmc_stalls_convx3_conv3_2.txt showed 55 % issue stalls with every K-step's fragment reads
// placed right in front of its MFMAs): the fragments of K-step s+1 are read during the MFMAs of K-step s (two fragment sets),
// the global loads of block c+1 are issued in K-step 0 of block c and written to the other LDS buffer in K-steps 2 and 3 (one
// register set: two K-steps of MFMAs cover the load), the block barrier sits before K-step 4, whose MFMAs cover the reads of
// block c+1's first fragments.  sched_group_barrier pins that interleave.
template <int CT, int PR, int ROWS, int F16 = 0, int FMT = 0>
__global__ __launch_bounds__(64 * ROWS) void conv3x3_x3_kernel(const void* __restrict__ in_, const uint4* __restrict__ wpk,
                                                               const float* __restrict__ bias, void* __restrict__ out_,
                                                               int H, int W, int Cin, int Cout, int relu, int main_tiles,
                                                               int tail_ks, float* __restrict__ part, int tail_first) {
  constexpr bool kInPk = (FMT & kFmtInPacked) != 0, kOutPk = (FMT & kFmtOutPacked) != 0;
  constexpr int NT = 64 * ROWS;
  constexpr int R = ROWS * PR;                               // pixel rows per workgroup
  constexpr int kHaloPix = (R + 2) * kX3HaloCols;
  constexpr int KS = F16 ? 9 : 5;                            // MFMA K-steps per block
  constexpr int kWPitch = F16 ? kF16WPitch : kX3WPitch;
  // staging items per halo pixel (16 bytes of HBM each): bf16x3: two -- fp32: channels 0-3 / 4-7, packed: hi x8 / lo x8;
  // f16: fp32 input: four (plane h = 0 / 1 of the block, channels 0-3 / 4-7), packed: two (plane h)
  constexpr int kIpp = (F16 && !kInPk) ? 4 : 2;
  constexpr int kHaloItems = kHaloPix * kIpp;
  constexpr int kHPer = (kHaloItems + NT - 1) / NT;
  constexpr int NCO = 32 * CT;
  constexpr int kWVec = NCO * (kWPitch / 4);                 // uint4 items per weight panel
  constexpr int kWPer = (kWVec + NT - 1) / NT;
  constexpr int kHaloDw = kHaloPix * kX3PixPitch;
  constexpr int kWDw = NCO * kWPitch;
  extern __shared__ __attribute__((aligned(16))) unsigned s_mem[];   // halo[2] then weights[2]
  unsigned* const s_halo = s_mem;
  unsigned* const s_w = s_mem + 2 * kHaloDw;
  const uint4* const in = reinterpret_cast<const uint4*>(in_);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kb = lane >> 5;
  // Tiles are numbered column tile fastest, then row tile, then channel tile.  Blocks [0, main_tiles) compute one whole tile.
  // The tiles behind them (the last, partly filled round of resident workgroups -- or all tiles of a small map) are cut into
  // tail_ks K ranges, one block each, whose raw partial sums go to `part` ([tail tile][range][row][channel block][col][8])
  // and are finished by x3_tail_reduce_kernel.
  const int nx = (W + kX3Cols - 1) / kX3Cols, ny = (H + R - 1) / R;
  // tail_first (CONVX3_TAIL=2): the tail's K ranges take the low block numbers and start with the launch
  const int n_tail_blocks = (int)gridDim.x - main_tiles;
  int tile = blockIdx.x, split = 0, ksplit = 1;
  if (tail_first) tile = tile < n_tail_blocks ? main_tiles + tile : tile - n_tail_blocks;
  if (tile >= main_tiles) {
    const int q = tile - main_tiles;
    ksplit = tail_ks;
    tile = main_tiles + q / tail_ks;
    split = q - (tile - main_tiles) * tail_ks;
  }
  const int tx = tile % nx, tyz = tile / nx;
  const int w0 = tx * kX3Cols, h0 = (tyz % ny) * R, co0 = (tyz / ny) * NCO;
  const int nplanes = Cin >> 3;                              // c8 planes of the input
  const int nchunks = (F16 ? (nplanes + 1) >> 1 : nplanes) / ksplit;
  const int chunk0 = split * nchunks;

  // ---- staging assignment (fixed per thread): every thread always loads and stores; surplus threads repeat the last item
  int h_dst[kHPer];
  unsigned h_src[kHPer];                                     // byte offset from the block's first plane (< 2^32)
  unsigned h_keep[kHPer];
  unsigned h_second[kHPer];                                  // f16: all ones for items of the block's second plane
  const unsigned plane_bytes = (unsigned)H * W * ((F16 && kInPk) ? 16 : 32);
#pragma unroll
  for (int u = 0; u < kHPer; ++u) {
    const int q = min(tid + u * NT, kHaloItems - 1);
    const int pix = q / kIpp, sub = q - pix * kIpp;
    const int r = pix / kX3HaloCols, c = pix - r * kX3HaloCols;
    const int gh = h0 - 1 + r, gw = w0 - 1 + c;
    const bool inside = gh >= 0 && gh < H && gw >= 0 && gw < W;
    const unsigned gp = (unsigned)(min(max(gh, 0), H - 1) * W + min(max(gw, 0), W - 1));
    if (!F16) {                      // sub: fp32 -> channels 4 sub .. +3 (hi at +2 sub, lo at +4 + 2 sub); packed -> hi / lo
      h_dst[u] = pix * kX3PixPitch + sub * (kInPk ? 4 : 2);
      h_src[u] = (gp * 2 + sub) * 16;
      h_second[u] = 0u;
    } else if (kInPk) {              // sub = plane
      h_dst[u] = pix * kX3PixPitch + sub * 4;
      h_src[u] = gp * 16;
      h_second[u] = sub ? 0xFFFFFFFFu : 0u;
    } else {                         // sub = plane * 2 + channel half
      h_dst[u] = pix * kX3PixPitch + sub * 2;
      h_src[u] = (gp * 2 + (sub & 1)) * 16;
      h_second[u] = (sub >> 1) ? 0xFFFFFFFFu : 0u;
    }
    h_keep[u] = inside ? 0xFFFFFFFFu : 0u;
  }
  int w_idx[kWPer];
#pragma unroll
  for (int u = 0; u < kWPer; ++u) w_idx[u] = min(tid + u * NT, kWVec - 1);
  // bytes of one block in HBM; the block's base is wave-uniform, per-thread offsets are 32-bit
  const long plane = (long)plane_bytes * (F16 ? 2 : 1);

  // (initialised: hipcc keeps arrays that a lambda writes first as allocas -> scratch otherwise)
  struct Regs {
    uint4 h[kHPer];
    uint4 w[kWPer];
  };
  constexpr bool kPipe = CT * PR >= 4;                       // see "Schedule" above; small register tiles keep the round-1 loop
  Regs G, G1;                                                // G1: second register set of the round-1 loop only
#pragma unroll
  for (int u = 0; u < kHPer; ++u) G.h[u] = G1.h[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < kWPer; ++u) G.w[u] = G1.w[u] = make_uint4(0, 0, 0, 0);
  // f16, odd number of c8 planes: the last block has no second plane -- its items read the first plane again (any valid
  // address) and are stored as zeros
  auto second_missing = [&](int c) { return F16 && 2 * (chunk0 + min(c, nchunks - 1)) + 1 >= nplanes; };
  auto load_chunk = [&](int c, Regs& G) {
    const unsigned sec = second_missing(c) ? 0u : plane_bytes;
    c = chunk0 + min(c, nchunks - 1);
    const char* src = reinterpret_cast<const char*>(in) + (long)c * plane;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) G.h[u] = *reinterpret_cast<const uint4*>(src + (h_src[u] + (h_second[u] & sec)));
    const char* wsrc = reinterpret_cast<const char*>(wpk + ((long)c * Cout + co0) * (kWPitch / 4));
#pragma unroll
    for (int u = 0; u < kWPer; ++u) G.w[u] = *reinterpret_cast<const uint4*>(wsrc + (unsigned)(w_idx[u] * 16));
  };
  // live == false: a phantom block behind an odd block count -- its halo is stored as zeros, so it multiplies to nothing
  auto store_halo = [&](int buf, const Regs& G, int c) {      // c: the block held by G
    unsigned* hdst = s_halo + buf * kHaloDw;
    const bool live = c < nchunks;
    const unsigned nosec = second_missing(c) ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) {
      const unsigned keep = live ? (h_keep[u] & ~(h_second[u] & nosec)) : 0u;
      uint4 v = G.h[u];
      if (kInPk) {
        v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
        *reinterpret_cast<uint4*>(hdst + h_dst[u]) = v;
        continue;
      }
      const float4 x = __builtin_bit_cast(float4, v);
      if (F16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 hv = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
        uint2 hi = F16 == 2 ? x3_bf16x4(x) : __builtin_bit_cast(uint2, hv);     // F16 == 2: plain bf16 (one product per term)
        hi.x &= keep; hi.y &= keep;
        *reinterpret_cast<uint2*>(hdst + h_dst[u]) = hi;
        continue;
      }
      uint2 hi, lo;
      x3_split4(x, hi, lo);
      hi.x &= keep; hi.y &= keep;
      lo.x &= keep; lo.y &= keep;
      *reinterpret_cast<uint2*>(hdst + h_dst[u]) = hi;
      *reinterpret_cast<uint2*>(hdst + h_dst[u] + 4) = lo;
    }
  };
  auto store_weights = [&](int buf, const Regs& G) {
    uint4* wdst = reinterpret_cast<uint4*>(s_w + buf * kWDw);
#pragma unroll
    for (int u = 0; u < kWPer; ++u) wdst[w_idx[u]] = G.w[u];
  };

  f32x16 acc[PR][CT];
#pragma unroll
  for (int r = 0; r < PR; ++r)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;

  // per-lane fragment addresses.  bf16x3: K-step s reads tap 2s + kb (slot 9 = zero weights; its pixel read repeats tap 8),
  // hi then lo 16 bytes.  f16: K-step s = tap s, lane half kb = channels 8 kb .. of the block's 16.
  int p_off[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = min(2 * s + kb, 8);
    p_off[s] = ((wave * PR + tap / 3) * kX3HaloCols + j + tap % 3) * kX3PixPitch;
  }
  const int p_f16 = ((wave * PR) * kX3HaloCols + j) * kX3PixPitch + kb * 4;   // + ((tap / 3) * 34 + tap % 3) * 12
  const int w_base = j * kWPitch + kb * (F16 ? 4 : 8);       // + t * 32 * pitch + s * (F16 ? 8 : 16) (+ 4 for lo)

  struct Frags { uint4 bh[PR], bl[PR], ah[CT], al[CT]; };   // f16: bl / al unused
  auto read_frags = [&](int buf, int s, Frags& f) {
    if (MNC_X3_ABL & 4) {                                    // ablation: no fragment reads, opaque register contents
      auto opaque = [](uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
#pragma unroll
      for (int r = 0; r < PR; ++r) { opaque(f.bh[r]); opaque(f.bl[r]); }
#pragma unroll
      for (int t = 0; t < CT; ++t) { opaque(f.ah[t]); opaque(f.al[t]); }
      return;
    }
    const unsigned* sh = s_halo + buf * kHaloDw;
    const unsigned* sw = s_w + buf * kWDw;
    const int po = F16 ? p_f16 + ((s / 3) * kX3HaloCols + s % 3) * kX3PixPitch : p_off[F16 ? 0 : s];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const unsigned* p = sh + po + r * kX3HaloCols * kX3PixPitch;
      f.bh[r] = *reinterpret_cast<const uint4*>(p);
      if (!F16) f.bl[r] = *reinterpret_cast<const uint4*>(p + 4);
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const unsigned* p = sw + w_base + t * 32 * kWPitch + s * (F16 ? 8 : 16);
      f.ah[t] = *reinterpret_cast<const uint4*>(p);
      if (!F16) f.al[t] = *reinterpret_cast<const uint4*>(p + 4);
    }
  };
  auto mfmas = [&](const Frags& f) {
    if (F16) {
#pragma unroll
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t)
          acc[r][t] = F16 == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.ah[t]), x3_as_bf16x8(f.bh[r]), acc[r][t], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_as_f16x8(f.ah[t]), x3_as_f16x8(f.bh[r]), acc[r][t], 0, 0, 0);
      return;
    }
    // term outermost: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int t = 0; t < CT; ++t)
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.al[t]), x3_as_bf16x8(f.bh[r]), acc[r][t], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int t = 0; t < CT; ++t)
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.ah[t]), x3_as_bf16x8(f.bl[r]), acc[r][t], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int t = 0; t < CT; ++t)
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.ah[t]), x3_as_bf16x8(f.bh[r]), acc[r][t], 0, 0, 0);
  };
  // MFMAs have no side effects, so instruction selection may drift them across a region boundary, which breaks the per-region
  // counts; an empty asm that "modifies" the accumulators keeps each K-step's MFMAs on its side (as in gemm_x3.hip)
  auto pin_acc = [&]() {
#pragma unroll
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int t = 0; t < CT; ++t) asm volatile("" : "+a"(acc[r][t]));
  };
  constexpr int kNM = (F16 ? 1 : 3) * PR * CT;
  constexpr int kNR = (F16 ? 1 : 2) * (PR + CT);
  constexpr int kNWH = kHPer * ((kInPk || F16) ? 1 : 2);
  constexpr int kS = kNM >= 2 ? kNM / 2 : 1;
  // block c sits in LDS[buf]; f[P] holds its K-step-0 fragments; on return f[P ^ 1] holds those of block c + 1 (KS is odd).
  // K-step 0: global loads of block c + 1;  KS - 3: halo stores;  KS - 2: weight stores, then the barrier;  KS - 1: first
  // fragments of block c + 1.
  Frags f[2] = {};
  auto step = [&](int c, int buf, auto parity) {
    constexpr int P = decltype(parity)::value;
    x3_static_for<0, KS>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      constexpr int cur = (P + ks) & 1;
      if (ks < KS - 1) read_frags(buf, ks + 1, f[cur ^ 1]);
      else read_frags(buf ^ 1, 0, f[cur ^ 1]);           // K-step 0 of the next block (of the zero-filled phantom at the end)
      if (ks == 0 && !(MNC_X3_ABL & 1)) load_chunk(c + 1, G);
      if (ks == KS - 3 && !(MNC_X3_ABL & 2)) store_halo(buf ^ 1, G, c + 1);
      if (ks == KS - 2 && !(MNC_X3_ABL & 2)) store_weights(buf ^ 1, G);
      mfmas(f[cur]);
      X3Pin<0, kS, kNM, kNR, ks == KS - 3 ? kNWH : (ks == KS - 2 ? kWPer : 0), ks == 0 ? kHPer + kWPer : 0>::run();
      pin_acc();
      if (ks == KS - 2) __syncthreads();
    });
  };
  // Small register tiles (1 or 2 MFMAs per K-step and fragment set; 5-6 waves per SIMD): two K-steps of MFMAs do not cover
  // a global load, so the round-1 loop is kept -- two register sets (the loads for block c+2 are issued while block c is
  // multiplied and block c+1, requested one iteration earlier, is written to LDS), fragments read right before their MFMAs,
  // the other waves of the SIMD cover the LDS latency.
  auto step_simple = [&](int c, int buf, Regs& cur, Regs& nxt) {
    load_chunk(c + 2, nxt);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      read_frags(buf, ks, f[0]);
      mfmas(f[0]);
    }
    store_halo(buf ^ 1, cur, c + 1);
    store_weights(buf ^ 1, cur);
    __syncthreads();
  };

  load_chunk(0, G);
  store_halo(0, G, 0);
  store_weights(0, G);
  if (kPipe) {
    __syncthreads();
    read_frags(0, 0, f[0]);
    for (int c = 0; c < nchunks; c += 2) {
      step(c, 0, std::integral_constant<int, 0>());
      step(c + 1, 1, std::integral_constant<int, 1>());   // for an odd block count: the zero-filled phantom block
    }
  } else {
    load_chunk(1, G);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      step_simple(c, 0, G, G1);
      step_simple(c + 1, 1, G1, G);
    }
  }

  // ---- epilogue: D[row = cout (reg&3)+8*(reg>>2)+4*kb][col = pixel j] ----
  const int ow = w0 + j;
  if (ksplit > 1) {
    float4* pp = reinterpret_cast<float4*>(part) + (long)((tile - main_tiles) * ksplit + split) * (R * NCO * 8);
#pragma unroll
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          pp[(((wave * PR + r) * (NCO / 8) + t * 4 + g) * 32 + j) * 2 + kb] =
              make_float4(acc[r][t][4 * g + 0], acc[r][t][4 * g + 1], acc[r][t][4 * g + 2], acc[r][t][4 * g + 3]);
    return;
  }
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    const int oh = h0 + wave * PR + r;
    if (oh < H && ow < W) {                                  // lanes j and j + 32 (the two channel halves of a pixel) agree
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b = *reinterpret_cast<const float4*>(bias + co0 + t * 32 + g * 8 + kb * 4);
          v[g] = make_float4(acc[r][t][4 * g + 0] + b.x, acc[r][t][4 * g + 1] + b.y, acc[r][t][4 * g + 2] + b.z,
                             acc[r][t][4 * g + 3] + b.w);
          if (relu) { v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f); }
        }
        const long pix0 = ((long)((co0 >> 3) + t * 4) * H + oh) * W + ow;          // + g * H * W
        const long gstride = (long)H * W;
        if (!kOutPk) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out_) + (pix0 + g * gstride) * 8 + kb * 4) = v[g];
        } else if (F16) {
          // 16-byte stores: v_permlane32_swap hands lane j the whole pixel of channel block g and lane j + 32 that of g + 1
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const uint2 A = x3_f16x4(v[g]), B = x3_f16x4(v[g + 1]);
            const auto sx = __builtin_amdgcn_permlane32_swap(A.x, B.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(A.y, B.y, false, false);
            reinterpret_cast<uint4*>(out_)[pix0 + (g + kb) * gstride] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        } else {
          // lane j stores the pixel's hi x8 (its own four channels and lane j + 32's), lane j + 32 the lo x8
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hi, lo;
            x3_split4(v[g], hi, lo);
            const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
            reinterpret_cast<uint4*>(out_)[(pix0 + g * gstride) * 2 + kb] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
      }
    }
  }
}

// OIHW fp32 -> [Cin/8][Cout][21 uint4]: slot t < 9: (hi x8 | lo x8) of w[co][cb*8 .. +8][tap t]; slot 9 and the pad: zero
// f16 != 0: hi = the 8 values rounded to fp16, lo = zero (conv3x3_x3_kernel<.., F16 = 1>)
__global__ void pack_conv_x3_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin, int f16) {
  const long total = (long)(Cin >> 3) * Cout * 11;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int slot = (int)(i % 11);
    const long row = i / 11;
    const int co = (int)(row % Cout), cb = (int)(row / Cout);
    uint4* dst = out + row * (kX3WPitch / 4);
    if (slot == 10) {
      dst[20] = make_uint4(0, 0, 0, 0);
      continue;
    }
    uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
    if (slot < 9) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = w[((long)co * Cin + cb * 8 + e) * 9 + slot];
      if (f16) {
        const f16x8 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3],
                          (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
        hi = __builtin_bit_cast(uint4, hv);
      } else {
        x3_split8_rne(x, hi, lo);
      }
    }
    dst[slot * 2] = hi;
    dst[slot * 2 + 1] = lo;
  }
}

// f16 layout: OIHW fp32 -> [ceil(Cin/16)][Cout][19 uint4]: tap t < 9: uint4 2t, 2t+1 = the 16 channels of the block rounded to
// fp16 (channels past Cin: zero); uint4 18: pad
template <int BF16>      // BF16 = 1: the same layout in bf16 (nearest even) for the plain "bf16" mode
__global__ void pack_conv_f16_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin) {
  const int nblk = (Cin + 15) / 16;
  const long total = (long)nblk * Cout * 19;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int slot = (int)(i % 19);
    const long row = i / 19;
    const int co = (int)(row % Cout), blk = (int)(row / Cout);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (slot < 18) {
      const int tap = slot >> 1, c0 = blk * 16 + (slot & 1) * 8;
      if (BF16) {
        bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = c0 + e < Cin ? (__bf16)w[((long)co * Cin + c0 + e) * 9 + tap] : (__bf16)0.f;
        v = __builtin_bit_cast(uint4, h);
      } else {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = c0 + e < Cin ? (_Float16)w[((long)co * Cin + c0 + e) * 9 + tap] : (_Float16)0.f;
        v = __builtin_bit_cast(uint4, h);
      }
    }
    out[row * 19 + slot] = v;
  }
}

static int x3_grid_for(long total) {
  const long g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// out = act(sum of a tail tile's K ranges + bias), in the requested activation format; one thread per pixel and channel block
template <int F16, bool PACKED>
__global__ __launch_bounds__(256) void x3_tail_reduce_kernel(const float4* __restrict__ part, const float* __restrict__ bias,
                                                             void* __restrict__ out, int H, int W, int R, int NCO, int main_tiles,
                                                             int ntail, int ks, int relu) {
  const int nx = (W + kX3Cols - 1) / kX3Cols, ny = (H + R - 1) / R;
  const int per_tile = R * (NCO / 8) * 32;
  const long total = (long)ntail * per_tile;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tl = (int)(i / per_tile), e = (int)(i - (long)tl * per_tile);
    const int j = e & 31, cb = (e >> 5) % (NCO / 8), row = (e >> 5) / (NCO / 8);
    const int tile = main_tiles + tl;
    const int tx = tile % nx, tyz = tile / nx;
    const int ow = tx * kX3Cols + j, oh = (tyz % ny) * R + row, co = (tyz / ny) * NCO + cb * 8;
    if (oh >= H || ow >= W) continue;
    const float4* p = part + ((long)tl * ks * per_tile + e) * 2;
    float4 a = p[0], c = p[1];
    for (int s = 1; s < ks; ++s) {
      const float4 a1 = p[(long)s * per_tile * 2], c1 = p[(long)s * per_tile * 2 + 1];
      a.x += a1.x; a.y += a1.y; a.z += a1.z; a.w += a1.w;
      c.x += c1.x; c.y += c1.y; c.z += c1.z; c.w += c1.w;
    }
    const float4 b0 = *reinterpret_cast<const float4*>(bias + co), b1 = *reinterpret_cast<const float4*>(bias + co + 4);
    a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
    c.x += b1.x; c.y += b1.y; c.z += b1.z; c.w += b1.w;
    if (relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
    }
    x3_store8<F16, PACKED>(out, ((long)(co >> 3) * H + oh) * W + ow, a, c);
  }
}

static int x3_grid_for(long total);

template <int CT, int PR, int ROWS, int F16, int FMT>
static int launch_x3(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out, int H, int W,
                     int Cin, int Cout, int relu, int force_ks) {
  constexpr int R = ROWS * PR, NCO = 32 * CT;
  constexpr size_t lds = 2 * 4 * ((size_t)(R + 2) * kX3HaloCols * kX3PixPitch + (size_t)32 * CT * (F16 ? kF16WPitch : kX3WPitch));
  static_assert(lds <= 160 * 1024, "conv3x3_x3: LDS budget");
  auto kern = conv3x3_x3_kernel<CT, PR, ROWS, F16, FMT>;
  static std::atomic<unsigned long long> attr_set{0};          // one bit per device: function attributes are per device
  static std::atomic<int> resident{0};                         // workgroups of this kernel the device holds at once
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0, cus = 0;
    MNC_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * ROWS, lds));
    MNC_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    resident.store(per_cu * cus > 0 ? per_cu * cus : 512, std::memory_order_relaxed);
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const int cap = resident.load(std::memory_order_relaxed);
  const int ntiles = cdiv(W, kX3Cols) * cdiv(H, R) * (Cout / NCO);
  const int blocks = F16 ? (Cin / 8 + 1) / 2 : Cin / 8;        // K blocks: 16 channels in the f16 layout, 8 otherwise
  auto best_ks = [&](int limit) {                              // largest of 4, 2 that divides the K blocks and leaves >= 4 per range
    for (int k = 4; k >= 2; k >>= 1)
      if (k <= limit && blocks % k == 0 && blocks / k >= 4) return k;
    return 1;
  };
  // K splits: for a small map (fewer tiles than half the resident workgroups) every tile is split, as in mnc_conv3x3;
  // otherwise only the tiles of the last round when that round would fill at most a quarter of the device -- 608 tiles on 512
  // resident workgroups (conv3_x) otherwise take two rounds, the second on 96 compute units with one wave per SIMD.
  int main_tiles = ntiles, ks = 1;
  if (force_ks > 0) {
    if (force_ks > 1 && blocks % force_ks == 0) main_tiles = 0, ks = force_ks;
  } else if (ntiles * 2 <= cap) {
    ks = best_ks(4);
    if (ks > 1) main_tiles = 0;
  } else {
    const bool tail_on = tune(ctx, T_CONVX3_TAIL, 1) != 0;
    const int tail = ntiles % cap;
    if (tail_on && tail > 0 && tail * 4 <= cap) {
      ks = best_ks(cap / tail);
      if (ks > 1) main_tiles = ntiles - tail;
    }
  }
  const int ntail = ntiles - main_tiles;
  float* part = nullptr;
  if (ntail > 0) {
    int rc = ensure_scratch(ctx, (size_t)ntail * ks * R * NCO * 32 * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  hipLaunchKernelGGL(kern, dim3(main_tiles + ntail * ks), dim3(64 * ROWS), lds, ctx->stream, d_in, (const uint4*)d_wpk, d_bias,
                     d_out, H, W, Cin, Cout, relu, main_tiles, ks, part,
                     main_tiles > 0 && ntail > 0 && tune(ctx, T_CONVX3_TAIL, 1) == 2 ? 1 : 0);
  if (ntail > 0)
    hipLaunchKernelGGL((x3_tail_reduce_kernel<F16, (FMT & kFmtOutPacked) != 0>), dim3(x3_grid_for((long)ntail * R * (NCO / 8) * 32)),
                       dim3(256), 0, ctx->stream, (const float4*)part, d_bias, d_out, H, W, R, NCO, main_tiles, ntail, ks, relu);
  return MNC_OK;
}

// Pooling MAX 2x2/2 (Caffe's ceil output size) on packed activations.  f16: the maximum of the rounded values is the rounded
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
    if (F16) {
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

}  // namespace mnc

using namespace mnc;

static int pack_conv_lowp(mnc_ctx* ctx, const char* name, const float* d_oihw, void* d_packed, int Cout, int Cin, int f16) {
  MNC_REQUIRE(ctx && d_oihw && d_packed && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0, "%s: bad argument", name);
  LaunchScope ls(ctx, name);
  if (f16) {
    auto kern = f16 == 2 ? pack_conv_f16_kernel<1> : pack_conv_f16_kernel<0>;
    hipLaunchKernelGGL(kern, dim3(x3_grid_for((long)((Cin + 15) / 16) * Cout * 19)), dim3(256), 0, ctx->stream, d_oihw,
                       (uint4*)d_packed, Cout, Cin);
    return ls.finish("pack_conv_f16_kernel");
  }
  hipLaunchKernelGGL(pack_conv_x3_kernel, dim3(x3_grid_for((long)(Cin / 8) * Cout * 11)), dim3(256), 0, ctx->stream, d_oihw,
                     (uint4*)d_packed, Cout, Cin, 0);
  return ls.finish("pack_conv_x3_kernel");
}

template <int F16, int FMT>
static int conv3x3_lowp(mnc_ctx* ctx, const char* name, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out,
                        int H, int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(ctx && d_in && d_wpk && d_bias && d_out, "%s: null pointer", name);
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "%s: unsupported shape H=%d W=%d Cin=%d Cout=%d (need Cin%%8==0, Cout%%32==0)", name, H, W, Cin, Cout);
  // (channel tiles CT, pixel rows PR) per wave.  Measured on MI355X (tools/kernel_bench.py convx3, round 1): the 2x2
  // register tile (8 LDS fragment reads per 12 MFMAs, 2 workgroups per CU) wins wherever it still yields >= 512 workgroups
  // (conv1_2 .. conv3_3: 280-335 TF/s fp32-equivalent); on the small maps (conv4_x, conv5_x, rpn_conv) the 1x1 tile with
  // 4x the workgroups and 5 waves per SIMD is faster (175-285 vs 120-250).  4x2 (1 workgroup per CU) never wins.
  int ct = 1, pr = 1;
  if (Cout % 64 == 0 && (long)cdiv(W, kX3Cols) * cdiv(H, 8) * (Cout / 64) >= 512) ct = 2, pr = 2;
  if (tune_set(ctx, T_CONVX3_TILE)) {                        // "CT,PR" override (mnc_ctx_set_tuning: CT * 1000 + PR)
    const int a = tune(ctx, T_CONVX3_TILE, 0) / 1000, b = tune(ctx, T_CONVX3_TILE, 0) % 1000;
    if ((a == 1 || a == 2) && (b == 1 || b == 2) && Cout % (32 * a) == 0 && !(a == 1 && b == 2)) {
      ct = a;
      pr = b;
    }
  }
  int force_ks = 0;                                          // CONV_KSPLIT: 1 = no splits at all, k > 1 = every tile in k ranges
  if (tune_set(ctx, T_CONV_KSPLIT)) {
    const int v = tune(ctx, T_CONV_KSPLIT, 0);
    if (v >= 1 && v <= 8) force_ks = v;
  }
  const double flops = 2.0 * H * W * 9.0 * Cin * Cout;
  const double ib = (FMT & kFmtInPacked) && F16 ? 2.0 : 4.0, ob = (FMT & kFmtOutPacked) && F16 ? 2.0 : 4.0;
  const double bytes = (double)H * W * (Cin * ib + Cout * ob) + 4.0 * 9.0 * Cin * Cout;
  LaunchScope ls(ctx, name, flops, bytes);
  int rc = MNC_OK;
  if (ct == 2 && pr == 2) rc = launch_x3<2, 2, 4, F16, FMT>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, force_ks);
  else if (ct == 2 && pr == 1) rc = launch_x3<2, 1, 4, F16, FMT>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, force_ks);
  else rc = launch_x3<1, 1, 4, F16, FMT>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, force_ks);
  if (rc) return rc;
  return ls.finish("conv3x3_x3_kernel");
}

template <int F16>
static int conv3x3_lowp_fmt(mnc_ctx* ctx, const char* name, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out,
                            int H, int W, int Cin, int Cout, int relu, int in_packed, int out_packed) {
  const int fmt = (in_packed ? kFmtInPacked : 0) | (out_packed ? kFmtOutPacked : 0);
  switch (fmt) {
    case 0: return conv3x3_lowp<F16, 0>(ctx, name, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
    case 1: return conv3x3_lowp<F16, 1>(ctx, name, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
    case 2: return conv3x3_lowp<F16, 2>(ctx, name, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
    default: return conv3x3_lowp<F16, 3>(ctx, name, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
  }
}

template <int F16>
static int maxpool2_packed(mnc_ctx* ctx, const char* name, const void* d_in, void* d_out, int C, int H, int W) {
  MNC_REQUIRE(ctx && d_in && d_out && C > 0 && C % 8 == 0 && H > 0 && W > 0, "%s: bad argument", name);
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const double b = (F16 ? 2.0 : 4.0) * C;
  LaunchScope ls(ctx, name, 0.0, b * ((double)H * W + (double)OH * OW));
  hipLaunchKernelGGL(maxpool2_packed_kernel<F16>, dim3(x3_grid_for((long)(C / 8) * OH * OW)), dim3(256), 0, ctx->stream,
                     (const uint4*)d_in, (uint4*)d_out, C / 8, H, W, OH, OW);
  return ls.finish("maxpool2_packed_kernel");
}

extern "C" {

int mnc_pack_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return pack_conv_lowp(ctx, "pack_conv3x3_bf16x3", d_oihw, d_packed, Cout, Cin, 0);
}

int mnc_pack_conv3x3_f16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return pack_conv_lowp(ctx, "pack_conv3x3_f16", d_oihw, d_packed, Cout, Cin, 1);
}

int mnc_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W,
                       int Cin, int Cout, int relu) {
  return conv3x3_lowp<0, 0>(ctx, "conv3x3_bf16x3", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
}

int mnc_conv3x3_f16(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                    int Cout, int relu) {
  return conv3x3_lowp<1, 0>(ctx, "conv3x3_f16", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
}

// Plain bf16 (round 4; BASELINE configs[2] "bf16 convs via MFMA" as written): the f16 kernel with both operands rounded to bf16
// (weights once at load, activations while the halo is staged), ONE v_mfma_f32_32x32x16_bf16 per term, fp32 accumulation.  fp32 c8
// tensors in and out (no packed 2-byte form: the mode exists to be measured, not to be fast -- 8 mantissa bits put a layer at ~4e-3
// of its range and the network far outside the 1e-3 bar; f16 runs the same pipe at the same rate with 11 bits).
int mnc_pack_conv3x3_bf16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return pack_conv_lowp(ctx, "pack_conv3x3_bf16", d_oihw, d_packed, Cout, Cin, 2);
}

int mnc_conv3x3_bf16(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                     int Cout, int relu) {
  return conv3x3_lowp<2, 0>(ctx, "conv3x3_bf16", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
}

int mnc_conv3x3_bf16x3_pk(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out, int H, int W,
                          int Cin, int Cout, int relu, int in_packed, int out_packed) {
  return conv3x3_lowp_fmt<0>(ctx, "conv3x3_bf16x3", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, in_packed, out_packed);
}

int mnc_conv3x3_f16_pk(mnc_ctx* ctx, const void* d_in, const void* d_wpk, const float* d_bias, void* d_out, int H, int W, int Cin,
                       int Cout, int relu, int in_packed, int out_packed) {
  return conv3x3_lowp_fmt<1>(ctx, "conv3x3_f16", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, in_packed, out_packed);
}

int mnc_maxpool2_c8_bf16x3(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W) {
  return maxpool2_packed<0>(ctx, "maxpool2_c8_bf16x3", d_in, d_out, C, H, W);
}

int mnc_maxpool2_c8_f16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W) {
  return maxpool2_packed<1>(ctx, "maxpool2_c8_f16", d_in, d_out, C, H, W);
}

int mnc_act_pack(mnc_ctx* ctx, const float* d_c8, void* d_packed, size_t n, int f16) {
  MNC_REQUIRE(ctx && d_c8 && d_packed && n > 0 && n % 8 == 0, "mnc_act_pack: bad argument");
  LaunchScope ls(ctx, "act_pack", 0.0, (f16 ? 6.0 : 8.0) * n);
  if (f16 == 2) hipLaunchKernelGGL(pack_act_kernel<2>, dim3(x3_grid_for((long)(n / 8))), dim3(256), 0, ctx->stream, (const float4*)d_c8, d_packed, (long)(n / 8));
  else if (f16) hipLaunchKernelGGL(pack_act_kernel<1>, dim3(x3_grid_for((long)(n / 8))), dim3(256), 0, ctx->stream, (const float4*)d_c8, d_packed, (long)(n / 8));
  else hipLaunchKernelGGL(pack_act_kernel<0>, dim3(x3_grid_for((long)(n / 8))), dim3(256), 0, ctx->stream, (const float4*)d_c8, d_packed, (long)(n / 8));
  return ls.finish("pack_act_kernel");
}

int mnc_act_unpack(mnc_ctx* ctx, const void* d_packed, float* d_c8, size_t n, int f16) {
  MNC_REQUIRE(ctx && d_c8 && d_packed && n > 0 && n % 8 == 0, "mnc_act_unpack: bad argument");
  LaunchScope ls(ctx, "act_unpack", 0.0, (f16 ? 6.0 : 8.0) * n);
  if (f16 == 2) hipLaunchKernelGGL(unpack_act_kernel<2>, dim3(x3_grid_for((long)(n / 4))), dim3(256), 0, ctx->stream, (const unsigned*)d_packed, (float4*)d_c8, (long)(n / 4));
  else if (f16) hipLaunchKernelGGL(unpack_act_kernel<1>, dim3(x3_grid_for((long)(n / 4))), dim3(256), 0, ctx->stream, (const unsigned*)d_packed, (float4*)d_c8, (long)(n / 4));
  else hipLaunchKernelGGL(unpack_act_kernel<0>, dim3(x3_grid_for((long)(n / 4))), dim3(256), 0, ctx->stream, (const unsigned*)d_packed, (float4*)d_c8, (long)(n / 4));
  return ls.finish("unpack_act_kernel");
}

}  // extern "C"
