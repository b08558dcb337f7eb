// InnerProduct on the bf16 matrix pipe with fp32-class accuracy ("bf16x3" split precision) for gfx950.
//
// Every fp32 operand is split into two bf16 terms, x = hi + lo (x3_split.h); a product is evaluated as
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with three v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in
// the fp32 accumulator).  The dropped terms are O(2^-16) relative, i.e. ~1e-5 per product and less on a dot product --
// two orders of magnitude inside the 1e-3 parity bar (tests/test_gpu_ops.py::test_fc_bf16x3), while the matrix pipe
// runs the three bf16 MFMAs 5.3x faster than the eight fp32 MFMAs they replace (16 K-values in 3 x 32 cycles instead of
// 8 x 64).  At M = 300 that moves the FC layers from MFMA-bound to weight-streaming-bound.
//
// Same tiling as gemm.hip (320 rows x 128 columns x one K split per workgroup, 32-deep stages, one barrier per stage):
//   * weights are split ONCE at load (mnc_pack_fc_bf16x3) into [N/128 column tiles][K/32 stages][128][hi x8 | lo x8 per
//     8 values] bf16 -- 4 bytes per value like fp32, and a workgroup's weight panel of one stage is ONE contiguous 16 KB;
//   * activations are fp32 at the interface and split once per call into the same [M][K/8][hi x8 | lo x8] form in the
//     context's scratch arena (a 10 us elementwise pass; splitting them while staging cost more VALU time per stage than
//     the 60 MFMAs it was meant to hide behind, 256 times over);
//   * LDS rows are 4 groups x 32 B + 16 B pad (pitch 36 dwords, conflict-free ds_read_b128); lane (row, kb) of K-step ks
//     reads group 2*ks + kb: hi and lo are two adjacent 16-byte fragments.
#include <atomic>
#include <cstdlib>

#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXBN = 128, kXBK = 32;
constexpr int kXPitch = 36;            // dwords per LDS row: 4 groups x 8 dwords + 4 pad
constexpr int kXBVec = kXBN * 8;       // uint4 items of the weight panel per stage (128 rows x 128 B)
constexpr int kXBPer = kXBVec / 256;   // 4

__device__ __forceinline__ float x3_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 1.0f / (1.0f + expf(-v));
  return v;
}

// kWR = waves along M: 1 -> a wave owns all kMT row tiles x one 32-column tile (2 + 2*kMT fragment reads per 3*kMT MFMAs);
// 2 -> a wave owns kMT/2 row tiles x two column tiles (4 + kMT reads for the same MFMAs: 14 instead of 22 at kMT = 10 --
// LDS fragment bandwidth, not the matrix pipe, is what bounds this kernel).
// ABL != 0: ablation builds for tuning (MNC_FCX3_ABL): 1 = no global loads / LDS stores in the loop, 2 = additionally no
// barrier, 3 = additionally no LDS fragment reads (MFMAs on constant registers).
// F16 != 0: the same kernel on v_mfma_f32_32x32x16_f16 with ONE product per term ("f16" math mode, BASELINE configs[4]):
// operands are fp16 (weights converted once at load, activations once per call, both stage-major), a stage is 64 K-values
// -- the same 128 bytes per row as 32 split bf16 pairs, so staging, LDS layout and pipeline are unchanged; a half-stage
// (what the split kernel calls a K-step) is two 16-value MFMA steps, two MFMAs per accumulator tile instead of three.
template <int kMT, int kWR, int ABL = 0, int F16 = 0>
__global__ __launch_bounds__(256) void fc_x3_kernel(const uint4* __restrict__ Ax, const uint4* __restrict__ Wx,
                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                    float* __restrict__ part, int M, int N, int K, int ldc, int kper,
                                                    int act, int fused, int tn_, int splits_, int tm_, int mstride) {
  constexpr int kBM = 32 * kMT;
  constexpr int kAPer = (kBM * 8 + 255) / 256;          // uint4 items of the (pre-split) A panel per thread and stage
  __shared__ __attribute__((aligned(16))) unsigned sA[2][kBM * kXPitch];
  __shared__ __attribute__((aligned(16))) unsigned sB[2][kXBN * kXPitch];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kb = lane >> 5;
  int bn, split, bmz;
  if (tm_ < 0) {            // row block fastest (see fc_lowp): the workgroups that share a weight panel are neighbours
    int a, b, c;
    xcd_decode(blockIdx.x, -tm_, tn_, splits_, a, b, c);
    bmz = a; bn = b; split = c;
  } else {
    xcd_decode(blockIdx.x, tn_, splits_, tm_, bn, split, bmz);
  }
  const int n0 = bn * kXBN, m0 = bmz * kBM;
  constexpr int kStageK = F16 ? 64 : kXBK;     // K values per stage (128 bytes per row either way)
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const int nstages = (kend - kbeg) / kStageK;
  const int mrows = min(M - m0, kBM);
  const int mtiles = (mrows + 31) >> 5;

  // staging map, the same for both operands: item q -> row q>>3, uint4 q&7 of the row's 128 bytes of this stage.  Both
  // operands arrive pre-split AND stage-major, so a workgroup's panel of one stage is one contiguous run in memory (16 KB of
  // weights, M x 128 B of activations): with row-major [N][K] weights a stage touched 128 rows 100 KB apart -- 128 DRAM
  // pages / TLB entries for 16 KB -- and the 411 MB fc6 matrix streamed at 1.7 TB/s.
  // Every staging load/store below is UNCONDITIONAL (ragged items are clamped onto the last row and simply rewrite it):
  // a load under a branch makes hipcc lose its vmcnt bookkeeping and drain the whole prefetch pipeline with
  // s_waitcnt vmcnt(0) every stage.
  const uint4* a_src[kAPer];
  int a_dst[kAPer];
#pragma unroll
  for (int u = 0; u < kAPer; ++u) {
    const int q = tid + u * 256, r = min(q >> 3, kBM - 1), c = q & 7;
    const int gr = m0 + min(r, mrows - 1);
    a_src[u] = Ax + (((long)(kbeg / kStageK) * mstride + gr) << 3) + c;      // [stage][row of mstride][8]
    a_dst[u] = r * kXPitch + c * 4;
  }
  const uint4* b_src[kXBPer];
  int b_dst[kXBPer];
#pragma unroll
  for (int u = 0; u < kXBPer; ++u) {
    const int q = tid + u * 256, r = q >> 3, c = q & 7;
    b_src[u] = Wx + ((((long)bn * (K / kStageK) + kbeg / kStageK) * kXBN + r) << 3) + c;   // [column tile][stage][128][8]
    b_dst[u] = r * kXPitch + c * 4;
  }
  const long a_step = (long)mstride << 3, b_step = (long)kXBN << 3;   // uint4 per stage (the panel may hold more rows than this call multiplies)
  // Two register sets (R0/R1): the loads of stage s+2 are in flight while stage s is multiplied and stage s+1 -- already
  // in registers -- is written to the free LDS buffer.  One barrier per stage; the global-load latency gets a whole stage
  // to hide.  The main loop has no VALU work besides addressing: splitting the activations here cost ~140 VALU
  // instructions per thread and stage next to 60 MFMAs, and every one of the 256 workgroups re-split the same panel.
  struct Regs { uint4 a[kAPer]; uint4 b[kXBPer]; };
  Regs R0, R1;
#pragma unroll
  for (int u = 0; u < kAPer; ++u) R0.a[u] = R1.a[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < kXBPer; ++u) R0.b[u] = R1.b[u] = make_uint4(0, 0, 0, 0);

  // load_stage / store_stage are branch-free: the stage index is clamped to the last real stage and a phantom stage
  // (index >= nstages, needed when the stage count is odd) is stored as zeros, so it multiplies to nothing.  A uniform
  // branch around the loads would leave the compiler unsure how many loads are outstanding at the join and it then
  // over-waits (s_waitcnt vmcnt(0..12) on the loads it has just issued).
  auto load_stage = [&](int s, Regs& R) {
    const long st = min(s, nstages - 1);
#pragma unroll
    for (int u = 0; u < kAPer; ++u) R.a[u] = a_src[u][st * a_step];
#pragma unroll
    for (int u = 0; u < kXBPer; ++u) R.b[u] = b_src[u][st * b_step];
  };
  auto store_stage = [&](int buf, const Regs& R, bool live) {
    const unsigned keep = live ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int u = 0; u < kAPer; ++u) {
      uint4 v = R.a[u];
      v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
      *reinterpret_cast<uint4*>(&sA[buf][a_dst[u]]) = v;
    }
#pragma unroll
    for (int u = 0; u < kXBPer; ++u) *reinterpret_cast<uint4*>(&sB[buf][b_dst[u]]) = R.b[u];
  };

  constexpr int TR = kMT / kWR;                 // row tiles per wave
  constexpr int TC = kWR;                       // 32-column tiles per wave (4 waves cover 128 columns)
  static_assert(kMT % kWR == 0 && (kWR == 1 || kWR == 2), "wave grid");
  const int wr = kWR == 1 ? 0 : wave >> 1;      // wave's position along M
  const int wc = kWR == 1 ? wave : wave & 1;    // ... and along N
  f32x16 acc[TR][TC];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;

  // split bf16: K-step ks = dwords [ks*16, ks*16+16) of the row: lane half kb takes 8 of them, hi then lo.
  // f16: half-stage ks = the same 16 dwords = two 16-value MFMA steps q of 8 dwords each: lane half kb takes 4 of them.
  constexpr int kKbOff = F16 ? 4 : 8, kTermOff = F16 ? 8 : 4;
  const int a_base = (wr * TR * 32 + j) * kXPitch + kb * kKbOff;      // + t*32*pitch + ks*16 (+ kTermOff)
  const int b_base = (wc * TC * 32 + j) * kXPitch + kb * kKbOff;      // + c*32*pitch + ks*16 (+ kTermOff)
  // Software pipeline (one wave per SIMD here, so nothing hides a stall but the wave's own MFMAs):
  //   * the fragments of a K-step are read from LDS while the MFMAs of the PREVIOUS K-step run (two fragment sets, F0 / F1);
  //   * the global loads of stage s+2 and the LDS writes of stage s+1 are issued during K-step 0 of stage s;
  //   * the stage barrier sits between K-step 0 and K-step 1, so the first fragments of stage s+1 are prefetched during
  //     K-step 1 of stage s and no LDS latency is exposed at the stage boundary;
  //   * sched_group_barrier pins the interleave (2 MFMAs : 1 ds_read : 1 ds_write : 1 global load); left alone hipcc
  //     clusters each class, and the ablation (MNC_FCX3_ABL) showed the three phases simply adding up:
  //     MFMA 112 us + fragment reads 43 us + staging 90 us = 245 us for fc6.
  struct Frags { uint4 ah[TR], al[TR], bh[TC], bl[TC]; };     // f16: ah/bh = MFMA step 0, al/bl = MFMA step 1 of the half-stage
  auto read_frags = [&](int buf, int ks, Frags& f) {
    const unsigned* pa = sA[buf];
    const unsigned* pb = sB[buf];
    if (ABL == 3) {
      uint4 k = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
      asm volatile("" : "+v"(k.x), "+v"(k.y), "+v"(k.z), "+v"(k.w));
#pragma unroll
      for (int c = 0; c < TC; ++c) f.bh[c] = f.bl[c] = k;
#pragma unroll
      for (int t = 0; t < TR; ++t) f.ah[t] = f.al[t] = k;
      return;
    }
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      f.bh[c] = *reinterpret_cast<const uint4*>(pb + b_base + c * 32 * kXPitch + ks * 16);
      f.bl[c] = *reinterpret_cast<const uint4*>(pb + b_base + c * 32 * kXPitch + ks * 16 + kTermOff);
    }
    // NO per-tile branch: all row tiles are always multiplied (rows past M hold clamped copies and are never stored)
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      f.ah[t] = *reinterpret_cast<const uint4*>(pa + a_base + t * 32 * kXPitch + ks * 16);
      f.al[t] = *reinterpret_cast<const uint4*>(pa + a_base + t * 32 * kXPitch + ks * 16 + kTermOff);
    }
  };
  auto mfmas = [&](const Frags& f) {                 // term outermost: consecutive MFMAs never share an accumulator
    if (F16) {
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int c = 0; c < TC; ++c)
          acc[t][c] = F16 == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.ah[t]), x3_as_bf16x8(f.bh[c]), acc[t][c], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_as_f16x8(f.ah[t]), x3_as_f16x8(f.bh[c]), acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int c = 0; c < TC; ++c)
          acc[t][c] = F16 == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.al[t]), x3_as_bf16x8(f.bl[c]), acc[t][c], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_as_f16x8(f.al[t]), x3_as_f16x8(f.bl[c]), acc[t][c], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < TC; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.al[t]), x3_as_bf16x8(f.bh[c]), acc[t][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < TC; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.ah[t]), x3_as_bf16x8(f.bl[c]), acc[t][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < TC; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.ah[t]), x3_as_bf16x8(f.bh[c]), acc[t][c], 0, 0, 0);
  };
  // MFMAs have no side effects, so instruction selection is free to drift them across the barrier, which breaks the
  // per-region counts below; an empty asm that "modifies" the accumulators (they live in AGPRs: no instruction results)
  // keeps each K-step's MFMAs on its side.
  auto pin_acc = [&]() {
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < TC; ++c) asm volatile("" : "+a"(acc[t][c]));
  };
  constexpr int kNFrag = 2 * (TR + TC), kNMfma = (F16 ? 2 : 3) * TR * TC;
  constexpr int kNStage = kAPer + kXBPer;            // global loads (= LDS writes) per thread and stage
  constexpr int kSlots = kNMfma / 2;                 // interleave slots of one K-step: 2 MFMAs each
  // stage s sits in LDS[buf] and its K-step-0 fragments in f0; stage s+1 is in `cur`; stage s+2 is requested into `nxt`
  auto step = [&](int s, int buf, Regs& cur, Regs& nxt, Frags& f0) {
    Frags f1;
    read_frags(buf, 1, f1);
    if (ABL == 0) store_stage(buf ^ 1, cur, s + 1 < nstages);
    if (ABL == 0) load_stage(s + 2, nxt);
    mfmas(f0);
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (ABL != 3 && i < kNFrag) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (ABL == 0 && i < kNStage) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      if (ABL == 0 && i < kNStage) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    pin_acc();
    if (ABL < 2) __syncthreads();
    read_frags(buf ^ 1, 0, f0);                      // K-step 0 of the next stage (of the zero-filled phantom at the end)
    mfmas(f1);
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (ABL != 3 && i < kNFrag) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    pin_acc();
  };

  if (nstages > 0) {
    Frags F0;
    load_stage(0, R0);
    store_stage(0, R0, true);
    load_stage(1, R0);
    __syncthreads();
    read_frags(0, 0, F0);
    for (int s = 0; s < nstages; s += 2) {
      step(s, 0, R0, R1, F0);
      step(s + 1, 1, R1, R0, F0);   // for an odd stage count the last call multiplies the zero-filled phantom stage
    }
  }

#pragma unroll
  for (int c = 0; c < TC; ++c) {
    const int n = n0 + (wc * TC + c) * 32 + j;
    if (n >= N) continue;
    const float bv = fused ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      const int tile = wr * TR + t;
      if (tile < mtiles) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * kb;
          if (m < M) {
            if (fused) out[(long)m * ldc + n] = x3_act(acc[t][c][e] + bv, act);
            else part[((long)split * M + m) * N + n] = acc[t][c][e];
          }
        }
      }
    }
  }
}

// fp32 row-major [rows][K] -> split bf16, stage-major: out[((tile * S + s) * tile_rows + r) * 8 + 2*g + {hi, lo}] with
// row = tile * tile_rows + r, S = K/32 stages, g = 8-value group inside the stage.  Rows >= rows (padding of the last
// tile) are written as zeros.  Weights: tile_rows = 128 (column tiles of the GEMM).  Activations: one tile of `rows` rows.
__global__ void pack_x3_kernel(const float* __restrict__ in, uint4* __restrict__ out, int rows, int K, int tile_rows,
                               int tiles) {
  const int S = K / kXBK;
  const long total = (long)tiles * S * tile_rows * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i & 3);
    long t = i >> 2;
    const int r = (int)(t % tile_rows);
    t /= tile_rows;
    const int st = (int)(t % S);
    const int tile = (int)(t / S);
    const long row = (long)tile * tile_rows + r;
    uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
    if (row < rows) {
      const float4* p = reinterpret_cast<const float4*>(in + row * K + st * kXBK + g * 8);
      const float4 a = p[0], b = p[1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      x3_split8_rne(x, hi, lo);
    }
    out[i * 2] = hi;
    out[i * 2 + 1] = lo;
  }
}

// fp32 row-major [rows][K] -> fp16 (round to nearest even), stage-major with 64-value stages:
// out[((tile * S + s) * tile_rows + r) * 8 + g] = 8 halves of values s*64 + g*8 .. +7 of row tile * tile_rows + r, S = K/64.
template <int BF16>      // BF16 = 1: the same stage-major layout in bf16 (nearest even), the plain "bf16" math mode (round 4)
__global__ void pack_f16_kernel(const float* __restrict__ in, uint4* __restrict__ out, int rows, int K, int tile_rows,
                                int tiles) {
  const int S = K / 64;
  const long total = (long)tiles * S * tile_rows * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i & 7);
    long t = i >> 3;
    const int r = (int)(t % tile_rows);
    t /= tile_rows;
    const int st = (int)(t % S);
    const int tile = (int)(t / S);
    const long row = (long)tile * tile_rows + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < rows) {
      const float4* p = reinterpret_cast<const float4*>(in + row * K + st * 64 + g * 8);
      const float4 a = p[0], b = p[1];
      if (BF16) {
        const uint2 lo = x3_bf16x4(a), hi = x3_bf16x4(b);
        v = make_uint4(lo.x, lo.y, hi.x, hi.y);
      } else {
        f16x8 h = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z,
                   (_Float16)b.w};
        v = __builtin_bit_cast(uint4, h);
      }
    }
    out[i] = v;
  }
}

// ---- fp16 / split-bf16 InnerProduct, 256-column workgroup tiles, operand panels copied by LDS-DMA (round 3) ----------------------------------
// fc_x3_kernel's 320 x 128 tile moves (320 + 128) x 128 B per 64-deep stage for 5.2 MFLOP, and at 1000 RoIs re-streams the activation
// panel per 128-column tile (32 x 100 MB) plus the weights per row block (4 x 411 MB): 4.8 GB per call.  The lever is bytes per
// flop (energy: the pipe is power limited -- 1.75 GHz under this kernel, DESIGN.md section 9 item 4 -- not a full load path): a
// 256-column tile reads the activations half as often.  This kernel: workgroup = 8 waves (two per SIMD) = 32 kMT rows x
// 256 columns, wave (wm, wn) owns kMT/2 row tiles x columns [64 wn, 64 wn + 64) (10 or 8 accumulator tiles = 160 / 128 AGPRs);
// a stage's panels ([32 kMT][128 B] activations + [256][128 B] weights, both already stage-major and contiguous in memory) go
// global -> LDS with global_load_lds_dwordx4 -- no staging registers, which is what lets eight 10-tile waves fit the register
// file -- into the XOR-swizzled unpadded row layout of fc_mfma_dma_kernel (gemm.hip: row r keeps its 16-byte chunk c in slot
// c ^ ((r >> 1) & 7); conflict-free ds_read_b128 fragments), double-buffered: 2 x 72 KB of LDS at kMT = 10.
// F16 = 0: the split-bf16 products (three MFMAs per term, 32-deep stages) on the same panels.
// Same operands, same K order per accumulator and the same epilogue as fc_x3_kernel: results differ from it only by
// the K-split grouping (fewer column tiles -> more splits).
// MNC_LP_ABL (tuning builds only, -DMNC_LP_ABL=bits, wrong results; tools/lp_abl.sh): 1 no copies inside the loop, 2 no fragment
// reads inside the loop, 4 no barrier, 8 the copies inside the loop issued out of range (the instruction without its memory
// traffic), 16 / 32 no weight / activation copies inside the loop -- what each costs (profiles/r06_fc_lowp.txt)
#ifndef MNC_LP_ABL
#define MNC_LP_ABL 0
#endif
#ifndef MNC_LP_PIPE
#define MNC_LP_PIPE 1      // (tuning: 0 = rounds 3-5's single fragment set in the fp16 / bf16 build too)
#endif
template <int kMT, int F16>
__global__ __launch_bounds__(512) void fc_lowp_dma_kernel(const uint4* __restrict__ Ax, const uint4* __restrict__ Wx,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         float* __restrict__ part, int M, int N, int K, int ldc, int kper, int act,
                                                         int fused, int tn_, int splits_, int tm_, int mstride,
                                                         const uint4* __restrict__ Ax1 = nullptr, const uint4* __restrict__ Wx1 = nullptr,
                                                         const float* __restrict__ bias1 = nullptr, float* __restrict__ out1 = nullptr,
                                                         int pair_tn = 0) {
  constexpr int kBM = 32 * kMT, kBN2 = 256;
  constexpr int kRows = kBM + kBN2;                  // operand rows per stage: activations, then weights
  constexpr int kBuf = kRows * 128;                  // bytes per stage buffer
  constexpr int kPieces = kRows / 8;                 // 1 KB DMA pieces per stage (8 rows each)
  constexpr int kPer = kPieces / 8;                  // per wave
  constexpr int TR = kMT / 2;                        // row tiles per wave
  static_assert(kMT % 2 == 0 && kPieces % 8 == 0, "wave grid");
  extern __shared__ __attribute__((aligned(1024))) char s_lp[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int j = lane & 31, kb = lane >> 5;
  int bn, split, bmz;
  if (tm_ < 0) {            // row block fastest: the workgroups that share a weight panel are neighbours on one XCD
    int a, b, c;
    xcd_decode(blockIdx.x, -tm_, tn_, splits_, a, b, c);
    bmz = a; bn = b; split = c;
  } else {
    xcd_decode(blockIdx.x, tn_, splits_, tm_, bn, split, bmz);
  }
  // PAIR (round 6, mnc_fc_lowp_pair: fc6 + fc6_mask, fc7 + fc7_mask): two products of one shape in one launch -- tn_ counts the
  // column tiles of both, pair_tn those of one; twice the tiles fill the chip with half the K ranges (half the partial sums)
  const int which = (pair_tn && bn >= pair_tn) ? 1 : 0;      // (block-uniform)
  bn -= which * pair_tn;
  if (which) { Ax = Ax1; Wx = Wx1; bias = bias1; out = out1; }
  const int n0 = bn * kBN2, m0 = bmz * kBM;
  constexpr int kStageK = F16 ? 64 : kXBK;           // K values per stage: 128 bytes per row in either format
  const int S = K / kStageK;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const int stage0 = kbeg / kStageK, nstages = (MNC_LP_ABL & 64) ? 0 : (kend - kbeg) / kStageK;   // (64: tuning, no loop at all)
  const int mrows = min(M - m0, kBM);
  const int mtiles = (mrows + 31) >> 5;
  const int npanels = (N + 127) >> 7;

  // Piece p = wave + 8 i covers buffer bytes [1024 p, 1024 p + 1024): lane L fills slot 64 p + L = row r = 8 p + (L >> 3), chunk
  // slot L & 7, i.e. it fetches chunk c = (L & 7) ^ ((r >> 1) & 7) of that row -- and (r >> 1) & 7 = 4 (wave & 1) + (L >> 4) for
  // every i, so c is one lane constant.  r = r0 + 64 i with r0 = 8 wave + (L >> 3) < 64: pieces i < kBM / 64 are activation rows
  // (stage stride mstride x 128 B), the others weight rows r0 + 64 k of the two 128-row panels of this column tile (stage stride
  // 128 x 128 B).  Nothing per piece is kept in registers: the addresses are rebuilt from (r0, c) at every stage.
  static_assert(kBM % 64 == 0, "whole pieces per operand");
  constexpr int kPerA = kBM / 64;
  const int r0 = wave * 8 + (lane >> 3);
  const int cch = (lane & 7) ^ ((wave & 1) * 4 + (lane >> 4));
  const int panel0 = min(bn * 2, npanels - 1), panel1 = min(bn * 2 + 1, npanels - 1);     // a panel past N re-reads the last one
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_lp;
  // (inline assembly, as in fc_mfma_dma_kernel: behind the builtin hipcc makes every later ds_read wait for vmcnt(0))
  // The copies go through buffer descriptors rebuilt per stage on the SCALAR unit (base = operand + stage offset, 64-bit scalar
  // adds), the lane's part a 32-bit offset that does not change from stage to stage: no 64-bit lane address arithmetic -- VALU
  // instructions cost an MFMA stream 3-5 cycles each, reduced precision too (tools/probes/valu_under_f16_mfma_probe.hip); the loop
  // had 24 of them per stage and wave.  Same bytes to the same places as global_load_lds_dwordx4 with lane addresses (round 3).
  typedef int i32x4d __attribute__((ext_vector_type(4)));
  auto make_rsrc = [](const uint4* base) {
    const unsigned long a = (unsigned long)base;
    i32x4d r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
    r.z = 0x70000000;                                // (no lane relies on the range check: rows past M are clamped)
    r.w = 0x00020000;
    return r;
  };
  int voff_a[kPerA];
#pragma unroll
  for (int i = 0; i < kPerA; ++i) voff_a[i] = min(r0 + 64 * i, mrows - 1) * 128 + cch * 16;      // rows past M re-read the last valid row; never stored
  const int voff_w0 = r0 * 128 + cch * 16, voff_w1 = (r0 + 64) * 128 + cch * 16;
  // one descriptor per operand panel, built once (round 6; rounds 3-5 rebuilt three per stage: six v_readfirstlane + 64-bit scalar
  // arithmetic per stage and wave); the stage is the instruction's SCALAR offset: st x mstride x 128 bytes into the activations
  // (< 4 GB for M x K below 2e9 values -- fc_lowp checks), st x 16 KB into a weight panel
  const i32x4d ra = make_rsrc(Ax + (((long)stage0 * mstride + m0) << 3));
  const i32x4d rw0 = make_rsrc(Wx + (((long)panel0 * S + stage0) << 10)), rw1 = make_rsrc(Wx + (((long)panel1 * S + stage0) << 10));
  const unsigned a_stage_bytes = (unsigned)mstride * 128u;
  auto dma_piece = [&](int i, int s, int buf_byte) {
    const unsigned st = (unsigned)min(s, nstages - 1);  // past the end: the last stage once more, never multiplied
    const unsigned so_a = st * a_stage_bytes, so_w = st << 14;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf_byte + (unsigned)(wave + 8 * i) * 1024u);
    if (i < kPerA) {
      if ((MNC_LP_ABL & 32) && s >= 2) return;       // tuning: no activation copies inside the loop
      const int vo = ((MNC_LP_ABL & 8) && s >= 2) ? 0x7FFFFFF0 : voff_a[i];
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(ra), "s"(so_a), "s"(l) : "memory");
    } else {
      if ((MNC_LP_ABL & 16) && s >= 2) return;       // tuning: no weight copies inside the loop
      const int k = i - kPerA;
      const i32x4d rw = (k >> 1) ? rw1 : rw0;
      const int vo = ((MNC_LP_ABL & 8) && s >= 2) ? 0x7FFFFFF0 : ((k & 1) ? voff_w1 : voff_w0);
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rw), "s"(so_w), "s"(l) : "memory");
    }
  };
  auto dma_stage = [&](int s, int buf_byte) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) dma_piece(i, s, buf_byte);
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  f32x16 acc[TR][2];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;

  // fp16: MFMA step q (16 K-values, 4 per stage) reads the 16-byte chunk 2 q + kb of a row.  Split bf16: K-step ks (16 K-values, 2
  // per stage) is the groups 2 ks and 2 ks + 1 of 8 values, lane half kb takes group g = 2 ks + kb: chunk 2 g = its hi halves,
  // 2 g + 1 = its lo halves.  Chunk c of row r sits in slot c ^ ((r >> 1) & 7); tile offsets (32 rows = 4 KB) leave
  // (r >> 1) & 7 = (j >> 1) & 7 unchanged.  Byte offsets inside a buffer.
  constexpr int kSteps = F16 ? 4 : 2;
  const int swz = (j >> 1) & 7;
  int coff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) coff[q] = F16 ? ((2 * q + kb) ^ swz) * 16 : ((2 * (2 * (q >> 1) + kb) + (q & 1)) ^ swz) * 16;   // bf16x3: [ks][hi, lo]
  const int a_row = (wm * TR * 32 + j) * 128, b_row = (kBM + wn * 64 + j) * 128;
  struct Frags { uint4 a[TR]; uint4 b[2]; uint4 al[F16 ? 1 : TR]; uint4 bl[F16 ? 1 : 2]; };
  auto read_frags = [&](int buf_byte, int q, Frags& f) {
    const char* base = s_lp + buf_byte + coff[F16 ? q : 2 * q];
    f.b[0] = *reinterpret_cast<const uint4*>(base + b_row);
    f.b[1] = *reinterpret_cast<const uint4*>(base + b_row + 4096);
#pragma unroll
    for (int t = 0; t < TR; ++t) f.a[t] = *reinterpret_cast<const uint4*>(base + a_row + t * 4096);
    if (!F16) {
      const char* lo = s_lp + buf_byte + coff[2 * q + 1];
      f.bl[0] = *reinterpret_cast<const uint4*>(lo + b_row);
      f.bl[1] = *reinterpret_cast<const uint4*>(lo + b_row + 4096);
#pragma unroll
      for (int t = 0; t < TR; ++t) f.al[t] = *reinterpret_cast<const uint4*>(lo + a_row + t * 4096);
    }
  };
  auto mfmas = [&](const Frags& f) {
    if (F16) {
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          acc[t][c] = F16 == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.a[t]), x3_as_bf16x8(f.b[c]), acc[t][c], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_as_f16x8(f.a[t]), x3_as_f16x8(f.b[c]), acc[t][c], 0, 0, 0);
      return;
    }
    // a_lo b_hi + a_hi b_lo + a_hi b_hi, term outermost -- fc_x3_kernel's order
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.al[t]), x3_as_bf16x8(f.b[c]), acc[t][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.a[t]), x3_as_bf16x8(f.bl[c]), acc[t][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_as_bf16x8(f.a[t]), x3_as_bf16x8(f.b[c]), acc[t][c], 0, 0, 0);
  };
  // The accumulators stay in ARCHITECTURAL registers ("+v"): a 512-thread workgroup leaves a wave 256 registers, and as soon as a
  // function touches AGPRs hipcc splits that budget 128 / 128 -- the 160 accumulator registers of kMT = 10 then spill (800
  // v_accvgpr moves and 53 scratch accesses per stage).  MFMA reads and writes VGPR accumulators at the same rate on gfx950.
  auto pin_acc = [&]() {
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(acc[t][c]));
  };

  if (nstages > 0) {
    dma_stage(0, 0);
    dma_stage(1, kBuf);
    dma_wait();
    __syncthreads();
    int cur = 0;                                     // byte offset of the buffer stage s sits in
    if constexpr (F16 != 0 && MNC_LP_PIPE) {
      // fp16 / bf16 (round 6): TWO fragment sets (28 registers each beside the 160 accumulators: 244 of 256), software-pipelined as
      // fc_mfma_dma16_kernel's -- the fragments of K-step q + 1 are read under the MFMAs of step q, step 0 of the next stage behind
      // the stage barrier under the last step.  With one set the seven ds_read_b128 of a step and their wait stood in front of its ten
      // MFMAs in every wave and only the partner wave of the SIMD could cover them (ablation, fc6 + fc6_mask: 15 of 135 us;
      // profiles/r06_fc_lowp.txt).  sched_group_barrier pins one read behind each of a step's first seven MFMAs.
      constexpr int kNM = 2 * TR, kNR = TR + 2;
      auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < kNM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < kNR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      };
      Frags f0, f1;
      read_frags(0, 0, f0);
      for (int s = 0; s < nstages; ++s) {
        read_frags(cur, 1, f1);
        mfmas(f0);
        interleave();
        read_frags(cur, 2, f0);
        mfmas(f1);
        interleave();
        read_frags(cur, 3, f1);
        mfmas(f0);
        interleave();
        pin_acc();
        dma_wait();                                  // stage s + 1 has landed ...
        if (!(MNC_LP_ABL & 4)) __syncthreads();      // ... for every wave, and every wave holds its last fragments of buffer `cur`
        __builtin_amdgcn_sched_barrier(0);
        if (!(MNC_LP_ABL & 1)) dma_stage(s + 2, cur);  // refill the buffer just released: a whole stage to land (one copy behind
        __builtin_amdgcn_sched_barrier(0);           // each of the last MFMAs instead of this run: measured, no difference)
        read_frags(kBuf - cur, 0, f0);               // step 0 of stage s + 1 (behind the last stage: a copy of it, never multiplied)
        mfmas(f1);
        interleave();
        pin_acc();
        cur = kBuf - cur;
      }
    } else {
      // One fragment set (split bf16: 56 registers a set): two waves share each SIMD -- one wave's fragment reads run under its
      // partner's MFMAs.
      Frags f;
      for (int s = 0; s < nstages; ++s) {
#pragma unroll
        for (int q = 0; q < kSteps - 1; ++q) {
          if (!(MNC_LP_ABL & 2) || s == 0) read_frags(cur, q, f);
          mfmas(f);
        }
        if (!(MNC_LP_ABL & 2) || s == 0) read_frags(cur, kSteps - 1, f);
        pin_acc();
        dma_wait();                                  // stage s + 1 has landed ...
        if (!(MNC_LP_ABL & 4)) __syncthreads();      // ... for every wave, and every wave holds its last fragments of buffer `cur`
        __builtin_amdgcn_sched_barrier(0);
        if (!(MNC_LP_ABL & 1)) dma_stage(s + 2, cur);  // refill the buffer just released: a whole stage to land
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f);                                    // the last step
        pin_acc();
        cur = kBuf - cur;
      }
    }
    dma_wait();                                      // the copies issued by the last two stages have landed before the LDS is released
    __syncthreads();
  }

#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int n = n0 + wn * 64 + c * 32 + j;
    if (n >= N) continue;
    const float bv = fused ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      const int tile = wm * TR + t;
      if (tile < mtiles) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * kb;
          if (m < M) {
            if (fused) out[(long)m * ldc + n] = x3_act(acc[t][c][e] + bv, act);
            else part[((long)(which * splits_ + split) * M + m) * N + n] = acc[t][c][e];
          }
        }
      }
    }
  }
}

// The inverse of the stage-major activation forms (round 6: a consumer or the host that needs fp32 rows of a tensor its producer wrote
// in the stage-major form only): fmt 1: [K/64][M][64] fp16 -> the values; fmt 2: [K/32][M][4][hi x8 | lo x8] bf16 -> hi + lo.
// One thread per 8 consecutive K values of a row.
__global__ void unpack_act_sm_kernel(const uint4* __restrict__ sm, float* __restrict__ out, long M, long K, int fmt) {
  const long total = M * (K >> 3);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (K >> 3), k = (i % (K >> 3)) << 3;
    float v[8];
    if (fmt == 1) {
      const f16x8 h = x3_as_f16x8(sm[((k >> 6) * M + r) * 8 + ((k & 63) >> 3)]);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    } else if (fmt == 3) {
      const uint4 u = sm[((k >> 6) * M + r) * 8 + ((k & 63) >> 3)];
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(w[e] << 16);
        v[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
      }
    } else {
      const uint4* p = sm + (((k >> 5) * M + r) * 4 + ((k & 31) >> 3)) * 2;
      const uint4 a = p[0], b = p[1];
      const unsigned wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(wa[e] << 16) + __uint_as_float(wb[e] << 16);
        v[2 * e + 1] = __uint_as_float(wa[e] & 0xFFFF0000u) + __uint_as_float(wb[e] & 0xFFFF0000u);
      }
    }
    float4* dst = reinterpret_cast<float4*>(out + r * K + k);
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

static int f16_pack_launch(mnc_ctx* ctx, const float* d_in, uint4* d_out, int rows, int K, int tile_rows, int tiles, int bf16 = 0) {
  const long total = (long)tiles * (K / 64) * tile_rows * 8;
  long g = (total + 255) / 256;
  if (g > 65536) g = 65536;
  if (bf16) hipLaunchKernelGGL(pack_f16_kernel<1>, dim3((int)g), dim3(256), 0, ctx->stream, d_in, d_out, rows, K, tile_rows, tiles);
  else hipLaunchKernelGGL(pack_f16_kernel<0>, dim3((int)g), dim3(256), 0, ctx->stream, d_in, d_out, rows, K, tile_rows, tiles);
  return MNC_OK;
}

static int x3_pack_launch(mnc_ctx* ctx, const float* d_in, uint4* d_out, int rows, int K, int tile_rows, int tiles) {
  const long total = (long)tiles * (K / kXBK) * tile_rows * 4;
  long g = (total + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(pack_x3_kernel, dim3((int)g), dim3(256), 0, ctx->stream, d_in, d_out, rows, K, tile_rows, tiles);
  return MNC_OK;
}

}  // namespace mnc

using namespace mnc;

// Shared launcher of the split-bf16 (F16 = 0) and fp16 (F16 = 1) InnerProducts.  The activations are multiplied from their
// stage-major 2-byte form [K/stage][mstride rows][128 B]: either `d_pre` -- already in that form, written by the producer of the
// tensor (mnc_roi_warp_sm, mnc_maxpool2_rhwc_sm, mnc_mask_pool_sm, mnc_fc_pack_act) -- or `d_a`, fp32 row-major, converted here
// into the scratch arena (one elementwise pass per call: 10-200 us that the producers' epilogues make unnecessary).
template <int F16>
static int fc_lowp(mnc_ctx* ctx, const char* what, const float* d_a, const uint4* d_pre, int mstride, const void* d_w_packed,
                   const float* d_bias, float* d_out, int M, int N, int K, int ldc, int act, void* d_osm = nullptr, int osm_fmt = 0,
                   long osm_rows = 0, long osm_row0 = 0) {
  constexpr int kStage = F16 ? 64 : kXBK;
  if (M == 0) return MNC_OK;
  // Several row blocks (M > 320: the 1000-RoI ResNet configuration, CFM): every block streams its whole weight panel, so a
  // block costs about (rows + 128) -- measured: the 40-row tail of M = 1000 took 0.19 of the time of the 960 rows before it.
  // 256-row blocks (fc_x3_kernel<8, 2>) in ONE launch when that is cheaper than 320-row blocks plus a tail launch:
  // M = 1000: 4 x (256 + 128) = 1536 against 3 x 448 + 288 = 1632; M = 960 or 2000 stay on 320-row blocks.
  bool rows256 = false;
  if (M > 320 && 2.0 * M * (double)N * K >= 2.0e9) {
    const int tail = M % 320;
    const long cost320 = (long)(M / 320) * 448 + (tail == 0 ? 0 : tail <= 160 ? 288 : 448);
    const long cost256 = (long)cdiv(M, 256) * 384;
    rows256 = cost256 < cost320 && !tune(ctx, T_FC_NO256, 0);
  }
  // full 320-row blocks and a ragged tail of at most 160 rows are two launches, each with its own tile height (see mnc_fc)
  if (!rows256 && M > 320 && M % 320 != 0 && M % 320 <= 160 && 2.0 * M * (double)N * K >= 2.0e9 && !tune(ctx, T_FC_NOTAIL, 0)) {
    const int head = M / 320 * 320;
    int rc = fc_lowp<F16>(ctx, what, d_a, d_pre, mstride, d_w_packed, d_bias, d_out, head, N, K, ldc, act, d_osm, osm_fmt, osm_rows,
                          osm_row0);
    if (rc) return rc;
    return fc_lowp<F16>(ctx, what, d_a ? d_a + (size_t)head * K : nullptr, d_pre ? d_pre + (size_t)head * 8 : nullptr, mstride,
                        d_w_packed, d_bias, d_out + (size_t)head * ldc, M - head, N, K, ldc, act, d_osm, osm_fmt, osm_rows,
                        osm_row0 + head);
  }
  // row tiles per workgroup: 2 (64 rows) for the small GEMMs, else the smallest of {5, 10} that covers M in one block
  const bool small = 2.0 * M * (double)N * K < 2.0e9;
  int mt = small ? 2 : (M <= 160 ? 5 : 10);
  // 320-row blocks stream the weights once but need many K splits to fill the chip; when a split would be shorter than 64 stages
  // (bf16x3; 32 for fp16), 160-row blocks (twice the tiles, half the splits and half the partial-sum traffic) are faster
  // (measured at M = 300, bf16x3: fc7 59 vs 69 us, fc6_maskest 132 vs 151 us, fc6 274 vs 262 us)
  // Round 6, throughput plan: ONE row block runs on the 256-column LDS-DMA kernel from N = 256 on, in fc_lowp_ranges' K ranges, split
  // bf16 included -- fc6_maskest (N = 256, K = 100352): one column tile x 49 ranges of 2048 K values on 49 CUs, each operand read
  // once (111 MB), instead of 2 x 2 tiles x 64 ranges on the 160-row register-staged kernel.  The launch is longer (fp16 37 -> 67 us,
  // split bf16 65 -> 159) and costs a fifth of the CU time: f16 985 -> 991 images/s, mixed 616 -> 620, bf16x3 488 -> 491 (two runs
  // each, profiles/r06_fc_ranges.txt); the latency plan keeps the old choice.
  const bool wide1 = tune(ctx, T_FCX3_WIDE, 1) != 0 && !tune_set(ctx, T_FCX3_TILE) && !plan_latency(ctx) && mt == 10 && M <= 320 && N % 256 == 0 && N >= 256 &&
                     (K / kStage) / fc_lowp_ranges(ctx, 256, K, N / 256) >= (F16 ? 8 : 16);
  if (!wide1 && mt == 10 && (K / kStage) / cdiv(256, cdiv(N, kXBN) * cdiv(M, 320)) < (F16 ? 32 : 64)) mt = 5;
  if (rows256) mt = 8;
  if (tune_set(ctx, T_FCX3_TILE)) {
    const int v = tune(ctx, T_FCX3_TILE, 0);
    if (v == 2 || v == 5 || v == 8 || v == 10) mt = v;
  }
  const int bm = 32 * mt;
  const int stages = K / kStage, tm = cdiv(M, bm);
  // 256- or 320-row blocks, N a multiple of 256: the 256-column LDS-DMA kernel (fc_lowp_dma_kernel) -- half the activation
  // bytes per flop (the reduced-precision pipe is power limited, DESIGN.md section 9 item 4: bytes are energy) -- when its K splits keep at least 8 stages
  // (fc7 at 300 RoIs would get 4: prologue and epilogue of a workgroup then outweigh the traffic saved).  FCX3_WIDE=0: off.
  bool wide = false;
  // (throughput plan, round 6: from N = 256 on -- one column tile per row block; ResNet-50 configuration, fc6_maskest at 1000 RoIs:
  // f16 260.5 -> 264.6 images/s, mixed 130.2 -> 134.2)
  if ((mt == 8 || mt == 10) && N % 256 == 0 && N >= (plan_latency(ctx) ? 512 : 256) && tune(ctx, T_FCX3_WIDE, 1) != 0) {
    const int sp = cdiv(256, (N / 256) * tm);
    wide = stages / (sp > 0 ? sp : 1) >= (F16 ? 8 : 16);
    // split bf16 at one row block (300 RoIs): three MFMAs per term make the operand bytes a smaller share of the work, and the doubled
    // K splits cost in the reduction what the kernel gains (fc6: 213.9 + 10.8 us vs 204.5 + 16.9 us) -- the 128-column kernel stays
    if (!F16 && tm == 1 && !tune_set(ctx, T_FCX3_WIDE)) wide = false;
  }
  if (wide1) wide = true;
  // (the 256-column kernel addresses a stage of the activations by a 32-bit scalar offset: stages x m_stride x 128 bytes)
  if (wide && (double)stages * (double)(d_pre ? mstride : M) * 128.0 >= 4.0e9) wide = false;
  const int bn_w = wide ? 256 : kXBN;
  const int tn = cdiv(N, bn_w);
  int splits = cdiv(mt == 2 ? 512 : 256, tn * tm);
  const int min_stages = F16 ? (mt == 2 ? 1 : 4) : (mt == 2 ? 2 : 8);
  if (splits > stages / min_stages) splits = stages / min_stages;
  if (splits < 1) splits = 1;
  if (tm == 1 && mt != 2) {
    const int full = splits;
    splits = fc_lowp_ranges(ctx, splits, K, tn);      // (round 6: CU time, not launch time -- mnc_internal.h)
    if (splits == 1 && full > 1 && d_osm && (ldc != N || osm_rows != M || osm_row0 != 0)) splits = 2;
  }
  if (tm > 1 && mt != 2)     // several row blocks: split count by cost (mnc_internal.h: choose_splits)
    // (round 6, throughput plan: the K ranges fill HALF the chip here too -- ResNet-50 configuration, 1000 RoIs, four images in
    // flight: f16 249 -> 262 images/s, mixed 126.5 -> 131.6 with 128 slots, 259 with 64; FC_SLOTS overrides)
    splits = choose_splits(tn * tm, stages, min_stages, tune(ctx, T_FC_SLOTS, plan_latency(ctx) ? 256 : 128),
                           (double)bm * bn_w * kStage * 2.0 / (F16 ? 2000.0e3 : 1050.0e3), 4.0 * M * (double)N);
  const int kper = cdiv(stages, splits) * kStage;
  splits = cdiv(K, kper);
  // scratch arena: [split-K partials | the activations in their 2-byte stage-major form (when they arrive as fp32)]
  const size_t part_bytes = splits > 1 ? (((size_t)splits * M * N * 4 + 255) & ~(size_t)255) : 0;
  int rc = ensure_scratch(ctx, part_bytes + (d_pre ? 0 : (size_t)M * K * (F16 ? 2 : 4)));
  if (rc) return rc;
  float* part = splits > 1 ? (float*)ctx->scratch : nullptr;
  const uint4* d_ax = d_pre;
  if (!d_pre) {
    uint4* conv = (uint4*)((char*)ctx->scratch + part_bytes);
    LaunchScope ls(ctx, F16 ? "fc_f16_convert" : "fc_bf16x3_split", 0.0, (F16 ? 6.0 : 8.0) * M * (double)K);
    if (F16) f16_pack_launch(ctx, d_a, conv, M, K, M, 1, F16 == 2);
    else x3_pack_launch(ctx, d_a, conv, M, K, M, 1);
    rc = ls.finish(F16 ? "pack_f16_kernel" : "pack_x3_kernel");
    if (rc) return rc;
    d_ax = conv;
    mstride = M;
  }
  const double flops = 2.0 * M * (double)N * K;
  const double bytes = (F16 ? 2.0 : 4.0) * ((double)N * K + (double)M * K) + 4.0 * (double)M * N;
  // Block order with several row blocks: row block fastest, so the workgroups that multiply the same weight panel are
  // neighbours on one XCD and stream it from L2 together instead of once per row block from HBM (measured, fp16, M = 960:
  // N = 4096, K = 50176: 612 -> 555 us; M = 2000, K = 25088: 302 -> 282 us; with two column tiles (N = 256) it is 3 % slower,
  // so only from 8 column tiles on).  MNC_FC_ORDER=0 / 1 forces the column-tile-fastest / row-block-fastest order.
  const bool rows_fastest = tune_set(ctx, T_FC_ORDER) ? tune(ctx, T_FC_ORDER, 0) == 1 : tn >= 8;
  const int tm_arg = (rows_fastest && tm > 1) ? -tm : tm;
  {
    LaunchScope ls(ctx, F16 == 2 ? (small ? "fc_bf16_small" : "fc_bf16") : F16 ? (small ? "fc_f16_small" : "fc_f16") : (small ? "fc_bf16x3_small" : "fc_bf16x3"), flops, bytes);
#define MNC_X3_LAUNCH(MT, WR, A)                                                                                              \
  hipLaunchKernelGGL((fc_x3_kernel<MT, WR, A, F16>), dim3(tn * splits * tm), dim3(256), 0, ctx->stream, d_ax,                 \
                     (const uint4*)d_w_packed, d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, \
                     tm_arg, mstride)
    if (wide) {
      {
#define MNC_WIDE_LAUNCH(MT)                                                                                                     \
  do {                                                                                                                          \
    constexpr int lds = 2 * (32 * MT + 256) * 128;                                                                              \
    static std::atomic<unsigned long long> attr_set{0};            /* one bit per device */                                    \
    const unsigned long long bit = 1ull << (ctx->device & 63);                                                                  \
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {                                                                    \
      MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_lowp_dma_kernel<MT, F16>),                                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                        \
      attr_set.fetch_or(bit, std::memory_order_relaxed);                                                                        \
    }                                                                                                                           \
    hipLaunchKernelGGL((fc_lowp_dma_kernel<MT, F16>), dim3(tn * splits * tm), dim3(512), lds, ctx->stream, d_ax,                      \
                       (const uint4*)d_w_packed, d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, \
                       tm_arg, mstride);                                                                                        \
  } while (0)
        if (mt == 8) MNC_WIDE_LAUNCH(8);
        else MNC_WIDE_LAUNCH(10);
#undef MNC_WIDE_LAUNCH
      }
    }
    else if (mt == 2) MNC_X3_LAUNCH(2, 2, 0);
    else if (mt == 5) MNC_X3_LAUNCH(5, 1, 0);
    else if (mt == 8) MNC_X3_LAUNCH(8, 2, 0);
    else if constexpr (F16 != 0) {
      MNC_X3_LAUNCH(10, 2, 0);
    } else {
#ifdef MNC_TUNING
      const int abl = tune(ctx, T_FCX3_ABL, 0);                     // ablation builds of the split-bf16 kernel
      if (abl == 1) MNC_X3_LAUNCH(10, 2, 1);
      else if (abl == 2) MNC_X3_LAUNCH(10, 2, 2);
      else if (abl == 3) MNC_X3_LAUNCH(10, 2, 3);
      else
#endif
      MNC_X3_LAUNCH(10, 2, 0);
    }
#undef MNC_X3_LAUNCH
    rc = ls.finish(F16 ? "fc_x3_kernel<f16>" : "fc_x3_kernel");
    if (rc) return rc;
  }
  bool osm_done = false;
  if (splits > 1) {
    LaunchScope ls(ctx, "fc_reduce", 0.0, 4.0 * ((double)splits + 1.0) * M * N);
    osm_done = fc_reduce_launch_sm(ctx->stream, part, d_bias, d_out, M, N, ldc, splits, act, d_osm, osm_fmt, osm_rows, osm_row0);
    rc = ls.finish("fc_reduce_kernel");
    if (rc) return rc;
  }
  if (d_osm && !osm_done) {
    // no reduction pass to write it from (one K split: the GEMM's epilogue stored the rows), or a shape the reduction's vector
    // path does not take: convert the rows just written (row-major, ldc == N required by the entry point in that case)
    MNC_REQUIRE(ldc == N && osm_rows == M && osm_row0 == 0, "%s: the second output needs a K-split reduction or dense rows", what);
    LaunchScope ls(ctx, osm_fmt != 2 ? "fc_f16_convert" : "fc_bf16x3_split", 0.0, (osm_fmt != 2 ? 6.0 : 8.0) * M * (double)N);
    if (osm_fmt != 2) f16_pack_launch(ctx, d_out, (uint4*)d_osm, M, N, M, 1, osm_fmt == 3);
    else x3_pack_launch(ctx, d_out, (uint4*)d_osm, M, N, M, 1);
    return ls.finish("pack kernel");
  }
  return MNC_OK;
}

// Two reduced-precision InnerProducts of one shape (fp16 / plain bf16) as ONE launch of the 256-column LDS-DMA kernel: the box and
// the mask branch of a head stage (fc6 + fc6_mask, fc7 + fc7_mask; test.prototxt:584-627, 652-696).  Twice the column tiles fill
// the chip with half the K ranges: fc6 at 300 RoIs writes 2 x 39 MB of partial sums instead of 2 x 79, fc7 (whose 64 stages are
// too few for the 256-column kernel alone) gets it with 8-stage ranges.  Shapes the paired kernel does not take (one row block of
// 161..320 rows, N a multiple of 256, >= 8 stages per range) run as two fc_lowp calls -- every executor of a graph calls this entry
// point for the same layers, so they keep the same bits either way.
template <int F16>
static int fc_lowp_pair(mnc_ctx* ctx, const char* what, const float* d_a0, const uint4* d_pre0, const float* d_a1, const uint4* d_pre1,
                        int mstride, const void* d_w0, const void* d_w1, const float* d_bias0, const float* d_bias1, float* d_out0,
                        float* d_out1, int M, int N, int K, int ldc, int act, void* d_osm0, void* d_osm1, int osm_fmt) {
  constexpr int kStage = F16 ? 64 : kXBK;
  if (M == 0) return MNC_OK;
  const int stages = K / kStage, tn = N / 256;
  // Several row blocks (round 6, throughput plan only; the ResNet-50 configuration's 1000 RoIs): the pair runs as ONE launch too when
  // fc_lowp would run each product as one launch -- 256-row blocks, or 320-row blocks without a ragged tail of <= 160 rows
  int mt = 10, tm = 1;
  bool multi = false;
  if (M > 320 && !plan_latency(ctx)) {
    const int tail = M % 320;
    const long cost320 = (long)(M / 320) * 448 + (tail == 0 ? 0 : tail <= 160 ? 288 : 448);
    const long cost256 = (long)cdiv(M, 256) * 384;
    const bool rows256 = cost256 < cost320 && !tune(ctx, T_FC_NO256, 0);
    multi = rows256 || tail == 0 || tail > 160 || tune(ctx, T_FC_NOTAIL, 0);
    mt = rows256 ? 8 : 10;
    tm = cdiv(M, 32 * mt);
  }
  bool paired = M > 160 && (M <= 320 || multi) && N % 256 == 0 && N >= 512 && 2.0 * M * (double)N * K >= 2.0e9 && tune(ctx, T_FCX3_WIDE, 1) != 0 &&
                (double)stages * (double)((d_pre0 || d_pre1) ? mstride : M) * 128.0 < 4.0e9 &&
                !tune_set(ctx, T_FCX3_TILE) && tune(ctx, T_FUSE_SMALL, 1) != 0;
  int splits = 1;
  if (paired && tm > 1) {
    splits = choose_splits(2 * tn * tm, stages, F16 ? 4 : 8, tune(ctx, T_FC_SLOTS, 128),
                           (double)(32 * mt) * 256 * kStage * 2.0 / (F16 ? 2000.0e3 : 1050.0e3), 8.0 * M * (double)N);
    paired = stages / splits >= (F16 ? 8 : 16);
    if (splits == 1 && stages >= 2 * (F16 ? 8 : 16) && (d_osm0 || d_osm1) && ldc != N) splits = 2;
  } else if (paired) {
    splits = cdiv(256, 2 * tn);
    if (splits > stages / (F16 ? 4 : 8)) splits = stages / (F16 ? 4 : 8);
    if (splits < 1) splits = 1;
    paired = stages / splits >= (F16 ? 8 : 16);                // (the 256-column kernel's own bar: fc_lowp)
    const int full = splits;
    splits = fc_lowp_ranges(ctx, splits, K, 2 * tn, true);     // (round 6: CU time, not launch time -- mnc_internal.h)
    // (a second output in stage-major form is written by the reduction pass; without one only dense rows can be converted)
    if (splits == 1 && full > 1 && (d_osm0 || d_osm1) && ldc != N) splits = 2;
  }
  if (!paired) {
    int rc = fc_lowp<F16>(ctx, what, d_a0, d_pre0, d_a0 ? M : mstride, d_w0, d_bias0, d_out0, M, N, K, ldc, act, d_osm0,
                          d_osm0 ? osm_fmt : 0, M, 0);
    if (rc) return rc;
    return fc_lowp<F16>(ctx, what, d_a1, d_pre1, d_a1 ? M : mstride, d_w1, d_bias1, d_out1, M, N, K, ldc, act, d_osm1,
                        d_osm1 ? osm_fmt : 0, M, 0);
  }
  const int kper = cdiv(stages, splits) * kStage;
  splits = cdiv(K, kper);
  // scratch arena: [partial sums of both products | activations that arrive as fp32, in their 2-byte stage-major form]
  const size_t part_bytes = splits > 1 ? (((size_t)2 * splits * M * N * 4 + 255) & ~(size_t)255) : 0;
  const size_t conv_bytes = ((size_t)M * K * (F16 ? 2 : 4) + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, part_bytes + (d_pre0 ? 0 : conv_bytes) + (d_pre1 ? 0 : conv_bytes));
  if (rc) return rc;
  float* part = splits > 1 ? (float*)ctx->scratch : nullptr;
  const uint4* ax[2] = {d_pre0, d_pre1};
  int ms[2] = {mstride, mstride};
  {
    char* conv = (char*)ctx->scratch + part_bytes;
    const float* a32[2] = {d_a0, d_a1};
    for (int i = 0; i < 2; ++i) {
      if (ax[i]) continue;
      LaunchScope ls(ctx, F16 ? "fc_f16_convert" : "fc_bf16x3_split", 0.0, (F16 ? 6.0 : 8.0) * M * (double)K);
      if (F16) f16_pack_launch(ctx, a32[i], (uint4*)conv, M, K, M, 1, F16 == 2);
      else x3_pack_launch(ctx, a32[i], (uint4*)conv, M, K, M, 1);
      rc = ls.finish(F16 ? "pack_f16_kernel" : "pack_x3_kernel");
      if (rc) return rc;
      ax[i] = (const uint4*)conv;
      ms[i] = M;
      conv += conv_bytes;
    }
  }
  MNC_REQUIRE(ms[0] == ms[1], "%s: the two activation panels need one row stride", what);
  {
    const double flops = 4.0 * M * (double)N * K, bytes = (F16 ? 4.0 : 8.0) * ((double)N * K + (double)M * K) + 8.0 * (double)M * N;
    LaunchScope ls(ctx, F16 == 2 ? "fc_bf16" : F16 ? "fc_f16" : "fc_bf16x3", flops, bytes);
    static std::atomic<unsigned long long> attr_set{0};            // one bit per device
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
      MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_lowp_dma_kernel<10, F16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (320 + 256) * 128));
      MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_lowp_dma_kernel<8, F16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128));
      attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    // (several row blocks: row block fastest, so that the workgroups sharing a weight panel are neighbours on one XCD -- fc_lowp)
    const int tm_arg = tm > 1 ? -tm : 1;
    if (mt == 8)
      hipLaunchKernelGGL((fc_lowp_dma_kernel<8, F16>), dim3(2 * tn * splits * tm), dim3(512), 2 * (256 + 256) * 128, ctx->stream, ax[0],
                         (const uint4*)d_w0, d_bias0, d_out0, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, 2 * tn, splits, tm_arg,
                         ms[0], ax[1], (const uint4*)d_w1, d_bias1, d_out1, tn);
    else
      hipLaunchKernelGGL((fc_lowp_dma_kernel<10, F16>), dim3(2 * tn * splits * tm), dim3(512), 2 * (320 + 256) * 128, ctx->stream, ax[0],
                         (const uint4*)d_w0, d_bias0, d_out0, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, 2 * tn, splits, tm_arg,
                         ms[0], ax[1], (const uint4*)d_w1, d_bias1, d_out1, tn);
    rc = ls.finish("fc_lowp_dma_kernel<pair>");
    if (rc) return rc;
  }
  float* outs[2] = {d_out0, d_out1};
  const float* biases[2] = {d_bias0, d_bias1};
  void* osms[2] = {d_osm0, d_osm1};
  bool pair_done = false, pair_sm = false;
  if (splits > 1) {                                  // both products' K ranges summed by ONE launch (grid.y = 2) where they can share it
    LaunchScope ls(ctx, "fc_reduce", 0.0, 8.0 * ((double)splits + 1.0) * M * N);
    pair_done = fc_reduce_pair_launch_sm(ctx->stream, part, part + (size_t)splits * M * N, d_bias0, d_bias1, d_out0, d_out1, M, N, ldc, splits,
                                         act, d_osm0, d_osm1, osm_fmt, M, &pair_sm);
    rc = ls.finish("fc_reduce_kernel<pair>");
    if (rc) return rc;
  }
  for (int i = 0; i < 2; ++i) {
    bool osm_done = pair_done && pair_sm;
    if (splits > 1 && !pair_done) {
      LaunchScope ls(ctx, "fc_reduce", 0.0, 4.0 * ((double)splits + 1.0) * M * N);
      osm_done = fc_reduce_launch_sm(ctx->stream, part + (size_t)i * splits * M * N, biases[i], outs[i], M, N, ldc, splits, act, osms[i],
                                     osms[i] ? osm_fmt : 0, M, 0);
      rc = ls.finish("fc_reduce_kernel");
      if (rc) return rc;
    }
    if (osms[i] && !osm_done) {
      MNC_REQUIRE(ldc == N, "%s: the second output needs a K-split reduction or dense rows", what);
      LaunchScope ls(ctx, osm_fmt != 2 ? "fc_f16_convert" : "fc_bf16x3_split", 0.0, (osm_fmt != 2 ? 6.0 : 8.0) * M * (double)N);
      if (osm_fmt != 2) f16_pack_launch(ctx, outs[i], (uint4*)osms[i], M, N, M, 1, osm_fmt == 3);
      else x3_pack_launch(ctx, outs[i], (uint4*)osms[i], M, N, M, 1);
      rc = ls.finish("pack kernel");
      if (rc) return rc;
    }
  }
  return MNC_OK;
}

extern "C" {

int mnc_pack_fc_bf16x3(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K) {
  MNC_REQUIRE(ctx && d_w && d_packed && N > 0 && K > 0 && K % kXBK == 0, "mnc_pack_fc_bf16x3: bad argument (K%%32==0)");
  LaunchScope ls(ctx, "pack_fc_bf16x3");
  x3_pack_launch(ctx, d_w, (uint4*)d_packed, N, K, kXBN, cdiv(N, kXBN));
  return ls.finish("pack_x3_kernel");
}

int mnc_fc_bf16x3(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M, int N,
                  int K, int ldc, int act) {
  MNC_REQUIRE(ctx && d_a && d_w_packed && d_bias && d_out, "mnc_fc_bf16x3: null pointer");
  MNC_REQUIRE(M >= 0 && N > 0 && K > 0 && K % kXBK == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_bf16x3: unsupported shape M=%d N=%d K=%d ldc=%d act=%d (need K%%32==0)", M, N, K, ldc, act);
  return fc_lowp<0>(ctx, "mnc_fc_bf16x3", d_a, nullptr, M, d_w_packed, d_bias, d_out, M, N, K, ldc, act);
}

int mnc_fc_bf16x3_pre(mnc_ctx* ctx, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias, float* d_out,
                      int M, int N, int K, int ldc, int act) {
  MNC_REQUIRE(ctx && d_a_sm && d_w_packed && d_bias && d_out, "mnc_fc_bf16x3_pre: null pointer");
  MNC_REQUIRE(M >= 0 && m_stride >= M && N > 0 && K > 0 && K % kXBK == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_bf16x3_pre: unsupported shape M=%d (stride %d) N=%d K=%d ldc=%d act=%d (need K%%32==0)", M, m_stride, N, K,
              ldc, act);
  return fc_lowp<0>(ctx, "mnc_fc_bf16x3_pre", nullptr, (const uint4*)d_a_sm, m_stride, d_w_packed, d_bias, d_out, M, N, K, ldc, act);
}

int mnc_pack_fc_f16(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K) {
  MNC_REQUIRE(ctx && d_w && d_packed && N > 0 && K > 0 && K % 64 == 0, "mnc_pack_fc_f16: bad argument (K%%64==0)");
  LaunchScope ls(ctx, "pack_fc_f16");
  f16_pack_launch(ctx, d_w, (uint4*)d_packed, N, K, kXBN, cdiv(N, kXBN));
  return ls.finish("pack_f16_kernel");
}

// Plain bf16 (round 4, the "bf16" math mode): both operands rounded to bf16 (nearest even), one v_mfma_f32_32x32x16_bf16 per term.
int mnc_pack_fc_bf16(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K) {
  MNC_REQUIRE(ctx && d_w && d_packed && N > 0 && K > 0 && K % 64 == 0, "mnc_pack_fc_bf16: bad argument (K%%64==0)");
  LaunchScope ls(ctx, "pack_fc_bf16");
  f16_pack_launch(ctx, d_w, (uint4*)d_packed, N, K, kXBN, cdiv(N, kXBN), 1);
  return ls.finish("pack_f16_kernel<bf16>");
}

int mnc_fc_bf16(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M, int N, int K,
                int ldc, int act) {
  MNC_REQUIRE(ctx && d_a && d_w_packed && d_bias && d_out, "mnc_fc_bf16: null pointer");
  MNC_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 64 == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_bf16: unsupported shape M=%d N=%d K=%d ldc=%d act=%d (need K%%64==0)", M, N, K, ldc, act);
  return fc_lowp<2>(ctx, "mnc_fc_bf16", d_a, nullptr, M, d_w_packed, d_bias, d_out, M, N, K, ldc, act);
}

// InnerProduct in fp16 arithmetic (fp32 accumulate): the same launcher with 64-value stages.
int mnc_fc_f16(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M, int N, int K,
               int ldc, int act) {
  MNC_REQUIRE(ctx && d_a && d_w_packed && d_bias && d_out, "mnc_fc_f16: null pointer");
  MNC_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 64 == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_f16: unsupported shape M=%d N=%d K=%d ldc=%d act=%d (need K%%64==0)", M, N, K, ldc, act);
  return fc_lowp<1>(ctx, "mnc_fc_f16", d_a, nullptr, M, d_w_packed, d_bias, d_out, M, N, K, ldc, act);
}

int mnc_fc_f16_pre(mnc_ctx* ctx, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias, float* d_out,
                   int M, int N, int K, int ldc, int act) {
  MNC_REQUIRE(ctx && d_a_sm && d_w_packed && d_bias && d_out, "mnc_fc_f16_pre: null pointer");
  MNC_REQUIRE(M >= 0 && m_stride >= M && N > 0 && K > 0 && K % 64 == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_f16_pre: unsupported shape M=%d (stride %d) N=%d K=%d ldc=%d act=%d (need K%%64==0)", M, m_stride, N, K, ldc,
              act);
  return fc_lowp<1>(ctx, "mnc_fc_f16_pre", nullptr, (const uint4*)d_a_sm, m_stride, d_w_packed, d_bias, d_out, M, N, K, ldc, act);
}

// The general form: activations as fp32 rows (d_a) or already stage-major (d_a_sm, m_stride rows) -- exactly one of the two --
// and, optionally, the result rows a second time in the stage-major form of the NEXT reduced-precision InnerProduct
// (d_out_sm: [N/64][M][64] halves for out_sm_fmt 1, [N/32][M][4][hi x8 | lo x8] for 2; written by the K-split reduction).
int mnc_fc_f16_ex(mnc_ctx* ctx, const float* d_a, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias,
                  float* d_out, int M, int N, int K, int ldc, int act, void* d_out_sm, int out_sm_fmt) {
  MNC_REQUIRE(ctx && (d_a != nullptr) != (d_a_sm != nullptr) && d_w_packed && d_bias && d_out, "mnc_fc_f16_ex: null pointer / both inputs");
  MNC_REQUIRE(M >= 0 && (d_a || m_stride >= M) && N > 0 && K > 0 && K % 64 == 0 && ldc >= N && act >= 0 && act <= 2 &&
                  (!d_out_sm || (((out_sm_fmt == 1 || out_sm_fmt == 3) && N % 64 == 0) || (out_sm_fmt == 2 && N % 32 == 0))),
              "mnc_fc_f16_ex: unsupported shape M=%d N=%d K=%d ldc=%d act=%d out_sm_fmt=%d", M, N, K, ldc, act, out_sm_fmt);
  return fc_lowp<1>(ctx, "mnc_fc_f16_ex", d_a, (const uint4*)d_a_sm, d_a ? M : m_stride, d_w_packed, d_bias, d_out, M, N, K, ldc, act,
                    d_out_sm, d_out_sm ? out_sm_fmt : 0, M, 0);
}

int mnc_fc_bf16x3_ex(mnc_ctx* ctx, const float* d_a, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias,
                     float* d_out, int M, int N, int K, int ldc, int act, void* d_out_sm, int out_sm_fmt) {
  MNC_REQUIRE(ctx && (d_a != nullptr) != (d_a_sm != nullptr) && d_w_packed && d_bias && d_out, "mnc_fc_bf16x3_ex: null pointer / both inputs");
  MNC_REQUIRE(M >= 0 && (d_a || m_stride >= M) && N > 0 && K > 0 && K % kXBK == 0 && ldc >= N && act >= 0 && act <= 2 &&
                  (!d_out_sm || (((out_sm_fmt == 1 || out_sm_fmt == 3) && N % 64 == 0) || (out_sm_fmt == 2 && N % 32 == 0))),
              "mnc_fc_bf16x3_ex: unsupported shape M=%d N=%d K=%d ldc=%d act=%d out_sm_fmt=%d", M, N, K, ldc, act, out_sm_fmt);
  return fc_lowp<0>(ctx, "mnc_fc_bf16x3_ex", d_a, (const uint4*)d_a_sm, d_a ? M : m_stride, d_w_packed, d_bias, d_out, M, N, K, ldc, act,
                    d_out_sm, d_out_sm ? out_sm_fmt : 0, M, 0);
}

int mnc_fc_bf16_ex(mnc_ctx* ctx, const float* d_a, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias,
                   float* d_out, int M, int N, int K, int ldc, int act, void* d_out_sm, int out_sm_fmt) {
  MNC_REQUIRE(ctx && (d_a != nullptr) != (d_a_sm != nullptr) && d_w_packed && d_bias && d_out, "mnc_fc_bf16_ex: null pointer / both inputs");
  MNC_REQUIRE(M >= 0 && (d_a || m_stride >= M) && N > 0 && K > 0 && K % 64 == 0 && ldc >= N && act >= 0 && act <= 2 &&
                  (!d_out_sm || (out_sm_fmt == 3 && N % 64 == 0)),
              "mnc_fc_bf16_ex: unsupported shape M=%d N=%d K=%d ldc=%d act=%d out_sm_fmt=%d (the bf16 form is format 3)", M, N, K, ldc, act,
              out_sm_fmt);
  return fc_lowp<2>(ctx, "mnc_fc_bf16_ex", d_a, (const uint4*)d_a_sm, d_a ? M : m_stride, d_w_packed, d_bias, d_out, M, N, K, ldc, act,
                    d_out_sm, d_out_sm ? out_sm_fmt : 0, M, 0);
}

// fp32 row-major [M][K] -> the stage-major 2-byte activation form of mnc_fc_{f16,bf16x3}_pre (what those entry points' producers
// write in their epilogues): f16 != 0: [K/64][M][64 halves], M*K*2 bytes; else [K/32][M][(hi x8 | lo x8) x 4] bf16, M*K*4 bytes.
int mnc_fc_pack_act(mnc_ctx* ctx, const float* d_a, void* d_a_sm, int M, int K, int f16) {
  MNC_REQUIRE(ctx && d_a && d_a_sm && M > 0 && K > 0 && f16 >= 0 && f16 <= 2 && K % (f16 ? 64 : kXBK) == 0, "mnc_fc_pack_act: bad argument");
  LaunchScope ls(ctx, f16 ? "fc_f16_convert" : "fc_bf16x3_split", 0.0, (f16 ? 6.0 : 8.0) * M * (double)K);
  if (f16) f16_pack_launch(ctx, d_a, (uint4*)d_a_sm, M, K, M, 1, f16 == 2);      // (2, round 6: bf16 in fp16's layout = stage-major format 3)
  else x3_pack_launch(ctx, d_a, (uint4*)d_a_sm, M, K, M, 1);
  return ls.finish(f16 ? "pack_f16_kernel" : "pack_x3_kernel");
}

int mnc_fc_unpack_act(mnc_ctx* ctx, const void* d_a_sm, float* d_a, int M, int K, int fmt) {
  MNC_REQUIRE(ctx && d_a_sm && d_a && M > 0 && K > 0 && fmt >= 1 && fmt <= 3 && K % (fmt == 2 ? kXBK : 64) == 0,
              "mnc_fc_unpack_act: bad argument (fmt 1 / 3: K %% 64 == 0, fmt 2: K %% 32 == 0)");
  LaunchScope ls(ctx, "fc_act_unpack", 0.0, (fmt == 2 ? 8.0 : 6.0) * M * (double)K);
  long g = ((long)M * (K / 8) + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(unpack_act_sm_kernel, dim3((int)g), dim3(256), 0, ctx->stream, (const uint4*)d_a_sm, d_a, (long)M, (long)K, fmt);
  return ls.finish("unpack_act_sm_kernel");
}

// Two InnerProducts of one shape in reduced precision (mode 1 = fp16, 2 = plain bf16; each input as fp32 rows OR stage-major, each
// output optionally a second time in the next InnerProduct's stage-major form): see fc_lowp_pair above.  mode 0 = split bf16 (32-deep
// stages, >= 16 per K range).  Shapes the paired kernel does not take: exactly the two single calls.
int mnc_fc_lowp_pair(mnc_ctx* ctx, int mode, const float* d_a0, const void* d_a_sm0, const float* d_a1, const void* d_a_sm1, int m_stride,
                     const void* d_w0, const void* d_w1, const float* d_bias0, const float* d_bias1, float* d_out0, float* d_out1, int M,
                     int N, int K, int ldc, int act, void* d_out_sm0, void* d_out_sm1, int out_sm_fmt) {
  MNC_REQUIRE(ctx && mode >= 0 && mode <= 2 && (d_a0 != nullptr) != (d_a_sm0 != nullptr) && (d_a1 != nullptr) != (d_a_sm1 != nullptr) && d_w0 &&
                  d_w1 && d_bias0 && d_bias1 && d_out0 && d_out1, "mnc_fc_lowp_pair: null pointer / both inputs");
  MNC_REQUIRE(M >= 0 && ((d_a0 && d_a1) || m_stride >= M) && N > 0 && K > 0 && K % (mode ? 64 : kXBK) == 0 && ldc >= N && act >= 0 && act <= 2 &&
                  ((!d_out_sm0 && !d_out_sm1) || (((out_sm_fmt == 1 || out_sm_fmt == 3) && N % 64 == 0) || (out_sm_fmt == 2 && N % 32 == 0))),
              "mnc_fc_lowp_pair: unsupported shape M=%d N=%d K=%d ldc=%d act=%d out_sm_fmt=%d", M, N, K, ldc, act, out_sm_fmt);
  if (mode == 0)
    return fc_lowp_pair<0>(ctx, "mnc_fc_lowp_pair", d_a0, (const uint4*)d_a_sm0, d_a1, (const uint4*)d_a_sm1, m_stride, d_w0, d_w1, d_bias0,
                           d_bias1, d_out0, d_out1, M, N, K, ldc, act, d_out_sm0, d_out_sm1, out_sm_fmt);
  if (mode == 1)
    return fc_lowp_pair<1>(ctx, "mnc_fc_lowp_pair", d_a0, (const uint4*)d_a_sm0, d_a1, (const uint4*)d_a_sm1, m_stride, d_w0, d_w1, d_bias0,
                           d_bias1, d_out0, d_out1, M, N, K, ldc, act, d_out_sm0, d_out_sm1, out_sm_fmt);
  return fc_lowp_pair<2>(ctx, "mnc_fc_lowp_pair", d_a0, (const uint4*)d_a_sm0, d_a1, (const uint4*)d_a_sm1, m_stride, d_w0, d_w1, d_bias0,
                         d_bias1, d_out0, d_out1, M, N, K, ldc, act, d_out_sm0, d_out_sm1, out_sm_fmt);
}

}  // extern "C"
