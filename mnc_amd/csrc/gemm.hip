// InnerProduct layers of the MNC heads (test.prototxt:509-785, 824-1106) for gfx950:
//     out[M][N] = act(A[M][K] . W[N][K]^T + bias[N]),   M = #RoIs (<= 300), K up to 100352, N up to 4096.
//
// At M = 300 the fp32 arithmetic intensity is ~150 FLOP/B, far above the fp32-matrix ridge (157 TF / ~6 TB/s ~ 25),
// so these GEMMs are MFMA-bound, not weight-streaming-bound; the kernel is built around v_mfma_f32_32x32x2_f32
// (exact fp32).
//
// Tiling: a workgroup (4 waves) owns ALL rows of a 320-row block (10 MFMA row tiles; M = 300 fits one block, so the
// weights are streamed from HBM exactly once) x 128 output columns (wave w owns columns 32w..32w+31) x one K split.
// Per 32-deep K stage the A panel [320][32] and the W panel [128][32] go global -> registers -> LDS (row pitch 36
// floats: conflict-free ds_read_b128 fragments), double-buffered with one barrier per stage; the loads of stage s+1
// are issued before the 160 MFMAs per wave of stage s.  The MFMA A operand is the activation (row = RoI), the B
// operand the weight (column = output), so for a fixed accumulator register 32 lanes hold 32 consecutive outputs of
// one RoI -> 128-byte contiguous stores.
//
// Split-K (chosen so that tiles x splits ~ the 256 CUs) writes fp32 partials to the context scratch; a second kernel
// sums them in fixed order (deterministic) and applies bias + activation.  With one split the epilogue is fused.
#include <atomic>
#include <type_traits>
#include <cstdlib>

#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBN = 128, kBK = 32;       // kBK: K granularity of the interface (K % 32 == 0) and the default stage depth
// Two row-tile counts per workgroup: MT = 10 (320 rows: all RoIs in one block, weights streamed once -- the big FCs) and
// MT = 2 (64 rows: the small GEMMs -- mask_pred 256->441 and the 8192->{21,21,84} heads -- where 320-row blocks would
// leave most CUs idle; 55 KB of LDS lets two such workgroups share a CU).

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 1.0f / (1.0f + expf(-v));
  return v;
}

// grid: (ceil(N/128), splits, ceil(M/320)).  k range of split s: [s*kper, min(K, (s+1)*kper)), kper % 32 == 0.
// fused != 0: write act(acc + bias) to out (ldc); else write raw partials to part[split][M][N].
// ABL != 0: ablation builds for tuning (MNC_FC_ABL, kMT = 10 only): 1 = no global loads / LDS stores in the loop,
// 2 = additionally no barrier, 3 = additionally no LDS fragment reads, 4 = global loads issued and awaited but not stored.
// kSK = K values per stage (32 or 16).  16 halves the LDS footprint: the 160-row variant <5, 16> needs 46 KB and two or three
// workgroups share a CU, so one workgroup's staging / barrier phases are covered by another's MFMAs (at one workgroup per CU
// -- the 320-row variant -- the phases of the single wave per SIMD largely add up, see the ablation numbers below).
template <int kMT, int kSK = 32, int ABL = 0>
__global__ __launch_bounds__(256) void fc_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      float* __restrict__ part, int M, int N, int K, int ldc, int kper,
                                                      int act, int fused, int tn_, int splits_, int tm_) {
  constexpr int kBM = 32 * kMT;
  constexpr int kPitch = kSK + 4;          // floats per LDS row: (kSK + 4) / 4 is odd -> conflict-free ds_read_b128
  constexpr int kC4 = kSK / 4;             // float4 per row and stage
  constexpr int kAPer = (kBM * kC4 + 255) / 256;   // float4 staging items per thread for the A panel
  constexpr int kBPer = (kBN * kC4 + 255) / 256;   // ... and for the weight panel
  constexpr int kNG = kSK / 8;             // K-groups (8 values: one 16-byte fragment per lane) per stage
  static_assert(kSK == 32 || kSK == 16, "stage depth");
  __shared__ __attribute__((aligned(16))) float sA[2][kBM * kPitch];
  __shared__ __attribute__((aligned(16))) float sB[2][kBN * kPitch];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kk = lane >> 5;
  int bn, split, bmz;
  xcd_decode(blockIdx.x, tn_, splits_, tm_, bn, split, bmz);
  const int n0 = bn * kBN, m0 = bmz * kBM;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const int nstages = (kend - kbeg) / kSK;
  const int mrows = min(M - m0, kBM);
  const int mtiles = (mrows + 31) >> 5;

  // staging map: item q -> row q / kC4, float4 column q % kC4 (rows are K-contiguous in global memory); surplus items are
  // clamped onto the last row and simply rewrite it
  const float* a_src[kAPer];
  const float* b_src[kBPer];
  int a_dst[kAPer], b_dst[kBPer];
#pragma unroll
  for (int u = 0; u < kAPer; ++u) {
    const int q = tid + u * 256, r = min(q / kC4, kBM - 1), c4 = q % kC4;
    const int gr = m0 + min(r, mrows - 1);               // rows past M re-read the last valid row; never stored
    a_src[u] = A + (long)gr * K + kbeg + c4 * 4;
    a_dst[u] = r * kPitch + c4 * 4;
  }
#pragma unroll
  for (int u = 0; u < kBPer; ++u) {
    const int q = tid + u * 256, r = min(q / kC4, kBN - 1), c4 = q % kC4;
    const int gr = min(n0 + r, N - 1);
    b_src[u] = Wt + (long)gr * K + kbeg + c4 * 4;
    b_dst[u] = r * kPitch + c4 * 4;
  }
  // Two register sets (R0/R1): the loads of stage s+2 are in flight while stage s is multiplied and stage s+1 -- already
  // in registers -- is written to the free LDS buffer.  All staging is branch-free (clamped stage index; a phantom stage
  // behind an odd stage count is stored as zeros): loads under a branch make hipcc drain the prefetch pipeline with
  // s_waitcnt vmcnt(0) at every join.  (Initialised: uninitialised staging arrays become scratch, see conv.hip.)
  struct Regs { float4 a[kAPer]; float4 b[kBPer]; };
  Regs R0, R1;
#pragma unroll
  for (int u = 0; u < kAPer; ++u) R0.a[u] = R1.a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < kBPer; ++u) R0.b[u] = R1.b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_stage = [&](int s, Regs& R) {
    const long off = (long)min(s, nstages - 1) * kSK;
#pragma unroll
    for (int u = 0; u < kAPer; ++u) R.a[u] = *reinterpret_cast<const float4*>(a_src[u] + off);
#pragma unroll
    for (int u = 0; u < kBPer; ++u) R.b[u] = *reinterpret_cast<const float4*>(b_src[u] + off);
  };
  auto store_stage = [&](int buf, const Regs& R, bool live) {
    const unsigned keep = live ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int u = 0; u < kAPer; ++u) {
      float4 v = R.a[u];
      v.x = __uint_as_float(__float_as_uint(v.x) & keep);
      v.y = __uint_as_float(__float_as_uint(v.y) & keep);
      v.z = __uint_as_float(__float_as_uint(v.z) & keep);
      v.w = __uint_as_float(__float_as_uint(v.w) & keep);
      *reinterpret_cast<float4*>(&sA[buf][a_dst[u]]) = v;        // item u == row tile u
    }
#pragma unroll
    for (int u = 0; u < kBPer; ++u) *reinterpret_cast<float4*>(&sB[buf][b_dst[u]]) = R.b[u];
  };

  f32x16 acc[kMT];
#pragma unroll
  for (int t = 0; t < kMT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int a_base = j * kPitch + kk * 4;
  const int b_base = (wave * 32 + j) * kPitch + kk * 4;
  // Software pipeline (one wave per SIMD: only the wave's own MFMAs can hide its LDS and global latency; left alone the
  // phases simply add up -- fc6: MFMA 464 us + fragment reads 61 us + barrier 12 us + staging 82 us = 619 us):
  //   * the fragments of K-group kc+1 (8 k-values: one 16-byte read per row tile + one for the column) are read while the
  //     40 MFMAs of group kc run (two fragment sets);
  //   * the stage barrier sits before the LAST group, so group 0 of the next stage is prefetched during group 3;
  //   * global loads (stage s+2) and LDS writes (stage s+1) are spread over groups 0-2;
  //   * sched_group_barrier pins the interleave, the empty asm on the accumulators keeps each region's MFMAs inside it.
  struct Frags { float4 a[kMT]; float4 b; };
  auto read_frags = [&](int buf, int kc, Frags& f) {
    if (ABL == 3) {
      f.b = make_float4(1.f, 2.f, 3.f, 4.f);
      asm volatile("" : "+v"(f.b.x), "+v"(f.b.y), "+v"(f.b.z), "+v"(f.b.w));
#pragma unroll
      for (int t = 0; t < kMT; ++t) f.a[t] = f.b;
      return;
    }
    f.b = *reinterpret_cast<const float4*>(sB[buf] + b_base + kc * 8);
    // no per-tile branch: rows past M are clamped copies, multiplied but never stored
#pragma unroll
    for (int t = 0; t < kMT; ++t) f.a[t] = *reinterpret_cast<const float4*>(sA[buf] + a_base + t * 32 * kPitch + kc * 8);
  };
  auto mfmas = [&](const Frags& f) {                 // k outermost: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int t = 0; t < kMT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].x, f.b.x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < kMT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].y, f.b.y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < kMT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].z, f.b.z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < kMT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].w, f.b.w, acc[t], 0, 0, 0);
  };
  auto pin_acc = [&]() {
#pragma unroll
    for (int t = 0; t < kMT; ++t) asm volatile("" : "+a"(acc[t]));
  };
  constexpr int kNM = 4 * kMT;                       // MFMAs per K-group
  constexpr int kNR = kMT + 1;                       // fragment reads per K-group
  constexpr int kNS = kAPer + kBPer;                 // global loads (= LDS writes) per stage and thread
  // stage s sits in LDS[buf] with its group-0 fragments in f0; stage s+1 is in `cur`; stage s+2 is requested into `nxt`
  auto stage_io = [&](int s, int buf, Regs& cur, Regs& nxt) {
    if (ABL == 0) store_stage(buf ^ 1, cur, s + 1 < nstages);
    if (ABL == 4) {                                  // loads issued and awaited, never written to LDS
#pragma unroll
      for (int u = 0; u < kAPer; ++u) asm volatile("" :: "v"(cur.a[u].x));
#pragma unroll
      for (int u = 0; u < kBPer; ++u) asm volatile("" :: "v"(cur.b[u].x));
    }
  };
  auto step = [&](int s, int buf, Regs& cur, Regs& nxt, Frags& f0) {
    Frags f1;
    // ---- region 1: groups 0 .. kNG-2 (+ the reads for the last group), loads of stage s+2, LDS writes of stage s+1 ----
    if (ABL == 0 || ABL == 4) load_stage(s + 2, nxt);
    read_frags(buf, 1, f1);
    mfmas(f0);                                       // group 0
    stage_io(s, buf, cur, nxt);
    if (kNG == 4) {
      read_frags(buf, 2, f0);
      mfmas(f1);                                     // group 1
      read_frags(buf, 3, f1);
      mfmas(f0);                                     // group 2
    }
    constexpr int kG1 = kNG - 1;                     // groups in region 1
#pragma unroll
    for (int i = 0; i < kG1 * kNM; ++i) {            // one slot per MFMA; the other classes spread evenly over the slots
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ABL != 3 && (i + 1) * kG1 * kNR / (kG1 * kNM) > i * kG1 * kNR / (kG1 * kNM)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if ((ABL == 0 || ABL == 4) && (i + 1) * kNS / (kG1 * kNM) > i * kNS / (kG1 * kNM))
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (ABL == 0 && (i + 1) * kNS / (kG1 * kNM) > i * kNS / (kG1 * kNM)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    pin_acc();
    if (ABL < 2) __syncthreads();
    // ---- region 2: the last group, prefetching group 0 of the next stage (of the zero-filled phantom at the very end) ----
    read_frags(buf ^ 1, 0, f0);
    mfmas(f1);
#pragma unroll
    for (int i = 0; i < kNM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ABL != 3 && (i + 1) * kNR / kNM > i * kNR / kNM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
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

  // D[row = m (reg&3)+8*(reg>>2)+4*kk][col = n j]
  const int n = n0 + wave * 32 + j;
  if (n < N) {
    const float bv = fused ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < kMT; ++t) {
      if (t < mtiles) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
          if (m < M) {
            if (fused) out[(long)m * ldc + n] = apply_act(acc[t][e] + bv, act);
            else part[((long)split * M + m) * N + n] = acc[t][e];
          }
        }
      }
    }
  }
}

// ---- the 320-row kernel with its operand panels copied by LDS-DMA -------------------------------------------------------------
// Ablations of fc_mfma_kernel<10, 32> on fc6 (M = 300, N = 4096, K = 25088, random operands, tools/kernel_bench.py fc with
// MNC_FC_ABL): 621 us; 570 without the LDS stores of the staged panels; 520 without their global loads as well; 469 for the MFMAs
// alone -- one wave per SIMD pays for the 14 ds_write_b128 + 14 global loads per thread and stage (and their 190 VGPRs)
// although none of them is on a critical path.  Here a stage's panels ([320][32] activations + [128][32] weights = 56 KB) go
// global -> LDS with global_load_lds_dwordx4 (1 KB per wave instruction, 14 per wave and stage): no staging registers, no
// ds_write, no keep masks.  A DMA instruction places its 64 lanes' 16-byte pieces CONTIGUOUSLY in LDS, so rows cannot be
// padded; instead row r (128 bytes = 8 chunks) stores its k-chunk c in chunk slot c ^ ((r >> 1) & 7) -- the lane picks the
// global address for its slot -- and a fragment read of chunk c goes to that slot: within each 16-lane group of a
// ds_read_b128 (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a tile) the pairs (r & 1, (r >> 1) & 7) are all
// different, i.e. 16 distinct starts of 4 banks: conflict-free without padding.
// Pipeline as fc_mfma_kernel's (fragments of K-group g + 1 read during the MFMAs of group g, stage barrier before the last
// group), with the copy of stage s + 2 issued right BEHIND the barrier of stage s -- every wave is past its reads of that
// buffer -- so it has a full stage of MFMAs to land and the vmcnt(0) hipcc puts in front of a barrier while a DMA is in flight
// is the wait the next stage needs anyway.
// kWM = waves along M: 1 -> four waves, each all kMT row tiles x its 32 columns (one wave per SIMD); 2 -> eight waves, wave
// (wm, wn) owns kMT / 2 row tiles x columns 32 wn: half the accumulators, two waves per SIMD -- one wave's barrier and fragment
// waits run under its partner's MFMAs.
template <int kMT, int ABL = 0, int kWM = 1, int HALF = 0>
__global__ __launch_bounds__(256 * kWM) void fc_mfma_dma_kernel(const float* __restrict__ A, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          float* __restrict__ part, int M, int N, int K, int ldc, int kper,
                                                          int act, int fused, int tn_, int splits_, int tm_) {
  constexpr int kBM = 32 * kMT;
  constexpr int kRows = kBM + kBN;                   // operand rows per stage: activations, then weights
  constexpr int kNW = 4 * kWM;                       // waves
  constexpr int TR = kMT / kWM;                      // row tiles per wave
  constexpr int kPer = kRows * 128 / 1024 / kNW;     // DMA instructions per wave and stage (1 KB = 8 rows each)
  static_assert(kRows % (8 * kNW) == 0 && kMT % kWM == 0, "whole pieces and row tiles per wave");
  // Two stage buffers 64 KB apart (57 KB used each), so that every LDS offset switches buffer with one XOR.  The loop runs ONE
  // stage per iteration with the DMA issue as its last LDS-related instruction: hipcc's wait-count pass makes every ds_read that
  // FOLLOWS a DMA issue in straight-line code wait for vmcnt(0) ("may alias"; the reads go to the other buffer, the barrier
  // protocol orders the real dependence), but not the reads of the next iteration -- with two stages per iteration the second
  // stage's fragment reads each waited for the copy issued by the first.
  extern __shared__ __attribute__((aligned(1024))) char s_fc_dma[];
  constexpr int kBufXor = 65536;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;           // column group, row half
  const int j = lane & 31, kk = lane >> 5;
  int bn, split, bmz;
  xcd_decode(blockIdx.x, tn_, splits_, tm_, bn, split, bmz);
  const int n0 = bn * kBN, m0 = bmz * kBM;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const int nstages = (kend - kbeg) / 32;
  const int mrows = min(M - m0, kBM);
  const int mtiles = (mrows + 31) >> 5;

  // piece p = wave + kNW i covers buffer bytes [1024 p, 1024 p + 1024): lane L fills slot 64 p + L = (row r, chunk slot L & 7)
  const float* src[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int slot = (wave + kNW * i) * 64 + lane, r = slot >> 3, c = (slot & 7) ^ ((r >> 1) & 7);
    src[i] = (r < kBM ? A + (long)(m0 + min(r, mrows - 1)) * K               // rows past M re-read the last valid row; never stored
                      : Wt + (long)min(n0 + r - kBM, N - 1) * K) + kbeg + c * 4;
  }
  // The copy is issued through inline assembly: hipcc's wait-count pass makes the first ds_read behind a DMA builtin wait for
  // vmcnt(0) ("the LDS it writes may alias"), also across the loop's back edge, which would expose the whole global latency once
  // per stage.  The reads behind an issue go to the OTHER buffer; the real dependence -- stage s + 1 complete before anybody reads
  // it -- is the explicit s_waitcnt vmcnt(0) + barrier below.  (s_nop: one wait state between the write of M0 and its use.)
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_fc_dma;
  auto dma_piece = [&](int i, long off, int buf_byte) {
    if (ABL == 3 && wave + kNW * i >= kBM / 8) off = 0;            // tuning: weight pieces re-read stage 0 (activations stream)
    const float* g = src[i] + off;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf_byte + (unsigned)(wave + kNW * i) * 1024u);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l));
  };
  auto dma_stage = [&](int s, int buf_byte) {
    const long off = ABL == 1 ? 0 : (long)min(s, nstages - 1) * 32;   // past the end: the last stage once more, never multiplied
#pragma unroll
    for (int i = 0; i < kPer; ++i) dma_piece(i, off, buf_byte);
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  f32x16 acc[TR];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // HALF (M in (32 kMT - 32, 32 kMT - 16], e.g. 300 RoIs = 9 row tiles + 12 rows): the last row tile holds at most 16 live rows, so
  // the waves that own it (wm = kWM - 1) multiply it with v_mfma_f32_16x16x4_f32 on rows [32 (kMT - 1), +16) only -- 16 x 32
  // outputs as two 16-column halves, 4 K-values per instruction: four 32-cycle MFMAs per K-group instead of four 64-cycle ones,
  // i.e. 304 rows of matrix-pipe work instead of 320 (the padding of 300 -> 320 was 6.7 % of the pipe time).  Operand layout of
  // that instruction: lane l supplies A[row l % 16][k = l / 16] and B[col l % 16][k = l / 16]; D register i is row 4 (l / 16) + i.
  // A K-group (8 values = chunks 2 kc, 2 kc + 1 of a row) is two such MFMAs; each lane reads ONE float per operand and MFMA
  // (ds_read_b32 at element l / 16 of the chunk; 16 rows x 4 elements land on 64 different banks under the row swizzle).
  constexpr bool kHalf = HALF != 0;
  static_assert(!kHalf || (kWM == 2 && ABL == 0), "the half tile exists in the eight-wave product build only");
  const bool last_wave = kHalf && wm == kWM - 1;                 // wave-uniform
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v hacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // K order: a 32 x 32 x 2 MFMA on component q of the fragments multiplies k = q (lanes 0-31) and k = 4 + q (lanes 32-63) of the
  // K-group, so an output accumulates k0, k4, k1, k5, k2, k6, k3, k7.  The two 16 x 16 x 4 MFMAs keep that order: MFMA h takes
  // (k_2h, k_4+2h, k_2h+1, k_4+2h+1) from its lane groups 0..3, i.e. group g reads chunk 2 kc + (g & 1), element 2 h + (g >> 1).
  const int r16 = lane & 15, g4 = lane >> 4;
  int h_off[8];                                                   // [2 kc + h]: byte offset inside this lane's half-tile row
#pragma unroll
  for (int c = 0; c < 8; ++c)
    h_off[c] = ((2 * (c >> 1) + (g4 & 1)) ^ ((r16 >> 1) & 7)) * 16 + (2 * (c & 1) + (g4 >> 1)) * 4;
  const int ha_base = ((kMT - 1) * 32 + r16) * 128, hb_base = (kBM + wn * 32 + r16) * 128;
  int hflip = 0;                                                  // which stage buffer the half-tile offsets point into

  // fragment of K-group kc (8 k values): chunk 2 kc + kk of row (tile row j) -> slot (2 kc + kk) ^ ((j >> 1) & 7); the tile
  // offsets (32 rows = 4 KB) and the weight rows (kBM + 32 wave + j) leave (row >> 1) & 7 unchanged.  Byte offsets.
  int a_off[4], b_off[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const int c = (2 * kc + kk) ^ ((j >> 1) & 7);
    a_off[kc] = ((wm * TR * 32 + j) * 32 + c * 4) * 4;
    b_off[kc] = ((kBM + wn * 32 + j) * 32 + c * 4) * 4;
  }
  struct Frags { float4 a[TR]; float4 b; float ha[2]; float hb[2][2]; };
  // LAST: the wave's last row tile is the half tile (its 32 x 32 fragment is not read)
  auto read_frags = [&](int kc, int flip, Frags& f, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    f.b = *reinterpret_cast<const float4*>(s_fc_dma + (b_off[kc] ^ flip));
#pragma unroll
    for (int t = 0; t < (LAST ? TR - 1 : TR); ++t) f.a[t] = *reinterpret_cast<const float4*>(s_fc_dma + (a_off[kc] ^ flip) + t * 4096);
    if (LAST) {
      const int hf = hflip ^ flip;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f.ha[h] = *reinterpret_cast<const float*>(s_fc_dma + ((ha_base + h_off[2 * kc + h]) ^ hf));
        f.hb[h][0] = *reinterpret_cast<const float*>(s_fc_dma + ((hb_base + h_off[2 * kc + h]) ^ hf));
        f.hb[h][1] = *reinterpret_cast<const float*>(s_fc_dma + ((hb_base + 2048 + h_off[2 * kc + h]) ^ hf));
      }
    }
  };
  auto half_mfmas = [&](const Frags& f) {            // k 0-3 then k 4-7 of the group, the two 16-column halves alternating
    hacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha[0], f.hb[0][0], hacc[0], 0, 0, 0);
    hacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha[0], f.hb[0][1], hacc[1], 0, 0, 0);
    hacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha[1], f.hb[1][0], hacc[0], 0, 0, 0);
    hacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ha[1], f.hb[1][1], hacc[1], 0, 0, 0);
  };
  auto mfmas = [&](const Frags& f, auto last_tag) {  // k outermost: consecutive MFMAs go to different accumulators
    constexpr bool LAST = decltype(last_tag)::value;
    constexpr int NT = LAST ? TR - 1 : TR;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].x, f.b.x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].y, f.b.y, acc[t], 0, 0, 0);
    if (LAST) half_mfmas(f);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].z, f.b.z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t].w, f.b.w, acc[t], 0, 0, 0);
  };
  auto pin_acc = [&]() {
#pragma unroll
    for (int t = 0; t < TR; ++t) asm volatile("" : "+a"(acc[t]));
  };

  auto run = [&](auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    constexpr int NT = LAST ? TR - 1 : TR;
    constexpr int kNM = 4 * NT + (LAST ? 4 : 0), kNR = NT + 1 + (LAST ? 6 : 0);    // MFMAs / LDS reads per K-group
    Frags f0, f1;
    dma_stage(0, 0);
    dma_stage(1, kBufXor);
    dma_wait();
    __syncthreads();
    read_frags(0, 0, f0, last_tag);
    int cur = 0;                                     // byte offset of the buffer stage s sits in (scalar)
    for (int s = 0; s < nstages; ++s) {
      // stage s sits in buffer `cur` (a_off / b_off point into it) with its group-0 fragments in f0; stage s + 1 is landing in
      // (or already in) the other buffer
      read_frags(1, 0, f1, last_tag);
      mfmas(f0, last_tag);                           // group 0
      read_frags(2, 0, f0, last_tag);
      mfmas(f1, last_tag);                           // group 1
      read_frags(3, 0, f1, last_tag);
      mfmas(f0, last_tag);                           // group 2
#pragma unroll
      for (int i = 0; i < 3 * kNM; ++i) {            // one slot per MFMA; the fragment reads spread evenly over the slots
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if ((i + 1) * 3 * kNR / (3 * kNM) > i * 3 * kNR / (3 * kNM)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      pin_acc();
      dma_wait();                                    // stage s + 1 has landed ...
      __syncthreads();                               // ... for every wave, and nobody reads buffer `cur` any more
      read_frags(0, kBufXor, f0, last_tag);          // group 0 of stage s + 1
      {                                              // group 3: the reads under the first MFMAs, then one copy per MFMA
        const long off = ABL == 1 ? 0 : (long)min(s + 2, nstages - 1) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int i = q * NT + t;
            const float av = q == 0 ? f1.a[t].x : q == 1 ? f1.a[t].y : q == 2 ? f1.a[t].z : f1.a[t].w;
            const float bv = q == 0 ? f1.b.x : q == 1 ? f1.b.y : q == 2 ? f1.b.z : f1.b.w;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            if (i >= 4 * NT - kPer - 2 && i < 4 * NT - 2) {
              __builtin_amdgcn_sched_barrier(0);
              dma_piece(i - (4 * NT - kPer - 2), off, cur);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (LAST && q == 0) half_mfmas(f1);
        }
      }
      pin_acc();
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) { a_off[kc] ^= kBufXor; b_off[kc] ^= kBufXor; }
      hflip ^= kBufXor;
      cur ^= kBufXor;
    }
    dma_wait();                                      // the copies issued by the last two stages have landed before the LDS is released
    __syncthreads();
  };
  if (nstages > 0) {
    if (last_wave) run(std::true_type{});
    else run(std::false_type{});
  }

  // D[row = m (reg&3)+8*(reg>>2)+4*kk][col = n j]
  const int n = n0 + wn * 32 + j;
  if (n < N) {
    const float bv = fused ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      if (last_wave && t == TR - 1) continue;        // the half tile is stored below
      if (wm * TR + t < mtiles) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + (wm * TR + t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
          if (m < M) {
            if (fused) out[(long)m * ldc + n] = apply_act(acc[t][e] + bv, act);
            else part[((long)split * M + m) * N + n] = acc[t][e];
          }
        }
      }
    }
  }
  if (last_wave) {                                   // D[row = 4 (lane / 16) + reg][col = lane % 16] of the two 16-column halves
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int nh = n0 + wn * 32 + ch * 16 + r16;
      if (nh >= N) continue;
      const float bh = fused ? bias[nh] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + (kMT - 1) * 32 + 4 * g4 + e;
        if (m < M) {
          if (fused) out[(long)m * ldc + nh] = apply_act(hacc[ch][e] + bh, act);
          else part[((long)split * M + m) * N + nh] = hacc[ch][e];
        }
      }
    }
  }
}

// ---- the eight-wave LDS-DMA kernel on v_mfma_f32_16x16x4_f32 fragments ---------------------------------------------------------
// Same staging, swizzle, barrier protocol and workgroup tile (320 x 128, 32 K-values per stage) as fc_mfma_dma_kernel<10, 0, 2>;
// the wave's 160 x 32 outputs are 10 x 2 accumulators of 16 x 16 (4 registers each) instead of 5 of 32 x 32 (16 each).  Why: an
// fp32 MFMA's own result write-back (32x32x2: 4 KB per 4096 FLOP) competes with every other register-file writer -- LDS reads,
// the DMA's bookkeeping -- for the same ports (tools/probes/mfma_f32_mix_probe.hip: the same FLOPs with the same LDS reads beside
// them run 5 % faster as 16x16x4, 1 KB per 2048 FLOP), and the pipe is power limited, so cycles given back are clock given back.
//   * operand layout of the instruction: lane l supplies A[row l % 16][k = l / 16] and B[col l % 16][k = l / 16]; D register i is
//     row 4 (l / 16) + i, column l % 16.
//   * a K-group is 16 values = chunks 4 G .. 4 G + 3 of a row (G = 0, 1 per stage): lane (r = l % 16, g = l / 16) reads chunk 4 G + g
//     of row r of every 16-row sub-tile with ONE ds_read_b128 -- MFMA q of the group multiplies element q of those chunks, i.e.
//     k = q, 4 + q, 8 + q, 12 + q across its four lane groups.  Per 16 K-values and wave: 10 + 2 ds_read_b128 and 80 MFMAs of
//     32 cycles (the 32x32x2 kernel: 12 and 40 of 64).  Under the row swizzle (slot = chunk ^ ((row >> 1) & 7)) the 16 lanes of every
//     ds_read_b128 service group (rows {0-3, 12-15} of lane group g with rows {4-11} of g + 1, ...) start on 16 different 16-byte
//     slots of the 256-byte bank row: conflict-free, as before.
//   * K order of an output: stage by stage, group by group, q = 0..3, the four k of an MFMA in the hardware's order -- fixed, but
//     not the 32x32x2 kernel's (k0, k4, k1, k5, ...): results differ from it in the last bits.
//   * a block whose last 16-row sub-tile is dead (M in (288, 304], e.g. 300 RoIs) skips that sub-tile's MFMAs in the waves that
//     own it (LAST): the 32x32x2 kernel's half-tile special case is the general case here.
//   * ABL (tuning builds, wrong results): 1 no copies inside the loop, 2 no barrier inside the loop -- what they cost (kernel_bench fc, MNC_FC_DMA_ABL = 16 + ABL)
//   * BUF: the copies as buffer_load_dwordx4 ... lds (a descriptor per operand, the stage in the scalar offset, a 32-bit lane offset:
//     no 64-bit address arithmetic on the VALU the fp32 MFMAs share) instead of global_load_lds_dwordx4 with a 64-bit lane address;
//     ABL 4 (BUF only): every copy inside the loop issued with all lanes out of range -- the instruction without its memory traffic
// PAIR (round 5): two InnerProducts of one shape in ONE launch (mnc_fc_pair: fc6 + fc6_mask, fc7 + fc7_mask -- the box and the
// mask branch of a head stage, test.prototxt:584-627 / :652-696).  The column tiles of product 1 follow those of product 0
// (tn_ counts both; pair_tn = the tiles of one), so 2 x 32 tiles need 4 K ranges instead of 8 to fill the chip: a workgroup's
// K range is twice as long (prologue and epilogue paid once per 196 instead of 98 stages) and half as many partial sums are
// written and read back.
template <int kMT, int ABL = 0, int BUF = 0>
__global__ __launch_bounds__(512) void fc_mfma_dma16_kernel(const float* __restrict__ A_0, const float* __restrict__ Wt_0,
                                                            const float* __restrict__ bias_0, float* __restrict__ out_0,
                                                            float* __restrict__ part_0, int M, int N, int K, int ldc, int kper,
                                                            int act, int fused, int tn_, int splits_, int tm_, int drop_last,
                                                            const float* __restrict__ A_1,
                                                            const float* __restrict__ Wt_1, const float* __restrict__ bias_1,
                                                            float* __restrict__ out_1, int pair_tn) {
  constexpr int kBM = 32 * kMT;
  constexpr int kRows = kBM + kBN;
  constexpr int kNW = 8;
  constexpr int TS = kMT;                            // 16-row sub-tiles per wave (kMT / 2 row tiles x 2)
  constexpr int kPer = kRows * 128 / 1024 / kNW;     // DMA instructions per wave and stage
  static_assert(kRows % (8 * kNW) == 0 && kMT % 2 == 0, "whole pieces and row tiles per wave");
  extern __shared__ __attribute__((aligned(1024))) char s_fc_dma[];
  constexpr int kBufXor = 65536;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int r16 = lane & 15, g4 = lane >> 4;
  int bn, split, bmz;
  xcd_decode(blockIdx.x, tn_, splits_, tm_, bn, split, bmz);
  const int which = (pair_tn && bn >= pair_tn) ? 1 : 0;             // (block-uniform) which product of a pair this tile belongs to
  bn -= which * pair_tn;
  const float* __restrict__ A = which ? A_1 : A_0;
  const float* __restrict__ Wt = which ? Wt_1 : Wt_0;
  const float* __restrict__ bias = which ? bias_1 : bias_0;
  float* __restrict__ out = which ? out_1 : out_0;
  float* __restrict__ part = part_0;                  // K ranges' partial sums: slabs [tile (both products' column tiles)][range]
  const int n0 = bn * kBN, m0 = bmz * kBM;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const int nstages = (kend - kbeg) / 32;
  const int mrows = min(M - m0, kBM);

  const float* src[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int slot = (wave + kNW * i) * 64 + lane, r = slot >> 3, c = (slot & 7) ^ ((r >> 1) & 7);
    src[i] = (r < kBM ? A + (long)(m0 + min(r, mrows - 1)) * K
                      : Wt + (long)min(n0 + r - kBM, N - 1) * K) + kbeg + c * 4;
  }
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)s_fc_dma;
  // BUF: descriptors based at this workgroup's first row and K range (the lane offsets then stay far below 2 GB for any shape)
  typedef int i32x4d __attribute__((ext_vector_type(4)));
  auto make_rsrc = [](const float* base) {
    const unsigned long a = (unsigned long)base;
    i32x4d r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
    r.z = 0x70000000;
    r.w = 0x00020000;
    return r;
  };
  const i32x4d rs_a = make_rsrc(A + (long)m0 * K + kbeg), rs_w = make_rsrc(Wt + (long)min(n0, N - 1) * K + kbeg);
  int voff[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int slot = (wave + kNW * i) * 64 + lane, r = slot >> 3, c = (slot & 7) ^ ((r >> 1) & 7);
    voff[i] = (r < kBM ? min(r, mrows - 1) * K : (min(n0 + r - kBM, N - 1) - min(n0, N - 1)) * K) * 4 + c * 16;
  }
  auto dma_piece = [&](int i, long off, int buf_byte, bool in_loop = false) {
    const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf_byte + (unsigned)(wave + kNW * i) * 1024u);
    if (BUF) {
      const bool is_a = (wave + kNW * i) * 8 < kBM;                  // (a piece is eight whole rows: wave-uniform)
      const i32x4d rs = is_a ? rs_a : rs_w;
      const int so = __builtin_amdgcn_readfirstlane((int)off * 4);
      const int vo = ((ABL & 4) && in_loop) ? 0x7FFFFFF0 : voff[i];
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so), "s"(l) : "memory");
    } else {
      const float* g = src[i] + off;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l));
    }
  };
  auto dma_stage = [&](int s, int buf_byte) {
    const long off = (long)min(s, nstages - 1) * 32;
#pragma unroll
    for (int i = 0; i < kPer; ++i) dma_piece(i, off, buf_byte);
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc[TS][2];
#pragma unroll
  for (int i = 0; i < TS; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[i][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // byte offsets of this lane's chunk of K-group G in sub-tile 0 of the wave's rows / columns (+ 2048 per further sub-tile)
  int a_off[2], b_off[2];
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const int sl = ((4 * G + g4) ^ ((r16 >> 1) & 7)) * 16;
    a_off[G] = (wm * TS * 16 + r16) * 128 + sl;
    b_off[G] = (kBM + wn * 32 + r16) * 128 + sl;
  }
  const bool last_wave = drop_last && wm == 1;       // wave-uniform: this wave's last sub-tile holds no live row

  auto run = [&](auto last_tag) {
    constexpr int NS = decltype(last_tag)::value ? TS - 1 : TS;    // live sub-tiles
    constexpr int kNM = 8 * NS, kNR = NS + 2;                      // MFMAs / LDS reads per K-group
    struct Frags { f32x4v a[NS]; f32x4v b[2]; };
    auto read_frags = [&](int G, int flip, Frags& f) {
      f.b[0] = *reinterpret_cast<const f32x4v*>(s_fc_dma + (b_off[G] ^ flip));
      f.b[1] = *reinterpret_cast<const f32x4v*>(s_fc_dma + (b_off[G] ^ flip) + 2048);
#pragma unroll
      for (int i = 0; i < NS; ++i) f.a[i] = *reinterpret_cast<const f32x4v*>(s_fc_dma + (a_off[G] ^ flip) + i * 2048);
    };
    Frags f0, f1;
    dma_stage(0, 0);
    dma_stage(1, kBufXor);
    // only stage 0 has to have landed before the first MFMA (copies complete in issue order: kPer of this wave's may still be in
    // flight); stage 1 is waited for in front of the loop's first barrier, half a stage of MFMAs later.  All 256 workgroups pull
    // their first stages at once (29 MB): waiting for both cost every InnerProduct ~3 us of prologue (round 5).
    static_assert(kPer == 7, "the literal in the wait below");
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __syncthreads();
    read_frags(0, 0, f0);
    int cur = 0;
    for (int s = 0; s < nstages; ++s) {
      // stage s sits in buffer `cur` with its group-0 fragments in f0; stage s + 1 is landing in (or already in) the other buffer
      read_frags(1, 0, f1);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.a[i][q], f0.b[0][q], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.a[i][q], f0.b[1][q], acc[i][1], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < kNM; ++i) {                // one slot per MFMA; the fragment reads spread evenly over the first half
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if ((i + 1) * 2 * kNR / kNM > i * 2 * kNR / kNM && i * 2 < kNM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      dma_wait();                                    // stage s + 1 has landed ...
      if (!(ABL & 2)) __syncthreads();               // ... for every wave, and nobody reads buffer `cur` any more
      read_frags(0, kBufXor, f0);                    // group 0 of stage s + 1
      {                                              // group 1: the reads under the first MFMAs, then one copy every fourth MFMA
        const long off = (long)min(s + 2, nstages - 1) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.a[i][q], f1.b[0][q], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.a[i][q], f1.b[1][q], acc[i][1], 0, 0, 0);
            const int m = 2 * (q * NS + i);          // MFMAs issued before this pair
            if (m >= kNM - 4 * kPer - 8 && m < kNM - 8 && (m - (kNM - 4 * kPer - 8)) % 4 == 0) {
              __builtin_amdgcn_sched_barrier(0);
              if (!(ABL & 1)) dma_piece((m - (kNM - 4 * kPer - 8)) / 4, off, cur, true);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      }
#pragma unroll
      for (int G = 0; G < 2; ++G) { a_off[G] ^= kBufXor; b_off[G] ^= kBufXor; }
      cur ^= kBufXor;
    }
    dma_wait();                                      // the copies issued by the last two stages have landed before the LDS is released
    __syncthreads();
  };
  if (nstages > 0) {
    if (last_wave) run(std::true_type{});
    else run(std::false_type{});
  }

  const bool final_out = fused == 1;
  if (fused == 0) {
    // K ranges finished by the reduction launch (fc_reduce_slab_kernel): the accumulators as they sit in registers, [tile][range]
    // [wave][sub-tile i][column half c][lane] x 16 bytes -- one kilobyte per wave instruction (rounds 1-4 stored them row-major:
    // 80 four-byte stores per lane)
    constexpr int kSlabF4 = kNW * TS * 2 * 64;
    const int tile = bmz * tn_ + which * pair_tn + bn;
    f32x4v* slab = reinterpret_cast<f32x4v*>(part) + ((size_t)tile * splits_ + split) * kSlabF4 + wave * TS * 2 * 64 + lane;
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) slab[(i * 2 + c) * 64] = acc[i][c];
    return;
  }
  // D[row = 16 i + 4 (lane / 16) + reg][col = 16 c + lane % 16] of the wave's 160 x 32 outputs
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int n = n0 + wn * 32 + c * 16 + r16;
    if (n >= N) continue;
    const float bv = final_out ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * TS * 16 + i * 16 + 4 * g4 + e;
        if (m < M) {
          out[(long)m * ldc + n] = apply_act(acc[i][c][e] + bv, act);
        }
      }
  }
}

// The reduction behind fc_mfma_dma16_kernel's slabs (one or two products of a launch): one thread per 16-byte piece of a tile
// sums the K ranges in range order from zero (fc_reduce_kernel's additions: the same bits as rounds 1-4), adds the bias, applies
// the activation and writes the piece's four rows.  piece = (wave = 4 wm + wn, sub-tile i, column half c, lane = 16 g4 + r16):
// rows 160 wm + 16 i + 4 g4 + e, column 32 wn + 16 c + r16 of the tile.
__global__ __launch_bounds__(256) void fc_reduce_slab_kernel(const float* __restrict__ part, const float* __restrict__ bias0,
                                                             const float* __restrict__ bias1, float* __restrict__ out0,
                                                             float* __restrict__ out1, int M, int N, int ldc, int splits, int act,
                                                             int tn_all, int pair_tn, int ntiles) {
  constexpr int kSlabF4 = 8 * 10 * 2 * 64;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const f32x4v* p4 = reinterpret_cast<const f32x4v*>(part);
  const long total = (long)ntiles * kSlabF4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int tile = (int)(idx / kSlabF4), piece = (int)(idx - (long)tile * kSlabF4);
    const int lane = piece & 63, ic = (piece >> 6) % 20, wave = piece / (64 * 20);
    const int i = ic >> 1, c = ic & 1, wn = wave & 3, wm = wave >> 2, r16 = lane & 15, g4 = lane >> 4;
    const int bmz = tile / tn_all;
    int bn = tile - bmz * tn_all;
    const int which = (pair_tn && bn >= pair_tn) ? 1 : 0;
    bn -= which * pair_tn;
    const int n = bn * kBN + wn * 32 + c * 16 + r16;
    const int m0 = bmz * 320 + wm * 160 + i * 16 + 4 * g4;
    if (n >= N || m0 >= M) continue;
    const f32x4v* src = p4 + (long)tile * splits * kSlabF4 + piece;
    f32x4v v = {0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {
      const f32x4v a = src[(long)sp * kSlabF4], b = src[(long)(sp + 1) * kSlabF4], cc = src[(long)(sp + 2) * kSlabF4],
                   d = src[(long)(sp + 3) * kSlabF4];
      v = (((v + a) + b) + cc) + d;
    }
    for (; sp < splits; ++sp) v += src[(long)sp * kSlabF4];
    const float bv = (which ? bias1 : bias0)[n];
    float* o = (which ? out1 : out0) + (long)m0 * ldc + n;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (m0 + e < M) o[(long)e * ldc] = apply_act(v[e] + bv, act);
  }
}

// out = act(sum over splits (in split order) + bias).  VEC = 4: N % 4 == 0 and ldc % 4 == 0 -- one thread per four columns,
// 16-byte loads, four splits in flight; the per-element order of the additions is that of the scalar kernel.
// SM != 0 (VEC = 4 only): the result rows are written a second time in the stage-major 2-byte form the NEXT reduced-precision
// InnerProduct multiplies from (x3_split.h: sm_store4; sm_rows rows, this call's rows start at sm_row0) -- fc6 -> fc7 without
// a conversion pass.
// blockIdx.y == 1 (round 6: the reductions of a PAIR of products in one launch, mnc_fc_lowp_pair): the second set of pointers.
template <int VEC, int SM = 0>
__global__ __launch_bounds__(256) void fc_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                        float* __restrict__ out, int M, int N, int ldc, int splits,
                                                        int act, void* __restrict__ sm = nullptr, long sm_rows = 0,
                                                        long sm_row0 = 0, const float* __restrict__ part1 = nullptr,
                                                        const float* __restrict__ bias1 = nullptr, float* __restrict__ out1 = nullptr,
                                                        void* __restrict__ sm1 = nullptr) {
  if (blockIdx.y) { part = part1; bias = bias1; out = out1; sm = sm1; }
  const long total = (long)M * N;
  if (VEC == 4) {
    const int n4 = N >> 2;
    const long total4 = total >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
      const int n = (int)(i % n4) * 4;
      const long m = i / n4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      int s = 0;
      for (; s + 4 <= splits; s += 4) {
        const float4 a = p4[(long)s * total4 + i], b = p4[(long)(s + 1) * total4 + i], c = p4[(long)(s + 2) * total4 + i],
                     d = p4[(long)(s + 3) * total4 + i];
        v.x = (((v.x + a.x) + b.x) + c.x) + d.x;
        v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
        v.z = (((v.z + a.z) + b.z) + c.z) + d.z;
        v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
      }
      for (; s < splits; ++s) {
        const float4 a = p4[(long)s * total4 + i];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      const float4 y =
          make_float4(apply_act(v.x + b.x, act), apply_act(v.y + b.y, act), apply_act(v.z + b.z, act), apply_act(v.w + b.w, act));
      *reinterpret_cast<float4*>(out + m * ldc + n) = y;
      if (SM) sm_store4<SM>(sm, sm_rows, sm_row0 + m, n, y);          // lanes 2j / 2j+1 hold the halves of one 8-column group
    }
    return;
  }
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % N);
    const long m = idx / N;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[(long)s * total + idx];
    out[m * ldc + n] = apply_act(v + bias[n], act);
  }
}

void fc_reduce_launch(hipStream_t stream, const float* part, const float* bias, float* out, int M, int N, int ldc, int splits,
                      int act) {
  fc_reduce_launch_sm(stream, part, bias, out, M, N, ldc, splits, act, nullptr, 0, 0, 0);
}

// as above + the second output (sm_fmt 1 fp16 / 2 split bf16 / 3 bf16; needs N % 64 resp. % 32 == 0 and the vector path); returns false
// when the second output could not be written (the caller then converts the fp32 rows)
bool fc_reduce_launch_sm(hipStream_t stream, const float* part, const float* bias, float* out, int M, int N, int ldc, int splits,
                         int act, void* sm, int sm_fmt, long sm_rows, long sm_row0) {
  const bool vec = N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
  const long items = vec ? (long)M * N / 4 : (long)M * N;
  int g = (int)((items + 255) / 256);
  if (g > 4096) g = 4096;
  const bool sm_ok = sm && vec && (((sm_fmt == 1 || sm_fmt == 3) && N % 64 == 0) || (sm_fmt == 2 && N % 32 == 0));
  if (sm_ok && sm_fmt == 1)
    hipLaunchKernelGGL((fc_reduce_kernel<4, 1>), dim3(g), dim3(256), 0, stream, part, bias, out, M, N, ldc, splits, act, sm, sm_rows, sm_row0);
  else if (sm_ok && sm_fmt == 3)
    hipLaunchKernelGGL((fc_reduce_kernel<4, 3>), dim3(g), dim3(256), 0, stream, part, bias, out, M, N, ldc, splits, act, sm, sm_rows, sm_row0);
  else if (sm_ok)
    hipLaunchKernelGGL((fc_reduce_kernel<4, 2>), dim3(g), dim3(256), 0, stream, part, bias, out, M, N, ldc, splits, act, sm, sm_rows, sm_row0);
  else if (vec) hipLaunchKernelGGL((fc_reduce_kernel<4, 0>), dim3(g), dim3(256), 0, stream, part, bias, out, M, N, ldc, splits, act, nullptr, 0L, 0L);
  else hipLaunchKernelGGL((fc_reduce_kernel<1, 0>), dim3(g), dim3(256), 0, stream, part, bias, out, M, N, ldc, splits, act, nullptr, 0L, 0L);
  return sm_ok;
}

// The reductions of two products of one shape as ONE launch (grid.y = 2); false when the pair cannot share a launch (the caller then
// makes two fc_reduce_launch_sm calls): both need the vector path and the same second-output decision.
bool fc_reduce_pair_launch_sm(hipStream_t stream, const float* part0, const float* part1, const float* bias0, const float* bias1,
                              float* out0, float* out1, int M, int N, int ldc, int splits, int act, void* sm0, void* sm1, int sm_fmt,
                              long sm_rows, bool* sm_done) {
  const bool vec = N % 4 == 0 && ldc % 4 == 0 && ((reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1) |
                                                   reinterpret_cast<uintptr_t>(bias0) | reinterpret_cast<uintptr_t>(bias1)) & 15) == 0;
  if (!vec || (sm0 != nullptr) != (sm1 != nullptr)) return false;
  const long items = (long)M * N / 4;
  int g = (int)((items + 255) / 256);
  if (g > 4096) g = 4096;
  const bool sm_ok = sm0 && (((sm_fmt == 1 || sm_fmt == 3) && N % 64 == 0) || (sm_fmt == 2 && N % 32 == 0));
  if (sm_ok && sm_fmt == 1)
    hipLaunchKernelGGL((fc_reduce_kernel<4, 1>), dim3(g, 2), dim3(256), 0, stream, part0, bias0, out0, M, N, ldc, splits, act, sm0, sm_rows, 0L,
                       part1, bias1, out1, sm1);
  else if (sm_ok && sm_fmt == 3)
    hipLaunchKernelGGL((fc_reduce_kernel<4, 3>), dim3(g, 2), dim3(256), 0, stream, part0, bias0, out0, M, N, ldc, splits, act, sm0, sm_rows, 0L,
                       part1, bias1, out1, sm1);
  else if (sm_ok)
    hipLaunchKernelGGL((fc_reduce_kernel<4, 2>), dim3(g, 2), dim3(256), 0, stream, part0, bias0, out0, M, N, ldc, splits, act, sm0, sm_rows, 0L,
                       part1, bias1, out1, sm1);
  else
    hipLaunchKernelGGL((fc_reduce_kernel<4, 0>), dim3(g, 2), dim3(256), 0, stream, part0, bias0, out0, M, N, ldc, splits, act, (void*)nullptr, 0L,
                       0L, part1, bias1, out1, (void*)nullptr);
  *sm_done = sm_ok;
  return true;
}

__global__ void softmax_rows_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int M, int N) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* r = in + (long)m * ld_in;
  float mx = r[0];
  for (int i = 1; i < N; ++i) mx = fmaxf(mx, r[i]);
  float sum = 0.f;
  for (int i = 0; i < N; ++i) sum += expf(r[i] - mx);
  for (int i = 0; i < N; ++i) out[(long)m * N + i] = expf(r[i] - mx) / sum;
}

// N <= 64: one wavefront per row, one element per lane.  The exponentials are computed once, in parallel; the sum is
// taken in the SAME sequential order as the kernel above (every lane walks the row through shuffles), so both variants
// return identical bits.
__global__ __launch_bounds__(256) void softmax_rows_wave_kernel(const float* __restrict__ in, int ld_in,
                                                                float* __restrict__ out, int M, int N) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const bool live = lane < N;
  const float v = live ? in[(long)m * ld_in + lane] : -INFINITY;
  float mx = v;
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float e = live ? expf(v - mx) : 0.f;
  float sum = 0.f;
  for (int i = 0; i < N; ++i) sum += __shfl(e, i);
  if (live) out[(long)m * N + lane] = e / sum;
}

__global__ void eltwise_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = apply_act(in[i], op);
}

// [N][C][PH][PW] columns -> [N][PH][PW][C] columns
__global__ void pack_fc_kernel(const float* __restrict__ in, float* __restrict__ out, long N, int C, int P) {
  const long total = N * C * P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const int p = (int)(r % P);
    const long n = r / P;
    out[idx] = in[(n * C + c) * P + p];
  }
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_fc(mnc_ctx* ctx, const float* d_a, const float* d_w, const float* d_bias, float* d_out, int M, int N, int K,
           int ldc, int act) {
  MNC_REQUIRE(ctx && d_a && d_w && d_bias && d_out, "mnc_fc: null pointer");
  MNC_REQUIRE(M >= 0 && N > 0 && K > 0 && K % kBK == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc: unsupported shape M=%d N=%d K=%d ldc=%d act=%d (need K%%32==0)", M, N, K, ldc, act);
  if (M == 0) return MNC_OK;
  // Several 320-row blocks with a ragged tail (the CFM / ResNet configurations: 500-2000 RoIs per call): every block
  // multiplies all of its row tiles, so the full blocks and the tail are two launches, each with the tile height and split
  // count that suit it (M = 760: 2 x 320 + one 160-row block instead of 3 x 320).  Same stream: the second launch re-uses
  // the split-K scratch after the first one's reduction.  MNC_FC_NOTAIL=1 keeps one launch.
  if (M > 320 && M % 320 != 0 && M % 320 <= 160 && 2.0 * M * (double)N * K >= 2.0e9 && !tune(ctx, T_FC_NOTAIL, 0)) {
    const int head = M / 320 * 320;
    // (two launches cannot hand ONE set of partial sums to the caller: the deferred reduction is for single-launch products only --
    // both halves finish their own reduction, the caller sees deferred_splits == 1)
    const bool defer = ctx->defer_reduce;
    ctx->defer_reduce = false;
    int rc = mnc_fc(ctx, d_a, d_w, d_bias, d_out, head, N, K, ldc, act);
    if (!rc) rc = mnc_fc(ctx, d_a + (size_t)head * K, d_w, d_bias, d_out + (size_t)head * ldc, M - head, N, K, ldc, act);
    ctx->defer_reduce = defer;
    ctx->deferred_part = nullptr;
    ctx->deferred_splits = 1;
    return rc;
  }
  // small problems (< 2 GFLOP) use 64-row workgroups so that rows, column tiles and K splits together fill the chip; the
  // large ones the smallest of {160, 320} rows that covers M in one block (weights streamed once).  Measured at M = 300
  // (round 1): 320 rows x 32-deep stages, one workgroup per CU, and 160 rows x 16-deep stages, two per CU, are within 1 %
  // of each other on every FC of the heads (MNC_FC_TILE=5|10 overrides).
  const bool small = 2.0 * M * (double)N * K < 2.0e9;
  int mt = small ? 2 : (M <= 160 ? 5 : 10);            // row tiles per workgroup (all of them are always multiplied)
  // When the K splits of the 320-row variant would be shorter than 64 stages (fc7, fc6_maskest), 160-row blocks were a few per
  // cent faster than the REGISTER-STAGED 320-row kernel (round 1: 107 vs 112 us, 160 vs 169 us).  Against the LDS-DMA kernel
  // (K % 64 == 0) they lose: fc7 105.7 -> 100.7 us, fc6_maskest 159.6 -> 153.5 us on the 320-row DMA kernel (round 3,
  // kernel_bench fc), and the weights are streamed once instead of once per row block -- so the rule only applies without it.
  if (mt == 10 && (K % 64 != 0 || tune(ctx, T_FC_DMA, 1) == 0) && (K / 32) / cdiv(256, cdiv(N, kBN) * cdiv(M, 320)) < 64) mt = 5;
  if (tune_set(ctx, T_FC_TILE)) {
    const int v = tune(ctx, T_FC_TILE, 0);
    if (!small && (v == 5 || v == 10)) mt = v;
  }
  const int sk = mt == 5 ? 16 : 32;                    // K values per stage
  const int bm = 32 * mt;
  const int tn = cdiv(N, kBN), tm = cdiv(M, bm), stages = K / sk;
  // enough splits to give every CU its workgroups (two per CU except for the 320-row variant), but at least 64 K values
  // (8 stages for the large variants) per split
  int splits = cdiv(mt == 10 ? 256 : 512, tn * tm);
  // small variant: deep K (the N = 126 heads, K = 8192) gets at least 8 stages per split -- 32 splits instead of 103 cut
  // its reduction from 24 to 11 us and the total from 44 to 29 us; shallow K (mask_pred, K = 256) keeps 2
  const int min_stages = small ? (stages >= 64 ? 8 : 2) : (mt == 5 ? 16 : 8);
  if (splits > stages / min_stages) splits = stages / min_stages;
  // a small GEMM over at most 8 stages (mask_pred: K = 256) is not cut at all: four ranges of two stages each cost 9.5 us + a
  // 6.3 us reduction launch (kernel_bench fc, round 5) for 0.07 GFLOP; one range of eight stages writes the result itself
  if (small && stages <= 8) splits = 1;
  if (splits < 1) splits = 1;
  if (!small && tm == 1) splits = fc_split_div(ctx, splits, K);
  if (tm > 1 && !small)      // several row blocks: pick the split count by cost (see choose_splits); one block: as tuned above
    splits = choose_splits(tn * tm, stages, min_stages, mt == 10 ? 256 : 512,
                           (double)bm * kBN * sk * 2.0 / 460.0e3 * (mt == 10 ? 1.0 : 2.0), 4.0 * M * (double)N);
  int kper = cdiv(stages, splits) * sk;
  // LDS-DMA build of the 320-row kernel (fc_mfma_dma16_kernel; MNC_FC_DMA=0: the register-staged one).  Its loop walks one stage per
  // iteration, so a K range may hold an odd number of stages (rounds 3-5 rounded the ranges up to even counts, a leftover of the first
  // DMA build: fc6_maskest then ran 121 ranges of 26 stages on 242 CUs where 126 ranges of 25 fit 252 -- profiles/r06_fc_maskest.txt)
  const bool dma = mt == 10 && K % 64 == 0 && !tune_set(ctx, T_FC_ABL) && tune(ctx, T_FC_DMA, 1) != 0;
  if (dma && tune(ctx, T_FC_EVEN, 0)) kper = cdiv(kper, 64) * 64;
  splits = cdiv(K, kper);
  // (In-launch reduction of the K ranges by each tile's last arriver: built and measured in round 5 -- a loss here, the last arriver
  // reads 8 x 160 KB on one CU while 224 idle, profiles/r05_inlaunch_reduce.txt -- and removed in round 6.)
  // the eight-wave 16x16x4 kernel leaves its K ranges as SLABS (its accumulators in register layout, 160 KB per tile and range:
  // fc_reduce_slab_kernel); every other kernel as [range][M][N] rows (fc_reduce_kernel)
  bool slab = dma;
#ifdef MNC_TUNING
  if (dma && tune(ctx, T_FC_DMA_ABL, 0) < 16 &&
      (tune(ctx, T_FC_MFMA16, 1) == 0 || tune(ctx, T_FC_DMA_WAVES, 8) == 4 || tune(ctx, T_FC_DMA_ABL, 0)))
    slab = false;                                    // (the 32x32x2 builds of the tuning library)
#endif
  float* part = nullptr;
  if (splits > 1) {
    int rc = ensure_scratch(ctx, slab ? (size_t)tn * tm * splits * 163840 : (size_t)splits * M * N * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)N * K + (double)M * K + (double)M * N);
  {
    LaunchScope ls(ctx, small ? "fc_mfma_small" : "fc_mfma", flops, bytes);
#define MNC_FC_LAUNCH(MT, SK, A) hipLaunchKernelGGL((fc_mfma_kernel<MT, SK, A>), dim3(tn * splits * tm), dim3(256), 0, ctx->stream, \
                         d_a, d_w, d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, tm)
#define MNC_FC_DMA_LAUNCH(A, WM) MNC_FC_DMA_LAUNCH_H(A, WM, 0)
#define MNC_FC_DMA_LAUNCH_H(A, WM, H)                                                                                               \
  do {                                                                                                                              \
    static std::atomic<unsigned long long> attr_set{0};            /* one bit per device: function attributes are per device */   \
    const unsigned long long bit = 1ull << (ctx->device & 63);                                                                      \
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {                                                                        \
      MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_mfma_dma_kernel<10, A, WM, H>),                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                            \
      attr_set.fetch_or(bit, std::memory_order_relaxed);                                                                            \
    }                                                                                                                               \
    hipLaunchKernelGGL((fc_mfma_dma_kernel<10, A, WM, H>), dim3(tn * splits * tm), dim3(256 * WM), lds, ctx->stream, d_a, d_w, d_bias, \
                       d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, tm);                                  \
  } while (0)
    if (dma) {
      constexpr int lds = 65536 + (320 + kBN) * 32 * 4;
      // the product build: eight waves on 16 x 16 x 4 fragments (fc_mfma_dma16_kernel).  Tuning builds keep the 32 x 32 x 2 kernel it
      // replaced (FC_MFMA16=0; FC_DMA_WAVES=4: its four-wave form; FC_HALF=0: without the half tile; FC_DMA_ABL) for comparison:
      // fc6 562 -> 514 us, fc7 100 -> 94, fc6_maskest 153 -> 144 (kernel_bench fc --relu-input, same box).
      bool launched = false;
#ifdef MNC_TUNING
      const int waves = tune(ctx, T_FC_DMA_WAVES, 8), dabl = tune(ctx, T_FC_DMA_ABL, 0);
      if (dabl >= 16) {                              // ablations of the product kernel
        launched = true;
        const int drop = tm == 1 && M > 288 && M <= 304 ? 1 : 0;
        auto kern = dabl == 17 ? fc_mfma_dma16_kernel<10, 1> : dabl == 18 ? fc_mfma_dma16_kernel<10, 2> : dabl == 19 ? fc_mfma_dma16_kernel<10, 3>
                    : dabl == 20 ? fc_mfma_dma16_kernel<10, 0, 1> : dabl == 21 ? fc_mfma_dma16_kernel<10, 4, 1> : fc_mfma_dma16_kernel<10, 0>;
        MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kern, dim3(tn * splits * tm), dim3(512), lds, ctx->stream, d_a, d_w, d_bias, d_out, part, M, N, K, ldc, kper,
                           act, splits == 1 ? 1 : 0, tn, splits, tm, drop, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, 0);
      } else if (tune(ctx, T_FC_MFMA16, 1) == 0 || waves == 4 || dabl) {
        launched = true;
        if (dabl == 1) MNC_FC_DMA_LAUNCH(1, 1);      // 1: every copy re-reads stage 0 (L2-hot operands), 3: only the weight copies do
        else if (dabl == 3) MNC_FC_DMA_LAUNCH(3, 1);
        else if (waves == 4) MNC_FC_DMA_LAUNCH(0, 1);
        // one row block whose last row tile holds at most 16 rows (300 RoIs: 288 + 12): that tile on 16 x 16 x 4 MFMAs (FC_HALF=0: off)
        else if (tm == 1 && M > 288 && M <= 304 && tune(ctx, T_FC_HALF, 1) != 0) MNC_FC_DMA_LAUNCH_H(0, 2, 1);
        else MNC_FC_DMA_LAUNCH(0, 2);
      }
#endif
      if (!launched) {
        static std::atomic<unsigned long long> attr16{0};            // one bit per device: function attributes are per device
        const unsigned long long bit = 1ull << (ctx->device & 63);
        if (!(attr16.load(std::memory_order_relaxed) & bit)) {
          MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_mfma_dma16_kernel<10, 0, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
          MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_mfma_dma16_kernel<10, 0, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
          attr16.fetch_or(bit, std::memory_order_relaxed);
        }
        // one row block whose last 16-row sub-tile holds no live row (300 RoIs = 18 sub-tiles + 12 rows): its MFMAs are skipped
        const int drop = tm == 1 && M > 288 && M <= 304 ? 1 : 0;
        // copies through buffer descriptors (32-bit lane offsets from the workgroup's first row: 320 rows of K floats must stay
        // below the descriptor's range) -- same bytes to the same places, bit-identical results, 2-4 % faster than 64-bit lane
        // addresses (kernel_bench fc, MNC_FC_DMA_ABL=20 against 16: fc6 511 -> 500 us, fc7 92.9 -> 89.2, fc6_maskest 142 -> 137)
        const bool buf = 320.0 * (double)K * 4.0 < 1.8e9;
        hipLaunchKernelGGL((buf ? fc_mfma_dma16_kernel<10, 0, 1> : fc_mfma_dma16_kernel<10, 0, 0>), dim3(tn * splits * tm), dim3(512), lds,
                           ctx->stream, d_a, d_w, d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits,
                           tm, drop, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0);
      }
    }
    else if (mt == 2) MNC_FC_LAUNCH(2, 32, 0);
    else if (mt == 5) {
#ifdef MNC_TUNING
      const int abl = tune(ctx, T_FC_ABL, 0);
      if (abl == 1) MNC_FC_LAUNCH(5, 16, 1);
      else if (abl == 3) MNC_FC_LAUNCH(5, 16, 3);
      else
#endif
      MNC_FC_LAUNCH(5, 16, 0);
    } else {
#ifdef MNC_TUNING
      const int abl = tune(ctx, T_FC_ABL, 0);
      if (abl == 1) MNC_FC_LAUNCH(10, 32, 1);
      else if (abl == 2) MNC_FC_LAUNCH(10, 32, 2);
      else if (abl == 3) MNC_FC_LAUNCH(10, 32, 3);
      else if (abl == 4) MNC_FC_LAUNCH(10, 32, 4);
      else
#endif
      MNC_FC_LAUNCH(10, 32, 0);
    }
#undef MNC_FC_LAUNCH
#undef MNC_FC_DMA_LAUNCH
#undef MNC_FC_DMA_LAUNCH_H
    int rc = ls.finish("fc_mfma_kernel");
    if (rc) return rc;
  }
  if (ctx->defer_reduce && !slab) {                  // the caller's next kernel sums the ranges (mnc_internal.h; row layout only)
    ctx->deferred_part = splits > 1 ? part : nullptr;
    ctx->deferred_splits = splits > 1 ? splits : 1;
    return MNC_OK;
  }
  ctx->deferred_part = nullptr;
  ctx->deferred_splits = 1;
  if (splits > 1 && slab) {
    LaunchScope ls(ctx, "fc_reduce", 0.0, 4.0 * ((double)splits + 1.0) * M * N);
    long g = ((long)tn * tm * 10240 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(fc_reduce_slab_kernel, dim3((int)g), dim3(256), 0, ctx->stream, part, d_bias, (const float*)nullptr, d_out,
                       (float*)nullptr, M, N, ldc, splits, act, tn, 0, tn * tm);
    return ls.finish("fc_reduce_slab_kernel");
  }
  if (splits > 1) {
    LaunchScope ls(ctx, "fc_reduce", 0.0, 4.0 * ((double)splits + 1.0) * M * N);
    fc_reduce_launch(ctx->stream, part, d_bias, d_out, M, N, ldc, splits, act);
    return ls.finish("fc_reduce_kernel");
  }
  return MNC_OK;
}

// Two InnerProducts of one shape -- out_i[M][N] = act(a_i . w_i^T + b_i), i = 0, 1 -- as ONE launch of the eight-wave LDS-DMA
// kernel (+ one reduction launch) when that kernel would run each of them anyway (fp32, >= 2 GFLOP, 160 < M, K % 64 == 0, N and
// ldc multiples of 4); otherwise exactly two mnc_fc calls.  The paired launch cuts K into HALF as many ranges as mnc_fc would
// (twice the column tiles fill the chip), so its results differ from two mnc_fc calls in the last bits (other grouping of the
// partial sums); every executor of a graph must pair the same layers (engine.py: _plan_fusions; pipeline.hip: run_stage).
int mnc_fc_pair(mnc_ctx* ctx, const float* d_a0, const float* d_w0, const float* d_bias0, float* d_out0, const float* d_a1,
                const float* d_w1, const float* d_bias1, float* d_out1, int M, int N, int K, int ldc, int act) {
  MNC_REQUIRE(ctx && d_a0 && d_w0 && d_bias0 && d_out0 && d_a1 && d_w1 && d_bias1 && d_out1, "mnc_fc_pair: null pointer");
  MNC_REQUIRE(M >= 0 && N > 0 && K > 0 && K % kBK == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_pair: unsupported shape M=%d N=%d K=%d ldc=%d act=%d (need K%%32==0)", M, N, K, ldc, act);
  if (M == 0) return MNC_OK;
  // what mnc_fc would do with ONE of them: the 320-row LDS-DMA kernel in a single launch?  (small products run the 64-row kernel,
  // M <= 160 the 160-row one, several row blocks with a ragged tail two launches: none of these is paired)
  const bool big = 2.0 * M * (double)N * K >= 2.0e9;
  const bool ragged = M > 320 && M % 320 != 0 && M % 320 <= 160 && !tune(ctx, T_FC_NOTAIL, 0);
  const bool paired = big && M > 160 && !ragged;
  const bool vec = N % 4 == 0 && ldc % 4 == 0 && ((reinterpret_cast<uintptr_t>(d_out0) | reinterpret_cast<uintptr_t>(d_out1) |
                                                   reinterpret_cast<uintptr_t>(d_bias0) | reinterpret_cast<uintptr_t>(d_bias1)) & 15) == 0;
  bool fast = paired && vec && K % 64 == 0 && tune(ctx, T_FC_DMA, 1) != 0 && !tune_set(ctx, T_FC_ABL) && !tune_set(ctx, T_FC_TILE) &&
              320.0 * (double)K * 4.0 < 1.8e9 && !ctx->defer_reduce;
#ifdef MNC_TUNING
  if (tune(ctx, T_FC_MFMA16, 1) == 0 || tune(ctx, T_FC_DMA_WAVES, 8) == 4 || tune(ctx, T_FC_DMA_ABL, 0)) fast = false;
#endif
  if (!fast) {
    int rc = mnc_fc(ctx, d_a0, d_w0, d_bias0, d_out0, M, N, K, ldc, act);
    if (rc) return rc;
    return mnc_fc(ctx, d_a1, d_w1, d_bias1, d_out1, M, N, K, ldc, act);
  }
  const int tn = cdiv(N, kBN), tm = cdiv(M, 320), stages = K / 32;
  int splits = cdiv(256, 2 * tn * tm);
  if (splits > stages / 8) splits = stages / 8;
  if (splits < 1) splits = 1;
  if (tm > 1) splits = choose_splits(2 * tn * tm, stages, 8, 256, 320.0 * kBN * 32 * 2.0 / 460.0e3, 8.0 * M * (double)N);
  else splits = fc_split_div(ctx, splits, K);
  int kper = cdiv(cdiv(stages, splits) * 32, 64) * 64;
  splits = cdiv(K, kper);
  float* part = nullptr;
  if (splits > 1) {
    int rc = ensure_scratch(ctx, (size_t)2 * tn * tm * splits * 163840);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  constexpr int lds = 65536 + (320 + kBN) * 32 * 4;
  {
    const double flops = 4.0 * M * (double)N * K, bytes = 8.0 * ((double)N * K + (double)M * K + (double)M * N);
    LaunchScope ls(ctx, "fc_mfma", flops, bytes);
    static std::atomic<unsigned long long> attr{0};
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr.load(std::memory_order_relaxed) & bit)) {
      MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_mfma_dma16_kernel<10, 0, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr.fetch_or(bit, std::memory_order_relaxed);
    }
    const int drop = tm == 1 && M > 288 && M <= 304 ? 1 : 0;
    hipLaunchKernelGGL((fc_mfma_dma16_kernel<10, 0, 1>), dim3(2 * tn * splits * tm), dim3(512), lds, ctx->stream, d_a0, d_w0, d_bias0,
                       d_out0, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, 2 * tn, splits, tm, drop, d_a1, d_w1,
                       d_bias1, d_out1, tn);
    int rc = ls.finish("fc_mfma_dma16_kernel");
    if (rc) return rc;
  }
  if (splits > 1) {
    LaunchScope ls(ctx, "fc_reduce", 0.0, 8.0 * ((double)splits + 1.0) * M * N);
    long g = ((long)2 * tn * tm * 10240 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(fc_reduce_slab_kernel, dim3((int)g), dim3(256), 0, ctx->stream, part, d_bias0, d_bias1, d_out0, d_out1, M, N,
                       ldc, splits, act, 2 * tn, tn, 2 * tn * tm);
    return ls.finish("fc_reduce_slab_kernel");
  }
  return MNC_OK;
}

int mnc_softmax_rows_ld(mnc_ctx* ctx, const float* d_in, int ld_in, float* d_out, int M, int N) {
  MNC_REQUIRE(ctx && d_in && d_out && M >= 0 && N > 0 && ld_in >= N, "mnc_softmax_rows: bad argument");
  if (M == 0) return MNC_OK;
  LaunchScope ls(ctx, "softmax_rows");
  if (N <= 64)
    hipLaunchKernelGGL(softmax_rows_wave_kernel, dim3(cdiv(M, 4)), dim3(256), 0, ctx->stream, d_in, ld_in, d_out, M, N);
  else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(M, 64)), dim3(64), 0, ctx->stream, d_in, ld_in, d_out, M, N);
  return ls.finish("softmax_rows_kernel");
}

int mnc_softmax_rows(mnc_ctx* ctx, const float* d_in, float* d_out, int M, int N) {
  return mnc_softmax_rows_ld(ctx, d_in, N, d_out, M, N);
}

int mnc_eltwise(mnc_ctx* ctx, const float* d_in, float* d_out, size_t count, int op) {
  MNC_REQUIRE(ctx && (count == 0 || (d_in && d_out)) && (op == 1 || op == 2), "mnc_eltwise: bad argument");
  if (count == 0) return MNC_OK;
  LaunchScope ls(ctx, "eltwise");
  size_t g = (count + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(eltwise_kernel, dim3((int)g), dim3(256), 0, ctx->stream, d_in, d_out, count, op);
  return ls.finish("eltwise_kernel");
}

int mnc_copy2d(mnc_ctx* ctx, float* d_dst, int dst_ld, const float* d_src, int src_ld, int rows, int cols) {
  MNC_REQUIRE(ctx && rows >= 0 && cols >= 0 && dst_ld >= cols && src_ld >= cols, "mnc_copy2d: bad argument");
  if (rows == 0 || cols == 0) return MNC_OK;
  MNC_REQUIRE(d_dst && d_src, "mnc_copy2d: null pointer");
  MNC_HIP_TRY(hipMemcpy2DAsync(d_dst, (size_t)dst_ld * 4, d_src, (size_t)src_ld * 4, (size_t)cols * 4, rows,
                               hipMemcpyDeviceToDevice, ctx->stream));
  return MNC_OK;
}

int mnc_pack_fc_weights(mnc_ctx* ctx, const float* d_in, float* d_out, int N, int C, int PH, int PW) {
  MNC_REQUIRE(ctx && d_in && d_out && N > 0 && C > 0 && PH > 0 && PW > 0, "mnc_pack_fc_weights: bad argument");
  LaunchScope ls(ctx, "pack_fc");
  long total = (long)N * C * PH * PW;
  long g = (total + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(pack_fc_kernel, dim3((int)g), dim3(256), 0, ctx->stream, d_in, d_out, (long)N, C, PH * PW);
  return ls.finish("pack_fc_kernel");
}

}  // extern "C"
