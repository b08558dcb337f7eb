// Shared by the Winograd F(2x2, 3x3) kernels (conv_wino.hip, conv_wino_stream.hip): LDS geometry of a channel block and the
// packed weight panel (pack_conv3x3_wino_kernel, conv_wino.hip).
#pragma once

#include "mnc_internal.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWCols = 32;                 // pixel columns per workgroup (16 Winograd tiles)
constexpr int kWHaloCols = kWCols + 2;
constexpr int kWPixPitch = 12;             // floats per halo pixel in LDS (8 channels + 4 pad)
constexpr int kWRowPitch = 68;             // floats per (k half, output channel) weight row: 16 positions x 4 channels + 4 pad
constexpr int kWPanel = 2 * 32 * kWRowPitch;   // floats per (channel block, 32-channel tile) weight panel = 4352

// conv_wino_stream.hip (-DMNC_TUNING builds only): the layer as one stream of (tile, channel block) units cut into equal ranges, one per workgroup.
// wgs_per_slot: workgroups per resident slot (512 slots: 256 CUs x two workgroups); returns MNC_ERR_INVALID if the shape does not
// qualify (the caller then runs the tiled kernel).
int wino_stream_launch(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                       int Cin, int Cout, int relu, int pool, int wgs_per_slot, int xcd_order);

// conv_wino16.hip (-DMNC_TUNING builds only): conv3x3_wino2_kernel<2, 7, 1, *> re-tiled onto v_mfma_f32_16x16x4_f32 fragments (same arguments as launch_wino2)
int wino16_launch(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                  int Cout, int relu, int ksplit, float* part, int pool, int pix_a, int ksplit_b, int xcd_order);

}  // namespace mnc
