// Host build of mnc_amd/csrc/np_exp.h for tests/test_np_exp.py (g++ -O2 -mfma -ffp-contract=off -shared -fPIC).
#include "../mnc_amd/csrc/np_exp.h"
extern "C" void np_exp_f32_array(const float* in, float* out, long n) {
  for (long i = 0; i < n; ++i) out[i] = mnc::np_exp_f32(in[i]);
}
// bbox_transform_inv of one (box, delta) pair in the device kernels' float32 operation order (csrc/proposal.hip), host build
extern "C" void decode_box(const float* box, const float* d, float* out) {
  const float widths = box[2] - box[0] + 1.0f, heights = box[3] - box[1] + 1.0f;
  const float ctr_x = box[0] + 0.5f * widths, ctr_y = box[1] + 0.5f * heights;
  const float pcx = d[0] * widths + ctr_x, pcy = d[1] * heights + ctr_y;
  const float pw = mnc::np_exp_f32(d[2]) * widths, ph = mnc::np_exp_f32(d[3]) * heights;
  out[0] = pcx - 0.5f * pw; out[1] = pcy - 0.5f * ph; out[2] = pcx + 0.5f * pw; out[3] = pcy + 0.5f * ph;
}
