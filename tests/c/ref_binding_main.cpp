// What the reference's Cython extensions do at link level, without Cython: a C++ translation unit that sees `_nms` / `_mv`
// ONLY through the reference's own headers (lib/nms/gpu_nms.hpp, gpu_mv.hpp -- included from /root/reference when the test
// passes -DMNC_REF_HEADERS, otherwise through the two declarations below, copied character for character from those
// headers: they are the interface under test) and is linked with -lmnc_hip.  It resolves to the C++-mangled exports
// _Z4_nmsPiS_PKfiifi / _Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii, as gpu_nms.pyx:13-14 / gpu_mv.pyx:7-8 built with language='c++'
// (lib/setup.py:126-147) do.
//
//   ref_binding_main nms <dets.f32> <n> <thresh> <out.i32>            out = [num, keep[0..num)]
//   ref_binding_main mv  <dir> <N> <C> <R> <H> <W> <S>                reads boxes.f32 masks.f32 inds.i32 start.i32 wts.f32 from
//                                                                     <dir>, writes out_mask.f32 / out_box.i32 there
//   ref_binding_main link                                             prints the two function addresses (CPU link check)
#ifdef MNC_REF_HEADERS
#include "gpu_nms.hpp"
#include "gpu_mv.hpp"
#else
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
          int boxes_dim, float nms_overlap_thresh, int device_id);
void _mv(const float* all_boxes, const float* all_masks, const int all_boxes_num,
        const int* candidate_inds, const int* candidate_start, const float* candidate_weights, const int candidate_num,
        const int image_height, const int image_width, const int box_dim, const int mask_size, const int result_num,
        float* finalize_output_mask, int* finalize_output_box, const int device_id);
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

template <typename T>
static std::vector<T> slurp(const std::string& path, size_t count) {
  std::vector<T> v(count);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(v.data(), sizeof(T), count, f) != count) {
    fprintf(stderr, "cannot read %zu items from %s\n", count, path.c_str());
    exit(2);
  }
  fclose(f);
  return v;
}
template <typename T>
static void spit(const std::string& path, const T* p, size_t count) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, sizeof(T), count, f) != count) exit(3);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "link")) {
    printf("_nms %p _mv %p\n", (void*)&_nms, (void*)&_mv);
    return 0;
  }
  if (argc == 6 && !strcmp(argv[1], "nms")) {
    const int n = atoi(argv[3]);
    std::vector<float> dets = slurp<float>(argv[2], (size_t)n * 5);
    std::vector<int> out(n + 1, 0);
    _nms(out.data() + 1, out.data(), dets.data(), n, 5, (float)atof(argv[4]), 0);
    spit(argv[5], out.data(), (size_t)out[0] + 1);
    return 0;
  }
  if (argc == 9 && !strcmp(argv[1], "mv")) {
    const std::string d = argv[2];
    const int N = atoi(argv[3]), C = atoi(argv[4]), R = atoi(argv[5]), H = atoi(argv[6]), W = atoi(argv[7]), S = atoi(argv[8]);
    std::vector<float> boxes = slurp<float>(d + "/boxes.f32", (size_t)N * 4), masks = slurp<float>(d + "/masks.f32", (size_t)N * S * S),
                       wts = slurp<float>(d + "/wts.f32", C);
    std::vector<int> inds = slurp<int>(d + "/inds.i32", C), start = slurp<int>(d + "/start.i32", R);
    std::vector<float> om((size_t)R * S * S);
    std::vector<int> ob((size_t)R * 4);
    _mv(boxes.data(), masks.data(), N, inds.data(), start.data(), wts.data(), C, H, W, 4, S, R, om.data(), ob.data(), 0);
    spit(d + "/out_mask.f32", om.data(), om.size());
    spit(d + "/out_box.i32", ob.data(), ob.size());
    return 0;
  }
  fprintf(stderr, "usage: see the head of tests/c/ref_binding_main.cpp\n");
  return 1;
}
