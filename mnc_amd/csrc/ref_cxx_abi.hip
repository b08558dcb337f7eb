// The reference's link-level binding of nms.gpu_nms / nms.mv.
//
// /root/reference/lib/nms/gpu_nms.pyx:13-14 and gpu_mv.pyx:7-8 declare `_nms` / `_mv` with `cdef extern from "gpu_nms.hpp"`
// resp. "gpu_mv.hpp", and lib/setup.py:126-130, 143-147 build both extensions with language='c++' against nms_kernel.cu /
// mv_kernel.cu compiled by nvcc as C++.  The symbols the extension modules import are therefore the Itanium-mangled
//     _Z4_nmsPiS_PKfiifi                       void _nms(int*, int*, float const*, int, int, float, int)
//     _Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii        void _mv(float const*, float const*, int, int const*, int const*, float const*,
//                                                       int, int, int, int, int, int, float*, int*, int)
// not the unmangled names.  This translation unit defines exactly those two functions (C++ linkage, default visibility) on top
// of the status-returning C entry points, so that the reference's gpu_nms.pyx / gpu_mv.pyx / gpu_nms.hpp / gpu_mv.hpp build
// UNCHANGED with `libraries=['mnc_hip']` in place of the two .cu sources.  The C-linkage `_nms` / `_mv`
// (nms.hip, mv.hip) stay for dlsym-style callers; a C and a C++ function of one name cannot be declared in one translation
// unit, hence the separate file and the MNC_HIP_REF_CXX_NAMES switch of the header.
//
// tests/test_abi_cpu.py compiles a C++ program against the reference's headers (verbatim copies of their two declarations when
// /root/reference is not mounted) and links it to this library; tests/test_gpu_nms_mv.py runs it against the fixtures.
#define MNC_HIP_REF_CXX_NAMES 1
#include <cstdio>

#include "../../include/mnc_hip.h"

void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id) {
  if (mnc_nms(keep_out, num_out, boxes_host, boxes_num, boxes_dim, nms_overlap_thresh, device_id) != MNC_OK) {
    fprintf(stderr, "mnc_hip: _nms failed: %s\n", mnc_last_error());
    if (num_out) *num_out = 0;
  }
}

void _mv(const float* all_boxes, const float* all_masks, const int all_boxes_num, const int* candidate_inds,
         const int* candidate_start, const float* candidate_weights, const int candidate_num, const int image_height,
         const int image_width, const int box_dim, const int mask_size, const int result_num, float* finalize_output_mask,
         int* finalize_output_box, const int device_id) {
  if (mnc_mv(all_boxes, all_masks, all_boxes_num, candidate_inds, candidate_start, candidate_weights, candidate_num,
             image_height, image_width, box_dim, mask_size, result_num, finalize_output_mask, finalize_output_box,
             device_id) != MNC_OK)
    fprintf(stderr, "mnc_hip: _mv failed: %s\n", mnc_last_error());
}
