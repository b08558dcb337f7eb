"""`nms.cpu_nms.cpu_nms` -- the import the reference's nms_wrapper expects (lib/nms/nms_wrapper.py:9, :18-21).

This package has no CPU compute path: with cfg.USE_GPU_NMS = False the reference would switch to its Cython NMS (which is
not even equivalent to the GPU kernel: `ovr >= thresh` in lib/nms/cpu_nms.pyx:65 against a strict `>` in
nms_kernel.cu:71).  Here the switch fails loudly instead of silently computing on the host."""


def cpu_nms(dets, thresh):
    raise NotImplementedError("cfg.USE_GPU_NMS = False selects the reference's CPU NMS; this package computes on the MI355X "
                              "only (nms.gpu_nms.gpu_nms)")
