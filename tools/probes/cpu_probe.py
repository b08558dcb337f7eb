import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mnc_amd import models, synth
from oracle import host as ohost, native, net as onet
p = models.write_mnc_5stage_test_prototxt(); w = synth.synthetic_weights(p, 0)
im = np.random.default_rng(0).integers(0, 256, (600, 1000, 3), dtype=np.uint8)
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    onet.im_detect(w, im)
    t = time.perf_counter(); b, m, s = onet.im_detect(w, im); t1 = time.perf_counter() - t
    print("threads", nt, "net+host %.2fs" % t1, flush=True)
t = time.perf_counter(); ohost.gpu_mask_voting(m, b, s, 21, 100, 1000, 600); print("voting oracle C %.2fs" % (time.perf_counter() - t))
t = time.perf_counter(); ohost.gpu_mask_voting(m, b, s, 21, 100, 1000, 600, nms_fn=native.ref_gpu_nms, mv_fn=native.ref_mv); print("voting ref .cu-on-cpu %.2fs" % (time.perf_counter() - t))
