"""CPU: the N > 1 path (image sharding + fixed-shape instance all-gather) with world_size 2 over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mnc_amd import dist as mdist


def test_pack_unpack_roundtrip_and_cap():
    rng = np.random.default_rng(0)
    lb = [np.hstack([rng.uniform(0, 99, (7, 4)), rng.uniform(0, 1, (7, 1))]) for _ in range(20)]     # 140 > cap
    lm = [rng.uniform(0, 1, (7, 1, 21, 21)).astype(np.float32) for _ in range(20)]
    with pytest.warns(UserWarning, match="40 dropped"):       # truncation (ties at the voting threshold) is never silent
        rec, n = mdist.pack_instances(lm, lb)
    assert rec.shape == (100, 447) and n == 100
    boxes, classes, masks = mdist.unpack_instances(rec)
    assert boxes.shape == (100, 5) and classes[0] == 1 and classes[-1] == 15 and masks.shape == (100, 1, 21, 21)
    assert np.array_equal(masks[:7], lm[0])
    rec0, n0 = mdist.pack_instances([np.zeros((0, 1, 21, 21), np.float32)] * 20, [np.zeros((0, 5))] * 20)
    assert n0 == 0 and mdist.unpack_instances(rec0)[0].shape == (0, 5)
    assert mdist.shard_indices(8, 3, 8) == [3] and mdist.shard_indices(10, 1, 4) == [1, 5, 9]


def test_world_size_2_gloo():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(here, "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_OK" in r.stdout


def test_world_size_8_gloo():
    """De-risking the first 8-rank run: eight gloo ranks on the CPU -- sharding 19 images, the per-step gather
    with a ragged tail, and the life cycle of the weight container bench.py shares between the ranks of a node (LOCAL_RANK 0
    writes it under an unpredictable name, all map it, nothing is left behind)."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29583", os.path.join(here, "dist_worker8.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST8_OK" in r.stdout
