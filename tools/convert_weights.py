#!/usr/bin/env python3
"""Convert a Caffe weight file (.caffemodel protobuf or .caffemodel.h5 HDF5, as written by the reference's
SolverWrapper.snapshot, lib/caffeWrapper/SolverWrapper.py:100-114) to the .npz container, or just list it.

    python tools/convert_weights.py mnc_model.caffemodel.h5 mnc_model.npz
    python tools/convert_weights.py --list mnc_model.caffemodel
    python tools/convert_weights.py mnc_model.caffemodel.h5 mnc_model.mncw     (flat file for mnc_net_load_file / C hosts)

`caffe.Net(prototxt, path, caffe.TEST)` of this package reads all three containers directly; the conversion is only a
convenience (an .npz loads faster and is easy to inspect).  Shared parameters (soft links in the HDF5 container) come out as
one copy per layer name; the engine de-duplicates them on the device by `param { name }`."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnc_amd import caffemodel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst", nargs="?")
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    w = caffemodel.load_weights(args.src)
    total = 0
    for name, blobs in w.items():
        shapes = ["x".join(map(str, b.shape)) if b is not None else "-" for b in blobs]
        total += sum(b.size for b in blobs if b is not None)
        if args.list or not args.dst:
            print("%-28s %s" % (name, "  ".join(shapes)))
    print("%d layers, %.1f M parameters" % (len(w), total / 1e6))
    if args.dst:
        if args.dst.endswith(".mncw"):                 # flat container for mnc_net_load_file (non-Python hosts)
            caffemodel.save_flat(w, args.dst)
        else:
            caffemodel.save_npz(w, args.dst)
        print("wrote", args.dst)


if __name__ == "__main__":
    main()
