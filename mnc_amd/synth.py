"""Seeded synthetic weights for MNC graphs (there is no network here to fetch mnc_model.caffemodel.h5,
data/scripts/fetch_mnc_model.sh:6-13).  He-normal conv/FC (std = sqrt(2/fan_in)) keeps activations O(1) through the
13-layer trunk; the two box regressors get the small stds Caffe's fillers use for them; biases are zero.
Layers sharing parameters by `param { name }` (test.prototxt:514-515 <-> :829-834) get ONE entry, under the first
layer that declares the name -- the same convention as a Caffe HDF5 snapshot converted to .npz."""
import numpy as np

from . import prototxt

SMALL_STD = {"bbox_pred": 0.001, "rpn_bbox_pred": 0.01}


def layer_shapes(prototxt_path):
    """-> OrderedDict-like list of (layer name, kind, weight shape, bias shape) for every parameterised layer that
    owns its parameters, by tracing channel/geometry through the graph (batch and RoI counts are irrelevant)."""
    net = prototxt.parse_file(prototxt_path)
    geo = {}                                    # blob -> (C,) or (C, PH, PW)
    for name, shp in zip(net.all("input"), net.all("input_shape")):
        d = shp.all("dim")
        geo[name] = (d[1],) if len(d) == 4 else (d[-1],)
    owners, out = set(), []
    for L in net.all("layer"):
        typ, name = L.get1("type"), L.get1("name")
        bots, tops = L.all("bottom"), L.all("top")
        pnames = tuple(p.get1("name") for p in L.all("param"))
        shared = bool(pnames) and all(pnames) and pnames in owners
        if typ == "Convolution":
            cp = L.get1("convolution_param")
            cin, cout, k = geo[bots[0]][0], cp.get1("num_output"), cp.get1("kernel_size")
            geo[tops[0]] = (cout,)
            if not shared:
                out.append((name, "conv", (cout, cin, k, k), (cout,) if cp.get1("bias_term", True) else None))
        elif typ in ("BatchNorm", "Scale"):
            c = geo[bots[0]][0]
            geo[tops[0]] = geo[bots[0]]
            out.append((name, "bn" if typ == "BatchNorm" else "scale", (c,), (c,)))
        elif typ == "Eltwise":
            geo[tops[0]] = geo[bots[0]]
        elif typ == "InnerProduct":
            n = L.get1("inner_product_param").get1("num_output")
            k = int(np.prod(geo[bots[0]]))
            geo[tops[0]] = (n,)
            if not shared:
                out.append((name, "fc", (n, k), (n,)))
        elif typ == "Pooling":
            g = geo[bots[0]]
            geo[tops[0]] = g if len(g) == 1 else (g[0], g[1] // 2, g[2] // 2)
        elif typ == "ROIWarping":
            rp = L.get1("roi_warping_param")
            geo[tops[0]] = (geo[bots[0]][0], rp.get1("pooled_h"), rp.get1("pooled_w"))
        elif typ == "ROIPooling":
            rp = L.get1("roi_pooling_param")
            geo[tops[0]] = (geo[bots[0]][0], rp.get1("pooled_h"), rp.get1("pooled_w"))
        elif typ == "MaskResize":
            mp = L.get1("mask_resize_param")
            geo[tops[0]] = (1, mp.get1("output_height"), mp.get1("output_width"))
        elif typ == "MaskPooling":
            geo[tops[0]] = geo[bots[0]]
        elif typ == "Concat":
            geo[tops[0]] = (sum(geo[b][0] for b in bots),)
        elif typ == "Python":
            lay = L.get1("python_param").get1("layer")
            geo[tops[0]] = (1, 21, 21) if lay == "MaskLayer" else (5,)
        else:                                   # ReLU, Softmax, Sigmoid, Reshape: geometry-preserving here
            for t in tops:
                geo[t] = geo[bots[0]]
        if pnames and all(pnames):
            owners.add(pnames)
    return out


PIXEL_STD = 73.6      # std of uniform{0..255} pixels: folded into conv1_1 so that every later activation is O(1)
_CHUNK = 1 << 23


def _fill(flat, std, seed, layer_idx, pool):
    """flat[:] ~ N(0, std^2), generated in independent fixed-size chunks (deterministic for a given seed regardless of
    the number of worker threads; numpy releases the GIL inside standard_normal)."""
    n = flat.shape[0]

    def work(ci):
        lo, hi = ci * _CHUNK, min(n, (ci + 1) * _CHUNK)
        g = np.random.default_rng(np.random.SeedSequence(entropy=seed, spawn_key=(layer_idx, ci)))
        g.standard_normal(out=flat[lo:hi], dtype=np.float32)
        flat[lo:hi] *= np.float32(std)

    list(pool.map(work, range((n + _CHUNK - 1) // _CHUNK)))


def synthetic_weights(prototxt_path, seed=0):
    """{layer: [W float32, b float32]} in Caffe layout (conv OIHW, fc [N][K] with (c,h,w) column order); a convolution with
    `bias_term: false` gets [W]; BatchNorm gets Caffe's three blobs [mean, variance, moving-average factor] and Scale
    [gamma, beta], with statistics that keep the activations O(1) through a residual trunk."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    weights = {}
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
        for idx, (name, kind, wshape, bshape) in enumerate(layer_shapes(prototxt_path)):
            if kind in ("bn", "scale"):
                g = np.random.default_rng(np.random.SeedSequence(entropy=seed, spawn_key=(idx, 0)))
                c = wshape[0]
                if kind == "bn":          # stored un-normalised, as Caffe does: statistic * factor
                    factor = np.float32(2.0)
                    weights[name] = [(g.normal(0, 0.2, c) * factor).astype(np.float32),
                                     (g.uniform(0.6, 1.6, c) * factor).astype(np.float32), np.array([factor], np.float32)]
                else:
                    weights[name] = [g.uniform(0.4, 0.9, c).astype(np.float32), g.normal(0, 0.1, c).astype(np.float32)]
                continue
            fan_in = int(np.prod(wshape[1:]))
            std = SMALL_STD.get(name, float(np.sqrt(2.0 / fan_in)))
            if idx == 0:
                std /= PIXEL_STD
            w = np.empty(wshape, dtype=np.float32)
            _fill(w.reshape(-1), std, seed, idx, pool)
            weights[name] = [w, np.zeros(bshape, dtype=np.float32)] if bshape is not None else [w]
    return weights
