"""Image -> network input (reference: lib/utils/blob.py:17-106).  cv2 is not required: the INTER_LINEAR resize of a
float32 image is implemented here (OpenCV convention: source = (dst + 0.5)/scale - 0.5, border-clamped)."""
import numpy as np


def _linear_taps(n_dst, n_src, scale):
    src = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (1.0 / scale) - 0.5).astype(np.float32)   # OpenCV: scale = 1/inv_scale
    lo = np.floor(src).astype(np.int64)
    frac = (src - lo).astype(np.float32)
    under, over = lo < 0, lo >= n_src - 1
    frac[under | over] = 0.0
    lo[under] = 0
    lo[over] = n_src - 1
    return lo, np.minimum(lo + 1, n_src - 1), frac


def resize_linear(im, fx, fy):
    """cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for float32 HxWxC."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    if fx == 1.0 and fy == 1.0:
        return im
    x0, x1, ax = _linear_taps(int(round(w * fx)), w, fx)
    y0, y1, ay = _linear_taps(int(round(h * fy)), h, fy)
    one = np.float32(1.0)
    rows = im[:, x0] * (one - ax)[None, :, None] + im[:, x1] * ax[None, :, None]
    return (rows[y0] * (one - ay)[:, None, None] + rows[y1] * ay[:, None, None]).astype(np.float32)


def resize_to(im, width, height):
    """cv2.resize(im, (width, height)) (INTER_LINEAR) for a float32 HxW or HxWxC array: the scale is dst/src per axis."""
    im = np.asarray(im, dtype=np.float32)
    squeeze = im.ndim == 2
    if squeeze:
        im = im[:, :, None]
    h, w = im.shape[:2]
    x0, x1, ax = _linear_taps(int(width), w, float(width) / w)
    y0, y1, ay = _linear_taps(int(height), h, float(height) / h)
    one = np.float32(1.0)
    rows = im[:, x0] * (one - ax)[None, :, None] + im[:, x1] * ax[None, :, None]
    out = (rows[y0] * (one - ay)[:, None, None] + rows[y1] * ay[:, None, None]).astype(np.float32)
    return out[:, :, 0] if squeeze else out


def im_list_to_blob(ims):
    """Zero-padded stack of prepared HxWx3 images -> float32 [N,3,H,W]."""
    hmax = max(im.shape[0] for im in ims)
    wmax = max(im.shape[1] for im in ims)
    blob = np.zeros((len(ims), 3, hmax, wmax), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, :, :im.shape[0], :im.shape[1]] = im.transpose(2, 0, 1)
    return blob


def prep_im_for_blob(im, pixel_means, target_size, max_size):
    """Subtract the BGR means (in float64, rounded once to float32 -- what `float32_array -= float64_array` does in
    the reference), then scale the short side to target_size unless that pushes the long side past max_size."""
    im = im.astype(np.float32, copy=True)
    im -= pixel_means
    short, long_ = min(im.shape[0:2]), max(im.shape[0:2])
    scale = float(target_size) / float(short)
    if np.round(scale * long_) > max_size:
        scale = float(max_size) / float(long_)
    return resize_linear(im, scale, scale), scale


def prep_im_for_blob_cfm(im, input_scales):
    """Image pyramid of the CFM test path (lib/utils/blob.py:53-85): one level per target short side in `input_scales`
    (long side capped at cfg.TEST.MAX_SIZE), zero-padded into one [L,3,H,W] blob -> (blob, scale factor per level)."""
    from mnc_config import cfg
    im_orig = im.astype(np.float32, copy=True)
    im_orig -= cfg.PIXEL_MEANS
    short, long_ = np.min(im_orig.shape[0:2]), np.max(im_orig.shape[0:2])
    ims, factors = [], []
    for target_size in input_scales:
        scale = float(target_size) / float(short)
        if np.round(scale * long_) > cfg.TEST.MAX_SIZE:
            scale = float(cfg.TEST.MAX_SIZE) / float(long_)
        ims.append(resize_linear(im_orig, scale, scale))
        factors.append(scale)
    return im_list_to_blob(ims), np.array(factors)


def pred_rois_for_blob(im_rois, im_scales):
    """Boxes -> net `rois` rows (level, x1, y1, x2, y2) in float64 (lib/utils/blob.py:88-106): each box goes to the pyramid
    level where its scaled area is closest to 224 x 224 and is scaled by that level's factor."""
    im_rois = im_rois.astype(np.float64, copy=False)
    if len(im_scales) > 1:
        widths = im_rois[:, 2] - im_rois[:, 0] + 1
        heights = im_rois[:, 3] - im_rois[:, 1] + 1
        scaled_areas = (widths * heights)[:, np.newaxis] * (im_scales[np.newaxis, :] ** 2)
        levels = np.abs(scaled_areas - 224 * 224).argmin(axis=1)[:, np.newaxis]
    else:
        levels = np.zeros((im_rois.shape[0], 1), dtype=np.int64)
    return np.hstack((levels.astype(np.float64), im_rois * im_scales[levels]))
