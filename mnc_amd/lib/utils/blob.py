"""Image -> network input (reference: lib/utils/blob.py:17-106).  With a net of this package and a uint8 image the resize
runs on the GPU (`*_device` variants below, mnc_amd/prep.py); the numpy functions are the reference-shaped API and the
definition of the result.  cv2 is not required: the INTER_LINEAR resize of a
float32 image is implemented here (OpenCV convention: source = (dst + 0.5)/scale - 0.5, border-clamped)."""
import numpy as np


from mnc_amd.prep import linear_taps as _linear_taps      # one tap function for the numpy path and the device path


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


def cfm_scale_factors(im_shape, input_scales):
    """Scale factor of every pyramid level (lib/utils/blob.py:70-79): target short side, long side capped at TEST.MAX_SIZE."""
    from mnc_config import cfg
    short, long_ = np.min(im_shape[0:2]), np.max(im_shape[0:2])
    factors = []
    for target_size in input_scales:
        scale = float(target_size) / float(short)
        if np.round(scale * long_) > cfg.TEST.MAX_SIZE:
            scale = float(cfg.TEST.MAX_SIZE) / float(long_)
        factors.append(scale)
    return np.array(factors)


def prep_im_for_blob_cfm(im, input_scales):
    """Image pyramid of the CFM test path (lib/utils/blob.py:53-85): one level per target short side in `input_scales`
    (long side capped at cfg.TEST.MAX_SIZE), zero-padded into one [L,3,H,W] blob -> (blob, scale factor per level)."""
    from mnc_config import cfg
    im_orig = im.astype(np.float32, copy=True)
    im_orig -= cfg.PIXEL_MEANS
    factors = cfm_scale_factors(im_orig.shape, input_scales)
    return im_list_to_blob([resize_linear(im_orig, f, f) for f in factors]), factors


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


def prep_im_for_blob_device(net, im, pixel_means, target_size, max_size):
    """prep_im_for_blob + im_list_to_blob on the GPU of `net`: -> (DeviceArray [1,3,H',W'], scale).  Same values as
    im_list_to_blob([prep_im_for_blob(im, ...)[0]]) (bit-identical; tests/test_gpu_ops.py)."""
    short, long_ = min(im.shape[0:2]), max(im.shape[0:2])
    scale = float(target_size) / float(short)
    if np.round(scale * long_) > max_size:
        scale = float(max_size) / float(long_)
    return net.prep_image(im, pixel_means, [scale]), scale


def prep_im_for_blob_cfm_device(net, im, input_scales):
    """prep_im_for_blob_cfm on the GPU of `net`: -> (DeviceArray [L,3,H',W'], scale factors)."""
    from mnc_config import cfg
    factors = cfm_scale_factors(im.shape, input_scales)
    return net.prep_image(im, cfg.PIXEL_MEANS, list(factors)), factors


def can_prep_on_device(net, im):
    from mnc_config import cfg
    return (cfg.TEST.get("DEVICE_PREP", True) and hasattr(net, "prep_image") and isinstance(im, np.ndarray)
            and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3)
