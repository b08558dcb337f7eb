"""Image file reading without OpenCV (the reference uses cv2.imread, tools/demo.py:142, TesterWrapper.py:166)."""
import numpy as np


def imread(path):
    """uint8 HxWx3 in BGR channel order, like cv2.imread(path) (colour images; EXIF orientation ignored as in OpenCV 2/3).
    `.npy` files (HxWx3 uint8, already BGR) are read directly -- handy for synthetic test sets."""
    if str(path).endswith(".npy"):
        im = np.load(path)
        if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8:
            raise ValueError("%s: expected a uint8 HxWx3 array" % path)
        return im
    from PIL import Image
    with Image.open(path) as img:
        rgb = np.asarray(img.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])
