"""Global configuration of the MNC inference path -- same surface as the reference's `mnc_config`
(lib/mnc_config.py:8-206): a mutable attribute-dict `cfg` plus `cfg_from_file(yaml)`; `easydict` is not required.
Every key of the reference is declared (so that its experiment yml files merge); only the TEST keys and a handful of TRAIN
keys are read by the inference path (demo.py:59 passes cfg.TRAIN.MAX_SIZE to prep_im_for_blob)."""
import os

import numpy as np


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    __setattr__ = __setitem__


cfg = AttrDict()
cfg.MNC_MODE = True
cfg.CFM_MODE = False
cfg.EXP_DIR = "default"
cfg.USE_GPU_NMS = True                      # mnc_config.py:16
cfg.GPU_ID = 0                              # mnc_config.py:17
cfg.RNG_SEED = 3
cfg.EPS = 1e-14
cfg.PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])   # BGR, mnc_config.py:20
cfg.ROOT_DIR = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
cfg.DATA_DIR = os.path.join(cfg.ROOT_DIR, "data")
cfg.BINARIZE_THRESH = 0.4                   # mnc_config.py:26
cfg.MASK_SIZE = 21                          # mnc_config.py:28
# Not in the reference: which convention the three Caffe layers whose source is unavailable (ROIWarping, MaskResize, MaskPooling)
# follow -- fields of `mnc_layer_conventions` (include/mnc_hip.h), e.g. LAYER_CONVENTIONS: {warp_sample: 2, resize_mode: 1} in an
# experiment .yml.  Empty = oracle/SPEC.md (PARITY UNPINNED either way; SPEC.md section 6 lists the alternatives).
cfg.LAYER_CONVENTIONS = AttrDict()

# TRAIN: no training code exists in this package, but the reference's experiment files (experiments/cfgs/VGG16/*.yml) set
# TRAIN keys next to the TEST ones and cfg_from_file rejects unknown keys (mnc_config.py:174-176) -- so every key of
# mnc_config.py:31-108 is declared, with the reference's default.  Test-time readers: TRAIN.MAX_SIZE (demo.py:59,
# TesterWrapper.py:271), TRAIN.MIX_INDEX, TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED.
cfg.TRAIN = AttrDict(
    IMS_PER_BATCH=1, BATCH_SIZE=64, ASPECT_GROUPING=True, USE_FLIPPED=True, SCALES=(600,), MAX_SIZE=1000,
    SNAPSHOT_ITERS=5000, SNAPSHOT_INFIX='',
    FG_FRACTION=[0.3], FG_THRESH_HI=[1.0], FG_THRESH_LO=[0.5], BG_FRACTION=[0.85, 0.15], BG_THRESH_HI=[0.5, 0.1],
    BG_THRESH_LO=[0.1, 0.0], PROPOSAL_METHOD='gt',
    BBOX_REG=True, BBOX_NORMALIZE_TARGETS=True, BBOX_NORMALIZE_TARGETS_PRECOMPUTED=False, BBOX_THRESH=0.5,
    BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0), BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2),
    BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0),
    HAS_RPN=True, RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_CLOBBER_POSITIVES=False, RPN_FG_FRACTION=0.5,
    RPN_BATCHSIZE=256, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_MIN_SIZE=16,
    RPN_BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0), RPN_POSITIVE_WEIGHT=-1.0,
    MIX_INDEX=True, CFM_INPUT_MASK_SIZE=14, FG_DET_THRESH=0.5, FG_SEG_THRESH=0.5, FRACTION_SAMPLE=[0.3, 0.5, 0.2],
    THRESH_LO_SAMPLE=[0.5, 0.1, 0.0], THRESH_HI_SAMPLE=[1.0, 0.5, 0.1])
cfg.TEST = AttrDict(
    SCALES=(600,), MAX_SIZE=1000, NMS=0.3, HAS_RPN=True,
    RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, RPN_MIN_SIZE=16,   # mnc_config.py:121-129
    BBOX_REG=True, MASK_MERGE_IOU_THRESH=0.5, MASK_MERGE_NMS_THRESH=0.3,                   # :133-134
    CFM_INPUT_MASK_SIZE=14, MAX_ROIS_GPU=[2000], GROUP_SCALE=1, USE_TOP_K_MCG=0,
    USE_MASK_MERGE=True, USE_GPU_MASK_MERGE=True,
    # not in the reference: im_detect / _segmentation_forward leave boxes, masks and scores on the GPU for gpu_mask_voting
    DEVICE_RESULTS=True, DEVICE_PREP=True)


def get_output_dir(imdb, net):
    path = os.path.abspath(os.path.join(cfg.ROOT_DIR, "output", cfg.EXP_DIR, imdb.name))
    return path if net is None else os.path.join(path, net.name)


def _merge(user, default, prefix=""):
    for key, val in user.items():
        if key not in default:
            raise KeyError("{} is not a valid config key".format(prefix + key))
        cur = default[key]
        if isinstance(cur, dict):
            if not isinstance(val, dict):
                raise ValueError("config key {} expects a mapping".format(prefix + key))
            _merge(val, cur, prefix + key + ".")
            continue
        if isinstance(cur, np.ndarray):
            val = np.array(val, dtype=cur.dtype)
        elif isinstance(cur, tuple) and isinstance(val, list):
            val = tuple(val)
        elif type(cur) is not type(val) and not (isinstance(cur, float) and isinstance(val, int)):
            raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(cur), type(val), prefix + key))
        default[key] = val


def cfg_from_file(file_name):
    """Merge a YAML file into `cfg` with the reference's key/type checking (mnc_config.py:167-206)."""
    import yaml
    with open(file_name, "r") as f:
        user = yaml.safe_load(f) or {}
    _merge(user, cfg)
