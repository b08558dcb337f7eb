"""Image database factory (reference: lib/db/imdb.py:8-28).  Test-time surfaces only;
extra sets can be registered with add_imdb (used by the tests for a synthetic devkit)."""
from datasets.pascal_voc_det import PascalVOCDet
from datasets.pascal_voc_seg import PascalVOCSeg

_sets = {
    'voc_2012_seg_train': (lambda: PascalVOCSeg('train', '2012', 'data/VOCdevkitSDS/')),
    'voc_2012_seg_val': (lambda: PascalVOCSeg('val', '2012', 'data/VOCdevkitSDS/')),
    'voc_2007_trainval': (lambda: PascalVOCDet('trainval', '2007')),
    'voc_2007_test': (lambda: PascalVOCDet('test', '2007')),
}


def add_imdb(name, factory):
    _sets[name] = factory


def get_imdb(name):
    if name not in _sets:
        raise KeyError('Unknown dataset: {}'.format(name))
    return _sets[name]()


def list_imdbs():
    return list(_sets.keys())
