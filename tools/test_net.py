#!/usr/bin/env python3
"""Evaluate an MNC network on an image database (reference: tools/test_net.py:24-84): `--task seg`, the 5-stage
MNC graph this package accelerates; `--task det`, Faster R-CNN end2end, and `--task cfm`, convolutional feature masking over
MCG proposals (with `--cfg experiments/cfgs/VGG16/cfm.yml`: 5-level pyramid, levels grouped 3 + 2 per forward), on the same
kernels.

    python tools/test_net.py --gpu 0 --def models/VGG16/mnc_5stage/test.prototxt \\
        --net data/mnc_model/mnc_model.caffemodel.h5 --imdb voc_2012_seg_val --task seg
"""
import argparse
import os
import pprint
import sys
import time

import _init_paths  # noqa: F401
import caffe
from caffeWrapper.TesterWrapper import TesterWrapper
from db.imdb import get_imdb
from mnc_config import cfg, cfg_from_file


def parse_args():
    parser = argparse.ArgumentParser(description='Test an MNC network')
    parser.add_argument('--gpu', dest='gpu_id', help='GPU id to use', default=0, type=int)
    parser.add_argument('--def', dest='prototxt', help='prototxt file defining the network', default=None, type=str)
    parser.add_argument('--net', dest='caffemodel', help='model to test', default=None, type=str)
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to test', default='voc_2012_seg_val', type=str)
    parser.add_argument('--wait', dest='wait', help='wait until net file exists', default=True, type=bool)
    parser.add_argument('--comp', dest='comp_mode', help='competition mode', action='store_true')
    parser.add_argument('--set', dest='set_cfgs', help='set config keys', default=None, nargs=argparse.REMAINDER)
    parser.add_argument('--task', dest='task_name', help='set task name', default='seg', type=str)
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args()


if __name__ == '__main__':
    args = parse_args()
    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    cfg.GPU_ID = args.gpu_id
    print('Using config:')
    pprint.pprint(cfg)
    while not os.path.exists(args.caffemodel) and args.wait:
        print('Waiting for {} to exist...'.format(args.caffemodel))
        time.sleep(10)
    caffe.set_mode_gpu()
    caffe.set_device(args.gpu_id)
    imdb = get_imdb(args.imdb_name)
    _tester = TesterWrapper(args.prototxt, imdb, args.caffemodel, args.task_name)
    _tester.get_result()
