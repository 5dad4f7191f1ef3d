#!/usr/bin/env python3
"""Reference CLI (make_voc_list.py:29-38): python make_voc_list.py <train.txt> <data/voc_img_ann.npy>."""
import argparse
import sys

from k210_yolo_framework_amd.datatools import make_voc_list

if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('train_file', type=str, help='trian.txt file path')
    p.add_argument('output_file', type=str, help='output file path')
    a = p.parse_args(sys.argv[1:])
    make_voc_list(a.train_file, a.output_file)
