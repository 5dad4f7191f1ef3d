#!/usr/bin/env python3
"""Reference CLI (make_anchor_list.py:221-241); plotting (--is_plot) is not reproduced."""
import argparse
import sys

from k210_yolo_framework_amd.datatools import make_anchor_list

if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('train_set', type=str)
    p.add_argument('--max_iters', type=int, default=10)
    p.add_argument('--is_random', type=str, choices=['True', 'False'], default='True')
    p.add_argument('--is_plot', type=str, choices=['True', 'False'], default='False')
    p.add_argument('--in_hw', type=int, default=(224, 320), nargs='+')
    p.add_argument('--out_hw', type=int, default=(7, 10, 14, 20), nargs='+')
    p.add_argument('--low', type=float, default=(0.0, 0.0), nargs='+')
    p.add_argument('--high', type=float, default=(1.0, 1.0), nargs='+')
    p.add_argument('--anchor_num', type=int, default=3)
    a = p.parse_args(sys.argv[1:])
    c = make_anchor_list(a.train_set, tuple(a.in_hw), tuple(a.out_hw), a.anchor_num, a.is_random == 'True', a.low, a.high)
    print(f'[NOTE] Now anchors are :\n{c}')
