"""2-layer sampled GCN trainer — counterpart of the reference's examples/profile/pa_gcn.py
(same command line, same prints):  python examples/profile/pa_gcn.py --dataset DIR --gpu 0[,1,..]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import main

if __name__ == '__main__':
    main('gcn', 'GCN', n_hidden=32, lr=3e-2)     # pa_gcn.py:130,137
