"""2-layer sampled GraphSAGE-mean trainer — counterpart of the reference's
examples/profile/pa_gs.py:  python examples/profile/pa_gs.py --dataset DIR --gpu 0[,1,..]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import main

if __name__ == '__main__':
    main('graphsage', 'GraphSAGE', n_hidden=16, lr=1e-2)   # pa_gs.py:134,141
