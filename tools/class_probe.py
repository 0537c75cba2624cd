#!/usr/bin/env python3
"""Developer probe (GPU box): search-kernel time for all-straight / all-in-arc / mixed batches at several sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_probe import run
for B in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    for st in (True, False, None):
        run('f32', B, 2, straight=st, iters=9)
