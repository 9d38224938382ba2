"""Polynomial primitives on device-resident Goldilocks arrays: time per call (the scratch pool's effect shows at the small sizes).
    python tools/gpu_poly_one.py [field]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import poly
field = sys.argv[1] if len(sys.argv) > 1 else "gl64"
for lg in (12, 16, 20, 24):
    n = 1 << lg
    x = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    out = torch.empty_like(x)
    for name, fn in (("prefix_op add", lambda: poly.prefix_op(out, x, 0, field)), ("prefix_op mul", lambda: poly.prefix_op(out, x, 1, field))):
        for _ in range(3): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("%s 2^%d %s: %.3f ms" % (field, lg, name, e0.elapsed_time(e1) / 20), flush=True)
