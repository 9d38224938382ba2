"""One LDE (sppark_amd.LDE: Goldilocks 2^LG -> 2^(LG+LB)) a few times: the command behind a kernel trace / timing.
    python tools/gpu_lde_one.py [field] LG LB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
args = sys.argv[1:]
field = args.pop(0) if not args[0].isdigit() else "gl64"
lg, lb = int(args[0]), int(args[1])
words = {"gl64": 1, "bb31": 1}.get(field, 4)
dt = torch.int32 if field == "bb31" else torch.int64
hi = 0x78000000 if field == "bb31" else (1 << 62)
x = torch.randint(0, hi, ((1 << (lg + lb)) * words,), dtype=dt, device="cuda")
if words == 4: x[3::4] &= 0x0fffffffffffffff
torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    sppark_amd.LDE(0, x, lg, lb, field, stream=s)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    sppark_amd.LDE(0, x, lg, lb, field, stream=s)
e1.record(); torch.cuda.synchronize()
print("%s LDE 2^%d -> 2^%d: %.3f ms" % (field, lg, lg + lb, e0.elapsed_time(e1) / 10), flush=True)
