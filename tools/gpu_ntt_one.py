"""Run the forward NTT of a field (gl64, bb31, bls12_381, ...) a few times at 2^LG (for profiling): FIELD LG REPS [NR|NN|RN|RR]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import NTTInputOutputOrder as Ord
field = sys.argv[1] if len(sys.argv) > 1 else "gl64"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dt = torch.int32 if field == "bb31" else torch.int64
words = 4 if field not in ("gl64", "bb31") else 1             # 256-bit fields: four 64-bit words per element (any values < 2^252)
x = torch.randint(0, 2**30, ((1 << lg) * words,), dtype=dt, device="cuda")
torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
s = torch.cuda.current_stream().cuda_stream
order = getattr(Ord, sys.argv[4]) if len(sys.argv) > 4 else Ord.NR
for _ in range(reps):
    sppark_amd.NTT(0, x, order, field, stream=s)
torch.cuda.synchronize()
