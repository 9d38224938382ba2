"""Small transforms (2^8 .. 2^20), ours against the reference's HIP build, device-resident, forward NR: both timed by
events around REPS back-to-back calls issued from C-speed loops (ours: the bare ctypes entry point sppark_ntt, no Python
wrapper work per call; the reference: ref_ntt_dev_timed).  `only=<ours|ref> lg=<k>` runs one side once more in a loop
for a kernel trace."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle as O
from sppark_amd import ffi

REPS = 200
args = dict(a.split("=") for a in sys.argv[1:])
torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
stream = torch.cuda.current_stream().cuda_stream
for field, dt, eb in (("gl64", torch.int64, 8), ("bb31", torch.int32, 4), ("bls12_381", torch.int64, 32)):
    if args.get("field", field) != field:
        continue
    L = ffi.load(field)
    for lg in ([int(args["lg"])] if "lg" in args else range(8, 21)):
        n = 1 << lg
        x = torch.randint(0, 2**30, (n * (eb // 8 if eb >= 8 else 1),), dtype=dt, device="cuda")
        p = ctypes.c_void_p(x.data_ptr())
        out = []
        if args.get("only", "ours") == "ours":
            for _ in range(5):
                L.sppark_ntt(0, p, lg, 1, 0, 0, stream)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                L.sppark_ntt(0, p, lg, 1, 0, 0, stream)
            e1.record(); torch.cuda.synchronize()
            out.append("ours %.4f ms" % (e0.elapsed_time(e1) / REPS))
        if args.get("only", "ref") == "ref":
            out.append("reference %.4f ms" % O.ref_ntt_dev_ms(field, x.data_ptr(), lg, 1, 0, 0, REPS))
        print("%-9s 2^%-2d fwd NR: %s" % (field, lg, ", ".join(out)), flush=True)
