"""Worker of test_msm_rccl_exchange_many_ranks_with_a_test_double (not a test module): N "ranks" = N threads of this
process, each calling sppark_msm_rccl on its shard with a communicator of the RCCL test double (tests/emu/fake_rccl.cpp,
selected by SPPARK_RCCL_LIB before the library's first exchange).  Prints one JSON line: the ranks' codes and outputs.

    python tests/rccl_ranks_worker.py inputs.npz '[0, 1000, 1000, 7001]' <bad_rank or -1>
"""
import ctypes
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sppark_amd import ffi                                      # noqa: E402


def main():
    data = np.load(sys.argv[1])
    pts, sc = np.ascontiguousarray(data["pts"]), np.ascontiguousarray(data["sc"])
    cuts = json.loads(sys.argv[2])
    bad = int(sys.argv[3])
    n = len(cuts) - 1
    fake = ctypes.CDLL(os.environ["SPPARK_RCCL_LIB"])
    handles = (ctypes.c_void_p * n)()
    fake.fake_rccl_make(n, handles)
    L = ffi.load("bls12_381")
    outs = [np.full(144, 0xff, dtype=np.uint8) for _ in range(n)]
    codes = [None] * n

    def work(r):
        a, b = cuts[r], cuts[r + 1]
        p, s = pts[a:b], sc[a:b]
        stride = 8 if r == bad else pts.shape[1]                # below two field elements: the local MSM fails
        err = L.sppark_msm_rccl(outs[r].ctypes.data, p.ctypes.data if b > a else None, b - a,
                                s.ctypes.data if b > a else None, 0, stride, handles[r], None)
        codes[r] = err.code
        if err.message:
            L.drop_error_message(err.message)
    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    hung = [t.is_alive() for t in threads]
    print(json.dumps({"codes": codes, "hung": hung, "outs": [o.tobytes().hex() for o in outs]}), flush=True)
    os._exit(1 if any(hung) else 0)


if __name__ == "__main__":
    main()
