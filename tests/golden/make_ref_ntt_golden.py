"""tests/golden/ref_ntt_golden.json: outputs of the REFERENCE's own NTT (poc/ntt-cuda/cuda/ntt_api.cu compute_ntt, built for gfx950
through the reference's HIP path into oracle/_ref/libref_ntt_<field>.so -- oracle/Makefile: ref_ntt), recorded on an MI355X.

Run on the GPU box, where the libraries arrive prebuilt: every order x direction x type (ntt/ntt.cuh:33-36) at 2^1, 2^2, 2^3 and 2^5
(2^1 .. 2^3 for the 256-bit fields) on seeded inputs, all nine field libraries incl. both compile-time root conventions.  The file is DATA
(inputs and the reference's outputs as hex); tests/test_oracle.py holds the ORACLE's NTT restatement (oracle/ntt.hpp, CPU) against it in
the non-GPU suite -- the same pin the `-m gpu` tests apply live (tests/test_ntt_vs_reference_gpu.py), without needing a GPU.

    gpurun -- 'python tests/golden/make_ref_ntt_golden.py gpurun_out/ref_ntt_golden.json'      # then copy into tests/golden/
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import oracle as O                                             # noqa: E402
import recipe                                                  # noqa: E402

LIBS = ["gl64", "gl64_plonky2", "bb31", "bb31_canonical", "bls12_381", "bn254", "bls12_377", "pallas", "vesta"]
KIND = {"gl64_plonky2": "gl64", "bb31_canonical": "bb31"}


def main(path):
    cases = []
    for lib in LIBS:
        kind = KIND.get(lib, lib)
        for lg in ((1, 2, 3, 5) if kind in ("gl64", "bb31") else (1, 2, 3)):
            x = recipe.ntt_input(kind, lg, 7000 + lg)
            case = {"lib": lib, "kind": kind, "lg": lg, "dtype": str(x.dtype), "input": np.ascontiguousarray(x).tobytes().hex(), "expect": {}}
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        y = O.ref_compute_ntt(lib, x, order, direction, typ)
                        case["expect"]["%d%d%d" % (order, direction, typ)] = np.ascontiguousarray(y).tobytes().hex()
            cases.append(case)
    with open(path, "w") as f:
        json.dump({"what": "outputs of the reference's own compute_ntt (its HIP build for gfx950) on an MI355X; key = order direction type digits",
                   "generator": "tests/golden/make_ref_ntt_golden.py", "cases": cases}, f, indent=0)
    print("wrote %s: %d cases x 16 modes" % (path, len(cases)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_ntt_golden.json"))
