"""tests/golden/ref_field_golden.json: outputs of the REFERENCE's own device field classes, recorded on an MI355X.

Under __HIPCC__ the reference defines fp_t / fr_t over ff/mont_t.hip (ff/bls12-381.hpp:63-83, ff/alt_bn128.hpp:60-82,
ff/bls12-377.hpp:61-85, ff/pasta.hpp:55-79); oracle/_ref/libref_field_<curve>.so (oracle/Makefile: ref_field, oracle/ref_field_shim.cu)
is that code built for gfx950 where the sources lie, and this script -- run on the GPU box, where the libraries arrive prebuilt --
applies its operators + - * sqr() to() from() (ff/mont_t.hip:96-218) to fixed, seeded, edge-heavy operands and stores operands and
results as hex.  The file is DATA (inputs and the reference's outputs); tests/test_oracle.py holds the ORACLE's field (oracle/ff.hpp, a
different algorithm on the CPU) against it in the non-GPU suite, which pins the oracle's field layer to the reference itself.

    gpurun -- 'python tests/golden/make_ref_field_golden.py gpurun_out/ref_field_golden.json'    # then copy into tests/golden/
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O                                             # noqa: E402

CURVES = [(O.BLS12_381, "bls12_381"), (O.BN254, "bn254"), (O.BLS12_377, "bls12_377"), (O.PALLAS, "pallas"), (O.VESTA, "vesta")]
OPS = ["add", "sub", "mul", "sqr", "to", "from"]               # ref_field_shim.cu's op numbers 0 .. 5
N = 40


def operands(p, nb, seed):
    rng = random.Random(seed)
    R = 1 << (8 * nb)
    edge = [0, 1, 2, p - 1, p - 2, R % p, R * R % p, (p - 1) // 2, (p + 1) // 2, (R - 1) % p, (1 << 28) % p, (1 << 32) % p,
            ((1 << 64) - 1) % p, (1 << (8 * nb - 2)) % p]
    a = edge + [rng.randrange(p) for _ in range(N - len(edge))]
    b = list(reversed(edge)) + [rng.randrange(p) for _ in range(N - len(edge))]
    return a, b


def main(path):
    cases = []
    for curve, name in CURVES:
        for which, p, nb in ((0, O.FP_MODULUS[curve], O.FP_BYTES[curve]), (1, O.FR_MODULUS[curve], 32)):
            va, vb = operands(p, nb, 1000 * curve + which)
            a = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in va), dtype=np.uint8).copy()
            b = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in vb), dtype=np.uint8).copy()
            case = {"curve": name, "field": "fp" if which == 0 else "fr", "bytes": nb, "n": N, "modulus": hex(p),
                    "a": a.tobytes().hex(), "b": b.tobytes().hex(), "expect": {}}
            for op, opname in enumerate(OPS):
                case["expect"][opname] = O.ref_field_op(name, which, op, a, b).tobytes().hex()
            cases.append(case)
    with open(path, "w") as f:
        json.dump({"what": "outputs of the reference's own fp_t / fr_t (ff/mont_t.hip, built for gfx950) on an MI355X; little-endian Montgomery images",
                   "generator": "tests/golden/make_ref_field_golden.py", "ops": OPS, "cases": cases}, f, indent=0)
    print("wrote %s: %d cases x %d ops x %d elements" % (path, len(cases), len(OPS), N))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_field_golden.json"))
