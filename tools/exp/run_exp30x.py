"""GPU check of ff/montx_dev.hpp (LB = 28, BLS12-381 Fp): ops against Python big-ints, then throughput
of products / squares / a mixed-addition chain against the 32-bit-limb class."""
import ctypes, os, random, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
import torch  # noqa: F401  (HIP runtime first)
L = ctypes.CDLL(os.path.join(HERE, "libexp30x.so"))
L.exp_op.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
L.exp_bench.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
L.exp_bench.restype = ctypes.c_float
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
LB, NL = 28, 14
R = 1 << (LB * NL)
MASK = (1 << LB) - 1
def limbs(v): return [(v >> (LB * j)) & MASK if j < NL - 1 else v >> (LB * j) for j in range(NL)]
def val(l): return sum(int(x) << (LB * j) for j, x in enumerate(l))
random.seed(1)
n = 4096
def rnd(k):     # value < k*p
    r = random.random()
    if r < 0.05: return random.choice([0, 1, P - 1, P, P + 1, 2 * P, k * P - 1])
    return random.randrange(k * P)
A = [rnd(3) for _ in range(n)]; B = [rnd(3) for _ in range(n)]
a = np.array([limbs(v) for v in A], dtype=np.uint32); b = np.array([limbs(v) for v in B], dtype=np.uint32)
out = np.zeros_like(a)
def run(op, a_, b_):
    rc = L.exp_op(op, out.ctypes.data, a_.ctypes.data, b_.ctypes.data, n); assert rc == 0
    return [val(row) for row in out], out.copy()
Rinv = pow(R, P - 2, P)
got, raw = run(0, a, b)
assert all(g % P == x * y * Rinv % P and g < x * y // R + P + 1 for g, x, y in zip(got, A, B)), "mul"
assert (raw[:, :NL - 1] <= MASK).all(), "mul limbs normalised"
got, raw = run(1, a, b)
assert all(g % P == x * x * Rinv % P for g, x in zip(got, A)), "sqr"
# un-normalised first operand: a lazy difference a + 12p - c with fat limbs (B = 5), value < 15p
def fat(K, Bw):
    pl = limbs(K * P)
    return [pl[j] + ((Bw << LB) if j < NL - 1 else 0) - (Bw if j > 0 else 0) for j in range(NL)]
F = fat(12, 5)
assert val(F) == 12 * P
C = [random.randrange(11 * P) for _ in range(n)]
a_lazy = np.array([[x + f - y for x, f, y in zip(limbs(A[i]), F, limbs(C[i]))] for i in range(n)], dtype=np.uint32)
AL = [val(r) for r in a_lazy]
assert all(v == A[i] + 12 * P - C[i] for i, v in enumerate(AL)) and int(a_lazy.max()) < (1 << 31)
got, raw = run(0, a_lazy, b)
assert all(g % P == x * y * Rinv % P for g, x, y in zip(got, AL, B)), "mul lazy operand"
F2 = fat(12, 2)                       # sqr's contract: limbs < 2^30
a_lazy2 = np.array([[x + f - y for x, f, y in zip(limbs(A[i]), F2, limbs(C[i]))] for i in range(n)], dtype=np.uint32)
AL2 = [val(r) for r in a_lazy2]
assert int(a_lazy2.max()) < (1 << 30)
got, raw = run(1, a_lazy2, b)
assert all(g % P == x * x * Rinv % P for g, x in zip(got, AL2)), "sqr lazy operand"
got, raw = run(2, a, b)        # a + 3p - b (b < 2p needed: use b mod 2p)
B2 = [v % (2 * P) for v in B]; b2 = np.array([limbs(v) for v in B2], dtype=np.uint32)
got, raw = run(2, a, b2)
assert all(g == x + 3 * P - y for g, x, y in zip(got, A, B2)), "sub"
assert (raw[:, :NL - 1] <= MASK).all()
got, raw = run(3, a, b)
assert all(g == x + y for g, x, y in zip(got, A, B)), "add"
Z = [random.choice([0, P, 2 * P, 5 * P, 13 * P, 7 * P + 1, P - 1, random.randrange(14 * P), (random.randrange(14 * P) & ~MASK) | ((k * P) & MASK)]) for k in range(n)]
z = np.array([limbs(v) for v in Z], dtype=np.uint32)
got, raw = run(4, z, b)
assert all(int(raw[i, 0]) == (1 if Z[i] % P == 0 and Z[i] < 14 * P else 0) for i in range(n)), "is_zero_mod"
print("montx ops OK", flush=True)
x = np.array([limbs(random.randrange(P)) for _ in range(1024)], dtype=np.uint32)
blocks, iters = 256 * 8, 400
for which, name in ((0, "montx mul"), (1, "montx sqr"), (2, "mont32 mul"), (3, "mont32 sqr"), (4, "montx madd chain"), (5, "mont32 madd chain")):
    it = iters if which < 4 else 100
    ms = L.exp_bench(which, x.ctypes.data, blocks, it)
    print("%-18s %8.3f ms  %.3e ops/s" % (name, ms, blocks * 256 * it / (ms * 1e-3)), flush=True)
