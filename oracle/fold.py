"""ORACLE — TEST INFRASTRUCTURE ONLY (checker for full-size MSMs).

An MSM over n points that repeat with period U (P_i = B_{i mod U}, the shape of the reference's
own test/bench inputs, poc/msm-cuda/src/util.rs:11-38) equals the U-point MSM

    sum_j ( sum_{i = j mod U} s_i  mod r ) * B_j

so a 2^26-point result with INDEPENDENT UNIFORM scalars can be compared, bit for bit, with the
oracle on U = 2048 points.  fold_scalars computes the U class sums exactly (integer byte-column
sums on whatever device the scalars live on, then big-int arithmetic mod r on the host).

ANY period and a ragged tail (n not a multiple of U) are allowed.  What the period decides is what the
check can SEE: a gather-index error d (the kernel reads point i + d for entry i) is invisible whenever
d = 0 mod U.  U = 2048 is the reference's shape and is blind to every power-of-two slip from 2^11 on
(a wrong index group of the 4-byte sort records is +-k 2^IB, a 32-bit wrap of i or of i * stride is a
power of two); an odd prime U (2039) sees all of them: k 2^m = 0 mod 2039 only for k = 0 mod 2039.

weighted_sums gives sum s_i and sum i s_i as integers: on the all-distinct progression
P_i = (a + i b) G (sppark_g1_generate_progression) the MSM is (a sum s_i + b sum i s_i mod r) G."""
import numpy as np


def fold_scalars(scalars, period, r):
    """scalars: (n, 32) uint8 torch tensor or numpy array (n need not be a multiple of |period|).
    Returns the (period, 32) uint8 numpy array of class sums mod r."""
    n = scalars.shape[0]
    whole = n // period * period
    rows = max(period, (1 << 22) // period * period)
    if isinstance(scalars, np.ndarray):
        cols = np.zeros((period, 32), dtype=np.int64)
        for lo in range(0, whole, rows):            # bounded temporaries
            cols += scalars[lo:min(lo + rows, whole)].reshape(-1, period, 32).sum(axis=0, dtype=np.int64)
        if whole < n:
            cols[:n - whole] += scalars[whole:]
    else:
        import torch
        cols = torch.zeros((period, 32), dtype=torch.int64, device=scalars.device)
        for lo in range(0, whole, rows):
            blk = scalars[lo:min(lo + rows, whole)]
            cols += blk.reshape(-1, period, 32).sum(dim=0, dtype=torch.int64)
        if whole < n:
            cols[:n - whole] += scalars[whole:].to(torch.int64)
        cols = cols.cpu().numpy()
    out = np.zeros((period, 32), dtype=np.uint8)
    for j in range(period):
        v = 0
        for k in range(32):
            v += int(cols[j, k]) << (8 * k)
        out[j] = np.frombuffer((v % r).to_bytes(32, "little"), dtype=np.uint8)
    return out


def weighted_sums(scalars):
    """(sum_i s_i, sum_i i * s_i) as Python integers for an (n, 32) uint8 array / tensor of little-endian scalars,
    n < 2^30: byte-column sums in int64 (a column of sum i * byte is below 2^30 * 2^30 * 2^8 only in pieces, so the
    index is split: i = hi * 2^15 + lo)."""
    n = scalars.shape[0]
    assert n < (1 << 30)
    S0 = [0] * 32; S1 = [0] * 32
    step = 1 << 22
    if isinstance(scalars, np.ndarray):
        for lo in range(0, n, step):
            blk = scalars[lo:lo + step].astype(np.int64)
            idx = np.arange(lo, lo + blk.shape[0], dtype=np.int64)
            c0 = blk.sum(axis=0); cl = ((idx & 0x7fff)[:, None] * blk).sum(axis=0); ch = ((idx >> 15)[:, None] * blk).sum(axis=0)
            for k in range(32):
                S0[k] += int(c0[k]); S1[k] += int(cl[k]) + (int(ch[k]) << 15)
    else:
        import torch
        for lo in range(0, n, step):
            blk = scalars[lo:lo + step].to(torch.int64)
            idx = torch.arange(lo, lo + blk.shape[0], dtype=torch.int64, device=scalars.device)
            c0 = blk.sum(dim=0).cpu(); cl = ((idx & 0x7fff)[:, None] * blk).sum(dim=0).cpu(); ch = ((idx >> 15)[:, None] * blk).sum(dim=0).cpu()
            for k in range(32):
                S0[k] += int(c0[k]); S1[k] += int(cl[k]) + (int(ch[k]) << 15)
    return sum(S0[k] << (8 * k) for k in range(32)), sum(S1[k] << (8 * k) for k in range(32))
