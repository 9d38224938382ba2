"""ORACLE — TEST INFRASTRUCTURE ONLY (checker for full-size MSMs).

An MSM over n points that repeat with period U (P_i = B_{i mod U}, the shape of the reference's
own test/bench inputs, poc/msm-cuda/src/util.rs:11-38) equals the U-point MSM

    sum_j ( sum_{i = j mod U} s_i  mod r ) * B_j

so a 2^26-point result with INDEPENDENT UNIFORM scalars can be compared, bit for bit, with the
oracle on U = 2048 points.  fold_scalars computes the U class sums exactly (integer byte-column
sums on whatever device the scalars live on, then big-int arithmetic mod r on the host)."""
import numpy as np


def fold_scalars(scalars, period, r):
    """scalars: (n, 32) uint8 torch tensor or numpy array, n a multiple of |period|.
    Returns the (period, 32) uint8 numpy array of class sums mod r."""
    n = scalars.shape[0]
    assert n % period == 0
    if isinstance(scalars, np.ndarray):
        cols = np.zeros((period, 32), dtype=np.int64)
        rows = max(period, (1 << 22) // period * period)
        for lo in range(0, n, rows):                # bounded temporaries
            cols += scalars[lo:lo + rows].reshape(-1, period, 32).sum(axis=0, dtype=np.int64)
    else:
        import torch
        cols = torch.zeros((period, 32), dtype=torch.int64, device=scalars.device)
        rows = max(period, (1 << 22) // period * period)
        for lo in range(0, n, rows):
            blk = scalars[lo:lo + rows]
            cols += blk.reshape(-1, period, 32).sum(dim=0, dtype=torch.int64)
        cols = cols.cpu().numpy()
    out = np.zeros((period, 32), dtype=np.uint8)
    for j in range(period):
        v = 0
        for k in range(32):
            v += int(cols[j, k]) << (8 * k)
        out[j] = np.frombuffer((v % r).to_bytes(32, "little"), dtype=np.uint8)
    return out
