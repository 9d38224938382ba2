#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--lg 26 | --total-lg 28]

Headline (BASELINE.json metric): BLS12-381 G1 Pippenger MSM of 2^26 points, inputs resident in
HBM, on N GPUs of one node.  A step = one full MSM.

  N = 1                 2^26 points on one GPU (configs[2])
  N > 1, default        STRONG scaling: the SAME 2^26-point MSM cut into N contiguous shards, one
                        rank per GPU, one RCCL all-gather of the 144-byte partial sums per step,
                        combine on every rank -- "MSM points/sec (2^26) at 1/2/4/8 GPUs"
  --total-lg 28         strong scaling at 2^28 total (configs[3]: 2^25 per rank on 8 GPUs)
  --scaling weak        2^lg points PER RANK (a 2^lg * N-point MSM)
  --backend gloo        dry run of the world > 1 control flow on a 1-GPU box: the ranks share the GPU and
                        exchange through host memory (tests/test_bench_multirank_gpu.py); not a measurement

Inputs follow poc/msm-cuda/src/util.rs:11-38: 2^11 distinct points replicated cyclically
(point 3 = infinity), independent uniform scalars on [0, r) (rejection sampled).
The result of the LAST TIMED step is asserted bit-exact (affine) against the oracle: the point
period lets the whole MSM be folded into a 2048-point one (oracle/fold.py).

Secondary (same JSON line, key "ntt"): Goldilocks NTT 2^24 forward NR + inverse RN (configs[1]),
output asserted equal to the oracle's at the timed size.

One JSON line on rank 0; "roofline" is for the dominant kernel k_accumulate (HIP-event timed
inside the library on the launch stream), "roofline_alu" prices the same kernel against the
measured v_mad_u64_u32 issue peak, "cpu_baseline" is the reference's msm/pippenger.hpp
(oracle/_ref) on the host cores (N = 1 only)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import sppark_amd
from sppark_amd import multi_gpu, synth

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MSM_BYTES_PER_POINT = 128           # 96-byte affine point + 32-byte scalar
NTT_BYTES_PER_ELEM = 16             # read + write one u64 per transform
PERIOD = 2048                       # distinct points (util.rs:15 uses 2^11 too)
# measured issue peak of the one wide integer multiplier, v_mad_u64_u32: 0.181 wave-instr/clk/SIMD
# (profiles/r01_ubench2_instruction_rates.log) x 1024 SIMDs x 64 lanes x 2.4 GHz
# v_mad_u64_u32 issue rate, wave-instructions / clk / SIMD at the nominal 2.4 GHz, measured with
# tools/exp/ubench_carry.hip (profiles/r02_ubench_instruction_rates.log): 0.226 with 8 waves per SIMD,
# 0.178 with the 2 waves per SIMD that the 230 registers of k_accumulate allow
MAD_RATE_PEAK, MAD_RATE_2WAVES = 0.226, 0.178
MAD_PEAK_PER_S = MAD_RATE_PEAK * 1024 * 64 * 2.4e9
# ff/montx_dev.hpp, ec/xyzzx_dev.hpp madd: 8 products (14x14) + 2 squares (105) + 9 Montgomery reductions
# (14x14 each; Y3 is one reduced sum of two products)
MADS_PER_MIXED_ADD = 8 * 196 + 2 * 105 + 9 * 196
MADS_PER_MIXED_ADD_9 = 8 * 81 + 2 * 45 + 9 * 81       # alt_bn128 / Pasta: the same formulas on nine 29-bit limbs (round 5)
# G2 (wave-pair kernel): 8 Fp2 products (two sums of two base products, one reduction each: 2 x 3 x 196) + 2 Fp2 squares (2 x 2 x 196)
MADS_PER_G2_MIXED_ADD = 8 * 6 * 196 + 2 * 4 * 196


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_ntt_time(field, host_in, expect, lg, reps=7):
    """compute_ntt(device 0, HOST buffer, forward NR) in place: wall-clock ms (median, min) over |reps| calls
    after one warm-up; the output of the last call asserted equal to |expect|."""
    Ord = sppark_amd.NTTInputOutputOrder
    buf = host_in.copy()
    sppark_amd.NTT(0, buf, Ord.NR, field)
    times = []
    for _ in range(reps):
        buf[:] = host_in
        t1 = time.perf_counter()
        sppark_amd.NTT(0, buf, Ord.NR, field)                   # stream=None: the reference's compute_ntt, synchronous
        times.append((time.perf_counter() - t1) * 1e3)
    ok = bool((buf == expect).all())
    assert ok, "compute_ntt on a host buffer differs from the checked device path (%s)" % field
    med = float(np.median(times))
    nbytes = buf.nbytes
    return {"entry_point": "compute_ntt(device_id=0, host inout, 2^%d, NR, forward, standard)" % lg, "bytes_each_way": nbytes,
            "ms_median": med, "ms_min": float(min(times)), "elems_per_s": (1 << lg) / (med * 1e-3),
            "copy_gb_per_s": 2 * nbytes / (med * 1e-3) / 1e9, "equals_checked_device_output": ok,
            "note": "wall clock of the whole call: H2D + transform + D2H from pageable host memory (PCIe inclusive; never the headline value)"}


def pcie_peaks(nbytes=1 << 30, reps=4):
    """H2D / D2H rate of one pinned |nbytes| copy on this box (GB/s, best of |reps|): the denominator of the host-buffer paths"""
    hbuf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dbuf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = {"h2d": 0.0, "d2h": 0.0}
    for _ in range(reps):
        for key, dst, src in (("h2d", dbuf, hbuf), ("d2h", hbuf, dbuf)):
            e0.record(); dst.copy_(src, non_blocking=True); e1.record(); torch.cuda.synchronize()
            best[key] = max(best[key], nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    # pageable source (what mult_pippenger_inf / compute_ntt callers pass): the runtime stages it itself
    pbuf = np.empty(nbytes, dtype=np.uint8); pbuf[::4096] = 1
    pt = torch.from_numpy(pbuf)
    pg = 0.0
    for _ in range(2):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dbuf.copy_(pt); torch.cuda.synchronize()
        pg = max(pg, nbytes / (time.perf_counter() - t1) / 1e9)
    del hbuf, dbuf
    return best["h2d"], best["d2h"], pg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lg", type=int, default=26, help="log2 points: of the whole MSM (strong) / per GPU (weak)")
    ap.add_argument("--total-lg", type=int, default=None, help="strong scaling with 2^TOTAL_LG points in total (configs[3]: 28)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--ntt-lg", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--all-extras", action="store_true",
                    help="also the G2 MSM, the 256-bit-field NTT and the LDE (SURVEY 8(f) rows; their numbers are in DESIGN.md section 0 "
                         "from the evidence run -- left out of the default line so that the driver's run stays near 30 s)")
    ap.add_argument("--groups", type=int, default=0, help="window groups of the MSM pipeline (0 = automatic)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="exchange backend: nccl = RCCL over xGMI, one GPU per rank (what the driver runs); gloo = the ranks "
                         "share the visible GPU(s) round-robin and exchange through host memory -- a dry run of the whole "
                         "world > 1 control flow on a 1-GPU box, never a performance number")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())         # dry run: ranks share the visible devices
    torch.cuda.set_device(local_rank)
    dist = None
    # SPPARK_FORCE_DIST=1 exercises the exchange with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("SPPARK_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    assert args.gpus == world, "--gpus must equal WORLD_SIZE"

    scaling = args.scaling
    if args.total_lg is not None:
        scaling, total = "strong", 1 << args.total_lg
    elif scaling == "strong":
        total = 1 << args.lg
    else:
        total = world << args.lg
    # contiguous shards in whole periods of the point list (any rank count: 3, 5, 6, 7 GPUs give shards that
    # differ by one period), so that every shard folds onto the same 2048 distinct points for the checker
    assert total % PERIOD == 0, "the MSM must be a whole number of point periods"
    lo, hi = (PERIOD * v for v in multi_gpu.shard_bounds(total // PERIOD, world, rank))
    n = hi - lo                                                  # this rank's points
    assert n > 0, "more ranks than point periods"

    pts, base = synth.replicated_points(n, "bls12_381", PERIOD, 0x5eed5eed0001)
    sc = synth.uniform_scalars(n, "bls12_381", 0x5eed5eed0001 + rank)
    ctx = sppark_amd.MsmContext("bls12_381", device_id=-1, stream=torch.cuda.current_stream().cuda_stream)
    if args.groups:
        ctx.tune_pipeline(groups=args.groups)
    ctx.enable_timing(True)
    ctx.reserve(n, 96)

    def step():
        part = ctx.invoke(pts, sc)
        if use_dist:
            return multi_gpu.combine_partials(multi_gpu.all_gather_bytes(part))
        return part

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        result = step()
    accum_ms, sort_ms, dev_ms, step_ms = [], [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        result = step()                                         # synchronous: returns with the result on the host
        step_ms.append((time.perf_counter() - ts) * 1e3)
        accum_ms.append(ctx.kernel_ms(1)); sort_ms.append(ctx.kernel_ms(0)); dev_ms.append(ctx.kernel_ms(2))
    fence()
    elapsed = time.perf_counter() - t0
    acc_launches = int(ctx.kernel_ms(3))
    if use_dist:
        t = torch.tensor([elapsed] + step_ms, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                # every step: the slowest rank's time
        elapsed = float(t[0].item()); step_ms = [float(v) for v in t[1:]]

    # ---- CHECKER (outside the timed region): the last timed result against the oracle ------------
    # class sums of this rank's scalars (exact integer arithmetic), gathered over the ranks; rank 0
    # evaluates the folded 2048-point MSM with the oracle and compares affine coordinates bit for bit
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import oracle as O                                          # checker / cpu_baseline legs only
    from oracle import fold
    r_mod = O.FR_MODULUS[O.BLS12_381]
    folded = fold.fold_scalars(sc, PERIOD, r_mod)
    if use_dist:
        allf = multi_gpu.all_gather_bytes(folded.reshape(-1)).reshape(-1, PERIOD, 32)
    else:
        allf = folded.reshape(1, PERIOD, 32)
    parity = None
    if rank == 0:
        tot = np.zeros((PERIOD, 32), dtype=np.uint8)
        for j in range(PERIOD):
            v = sum(int.from_bytes(allf[k, j].tobytes(), "little") for k in range(allf.shape[0])) % r_mod
            tot[j] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
        expect = O.msm_affine(O.BLS12_381, base.cpu().numpy(), tot, algo=0, param=8)
        got = sppark_amd.to_affine(result)
        parity = {"timed_msm_equals_oracle": bool((got == expect).all()),
                  "how": "last timed result, affine, vs oracle MSM of the 2048 distinct points with the exactly folded scalars (oracle/fold.py)",
                  "result_affine_sha256": hashlib.sha256(got.tobytes()).hexdigest()}
        assert parity["timed_msm_equals_oracle"], "timed MSM result differs from the oracle"

    ntt = None
    if rank == 0 and not args.no_ntt:
        lg = args.ntt_lg
        g = torch.Generator(device="cuda"); g.manual_seed(2)
        x = (torch.randint(0, 2**62, (1 << lg,), dtype=torch.int64, device="cuda", generator=g))   # < p
        # A NON-NULL stream for the device-resident timings: on the NULL stream sppark_ntt is synchronous (it mirrors the
        # reference's compute_ntt), so every timed transform would carry a launch + host synchronisation round trip
        # (~12 us of a 0.1-0.2 ms transform, profiles/r04_ntt_host_overhead.log) that is not kernel time.  torch's
        # events below record on the current stream = this one.
        ntt_stream = torch.cuda.Stream()
        torch.cuda.synchronize(); torch.cuda.set_stream(ntt_stream)
        stream = ntt_stream.cuda_stream
        Ord = sppark_amd.NTTInputOutputOrder
        ref = x.clone()
        for _ in range(3):
            sppark_amd.NTT(0, x, Ord.NR, "gl64", stream=stream); sppark_amd.iNTT(0, x, Ord.RN, "gl64", stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(x, ref), "NTT round trip failed"
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps = 20
        # forward NR transforms back to back, then inverse RN ones (HIP events on the launch stream);
        # x goes through reps forward + reps inverse transforms and ends where it started
        fwd = inv = 1e30
        for _batch in range(3):                                 # best of 3 batches: the section starts right after
            e0.record()                                         # two seconds of MSM load, clocks settle during the first
            for _ in range(reps):
                sppark_amd.NTT(0, x, Ord.NR, "gl64", stream=stream)
            e1.record()
            for _ in range(reps):
                sppark_amd.iNTT(0, x, Ord.RN, "gl64", stream=stream)
            e2.record(); torch.cuda.synchronize()
            fwd = min(fwd, e0.elapsed_time(e1) / reps); inv = min(inv, e1.elapsed_time(e2) / reps)
        assert torch.equal(x, ref), "timed NTT sequence did not return to its input"
        sppark_amd.NTT(0, x, Ord.NN, "gl64", stream=stream)          # (first use of the bit-reversal kernel: untimed)
        e0.record()
        for _ in range(reps):
            sppark_amd.NTT(0, x, Ord.NN, "gl64", stream=stream)      # natural in, natural out (adds the bit reversal)
        e1.record(); torch.cuda.synchronize()
        fwd_nn = e0.elapsed_time(e1) / reps
        # coset NR (ntt/ntt.cuh:197-207 scales by the powers of the coset generator in a separate kernel; here they ride in
        # the passes' twiddle tables, csrc/ntt/ntt_r64_kernels.hpp r64_coset_mode): same protocol
        yc = ref.clone()
        sppark_amd.coset_NTT(0, yc, Ord.NR, "gl64", stream=stream)
        coset_dev = yc.clone()                                  # (checked below; copied to the host AFTER the timing: an idle
        coset_nr = 1e30                                         # device clocks down during a synchronous copy)
        for _batch in range(3):
            e0.record()
            for _ in range(reps):
                sppark_amd.coset_NTT(0, yc, Ord.NR, "gl64", stream=stream)
            e1.record(); torch.cuda.synchronize()
            coset_nr = min(coset_nr, e0.elapsed_time(e1) / reps)
        coset_host = coset_dev.cpu().numpy().view(np.uint64)
        del yc, coset_dev
        # CHECKER: the timed transform at the timed size against the oracle, whole array
        y = ref.clone()
        sppark_amd.NTT(0, y, Ord.NR, "gl64", stream=stream); torch.cuda.synchronize()
        y_host = y.cpu().numpy().view(np.uint64)
        y_ref = O.ntt_gl64(ref.cpu().numpy().view(np.uint64), O.NR)
        ntt_ok = bool((y_host == y_ref).all())
        sppark_amd.iNTT(0, y, Ord.RN, "gl64", stream=stream); torch.cuda.synchronize()
        ntt_ok = ntt_ok and bool(torch.equal(y, ref))
        assert ntt_ok, "timed NTT differs from the oracle"
        coset_ok = bool((coset_host == O.ntt_gl64(ref.cpu().numpy().view(np.uint64), O.NR, O.FORWARD, O.COSET)).all())
        assert coset_ok, "timed coset NTT differs from the oracle"
        # SURVEY 8(d) timing protocol (ii), through-the-FFI: compute_ntt on a HOST buffer, what every caller of the
        # reference hits (poc/ntt-cuda/src/lib.rs:7-118 -> ntt/ntt.cuh:215-244: H2D, transform, D2H); wall clock,
        # pageable numpy memory as a Rust Vec / Go slice is; output asserted against the device path checked above
        torch.cuda.synchronize(); torch.cuda.set_stream(torch.cuda.default_stream())
        through_ffi = host_ntt_time("gl64", ref.cpu().numpy().view(np.uint64), y_host, lg)
        # BASELINE beside it: the reference's OWN NTT (its HIP path, built for gfx950 from /root/reference into
        # oracle/_ref/libref_ntt_gl64.so by `make -C oracle ref_ntt`; test infrastructure, prebuilt, not the product) on
        # the same device-resident array: NTT::Base_dev_ptr (ntt/ntt.cuh:344-350), events on ITS stream around the same
        # number of back-to-back transforms, best of 3 batches; its output on the timed input == ours, bit for bit
        ref_build = None
        if O.ref_ntt_available("gl64"):
            xr = ref.clone(); torch.cuda.synchronize()
            O.ref_ntt_dev("gl64", xr.data_ptr(), lg, 1, 0, 0); torch.cuda.synchronize()
            same = bool((xr.cpu().numpy().view(np.uint64) == y_host).all())
            assert same, "the reference's own NTT and sppark_amd's differ on the timed input"
            xr.copy_(ref); torch.cuda.synchronize()
            O.ref_ntt_dev("gl64", xr.data_ptr(), lg, 1, 0, 1); torch.cuda.synchronize()
            same_coset = bool((xr.cpu().numpy().view(np.uint64) == coset_host).all())
            assert same_coset, "the reference's own coset NTT and sppark_amd's differ on the timed input"
            r_fwd, r_inv, r_nn = (min(O.ref_ntt_dev_ms("gl64", xr.data_ptr(), lg, o, d_, 0, reps) for _ in range(3))
                                  for o, d_ in ((1, 0), (2, 1), (0, 0)))
            r_coset = min(O.ref_ntt_dev_ms("gl64", xr.data_ptr(), lg, 1, 0, 1, reps) for _ in range(2))
            ref_build = {"what": "supranational/sppark's own NTT through its HIP path (hipcc -include util/cuda2hip.hpp, gfx950), "
                                 "same box, same device-resident array, NTT::Base_dev_ptr",
                         "forward_ms": r_fwd, "inverse_ms": r_inv, "forward_nn_ms": r_nn, "coset_nr_ms": r_coset,
                         "output_equals_ours": same, "coset_output_equals_ours": same_coset,
                         "speedup_forward": r_fwd / fwd, "speedup_inverse": r_inv / inv, "speedup_forward_nn": r_nn / fwd_nn,
                         "speedup_coset_nr": r_coset / coset_nr}
            del xr
        # HBM traffic of one transform: a CONSTANT from the committed rocprofv3 --pmc passes of this workload
        # (tools/make_ntt_pmc_traffic.py), newest round first; null for any other size
        ntt_traffic, ntt_traffic_note = None, ""
        for fn in ("r06_ntt_gl64_pmc.json", "r05_ntt_gl64_pmc.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    pmc = json.load(f)
                if pmc.get("lg") == lg and pmc.get("field") == "gl64":
                    ntt_traffic = (pmc["fetch_bytes"] + pmc["write_bytes"]) / 1e9
                    ntt_traffic_note = ("; traffic = the transform's launches' FETCH_SIZE (x2) + WRITE_SIZE from the committed rocprofv3 --pmc "
                                        "passes (profiles/%s: %.2f x the algorithmic bytes; a recorded constant, not measured in this run)"
                                        % (fn, pmc["traffic_over_algorithmic"]))
                    break
            except (OSError, ValueError, KeyError):
                pass
        ntt = {"metric": "Goldilocks NTT 2^%d elements/s (forward NR / inverse RN, device-resident)" % lg,
               "input": {"elements": 1 << lg, "values": "torch.randint(0, 2^62) on the device, generator seed 2: 2^%d independent values, all distinct positions" % lg,
                         "seed": 2},
               "timing": "HIP events around 20 back-to-back transforms on a non-null stream, best of 3 batches",
               "forward_ms": fwd, "inverse_ms": inv, "forward_nn_ms": fwd_nn, "coset_nr_ms": coset_nr,
               "coset_equals_oracle": coset_ok,
               "through_ffi": through_ffi,
               "reference_hip_build": ref_build,
               "forward_elems_per_s": (1 << lg) / (fwd * 1e-3), "inverse_elems_per_s": (1 << lg) / (inv * 1e-3),
               "pair_elems_per_s": (1 << lg) / ((fwd + inv) * 1e-3),
               "equals_oracle": ntt_ok, "output_sha256": hashlib.sha256(y_host.tobytes()).hexdigest(),
               "roofline": {"bound": "hbm", "achieved": NTT_BYTES_PER_ELEM * (1 << lg) / (fwd * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": NTT_BYTES_PER_ELEM * (1 << lg) / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": ntt_traffic, "traffic_unit": "GB per transform",
                            "note": "whole forward transform vs 16 B/element algorithmic (one read + one write of the array)" + ntt_traffic_note}}

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        # BASELINE configs[4]: alt_bn128 G1 MSM + BabyBear NTT (multi-field instantiation)
        bpts, bbase = synth.replicated_points(n, "bn254", PERIOD, 7)
        bsc = synth.uniform_scalars(n, "bn254", 7)
        # bases kept in the context (the reference's msm_t(points) + invoke(out, scalars),
        # msm/pippenger.cuh:351-385,604-605): the one-time conversion of the points into the
        # kernels' own records is then outside the call; NOT the headline value, which pays it
        ctx.set_points(pts)
        ctx.invoke(None, sc)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3):
            ctx.invoke(None, sc)
        torch.cuda.synchronize()
        extras["bls12_381_g1_msm_preloaded_bases_points_per_s"] = 3 * n / (time.perf_counter() - t1)
        # (the fixed-base mode of the same bases -- a 4.9 s one-time table build -- is measured by tools/gpu_msm_fixed.py,
        # profiles/r05_msm_fixed_base.log, not on every driver run)
        ctx.set_points(None)
        bctx = sppark_amd.MsmContext("bn254", device_id=-1, stream=torch.cuda.current_stream().cuda_stream)
        bctx.enable_timing(True)
        bout = bctx.invoke(bpts, bsc)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        b_acc = []
        for _ in range(3):
            bout = bctx.invoke(bpts, bsc)
            b_acc.append(bctx.kernel_ms(1))
        torch.cuda.synchronize()
        extras["alt_bn128_g1_msm_points_per_s"] = 3 * n / (time.perf_counter() - t1)
        # configs[4]: the same two rooflines for the nine-limb pipeline (96 B per point: 64-byte affine point + 32-byte scalar)
        b_ms, b_w = float(np.mean(b_acc)), bctx.plan(n)["windows"]
        b_mads = float(b_w) * n * MADS_PER_MIXED_ADD_9
        extras["alt_bn128_g1_msm_roofline"] = {
            "kernel": "k_accumulate", "kernel_ms": b_ms, "windows": b_w,
            "hbm": {"achieved": 96 * n / (b_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 96 * n / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "alu": {"bound": "v_mad_u64_u32 issue", "achieved": b_mads / (b_ms * 1e-3) / 1e12, "peak": MAD_PEAK_PER_S / 1e12,
                    "unit": "T mad/s", "frac": b_mads / (b_ms * 1e-3) / MAD_PEAK_PER_S,
                    "note": "%d windows x points x %d multiply-adds (8 products + 2 squares + 9 reductions on 9 limbs of 29 bits; "
                            "round 4: 10 limbs of 28 bits, 2010 multiply-adds); three waves per SIMD" % (b_w, MADS_PER_MIXED_ADD_9)}}
        bexp = O.msm_affine(O.BN254, bbase.cpu().numpy(), fold.fold_scalars(bsc, PERIOD, O.FR_MODULUS[O.BN254]), algo=0, param=8)
        extras["alt_bn128_g1_msm_equals_oracle"] = bool((sppark_amd.to_affine(bout, "bn254") == bexp).all())
        assert extras["alt_bn128_g1_msm_equals_oracle"]
        bctx.close(); del bpts, bsc
        y = torch.randint(0, 0x78000000, (1 << args.ntt_lg,), dtype=torch.int32, device="cuda")
        y0 = y.clone()
        ntt_stream = torch.cuda.Stream()                        # (non-null, as for the Goldilocks timings above)
        torch.cuda.synchronize(); torch.cuda.set_stream(ntt_stream)
        stream = ntt_stream.cuda_stream
        sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NR, "bb31", stream=stream); torch.cuda.synchronize()
        bb_out = y.cpu().numpy().view(np.uint32)
        extras["babybear_ntt_equals_oracle"] = bool((bb_out == O.ntt_bb31(y0.cpu().numpy().view(np.uint32), O.NR)).all())
        assert extras["babybear_ntt_equals_oracle"]
        extras["babybear_ntt_through_ffi"] = host_ntt_time("bb31", y0.cpu().numpy().view(np.uint32), bb_out, args.ntt_lg)
        for _ in range(3):
            sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NR, "bb31", stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NR, "bb31", stream=stream)
        e1.record(); torch.cuda.synchronize()
        extras["babybear_ntt_elems_per_s"] = 20 * (1 << args.ntt_lg) / (e0.elapsed_time(e1) * 1e-3)
        sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NN, "bb31", stream=stream)      # (first use: untimed)
        e0.record()
        for _ in range(20):
            sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NN, "bb31", stream=stream)
        e1.record(); torch.cuda.synchronize()
        extras["babybear_ntt_forward_nr_ms"] = (1 << args.ntt_lg) / extras["babybear_ntt_elems_per_s"] * 1e3
        extras["babybear_ntt_forward_nn_ms"] = e0.elapsed_time(e1) / 20
        torch.cuda.set_stream(torch.cuda.default_stream())
        if O.ref_ntt_available("bb31"):                         # the reference's own build beside it, as for Goldilocks
            torch.cuda.synchronize()
            extras["babybear_ntt_reference_hip_build"] = {
                "forward_nr_ms": min(O.ref_ntt_dev_ms("bb31", y.data_ptr(), args.ntt_lg, 1, 0, 0, 20) for _ in range(3)),
                "forward_nn_ms": min(O.ref_ntt_dev_ms("bb31", y.data_ptr(), args.ntt_lg, 0, 0, 0, 20) for _ in range(3))}
        # the "next" rows of SURVEY 8(f): G2 MSM, 256-bit-field NTT, low-degree extension (--all-extras)
        try:
            if not args.all_extras:
                raise StopIteration
            # input points: the 33 G2 points of a committed golden case (data fixture), replicated
            with open(os.path.join(ROOT, "tests", "golden", "msm_g2_golden.json")) as f:
                case = [c for c in json.load(f) if c["curve"] == "bls12_381" and c["n"] == 33 and "points" in c][0]
            g2 = np.frombuffer(bytes.fromhex(case["points"]), dtype=np.uint8).reshape(33, -1)
            lg2 = min(args.lg, 22)
            g2pts = torch.from_numpy(g2[np.arange(1 << lg2) % 33].copy()).cuda()
            g2sc = sc[:1 << lg2].contiguous()
            sppark_amd.multi_scalar_mult_fp2_arkworks(g2pts, g2sc)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                sppark_amd.multi_scalar_mult_fp2_arkworks(g2pts, g2sc)
            torch.cuda.synchronize()
            g2_s = (time.perf_counter() - t1) / 3
            extras["bls12_381_g2_msm_2^%d_points_per_s" % lg2] = (1 << lg2) / g2_s
            # the accumulation by wave pairs (msm/msm_g2c_kernels.hpp; round 5) against the same multiply-add peak.  The entry
            # point is the reference's one-shot mult_pippenger_fp2_inf, which has no per-kernel timer: the WHOLE call is the
            # denominator, so the fraction is a lower bound of the accumulation kernel's own
            g2_w = ctx.plan(1 << lg2)["windows"]                 # (the G2 plan is make_plan() of the same point count)
            g2_mads = float(g2_w) * (1 << lg2) * MADS_PER_G2_MIXED_ADD
            extras["bls12_381_g2_msm_roofline_alu"] = {
                "bound": "v_mad_u64_u32 issue", "achieved": g2_mads / g2_s / 1e12, "peak": MAD_PEAK_PER_S / 1e12, "unit": "T mad/s",
                "frac": g2_mads / g2_s / MAD_PEAK_PER_S, "ms_per_msm": g2_s * 1e3, "windows": g2_w,
                "note": "%d windows x points x %d multiply-adds (8 Fp2 products + 2 Fp2 squares, one Fp2 component per wave) over the "
                        "wall clock of the whole call (sort, conversion, bucket sums included): a lower bound for k_accumulate_g2c"
                        % (g2_w, MADS_PER_G2_MIXED_ADD)}
            del g2pts, g2sc
        except StopIteration:
            pass
        except Exception as ex:                                 # noqa: BLE001  (extras never fail the bench)
            extras["bls12_381_g2_msm_error"] = repr(ex)[:200]
        torch.cuda.synchronize(); torch.cuda.set_stream(ntt_stream)
        stream = ntt_stream.cuda_stream
        wlg = min(args.ntt_lg, 22) if args.all_extras else 0
        if wlg:
            wx = torch.randint(0, 2**62, ((1 << wlg) * 4,), dtype=torch.int64, device="cuda"); wx[3::4] &= 0x0fffffffffffffff
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):
                sppark_amd.NTT(0, wx, sppark_amd.NTTInputOutputOrder.NR, "bls12_381", stream=stream)
            e0.record()
            for _ in range(5):
                sppark_amd.NTT(0, wx, sppark_amd.NTTInputOutputOrder.NR, "bls12_381", stream=stream)
            e1.record(); torch.cuda.synchronize()
            extras["bls12_381_fr_ntt_2^%d_elems_per_s" % wlg] = 5 * (1 << wlg) / (e0.elapsed_time(e1) * 1e-3)
            del wx
            llg = min(args.ntt_lg, 22)
            ext = torch.randint(0, 2**62, (1 << (llg + 2),), dtype=torch.int64, device="cuda")
            for _ in range(2):
                sppark_amd.LDE(0, ext, llg, 2, "gl64", stream=stream)
            e0.record()
            for _ in range(5):
                sppark_amd.LDE(0, ext, llg, 2, "gl64", stream=stream)
            e1.record(); torch.cuda.synchronize()
            extras["goldilocks_lde_2^%d_to_2^%d_ms" % (llg, llg + 2)] = e0.elapsed_time(e1) / 5
            del ext
        torch.cuda.set_stream(torch.cuda.default_stream())
        # through-the-FFI path with HOST buffers (PCIe inclusive; never the headline value): what the
        # reference's Rust / Go callers use.  2^24 and the full 2^26, chunked copy under the arithmetic.
        h2d, d2h, h2d_pageable = pcie_peaks()
        extras["h2d_peak_gbs"], extras["d2h_peak_gbs"], extras["h2d_pageable_gbs"] = h2d, d2h, h2d_pageable
        extras["pcie_note"] = ("one 1 GiB hipMemcpy each way between PINNED host memory and the device, HIP events, best of 4; "
                               "h2d_pageable_gbs = the same copy from pageable memory (wall clock), which is what the host-buffer entry points receive")
        for key in ("babybear_ntt_through_ffi",):
            extras[key]["frac_of_h2d_peak"] = extras[key]["copy_gb_per_s"] / h2d
        if ntt is not None:
            ntt["through_ffi"]["frac_of_h2d_peak"] = ntt["through_ffi"]["copy_gb_per_s"] / h2d
        host = {}
        for lgh in sorted({min(args.lg, 24), args.lg}):
            m = 1 << lgh
            hp = np.zeros((m, 104), dtype=np.uint8); hp[:, :96] = pts[:m].cpu().numpy()
            hp[3::PERIOD, 96] = 1                                # Affine_inf_t: the flag byte marks infinity
            hs = sc[:m].cpu().numpy()
            # first call at this size: the pooled context grows its scratch and staging buffers inside the call
            # (hipFree + hipMalloc of GBs); the reported figure is the steady state a prover sees from its second
            # proof on, the first call's time is kept beside it
            t1 = time.perf_counter()
            sppark_amd.multi_scalar_mult_arkworks(hp, hs)
            dt_first = time.perf_counter() - t1
            dt = 1e30
            for _ in range(2):
                t1 = time.perf_counter()
                hout = sppark_amd.multi_scalar_mult_arkworks(hp, hs)
                dt = min(dt, time.perf_counter() - t1)
            hexp = O.msm_affine(O.BLS12_381, base.cpu().numpy(), fold.fold_scalars(sc[:m], PERIOD, r_mod), algo=0, param=8)
            ok = bool((sppark_amd.to_affine(hout) == hexp).all())
            assert ok, "host-buffer MSM differs from the oracle"
            host["2^%d" % lgh] = {"points": m, "seconds": dt, "seconds_first_call": dt_first, "points_per_s": m / dt, "equals_oracle": ok,
                                  "input_gb_per_s": m * (104 + 32) / dt / 1e9, "frac_of_h2d_peak": m * (104 + 32) / dt / 1e9 / h2d,
                                  "frac_of_h2d_pageable": m * (104 + 32) / dt / 1e9 / h2d_pageable}
            del hp, hs
        extras["mult_pippenger_inf_host_buffers"] = host
        # the per-GPU shards of the headline MSM at N = 2 / 4 / 8 on THIS GPU (device-resident, same inputs): an upper
        # bound for strong scaling read off a 1-GPU line (t_shard, before the all-gather); each result asserted
        shard = {}
        # (without the library's phase timers -- an event record per phase, ~20 us per MSM at 2^16, profiles/r05_msm_timing_overhead.log:
        # they exist for the roofline of the headline above, a caller's context has them off)
        ctx.enable_timing(False)
        for lgs in (25, 24, 23, 20, 16):
            if lgs >= args.lg:
                continue
            m = 1 << lgs
            sp, ss = pts[:m], sc[:m]
            sout = ctx.invoke(sp, ss)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            reps = 3 if lgs >= 23 else 20
            for _ in range(reps):
                sout = ctx.invoke(sp, ss)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / reps * 1e3
            sexp = O.msm_affine(O.BLS12_381, base.cpu().numpy(), fold.fold_scalars(ss, PERIOD, r_mod), algo=0, param=8)
            ok = bool((sppark_amd.to_affine(sout) == sexp).all())
            assert ok, "shard-size MSM differs from the oracle"
            shard["msm_ms_at_2^%d" % lgs] = {"ms": ms, "points_per_s": m / (ms * 1e-3), "windows": ctx.plan(m)["windows"], "equals_oracle": ok}
        extras["shard_sizes"] = shard
        # the headline size on points that do NOT repeat (poc/msm-cuda/tests/msm.rs:19-39 checks against an arbitrary-point
        # oracle; the BASELINE shape above has 2^11 distinct points): P_i = (a + i b) G generated on the device, the same
        # uniform scalars, the result against (sum s_i (a + i b) mod r) G -- integer arithmetic + one oracle scalar multiplication
        try:
            pa, pb = 0x243f6a8885a308d313198a2e03707344, 0xa4093822299f31d0082efa99
            dpts = torch.empty((n, 96), dtype=torch.uint8, device="cuda")
            t1 = time.perf_counter()
            sppark_amd.generate_progression(dpts, n, pa, pb, 96, "bls12_381")
            gen_s = time.perf_counter() - t1
            dout = ctx.invoke(dpts, sc)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                dout = ctx.invoke(dpts, sc)
            torch.cuda.synchronize()
            d_ms = (time.perf_counter() - t1) / 3 * 1e3
            s0, s1 = fold.weighted_sums(sc)
            dexp = O.g1_mul(O.BLS12_381, O.g1_generator(O.BLS12_381), (pa * s0 + pb * s1) % r_mod)
            ok = bool((sppark_amd.to_affine(dout) == dexp).all())
            assert ok, "MSM over all-distinct points differs from the known discrete logarithm"
            extras["all_distinct_points"] = {"distinct_points": n, "ms": d_ms, "points_per_s": n / (d_ms * 1e-3), "equals_known_discrete_log": ok,
                                             "generate_s": gen_s,
                                             "what": "P_i = (a + i b) G for i < n, device-resident, the headline's scalars; expected (sum s_i (a + i b) mod r) G"}
            del dpts
        except AssertionError:
            raise
        except Exception as ex:                                 # noqa: BLE001
            extras["all_distinct_points_error"] = repr(ex)[:200]
        ctx.enable_timing(True)
        sppark_amd.ffi.load("bls12_381").sppark_msm_release_cached()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = len(os.sched_getaffinity(0))
        # oracle/_ref (the reference's own msm/pippenger.hpp + util/thread_pool_t.hpp, compiled in place
        # by oracle/Makefile over the oracle's portable field; prebuilt, travels with the snapshot) when it
        # is there, otherwise the restatement ("port")
        use_ref = O.ref_available()
        if use_ref:
            cpu_msm = lambda p_, s_: O.ref_msm_affine(O.BLS12_381, p_, s_, nthreads=cores)
        else:
            cpu_msm = lambda p_, s_: O.msm_affine(O.BLS12_381, p_, s_, algo=0, param=cores)
        # size the sample for ~10 s of CPU work from a 2^16 probe (config 1 of BASELINE.json)
        hp = pts[:1 << 16].cpu().numpy(); hs = sc[:1 << 16].cpu().numpy()
        probe = 1e30
        for _ in range(2):                                      # (the first call also starts the thread pool)
            t1 = time.perf_counter()
            cpu_msm(hp, hs)
            probe = min(probe, time.perf_counter() - t1)
        lgm = 16
        while lgm < min(args.lg, 24) and probe * (1 << (lgm + 1 - 16)) * 0.6 < 6.0:      # (~10 s of CPU work: 2^23 points on 256 threads)
            lgm += 1
        m = 1 << lgm
        hp = pts[:m].cpu().numpy(); hs = sc[:m].cpu().numpy()
        t1 = time.perf_counter()
        ref = cpu_msm(hp, hs)
        dt = time.perf_counter() - t1
        got = sppark_amd.to_affine(ctx.invoke(pts[:m], sc[:m]))
        cpu = {"value": m / dt, "unit": "points/s", "cores": cores, "cpu_model": cpu_model(),
               "at_2^16": {"seconds": probe, "points_per_s": (1 << 16) / probe,
                           "what": "BASELINE configs[0]: the first 2^16 points through the same CPU path, %d threads" % cores},
               "kind": "reference" if use_ref else "port",
               "sample": "first 2^%d points of the same workload, %s (portable C++ field, not blst asm), %d threads, %.2f s"
                         % (m.bit_length() - 1,
                            "the reference's msm/pippenger.hpp + thread_pool_t compiled in place (oracle/_ref)" if use_ref
                            else "oracle restatement of msm/pippenger.hpp", cores, dt),
               "parity_with_gpu_on_sample": bool((got == ref).all())}
        assert cpu["parity_with_gpu_on_sample"], "GPU result differs from the CPU baseline on the sample"

    if rank == 0:
        a_ms = float(np.mean(accum_ms))
        achieved = MSM_BYTES_PER_POINT * n / (a_ms * 1e-3) / 1e9
        plan = ctx.plan(n)
        nwins = plan["windows"]
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the timed run, so the
        # value is a CONSTANT read from the committed rocprofv3 --pmc passes of this same workload
        # (profiles/r06_pmc_traffic.json, else r05 / r04 / ...), not an in-run measurement; null for any other workload.
        traffic, traffic_note = None, ""
        for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    pmc = json.load(f)
                if pmc.get("lg") == n.bit_length() - 1 and pmc.get("curve") == "bls12_381" and n & (n - 1) == 0:
                    traffic = (pmc["fetch_bytes"] + pmc["write_bytes"]) / 1e9
                    traffic_note = ("; traffic = GB per MSM (all k_accumulate launches of one step) from the committed rocprofv3 "
                                    "--pmc FETCH_SIZE + WRITE_SIZE passes (profiles/%s; a recorded constant, not measured in this run)" % fn)
                    break
            except (OSError, ValueError, KeyError):
                pass
        mads = float(nwins) * n * MADS_PER_MIXED_ADD
        xchg = "RCCL" if args.backend == "nccl" else "gloo (DRY RUN: ranks share a GPU, not a performance number)"
        line = {
            "metric": "MSM points/sec (BLS12-381 G1, 2^%d points%s)" % (total.bit_length() - 1, " in total over %d GPUs" % world if world > 1 else ""),
            "value": total * args.steps / elapsed, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": float(np.median(step_ms)), "ms_per_step_min": float(min(step_ms)),
            "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BLS12-381 G1 Pippenger MSM, 2^%d points in total, %d per GPU, device-resident inputs (%s)"
                                   % (total.bit_length() - 1, n,
                                      "BASELINE configs[2]" if world == 1 and total == 1 << 26 else
                                      "BASELINE configs[3]: sharded x%d, %s all-gather of partial sums" % (world, xchg) if total == 1 << 28 else
                                      "BASELINE metric: the 2^26 MSM sharded x%d, %s all-gather of partial sums" % (world, xchg) if total == 1 << 26 else
                                      "sharded x%d, %s all-gather of partial sums" % (world, xchg)),
                       "backend": args.backend if use_dist else None,
                       "curve": "bls12_381", "points_total": total, "points_per_gpu": n,
                       "window_bits": plan["window_bits"], "windows": nwins, "window_groups": acc_launches,
                       "distinct_points": PERIOD, "scalars": "uniform on [0, r), rejection sampled"},
            "parity": parity,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "GB per step (all launches)",
                         "algorithmic_gb_per_step": MSM_BYTES_PER_POINT * n / 1e9, "kernel": "k_accumulate",
                         "kernel_ms": a_ms, "launches_per_step": acc_launches, "kernel_ms_per_launch": a_ms / max(1, acc_launches),
                         "note": "one step runs k_accumulate once per window group (%d launches, each all points x 1/%d of the "
                                 "windows = 1/%d of the 128 B/point); achieved = 128 B x points / summed launch time. MSM is "
                                 "integer-multiplier bound, not HBM bound (SURVEY F11): see roofline_alu" % (acc_launches, acc_launches, acc_launches) + traffic_note},
            "roofline_alu": {"bound": "v_mad_u64_u32 issue", "achieved": mads / (a_ms * 1e-3) / 1e12, "peak": MAD_PEAK_PER_S / 1e12,
                             "unit": "T mad/s (32x32+64 multiply-adds, lane level)", "frac": mads / (a_ms * 1e-3) / MAD_PEAK_PER_S,
                             "frac_of_peak_at_kernel_occupancy": mads / (a_ms * 1e-3) / (MAD_RATE_2WAVES * 1024 * 64 * 2.4e9),
                             "note": "%d windows x points mixed additions x %d multiply-adds each (8 products + 2 squares + 9 "
                                     "reductions on 14 limbs of 28 bits) against the measured issue peak of the instruction, "
                                     "%.3f wave-instr/clk/SIMD at 8 waves per SIMD x 1024 SIMDs x 64 lanes x 2.4 GHz "
                                     "(profiles/r02_ubench_instruction_rates.log); the kernel's 230 registers allow 2 waves per "
                                     "SIMD, where the same micro-benchmark reaches %.3f, and one in five of its instructions is "
                                     "not a multiply-add" % (nwins, MADS_PER_MIXED_ADD, MAD_RATE_PEAK, MAD_RATE_2WAVES)},
            "phases_ms": {"before_first_accumulate": float(np.mean(sort_ms)), "accumulate": a_ms, "device_total": float(np.mean(dev_ms))},
            "cpu_baseline": cpu, "ntt": ntt, "extras": extras,
        }
        print(json.dumps(line))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
