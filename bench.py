#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W

Headline: BLS12-381 G1 Pippenger MSM, 2^26 points per GPU, inputs resident in
HBM (BASELINE.json metric / configs[2]; weak scaling: every rank owns 2^26 points
of a 2^26*N-point MSM, one RCCL all-gather of the 144-byte partial results per
step, combine on every rank).  A step = one full MSM.  Secondary (same JSON line,
key "ntt"): Goldilocks NTT 2^24 forward NR + inverse RN (configs[1]).

One JSON line on rank 0; "roofline" is for the dominant kernel k_accumulate
(HIP-event timed inside the library on the launch stream), "cpu_baseline" is the
oracle's restatement of msm/pippenger.hpp on the host cores (N=1 only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import sppark_amd
from sppark_amd import multi_gpu

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MSM_BYTES_PER_POINT = 128           # 96-byte affine point + 32-byte scalar
NTT_BYTES_PER_ELEM = 16             # read + write one u64 per transform


def make_msm_inputs(lg, seed, curve="bls12_381", fb=48):
    """poc/msm-cuda/src/util.rs:11-38 shape: 2^11 distinct points replicated,
    index 3 = infinity, independent uniform scalars (254-bit, all < r)."""
    n = 1 << lg
    base = torch.zeros((2048, 2 * fb), dtype=torch.uint8, device="cuda")
    sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 2 * fb, curve)
    pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
    pts[3] = 0
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    sc[:, 31] &= 0x3f
    return pts, sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lg", type=int, default=26, help="log2 points per GPU (default: the BASELINE size)")
    ap.add_argument("--ntt-lg", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    # SPPARK_FORCE_DIST=1 exercises the RCCL exchange with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("SPPARK_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert args.gpus == world, "--gpus must equal WORLD_SIZE"

    n = 1 << args.lg
    pts, sc = make_msm_inputs(args.lg, 0x5eed5eed0001 + rank)
    ctx = sppark_amd.MsmContext("bls12_381", device_id=-1, stream=torch.cuda.current_stream().cuda_stream)
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
    accum_ms, sort_ms, dev_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
        accum_ms.append(ctx.kernel_ms(1)); sort_ms.append(ctx.kernel_ms(0)); dev_ms.append(ctx.kernel_ms(2))
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ntt = None
    if rank == 0 and not args.no_ntt:
        lg = args.ntt_lg
        g = torch.Generator(device="cuda"); g.manual_seed(2)
        x = (torch.randint(0, 2**62, (1 << lg,), dtype=torch.int64, device="cuda", generator=g))   # < p
        stream = torch.cuda.current_stream().cuda_stream
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
        e0.record()
        for _ in range(reps):
            sppark_amd.NTT(0, x, Ord.NN, "gl64", stream=stream)      # natural in, natural out (adds the bit reversal)
        e1.record(); torch.cuda.synchronize()
        fwd_nn = e0.elapsed_time(e1) / reps
        ntt = {"metric": "Goldilocks NTT 2^%d elements/s (forward NR / inverse RN, device-resident)" % lg,
               "timing": "HIP events around 20 back-to-back transforms, best of 3 batches",
               "forward_ms": fwd, "inverse_ms": inv, "forward_nn_ms": fwd_nn,
               "forward_elems_per_s": (1 << lg) / (fwd * 1e-3), "inverse_elems_per_s": (1 << lg) / (inv * 1e-3),
               "pair_elems_per_s": (1 << lg) / ((fwd + inv) * 1e-3),
               "roofline": {"bound": "hbm", "achieved": NTT_BYTES_PER_ELEM * (1 << lg) / (fwd * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": NTT_BYTES_PER_ELEM * (1 << lg) / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "whole forward transform (3 register-radix passes, 0.8 GB of traffic) vs 16 B/element algorithmic"}}

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        # BASELINE configs[4]: alt_bn128 G1 MSM + BabyBear NTT (multi-field instantiation)
        bpts, bsc = make_msm_inputs(args.lg, 7, "bn254", 32)
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
        ctx.set_points(None)
        bctx = sppark_amd.MsmContext("bn254", device_id=-1, stream=torch.cuda.current_stream().cuda_stream)
        bctx.invoke(bpts, bsc)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3):
            bctx.invoke(bpts, bsc)
        torch.cuda.synchronize()
        extras["alt_bn128_g1_msm_points_per_s"] = 3 * n / (time.perf_counter() - t1)
        bctx.close(); del bpts, bsc
        y = torch.randint(0, 0x78000000, (1 << args.ntt_lg,), dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NR, "bb31", stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            sppark_amd.NTT(0, y, sppark_amd.NTTInputOutputOrder.NR, "bb31", stream=stream)
        e1.record(); torch.cuda.synchronize()
        extras["babybear_ntt_elems_per_s"] = 20 * (1 << args.ntt_lg) / (e0.elapsed_time(e1) * 1e-3)
        # the "next" rows of SURVEY 8(f): G2 MSM, 256-bit-field NTT, low-degree extension
        try:
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
            extras["bls12_381_g2_msm_2^%d_points_per_s" % lg2] = 3 * (1 << lg2) / (time.perf_counter() - t1)
            del g2pts, g2sc
        except Exception as ex:                                 # noqa: BLE001  (extras never fail the bench)
            extras["bls12_381_g2_msm_error"] = repr(ex)[:200]
        stream = torch.cuda.current_stream().cuda_stream
        wlg = min(args.ntt_lg, 22)
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
        # through-the-FFI path with HOST buffers (PCIe inclusive; never the headline value)
        lgh = min(args.lg, 24)
        hp = np.zeros(((1 << lgh), 104), dtype=np.uint8); hp[:, :96] = pts[:1 << lgh].cpu().numpy()
        hs = sc[:1 << lgh].cpu().numpy()
        sppark_amd.multi_scalar_mult_arkworks(hp[:4096], hs[:4096])
        t1 = time.perf_counter()
        sppark_amd.multi_scalar_mult_arkworks(hp, hs)
        extras["mult_pippenger_inf_host_buffers"] = {"points": 1 << lgh, "seconds": time.perf_counter() - t1,
                                                     "points_per_s": (1 << lgh) / (time.perf_counter() - t1)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import oracle as O                                      # cpu_baseline leg only
        cores = len(os.sched_getaffinity(0))
        # oracle/_ref (the reference's own msm/pippenger.hpp + util/thread_pool_t.hpp, compiled in place
        # by oracle/Makefile over the oracle's portable field; prebuilt, travels with the snapshot) when it
        # is there, otherwise the restatement ("port")
        use_ref = O.ref_available()
        if use_ref:
            cpu_msm = lambda p_, s_: O.ref_msm_affine(O.BLS12_381, p_, s_, nthreads=cores)
        else:
            cpu_msm = lambda p_, s_: O.msm_affine(O.BLS12_381, p_, s_, algo=0, param=cores)
        # size the sample for ~15 s of CPU work from a 2^16 probe (config 1 of BASELINE.json)
        hp = pts[:1 << 16].cpu().numpy(); hs = sc[:1 << 16].cpu().numpy()
        t1 = time.perf_counter()
        cpu_msm(hp, hs)
        probe = time.perf_counter() - t1
        lgm = 16
        while lgm < min(args.lg, 24) and probe * (1 << (lgm + 1 - 16)) * 0.6 < 15.0:
            lgm += 1
        m = 1 << lgm
        hp = pts[:m].cpu().numpy(); hs = sc[:m].cpu().numpy()
        t1 = time.perf_counter()
        ref = cpu_msm(hp, hs)
        dt = time.perf_counter() - t1
        got = sppark_amd.to_affine(ctx.invoke(pts[:m], sc[:m]))
        cpu = {"value": m / dt, "unit": "points/s", "cores": cores, "kind": "reference" if use_ref else "port",
               "sample": "first 2^%d points of the same workload, %s (portable C++ field, not blst asm), %d threads, %.2f s"
                         % (m.bit_length() - 1,
                            "the reference's msm/pippenger.hpp + thread_pool_t compiled in place (oracle/_ref)" if use_ref
                            else "oracle restatement of msm/pippenger.hpp", cores, dt),
               "parity_with_gpu_on_sample": bool((got == ref).all())}

    if rank == 0:
        a_ms = float(np.mean(accum_ms))
        achieved = MSM_BYTES_PER_POINT * n / (a_ms * 1e-3) / 1e9
        plan = ctx.plan(n)
        nwins = plan["windows"]
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the
        # timed run, so the value comes from the committed rocprofv3 --pmc passes of this
        # same workload (profiles/r01_pmc_traffic.json); null for any other workload.
        traffic, traffic_note = None, ""
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc.get("lg") == args.lg and pmc.get("curve") == "bls12_381":
                traffic = (pmc["fetch_bytes"] + pmc["write_bytes"]) / 1e9
                traffic_note = ("; traffic = GB per launch from rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes, "
                                "profiles/r01_pmc_traffic.json): the 112-byte point-record gathers pull whole 128-byte lines, "
                                "hidden behind the multiplier-bound arithmetic")
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "MSM points/sec (BLS12-381 G1, 2^%d points per GPU)" % args.lg,
            "value": world * n * args.steps / elapsed, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BLS12-381 G1 Pippenger MSM, 2^%d points per GPU, device-resident inputs "
                                   "(BASELINE configs[2]%s)" % (args.lg, "; sharded x%d with RCCL all-gather of partial sums" % world if world > 1 else ""),
                       "curve": "bls12_381", "points_per_gpu": n, "window_bits": plan["window_bits"], "windows": nwins,
                       "distinct_points": 2048, "scalars": "uniform 254-bit"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "GB per launch",
                         "algorithmic_gb_per_launch": achieved * a_ms * 1e-3, "kernel": "k_accumulate",
                         "kernel_ms": a_ms,
                         "note": "MSM is integer-multiplier bound, not HBM bound (SURVEY F11): the kernel does "
                                 "%d mixed additions per launch = %.3e additions/s against 7.29e9/s for the same "
                                 "addition chain on registers (profiles/r01_montx_vs_mont32.log); rocprofv3 PMC: the vector ALU "
                                 "issues 83 %% of the kernel's cycles, instruction-cache hit rate > 99.999 %% "
                                 "(profiles/r01_accumulate_pmc.txt)"
                                 % (nwins * n, nwins * n / (a_ms * 1e-3)) + traffic_note},
            "phases_ms": {"digits_sort": float(np.mean(sort_ms)), "accumulate": a_ms, "device_total": float(np.mean(dev_ms))},
            "cpu_baseline": cpu, "ntt": ntt, "extras": extras,
        }
        print(json.dumps(line))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
