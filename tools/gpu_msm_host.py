"""PCIe-inclusive MSM (host buffers through mult_pippenger_inf / a context): chunk-size sweep.
python tools/gpu_msm_host.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sppark_amd
from sppark_amd import synth
for lg in (24, 26):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1)
    hp = np.zeros((n, 104), dtype=np.uint8); hp[:, :96] = pts.cpu().numpy(); hp[3::2048, 96] = 1
    hs = synth.uniform_scalars(n, "bls12_381", 1).cpu().numpy()
    del pts
    ctx = sppark_amd.MsmContext("bls12_381")
    ref = None
    for chunk_lg in (0, 20, 21, 22, 23, 24, lg):
        for groups in (0, 1):
            ctx.tune_pipeline(groups=groups, chunk_points=(1 << chunk_lg) if chunk_lg else 0)
            ctx.invoke(hp, hs, ffi_affine_sz=104)
            best = 1e9
            for _ in range(2):
                t = time.perf_counter(); out = ctx.invoke(hp, hs, ffi_affine_sz=104); best = min(best, time.perf_counter() - t)
            a = sppark_amd.to_affine(out)
            ref = a if ref is None else ref
            print("2^%d host buffers, chunk 2^%d groups %d: %d chunks, %.1f ms, %.3e points/s, scratch %.1f GB %s"
                  % (lg, chunk_lg, groups, ctx.last_chunks(), best * 1e3, n / best, ctx.scratch_bytes() / 1e9, "OK" if (a == ref).all() else "MISMATCH"), flush=True)
    ctx.close()
    t = time.perf_counter(); sppark_amd.multi_scalar_mult_arkworks(hp, hs); t1 = time.perf_counter() - t
    t = time.perf_counter(); sppark_amd.multi_scalar_mult_arkworks(hp, hs); t2 = time.perf_counter() - t
    print("2^%d mult_pippenger_inf one-shot: first %.1f ms, second %.1f ms" % (lg, t1 * 1e3, t2 * 1e3), flush=True)
    # pinned host memory for comparison
    tp = torch.from_numpy(hp).pin_memory(); ts = torch.from_numpy(hs).pin_memory()
    t = time.perf_counter(); sppark_amd.multi_scalar_mult_arkworks(tp.numpy(), ts.numpy()); t2 = time.perf_counter() - t
    t = time.perf_counter(); sppark_amd.multi_scalar_mult_arkworks(tp.numpy(), ts.numpy()); t3 = time.perf_counter() - t
    print("2^%d mult_pippenger_inf one-shot, pinned inputs: %.1f / %.1f ms" % (lg, t2 * 1e3, t3 * 1e3), flush=True)
    sppark_amd.ffi.load("bls12_381").sppark_msm_release_cached()
    del hp, hs, tp, ts
