"""PCIe-inclusive MSM (host buffers through mult_pippenger_inf / a context): chunk-size sweep with pageable
and with pinned source memory, and what hipHostRegister of the caller's buffers would cost.
    python tools/gpu_msm_host.py [LG ...]      (default 24 26)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sppark_amd
from sppark_amd import synth
for lg in ([int(a) for a in sys.argv[1:] if not a.startswith('-')] or [24, 26]):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1)
    hp = np.zeros((n, 104), dtype=np.uint8); hp[:, :96] = pts.cpu().numpy(); hp[3::2048, 96] = 1
    hs = synth.uniform_scalars(n, "bls12_381", 1).cpu().numpy()
    del pts
    gb = (hp.nbytes + hs.nbytes) / 1e9
    ctx = sppark_amd.MsmContext("bls12_381")
    ref = None
    def sweep(tag, P, S):
        global ref
        for chunk_lg in (0, 22, 23, 24, -5, -6, -7):               # negative: n / k points per chunk (k chunks)
            if chunk_lg > lg: continue
            cp = (1 << chunk_lg) if chunk_lg > 0 else (-(-n // -chunk_lg) if chunk_lg < 0 else 0)
            ctx.tune_pipeline(groups=1, chunk_points=cp)
            ctx.invoke(P, S, ffi_affine_sz=104)
            best = 1e9
            for _ in range(3):
                t = time.perf_counter(); out = ctx.invoke(P, S, ffi_affine_sz=104); best = min(best, time.perf_counter() - t)
            a = sppark_amd.to_affine(out)
            ref = a if ref is None else ref
            print("2^%d %s, chunk %s: %d chunks, %.1f ms, %.3e points/s, %.1f GB/s of input %s"
                  % (lg, tag, "auto" if not chunk_lg else "2^%d" % chunk_lg if chunk_lg > 0 else "n/%d" % -chunk_lg, ctx.last_chunks(), best * 1e3, n / best, gb / best,
                     "OK" if (a == ref).all() else "MISMATCH"), flush=True)
    sweep("pageable host buffers", hp, hs)
    if "--registered" not in sys.argv:
        ctx.close(); del hp, hs; continue
    # the caller's own buffers registered in place (what the library could do per call): cost of the registration
    rt = torch.cuda.cudart()
    for rep in range(2):
        t = time.perf_counter()
        e1 = rt.cudaHostRegister(hp.ctypes.data, hp.nbytes, 0); e2 = rt.cudaHostRegister(hs.ctypes.data, hs.nbytes, 0)
        treg = time.perf_counter() - t
        if rep == 0:
            sweep("registered (hipHostRegister) buffers", hp, hs)
        t = time.perf_counter()
        rt.cudaHostUnregister(hp.ctypes.data); rt.cudaHostUnregister(hs.ctypes.data)
        tun = time.perf_counter() - t
        print("2^%d hipHostRegister of %.1f GB: %.1f ms (%s %s), unregister %.1f ms" % (lg, gb, treg * 1e3, e1, e2, tun * 1e3), flush=True)
    ctx.close()
    t = time.perf_counter(); sppark_amd.multi_scalar_mult_arkworks(hp, hs); t1 = time.perf_counter() - t
    t = time.perf_counter(); sppark_amd.multi_scalar_mult_arkworks(hp, hs); t2 = time.perf_counter() - t
    print("2^%d mult_pippenger_inf one-shot (pageable): first %.1f ms, second %.1f ms" % (lg, t1 * 1e3, t2 * 1e3), flush=True)
    sppark_amd.ffi.load("bls12_381").sppark_msm_release_cached()
    del hp, hs
