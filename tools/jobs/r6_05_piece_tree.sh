#!/bin/bash
# Round 6, job 5: the piece tree of the small sizes: its GPU test (BLS12-381 only here: the other curves' libraries are
# rebuilt later), the small-size MSM tests around it, then the A/B (tune_tail 5 = the fan-in tree as before).
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest $R/tests/test_msm_gpu.py -x -q --timeout 600 -k "bls12_381 or (not bn254 and not bls12_377 and not pallas and not vesta)" > $O/r6_05_pytest.log 2>&1; tail -4 $O/r6_05_pytest.log
timeout 600 python $R/tools/gpu_msm_tail.py ab 10 12 14 15 16 17 18 > $O/r6_05_msm_sizes.log 2>&1; cat $O/r6_05_msm_sizes.log
