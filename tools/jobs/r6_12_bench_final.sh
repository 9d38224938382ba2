#!/bin/bash
# Round 6, job 12: the default bench line on the final tree, timed as the driver runs it; the NTT part twice more.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
SECONDS=0
timeout 900 python $R/bench.py > $O/r6e_bench_final.json 2> $O/r6e_bench_final.err; echo "bench.py wall: $SECONDS s" | tee $O/r6e_bench_wall.txt
python -c "
import json; d=json.load(open('$O/r6e_bench_final.json')); n=d['ntt']
print(d['value'], d['ms_per_step'], 'ntt', n['forward_ms'], n['inverse_ms'], n['forward_nn_ms'], n['coset_nr_ms'], n['reference_hip_build']['coset_nr_ms'])"
for i in 1 2; do timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); n=d['ntt']; print('ntt', n['forward_ms'], n['inverse_ms'], n['forward_nn_ms'], n['coset_nr_ms'])"; done
