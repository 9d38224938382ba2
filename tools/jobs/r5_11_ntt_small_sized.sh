#!/bin/bash
# (Kept as the record of how profiles/r05_ntt_small_sized_ab.log was produced: the variant libraries lib_old / lib_b / lib_nosq / lib_tuning were
# built from switches of the tree of that moment -- two pairs per lane, squared twiddles, DPP selects -- which the job's result removed.)
# Round 5, job 11: k_ntt_small with (a) its twiddles from the level rows of ntt_tables::inner (consecutive lanes read
# consecutive entries), (b) instances compiled for one size (2^8 ... 2^11), (c) the in-wave exchanges at distance <= 8 as
# selects with a DPP source.  One box: parity first (the NTT GPU tests), then ours only for the library of the commit before
# (lib_old), the new one (lib), the new one without (c) (lib_b), BabyBear with every twiddle loaded (lib_nosq), the tuning
# build with two pairs per lane from 2^10 / 2^9, without (b), and with the 256-bit fields' limit at 2^10; then the new one
# against the reference's build in all four orders.
mkdir -p gpurun_out; out=gpurun_out/r5_11; : > $out.ntt_small.log
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out.tests.log
run() { # label, fields, order, env...
  local label=$1 fields=$2 order=$3; shift 3
  echo "== $label, order $order" | tee -a $out.ntt_small.log
  env "$@" timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours lgs=8-11 fields=$fields order=$order 2>&1 | grep "^gl64\|^bb31\|^bls12_381" | tee -a $out.ntt_small.log
}
for o in 1 2; do
  run "lib_old (the commit before)" gl64,bb31,bls12_381 $o SPPARK_LIBDIR=lib_old
  run "lib (a + b + c)" gl64,bb31,bls12_381 $o SPPARK_LIBDIR=lib
  run "lib_b (a + b)" gl64,bb31 $o SPPARK_LIBDIR=lib_b
done
run "lib_nosq (a + b + c, BabyBear loads every twiddle)" bb31 1 SPPARK_LIBDIR=lib_nosq
run "lib_tuning, run-time-size instance (a + c)" gl64,bb31 1 SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_SIZED=0
run "lib_tuning, two pairs per lane from 2^10" gl64,bb31 1 SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_Q2=10
run "lib_tuning, two pairs per lane from 2^9" gl64,bb31 1 SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_Q2=9
run "lib_tuning, one pair per lane at 2^11" gl64,bb31 1 SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_Q2=99
run "lib_tuning, 256-bit fields up to 2^10 in one work-group" bls12_381 1 SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_MAX=10
for o in 1 2 0 3; do
  echo "== lib against the reference's build, order $o" | tee -a $out.ntt_small.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o lgs=8-12 2>&1 | grep "^gl64\|^bb31\|^bls12_381\|^bn254\|rows" | tee -a $out.ntt_small.log
done
