# gpurun_out/r6e_* (tools/jobs/r6_evidence.sh) -> the tracked names under profiles/
set -e
cd "$(dirname "$0")/.."
c() { cp "gpurun_out/$1" "profiles/$2"; }
c r6e_bench_final.json r06_bench_line.json
c r6e_bench_headline_rocprofv3_summary.txt r06_bench_headline_rocprofv3_summary.txt
c r6e_bench_rocprofv3_summary.txt r06_bench_rocprofv3_summary.txt
c r6e_pmc_traffic.json r06_pmc_traffic.json
c r6e_ntt_gl64_pmc.json r06_ntt_gl64_pmc.json
c pmc_msm_acc6.txt r06_msm_accumulate_sq_pmc.txt
c pmc_bn254_acc6.txt r06_bn254_accumulate_sq_pmc.txt
c r6e_msm_timeline_2p26.txt r06_msm_timeline_2p26.txt
c r6e_msm_timeline_all_2p16.txt r06_msm_timeline_2p16.txt
c r6e_msm_timeline_all_2p12.txt r06_msm_timeline_2p12.txt
c r6e_msm_sizes.log r06_msm_sizes.log
c r6e_msm_small_wall.log r06_msm_small_wall.log
c r6e_msm_small_grid_2p13_2p14.log r06_msm_small_grid_2p13_2p14.log
c r6e_msm_fixed_base.log r06_msm_fixed_base.log
c r6e_msm_bn254.log r06_msm_bn254.log
c r6e_msm_g2.log r06_msm_g2.log
c r6e_msm_other_curves.log r06_msm_other_curves.log
c r6e_ntt_bench.log r06_ntt_bench.log
c r6e_ntt_orders.log r06_ntt_orders.log
c r6e_ntt_vs_reference.log r06_ntt_vs_reference_2p16_2p26.log
c r6e_ntt_lde.log r06_ntt_lde.log
c r6e_pytest_gpu.log r06_pytest_gpu.log
git status --short profiles | head -40
