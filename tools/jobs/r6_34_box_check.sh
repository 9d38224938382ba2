# Is the box whole?  CU count, clocks, the headline size.  Outputs: gpurun_out/r6_34_box_check.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
(python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, 'CUs', p.total_memory>>30, 'GiB')"
 rocm-smi --showclocks 2>/dev/null | head -20
 timeout 300 python tools/gpu_msm_tail.py ab 16 20 26 2>&1 | grep -v amdgpu | grep "auto") > $R/gpurun_out/r6_34_box_check.log 2>&1
cat $R/gpurun_out/r6_34_box_check.log
