#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc, counters in their own passes) of the config-2 kernels at the end of round 3:
# fp32 (k_conv_tile_f32x3_ws, k_wgrad_f32x3) and bf16 (k_conv_tile_bf16<128,64,..,deep>, k_wgrad_bf16).
set +e
for dt in f32 bf16; do
  DTYPE=$dt bash scripts/gpu_pmc_final.sh r03_pmc_$dt > gpurun_out/r03_pmc_$dt.log 2>&1
  grep -v amdgpu.ids gpurun_out/r03_pmc_$dt.log | tail -12
done
