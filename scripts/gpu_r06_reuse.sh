timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bf16.py tests/test_gpu_minkunet.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | tail -6 | tee gpurun_out/r06_reuse_tests.log
for r in 1 2; do for tag in "" noreuse; do
  ME_AMD_LIB_TAG=$tag timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 30 --warmup 8 --cpu-budget 0 --pmc off --extra-workloads off > gpurun_out/r06_unet_reuse_${tag:-on}_$r.json 2> gpurun_out/r06_unet_reuse_${tag:-on}_$r.err
  python -c "import json; l=json.loads(open('gpurun_out/r06_unet_reuse_${tag:-on}_$r.json').read().strip().splitlines()[-1]); print('unet ${tag:-reuse} $r', l['ms_per_step'], l['value'])"
done; done
