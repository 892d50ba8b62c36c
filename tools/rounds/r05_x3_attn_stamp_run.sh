python -m pytest tests/test_gpu_fp32.py -x -q -k "attention" 2>&1 | tail -2; python tools/ab_attn_x3.py 2>&1 | grep -v amdgpu.ids
(cd keds_amd/csrc && make -j8 EXTRA="-DKEDS_AX_DBG=16" > /tmp/mk.log 2>&1) && python tools/ax_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_x3_attn_stamps_v3.txt
(cd keds_amd/csrc && make -j8 EXTRA="" > /tmp/mk.log 2>&1)
