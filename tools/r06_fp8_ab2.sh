#!/bin/bash
# round 6: MX scale bytes through LDS (whole-dword stores) against the previous kernel, alone and in the step, same box
# (needs the older kernel beside it, which is not kept in the tree: before the gpurun call, in the build container:
#  mkdir -p tools/ab_old && git show <commit of the older kernel>:keds_amd/csrc/gemm_fp8.hip > tools/ab_old/gemm_fp8_prev.hip -- round 6 used b4a0f71^ / the commit before the scale-LDS experiment)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
forms() { ROUNDS=3 timeout 300 python tools/fp8_forms.py 2>&1 | grep -v amdgpu.ids | sed 's/8 waves:.*4 waves/4 waves/'; }
line() { for i in 1 2; do timeout 300 python bench.py --precision fp8 --steps 40 --warmup 4 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python3 tools/ab_line.py; done; }
{
cp keds_amd/csrc/gemm_fp8.hip /tmp/new.hip
echo "### scale bytes through LDS"; forms; line
echo "### parity"; timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | tail -3
cp tools/ab_old/gemm_fp8_prev.hip keds_amd/csrc/gemm_fp8.hip
echo "### previous kernel (16-bit scale stores from the row groups)"; build "" && { forms; line; }
cp /tmp/new.hip keds_amd/csrc/gemm_fp8.hip
build ""
echo "### scale bytes through LDS, again"; forms; line
} 2>&1 | tee $O/fp8_scale_lds_ab.txt
