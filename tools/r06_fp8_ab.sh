#!/bin/bash
# round 6: tile-switch arrangement of the 4-wave MXFP8 kernel, same box: the round-5 kernel, then this round's with each piece off
# (needs the older kernel beside it, which is not kept in the tree: before the gpurun call, in the build container:
#  mkdir -p tools/ab_old && git show <commit of the older kernel>:keds_amd/csrc/gemm_fp8.hip > tools/ab_old/gemm_fp8_r05.hip -- round 6 used 235db3d / the commit before the scale-LDS experiment)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
forms() { ROUNDS=3 timeout 300 python tools/fp8_forms.py 2>&1 | grep -v amdgpu.ids | sed 's/8 waves:.*4 waves/4 waves/'; }
line() { timeout 300 python bench.py --precision fp8 --steps 40 --warmup 4 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python3 tools/ab_line.py; }
{
cp keds_amd/csrc/gemm_fp8.hip /tmp/new.hip
cp tools/ab_old/gemm_fp8_r05.hip keds_amd/csrc/gemm_fp8.hip
echo "### round-5 kernel"; build "" && { forms; line; }
cp /tmp/new.hip keds_amd/csrc/gemm_fp8.hip
echo "### round-6 kernel"; build "" && { forms; line; }
echo "### parity of the round-6 kernel"; timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | tail -3
for V in "-DKEDS_FQ_FOLD=0" "-DKEDS_FQ_VMCNT=0" "-DKEDS_FQ_FASTSTART=0" "-DKEDS_FQ_FOLD4=1"; do
  echo "### $V"; build "$V" && { forms; line; }
done
build ""
echo "### round-6 kernel again"; forms; line
} 2>&1 | tee $O/fp8_tile_switch_ab.txt
