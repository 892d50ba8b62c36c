#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "quad" > $O/pytest_quad.log 2>&1; echo "pytest quad rc=$?"; tail -5 $O/pytest_quad.log
timeout 900 python tools/ab_quad.py > $O/ab_quad2.txt 2>&1; cat $O/ab_quad2.txt
