#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline"
run() { $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['stage_ms_per_step']['gemm'],3), d['side_lane_rows'])"; }
for rnd in 1 2; do
for cfg in "side=1 quad=auto" "side=0 quad=auto" "side=1 quad=0" "side=0 quad=0" "side=1 quad=1"; do
  unset KEDS_SIDE_STREAM KEDS_GEMM_QUAD
  case "$cfg" in *side=0*) export KEDS_SIDE_STREAM=0;; esac
  case "$cfg" in *quad=0*) export KEDS_GEMM_QUAD=0;; *quad=1*) export KEDS_GEMM_QUAD=1;; esac
  echo "$cfg: $(run)"
done; done
