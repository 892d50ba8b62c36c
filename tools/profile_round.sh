#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command + two PMC passes (FETCH_SIZE, WRITE_SIZE) of two
# bench-identical steps.  (--no-verify: the untimed legs behind the timed region -- verification, the fp32 / fp32x3 points -- stay
# out of the table, whose rows are then the launches of the timed steps + warm-up only.)  Run on the GPU box from the repo root:  bash tools/profile_round.sh <tag>
tag=${1:-r03_final}
export TMPDIR=/tmp
W=/tmp/keds_prof_$tag; rm -rf $W; mkdir -p $W gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > $W/bench.log 2>&1
echo "stats rc=$?"; tail -2 $W/bench.log | cut -c1-400
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc/a -o p -- python3 tools/pmc_step.py > $W/pmc_a.log 2>&1
echo "pmc fetch rc=$?"; tail -2 $W/pmc_a.log | cut -c1-300
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc/b -o p -- python3 tools/pmc_step.py > $W/pmc_b.log 2>&1
echo "pmc write rc=$?"; tail -2 $W/pmc_b.log | cut -c1-300
python3 tools/parse_pmc.py $W/pmc gpurun_out/${tag}_pmc_traffic.json > /dev/null
f=$(find $W/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
grep '^{' $W/bench.log | tail -1 > gpurun_out/${tag}_bench.json
ls -la gpurun_out/${tag}_*
