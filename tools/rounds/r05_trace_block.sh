#!/bin/bash
# Round 5: one block of the default step out of a kernel trace, both queues (when do the side lane's launches actually run?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; W=/tmp/ktb; rm -rf $W; mkdir -p $W gpurun_out/r05
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $W/t -o b -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify > $W/b.log 2>&1
f=$(find $W/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/r05/r05_trace_block.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full block of the trace: from the 3rd-last attention launch to the 2nd-last
att = [i for i, r in enumerate(rows) if "attention_s257" in r["Kernel_Name"]]
i0, i1 = att[-14], att[-12]
t0 = int(rows[i0]["Start_Timestamp"])
qs = sorted({r["Queue_Id"] for r in rows[i0:i1 + 1]})
print("queues:", qs, "(two blocks; times in us from the first attention launch's start)")
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f'{"main" if r["Queue_Id"] == rows[i0]["Queue_Id"] else "   side":8s} {n:46s} start {s:9.1f}  end {e:9.1f}  dur {e - s:7.1f}  grid {r["Grid_Size_X"]:>7s}')
PY
