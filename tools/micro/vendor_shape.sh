#!/bin/bash
# launch shape of the vendor BLAS kernels next to ours (rocprofv3 --kernel-trace CSV columns), reference point only
export TMPDIR=/tmp; W=/tmp/vs_$$; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --output-format csv -d $W/t -o run -- python3 tools/micro/vendor_shape.py > $W/run.log 2>&1
f=$(find $W/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seen = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not ("Cijk" in n or "gemm_bt" in n): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (n, r["Workgroup_Size_X"], r["Grid_Size_X"], r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))
    seen.setdefault(k, []).append(d)
for k, v in seen.items():
    v.sort()
    print(f"{k[0][:200]}\n    wg {k[1]} grid {k[2]} lds {k[3]} vgpr {k[4]} agpr {k[5]} sgpr {k[6]} scratch {k[7]}  median {v[len(v)//2]:.1f} us (n={len(v)})")
PY
