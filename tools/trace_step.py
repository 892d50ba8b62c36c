#!/usr/bin/env python3
"""One step of a bench command out of a rocprofv3 --kernel-trace CSV: the launches between two consecutive occurrences of a marker
kernel (default: the first kernel of a step, preprocess / patch embedding), in time order on all queues, consecutive launches of the
same kernel folded into one line (count, total, mean), with the wall time of each phase.
Usage: python tools/trace_step.py <kernel_trace.csv> [marker-substring] [skip-substring ...]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "im2col"
key_q = "Queue_Id" if "Queue_Id" in rows[0] else "Queue_ID"


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([\w:]+(?:<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:52]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r[key_q]) for r in rows)
idx = [i for i, e in enumerate(ev) if marker in e[2]]
if len(idx) < 3:
    names = sorted({e[2] for e in ev})
    sys.exit("marker %r found %d times; kernels: %s" % (marker, len(idx), ", ".join(names)[:3000]))
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = ev[a][0]
print(f"step between launches {a} and {b}: {(ev[b][0] - t0) / 1e3:.1f} us, {b - a} launches")
i = a
while i < b:
    j = i
    tot = 0
    while j < b and ev[j][2] == ev[i][2] and ev[j][3] == ev[i][3]:
        tot += ev[j][1] - ev[j][0]
        j += 1
    n = j - i
    print(f"  q{ev[i][3]:>3s} t={(ev[i][0] - t0) / 1e3:9.1f}  {ev[i][2]:52s} x{n:3d}  total {tot / 1e3:8.1f}  mean {tot / n / 1e3:7.1f} us")
    i = j
