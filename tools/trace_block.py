#!/usr/bin/env python3
"""One steady-state tower block out of a rocprofv3 --kernel-trace CSV of bench.py: every launch (both queues) between two
consecutive attention launches in the middle of the trace, with start / end relative to the first and the gap to the previous
launch on the same queue; then the mean duration of every (kernel, position in the block) over all blocks.
Usage: python tools/trace_block.py <kernel_trace.csv> [attention-kernel-substring]"""
import csv
import re
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
att = sys.argv[2] if len(sys.argv) > 2 else "attention_s257"
key_q = "Queue_Id" if "Queue_Id" in rows[0] else "Queue_ID"


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([\w:]+(?:<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:44]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r[key_q]) for r in rows)
idx = [i for i, e in enumerate(ev) if att in e[2]]
mid = len(idx) // 2
a, b = idx[mid], idx[mid + 1]
t0 = ev[a][0]
last_end = {}
print(f"block between attention launches {mid} and {mid + 1} ({(ev[b][0] - t0) / 1e3:.1f} us):")
for s, e, n, q in ev[a:b + 1]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print(f"  q{q:>3s}  {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n}")
    last_end[q] = e
# per (kernel, occurrence within a block) means
occ = defaultdict(list)
for i0, i1 in zip(idx[:-1], idx[1:]):
    seen = defaultdict(int)
    for s, e, n, q in ev[i0:i1]:
        occ[(n, seen[n])].append((e - s) / 1e3)
        seen[n] += 1
print("mean duration by (kernel, occurrence in the block), blocks with the full pattern:")
for (n, k), v in sorted(occ.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= len(idx) // 2:
        print(f"  {n:44s} #{k}  n {len(v):4d}  mean {statistics.mean(v):7.1f}  median {statistics.median(v):7.1f} us")
