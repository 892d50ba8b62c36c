#!/usr/bin/env python3
"""Where the main lane of a tower pass waits: from a rocprofv3 --kernel-trace CSV of bench.py, the gap in front of every
launch of the big kernels on the stream that carries them (end of the previous kernel on that stream -> start of this one),
by kernel.  A join that waits for the side lane shows up as a long gap in front of the attention kernel.
Usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import re
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
key_q = "Queue_Id" if "Queue_Id" in rows[0] else "Queue_ID" if "Queue_ID" in rows[0] else None
by_q = defaultdict(list)
for r in rows:
    by_q[r.get(key_q, "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
main_q = max(by_q, key=lambda q: sum(e - s for s, e, _ in by_q[q]))          # the queue with the most kernel time
ev = sorted(by_q[main_q])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
    return n[-48:]


gaps, durs = defaultdict(list), defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ev[:-1], ev[1:]):
    gaps[short(n1)].append((s1 - e0) / 1e3)
    durs[short(n1)].append((e1 - s1) / 1e3)
print(f"queue {main_q}: {len(ev)} launches; other queues: " + ", ".join(f"{q}: {len(v)}" for q, v in by_q.items() if q != main_q))
print(f"{'kernel':50s} {'launches':>8s} {'gap median':>11s} {'gap mean':>9s} {'gap p90':>8s} {'dur mean':>9s}   (us)")
for n in sorted(gaps, key=lambda k: -sum(durs[k])):
    g = sorted(gaps[n])
    if len(g) < 8:
        continue
    print(f"{n:50s} {len(g):8d} {statistics.median(g):11.2f} {statistics.mean(g):9.2f} {g[int(0.9 * len(g))]:8.2f} {statistics.mean(durs[n]):9.1f}")
tot_gap = sum(sum(v) for v in gaps.values())
tot = ev[-1][1] - ev[0][0]
print(f"sum of gaps on the main queue: {tot_gap / 1e3:.2f} ms of {tot / 1e6:.2f} ms traced")
