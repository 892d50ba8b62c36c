#!/usr/bin/env python3
"""After a full `pytest -m gpu` run on the MI355X (gpurun merges gpurun_out/gpu_test_metrics.jsonl back): write the measured
parity numbers of the build to the TRACKED profiles/r03_parity.json, and -- only with --rebase, after a deliberate numerics
change -- refresh tests/golden/parity_baseline.json, the table the parity assertions take their 2x limits from."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
recs = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "gpu_test_metrics.jsonl")) if l.strip()]
last = collections.OrderedDict()
for r in recs:
    last[r["test"]] = r
rows = {k: {"rel_l2": v["rel_l2"], "min_cosine": v["min_cosine"]} for k, v in last.items() if "rel_l2" in v and "min_cosine" in v}
other = {k: {a: b for a, b in v.items() if a != "test"} for k, v in last.items() if k not in rows}
out = {"what": "parity of the HIP path against the reference-minted golden vectors (tests/golden/*.npz) and the CPU oracle, as measured "
               "by `pytest tests -m gpu` on one MI355X; rel_l2 = ||got - want|| / ||want||, min_cosine over rows",
       "rows": rows, "other_metrics": other}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_parity.json"), "w"), indent=1, sort_keys=True)
print(f"profiles/r03_parity.json: {len(rows)} parity rows, {len(other)} other metric rows")
if "--rebase" in sys.argv:
    path = os.path.join(ROOT, "tests", "golden", "parity_baseline.json")
    old = json.load(open(path)) if os.path.exists(path) else {"rows": {}}
    old["rows"].update(rows)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    print(f"tests/golden/parity_baseline.json: {len(old['rows'])} rows")
