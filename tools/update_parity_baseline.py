#!/usr/bin/env python3
"""After a full `pytest -m gpu` run on the MI355X (gpurun merges gpurun_out/gpu_test_metrics.jsonl back): write the measured
parity numbers of the build to the TRACKED profiles/r<NN>_parity.json (ROUND, default r04) together with the digest of the
kernel sources they were measured on (every record carries it: tests/gpu_util.report), and -- only with --rebase, after a
deliberate numerics change -- refresh tests/golden/parity_baseline.json, the table the parity assertions take their 2x limits
from.  bench.py reads the newest profiles/r*_parity.json for its `recall_parity` statement and drops it when the digest no
longer matches the sources."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = os.environ.get("ROUND", "r04")
recs = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "gpu_test_metrics.jsonl")) if l.strip()]
last = collections.OrderedDict()
for r in recs:
    last[r["test"]] = r
digests = {v.get("csrc_sha16") for v in last.values()}
if len(digests) != 1 or None in digests:
    raise SystemExit(f"gpu_test_metrics.jsonl mixes runs of different sources ({digests}): delete it and re-run pytest -m gpu")
rows = {k: {"rel_l2": v["rel_l2"], "min_cosine": v["min_cosine"]} for k, v in last.items() if "rel_l2" in v and "min_cosine" in v}
other = {k: {a: b for a, b in v.items() if a not in ("test", "csrc_sha16")} for k, v in last.items() if k not in rows}
out = {"what": "parity of the HIP path against the reference-minted golden vectors (tests/golden/*.npz) and the CPU oracle, as measured "
               "by `pytest tests -m gpu` on one MI355X; rel_l2 = ||got - want|| / ||want||, min_cosine over rows",
       "csrc_sha16": digests.pop(), "rows": rows, "other_metrics": other}
path = os.path.join(ROOT, "profiles", f"{tag}_parity.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(f"{path}: {len(rows)} parity rows, {len(other)} other metric rows, sources {out['csrc_sha16']}")
if "--rebase" in sys.argv:
    path = os.path.join(ROOT, "tests", "golden", "parity_baseline.json")
    old = json.load(open(path)) if os.path.exists(path) else {"rows": {}}
    old["rows"].update(rows)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    print(f"tests/golden/parity_baseline.json: {len(old['rows'])} rows")
