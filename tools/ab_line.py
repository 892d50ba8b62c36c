#!/usr/bin/env python3
"""One line per bench run for the same-box A/B scripts: reads the bench JSON line on stdin."""
import json
import sys
d = json.loads(sys.stdin.read())
st = d["stage_ms_per_step"]
ws = d["roofline_scan"].get("whole_search") or {}
print("   %.0f img/s  %.3f ms/step  gemm %.3f ms  attn %.3f ms  frac %.4f  search %.1f us  guard tripped: %s" % (
    d["value"], d["ms_per_step"], st["gemm"] or 0.0, st["attention"] or 0.0, d["roofline"]["frac"], ws.get("ms", 0.0) * 1e3, d["numerics_guard"]["tripped"]))
