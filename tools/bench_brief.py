#!/usr/bin/env python3
"""Print the few numbers of a bench.py JSON line that the A/B scripts look at (stdin or a file)."""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"]), round(d["ms_per_step"], 4), d.get("host_us_per_step"))
print([(e["kernel"][:14], round(e["ms_per_launch"], 4), e.get("cosmologies_per_launch")) for e in [r] + r.get("roofline_others", [])])
