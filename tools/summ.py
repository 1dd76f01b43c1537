#!/usr/bin/env python3
"""Summaries of bench.py JSON lines written by tools/gpu_steps.sh: tools/summ.py DIR name..."""
import json
import sys

d = sys.argv[1]
for n in sys.argv[2:]:
    try:
        j = json.loads(open(f"{d}/{n}.out").read().strip().splitlines()[-1])
        r = j.get("roofline", {})
        print(f"{n:14s} {j['value']:9.0f} evals/s  {j['ms_per_step']:.4f} ms/step  valid={j.get('valid')}  resum in-region {r.get('ms_per_launch', 0):.4f} alone {r.get('ms_per_launch_alone', 0):.4f}")
    except Exception as exc:  # noqa: BLE001
        print(f"{n:14s} ERR {exc!r}")
        try:
            print(open(f"{d}/{n}.err").read()[-700:])
        except OSError:
            pass
