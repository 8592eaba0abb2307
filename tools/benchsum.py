#!/usr/bin/env python3
"""Run bench.py with several argument sets and print one summary line each (developer tool)."""
import json
import subprocess
import sys

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
for spec in sys.argv[1:]:
    extra = spec.split()
    env = dict(__import__("os").environ)
    args = []
    for e in extra:
        if "=" in e and not e.startswith("--"):
            k, v = e.split("=", 1)
            env[k] = v
        else:
            args.append(e)
    p = subprocess.run([sys.executable, ROOT + "/bench.py", "--no-cpu"] + args, capture_output=True, text=True, env=env)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        r = d["roofline"]
        ee = d.get("early_exit") or {}
        hf = d.get("fp16_messages") or {}
        print("%-40s value %8.1f ms/step %7.2f | cn %6.0f GB/s %.3f ms | vn %6.0f GB/s %.3f ms | whole %.3f | early %s" % (
            spec, d["value"], d["ms_per_step"], r["achieved"], r["avg_launch_ms"], r["vn_update"]["achieved"],
            r["vn_update"]["avg_pass_ms"], r["whole_step"]["frac"], ("%.0f Mbit/s %.1f it" % (ee["value"], ee["avg_iterations"])) if ee else "-") + ((" | fp16 fixed %.0f early %.0f Mbit/s FER %.4f cn %.0f GB/s" % (
            hf["fixed"]["value"], hf["early_exit"]["value"], hf["early_exit"]["fer"], hf["fixed"]["cn_update_GBs"])) if hf else ""), flush=True)
    except Exception as ex:
        print(spec, "FAILED", ex, p.stderr[-2000:], flush=True)
