#!/usr/bin/env python3
"""Same-box A/B of two builds of the library on the per-shape micro-benchmark (boxes of the pool differ by several percent, and a
profiled clock differs from an unprofiled one: never compare numbers taken on different boxes).
    python tools/ab_ops.py --a eta-inversion_amd/etainv/lib/libetainv_hip.so --b .../libetainv_hip_<variant>.so [--only lin] [--rounds 3]
Runs tools/bench_ops.py under each library in alternation and prints the best time per shape for both."""
import argparse
import collections
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
ap = argparse.ArgumentParser()
ap.add_argument("--a", required=True)
ap.add_argument("--b", required=True)
ap.add_argument("--only", default="")
ap.add_argument("--rows", default="128")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--env-a", default="", help="extra VAR=value (comma separated) for arm A")
ap.add_argument("--env-b", default="")
a = ap.parse_args()
best = {"a": collections.OrderedDict(), "b": collections.OrderedDict()}
for r in range(a.rounds):
    for arm, lib, extra in (("a", a.a, a.env_a), ("b", a.b, a.env_b)):
        env = {**os.environ, "ETAINV_LIB": str(Path(lib).resolve())}
        for kv in filter(None, extra.split(",")):
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_ops.py"), "--rows", a.rows] + (["--only", a.only] if a.only else []),
                             env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            m = re.match(r"(.+?)\s+([0-9.]+) ms\s+([0-9.]+) (TFLOP/s|GB/s)", line)
            if m:
                name, ms = m.group(1).strip(), float(m.group(2))
                best[arm][name] = min(ms, best[arm].get(name, 1e9))
print(f"{'shape':34s} {'A ms':>9s} {'B ms':>9s}   B/A")
for name in best["a"]:
    if name in best["b"]:
        print(f"{name:34s} {best['a'][name]:9.3f} {best['b'][name]:9.3f}   {best['b'][name] / best['a'][name]:.3f}")
