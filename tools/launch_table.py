#!/usr/bin/env python3
"""Join the launch trace of one UNet call (ETAINV_TRACE_IGEMM=1: one stderr line per igemm launch with its shape and the kernel family it was routed to)
with the event times of the same launches (tools/unet_call.py --shapes --dump) and print time per (shape, route), largest first.

    ETAINV_TRACE_IGEMM=1 python tools/unet_call.py --rows 128 --calls 2 --shapes --dump launches.json 2> trace.txt
    python tools/launch_table.py launches.json trace.txt
"""
import collections
import json
import sys

recs = json.load(open(sys.argv[1]))["igemm"]
lines = [l.strip() for l in open(sys.argv[2]) if l.startswith("igemm M=")]
lines = lines[-len(recs):]                       # the last call's launches
assert len(lines) == len(recs), (len(lines), len(recs))
tab = collections.OrderedDict()
for r, l in zip(recs, lines):
    e = tab.setdefault(l, [0, 0.0, 0.0])
    e[0] += 1
    e[1] += r["ms"]
    e[2] += r["flops"]
tot = sum(e[1] for e in tab.values())
by_route = collections.defaultdict(lambda: [0, 0.0, 0.0])
print(f"{len(recs)} launches, {tot:.2f} ms")
for l, (n, ms, fl) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.3f} ms  x{n:3d}  {fl / ms / 1e9:7.1f} TF/s  {l[6:]}")
    rt = l.rsplit("route=", 1)[1]
    by_route[rt][0] += n
    by_route[rt][1] += ms
    by_route[rt][2] += fl
print()
for rt, (n, ms, fl) in sorted(by_route.items(), key=lambda kv: -kv[1][1]):
    print(f"route {rt:10s} {n:4d} launches {ms:8.3f} ms ({ms / tot:5.1%})  {fl / ms / 1e9:7.1f} TF/s")
