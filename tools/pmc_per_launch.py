#!/usr/bin/env python3
"""Per-LAUNCH HBM traffic of the implicit-GEMM kernel against its algorithmic bytes: where the fabric over-fetch sits.

Inputs: the launch list of one UNet call (tools/unet_call.py --shapes --dump launches.json: event time, FLOPs and algorithmic bytes of every igemm launch
in launch order) and the counter_collection.csv files of two rocprofv3 --pmc passes over the SAME command (FETCH_SIZE and WRITE_SIZE, separate passes,
KiB units, FETCH_SIZE x 2 for 16-byte-per-lane streaming reads on gfx950: MI355X_MICROARCH.md, HBM section).  The igemm dispatches of the LAST call in
the counter files are matched to the launch list by order.  Output: one row per distinct (FLOPs, bytes) shape: launches, ms, measured / algorithmic bytes.

    python tools/pmc_per_launch.py launches.json fetch/counter_collection.csv write/counter_collection.csv > profiles/rNN_pmc_per_shape.json
"""
import collections
import csv
import json
import sys


def igemm_dispatches(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and any(t in r["Kernel_Name"] for t in ("igemm_kernel", "pp_conv_kernel", "pp_conv2_kernel", "pp_dualn_kernel", "pp_gemm_kernel"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    launches = json.load(open(sys.argv[1]))["igemm"]
    n = len(launches)
    fetch, write = igemm_dispatches(sys.argv[2], "FETCH_SIZE"), igemm_dispatches(sys.argv[3], "WRITE_SIZE")
    assert len(fetch) >= n and len(write) >= n and len(fetch) % n == 0, (len(fetch), len(write), n)
    fetch, write = fetch[-n:], write[-n:]
    agg = collections.OrderedDict()
    for l, f, w in zip(launches, fetch, write):
        key = (l["flops"], l["bytes"])
        a = agg.setdefault(key, {"launches": 0, "ms": 0.0, "fetch": 0.0, "write": 0.0})
        a["launches"] += 1
        a["ms"] += l["ms"]
        a["fetch"] += f * 1024 * 2
        a["write"] += w * 1024
    out, tot_m, tot_a = [], 0.0, 0.0
    for (flops, nbytes), a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        meas = a["fetch"] + a["write"]
        tot_m += meas
        tot_a += nbytes * a["launches"]
        out.append({"flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes, "launches": a["launches"], "ms_total": a["ms"],
                    "tflops": flops * a["launches"] / a["ms"] / 1e9, "fetch_bytes_per_launch": a["fetch"] / a["launches"],
                    "write_bytes_per_launch": a["write"] / a["launches"], "measured_over_algorithmic": meas / max(nbytes * a["launches"], 1.0),
                    "intensity_flop_per_byte": flops / max(nbytes, 1.0)})
    print(json.dumps({"igemm_launches": n, "measured_bytes": tot_m, "algorithmic_bytes": tot_a, "measured_over_algorithmic": tot_m / tot_a, "shapes": out}, indent=1))


if __name__ == "__main__":
    main()
