#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files: per-kernel-class HBM traffic (FETCH_SIZE doubled for wide
coalesced reads on gfx950, WRITE_SIZE exact; both in KiB units -> bytes), per MI355X_MICROARCH.md section HBM."""
import collections
import csv
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from srcstamp import kernel_src_sha16  # noqa: E402

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        cls = "igemm" if any(t in k for t in ("igemm_kernel", "pp_conv_kernel", "pp_conv2_kernel", "pp_dualn_kernel", "pp_gemm_kernel")) else "self_attn" if "self_attn" in k else "cross_attn" if "cross_attn" in k else \
              "groupnorm" if "gn_" in k else "layernorm" if ("layernorm" in k or "ln_finalize" in k or "row_stats" in k) else "other"
        agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[cls][r["Counter_Name"]] += 1
out = {}
for cls, v in agg.items():
    n = max(calls[cls].values())
    fetch = v.get("FETCH_SIZE", 0.0) * 1024 * 2      # KiB units; x2 gfx950 correction for 16-B/lane streaming reads
    write = v.get("WRITE_SIZE", 0.0) * 1024
    out[cls] = {"launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
                "hbm_bytes_per_launch": (fetch + write) / n}
out["kernel_src_sha16"] = kernel_src_sha16()
print(json.dumps(out, indent=1))
