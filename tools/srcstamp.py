#!/usr/bin/env python3
"""Fingerprint of the kernel sources (csrc/*.hip, *.h, engine.cpp + include/etainv.h): the PMC summaries under profiles/ carry it, and bench.py
reports a file-sourced counter figure only when the file was taken on the kernels it is running (otherwise null + a `stale` note)."""
import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def kernel_src_sha16() -> str:
    h = hashlib.sha256()
    files = sorted((ROOT / "eta-inversion_amd" / "csrc").glob("*.hip")) + sorted((ROOT / "eta-inversion_amd" / "csrc").glob("*.h")) + \
        [ROOT / "eta-inversion_amd" / "csrc" / "engine.cpp", ROOT / "include" / "etainv.h"]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_src_sha16())
