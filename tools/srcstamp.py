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


def stamped_figure(profiles_dir, pattern, src_sha, pick, what):
    """(value, source note) from the newest profiles/<pattern> file IF it was taken on the kernel sources `src_sha`; otherwise (None, "stale: ...").
    `pick(json) -> value or None`.  bench.py's `roofline.traffic` / `roofline.mfma_busy_frac` come through here."""
    import json
    for cand in sorted(Path(profiles_dir).glob(pattern), reverse=True):
        d = json.load(open(cand))
        if d.get("kernel_src_sha16") != src_sha:
            return None, f"stale: profiles/{cand.name} was taken on kernel sources {d.get('kernel_src_sha16', 'unstamped')}, this run is {src_sha}"
        v = pick(d)
        if v is None:
            return None, f"profiles/{cand.name} holds no {what}"
        return v, f"profiles/{cand.name} ({what}, same kernel sources {src_sha})"
    return None, None


if __name__ == "__main__":
    print(kernel_src_sha16())
