#!/usr/bin/env python3
"""Compare a freshly derived S = 50 fp32 oracle run (tests/parity_s50.py --oracle-worker ... --worker-out X.pt) with the committed trace
tests/golden/oracle_cache/s50_pair<i>.npz: best-of-n choices, losses, checkpoints of both trajectories, the edit word's map, the final latents.

    python tests/parity_s50.py --oracle-worker --pair 0 --kind fp32 --threads 7 --worker-out /tmp/s50_pair0.pt      # ~2 h on 8 cores
    python tools/compare_s50_rederived.py /tmp/s50_pair0.pt
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests.oracle_cache import CACHE_DIR, load  # noqa: E402

new = torch.load(sys.argv[1])
ref = load(CACHE_DIR / f"s50_pair{new['pair']}.npz")
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
idx = [new["steps"].index(s) for s in ref["steps"]]
fin = torch.isfinite(ref["losses"])                       # (steps with eta = 0 carry inf: the same places in both, compared separately)
rows = {"best-of-n choices equal": bool(torch.equal(new["best"], ref["best"])),
        "non-finite loss entries at the same places": bool(torch.equal(torch.isfinite(new["losses"]), fin)),
        "losses (finite entries) rel L2": rel(new["losses"][fin], ref["losses"][fin]),
        "inversion checkpoints rel L2": rel(new["inv"][idx], ref["inv"]),
        "backward checkpoints rel L2": rel(new["bwd"][idx], ref["bwd"]),
        "edit-word map rel L2": rel(new["map"], ref["map"]),
        "final latents rel L2": rel(new["out"], ref["out"])}
print(f"pair {new['pair']}, S = {new['S']}, L = {new['L']}: derived in {new['seconds']:.0f} s on {new['threads']} threads (committed: {ref['oracle_seconds']:.0f} s on {ref['oracle_threads']})")
for k, v in rows.items():
    print(f"  {k}: {v if isinstance(v, bool) else format(v, '.2e')}")
ok = all(v for v in rows.values() if isinstance(v, bool)) and all(v <= 2e-5 for v in rows.values() if not isinstance(v, bool))
print("EQUAL (within 2e-5: thread-count-dependent summation order of the CPU GEMMs)" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
