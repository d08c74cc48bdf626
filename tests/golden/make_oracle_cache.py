#!/usr/bin/env python3
"""Run the CPU oracle's legs of the `-m gpu` tests once and store their outputs under tests/golden/oracle_cache/ (see tests/oracle_cache.py).

    python tests/golden/make_oracle_cache.py                 # every leg that has no file yet
    python tests/golden/make_oracle_cache.py --force         # recompute everything
    python tests/golden/make_oracle_cache.py --only realsize # legs whose name contains the pattern
    python tests/golden/make_oracle_cache.py --list           # have / miss / STALE (written from another oracle/*.py or leg source: MANIFEST.json)

CPU only (the legs regenerate their inputs from seeds and never touch the engine); about an hour on 8 cores, dominated by the L = 64 / 96
loops and their emulated-16-bit floors.  The S = 50 free-running runs of `tests/parity_s50.py` (50 min per pair on the GPU box's host) are
converted, not recomputed: `--from-parity-cache DIR` reads the `.pt` files that script wrote and stores the checkpoints the S = 50 test uses."""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "eta-inversion_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402

LEG_MODULES = ["tests.test_unet_gpu", "tests.test_e2e_gpu", "tests.test_fp32_gpu", "tests.test_realsize_gpu", "tests.test_configs_gpu"]


def import_legs():
    import importlib
    for m in LEG_MODULES:
        importlib.import_module(m)
    from tests import oracle_cache
    return oracle_cache


def convert_parity_cache(src: Path, oc):
    """profiles/_cache/parity_S50_L64_pair{i}_{fp32,fp16,bf16}.pt (tests/parity_s50.py oracle workers) -> tests/golden/oracle_cache/s50_*.npz:
    the fp32 oracle's checkpoints (kept as float32), and for the emulated 16-bit executions only their recorded distances to it (the floors)."""
    from tests.parity_s50 import compare
    keep = (1, 5, 10, 25, 50)
    for f in sorted(src.glob("parity_S50_L64_pair*_fp32.pt")):
        d = torch.load(f)
        i = d["pair"]
        idx = [d["steps"].index(s) for s in keep]
        out = {"S": d["S"], "L": d["L"], "pair": i, "steps": list(keep), "inv": d["inv"][idx], "bwd": d["bwd"][idx], "map": d["map"],
               "best": d["best"], "losses": d["losses"], "out": d["out"], "oracle_seconds": d["seconds"], "oracle_threads": d["threads"], "floors": {}}
        for kind in ("fp16", "bf16"):
            g = src / f"parity_S50_L64_pair{i}_{kind}.pt"
            if g.exists():
                c = compare(torch.load(g), d)
                out["floors"][kind] = {"final_edit_rel_l2": c["final_edit_rel_l2"], "best_of_n_agree": c["best_of_n_agree"],
                                       "edit_rel_l2": {str(r["step"]): r["edit_rel_l2"] for r in c["per_step"] if r["step"] in keep},
                                       "inv_rel_l2": {str(r["step"]): r["inv_rel_l2"] for r in c["per_step"] if r["step"] in keep}}
        path = oc.CACHE_DIR / f"s50_pair{i}.npz"
        oc.save(path, out)
        print(f"wrote {path.name} ({path.stat().st_size / 1e6:.2f} MB), floors: {list(out['floors'])}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--from-parity-cache", default=None)
    ap.add_argument("--stamp-manifest", action="store_true", help="record today's oracle / leg fingerprints for every committed result WITHOUT recomputing it: "
                    "only right after `ETAINV_SLOW=1 pytest tests/test_oracle_cache.py` has shown that the files equal the live oracle")
    a = ap.parse_args()
    os.environ["ETAINV_ORACLE"] = "write"
    torch.set_num_threads(a.threads)
    oc = import_legs()
    if a.from_parity_cache:
        return convert_parity_cache(Path(a.from_parity_cache), oc)
    if a.stamp_manifest:
        n = 0
        for name, (fn, cases) in oc.LEGS.items():
            for args in cases:
                key = oc.key_of(name, args)
                if (oc.CACHE_DIR / f"{key}.npz").exists():
                    oc.record(key, fn)
                    n += 1
        print(f"stamped {n} entries: oracle {oc.oracle_fingerprint()}")
        return
    todo = []
    for name, (fn, cases) in oc.LEGS.items():
        for args in cases:
            key = oc.key_of(name, args)
            if a.only and a.only not in key:
                continue
            exists = (oc.CACHE_DIR / f"{key}.npz").exists()
            if a.list:
                why = oc.stale_reason(key, fn) if exists else None
                print(("STALE " if why else "have  " if exists else "miss  ") + key + (f"   <- {why}" if why else "") +
                      ("" if not exists or key in oc.read_manifest() else "   (no manifest entry)"))
            elif a.force or not exists:
                todo.append((name, args, key))
    if a.list:
        return
    # legs that reuse another leg's cached result (the floors read the fp32 run) find it because dependencies are declared first in each module
    mod = {m.rsplit(".", 1)[-1]: sys.modules[m] for m in LEG_MODULES}
    t_all = time.time()
    for _, _, key in todo:
        (oc.CACHE_DIR / f"{key}.npz").unlink(missing_ok=True)
    for name, args, key in todo:
        t0 = time.time()
        m, f = name.split(".")
        getattr(mod[m], f)(*args)                       # the decorated wrapper: computes and writes in "write" mode
        path = oc.CACHE_DIR / f"{key}.npz"
        print(f"{key}: {time.time() - t0:.0f} s, {path.stat().st_size / 1e3:.0f} kB", flush=True)
    print(f"{len(todo)} legs in {time.time() - t_all:.0f} s; cache holds {sum(p.stat().st_size for p in oc.CACHE_DIR.glob('*.npz')) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
