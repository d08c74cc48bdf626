#!/usr/bin/env python3
"""Full-length parity of the native loop against the CPU oracle ("Next round" item 1a): etainv + prompt-to-prompt,
L = 64, S = 50 (the benchmark's configuration: eta [[0.6, 0], [1, 0.7]], n = 10 candidates, cfg 7.5 / 1), free-running.

  subjects   hip_fp16 / hip_bf16 / hip_fp32  the native engine (C ABI) in each operand precision it offers
             ref_fp16 / ref_bf16             the oracle with the REFERENCE's 16-bit path emulated (oracle/lowprec.py): what diffusers'
                                              own fp16 / bf16 execution loses against fp32 on the same weights -- the fair yardstick
  reference  the fp32 oracle (oracle/loop.py EtaInversionOracle + oracle/ptp.py controller), one run per pair

Reports, at steps 1, 5, 10, 25, 50 (and a few in between): rel-L2 / max-abs of the inversion trajectory and of the edited (target)
latent, the per-step best-of-n agreement count (with the oracle's relative loss gap at every disagreement), and the share of the final
edited latent's elements inside north_star's rtol 1e-3 / atol 1e-4.

The oracle costs ~2 s per UNet sample-forward on the GPU box's host (300 per run): every (pair, kind) run is a worker PROCESS of this
script; they run concurrently on disjoint thread budgets while the GPU runs finish in seconds.  Worker results are cached (--cache,
default profiles/_cache/, git-ignored but shipped to the GPU box) so that a later call -- e.g. after a kernel change -- only redoes the
GPU side.

    python tests/parity_s50.py --out gpurun_out/r03_parity_S50.json            # on the GPU box (through gpurun)
    python tests/parity_s50.py --S 4 --L 16 --pairs 1 --subjects ref_fp16      # CPU-only smoke of the harness
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "eta-inversion_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ETA = [[0.6, 0], [1, 0.7]]
PTP_CFG = dict(is_replace_controller=False, cross_replace_steps={"default_": .4}, self_replace_steps=.6)
KINDS = {"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}


def keep_steps(S):
    return sorted({s for s in (1, 2, 3, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50) if s <= S} | {S})


def inputs(n_pairs, L):
    g = torch.Generator().manual_seed(123)
    pairs = json.load(open(ROOT / "tests" / "golden" / "prompt_pairs.json"))
    chosen = [pairs[0], pairs[3], pairs[1], pairs[2]][:n_pairs]
    z0 = 0.8 * torch.randn(n_pairs, 4, L, L, generator=g)
    ctx_src = torch.randn(n_pairs, 2, 77, 768, generator=g)
    ctx_tgt = torch.randn(n_pairs, 2, 77, 768, generator=g)
    ctx_tgt[:, 0] = ctx_src[:, 0]
    return chosen, z0, ctx_src, ctx_tgt


# ------------------------------------------------------------------------------------------------ oracle worker
def oracle_worker(a):
    from oracle import loop as oloop, ptp as optp
    from oracle.unet import build_unet
    from oracle.lowprec import LowPrecisionUNet
    torch.set_num_threads(a.threads)
    S, L, i = a.S, a.L, a.pair
    pairs, z0, ctx_src, ctx_tgt = inputs(a.pairs, L)
    src, tgt = pairs[i]
    unet = build_unet(0)
    if KINDS[a.kind] is not None:
        unet = LowPrecisionUNet(unet, KINDS[a.kind])
    noise = oloop.noise_table(S, 10, L, seed=0)
    tok = optp.WordTokenizer()
    t0 = time.time()
    with torch.no_grad():
        o = oloop.EtaInversionOracle(unet, S=S, eta=ETA, L=L, use_mask=True)
        inv = o.invert(z0[i:i + 1], ctx_src[i], src)
        bw, tw = src.split(" ")[1], tgt.split(" ")[1]
        controller = optp.make_edit_controller(src, tgt, S, tok, blend_words=((bw,), (tw,)), equilizer_params={"words": (tw,), "values": (2,)},
                                               res=L // 4, thres_n=(L // 2) ** 2, **PTP_CFG)
        trace = []
        out = o.sample(inv, ctx_src[i], ctx_tgt[i], noise, edit_word_idx=(1, 1), controller=controller, trace=trace)
    ks = keep_steps(S)
    inv_lat = torch.cat(inv["latents"])                                              # (S+1, 4, L, L)
    res = {"S": S, "L": L, "pair": i, "kind": a.kind, "seconds": time.time() - t0, "threads": a.threads, "steps": ks,
           "inv": torch.stack([inv_lat[s] for s in ks]), "inv_all_norm": inv_lat.flatten(1).norm(dim=1),
           "map": torch.stack(inv["attn_maps_mean"])[1, 0],                           # the edit word's mean map
           "bwd": torch.stack([trace[s - 1]["latent"] for s in ks]),                  # (len, 2, 4, L, L) latent AFTER backward step s
           "best": torch.tensor([t["best"] for t in trace]), "losses": torch.stack([t["losses"] for t in trace]), "out": out}
    torch.save(res, a.worker_out)
    print(f"oracle worker pair {i} {a.kind}: {res['seconds']:.0f} s on {a.threads} threads", flush=True)


# ------------------------------------------------------------------------------------------------ native runs
def native_run(dtype, S, L, n_pairs, batch=None, select=None, engine=None):
    """batch (>= n_pairs): the n_pairs oracle-traced pairs are images 0 .. n_pairs-1 of a `batch`-image call; the other images cycle through the same
    prompt pairs with latents / contexts of their own (tests/test_bench_shape_gpu.py: the benchmark's B = 32).  Returns the runs of ALL images.
    select: run only these images of that input set (a smaller call on the same inputs); engine: reuse an Engine instead of building one."""
    from oracle import ptp as optp                       # host-side table builders only (checker infrastructure, like the tests)
    from etainv.engine import Engine
    from etainv.pipeline import EtaLoop, PtpTables, noise_table
    pairs, z0, ctx_src, ctx_tgt = inputs(n_pairs, L)
    B = n_pairs
    if batch is not None and batch > n_pairs:
        g = torch.Generator().manual_seed(321)
        extra = batch - n_pairs
        pairs = pairs + [pairs[i % n_pairs] for i in range(extra)]
        z0 = torch.cat([z0, 0.8 * torch.randn(extra, 4, L, L, generator=g)])
        xs, xt = torch.randn(extra, 2, 77, 768, generator=g), torch.randn(extra, 2, 77, 768, generator=g)
        xt[:, 0] = xs[:, 0]
        ctx_src, ctx_tgt = torch.cat([ctx_src, xs]), torch.cat([ctx_tgt, xt])
        B = batch
    if select is not None:
        sel = list(select)
        pairs, z0, ctx_src, ctx_tgt, B = [pairs[i] for i in sel], z0[sel], ctx_src[sel], ctx_tgt[sel], len(sel)
    tok = optp.WordTokenizer()
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    mp, al, eq, ba, ca = [], [], [], [], []
    for b, (src, tgt) in enumerate(pairs):
        ws = src.split(" ")
        tokens[b, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
        bw, tw = ws[1], tgt.split(" ")[1]
        m, al_ = optp.refinement_mapper(src, tgt, tok)
        mp.append(m); al.append(al_)
        eq.append(optp.equalizer(tgt, (tw,), (2,), tok))
        ba.append(optp.blend_alpha_layers([src, tgt], ((bw,), (tw,)), tok))
        ca.append(optp.time_words_alpha([src, tgt], S, {"default_": .4}, tok)[:, 0])
    eng = engine
    if eng is None:
        eng = Engine(dtype=dtype, max_unet_batch=4 * B, latent_size=L, max_img=B)
        eng.load_synthetic(0)
    ptp = PtpTables(np.stack(mp), np.stack(al), np.stack(ca, 1), 0.6, S, equalizer=np.stack(eq), blend_alpha=np.stack(ba))
    loop = EtaLoop(eng, S=S, eta=ETA, use_mask=True)
    nz = noise_table(S, 10, L, seed=0)
    t0 = time.time()
    inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
    trace = []
    out = loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), nz, edit_word=torch.tensor([1] * B), ptp=ptp, trace=trace)
    torch.cuda.synchronize()
    secs = time.time() - t0
    ks = keep_steps(S)
    lat = inv["latents"].cpu()                                                        # (S+1, B, 4, L, L)
    runs = []
    for b in range(B):
        runs.append({"steps": ks, "inv": torch.stack([lat[s, b] for s in ks]), "map": inv["maps_mean"][b, 1].cpu(),
                     "bwd": torch.stack([torch.stack([trace[s - 1]["latent"][b].cpu(), trace[s - 1]["latent"][B + b].cpu()]) for s in ks]),
                     "best": torch.tensor([int(t["best"][b]) for t in trace]), "out": torch.stack([out[b].cpu(), out[B + b].cpu()]),
                     # the native per-candidate losses of every best-of-n step (zeros at the eta = 0 steps, which choose nothing)
                     "losses": torch.stack([t["losses"][b].cpu() if t.get("losses") is not None else torch.zeros(10) for t in trace]),
                     "seconds": secs})
    if engine is None:
        eng.close()
    return runs


# ------------------------------------------------------------------------------------------------ comparison
def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def mabs(a, b):
    return float((a.double() - b.double()).abs().max())


def compare(sub, ref):
    """one subject run vs the fp32 oracle run of the same pair"""
    ks = ref["steps"]
    rows = []
    for k, s in enumerate(ks):
        rows.append({"step": s,
                     "inv_rel_l2": rel(sub["inv"][k], ref["inv"][k]), "inv_max_abs": mabs(sub["inv"][k], ref["inv"][k]),
                     "edit_rel_l2": rel(sub["bwd"][k][1], ref["bwd"][k][1]), "edit_max_abs": mabs(sub["bwd"][k][1], ref["bwd"][k][1]),
                     "src_rel_l2": rel(sub["bwd"][k][0], ref["bwd"][k][0])})
    S = len(ref["best"])
    agree = (sub["best"] == ref["best"])
    flips = []
    for i in (~agree).nonzero().flatten().tolist():
        ls = ref["losses"][i]
        flips.append({"bwd_step": i + 1, "native": int(sub["best"][i]), "oracle": int(ref["best"][i]),
                      "oracle_rel_loss_gap": float(abs(ls[int(sub["best"][i])] - ls[int(ref["best"][i])]) / ls[int(ref["best"][i])])})
    a, b = sub["out"][1].double(), ref["out"][1].double()
    within = float(((a - b).abs() <= 1e-4 + 1e-3 * b.abs()).double().mean())
    return {"per_step": rows, "best_of_n_agree": int(agree.sum()), "best_of_n_steps": S, "best_of_n_flips": flips,
            "final_edit_rel_l2": rel(a, b), "final_edit_max_abs": mabs(a, b), "final_edit_abs_max_of_ref": float(b.abs().max()),
            "final_edit_frac_within_rtol1e-3_atol1e-4": within, "final_edit_allclose_rtol1e-3_atol1e-4": bool(within == 1.0),
            "final_src_rel_l2": rel(sub["out"][0], ref["out"][0]), "edit_word_map_rel_l2": rel(sub["map"], ref["map"])}


def physical_cores():
    try:
        kv = dict(ln.split(":", 1) for ln in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines() if ":" in ln)
        n = int(kv["Socket(s)"].strip()) * int(kv["Core(s) per socket"].strip())
    except Exception:
        n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=50)
    ap.add_argument("--L", type=int, default=64)
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--subjects", nargs="*", default=["hip_fp16", "hip_bf16", "hip_fp32", "ref_fp16", "ref_bf16"])
    ap.add_argument("--cache", default=str(ROOT / "profiles" / "_cache"))
    ap.add_argument("--cache-out", default=None, help="where NEW worker results are written (default: --cache); on the GPU box: gpurun_out/parity_cache")
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-workers", type=int, default=2, help="concurrent oracle processes (the host's memory bandwidth, not its core count, bounds the CPU UNet: "
                                                               "6 at once ran 4.7x slower each on the GPU box)")
    ap.add_argument("--ref-pairs", type=int, default=None, help="run the ref_* (emulated 16-bit reference) subjects on the first N pairs only")
    # worker mode
    ap.add_argument("--oracle-worker", action="store_true")
    ap.add_argument("--pair", type=int, default=0)
    ap.add_argument("--kind", default="fp32")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--worker-out", default=None)
    a = ap.parse_args()
    if a.oracle_worker:
        return oracle_worker(a)

    cache, cache_out = Path(a.cache), Path(a.cache_out or a.cache)
    cache_out.mkdir(parents=True, exist_ok=True)
    kinds = ["fp32"] + [s[4:] for s in a.subjects if s.startswith("ref_")]
    name = lambda i, k: f"parity_S{a.S}_L{a.L}_pair{i}_{k}.pt"
    ref_pairs = a.pairs if a.ref_pairs is None else a.ref_pairs
    todo = [(i, k) for i in range(a.pairs) for k in kinds if (k == "fp32" or i < ref_pairs)
            and not (cache / name(i, k)).exists() and not (cache_out / name(i, k)).exists()]
    procs = []
    if todo:
        nthreads = max(1, physical_cores() // min(len(todo), a.max_workers))
        print(f"{len(todo)} oracle runs to do ({todo}), {nthreads} threads each", flush=True)
    pending = list(todo)

    def start_more():
        while pending and len([p for p in procs if p[0].poll() is None]) < a.max_workers:
            i, k = pending.pop(0)
            cmd = [sys.executable, __file__, "--oracle-worker", "--S", str(a.S), "--L", str(a.L), "--pairs", str(a.pairs), "--pair", str(i), "--kind", k,
                   "--threads", str(nthreads), "--worker-out", str(cache_out / name(i, k))]
            env = dict(os.environ, OMP_NUM_THREADS=str(nthreads), MKL_NUM_THREADS=str(nthreads))
            procs.append((subprocess.Popen(cmd, env=env), i, k))
    start_more()

    # GPU runs while the oracle workers occupy the host cores
    native = {}
    hip = [s for s in a.subjects if s.startswith("hip_")]
    if hip:
        assert torch.cuda.is_available(), "hip_* subjects need a GPU (no CPU fallback); pass --subjects ref_fp16 ... for a CPU-only harness check"
        for s in hip:
            dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[s[4:]]
            try:
                native[s] = native_run(dt, a.S, a.L, a.pairs)
                print(f"{s}: native run of {a.pairs} pairs took {native[s][0]['seconds']:.1f} s", flush=True)
            except Exception as ex:                                 # (e.g. an engine built without the fp32-operand path)
                print(f"{s}: NOT RUN -- {type(ex).__name__}: {ex}", flush=True)
                native[s] = None
    while pending or any(p[0].poll() is None for p in procs):
        start_more()
        time.sleep(2)
    bad = [(i, k, p.returncode) for p, i, k in procs if p.returncode != 0]
    assert not bad, f"oracle workers failed: {bad}"

    load = lambda i, k: torch.load(cache_out / name(i, k) if (cache_out / name(i, k)).exists() else cache / name(i, k))
    report = {"config": {"S": a.S, "L": a.L, "pairs": a.pairs, "eta": ETA, "editor": "ptp (Refine + Reweight + LocalBlend)", "n_candidates": 10,
                         "reference": "fp32 CPU oracle (oracle/loop.py), free-running", "tolerance_north_star": "rtol 1e-3 / atol 1e-4"},
              "oracle_seconds": {}, "subjects": {}}
    for i in range(a.pairs):
        ref = load(i, "fp32")
        report["oracle_seconds"][f"pair{i}"] = ref["seconds"]
        for s in a.subjects:
            if s.startswith("hip_"):
                if native.get(s) is None:
                    report["subjects"].setdefault(s, {})[f"pair{i}"] = None
                    continue
                sub = native[s][i]
            else:
                if i >= ref_pairs:
                    continue
                sub = load(i, s[4:])
            report["subjects"].setdefault(s, {})[f"pair{i}"] = compare(sub, ref)
    # summary table
    print(f"\n== etainv+ptp L={a.L} S={a.S}, free-running, vs the fp32 oracle (worst over {a.pairs} pairs) ==")
    for s, per in report["subjects"].items():
        runs = [r for r in per.values() if r]
        if not runs:
            print(f"{s:9s} not run")
            continue
        line = f"{s:9s} best-of-n agree {min(r['best_of_n_agree'] for r in runs)}/{runs[0]['best_of_n_steps']}  final edit rel-L2 {max(r['final_edit_rel_l2'] for r in runs):.2e} " \
               f"max-abs {max(r['final_edit_max_abs'] for r in runs):.2e}  within tol {min(r['final_edit_frac_within_rtol1e-3_atol1e-4'] for r in runs):.4f}  inv(S) rel-L2 " \
               f"{max(r['per_step'][-1]['inv_rel_l2'] for r in runs):.2e}"
        print(line)
        for st in (1, 5, 10, 25, 50):
            rows = [row for r in runs for row in r["per_step"] if row["step"] == st]
            if rows:
                print(f"          step {st:2d}: inversion rel-L2 {max(x['inv_rel_l2'] for x in rows):.2e} max-abs {max(x['inv_max_abs'] for x in rows):.2e} | "
                      f"edited rel-L2 {max(x['edit_rel_l2'] for x in rows):.2e} max-abs {max(x['edit_max_abs'] for x in rows):.2e}")
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        json.dump(report, open(a.out, "w"), indent=1)
        print("wrote", a.out)


if __name__ == "__main__":
    main()
