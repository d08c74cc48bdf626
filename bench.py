#!/usr/bin/env python3
"""Headline benchmark: images/sec of SD1.5-shaped 512x512, 50-step `etainv + ptp` (forward DDIM inversion +
eta-scheduled backward edit with prompt-to-prompt attention control), batch-sharded over N GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype fp16|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = the whole hot path over one batch of B synthetic image pairs per GPU: 50 forward UNet calls (B rows:
the uncond half is skipped because guidance_scale_fwd == 1 multiplies it out) + 50 backward UNet calls (4B rows) +
the fused eta / CFG / best-of-n step, word-map and LocalBlend kernels.  The 30 backward steps whose eta(t) is 0 skip the uncond source row
(its noise prediction is dead work there) and in the 20 of them that come after prompt-to-prompt's source injection has ended the cond source row
leaves the UNet after the last layer that needs it (etainv/pipeline.py): 207.3 sample-forward equivalents = 166.6 TFLOP per image executed (the
reference's call pattern issues 300 = 241.0); `end_to_end_mfma_frac` is computed from the executed count.  Inputs (latents, contexts, noise table, edit tables, synthetic SD1.x-shaped weights) are resident
in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "eta-inversion_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402

S_STEPS = 50
L = 64
F_UNET_TFLOP = 0.8033           # per sample-forward at L = 64
FWD_PER_IMAGE = 250             # 50 x 1 (forward, cond only) + 50 x 4 (backward)
# BASELINE.json configurations that fit one GPU.  3 is the one `metric` is quoted on (the default); 2 and 5 are reported through the same
# code path and JSON fields with `--config 2|5` (their own metric string; never the headline line).
CONFIGS = {
    2: dict(name="etainv+simple", L=64, S=50, batch=1, dtype="fp16", eta=[[0.6, 0], [1, 0.7]], editor="simple", f_unet=0.8033),
    3: dict(name="etainv+ptp", L=64, S=50, batch=32, dtype="bf16", eta=[[0.6, 0], [1, 0.7]], editor="ptp", f_unet=0.8033),
    5: dict(name="etainv+masactrl", L=96, S=100, batch=8, dtype="fp16", eta=(0.0, 0.4), editor="masactrl", f_unet=2.1481),
}
MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def ptp_tables(B, S):
    """Synthetic PIE-style edit: identity alignment with token 2 replaced, equalizer 2.0 on it, blend word =
    the same token, cross_replace_steps 0.4, self_replace_steps 0.6, edit_word_idx = (1, 1)."""
    from etainv.pipeline import PtpTables
    mapper = np.tile(np.arange(77, dtype=np.int32), (B, 1))
    alphas = np.ones((B, 77), np.float32)
    mapper[:, 2], alphas[:, 2] = -1, 0.0
    eq = np.ones((B, 77), np.float32)
    eq[:, 2] = 2.0
    blend = np.zeros((B, 2, 77), np.float32)
    blend[:, :, 2] = 1.0
    ca = np.zeros((S + 1, B, 77), np.float32)
    ca[: int(0.4 * (S + 1))] = 1.0
    return PtpTables(mapper, alphas, ca, 0.6, S, equalizer=eq, blend_alpha=blend)


def make_inputs(B, rank, dev, L=L):
    g = torch.Generator().manual_seed(1000 + rank)
    z0 = (0.18215 * 5.0 * torch.randn(B, 4, L, L, generator=g)).to(dev)
    ctx_src = torch.randn(B, 2, 77, 768, generator=g).to(dev)
    ctx_tgt = torch.randn(B, 2, 77, 768, generator=g).to(dev)
    ctx_tgt[:, 0] = ctx_src[:, 0]
    tokens = torch.arange(1, 9, dtype=torch.int32).repeat(B, 1).to(dev)     # 8-word prompts
    edit_word = torch.ones(B, dtype=torch.int64)
    return z0, ctx_src, ctx_tgt, tokens, edit_word


def host_cpu_info():
    """os.cpu_count(), the lscpu model string and the physical-core count of this host."""
    info = {"cpu_count": os.cpu_count() or 1, "model": None, "physical_cores": None, "sockets": None}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for ln in out.splitlines():
            if ":" in ln:
                k, v = ln.split(":", 1)
                kv.setdefault(k.strip(), v.strip())      # (first occurrence: the per-NUMA-node lines repeat some keys)
        info["model"] = kv.get("Model name")
        sockets, cps = int(kv.get("Socket(s)", 0) or 0), int(kv.get("Core(s) per socket", 0) or 0)
        if sockets and cps:
            info["physical_cores"], info["sockets"] = sockets * cps, sockets
    except Exception:
        pass
    try:                                                  # cores this process may actually run on (cgroup / affinity limits)
        info["usable_threads"] = len(os.sched_getaffinity(0))
    except Exception:
        info["usable_threads"] = info["cpu_count"]
    return info


def cpu_baseline(S_cpu=5):
    """The CPU oracle (fp32 PyTorch restatement, kind "port") on the same workload shape, as BASELINE.md section 3 lays it out: 1 image,
    etainv + ptp, L = 64, S = 5 of the 50 steps end to end (30 UNet sample-forwards in the reference's call pattern), one warm-up UNet
    forward, linear extrapolation to 50 steps (the per-step cost is constant), images/s = 1 / that.  Threads = the host's physical cores
    (not its hardware threads: 256 SMT threads oversubscribe PyTorch-CPU's small ops, measured 66 s per sample-forward), capped by what this
    process may run on and by ETAINV_CPU_THREADS.  S = 5 as BASELINE.md says (30 sample-forwards: 60-90 s on the GPU box's host); only a host
    where one sample-forward takes more than 8 s gets S = 2 so that the default bench run stays within minutes; the JSON says which S was timed."""
    from oracle.unet import build_unet
    from oracle import loop as oloop, ptp as optp
    host = host_cpu_info()
    phys = min(host["physical_cores"] or host["cpu_count"], host["usable_threads"])
    if os.environ.get("ETAINV_CPU_THREADS"):
        phys = min(phys, int(os.environ["ETAINV_CPU_THREADS"]))
    unet = build_unet(0)
    g = torch.Generator().manual_seed(1000)
    z0 = 0.18215 * 5.0 * torch.randn(1, 4, L, L, generator=g)
    ctx_s, ctx_t = torch.randn(2, 77, 768, generator=g), torch.randn(2, 77, 768, generator=g)
    src, tgt = "a b c d e f g h", "a x c d e f g h"
    tok = optp.WordTokenizer()
    with torch.no_grad():
        # thread count: all physical cores unless fewer run the UNet FASTER -- PyTorch-CPU's fp32 convolutions of this graph are
        # memory-bandwidth bound and lose to their own synchronisation past a few dozen threads (measured on the 2 x 64-core host of the GPU box:
        # 6.0 s per sample-forward on 128 threads, about 2 s on 32) -- the baseline is the best of the two, and says which
        tried = {}
        for n_thr in sorted({phys, min(32, phys)}, reverse=True):
            torch.set_num_threads(n_thr)
            unet(z0, torch.tensor(1), encoder_hidden_states=ctx_s[:1])     # warm-up
            t0 = time.time()
            unet(z0, torch.tensor(1), encoder_hidden_states=ctx_s[:1])
            tried[n_thr] = time.time() - t0
        cores = min(tried, key=tried.get)
        torch.set_num_threads(cores)
        t_fwd = tried[cores]
        host["one_sample_forward_s_by_threads"] = {str(k): v for k, v in tried.items()}
        if t_fwd > 8.0:                                                # very slow host (30 sample-forwards would take > 4 min): keep the sample bounded
            S_cpu = 2
        noise = oloop.noise_table(S_cpu, 10, L, seed=0)
        t0 = time.time()
        o = oloop.EtaInversionOracle(unet, S=S_cpu, eta=[[0.6, 0], [1, 0.7]], L=L)
        inv = o.invert(z0, ctx_s, src)
        ctrl = optp.make_edit_controller(src, tgt, S_cpu, tok, cross_replace_steps={"default_": .4}, self_replace_steps=.6,
                                         blend_words=(("b",), ("x",)), equilizer_params={"words": ("x",), "values": (2,)})
        o.sample(inv, ctx_s, ctx_t, noise, edit_word_idx=(1, 1), controller=ctrl)
        dt = time.time() - t0
    return {"value": 1.0 / (dt * S_STEPS / S_cpu), "unit": "images/s", "cores": cores, "kind": "port",
            "host": host, "steps_timed": S_cpu, "seconds_timed": dt, "seconds_extrapolated_S50": dt * S_STEPS / S_cpu,
            "one_sample_forward_s": t_fwd,
            "sample": f"1 image, etainv+ptp 512x512, {S_cpu} of 50 DDIM steps on the CPU oracle (fp32, reference call pattern: "
                      f"{6 * S_cpu} UNet sample-forwards) in {dt:.1f} s on {cores} threads of {host['physical_cores']} physical cores ({host['model']}; the faster of "
                      f"{sorted(tried)} threads), "
                      f"extrapolated linearly to 50 steps"}


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <argv>` as a child process (never exec: a process that may
    initialise the GPU must not be replaced), relay rank 0's JSON line on stdout, everything else on stderr, and return the exit code:
    the child's, or 1 when it exited 0 without a JSON line that reports n_gpus == n."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            try:
                json.loads(t)
                line = t
                continue
            except ValueError:
                pass
        sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank child exited with code {rc}", file=sys.stderr)
        return rc
    if line is None or json.loads(line).get("n_gpus") != n:
        print(f"bench.py: the {n}-rank child produced no JSON line with n_gpus == {n}", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def stub_main(a, world, rank):
    """Tests only (`--stub`): the launcher, the rank protocol (barrier + synchronise on both sides of exactly K timed steps, MAX over ranks,
    one JSON line from rank 0) and the final all_gather, over gloo on the CPU with a placeholder workload.  Never a bench line: data = "stub"."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "RANK" in os.environ:
        dist.init_process_group("gloo")
    on = dist.is_initialized()
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(a.batch, 4, 8, 8, generator=g)

    def one_step():
        out = torch.tanh(x @ x.transpose(-1, -2))
        if on:
            gathered = [torch.empty_like(out) for _ in range(world)]
            dist.all_gather(gathered, out)
        return out

    for _ in range(a.warmup):
        one_step()
    if on:
        dist.barrier()
    t0 = time.time()
    for _ in range(a.steps):
        one_step()
    if on:
        dist.barrier()
    dt = time.time() - t0
    if on:
        tmax = torch.tensor([dt])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": a.batch * world * a.steps / max(dt, 1e-9), "unit": "images/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "stub", "config": {"workload": "launcher / rank-protocol stub (gloo, CPU)"}}), flush=True)
    if on:
        dist.destroy_process_group()
    return 0


def nets_figure(eng, dtype, B, S, L, dev, steps):
    """Secondary figure (never `value`): images/s of `BatchEditor.edit` -- what one rank of the PIE-Bench sweep (eval.py) runs per batch -- on B
    synthetic 8L x 8L images with prompts: VAE encode, 4 text-encoder calls, per-image prompt-to-prompt tables, the loop, 2 VAE decodes per
    image.  Same engine (weights resident); the third-party networks get seeded synthetic weights."""
    from types import SimpleNamespace
    from etainv.batch import BatchEditor
    from etainv.nets import NativeCLIPText, NativeVAE
    from modules.schedulers import DDIMScheduler
    from modules.utils.tokenizer import load_tokenizer
    pipe = SimpleNamespace(device=dev, engine=eng, tokenizer=load_tokenizer(), text_encoder=NativeCLIPText(None, dtype, 0), vae=NativeVAE(None, dtype, 0),
                           scheduler=DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False))
    editor = BatchEditor(pipe, num_inference_steps=S, edit_method="ptp")
    g = torch.Generator().manual_seed(77)
    nouns = ["cat", "dog", "horse", "tiger", "bird", "house", "tower", "bridge"]
    samples = []
    for b in range(B):
        s_w, t_w = nouns[b % 8], nouns[(b + 3) % 8]
        src, tgt = f"a {s_w} sitting next to a mirror", f"a {t_w} sitting next to a mirror"
        samples.append(dict(image=(torch.rand(1, 3, 8 * L, 8 * L, generator=g) * 2 - 1).to(dev), source_prompt=src, target_prompt=tgt, edit_word_idx=(1, 1),
                            ptp=dict(is_replace_controller=False, prompts=[src, tgt], cross_replace_steps={"default_": .4}, self_replace_steps=.6,
                                     blend_words=((s_w,), (t_w,)), equilizer_params={"words": (t_w,), "values": (2,)})))
    editor.edit(samples)                                   # warm-up (allocations of the nets' workspaces)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        res = editor.edit(samples)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    assert all(r is not None and torch.isfinite(r["image"]).all() for r in res)
    return {"value": B / dt, "unit": "images/s", "ms_per_step": 1e3 * dt, "steps": steps,
            "what": "image -> image BatchEditor.edit per batch: VAE encode + text encoder + prompt-to-prompt tables + etainv loop + VAE decode of both rows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configuration (3 = the headline metric)")
    ap.add_argument("--batch", type=int, default=None, help="image pairs per GPU per step (default: the configuration's)")
    ap.add_argument("--dtype", default=None, choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-nets", action="store_true", help="add a secondary figure: image -> image BatchEditor.edit (VAE encode, CLIP, loop, 2 VAE decodes)")
    ap.add_argument("--all-rows", action="store_true", help="run every UNet row the reference's call pattern has in the backward pass (250 sample-forwards per image: "
                                                           "no dead-row skipping, no early exit of the cond source rows) -- for A/B; the default is the engine as shipped")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # tests only: launcher / rank protocol over gloo, no engine, no GPU
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    if a.all_rows:                                 # the reference's backward row count AND every row through every layer (no shared prefix)
        os.environ["ETAINV_NO_DEAD_ROW_SKIP"] = "1"
        os.environ["ETAINV_NO_PREFIX_SHARE"] = "1"
    if a.batch is None:
        a.batch = int(os.environ.get("ETAINV_BENCH_BATCH", cfg["batch"])) if a.config == 3 else cfg["batch"]
    if a.dtype is None:
        a.dtype = os.environ.get("ETAINV_BENCH_DTYPE", cfg["dtype"]) if a.config == 3 else cfg["dtype"]
    L, S_STEPS, F_UNET_TFLOP = cfg["L"], cfg["S"], cfg["f_unet"]
    FWD_PER_IMAGE = 5 * S_STEPS

    # `python bench.py --gpus N` with N > 1 and no torch.distributed.run environment: this process becomes the launcher (it has not touched
    # the GPU: importing torch does not) and the N ranks run as its CHILD process tree; a rank count that differs from --gpus is an error,
    # never a silent one-GPU run.
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus} "
              f"(or plain `python bench.py --gpus {a.gpus}`, which starts the ranks itself)", file=sys.stderr)
        sys.exit(2)
    if a.stub:
        return stub_main(a, world, rank)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        print(f"bench.py: rank {rank} needs cuda:{local}; {torch.cuda.device_count()} device(s) visible -- no CPU fallback", file=sys.stderr)
        sys.exit(3)
    dist = None
    if world > 1 or "RANK" in os.environ:          # under torch.distributed.run (also with one rank: exercises the RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        # RCCL prints a version banner on STDOUT when the communicator is created; keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    from etainv import _capi
    from etainv.engine import Engine
    from etainv.pipeline import EtaLoop, noise_table
    lib = _capi.load()
    B = a.batch
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    eng = Engine(dtype=dtype, max_unet_batch=4 * B, latent_size=L, max_img=B, device=str(dev))
    eng.load_default(0)
    loop = EtaLoop(eng, S=S_STEPS, eta=cfg["eta"], noise_sample_count=10)
    z0, ctx_src, ctx_tgt, tokens, edit_word = make_inputs(B, rank, dev, L)
    noise = noise_table(S_STEPS, 10, L, seed=0, device=dev)
    tables = ptp_tables(B, S_STEPS) if cfg["editor"] == "ptp" else None
    masa = (4, 10) if cfg["editor"] == "masactrl" else None          # MasactrlEditor defaults (masactrl_editor.py:22)

    def one_step():
        inv = loop.invert(z0, ctx_src, tokens)
        out = loop.sample(inv, ctx_src, ctx_tgt, noise, edit_word=edit_word, ptp=tables, masactrl=masa)
        if dist is not None:       # the path's only exchange: final gather of the edited latents (32 KiB / image)
            gathered = [torch.empty_like(out) for _ in range(world)]
            dist.all_gather(gathered, out)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    barrier()
    rows0 = loop.rows_executed
    t0 = time.time()
    for _ in range(a.steps):
        out = one_step()
    barrier()
    dt = time.time() - t0
    # UNet sample-forwards the timed steps actually issued per image: S cond rows forward + 4 per backward step, minus the uncond source row of the
    # backward steps whose eta(t) is 0 (dead work: EtaLoop.skip_dead_source_rows) and the tail of the cond source row once nothing is injected from it
    # (EtaLoop.src_exit) -- 207.3 instead of 250 with the paper's eta schedule and the PIE prompt-to-prompt settings
    FWD_PER_IMAGE = (loop.rows_executed - rows0) / (B * a.steps)
    assert torch.isfinite(out).all(), "non-finite edited latents"
    if dist is not None:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    with_nets = None
    if a.with_nets and cfg["editor"] == "ptp":
        with_nets = nets_figure(eng, dtype, B, S_STEPS, L, dev, max(1, min(a.steps, 3)))
    # per-kernel-class HIP-event timing (roofline fields) on ONE extra step OUTSIDE the timed region: the events sit on the launch
    # stream around every launch and would otherwise add their own (small) cost to `value`
    lib.etainv_prof_reset()
    lib.etainv_prof_enable(1)
    tp0 = time.time()
    one_step()
    torch.cuda.synchronize()
    dt_prof = time.time() - tp0
    lib.etainv_prof_enable(0)

    def prof(cls):
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        _capi.check(lib.etainv_prof_read(cls, C.byref(ms), C.byref(work), C.byref(n)))
        return ms.value, work.value, n.value

    ig_ms, ig_flop, ig_n = prof(0)
    sa_ms, sa_flop, sa_n = prof(1)
    ca_ms, ca_flop, ca_n = prof(2)
    gn_ms, gn_bytes, gn_n = prof(3)
    ln_ms, ln_bytes, ln_n = prof(4)
    split, split_n = (C.c_double * 6)(), (C.c_int64 * 2)()
    ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)          # 312.5 FLOP per byte
    _capi.check(lib.etainv_prof_split(0, ridge, split, split_n))
    # fabric-side bytes per igemm launch: a PMC figure (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/unet_call.py
    # --rows 128, FETCH_SIZE x2 per MI355X_MICROARCH.md, aggregated by tools/pmc_traffic.py), NOT measured by this run: the file is named
    # in the JSON and belongs to the kernels of the round it was taken in
    # (taken at 128 UNet rows = the default workload; null for the other configurations, whose launches have other sizes)
    # A file is used only when its `kernel_src_sha16` (tools/srcstamp.py: hash of csrc/ + include/etainv.h, written by the aggregators) equals that of the
    # sources this run was built from; otherwise the field is null and `*_source` says which file was stale.
    sys.path.insert(0, str(ROOT / "tools"))
    from srcstamp import kernel_src_sha16, stamped_figure
    src_sha = kernel_src_sha16()
    traffic = traffic_src = mfma_busy = mfma_busy_src = None
    if 4 * B == 128:
        traffic, traffic_src = stamped_figure(ROOT / "profiles", "r*_pmc_traffic_rows128.json", src_sha, lambda d: d.get("igemm", {}).get("hbm_bytes_per_launch"),
                                              "fabric bytes per implicit-GEMM launch of one 128-row UNet call")
        # matrix-pipe busy share of the same kernels from the SQ counters (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over tools/unet_call.py
        # --rows 128, aggregated by tools/pmc_sq.py): file-sourced like `traffic`, named in the JSON
        mfma_busy, mfma_busy_src = stamped_figure(ROOT / "profiles", "r*_pmc_sq_rows128.json", src_sha, lambda d: (d.get("summary", {}).get("igemm_family") or {}).get("mfma_busy_frac"),
                                                  "cycle-weighted MFMA-busy share of the implicit-GEMM kernels over one 128-row UNet call")
    images = B * world * a.steps
    value = images / dt
    # MFMA FLOPs the kernels EXECUTED per image (implicit GEMMs + both attentions of the profiled step: the launchers record 2 M N K / 4 B h N^2 d
    # of every launch), i.e. after dead rows, early exits and the shared context-independent prefix -- not a row count times a constant
    exec_tflop_per_image = (ig_flop + sa_flop + ca_flop) / B / 1e12
    if rank == 0:
        achieved = ig_flop / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        line = {
            "metric": f"images/sec SD1.5 {8 * L}^2 {S_STEPS}-step {cfg['name']}", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{cfg['name']}, SD1.5-shaped UNet (seeded synthetic weights), {8 * L}x{8 * L} ({L}x{L} latents), {S_STEPS} DDIM steps, "
                                   f"{B} image pairs per GPU per step, eta {cfg['eta']}, n=10 noise candidates, cfg 7.5/1",
                       "baseline_config": a.config,
                       "images_per_gpu": B, "unet_sample_forwards_per_image": FWD_PER_IMAGE,
                       "unet_sample_forwards_per_image_reference": 6 * S_STEPS,
                       "row_economy": "executed count = S cond rows forward (guidance_scale_fwd == 1 multiplies the uncond half out) + per backward step 4 rows while eta(t) > 0, "
                                      "3 rows [u_t, c_s, c_t] while eta(t) == 0 (the source row is replayed and its guided noise only picks a noise sample that is multiplied by "
                                      "eta = 0), the cond source row leaving the UNet after transformer block 12 / 9 once no cross replacement / no self-replace reads it; same edited "
                                      "latents (tests/test_e2e_gpu.py, profiles/r03_parity_S50.json); `--all-rows` runs 5 S rows per image",
                       "tflop_per_image": exec_tflop_per_image,
                       "tflop_per_image_note": "executed MFMA FLOPs (sum over the launches of one step, after row skipping, early exits and prefix sharing); "
                                               f"row count x {F_UNET_TFLOP} TFLOP would give {FWD_PER_IMAGE * F_UNET_TFLOP:.1f}, the reference's call pattern "
                                               f"{6 * S_STEPS * F_UNET_TFLOP:.1f}",
                       "sharding": f"batch-shard x{world}, final all_gather of latents"},
            "end_to_end_mfma_frac": value / world * exec_tflop_per_image / MFMA_PEAK_TFLOPS,
            "roofline": {"bound": "mfma", "kernel": "implicit-GEMM family: pp_conv2_kernel / pp_conv_kernel (conv3x3, PATCH ping-pong, dual-M / 256-pixel tiles), pp_dualn_kernel (1x1 / Linear / GEGLU, dual-N "
                                                    "ping-pong), igemm_kernel (strided / upsampling / small launches)", "achieved": achieved,
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_busy_src,
                         "note": "achieved = GEMM FLOPs only over the kernel's whole duration; its epilogues also carry bias / time row / residual / GEGLU and, "
                                 "since round 2, the LayerNorm and GroupNorm statistics that were separate passes (ETAINV_LN_UNFUSED / ETAINV_GN_UNFUSED move them "
                                 "back out: higher igemm TFLOP/s, lower images/s)",
                         "launches": ig_n, "avg_launch_ms": ig_ms / max(ig_n, 1), "share_of_wall": ig_ms * 1e-3 / dt_prof,
                         "measured_on": "one extra step after the timed region, HIP events on the launch stream"},
            # the same igemm launches split by which roofline bounds them (algorithmic intensity of the launch vs the 312.5 FLOP/B ridge): the
            # K <= 640 projections of the transformer blocks move more bytes than the MFMA peak could consume
            "igemm_by_bound": {
                "mfma_bound": {"launches": split_n[0], "tflops": split[1] / max(split[0], 1e-9) / 1e9, "frac_of_mfma_peak": split[1] / max(split[0], 1e-9) / 1e9 / MFMA_PEAK_TFLOPS,
                               "share_of_igemm_time": split[0] / max(split[0] + split[3], 1e-9)},
                "hbm_bound": {"launches": split_n[1], "algorithmic_gbs": split[5] / max(split[3], 1e-9) / 1e6, "frac_of_hbm_peak": split[5] / max(split[3], 1e-9) / 1e6 / HBM_PEAK_GBS,
                              "tflops": split[4] / max(split[3], 1e-9) / 1e9, "share_of_igemm_time": split[3] / max(split[0] + split[3], 1e-9)}},
            "other_kernels": {
                "self_attention": {"tflops": sa_flop / max(sa_ms, 1e-9) / 1e9, "share_of_wall": sa_ms * 1e-3 / dt_prof, "launches": sa_n},
                "cross_attention": {"tflops": ca_flop / max(ca_ms, 1e-9) / 1e9, "share_of_wall": ca_ms * 1e-3 / dt_prof, "launches": ca_n},
                "groupnorm": {"gbs": gn_bytes / max(gn_ms, 1e-9) / 1e6, "frac_hbm": gn_bytes / max(gn_ms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                              "share_of_wall": gn_ms * 1e-3 / dt_prof, "launches": gn_n},
                "layernorm": {"gbs": ln_bytes / max(ln_ms, 1e-9) / 1e6, "frac_hbm": ln_bytes / max(ln_ms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                              "share_of_wall": ln_ms * 1e-3 / dt_prof, "launches": ln_n}},
        }
        if with_nets is not None:
            line["with_nets"] = with_nets
        if world == 1 and not a.no_cpu_baseline and a.config == 3:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
