#!/usr/bin/env python3
"""PIE-Bench sweep on the batched native engine (reference eval.py:65-106 + utils/eval_utils.py:209-260).

    python eval.py --data_path data/eval/PIE-Bench_v1 --output result/pie_etainv_ptp [--batch 32] [--steps 50] [--prec fp16]
                   [--limit N] [--categories 1_change_object ...] [--edit_method ptp] [--override]
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 eval.py ...      (one rank per GPU)

Differences from the reference driver, by design: images are edited B at a time (the engine batches independent pairs), the data
set is sharded over ranks by image (rank r takes i = r mod world; the reference shards by config), and the only exchange is the
final gather of the edited latents.  Kept: prompts / ptp config / edit_word_idx per sample from PieBenchData, the output name
`imgs/{i:04d}_{source}_{target}.png`, skip-existing resume, silently skipping samples whose edit returns None."""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from dataset.pie_bench_data import PieBenchData, edit_image_name  # noqa: E402
from etainv.batch import BatchEditor  # noqa: E402
from etainv.shard import shard_indices  # noqa: E402
from modules import load_diffusion_model  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="PIE-Bench sweep: etainv + {ptp, simple, masactrl} on the MI355X engine")
    ap.add_argument("--data_path", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--model", default="CompVis/stable-diffusion-v1-4")
    ap.add_argument("--edit_method", default="ptp", choices=["ptp", "simple", "masactrl"])
    ap.add_argument("--batch", type=int, default=32, help="image pairs per engine call and GPU")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--prec", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--categories", nargs="+", default=None)
    ap.add_argument("--override", action="store_true", help="re-edit images whose output file exists")
    ap.add_argument("--save_latents", action="store_true", help="gather the edited latents of all ranks to rank 0 -> <output>/latents.pt")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    # "nccl" = RCCL over xGMI on the GPU box; ETAINV_DIST_BACKEND=gloo runs the same sharding / gather / resume plumbing on CPU (tests)
    backend = os.environ.get("ETAINV_DIST_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend)
    data = PieBenchData(a.data_path, skip_img_load=True, limit=a.limit, categories=a.categories)
    out_dir = Path(a.output) / "imgs"
    out_dir.mkdir(parents=True, exist_ok=True)
    todo = []
    for i in shard_indices(len(data), rank, world):
        s = data[i]
        f = out_dir / f"{edit_image_name(i, s['source_prompt'], s['edit']['target_prompt'])}.png"
        if a.override or not f.exists():                           # skip-existing resume (eval_utils.py:256-258)
            todo.append((i, s, f))
    pipe, (preproc, postproc) = load_diffusion_model(a.model, f"cuda:{local}" if world > 1 and backend == "nccl" else "cuda", variant=a.prec,
                                                     latent_size=a.size // 8, max_img=a.batch)
    editor = BatchEditor(pipe, num_inference_steps=a.steps, edit_method=a.edit_method)
    from PIL import Image
    t0, done, latents = time.time(), 0, {}
    for b0 in range(0, len(todo), a.batch):
        chunk = todo[b0:b0 + a.batch]
        samples = [dict(image=preproc(s["image_file"]), source_prompt=s["source_prompt"], target_prompt=s["edit"]["target_prompt"],
                        edit_word_idx=s["edit_word_idx"], ptp=s["edit"].get("ptp")) for _, s, _ in chunk]
        for (i, _, f), res in zip(chunk, editor.edit(samples)):
            if res is None:
                continue                                            # failed edit: skipped like the reference (eval.py:103-105)
            Image.fromarray(postproc(res["image"])).save(str(f))
            latents[i] = res["latent"][0].cpu()
            done += 1
    if world > 1:
        import torch.distributed as dist
        if a.save_latents:                                         # the one exchange step: RCCL all_gather of 32 KiB per image
            from etainv.shard import gather_latents
            idx, L = shard_indices(len(data), rank, world), a.size // 8
            mine = torch.zeros(len(idx), 4, L, L, device=pipe.device)
            for k, i in enumerate(idx):
                if i in latents:
                    mine[k] = latents[i].to(pipe.device)
            full = gather_latents(mine, len(data), rank, world)
            if rank == 0:
                torch.save(full.cpu(), str(Path(a.output) / "latents.pt"))
        dist.barrier()
        dist.destroy_process_group()
    elif a.save_latents:
        torch.save(latents, str(Path(a.output) / "latents.pt"))
    dt = time.time() - t0
    print(f"[rank {rank}] edited {done} of {len(todo)} images in {dt:.1f}s ({done / max(dt, 1e-9):.3f} images/s)")


if __name__ == "__main__":
    main()
