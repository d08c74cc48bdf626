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
    ap.add_argument("--io_threads", type=int, default=8, help="worker threads for image decode / resize and PNG encoding (overlap the engine)")
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
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from modules.models import StablePreprocess
    # Host work off the engine's critical path: file decode + resize of the NEXT batch and PNG encoding of the PREVIOUS one run on worker
    # threads (PIL / zlib / numpy release the GIL) while this thread enqueues the current batch's kernels; the device copies stay here.
    pre_cpu = StablePreprocess("cpu", size=a.size)
    if a.io_threads > 0:
        pool = ThreadPoolExecutor(max_workers=a.io_threads)
    else:                                    # --io_threads 0: the host work runs inline on this thread (the unoverlapped baseline)
        from concurrent.futures import Future

        class _Inline:
            def submit(self, fn, *args):
                f = Future()
                f.set_result(fn(*args))
                return f

            def shutdown(self):
                pass
        pool = _Inline()
    chunks = [todo[b0:b0 + a.batch] for b0 in range(0, len(todo), a.batch)]
    load = lambda chunk: [pool.submit(pre_cpu, s["image_file"]) for _, s, _ in chunk]
    t0, done, latents, saves = time.time(), 0, {}, []
    t_wait_load = t_edit = t_post = 0.0
    nxt = load(chunks[0]) if chunks else None
    for ci, chunk in enumerate(chunks):
        ta = time.time()
        images = [f.result().to(pipe.device) for f in nxt]
        nxt = load(chunks[ci + 1]) if ci + 1 < len(chunks) else None
        tb = time.time()
        samples = [dict(image=img, source_prompt=s["source_prompt"], target_prompt=s["edit"]["target_prompt"],
                        edit_word_idx=s["edit_word_idx"], ptp=s["edit"].get("ptp")) for img, (_, s, _) in zip(images, chunk)]
        results = editor.edit(samples)
        ok = [(i, f, res) for (i, _, f), res in zip(chunk, results) if res is not None]   # failed edits: skipped like the reference (eval.py:103-105)
        if ok:
            # one device -> host transfer for the whole batch (this is also where the batch's kernels are waited for)
            imgs = (torch.cat([res["image"] for _, _, res in ok]) / 2 + 0.5).clamp(0, 1).mul(255).permute(0, 2, 3, 1).to(torch.uint8).cpu().numpy()
            lats = torch.cat([res["latent"] for _, _, res in ok]).cpu()
            tc = time.time()
            for k, (i, f, _) in enumerate(ok):
                saves.append(pool.submit(lambda arr, path: Image.fromarray(arr).save(path), imgs[k], str(f)))
                latents[i] = lats[k]
                done += 1
        else:
            tc = time.time()
        # bound the encodes in flight (a slow disk must not let uint8 images pile up in host memory for the whole sweep) and surface a failed save
        # now, not after all the GPU work: everything but the last two batches' worth is waited for
        while len(saves) > 2 * a.batch:
            saves.pop(0).result()
        t_wait_load += tb - ta
        t_edit += tc - tb
        t_post += time.time() - tc
    for f in saves:
        f.result()
    pool.shutdown()
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
    print(f"[rank {rank}] edited {done} of {len(todo)} images in {dt:.1f}s ({done / max(dt, 1e-9):.3f} images/s); main thread: waited {t_wait_load:.2f}s for decoded "
          f"inputs, {t_edit:.2f}s in BatchEditor.edit + device->host, {t_post:.2f}s handing results to the PNG workers")


if __name__ == "__main__":
    main()
