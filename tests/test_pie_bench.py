"""PIE-Bench loader mirror (eta-inversion_amd/dataset/pie_bench_data.py) vs records produced by the reference's own
dataset/pie_bench_data.py on a synthetic mapping file (tests/golden/pie_bench.json, masks bit-packed in pie_bench_masks.npz)."""
import json
import os
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden"


def test_records_and_masks(tmp_path):
    from dataset.pie_bench_data import PieBenchData, edit_image_name
    g = json.loads((GOLD / "pie_bench.json").read_text())
    masks = np.load(GOLD / "pie_bench_masks.npz")
    (tmp_path / "mapping_file.json").write_text(json.dumps(g["mapping"]))
    data = PieBenchData(str(tmp_path), skip_img_load=True)
    assert len(data) == len(g["records"])
    for i, want in enumerate(g["records"]):
        s = data[i]
        assert s["source_prompt"] == want["source_prompt"] and s["target_prompt"] == want["target_prompt"]
        assert os.path.relpath(s["image_file"], tmp_path) == want["image_rel"]
        assert s["edit_word_idx"] == want["edit_word_idx"]
        assert json.loads(json.dumps(s["edit"]["ptp"])) == want["ptp"]
        assert s["image"] is None
        m = s["mask"].numpy()
        assert m.shape == (512, 512) and m.dtype == np.float32
        assert np.array_equal(np.packbits(m.astype(np.uint8)), masks[f"mask{i}"])          # bit-exact
        assert float(m.sum()) == want["mask_sum"]
    # bracket stripping, missing blend word -> None, clipping of a run that passes the end of the image
    assert "[" not in data[0]["source_prompt"] and data[3]["edit_word_idx"][1] is None and data[2]["edit"]["ptp"]["blend_words"] is None
    assert edit_image_name(7, "a b", "a c") == "0007_a b_a c"


def test_limit_and_categories(tmp_path):
    from dataset.pie_bench_data import PieBenchData
    g = json.loads((GOLD / "pie_bench.json").read_text())
    (tmp_path / "mapping_file.json").write_text(json.dumps(g["mapping"]))
    assert len(PieBenchData(str(tmp_path), skip_img_load=True, limit=2)) == 2
    assert len(list(PieBenchData(str(tmp_path), skip_img_load=True, limit=3))) == 3
    assert list(PieBenchData.categories["9_change_style"]) == list(range(620, 700))


def test_eval_driver_host_logic(tmp_path, monkeypatch):
    """eval.py without a GPU: rank sharding, output naming, None results skipped, skip-existing resume -- engine and editor
    replaced by fakes (the real ones are exercised by tests/test_batch_gpu.py)."""
    import sys
    import torch
    from PIL import Image
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "eta-inversion_amd"))
    import eval as pie_eval
    g = json.loads((GOLD / "pie_bench.json").read_text())
    root = tmp_path / "pie"
    (root / "annotation_images" / "0_random_140").mkdir(parents=True)
    (root / "mapping_file.json").write_text(json.dumps(g["mapping"]))
    for rec in g["records"]:
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(str(root / rec["image_rel"]))
    calls = []

    class FakePipe:
        device = "cpu"

    class FakeEditor:
        def __init__(self, pipe, num_inference_steps=50, edit_method="ptp"):
            pass

        def edit(self, samples):
            calls.append([s["source_prompt"] for s in samples])
            return [None if s["edit_word_idx"] is None or None in s["edit_word_idx"] else
                    {"image": torch.zeros(1, 3, 16, 16), "latent": torch.zeros(1, 4, 2, 2)} for s in samples]
    monkeypatch.setattr(pie_eval, "load_diffusion_model",
                        lambda *a, **k: (FakePipe(), (lambda f: torch.zeros(1, 3, 16, 16), lambda img: np.zeros((16, 16, 3), np.uint8))))
    monkeypatch.setattr(pie_eval, "BatchEditor", FakeEditor)
    out = tmp_path / "res"
    argv = ["--data_path", str(root), "--output", str(out), "--batch", "2", "--steps", "2", "--size", "16"]
    for rank in (0, 1):                                   # two ranks, run one after the other (no collective without --save_latents)
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "1")             # keep torch.distributed out of the CPU test; sharding is shard_indices' job
        pie_eval.main(argv)
    names = sorted(f.name for f in (out / "imgs").glob("*.png"))
    want = sorted(f"{i:04d}_{r['source_prompt']}_{r['target_prompt']}.png" for i, r in enumerate(g["records"]) if None not in r["edit_word_idx"])
    assert names == want
    assert all(len(c) <= 2 for c in calls)                # batches of at most --batch samples
    n_calls = len(calls)
    pie_eval.main(argv)                                   # resume: only the samples without an output (the None ones) are retried
    assert sum(len(c) for c in calls[n_calls:]) == len(g["records"]) - len(want)


def _eval_rank(rank, world, port, root, out, q):
    """one rank of eval.py under gloo with a fake engine: sharding, skip-existing, None results, latent gather, latents.pt on rank 0"""
    import sys
    import torch
    for p in (str(Path(__file__).resolve().parents[1]), str(Path(__file__).resolve().parents[1] / "eta-inversion_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ETAINV_DIST_BACKEND="gloo")
    import eval as pie_eval
    seen = []

    class FakePipe:
        device = "cpu"

    class FakeEditor:
        def __init__(self, pipe, num_inference_steps=50, edit_method="ptp"):
            pass

        def edit(self, samples):
            seen.extend(s["source_prompt"] for s in samples)
            return [None if None in s["edit_word_idx"] else
                    {"image": torch.zeros(1, 3, 16, 16), "latent": torch.full((1, 4, 2, 2), float(len(s["source_prompt"])))} for s in samples]
    pie_eval.load_diffusion_model = lambda *a, **k: (FakePipe(), (lambda f: torch.zeros(1, 3, 16, 16), lambda img: np.zeros((16, 16, 3), np.uint8)))
    pie_eval.BatchEditor = FakeEditor
    pie_eval.main(["--data_path", root, "--output", out, "--batch", "2", "--steps", "2", "--size", "16", "--save_latents"])
    q.put((rank, seen))


def test_eval_two_ranks_gloo(tmp_path):
    """eval.py end to end with WORLD_SIZE = 2 (gloo, fake engine): what the 8-GPU sweep runs apart from the kernels -- image i on rank
    i % 2, every output written exactly once, edited latents of both ranks gathered in image order into latents.pt on rank 0."""
    import torch
    import torch.multiprocessing as mp
    from PIL import Image
    g = json.loads((GOLD / "pie_bench.json").read_text())
    root = tmp_path / "pie"
    (root / "annotation_images" / "0_random_140").mkdir(parents=True)
    (root / "mapping_file.json").write_text(json.dumps(g["mapping"]))
    for rec in g["records"]:
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(str(root / rec["image_rel"]))
    out = tmp_path / "res"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_eval_rank, args=(r, 2, port, str(root), str(out), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    recs = g["records"]
    for r in (0, 1):
        assert res[r] == [recs[i]["source_prompt"] for i in range(r, len(recs), 2)]          # rank r edits images r, r + 2, ...
    names = sorted(f.name for f in (out / "imgs").glob("*.png"))
    assert names == sorted(f"{i:04d}_{r['source_prompt']}_{r['target_prompt']}.png" for i, r in enumerate(recs) if None not in r["edit_word_idx"])
    lat = torch.load(str(out / "latents.pt"))
    assert lat.shape == (len(recs), 4, 2, 2)
    for i, r in enumerate(recs):
        want = 0.0 if None in r["edit_word_idx"] else float(len(r["source_prompt"]))
        assert float(lat[i, 0, 0, 0]) == want
