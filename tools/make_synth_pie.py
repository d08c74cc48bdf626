#!/usr/bin/env python3
"""A synthetic data tree in PIE-Bench v1's published layout (mapping_file.json + annotation_images/<type>/<id>.jpg), for measuring the sweep
driver's per-rank share on one GPU: N images 512 x 512 (smooth random colour fields, JPEG), bracketed prompts with a
one-word edit, "blended_word" pairs and run-length masks -- the same record shape tests/golden/make_golden.py gen_pie_bench feeds the
reference's own loader.  No network, no real images: this measures throughput (decode, resize, VAE, loop, PNG), not edit quality.

    python tools/make_synth_pie.py --out /tmp/pie_synth --n 88
"""
import argparse
import json
from pathlib import Path

import numpy as np

NOUNS = ["cat", "dog", "horse", "tiger", "bird", "house", "tower", "bridge", "boat", "car", "tree", "flower", "lake", "river", "mountain", "chair"]
PLACES = ["next to a mirror", "on the grass", "near the lake", "in the snow", "under a tree", "on a table", "by the road", "in the garden"]


def smooth_image(rng, size=512):
    """low-frequency colour field + a blob: compresses like a photograph (JPEG ~ 30-60 KB), cheap to make"""
    from PIL import Image
    small = rng.integers(0, 256, size=(8, 8, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small).resize((size, size), Image.BICUBIC)).astype(np.float32)
    yy, xx = np.mgrid[0:size, 0:size]
    cy, cx, r = rng.integers(128, 384, size=3)
    blob = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2.0 * (40 + r / 4) ** 2))[..., None]
    img = img * (1 - 0.6 * blob) + rng.integers(0, 256, size=3) * 0.6 * blob
    img += rng.normal(0, 4, size=img.shape)
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8))


def rle_disc(cy, cx, r, size=512):
    runs = []
    for y in range(max(0, cy - r), min(size, cy + r + 1)):
        half = int((r * r - (y - cy) ** 2) ** 0.5)
        x0, x1 = max(0, cx - half), min(size - 1, cx + half)
        runs += [y * size + x0, x1 - x0 + 1]
    return runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=88)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    root = Path(a.out)
    (root / "annotation_images" / "0_random_140").mkdir(parents=True, exist_ok=True)
    mapping = {}
    for k in range(a.n):
        s, t = rng.choice(len(NOUNS), size=2, replace=False)
        place = PLACES[int(rng.integers(len(PLACES)))]
        src, tgt = f"a [{NOUNS[s]}] sitting {place}", f"a [{NOUNS[t]}] sitting {place}"
        rel = f"0_random_140/{k:012d}.jpg"
        smooth_image(rng).save(str(root / "annotation_images" / rel), quality=90)
        cy, cx, r = (int(v) for v in rng.integers(150, 360, size=3))
        mapping[f"{k:012d}"] = {"image_path": rel, "original_prompt": src, "editing_prompt": tgt, "editing_instruction": f"change the {NOUNS[s]} to a {NOUNS[t]}",
                                "editing_type_id": "0", "blended_word": f"{NOUNS[s]} {NOUNS[t]}", "mask": rle_disc(cy, cx, r // 3 + 20)}
    (root / "mapping_file.json").write_text(json.dumps(mapping))
    print(f"wrote {a.n} images + mapping_file.json under {root}")


if __name__ == "__main__":
    main()
