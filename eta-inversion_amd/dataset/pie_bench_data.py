"""PIE-Bench v1 annotations -> edit jobs for the batch driver (etainv/batch.py, eval.py).

Written from the dataset's published layout, not from the reference loader's code; it yields the records the reference's
dataset/pie_bench_data.py yields (checked field by field and bit by bit against records produced by that loader on a synthetic mapping
file: tests/test_pie_bench.py, tests/golden/pie_bench.json).

Layout of `<root>/mapping_file.json`: {image id: {"image_path", "original_prompt", "editing_prompt", "editing_instruction",
"editing_type_id", "blended_word", "mask"}}.  Prompts mark the edited words with [brackets]; "blended_word" is "<source word> <target
word>" or ""; "mask" is a run-length list [start0, len0, start1, len1, ...] over the row-major 512 x 512 image.  The 700 images come in
ten consecutive editing types (140 random, then blocks of 80 / 40)."""
import copy
import json
import os
from typing import Any, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

_TYPE_BLOCKS = (("0_random", 140), ("1_change_object", 80), ("2_add_object", 80), ("3_delete_object", 80), ("4_change_attribute_content", 40),
                ("5_change_attribute_pose", 40), ("6_change_attribute_color", 40), ("7_change_attribute_material", 40), ("8_change_background", 80),
                ("9_change_style", 80))


def _type_ranges() -> Dict[str, range]:
    out, first = {}, 0
    for name, count in _TYPE_BLOCKS:
        out[name] = range(first, first + count)
        first += count
    return out


CATEGORIES = _type_ranges()
MASK_SHAPE = (512, 512)
# prompt-to-prompt settings the benchmark runs every image with: refinement controller, cross / self replacement for 40 % / 60 % of the steps,
# LocalBlend on the blended word pair and the target word re-weighted by 2
_PTP_FIXED = {"is_replace_controller": False, "cross_replace_steps": {"default_": .4}, "self_replace_steps": 0.6}


def _plain(prompt: str) -> str:
    return prompt.translate({ord("["): None, ord("]"): None})


def _ptp_settings(source: str, target: str, blended: str) -> Dict[str, Any]:
    pair = blended.split(" ") if blended else None
    cfg = {"is_replace_controller": _PTP_FIXED["is_replace_controller"], "prompts": [source, target],
           "cross_replace_steps": dict(_PTP_FIXED["cross_replace_steps"]), "self_replace_steps": _PTP_FIXED["self_replace_steps"],
           "blend_words": None, "equilizer_params": None}
    if pair:
        cfg["blend_words"] = ((pair[0],), (pair[1],))
        cfg["equilizer_params"] = {"words": (pair[1],), "values": (2,)}
    return cfg


def _job(root: str, entry: Dict[str, Any]) -> Dict[str, Any]:
    source, target = _plain(entry["original_prompt"]), _plain(entry["editing_prompt"])
    image_file = os.path.join(f"{root}/annotation_images", entry["image_path"])
    return {"name": image_file, "source_prompt": source, "target_prompt": target, "image_file": image_file,
            "edit": {"target_prompt": target, "ptp": _ptp_settings(source, target, entry["blended_word"])}, "mask": entry["mask"]}


def _position(words: List[str], word: Optional[str]) -> Optional[int]:
    return words.index(word) if word in words else None


def decode_mask(runs: Sequence[int], shape=MASK_SHAPE) -> torch.Tensor:
    """Run-length list -> float32 foreground mask.  Runs are clipped at the end of the image and the outermost one-pixel frame is always
    foreground (the benchmark's convention for its metrics).  Vectorised: +1 / -1 marks at run starts / ends, one cumulative sum."""
    total = shape[0] * shape[1]
    pairs = np.asarray(runs, dtype=np.int64).reshape(-1, 2)
    starts = np.clip(pairs[:, 0], 0, total)
    ends = np.clip(pairs[:, 0] + np.maximum(pairs[:, 1], 0), 0, total)
    marks = np.zeros(total + 1, dtype=np.int64)
    np.add.at(marks, starts, 1)
    np.add.at(marks, ends, -1)
    mask = (np.cumsum(marks[:-1]) > 0).astype(np.float32).reshape(shape)
    mask[[0, -1], :] = 1
    mask[:, [0, -1]] = 1
    return torch.from_numpy(mask)


class PieBenchData:
    """Indexable / iterable set of edit jobs.  `categories` keeps only the named editing types (by position in the file, as the benchmark
    orders them), `limit` truncates, `skip_img_load` leaves `image` None (prompt- and mask-only uses)."""
    categories = CATEGORIES

    def __init__(self, data_path: str = "data/eval/PIE-Bench_v1", skip_img_load: bool = False, limit: Optional[int] = None,
                 categories: Optional[Sequence[str]] = None) -> None:
        with open(f"{data_path}/mapping_file.json", "r") as f:
            entries = list(json.load(f).values())
        jobs = [_job(data_path, e) for e in entries]
        if categories is not None:
            jobs = [jobs[i] for name in categories for i in CATEGORIES[name]]
        self.edit_prompts = jobs
        self.skip_img_load, self.limit = skip_img_load, limit

    def mask_decode(self, encoded_mask: Sequence[int], image_shape=MASK_SHAPE) -> torch.Tensor:
        return decode_mask(encoded_mask, image_shape)

    def __len__(self) -> int:
        return self.limit if self.limit is not None else len(self.edit_prompts)

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        return (self[i] for i in range(len(self)))

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        job = copy.deepcopy(self.edit_prompts[idx])
        ptp = job["edit"]["ptp"]
        blend = ptp["blend_words"]
        source_word, target_word = (blend[0][0], blend[1][0]) if blend is not None else (None, None)
        source, target = ptp["prompts"]
        job["image"] = None
        if not self.skip_img_load:
            from PIL import Image
            job["image"] = np.array(Image.open(job["image_file"]))[:, :, :3]
        job["mask"] = decode_mask(job["mask"])
        job["edit_word_idx"] = [_position(source.split(" "), source_word), _position(target.split(" "), target_word)]
        return job

    def __repr__(self) -> str:
        return f"PieBenchData({len(self)} samples)"


def edit_image_name(i: int, source_prompt: str, target_prompt: str) -> str:
    """file stem of an edited image (reference utils/eval_utils.py:209-222): `{i:04d}_{source}_{target}`"""
    return f"{i:04d}_{source_prompt}_{target_prompt}"
