"""PIE-Bench (DirectInversion) annotation loader -- same records as reference dataset/pie_bench_data.py:30-158.

mapping_file.json entry -> {name, source_prompt, target_prompt, image_file, edit{target_prompt, ptp{...}}, mask}; indexing a
sample adds the image (H,W,3 uint8, or None with skip_img_load), the decoded foreground mask and edit_word_idx.  Host logic only
(json, RLE decode); the RLE decode is vectorised but keeps the reference's clipping and its all-ones one-pixel frame."""
import copy
import json
import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch

CATEGORIES = {                       # reference :13-24
    '0_random': range(0, 140), '1_change_object': range(140, 220), '2_add_object': range(220, 300),
    '3_delete_object': range(300, 380), '4_change_attribute_content': range(380, 420), '5_change_attribute_pose': range(420, 460),
    '6_change_attribute_color': range(460, 500), '7_change_attribute_material': range(500, 540), '8_change_background': range(540, 620),
    '9_change_style': range(620, 700)}


class PieBenchData:
    categories = CATEGORIES

    def __init__(self, data_path: str = "data/eval/PIE-Bench_v1", skip_img_load: bool = False, limit: Optional[int] = None,
                 categories=None) -> None:
        with open(f"{data_path}/mapping_file.json", "r") as f:
            mapping = json.load(f)
        labels = []
        for _, item in mapping.items():
            original_prompt = item["original_prompt"].replace("[", "").replace("]", "")      # :47-48
            editing_prompt = item["editing_prompt"].replace("[", "").replace("]", "")
            image_path = os.path.join(f"{data_path}/annotation_images", item["image_path"])
            blended_word = item["blended_word"].split(" ") if item["blended_word"] != "" else []
            ptp_cfg = dict(                                                                     # :59-70
                is_replace_controller=False,
                prompts=[original_prompt, editing_prompt],
                cross_replace_steps={'default_': .4, },
                self_replace_steps=0.6,
                blend_words=(((blended_word[0],), (blended_word[1],))) if len(blended_word) else None,
                equilizer_params={"words": (blended_word[1],), "values": (2,)} if len(blended_word) else None)
            labels.append(dict(name=image_path, source_prompt=original_prompt, target_prompt=editing_prompt, image_file=image_path,
                               edit=dict(target_prompt=editing_prompt, ptp=ptp_cfg), mask=item["mask"]))
        if categories is not None:
            ind = sum([list(CATEGORIES[cat]) for cat in categories], [])
            labels = [labels[i] for i in ind]
        self.edit_prompts = labels
        self.skip_img_load = skip_img_load
        self.limit = limit

    def mask_decode(self, encoded_mask: List[int], image_shape=(512, 512)) -> torch.Tensor:
        """RLE (start, length) pairs over the flattened image -> float32 mask with a one-pixel frame of ones (:92-108)."""
        length = image_shape[0] * image_shape[1]
        mask = np.zeros((length,), dtype=np.float32)
        enc = np.asarray(encoded_mask, dtype=np.int64).reshape(-1, 2) if len(encoded_mask) else np.zeros((0, 2), np.int64)
        for start, run in enc:
            n = min(int(run), length - int(start))
            if n > 0:
                mask[int(start):int(start) + n] = 1
        mask = mask.reshape(image_shape[0], image_shape[1])
        mask[0, :] = 1
        mask[-1, :] = 1
        mask[:, 0] = 1
        mask[:, -1] = 1
        return torch.from_numpy(mask)

    def __len__(self) -> int:
        return len(self.edit_prompts) if self.limit is None else self.limit

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        edit_prompt = self.edit_prompts[idx]
        image = None
        if not self.skip_img_load:
            from PIL import Image
            image = np.array(Image.open(edit_prompt["image_file"]))[:, :, :3]
        mask = self.mask_decode(edit_prompt["mask"])
        ptp = edit_prompt["edit"]["ptp"]
        if ptp["blend_words"] is not None:
            edit_word_src, edit_word_target = ptp["blend_words"][0][0], ptp["blend_words"][1][0]
        else:
            edit_word_src, edit_word_target = None, None
        source_prompt, target_prompt = ptp["prompts"]
        edit_word_idx = [None, None]                                                           # :137-147
        try:
            edit_word_idx[0] = source_prompt.split(" ").index(edit_word_src)
        except ValueError:
            pass
        try:
            edit_word_idx[1] = target_prompt.split(" ").index(edit_word_target)
        except ValueError:
            pass
        return {**copy.deepcopy(edit_prompt), "image": image, "mask": mask, "edit_word_idx": edit_word_idx}

    def __repr__(self) -> str:
        return f"PieBenchData({len(self)} samples)"


def edit_image_name(i: int, source_prompt: str, target_prompt: str) -> str:
    """file stem of an edited image (reference utils/eval_utils.py:209-222): `{i:04d}_{source}_{target}`"""
    return f"{i:04d}_{source_prompt}_{target_prompt}"
