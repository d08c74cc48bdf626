#!/usr/bin/env python3
"""Single-image edit on the MI355X engine; command line of the reference's edit_image.py:133-149 (same flags, same
default prompt-to-prompt config, same outputs `<output>` + `<output stem>_inv<suffix>`, `Saved result to` / `Took` lines).
The UNet, the VAE, the CLIP text encoder and the BPE tokenizer all run natively; without `ETAINV_SD_PATH` (a local diffusers snapshot) they carry
seeded synthetic weights and a word-level tokenizer (plumbing run: no checkpoint exists offline).  `--prec` absent = fp32 like the reference."""
import argparse
import time
from pathlib import Path
from typing import List, Tuple

import torch
from PIL import Image

from modules import load_diffusion_model, load_editor, load_inverter
from modules.inversion.diffusion_inversion import DiffusionInversion


def split_to_words(prompt: str) -> List[str]:
    return (prompt[:-1] if prompt[-1] == "." else prompt).split(" ")


def get_edit_word(source_prompt: str, target_prompt: str) -> Tuple[str, str]:
    s, t = split_to_words(source_prompt), split_to_words(target_prompt)
    if len(s) != len(t):
        return None
    diffs = [(a, b) for a, b in zip(s, t) if a != b]
    return diffs[0] if len(diffs) == 1 else None


@torch.no_grad()
def main(input, model, source_prompt, target_prompt, output, inv_method, edit_method, scheduler, steps, guidance_scale_bwd,
         guidance_scale_fwd, edit_cfg, prec) -> None:
    torch.manual_seed(0)
    input = Path(input)
    if output is None:
        output = str(input.parent / (input.name + "_inv" + input.suffix))
    ldm_stable, (preproc, postproc) = load_diffusion_model(model, "cuda", variant=prec)
    if edit_cfg is None and edit_method in ("ptp", "etaedit"):
        blended_word = get_edit_word(source_prompt, target_prompt)
        if blended_word is None:
            print("Provide a edit_cfg for prompt-to-prompt if source and target prompt differ in more than one word.")
            return
        edit_cfg = dict(is_replace_controller=False, prompts=[source_prompt, target_prompt], cross_replace_steps={'default_': .4},
                        self_replace_steps=0.6, blend_words=((blended_word[0],), (blended_word[1],)),
                        equilizer_params={"words": (blended_word[1],), "values": (2,)})
        print(f"Using default ptp config:\n{edit_cfg}")
    elif edit_cfg is not None:
        import yaml
        edit_cfg = yaml.safe_load(Path(edit_cfg).read_text())
    inverter = load_inverter(model=ldm_stable, type=inv_method, scheduler=scheduler, num_inference_steps=steps,
                             guidance_scale_bwd=guidance_scale_bwd, guidance_scale_fwd=guidance_scale_fwd)
    editor = load_editor(inverter=inverter, type=edit_method)
    image = preproc(input)
    idx = next((i for i, (s, t) in enumerate(zip(source_prompt.split(" "), target_prompt.split(" "))) if s != t), None)
    inv_cfg = dict(edit_word_idx=(idx, idx))
    t1 = time.time()
    edit_res = editor.edit(image, source_prompt, target_prompt, cfg=edit_cfg, inv_cfg=inv_cfg)
    torch.cuda.synchronize()
    t2 = time.time()
    Image.fromarray(postproc(edit_res["image"])).save(output)
    if "image_inv" in edit_res:
        out_inv = Path(output)
        Image.fromarray(postproc(edit_res["image_inv"])).save(str(out_inv.parent / (out_inv.stem + "_inv" + out_inv.suffix)))
    print(f"Saved result to {output}")
    print(f"Took {t2 - t1}s")


def parse_args():
    from modules import get_edit_methods, get_inversion_methods
    p = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter, description="Edits a single image.")
    p.add_argument("--input", required=True, help="Path to image to invert.")
    p.add_argument("--model", default="CompVis/stable-diffusion-v1-4", help="Diffusion Model.")
    p.add_argument("--source_prompt", required=True, help="Prompt to use for inversion.")
    p.add_argument("--target_prompt", required=True, help="Prompt to use for inversion.")
    p.add_argument("--output", help="Path for output image.")
    p.add_argument("--inv_method", metavar="INV_METHOD", choices=get_inversion_methods(), default="etainv", help="Inversion method.")
    p.add_argument("--edit_method", metavar="EDIT_METHOD", choices=get_edit_methods(), default="ptp", help="Editing method.")
    p.add_argument("--edit_cfg", help="Path to yaml file for editor configuration. Often needed for prompt-to-prompt.")
    p.add_argument("--scheduler", help="Which scheduler to use.", choices=DiffusionInversion.get_available_schedulers())
    p.add_argument("--steps", type=int, help="How many diffusion steps to use.")
    p.add_argument("--guidance_scale_bwd", type=int, help="Classifier free guidance scale to use for backward diffusion (denoising).")
    p.add_argument("--guidance_scale_fwd", type=int, help="Classifier free guidance scale to use for forward diffusion (inversion).")
    p.add_argument("--prec", choices=["fp16", "fp32", "bf16"], help="Precision for diffusion (default: fp32, like the reference; fp16 / bf16 = MFMA throughput modes).")
    return vars(p.parse_args())


if __name__ == "__main__":
    main(**parse_args())
