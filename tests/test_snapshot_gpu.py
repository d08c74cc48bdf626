"""The real-weights route: a diffusers-layout snapshot on disk (unet/, vae/, text_encoder/ safetensors + tokenizer vocab / merges)
selected by ETAINV_SD_PATH.  No checkpoint exists offline, so the snapshot is written from the seeded synthetic weights -- fp16 files,
pre-0.18 diffusers attention names in the VAE -- and the loaded pipeline must reproduce the synthetic-weight pipeline."""
import json
import os
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_snapshot(root: Path):
    from safetensors.torch import save_file
    from oracle.clip import build_clip
    from tests.oracle_cache import oracle_unet      # one fp32 oracle UNet per session
    from oracle.vae import build_vae
    for sub, sd, fname in (("unet", oracle_unet().state_dict(), "diffusion_pytorch_model.fp16.safetensors"),
                           ("text_encoder", build_clip(0).state_dict(), "model.fp16.safetensors")):
        (root / sub).mkdir(parents=True)
        save_file({k: v.half().contiguous() for k, v in sd.items()}, str(root / sub / fname))
    old = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}        # pre-0.18 diffusers VAE attention names
    vae = {}
    for k, v in build_vae(0).state_dict().items():
        if ".attentions." in k:
            for new, o in old.items():
                if f".{new}." in k:
                    k = k.replace(f".{new}.", f".{o}.")
                    if k.endswith(".weight"):
                        v = v[:, :, None, None]                                             # stored as 1x1 convs back then
        vae[k] = v.half().contiguous()
    (root / "vae").mkdir()
    save_file(vae, str(root / "vae" / "diffusion_pytorch_model.fp16.safetensors"))
    sys.path.insert(0, str(Path(__file__).parent))
    from test_host_logic import _toy_bpe
    (root / "tokenizer").mkdir()
    _toy_bpe(root / "tokenizer")


def test_pipeline_from_local_snapshot(tmp_path, monkeypatch):
    from modules import load_diffusion_model
    from modules.utils.tokenizer import ClipBPETokenizer
    _write_snapshot(tmp_path)
    monkeypatch.delenv("ETAINV_SD_PATH", raising=False)
    ref, _ = load_diffusion_model("CompVis/stable-diffusion-v1-4", "cuda", variant="fp16", latent_size=16, max_img=1)
    monkeypatch.setenv("ETAINV_SD_PATH", str(tmp_path))
    pipe, (preproc, postproc) = load_diffusion_model("CompVis/stable-diffusion-v1-4", "cuda", variant="fp16", latent_size=16, max_img=1)
    assert isinstance(pipe.tokenizer, ClipBPETokenizer)
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(2, 4, 16, 16, generator=g).cuda(), torch.randn(2, 77, 768, generator=g).cuda()
    # the fp16 files round the fp32-kept parameters (norm scales, biases) as well: equal up to that rounding
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel(pipe.unet(x, 321, encoder_hidden_states=ctx)["sample"], ref.unet(x, 321, encoder_hidden_states=ctx)["sample"]) < 5e-3
    z = torch.randn(1, 4, 16, 16, generator=g).cuda()
    assert rel(pipe.vae.decode(z)["sample"], ref.vae.decode(z)["sample"]) < 5e-3
    img = torch.rand(1, 3, 128, 128, generator=g).cuda() * 2 - 1
    assert rel(pipe.vae.encode(img)["latent_dist"].mean, ref.vae.encode(img)["latent_dist"].mean) < 5e-3
    ids = pipe.tokenizer(["a cat sitting on a wooden chair"], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    ids = ids.clamp(max=49407)
    assert rel(pipe.text_encoder(ids.cuda())[0], ref.text_encoder(ids.cuda())[0]) < 5e-3
    # and the plugin API runs end to end on it (real BPE tokens -> word indices -> ptp tables)
    from modules import load_editor, load_inverter
    inv = load_inverter(type="etainv", model=pipe, scheduler="ddim", num_inference_steps=2, eta=[[0.6, 0], [1, 0.7]])
    ed = load_editor(type="ptp", inverter=inv)
    src, tgt = "a cat sitting on a wooden chair", "a dog sitting on a wooden chair"
    cfg = dict(is_replace_controller=False, cross_replace_steps={"default_": .4}, self_replace_steps=.6, blend_words=(("cat",), ("dog",)),
               equilizer_params={"words": ("dog",), "values": (2,)})
    res = ed.edit(img, src, tgt, cfg=cfg, inv_cfg=dict(edit_word_idx=(1, 1)))
    assert res is not None and res["image"].shape == (1, 3, 128, 128) and torch.isfinite(res["image"]).all()
