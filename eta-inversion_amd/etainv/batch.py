"""B independent (source, target) edits per engine call: the batched counterpart of `Editor.edit` (reference
modules/editing/editor.py:67-118 handles exactly one image; eval.py:89-101 loops over the data set one image at a time).

Host logic only: tokenisation, per-image prompt-to-prompt tables (built by the same `ptp.make_controller` the one-image API uses),
stacking them over the batch, and the order of the native calls (VAE encode -> EtaLoop.invert -> EtaLoop.sample -> VAE decode)."""
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .pipeline import EtaLoop, PtpTables, noise_table


class BatchEditor:
    def __init__(self, pipe, num_inference_steps=50, eta=((0.6, 0), (1, 0.7)), noise_sample_count=10, seed=0, guidance_scale_bwd=7.5,
                 guidance_scale_fwd=1.0, use_mask=True, mask_thres=0.2, edit_method="ptp"):
        assert edit_method in ("ptp", "simple", "masactrl")
        self.pipe, self.S, self.method = pipe, num_inference_steps, edit_method
        self.n_cand, self.seed, self.use_mask = noise_sample_count, seed, use_mask
        pipe.scheduler.set_timesteps(num_inference_steps)
        self.loop = EtaLoop(pipe.engine, S=num_inference_steps, guidance_scale_bwd=guidance_scale_bwd, guidance_scale_fwd=guidance_scale_fwd,
                            eta=eta,
                            noise_sample_count=noise_sample_count, use_mask=use_mask, mask_thres=mask_thres)
        self._noise = None

    # ------------------------------------------------------------------ helpers
    def _embed(self, prompts: List[str]) -> torch.Tensor:
        tok = self.pipe.tokenizer
        ids = tok(prompts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
        return self.pipe.text_encoder(ids.to(self.pipe.device))[0].float()

    @staticmethod
    def _word_tokens(prompts: List[str]) -> torch.Tensor:
        rows = [[p.split(" ").index(w) + 1 for w in p.split(" ")] for p in prompts]      # ptp_editor.py:72 (first occurrence)
        if max(len(r) for r in rows) > 75:
            raise IndexError("a prompt has more than 75 whitespace words: word maps index the 77-token context (ptp.py:296)")
        W = max(len(r) for r in rows)
        return torch.tensor([r + [0] * (W - len(r)) for r in rows], dtype=torch.int32)

    def _ptp_tables(self, samples) -> Optional[PtpTables]:
        from modules.utils import ptp
        tabs = []
        for s in samples:
            cfg = {k: v for k, v in (s["ptp"] or {}).items() if k != "prompts"}
            tabs.append(ptp.make_controller(self.pipe, prompts=[s["source_prompt"], s["target_prompt"]], **cfg))
        t = [c.tables() for c in tabs]
        assert all(x["mapper"] is not None for x in t) or all(x["replace_mat"] is not None for x in t), "one controller kind per batch"
        stack = lambda key, fill: None if all(x[key] is None for x in t) else np.stack([fill if x[key] is None else x[key] for x in t])
        return PtpTables(stack("mapper", None), stack("alphas", None), np.stack([x["cross_alpha"] for x in t], 1), tabs[0].self_replace_steps,
                         self.S, equalizer=stack("equalizer", np.ones(77, np.float32)), blend_alpha=stack("blend_alpha", np.zeros((2, 77), np.float32)),
                         replace_mat=stack("replace_mat", None), device=self.pipe.device)

    # ------------------------------------------------------------------ API
    @torch.no_grad()
    def edit(self, samples: List[Dict[str, Any]]) -> List[Optional[Dict[str, torch.Tensor]]]:
        """samples: dicts with image (1,3,H,W) in [-1,1], source_prompt, target_prompt, edit_word_idx (pair or None), ptp (cfg dict
        of reference ptp.make_controller or None).  Returns one result dict per sample in order, None where the reference
        returns None (an edit word missing from its prompt, eta_inversion.py:385-386)."""
        out: List[Optional[Dict[str, torch.Tensor]]] = [None] * len(samples)
        ok = [i for i, s in enumerate(samples)
              if not self.use_mask or (s.get("edit_word_idx") is not None and None not in tuple(s["edit_word_idx"]))]
        if not ok:
            return out
        sel = [samples[i] for i in ok]
        n, L = len(sel), self.loop.L
        assert n <= self.pipe.engine.max_img, "batch larger than the engine was created for"
        dev = self.pipe.device
        img = torch.cat([s["image"].to(dev).float() for s in sel])
        z0 = (self.pipe.vae.encode(img)["latent_dist"].mean * 0.18215).float().contiguous()
        unc = self._embed([""])[0]
        c_src, c_tgt = self._embed([s["source_prompt"] for s in sel]), self._embed([s["target_prompt"] for s in sel])
        ctx_src = torch.stack([unc.expand_as(c_src), c_src], 1).contiguous()              # (n,2,77,768)
        ctx_tgt = torch.stack([unc.expand_as(c_tgt), c_tgt], 1).contiguous()
        tokens = self._word_tokens([s["source_prompt"] for s in sel]).to(dev) if self.use_mask else None
        inv = self.loop.invert(z0, ctx_src, tokens)
        if self._noise is None:
            self._noise = noise_table(self.S, self.n_cand, L, self.seed, device=dev)      # reseeded per image in the reference: one table
        ew = torch.tensor([s["edit_word_idx"][0] for s in sel]) if self.use_mask else None
        ptp = self._ptp_tables(sel) if self.method == "ptp" else None
        masa = (4, 10) if self.method == "masactrl" else None
        x = self.loop.sample(inv, ctx_src, ctx_tgt, self._noise, edit_word=ew, ptp=ptp, masactrl=masa)   # (2n,4,L,L): [src.., tgt..]
        images = self.pipe.vae.decode(x / 0.18215)["sample"]
        for k, i in enumerate(ok):
            out[i] = {"image_inv": images[k:k + 1], "image": images[n + k:n + k + 1], "latent_inv": x[k:k + 1], "latent": x[n + k:n + k + 1]}
        return out
