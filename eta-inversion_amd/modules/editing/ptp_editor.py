"""Prompt-to-prompt editor (reference modules/editing/ptp_editor.py:17-156).  The controller carries the declarative
tables (modules/utils/ptp.py); the attention edits themselves run in the HIP kernels."""
from typing import Any, Dict, Optional

from ..utils import ptp
from .controller import ControllerBase
from .editor import ControllerBasedEditor


class PromptToPromptControllerBase(ControllerBase):
    """Per-step callbacks of the reference (ptp_editor.py:17-98) on the native engine: `begin_step` hands the declarative attention
    control of this step to `model.unet` (what register_attention_control does with Python hooks, ptp_utils.py:196-302), `end_step`
    applies LocalBlend (controller.step_callback, ptp.py:98-101) and advances the step.  The batched device loop
    (etainv.pipeline.EtaLoop) reads the same tables directly and does not call these."""

    def __init__(self, model, controller) -> None:
        self.model, self.controller, self.step_idx = model, controller, None
        self._tables = None

    def begin(self) -> None:
        self.step_idx = 0
        e = self.model.engine                          # AttentionStore.reset (ptp.py:169-172); the backward-pass store keeps the (L/4)^2 layers (LocalBlend)
        e.maps_configure(4) if e.map_div != 4 else e.maps_reset()

    def end(self) -> None:
        self.model.unet.attn_ctrl = None

    def device_tables(self):
        if self._tables is None:
            from etainv.pipeline import PtpTables
            t = self.controller.tables()
            st = lambda a: None if a is None else a[None]
            self._tables = PtpTables(st(t["mapper"]), st(t["alphas"]), t["cross_alpha"][:, None], self.controller.self_replace_steps,
                                     self.controller.num_steps, equalizer=st(t["equalizer"]), blend_alpha=st(t["blend_alpha"]),
                                     replace_mat=st(t["replace_mat"]), device=self.model.device)
        return self._tables

    def begin_step(self, latent, *args, **kwargs):
        from etainv import _capi
        from etainv.engine import AttnControl
        p, i, L = self.device_tables(), self.step_idx, self.model.engine.L
        self.model.unet.attn_ctrl = AttnControl(mode=_capi.ATTN_PTP, n_img=1, store_maps=True, mapper=p.mapper, alphas=p.alphas,
                                                replace_mat=p.replace_mat, equalizer=p.equalizer, cross_alpha=p.cross_alpha[min(i, p.cross_alpha.shape[0] - 1)],
                                                self_replace_active=p.self_lo <= i < p.self_hi, self_max_tokens=(L // 2) ** 2)
        return latent

    def end_step(self, latent, noise_pred=None, t=None):
        p = self.device_tables()
        lb = self.controller.local_blend
        if lb is not None and (self.step_idx + 1) > lb.start_blend:      # LocalBlend.__call__ (ptp.py:31-47): counter already incremented
            latent = self.model.engine.local_blend(latent.float().contiguous(), 1, p.blend_alpha, lb.th[0])
        self.model.unet.attn_ctrl = None
        self.step_idx += 1
        return latent

    def get_attention_map(self, mask_idx, res=None, from_where=None, prompt_idx=0, num_prompts=2, resize=None):
        """word map of one prompt from the backward-pass store, averaged over the steps done (ptp_editor.py:43-85): (1, L, L)"""
        import torch
        e = self.model.engine
        from etainv.pipeline import attn_layer_selection
        assert num_prompts == 2 and (resize is None or resize == e.L)
        div, mask = attn_layer_selection(e.L, res, from_where if from_where is not None else ("up", "down"))
        if div != e.map_div:
            raise NotImplementedError(f"the backward-pass store holds the (L/{e.map_div})^2 cross layers (LocalBlend reads those), not res = {res}")
        tok = torch.tensor([[mask_idx + 1]], dtype=torch.int32, device=self.model.device)
        out = torch.empty(1, 1, e.L, e.L, dtype=torch.float32, device=self.model.device)
        e.word_maps_ex(1, tok, self.step_idx + 1, prompt_idx, mask, out)
        return out[0]


class PromptToPromptController(PromptToPromptControllerBase):
    def __init__(self, model, source_prompt: str, target_prompt: str, inv_res: Optional[Dict[str, Any]] = None, **kwargs) -> None:
        self.source_prompt, self.target_prompt = source_prompt, target_prompt
        self.ptp_cfg = {**kwargs}
        if "prompts" in self.ptp_cfg:
            assert self.ptp_cfg["prompts"] == [source_prompt, target_prompt]
            self.ptp_cfg.pop("prompts")
        super().__init__(model, ptp.make_controller(model, prompts=[source_prompt, target_prompt], **self.ptp_cfg))

    def copy(self, **kwargs):
        return PromptToPromptController(self.model, self.source_prompt, self.target_prompt, **self.ptp_cfg)


class PromptToPromptEditor(ControllerBasedEditor):
    def make_controller(self, image, source_prompt, target_prompt, **kwargs):
        return PromptToPromptController(model=self.inverter.model, source_prompt=source_prompt, target_prompt=target_prompt, **kwargs)
