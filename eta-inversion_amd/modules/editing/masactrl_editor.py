"""MasaCtrl editor (reference modules/editing/masactrl_editor.py:12-69): mutual self-attention from denoising step
`step` on, in transformer blocks >= `layer`; `total_steps` stays 50 as in the reference (masactrl.py:20)."""
from .controller import ControllerBase
from .editor import Editor, _split


class MasactrlController(ControllerBase):
    def __init__(self, step: int, layer: int, model=None):
        self.step, self.layer, self.model, self.step_idx = step, layer, model, 0
        print("MasaCtrl at denoising steps: ", list(range(step, 50)))
        print("MasaCtrl at U-Net layers: ", list(range(layer, 16)))

    # per-step API (the batched device loop reads step / layer directly): hand this step's declarative control to model.unet
    def begin(self) -> None:
        self.step_idx = 0

    def end(self) -> None:
        if self.model is not None:
            self.model.unet.attn_ctrl = None

    def begin_step(self, latent, *args, **kwargs):
        if self.model is not None:
            from etainv import _capi
            from etainv.engine import AttnControl
            self.model.unet.attn_ctrl = AttnControl(mode=_capi.ATTN_MASA, n_img=1, masa_active=self.step <= self.step_idx < 50,
                                                    masa_first_block=self.layer)
        return latent

    def end_step(self, latent, noise_pred=None, t=None):
        if self.model is not None:
            self.model.unet.attn_ctrl = None
        self.step_idx += 1
        return latent


class MasactrlEditor(Editor):
    def __init__(self, inverter, no_null_source_prompt: bool = True, step: int = 4, layer: int = 10) -> None:
        self.inverter, self.model = inverter, inverter.model
        self.no_null_source_prompt, self.step, self.layer = no_null_source_prompt, step, layer

    def edit(self, image, source_prompt, target_prompt, cfg=None, inv_cfg=None):
        assert cfg is None, f"{cfg}"
        inv_cfg = {} if inv_cfg is None else inv_cfg
        src_context = self.inverter.create_context("" if not self.no_null_source_prompt else source_prompt)
        target_context = self.inverter.create_context(target_prompt)
        inv_res = self.inverter.invert(image, context=src_context, prompt=source_prompt, inv_cfg=inv_cfg)
        with self.inverter.use_controller(MasactrlController(self.step, self.layer, self.model)):
            edit_res = self.inverter.sample(inv_res, context=[src_context, target_context])
        return None if edit_res is None else _split(edit_res)
