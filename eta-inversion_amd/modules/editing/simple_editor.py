"""Plain denoising with the target prompt (reference modules/editing/simple_editor.py:8-51)."""
from .editor import Editor, _split


class SimpleEditor(Editor):
    def __init__(self, inverter, no_source_backward: bool = False) -> None:
        self.inverter, self.model, self.no_source_backward = inverter, inverter.model, no_source_backward

    def edit(self, image, source_prompt, target_prompt, cfg=None, inv_cfg=None):
        assert cfg is None
        src_context = self.inverter.create_context(source_prompt)
        target_context = self.inverter.create_context(target_prompt)
        inv_res = self.inverter.invert(image, prompt=source_prompt, context=src_context, guidance_scale_fwd=1, inv_cfg=inv_cfg)
        if not self.no_source_backward:
            edit_res = self.inverter.sample(inv_res, context=[src_context, target_context])
            return None if edit_res is None else _split(edit_res)
        edit_res = self.inverter.sample(inv_res, context=[target_context])      # target prompt only (simple_editor.py:45-51)
        return None if edit_res is None else {"image": edit_res["image"], "latent": edit_res["latent"]}
