"""Editors (reference modules/editing/editor.py:8-135): `Editor.edit(image, source_prompt, target_prompt, cfg, inv_cfg)`
returns {"image_inv", "image", "latent_inv", "latent"} (rows 0 / 1 of the [source, target] batch) or None."""
from typing import Any, Dict, Optional


class Editor:
    def edit(self, image, source_prompt: str, target_prompt: str, cfg: Optional[Dict[str, Any]] = None, **kwargs) -> Dict[str, Any]:
        raise NotImplementedError


def _split(edit_res):
    return {"image_inv": edit_res["image"][0:1], "image": edit_res["image"][1:2],
            "latent_inv": edit_res["latent"][0:1], "latent": edit_res["latent"][1:2]}


class ControllerBasedEditor(Editor):
    def __init__(self, inverter, no_source_backward: bool = False, dft_cfg: Optional[Dict[Any, str]] = None, fake_edit: bool = False) -> None:
        # no_source_backward: the backward pass runs the target prompt only (reference editor.py:100-116); fake_edit: no inversion, the
        # backward pass starts from cfg["zT_gt"] (editor.py:84-88).  Both are meaningful for the plain inverters (diffinv); `etainv` and
        # `dirinv` need the source row and refuse a single-row backward pass.
        self.inverter, self.no_source_backward, self.fake_edit = inverter, no_source_backward, fake_edit
        self.dft_cfg = dft_cfg if dft_cfg is not None else {}

    def make_controller(self, image, source_prompt, target_prompt, inv_res, **kwargs):
        raise NotImplementedError

    def edit(self, image, source_prompt, target_prompt, cfg=None, inv_cfg=None, **kwargs):
        cfg = {**self.dft_cfg} if cfg is None else {**cfg}
        inv_cfg = {} if inv_cfg is None else inv_cfg
        src_context = self.inverter.create_context(source_prompt)
        target_context = self.inverter.create_context(target_prompt)
        zT_gt = cfg.pop("zT_gt", None)
        if self.fake_edit:
            image = None
            inv_res = {"latents": [zT_gt.to(self.inverter.model.device)]}
        else:
            inv_res = self.inverter.invert(image, prompt=source_prompt, context=src_context, inv_cfg=inv_cfg)
        controller = self.make_controller(image=image, source_prompt=source_prompt, target_prompt=target_prompt, inv_res=inv_res, **cfg, **kwargs)
        with self.inverter.use_controller(controller):
            if not self.no_source_backward:
                edit_res = self.inverter.sample(inv_res, context=[src_context, target_context])
                return None if edit_res is None else _split(edit_res)
            edit_res = self.inverter.sample(inv_res, context=[target_context])
            return None if edit_res is None else {"image": edit_res["image"], "latent": edit_res["latent"]}
