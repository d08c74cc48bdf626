"""Prompt-to-prompt editor (reference modules/editing/ptp_editor.py:17-156).  The controller carries the declarative
tables (modules/utils/ptp.py); the attention edits themselves run in the HIP kernels."""
from typing import Any, Dict, Optional

from ..utils import ptp
from .controller import ControllerBase
from .editor import ControllerBasedEditor


class PromptToPromptControllerBase(ControllerBase):
    def __init__(self, model, controller) -> None:
        self.model, self.controller, self.step_idx = model, controller, None

    def begin(self) -> None:
        self.step_idx = 0

    def end_step(self, latent, noise_pred=None, t=None):
        self.step_idx += 1
        return latent


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
