"""Controller callbacks (reference modules/editing/controller.py:8-57): begin / end / begin_step / end_step / copy."""
from typing import Optional


class ControllerBase:
    def begin(self) -> None:
        pass

    def end(self) -> None:
        pass

    def begin_step(self, latent, *args, **kwargs):
        return latent

    def end_step(self, latent, noise_pred=None, t: Optional[int] = None):
        return latent

    def copy(self, **kwargs) -> "ControllerBase":
        raise NotImplementedError


class ControllerEmpty(ControllerBase):
    def copy(self, **kwargs) -> "ControllerEmpty":
        return self
