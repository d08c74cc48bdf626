from typing import Any


class DiffusionInverseScheduler:
    """Interface of the reference's modules/inverse_schedulers/diffusion_inverse_scheduler.py:5-29."""

    def set_timesteps(self, num_inference_steps: int) -> None:
        raise NotImplementedError

    def step(self, noise_pred, t, latent, *args, **kwargs) -> Any:
        raise NotImplementedError
