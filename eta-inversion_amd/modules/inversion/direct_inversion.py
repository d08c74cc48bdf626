"""Direct inversion (reference modules/inversion/direct_inversion.py:8-58): DDIM inversion, then a deterministic backward pass
whose source row is overwritten by the stored inversion latent after every step.  It is the eta = 0, mask-free special case of the
eta loop (the variance term vanishes, the best-of-n choice has no effect), so it runs on the same fused kernels."""
from typing import Optional

from .eta_inversion import EtaInversion


class DirectInversion(EtaInversion):
    def __init__(self, model, scheduler: Optional[str] = None, num_inference_steps: Optional[int] = None,
                 guidance_scale_bwd: Optional[float] = None, guidance_scale_fwd: Optional[float] = None, verbose: bool = False) -> None:
        super().__init__(model, scheduler, num_inference_steps, guidance_scale_bwd, guidance_scale_fwd, verbose, eta=(0.0, 0.0),
                         noise_sample_count=1, seed=0, use_mask=False)
