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

    def invert(self, image, prompt=None, context=None, guidance_scale_fwd=None, inv_cfg=None):
        """The reference's DirectInversion.invert forwards the per-call `guidance_scale_fwd` to the forward loop (direct_inversion.py:60-62,
        diffusion_inversion.py:401: `guidance_scale_fwd or self.guidance_scale_fwd`) -- SimpleEditor passes 1 -- unlike EtaInversion, whose
        predict_noise overrides it (eta_inversion.py:323-324)."""
        lp = self._loop
        saved = (lp.g_fwd, lp.g_fwd_table, lp.skip_uncond_fwd)
        g = guidance_scale_fwd or (self._g_fwd_pair or self.guidance_scale_fwd)
        try:
            if isinstance(g, (tuple, list)):
                import numpy as np
                lp.g_fwd, lp.g_fwd_table, lp.skip_uncond_fwd = 1.0, np.linspace(g[0], g[1], 1000), False
            else:
                lp.g_fwd, lp.g_fwd_table, lp.skip_uncond_fwd = float(g), None, float(g) == 1.0
            return super().invert(image, prompt, context, guidance_scale_fwd, inv_cfg)
        finally:
            lp.g_fwd, lp.g_fwd_table, lp.skip_uncond_fwd = saved
