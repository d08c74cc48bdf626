#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference):
    cd /tmp && python /root/repo/tests/golden/make_golden.py [--only NAME]

What it does:
  * installs import stubs for packages the image lacks (diffusers, cv2, torchvision, IPython).
    The `diffusers.DDIMScheduler` stub restates the published [3P] diffusers==0.21.1 scheduler
    (closed form) -- diffusers is not vendored in the reference, so that class
    cannot be imported; everything under `modules/` below IS the reference's own code;
  * pre-registers a bare `modules` package so the eager `modules/__init__.py` is skipped;
  * builds a fake pipeline whose `.unet` is the oracle's UNet restatement (toy widths, seeded
    synthetic weights) exposing 32 `Attention`-named submodules for the reference's hooks;
  * runs the reference's DDIMInverseScheduler / EtaInversion / ptp / seq_aligner / masactrl /
    editors on seeded inputs and stores inputs + outputs as small npz fixtures.
The fixtures travel with the repo; the reference source and this script's import of it do not
(nothing in tests/ reads /root/reference at test time).
"""
import argparse
import importlib.machinery
import json
import os
import sys
import types
import zlib
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))


# ----------------------------------------------------------------------------- stubs
def _mod(name, is_pkg=True, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=is_pkg)
    if is_pkg:
        m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class DDIMScheduler:
    """[3P] diffusers 0.21.1 DDIMScheduler, restated (epsilon prediction, leading spacing)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", **kw):
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                           steps_offset=steps_offset, prediction_type=prediction_type, **kw)
        assert beta_schedule == "scaled_linear"
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        return cls(**dict(config))

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.config.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        direction = (1 - a_p - std_dev_t ** 2) ** 0.5 * model_output
        prev = a_p ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std_dev_t * variance_noise
        return types.SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DPMSolverMultistepInverseSchedulerStub:
    """[3P] diffusers 0.21.1 DPMSolverMultistepInverseScheduler, restated (what the reference's wrapper touches: config, tables as fp32
    tensors, set_timesteps incl. noisiest_timestep, convert_model_output, the first / second order updates, model_outputs,
    lower_order_nums) for epsilon prediction, "dpmsolver++", midpoint, no Karras sigmas, no thresholding."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", solver_order=2,
                 prediction_type="epsilon", algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
                 lambda_min_clipped=-float("inf"), timestep_spacing="linspace", steps_offset=0, **kw):
        assert beta_schedule == "scaled_linear" and prediction_type == "epsilon" and algorithm_type == "dpmsolver++" and solver_type == "midpoint"
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                           solver_order=solver_order, prediction_type=prediction_type, algorithm_type=algorithm_type, solver_type=solver_type,
                           lower_order_final=lower_order_final, lambda_min_clipped=lambda_min_clipped, timestep_spacing=timestep_spacing,
                           steps_offset=steps_offset, **kw)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32).copy())
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0

    @classmethod
    def from_config(cls, config):
        return cls(**{k: v for k, v in dict(config).items() if k not in ("clip_sample", "set_alpha_to_one")})

    def set_timesteps(self, num_inference_steps=None, device=None):
        clipped_idx = torch.searchsorted(torch.flip(self.lambda_t, [0]), self.config.lambda_min_clipped).item()
        self.noisiest_timestep = self.config.num_train_timesteps - 1 - clipped_idx
        if self.config.timestep_spacing == "linspace":
            timesteps = np.linspace(0, self.noisiest_timestep, num_inference_steps + 1).round()[:-1].copy().astype(np.int64)
        elif self.config.timestep_spacing == "leading":
            step_ratio = (self.noisiest_timestep + 1) // (num_inference_steps + 1)
            timesteps = (np.arange(0, num_inference_steps + 1) * step_ratio).round()[:-1].copy().astype(np.int64)
            timesteps += self.config.steps_offset
        else:
            raise ValueError(self.config.timestep_spacing)
        _, unique_indices = np.unique(timesteps, return_index=True)
        timesteps = timesteps[np.sort(unique_indices)]
        self.timesteps = torch.from_numpy(timesteps)
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0

    def convert_model_output(self, model_output, timestep, sample):
        alpha_t, sigma_t = self.alpha_t[timestep], self.sigma_t[timestep]
        return (sample - sigma_t * model_output) / alpha_t

    def dpm_solver_first_order_update(self, model_output, timestep, prev_timestep, sample, noise=None):
        lambda_t, lambda_s = self.lambda_t[prev_timestep], self.lambda_t[timestep]
        alpha_t = self.alpha_t[prev_timestep]
        sigma_t, sigma_s = self.sigma_t[prev_timestep], self.sigma_t[timestep]
        h = lambda_t - lambda_s
        return (sigma_t / sigma_s) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * model_output

    def multistep_dpm_solver_second_order_update(self, model_output_list, timestep_list, prev_timestep, sample, noise=None):
        t, s0, s1 = prev_timestep, timestep_list[-1], timestep_list[-2]
        m0, m1 = model_output_list[-1], model_output_list[-2]
        lambda_t, lambda_s0, lambda_s1 = self.lambda_t[t], self.lambda_t[s0], self.lambda_t[s1]
        alpha_t = self.alpha_t[t]
        sigma_t, sigma_s0 = self.sigma_t[t], self.sigma_t[s0]
        h, h_0 = lambda_t - lambda_s0, lambda_s0 - lambda_s1
        r0 = h_0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        return (sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0 - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * D1


class _Dummy:
    def __init__(self, *a, **k):
        pass


def install_stubs():
    _mod("diffusers", DDIMScheduler=DDIMScheduler, DDPMScheduler=_Dummy, DPMSolverMultistepScheduler=_Dummy,
         DPMSolverMultistepInverseScheduler=DPMSolverMultistepInverseSchedulerStub, StableDiffusionPipeline=_Dummy)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_ddim", False, DDIMSchedulerOutput=_Dummy)
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.stable_diffusion")
    _mod("diffusers.pipelines.stable_diffusion.pipeline_stable_diffusion", False, StableDiffusionPipeline=_Dummy)
    _mod("diffusers.models")
    _mod("diffusers.models.unet_2d_condition", False, UNet2DConditionOutput=_Dummy, UNet2DConditionModel=_Dummy)
    _mod("diffusers.models.attention_processor", False, Attention=_Dummy)
    _mod("diffusers.models.resnet", False, ResnetBlock2D=_Dummy)
    _mod("cv2", False)
    tv = _mod("torchvision")
    tvu = _mod("torchvision.utils", False, save_image=lambda *a, **k: None)
    tv.utils = tvu
    ip = _mod("IPython")
    ipd = _mod("IPython.display", False, display=lambda *a, **k: None)
    ip.display = ipd
    pkg = types.ModuleType("modules")
    pkg.__path__ = [str(REF / "modules")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("modules", None, is_package=True)
    sys.modules["modules"] = pkg
    sys.path.insert(0, str(REF))
    import modules.utils.ptp_utils  # noqa: F401  (must precede modules.utils.ptp: circular import in the reference)


# ----------------------------------------------------------------------------- fake pipeline
class FakeTokenizer:
    """Word-level stand-in (same rule as oracle.ptp.WordTokenizer, restated so the fixtures do not
    depend on the oracle): id = 1000 + crc32(word) % 40000, BOS 49406, EOS/pad 49407."""
    model_max_length = 77

    def __init__(self):
        self.rev = {49406: "<|startoftext|>", 49407: "<|endoftext|>"}

    def encode(self, text):
        ids = [49406]
        for w in text.split(" "):
            if w == "":
                continue
            i = 1000 + zlib.crc32(w.encode()) % 40000
            self.rev[i] = w
            ids.append(i)
        return ids + [49407]

    def decode(self, ids):
        return " ".join(self.rev[int(i)] for i in ids)

    def __call__(self, texts, padding=None, max_length=77, truncation=True, return_tensors="pt"):
        rows = []
        for t in texts:
            ids = self.encode(t)[:max_length]
            rows.append(ids + [49407] * (max_length - len(ids)))
        return types.SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.int64))


def text_embed(ids: torch.Tensor, dim=768):
    """Deterministic stand-in text encoder: per-token-id seeded normal + per-position seeded normal."""
    out = torch.zeros(ids.shape[0], ids.shape[1], dim)
    for b in range(ids.shape[0]):
        for p in range(ids.shape[1]):
            g = torch.Generator().manual_seed(int(ids[b, p]))
            gp = torch.Generator().manual_seed(900000 + p)
            out[b, p] = torch.randn(dim, generator=g) + 0.3 * torch.randn(dim, generator=gp)
    return out


class FakeVAE:
    dtype = torch.float32

    def encode(self, image):     # "image" is already a latent / 0.18215 (VAE is outside the hot loop)
        return {"latent_dist": types.SimpleNamespace(mean=image)}

    def decode(self, z):
        return {"sample": z}


def make_pipe(unet):
    pipe = types.SimpleNamespace()
    pipe.device = torch.device("cpu")
    pipe.unet = unet
    pipe.vae = FakeVAE()
    pipe.tokenizer = FakeTokenizer()
    pipe.text_encoder = lambda ids: (text_embed(ids),)
    pipe.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   clip_sample=False, set_alpha_to_one=False)
    return pipe


def toy_unet(seed=0):
    from oracle.unet import build_unet
    return build_unet(seed, block_out_channels=(32, 64, 128, 128))


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    np.savez_compressed(OUT / f"{name}.npz", **conv)
    print(f"wrote {name}.npz: " + ", ".join(f"{k}{list(np.shape(v))}" for k, v in conv.items()))


# ----------------------------------------------------------------------------- generators
def gen_schedule():
    from modules.inversion.eta_inversion import EtaInversion
    out = {}
    for S in (10, 50, 100):
        inv = EtaInversion(make_pipe(toy_unet()), scheduler="ddim", num_inference_steps=S)
        out[f"t_bwd_{S}"] = inv.get_timesteps_backward().numpy()
        out[f"t_fwd_{S}"] = torch.stack(list(inv.get_timesteps_forward())).numpy()
        out[f"var_{S}"] = np.array([float(inv.scheduler_bwd._get_variance(int(t), int(t) - 1000 // S))
                                    for t in inv.get_timesteps_backward()])
    out["alphas_cumprod"] = inv.scheduler_bwd.alphas_cumprod.numpy()
    out["final_alpha_cumprod"] = np.array(float(inv.scheduler_bwd.final_alpha_cumprod))
    for key, eta in {"lin": (0.0, 0.4), "paper": [[0.6, 0], [1, 0.7]], "paper2": [[0.3, 0], [1, 0.2]],
                     "pow3": [[0.2, 0.1], [0.9, 0.8], 3], "const": 0.25}.items():
        out[f"etas_{key}"] = EtaInversion(make_pipe(toy_unet()), eta=eta).etas
    save("schedule", **out)


def gen_ddim_inverse():
    from modules.inverse_schedulers import DDIMInverseScheduler
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64)
    e = torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64)
    out = {"x": x, "eps": e}
    for S in (10, 50):
        for mode in ("sameshift", "samesame"):
            sch = DDIMInverseScheduler.from_scheduler(make_pipe(None).scheduler, inv_steps=mode)
            sch.set_timesteps(S)
            ts = [int(t) for t in sch.timesteps]
            for t in (ts[0], ts[1], ts[len(ts) // 2], ts[-1]):
                out[f"S{S}_{mode}_t{t}"] = sch.step(e, torch.tensor(t), x).prev_sample
    save("ddim_inverse", **out)


class _ConstUnet:
    """unet stand-in returning a fixed tensor (for eta-step known answers)."""
    dtype = torch.float32

    def __init__(self, out):
        self.out = out

    def __call__(self, x, t, encoder_hidden_states=None):
        return {"sample": self.out}


def gen_eta_step():
    from modules.inversion.eta_inversion import EtaInversion
    from tests.golden.recipes import ETA_CASES, eta_case_inputs, crc
    out = {}
    for name, (eta, t, fp16, use_mask) in ETA_CASES.items():
        inp = eta_case_inputs(name)
        latent, unet_out, src_prev, mask_map, noise = (inp[k] for k in ("latent", "unet_out", "src_prev", "mask_map", "noise"))
        dt = latent.dtype
        cu = _ConstUnet(unet_out)
        cu.dtype = dt
        pipe = make_pipe(cu)
        inv = EtaInversion(pipe, scheduler="ddim", num_inference_steps=50, eta=eta, use_mask=use_mask)
        inv.attn_maps_forward = {"mean": [mask_map, mask_map]}
        gen = torch.Generator().manual_seed(5)          # draws exactly `noise` (eta_inversion.py:156)
        ctx = torch.zeros(4, 77, 8, dtype=dt)
        with inv.use_controller(None):
            new, eps = inv.predict_step_backward(latent.clone(), torch.tensor(t), ctx, source_latent_prev=src_prev,
                                                 generator=gen, mask=None, edit_word_idx=(0, 0))
            res = inv.get_eta_variance_noise(src_prev, latent[:1], torch.tensor(t), eps[:1],
                                             torch.Generator().manual_seed(5))
        losses = torch.square(noise - inv.compute_optimal_variance_noise(src_prev, latent[:1], torch.tensor(t),
                                                                       inv.etas[t], eps[:1])).reshape(10, -1).mean(1)
        best = int(torch.argmin(losses).item())
        assert torch.equal(res["variance_noise"], noise[best]), (name, best)
        out.update({f"{name}/new": new.float(), f"{name}/best": np.array(best), f"{name}/losses": losses.float(),
                    f"{name}/eta": np.array(float(inv.etas[t])),
                    f"{name}/crc": np.array([crc(latent), crc(unet_out), crc(src_prev), crc(mask_map), crc(noise)])})
    save("eta_step", **out)


def gen_eta_step_modes():
    """non-default eta-mask modes of the reference's get_mask / predict_step_backward (eta_inversion.py:159-273)"""
    from modules.inversion.eta_inversion import EtaInversion
    from tests.golden.recipes import ETA_MODE_CASES, eta_case_inputs, crc
    out, t = {}, 980
    for name, mode in ETA_MODE_CASES.items():
        inp = eta_case_inputs(name)
        latent, unet_out, src_prev, mask_map, noise = (inp[k] for k in ("latent", "unet_out", "src_prev", "mask_map", "noise"))
        cu = _ConstUnet(unet_out)
        cu.dtype = latent.dtype
        inv = EtaInversion(make_pipe(cu), scheduler="ddim", num_inference_steps=50, eta=[[0.6, 0], [1, 0.7]], use_mask=True, mask_mode_cfg=mode)
        inv.attn_maps_forward = {"mean": [mask_map, mask_map], t: [mask_map, mask_map]}
        ctx = torch.zeros(4, 77, 8)
        with inv.use_controller(None):
            new, eps = inv.predict_step_backward(latent.clone(), torch.tensor(t), ctx, source_latent_prev=src_prev,
                                                 generator=torch.Generator().manual_seed(5), mask=mask_map, edit_word_idx=(0, 0))
        out.update({f"{name}/new": new.float(), f"{name}/crc": np.array([crc(latent), crc(unet_out), crc(src_prev), crc(mask_map), crc(noise)])})
    save("eta_step_modes", **out)


def gen_eta_step_dirinv():
    """the reference's target_dirinv / mask_dirinv options of predict_step_backward (eta_inversion.py:236-256)"""
    from modules.inversion.eta_inversion import EtaInversion
    from tests.golden.recipes import ETA_DIRINV_CASES, eta_case_inputs, crc, dirinv_gt_mask
    out, t = {}, 980
    for name, mode in ETA_DIRINV_CASES.items():
        inp = eta_case_inputs(name)
        latent, unet_out, src_prev, mask_map, noise = (inp[k] for k in ("latent", "unet_out", "src_prev", "mask_map", "noise"))
        cu = _ConstUnet(unet_out)
        cu.dtype = latent.dtype
        inv = EtaInversion(make_pipe(cu), scheduler="ddim", num_inference_steps=50, eta=[[0.6, 0], [1, 0.7]], use_mask=True, mask_mode_cfg=mode)
        inv.attn_maps_forward = {"mean": [mask_map, mask_map], t: [mask_map, mask_map]}
        with inv.use_controller(None):
            new, eps = inv.predict_step_backward(latent.clone(), torch.tensor(t), torch.zeros(4, 77, 8), source_latent_prev=src_prev,
                                                 generator=torch.Generator().manual_seed(5), mask=dirinv_gt_mask(mask_map), edit_word_idx=(0, 0))
        out.update({f"{name}/new": new.float(), f"{name}/crc": np.array([crc(latent), crc(unet_out), crc(src_prev), crc(mask_map), crc(noise)])})
    save("eta_step_dirinv", **out)


PROMPT_PAIRS = [
    ("a cat sitting next to a mirror", "a tiger sitting next to a mirror"),
    ("a photo of a house on a hill", "a photo of a wooden house on a snowy hill"),
    ("a dog", "a very fluffy dog running"),
    ("two birds sitting on a branch", "two origami birds sitting on a branch"),
    ("a bowl of soup", "a bowl of soup"),
]


def gen_ptp_tables():
    from modules.utils import seq_aligner, ptp_utils, ptp
    tok = FakeTokenizer()
    out = {}
    for i, (a, b) in enumerate(PROMPT_PAIRS):
        mapper, alphas = seq_aligner.get_refinement_mapper([a, b], tok)
        out[f"p{i}/mapper"], out[f"p{i}/alphas"] = mapper, alphas
        out[f"p{i}/ids_a"], out[f"p{i}/ids_b"] = np.array(tok.encode(a)), np.array(tok.encode(b))
        for S in (10, 50):
            out[f"p{i}/ctw_{S}"] = ptp_utils.get_time_words_attention_alpha([a, b], S, {"default_": 0.4}, tok)
        out[f"p{i}/ctw_word"] = ptp_utils.get_time_words_attention_alpha(
            [a, b], 50, {"default_": 0.8, b.split(" ")[1]: (0.1, 0.5)}, tok)
        w = b.split(" ")[1]
        pipe = types.SimpleNamespace(tokenizer=tok)
        out[f"p{i}/eq"] = ptp.get_equalizer(pipe, b, (w,), (2,))
        out[f"p{i}/inds_w1"] = ptp_utils.get_word_inds(b, w, tok)
        out[f"p{i}/inds_i2"] = ptp_utils.get_word_inds(b, 2, tok) if len(b.split(" ")) > 2 else np.array([])
        if len(a.split(" ")) == len(b.split(" ")):
            out[f"p{i}/replace"] = seq_aligner.get_replacement_mapper([a, b], tok)
    save("ptp_tables", **out)
    (OUT / "prompt_pairs.json").write_text(json.dumps(PROMPT_PAIRS, indent=1))


PTP_VARIANTS = {
    "refine": dict(is_replace_controller=False, cross_replace_steps={"default_": .4}, self_replace_steps=.6,
                   blend_words=(("cat",), ("tiger",)), equilizer_params={"words": ("tiger",), "values": (2,)}),
    "replace": dict(is_replace_controller=True, cross_replace_steps={"default_": .8}, self_replace_steps=.4,
                    blend_words=None, equilizer_params=None),
}


def gen_ptp_algebra():
    """Drive the reference controllers (Refine+Reweight+LocalBlend; Replace) and AttentionStore with the seeded
    random probabilities of tests/golden/recipes.py through 32 layers x a few steps, in the real layer order."""
    from modules.utils import ptp
    from modules.editing.ptp_editor import PromptToPromptControllerBase
    from tests.golden.recipes import drive_edit_controller, drive_store_controller
    S = 10
    src, tgt = PROMPT_PAIRS[0]
    pipe = make_pipe(types.SimpleNamespace(dtype=torch.float32))
    pipe.scheduler.set_timesteps(S)
    out = {}
    for variant, cfg in PTP_VARIANTS.items():
        ctrl = ptp.make_controller(pipe, [src, tgt], **cfg)
        ctrl.num_att_layers = 32
        for k, v in drive_edit_controller(ctrl).items():
            out[f"{variant}/{k}"] = v
        out[f"{variant}/cross_alpha"] = ctrl.cross_replace_alpha
    store_ctrl = PromptToPromptControllerBase(pipe, ptp.AttentionStore(max_size=16))
    store_ctrl.controller.num_att_layers = 32
    out["store/maps"] = drive_store_controller(
        store_ctrl.controller,
        lambda: torch.stack([store_ctrl.get_attention_map(src, w, res=16, from_where=["up", "down"], resize=64)
                             for w in src.split(" ")]))
    save("ptp_algebra", **out)


def gen_masactrl():
    from modules.utils.masactrl import MutualSelfAttentionControl
    ed = MutualSelfAttentionControl(4, 10)
    ed.num_att_layers = 32
    g = torch.Generator().manual_seed(77)
    out = {}
    heads, n, d = 8, 16, 8
    for step, layer in ((0, 0), (4, 19), (4, 20), (4, 21), (5, 31), (49, 26), (50, 26)):
        ed.cur_step, ed.cur_att_layer = step, layer
        is_cross = layer % 2 == 1
        q = torch.randn(4 * heads, n, d, generator=g)
        k = torch.randn(4 * heads, 77 if is_cross else n, d, generator=g)
        v = torch.randn(4 * heads, 77 if is_cross else n, d, generator=g)
        sim = torch.einsum("bid,bjd->bij", q, k) * d ** -0.5
        attn = sim.softmax(-1)
        o = ed(q, k, v, sim, attn, is_cross, "up", heads, scale=d ** -0.5)
        out.update({f"s{step}_l{layer}/q": q, f"s{step}_l{layer}/k": k, f"s{step}_l{layer}/v": v, f"s{step}_l{layer}/out": o})
    save("masactrl", **out)


def gen_e2e(S=3):
    """Full reference loop (EtaInversion + editors) on the toy-width oracle UNet at the reference's hard-coded
    64x64 latent size.  Pins the oracle's loop restatement end to end."""
    from modules.inversion.eta_inversion import EtaInversion
    from modules.editing.simple_editor import SimpleEditor
    from modules.editing.ptp_editor import PromptToPromptEditor
    from modules.editing.masactrl_editor import MasactrlEditor
    import modules.utils.masactrl as masa_mod
    src, tgt = PROMPT_PAIRS[0]
    unet = toy_unet(0)
    g = torch.Generator().manual_seed(2024)
    z0 = 0.8 * torch.randn(1, 4, 64, 64, generator=g)
    out = {"z0": z0}
    ptp_cfg = dict(is_replace_controller=False, prompts=[src, tgt], cross_replace_steps={"default_": .4},
                   self_replace_steps=0.6, blend_words=(("cat",), ("tiger",)),
                   equilizer_params={"words": ("tiger",), "values": (2,)})
    for name, eta, S_ in (("simple", (0.0, 0.4), S), ("ptp", [[0.6, 0], [1, 0.7]], 5), ("masactrl", (0.0, 0.4), 6)):
        pipe = make_pipe(unet)
        inv = EtaInversion(pipe, scheduler="ddim", num_inference_steps=S_, eta=eta, noise_sample_count=10, seed=0)
        if name == "simple":
            ed = SimpleEditor(inv)
            cfg = None
        elif name == "ptp":
            ed = PromptToPromptEditor(inv)
            cfg = {**ptp_cfg}
        else:
            ed = MasactrlEditor(inv, step=2, layer=10)
            cfg = None
            # MasaCtrl hard-codes total_steps=50; with S=6 steps 2..5 are active (masactrl.py:20,36)
        trace = []
        orig = inv.predict_step_backward

        def wrapped(*a, _orig=orig, **k):
            new, eps = _orig(*a, **k)
            trace.append((new.clone(), eps.clone()))
            return new, eps
        inv.predict_step_backward = wrapped
        captured = {}
        orig_inv = inv.invert

        def wrapped_inv(*a, _o=orig_inv, **k):
            r = _o(*a, **k)
            captured["inv"] = r
            return r
        inv.invert = wrapped_inv
        res = ed.edit(z0 / 0.18215, src, tgt, cfg=cfg, inv_cfg=dict(edit_word_idx=(1, 1)))
        invr = captured["inv"]
        out[f"{name}/S"] = np.array(S_)
        out[f"{name}/ctx_src"] = invr["context"]
        out[f"{name}/ctx_tgt"] = inv.create_context(tgt)
        out[f"{name}/inv_latents"] = torch.cat(invr["latents"])
        out[f"{name}/maps_mean"] = torch.stack(inv.attn_maps_forward["mean"])
        out[f"{name}/latent_inv"] = res["latent_inv"]
        out[f"{name}/latent"] = res["latent"]
        out[f"{name}/bwd_latents"] = torch.stack([t[0] for t in trace])
        out[f"{name}/bwd_eps"] = torch.stack([t[1] for t in trace])
    out["noise_crc"] = np.array(zlib.crc32(torch.randn((10, 1, 4, 64, 64), generator=torch.Generator().manual_seed(0)).numpy().tobytes()))
    save("e2e_toy", **out)



def gen_e2e_dirinv(S=2):
    """Reference DirectInversion (modules/inversion/direct_inversion.py) + PromptToPromptEditor on the toy-width oracle UNet:
    the eta = 0, mask-free special case of the same loop."""
    from modules.inversion.direct_inversion import DirectInversion
    from modules.editing.ptp_editor import PromptToPromptEditor
    src, tgt = PROMPT_PAIRS[0]
    unet = toy_unet(0)
    z0 = 0.8 * torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2025))
    ptp_cfg = dict(is_replace_controller=False, prompts=[src, tgt], cross_replace_steps={"default_": .4},
                   self_replace_steps=0.6, blend_words=(("cat",), ("tiger",)),
                   equilizer_params={"words": ("tiger",), "values": (2,)})
    pipe = make_pipe(unet)
    inv = DirectInversion(pipe, scheduler="ddim", num_inference_steps=S)
    captured = {}
    orig_inv = inv.invert

    def wrapped_inv(*a, _o=orig_inv, **k):
        r = _o(*a, **k)
        captured["inv"] = r
        return r
    inv.invert = wrapped_inv
    res = PromptToPromptEditor(inv).edit(z0 / 0.18215, src, tgt, cfg={**ptp_cfg}, inv_cfg=dict(edit_word_idx=(1, 1)))
    invr = captured["inv"]
    save("e2e_dirinv", z0=z0, S=np.array(S), ctx_src=invr["context"], ctx_tgt=inv.create_context(tgt), inv_latents=torch.cat(invr["latents"]),
         latent_inv=res["latent_inv"], latent=res["latent"])


def gen_e2e_diffinv(S=3):
    """Reference DiffusionInversion (`diffinv`, modules/inversion/diffusion_inversion.py) + SimpleEditor on the toy UNet, with and without
    the source row in the backward pass (no_source_backward, simple_editor.py:45-51)."""
    from modules.inversion.diffusion_inversion import DiffusionInversion
    from modules.editing.simple_editor import SimpleEditor
    src, tgt = PROMPT_PAIRS[0]
    unet = toy_unet(0)
    z0 = 0.8 * torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2027))
    out = {"z0": z0, "S": np.array(S)}
    for tag, nsb in (("pair", False), ("target_only", True)):
        inv = DiffusionInversion(make_pipe(unet), scheduler="ddim", num_inference_steps=S)
        captured = {}
        orig_inv = inv.invert

        def wrapped_inv(*a, _o=orig_inv, **k):
            r = _o(*a, **k)
            captured["inv"] = r
            return r
        inv.invert = wrapped_inv
        res = SimpleEditor(inv, no_source_backward=nsb).edit(z0 / 0.18215, src, tgt, inv_cfg=None)
        out[f"{tag}/latent"] = res["latent"]
        if not nsb:
            out["ctx_src"], out["ctx_tgt"] = captured["inv"]["context"], inv.create_context(tgt)
            out["inv_latents"] = torch.cat(captured["inv"]["latents"])
            out[f"{tag}/latent_inv"] = res["latent_inv"]
    save("e2e_diffinv", **out)


def gen_e2e_bwdmask(S=2):
    """EtaInversion with the eta mask taken from the BACKWARD-pass controller maps (mask_eta = bwd_source / bwd_source_target,
    eta_inversion.py:176-183) + ptp editor on the toy UNet."""
    from modules.inversion.eta_inversion import EtaInversion
    from modules.editing.ptp_editor import PromptToPromptEditor
    src, tgt = PROMPT_PAIRS[0]
    unet = toy_unet(0)
    z0 = 0.8 * torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2026))
    ptp_cfg = dict(is_replace_controller=False, prompts=[src, tgt], cross_replace_steps={"default_": .4},
                   self_replace_steps=0.6, blend_words=(("cat",), ("tiger",)),
                   equilizer_params={"words": ("tiger",), "values": (2,)})
    out = {"z0": z0, "S": np.array(S)}
    for mode in ("bwd_source", "bwd_source_target"):
        pipe = make_pipe(unet)
        inv = EtaInversion(pipe, scheduler="ddim", num_inference_steps=S, eta=(0.3, 0.6), noise_sample_count=10, seed=0,
                           mask_mode_cfg=dict(mask_eta=mode, thres=0.15))
        res = PromptToPromptEditor(inv).edit(z0 / 0.18215, src, tgt, cfg={**ptp_cfg}, inv_cfg=dict(edit_word_idx=(1, 1)))
        out[f"{mode}/latent_inv"] = res["latent_inv"]
        out[f"{mode}/latent"] = res["latent"]
    save("e2e_bwdmask", **out)


ATTNRES_CASES = {"res32": dict(attn_res=32), "up_only": dict(attn_from_where=["up"]), "mid8": dict(attn_res=8),
                 "down_bwd": dict(attn_from_where=["down"], mask_eta="bwd_source_target", thres=0.15)}


def gen_e2e_attnres(S=2):
    """EtaInversion with non-default `attn_res` / `attn_from_where` of the eta mask (eta_inversion.py:161-162: the (L/2)^2 cross layers, the up or
    the down blocks only, the mid block's 8 x 8 map; forward-pass and backward-pass sources) + ptp editor on the toy UNet."""
    from modules.inversion.eta_inversion import EtaInversion
    from modules.editing.ptp_editor import PromptToPromptEditor
    src, tgt = PROMPT_PAIRS[0]
    unet = toy_unet(0)
    z0 = 0.8 * torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2027))
    ptp_cfg = dict(is_replace_controller=False, prompts=[src, tgt], cross_replace_steps={"default_": .4},
                   self_replace_steps=0.6, blend_words=(("cat",), ("tiger",)),
                   equilizer_params={"words": ("tiger",), "values": (2,)})
    out = {"z0": z0, "S": np.array(S)}
    for name, mm in ATTNRES_CASES.items():
        pipe = make_pipe(unet)
        inv = EtaInversion(pipe, scheduler="ddim", num_inference_steps=S, eta=(0.3, 0.6), noise_sample_count=10, seed=0, mask_mode_cfg=dict(mm))
        res = PromptToPromptEditor(inv).edit(z0 / 0.18215, src, tgt, cfg={**ptp_cfg}, inv_cfg=dict(edit_word_idx=(1, 1)))
        out[f"{name}/maps_mean"] = torch.stack(inv.attn_maps_forward["mean"])
        out[f"{name}/latent_inv"] = res["latent_inv"]
        out[f"{name}/latent"] = res["latent"]
    save("e2e_attnres", **out)


def gen_pie_bench():
    """reference dataset/pie_bench_data.py on a synthetic mapping_file.json (the real PIE-Bench is not in the container):
    records, edit_word_idx and decoded RLE masks"""
    import tempfile
    sys.path.insert(0, str(REF))
    from dataset.pie_bench_data import PieBenchData
    rng = np.random.RandomState(7)
    mapping = {}
    rows = [("a [round] cake with orange frosting", "a [square] cake with orange frosting", "round square"),
            ("a cat sitting on a wooden chair", "a [dog] sitting on a wooden chair", "cat dog"),
            ("a woman with [long] hair", "a woman with [short] hair and a hat", ""),
            ("the house near the lake", "the house near the frozen lake", "lake river"),        # target word missing -> None
            ("a painting of a tower", "a painting of a bridge", "tower bridge")]
    for k, (src, tgt, bw) in enumerate(rows):
        runs, pos = [], 0
        for _ in range(int(rng.randint(0, 6))):
            pos += int(rng.randint(1, 60000))
            runs += [pos, int(rng.randint(1, 3000))]
        if k == 1:
            runs += [512 * 512 - 10, 500]                    # run clipped at the end of the image
        mapping[f"{k:012d}"] = {"image_path": f"0_random_140/{k:012d}.jpg", "original_prompt": src, "editing_prompt": tgt,
                                "editing_instruction": "x", "editing_type_id": "0", "blended_word": bw, "mask": runs}
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "mapping_file.json").write_text(json.dumps(mapping))
        data = PieBenchData(d, skip_img_load=True)
        recs, masks = [], {}
        for i in range(len(data)):
            s = data[i]
            masks[f"mask{i}"] = np.packbits(s["mask"].numpy().astype(np.uint8))
            recs.append({"source_prompt": s["source_prompt"], "target_prompt": s["target_prompt"], "image_rel": os.path.relpath(s["image_file"], d),
                         "edit_word_idx": s["edit_word_idx"], "ptp": json.loads(json.dumps(s["edit"]["ptp"])), "mask_sum": float(s["mask"].sum())})
    (OUT / "pie_bench.json").write_text(json.dumps({"mapping": mapping, "records": recs}, indent=1))
    save("pie_bench_masks", **masks)


def gen_resnet_block():
    """The reference's in-tree restatement of [3P] diffusers ResnetBlock2D.forward (modules/utils/pnp_utils.py:136-185, installed by
    register_conv_control_efficient on unet.up_blocks[1].resnets[1]) run on the ORACLE's ResnetBlock2D parameters: pins the residual block
    arithmetic of oracle/unet.py (GroupNorm -> SiLU -> conv3x3 -> + time_emb_proj(SiLU(temb)) -> GroupNorm -> SiLU -> conv3x3 -> + shortcut)
    at SD1.x width (cat input 2560 -> 1280 with a 1x1 shortcut, the block the reference patches) and at a same-width block (no shortcut)."""
    import torch.nn.functional as F
    from modules.utils.pnp_utils import register_conv_control_efficient
    from oracle.unet import ResnetBlock2D, synthetic_tensor
    out = {}
    for tag, (cin, cout, hw) in {"up1r1": (2560, 1280, 8), "same": (320, 320, 16)}.items():
        blk = ResnetBlock2D(cin, cout).eval()
        for n, prm in blk.named_parameters():
            prm.copy_(synthetic_tensor(f"pin.{tag}.{n}", prm.shape, 3))
        g = torch.Generator().manual_seed(cin + hw)
        x = torch.randn(2, cin, hw, hw, generator=g)
        temb = torch.randn(2, 1280, generator=g)
        # attributes of the diffusers module that the reference's forward reads and the oracle's module does not carry
        blk.nonlinearity, blk.upsample, blk.downsample = F.silu, None, None
        blk.time_embedding_norm, blk.output_scale_factor, blk.dropout = "default", 1.0, torch.nn.Identity()
        blk.t = 5
        fake = types.SimpleNamespace(unet=types.SimpleNamespace(up_blocks=[None, types.SimpleNamespace(resnets=[None, blk])]))
        register_conv_control_efficient(fake, torch.tensor([999]))      # t = 5 is not in the schedule: no feature injection
        y = blk.forward(x, temb)                                         # the reference's conv_forward
        # inputs and weights are regenerated from their seeds by the test (oracle.unet.synthetic_tensor + torch.Generator): only a probe is stored
        out.update({f"{tag}_x_probe": x.flatten()[:8], f"{tag}_y": y.to(torch.float32)})
    save("resnet_block", **out)


def gen_dpm_inverse():
    """The reference's OWN DPMSolverMultistepInverseScheduler (modules/inverse_schedulers/scheduling_dpmsolver_multistep_inverse.py:10-159:
    step-index lookup, the "sameshift" / "shiftshift" timestep shifts incl. the negative first timestep, prev_timestep / noisiest_timestep,
    order selection, model_outputs history) over the restated [3P] solver above, built the way DiffusionInversion.create_schedulers does
    (diffusion_inversion.py:139-165: from_config of the model's DDIM scheduler config).  Free-running recursion over all S steps on seeded
    noise predictions; the test regenerates x0 / eps from the seed (probes stored)."""
    import contextlib
    import io
    from modules.inverse_schedulers import DPMSolverMultistepInverseScheduler
    out = {}
    # [3P] diffusers 0.21.1's DDIMScheduler config carries timestep_spacing = "leading" (its default), which from_config hands to the DPM
    # scheduler -- the DDIMScheduler stub above has no such key, so it is added here; "linspace" (the DPM class's own default) is generated
    # as a second case
    for S, spacing in ((10, "leading"), (10, "linspace"), (50, "leading")):
        base_cfg = {**dict(make_pipe(None).scheduler.config), "timestep_spacing": spacing}
        fake_bwd = types.SimpleNamespace(config=base_cfg)
        for mode in ("samesame", "sameshift", "shiftshift"):
            key = f"S{S}_{spacing}_{mode}"
            sch = DPMSolverMultistepInverseScheduler.from_scheduler(fake_bwd, inv_steps=mode)
            sch.set_timesteps(S)
            ts = [int(t) for t in sch.timesteps]
            g = torch.Generator().manual_seed(1000 * S + len(mode))
            x = torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64)
            out[f"{key}_timesteps"] = np.asarray(ts, dtype=np.int64)
            out[f"{key}_x0_probe"] = x.flatten()[:4].clone()
            xs = []
            with contextlib.redirect_stdout(io.StringIO()):              # (the reference prints "t -> prev_t" every step)
                for j, t in enumerate(sch.timesteps):
                    eps = torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64)
                    x = sch.step(eps, t, x).prev_sample
                    xs.append(x.to(torch.float64))
            out[f"{key}_eps_last_probe"] = eps.flatten()[:4].clone()
            keep = list(range(S)) if S <= 10 else [0, 1, 2, 3, S // 2, S - 2, S - 1]
            out[f"{key}_steps"] = np.asarray(keep, dtype=np.int64)
            out[f"{key}_x"] = torch.stack([xs[j] for j in keep])
    save("dpm_inverse", **out)


GENS = {"schedule": gen_schedule, "ddim_inverse": gen_ddim_inverse, "eta_step": gen_eta_step, "eta_step_modes": gen_eta_step_modes, "eta_step_dirinv": gen_eta_step_dirinv,
        "ptp_tables": gen_ptp_tables, "ptp_algebra": gen_ptp_algebra, "masactrl": gen_masactrl, "e2e": gen_e2e, "e2e_dirinv": gen_e2e_dirinv, "e2e_bwdmask": gen_e2e_bwdmask, "e2e_attnres": gen_e2e_attnres, "pie_bench": gen_pie_bench,
        "resnet_block": gen_resnet_block, "e2e_diffinv": gen_e2e_diffinv, "dpm_inverse": gen_dpm_inverse}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    assert Path.cwd() != REPO, "run from a scratch cwd (reference has an import-time `rm -rf result/...`)"
    install_stubs()
    torch.set_grad_enabled(False)
    for k, f in GENS.items():
        if a.only in (None, k):
            print("==", k)
            f()
