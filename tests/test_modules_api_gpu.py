"""The reference's plugin surface on MI355X: load_diffusion_model / load_inverter / load_editor / Editor.edit and the
edit_image.py command line, at a 128x128 image (16x16 latents) with synthetic weights and the stand-in VAE / text encoder."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
SRC, TGT = "a cat sitting next to a mirror", "a tiger sitting next to a mirror"
PTP_CFG = dict(is_replace_controller=False, prompts=[SRC, TGT], cross_replace_steps={"default_": .4}, self_replace_steps=0.6,
               blend_words=(("cat",), ("tiger",)), equilizer_params={"words": ("tiger",), "values": (2,)})


@pytest.fixture(scope="module")
def pipe():
    import modules
    p, (pre, post) = modules.load_diffusion_model("sd15", "cuda", variant="fp16", latent_size=16)
    yield p, pre, post
    p.engine.close()


def _image():
    g = torch.Generator().manual_seed(3)
    return (torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).cuda()


@pytest.mark.parametrize("editor_name,cfg", [("simple", None), ("ptp", PTP_CFG), ("masactrl", None)])
def test_editor_edit_matches_direct_loop(pipe, editor_name, cfg):
    import modules
    from etainv.pipeline import EtaLoop, PtpTables, noise_table
    p, pre, post = pipe
    S = 6
    inverter = modules.load_inverter("etainv", model=p, scheduler="ddim", num_inference_steps=S, eta=[[0.6, 0], [1, 0.7]])
    editor = modules.load_editor(editor_name, inverter=inverter)
    image = _image()
    res = editor.edit(image, SRC, TGT, cfg=None if cfg is None else {**cfg}, inv_cfg=dict(edit_word_idx=(1, 1)))
    assert set(res) == {"image_inv", "image", "latent_inv", "latent"}
    assert res["latent"].shape == (1, 4, 16, 16) and res["image"].shape == (1, 3, 128, 128)
    assert torch.isfinite(res["latent"]).all()
    # same computation through the batched loop API directly
    loop = EtaLoop(p.engine, S=S, eta=[[0.6, 0], [1, 0.7]])
    z0 = inverter.encode(image).float()
    ctx_s, ctx_t = inverter.create_context(SRC)[None], inverter.create_context(TGT)[None]
    words = SRC.split(" ")
    tokens = torch.tensor([[words.index(w) + 1 for w in words]], dtype=torch.int32).cuda()
    inv = loop.invert(z0, ctx_s, tokens)
    ptp = masa = None
    if editor_name == "ptp":
        from modules.utils import ptp as ptp_mod
        t = ptp_mod.make_controller(p, [SRC, TGT], **{k: v for k, v in cfg.items() if k != "prompts"}).tables()
        ptp = PtpTables(t["mapper"][None], t["alphas"][None], t["cross_alpha"][:, None], 0.6, S, equalizer=t["equalizer"][None],
                        blend_alpha=t["blend_alpha"][None])
    elif editor_name == "masactrl":
        masa = (4, 10)
    out = loop.sample(inv, ctx_s, ctx_t, noise_table(S, 10, 16, seed=0), edit_word=torch.tensor([1]), ptp=ptp, masactrl=masa)
    assert torch.equal(out[1:2], res["latent"]) and torch.equal(out[0:1], res["latent_inv"])
    # the source row replays the inversion trajectory: latent_inv == z0 up to rounding of a + (b - a)
    torch.testing.assert_close(res["latent_inv"], z0, rtol=1e-5, atol=1e-5)
    img = post(res["image"])
    assert img.dtype == np.uint8 and img.shape == (128, 128, 3)


def test_per_step_plugin_path_matches_fast_path(pipe):
    """A user-defined controller (unknown to the engine) forces the callback-per-step path of the reference API
    (predict_step_backward + begin_step / end_step); with an identity controller it must agree with the fused loop."""
    import modules
    from modules.editing.controller import ControllerBase
    p, pre, post = pipe
    S = 4
    inverter = modules.load_inverter("etainv", model=p, scheduler="ddim", num_inference_steps=S)
    image = _image()
    ctx_s, ctx_t = inverter.create_context(SRC), inverter.create_context(TGT)
    inv_res = inverter.invert(image, prompt=SRC, context=ctx_s, inv_cfg=dict(edit_word_idx=(1, 1)))
    fast = inverter.sample(inv_res, context=[ctx_s, ctx_t])["latent"]

    class Custom(ControllerBase):
        calls = 0

        def end_step(self, latent, noise_pred=None, t=None):
            Custom.calls += 1
            return latent
    with inverter.use_controller(Custom()):
        slow = inverter.sample(inv_res, context=[ctx_s, ctx_t])["latent"]
    assert Custom.calls == S
    torch.testing.assert_close(slow, fast, rtol=1e-4, atol=1e-4)
    # granular forward API: predict_noise + step_forward == stored inversion trajectory
    lat = inv_res["latents"][0]
    t0 = inverter.get_timesteps_forward()[0]
    eps = inverter.predict_noise(lat, t0, ctx_s, 1, is_fwd=True)
    nxt = inverter.step_forward(eps, t0, lat).prev_sample
    torch.testing.assert_close(nxt, inv_res["latents"][1], rtol=1e-4, atol=1e-4)
    # unsupported input -> None like the reference (eta_inversion.py:385-386)
    assert inverter.invert(image, prompt=SRC, context=ctx_s, inv_cfg=dict(edit_word_idx=(None, None))) is None


@pytest.mark.parametrize("editor_name,mask_cfg", [("ptp", None), ("ptp", dict(mask_eta="bwd_source_target", thres=0.15)),
                                                  ("ptp", dict(mask_eta="fwd_mean", mask_dirinv="bwd_target", target_dirinv=0.5, thres=0.25)),
                                                  ("masactrl", None),
                                                  ("ptp", dict(mask_eta="bwd_source_target", thres=0.15, attn_from_where=["down"])),
                                                  ("ptp", dict(attn_res=8, attn_from_where=["up"]))])   # (L = 16: the (L/2)^2 layers of the up blocks)
def test_per_step_api_with_builtin_controllers(pipe, editor_name, mask_cfg):
    """The reference-style per-step path (controller.begin_step -> UNet -> get_mask -> eta step -> controller.end_step, one pair at a
    time) with the BUILT-IN prompt-to-prompt / MasaCtrl controllers, including the bwd_* mask sources that read the controller's
    attention store (eta_inversion.py:176-183) and a mask_dirinv from another source: must agree with the fused device loop."""
    import modules
    p, pre, post = pipe
    S = 5
    kw = dict(model=p, scheduler="ddim", num_inference_steps=S, eta=(0.3, 0.6))
    if mask_cfg is not None:
        kw["mask_mode_cfg"] = mask_cfg
    inverter = modules.load_inverter("etainv", **kw)
    editor = modules.load_editor(editor_name, inverter=inverter)
    image = _image()
    cfg = {**PTP_CFG} if editor_name == "ptp" else None
    fast = editor.edit(image, SRC, TGT, cfg=None if cfg is None else {**cfg}, inv_cfg=dict(edit_word_idx=(1, 1)))
    inverter.force_per_step = True
    slow = editor.edit(image, SRC, TGT, cfg=None if cfg is None else {**cfg}, inv_cfg=dict(edit_word_idx=(1, 1)))
    inverter.force_per_step = False
    torch.testing.assert_close(slow["latent_inv"], fast["latent_inv"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(slow["latent"], fast["latent"], rtol=1e-4, atol=1e-4)
    assert p.unet.attn_ctrl is None


def test_get_eta_variance_noise_reference_signature(pipe):
    """get_eta_variance_noise(latent_prev, latent, t, noise_pred, generator) with the reference's arguments and result keys
    (eta_inversion.py:330-375) vs the oracle's restatement (pinned by tests/golden/eta_step.npz)."""
    import modules
    from oracle import loop as oloop, schedule as sch
    p, pre, post = pipe
    S, L = 50, 16
    inverter = modules.load_inverter("etainv", model=p, scheduler="ddim", num_inference_steps=S, eta=[[0.6, 0], [1, 0.7]])
    g = torch.Generator().manual_seed(8)
    latent, eps = torch.randn(1, 4, L, L, generator=g), torch.randn(1, 4, L, L, generator=g)
    prev = 0.98 * latent + 0.05 * torch.randn(1, 4, L, L, generator=g)
    t = torch.tensor(860)
    res = inverter.get_eta_variance_noise(prev.cuda(), latent.cuda(), t, eps.cuda(), torch.Generator().manual_seed(5))
    assert {"eta", "variance_noise", "delta", "latent_prev", "latent_prev_rec", "loss"} <= set(res)
    cand = torch.randn((10, 1, 4, L, L), generator=torch.Generator().manual_seed(5))
    o = oloop.EtaInversionOracle(None, S=S, eta=[[0.6, 0], [1, 0.7]], L=L)
    eta, z, best, losses = o.eta_variance_noise(prev, latent, 860, eps, cand)
    rec = sch.ddim_eta_step(latent, eps, o.ac, 860, S, eta, noise=z)
    assert int(res["best_idx"].item()) == best and abs(res["eta"] - eta) < 1e-7
    torch.testing.assert_close(res["variance_noise"].cpu(), z, rtol=0, atol=0)
    torch.testing.assert_close(res["latent_prev_rec"].cpu(), rec, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res["delta"].cpu(), prev - rec, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(res["loss"].cpu(), losses[best], rtol=1e-4, atol=0)


def test_mask_cfg_validation(pipe):
    import modules
    p, pre, post = pipe
    L = p.engine.L
    with pytest.raises(NotImplementedError):                      # cross layers exist at L/2, L/4, L/8 only
        modules.load_inverter("etainv", model=p, num_inference_steps=4, mask_mode_cfg=dict(attn_res=L))
    with pytest.raises(NotImplementedError):                      # the backward-pass store keeps the (L/4)^2 layers
        modules.load_inverter("etainv", model=p, num_inference_steps=4, mask_mode_cfg=dict(attn_res=L // 2, mask_eta="bwd_source"))
    with pytest.raises(ValueError):                               # no (L/4)^2 layer in the mid block: the reference fails in torch.cat([])
        modules.load_inverter("etainv", model=p, num_inference_steps=4, mask_mode_cfg=dict(attn_from_where=["mid"]))
    inv = modules.load_inverter("etainv", model=p, num_inference_steps=4, mask_mode_cfg=dict(attn_res=L // 2, attn_from_where=["up"]))
    assert (inv._loop.attn_div, inv._loop.attn_layer_mask) == (2, 0x1c)
    inv = modules.load_inverter("etainv", model=p, num_inference_steps=4, mask_mode_cfg=dict(attn_res=L // 8, attn_from_where=["down"]))
    assert (inv._loop.attn_div, inv._loop.attn_layer_mask) == (8, 0x01)   # `res == 8` -> mid, whatever from_where says (ptp.py:293-294)
    with pytest.raises(ValueError):
        modules.load_inverter("etainv", model=p, num_inference_steps=4, mask_mode_cfg=dict(mask_eta="nope"))


def test_edit_image_cli(tmp_path):
    from PIL import Image
    src = tmp_path / "in.png"
    Image.fromarray((np.random.default_rng(0).random((96, 96, 3)) * 255).astype(np.uint8)).save(src)
    out = tmp_path / "out.png"
    r = subprocess.run([sys.executable, str(ROOT / "eta-inversion_amd" / "edit_image.py"), "--input", str(src), "--source_prompt", SRC,
                        "--target_prompt", TGT, "--output", str(out), "--inv_method", "etainv", "--edit_method", "ptp", "--steps", "3",
                        "--prec", "fp16"], capture_output=True, text=True, timeout=900,
                       env={**__import__("os").environ, "PYTHONPATH": str(ROOT / "eta-inversion_amd")})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Saved result to" in r.stdout and "Took" in r.stdout
    assert out.exists() and (tmp_path / "out_inv.png").exists()
    assert Image.open(out).size == (512, 512)


def test_dirinv_plugin_vs_oracle():
    """`load_inverter("dirinv")` + simple editor through the plugin API vs the CPU oracle run as eta = 0 / no mask
    (reference modules/inversion/direct_inversion.py:17-58)."""
    from modules import load_diffusion_model, load_inverter, load_editor, get_inversion_methods
    from oracle import loop as oloop
    from tests.oracle_cache import oracle_unet      # one fp32 oracle UNet per session
    assert "dirinv" in get_inversion_methods()
    S, L = 4, 16
    pipe, _ = load_diffusion_model("CompVis/stable-diffusion-v1-4", "cuda", variant="fp16", latent_size=L, max_img=1)
    inv = load_inverter(type="dirinv", model=pipe, scheduler="ddim", num_inference_steps=S)
    ed = load_editor(type="simple", inverter=inv)
    src, tgt = "a cat sitting on a chair", "a tiger sitting on a chair"
    g = torch.Generator().manual_seed(3)
    z0 = 0.8 * torch.randn(1, 4, L, L, generator=g)
    inv.encode = lambda image: image.to("cuda").float()          # feed the latent directly: the VAE is tested elsewhere
    res = ed.edit(z0, src, tgt, inv_cfg=dict(edit_word_idx=(1, 1)))
    ctx_s, ctx_t = inv.create_context(src).cpu(), inv.create_context(tgt).cpu()
    with torch.no_grad():
        o = oloop.EtaInversionOracle(oracle_unet(), S=S, eta=(0.0, 0.0), noise_sample_count=1, use_mask=False, L=L)
        oinv = o.invert(z0, ctx_s, src)
        z = o.sample(oinv, ctx_s, ctx_t, oloop.noise_table(S, 1, L, seed=0), edit_word_idx=(1, 1))
    rel = lambda a, b: ((a.float().cpu() - b).norm() / b.norm()).item()
    assert rel(res["latent_inv"], z[:1]) < 5e-3 and rel(res["latent"], z[1:]) < 3e-2
    # DirectInversion.invert honours the per-call guidance_scale_fwd (direct_inversion.py:60-62): CFG 3 in the inversion pass
    inv_res = inv.invert(z0, prompt=src, context=inv.create_context(src), guidance_scale_fwd=3.0)
    with torch.no_grad():
        o3 = oloop.EtaInversionOracle(oracle_unet(), S=S, eta=(0.0, 0.0), noise_sample_count=1, use_mask=False, L=L, guidance_scale_fwd=3.0)
        ref3 = torch.cat(o3.invert(z0, ctx_s, src)["latents"])
    assert rel(torch.cat(inv_res["latents"]), ref3) < 5e-3
    assert rel(torch.cat(inv_res["latents"]), torch.cat(oinv["latents"])) > 1e-2      # and it really differs from the scale-1 trajectory


@pytest.mark.parametrize("no_source_backward", [False, True])
def test_diffinv_plugin_vs_oracle(no_source_backward):
    """`load_inverter("diffinv")` + simple editor (with and without the source row in the backward pass) through the per-step plugin API
    vs the CPU oracle's restatement of the reference loops (diffusion_inversion.py:388-436, pinned by tests/golden/e2e_diffinv.npz)."""
    from modules import load_diffusion_model, load_inverter, load_editor
    from oracle import loop as oloop
    from tests.oracle_cache import oracle_unet      # one fp32 oracle UNet per session
    S, L = 4, 16
    pipe, _ = load_diffusion_model("CompVis/stable-diffusion-v1-4", "cuda", variant="fp16", latent_size=L, max_img=1)
    inv = load_inverter(type="diffinv", model=pipe, scheduler="ddim", num_inference_steps=S)
    ed = load_editor(type="simple", inverter=inv, no_source_backward=no_source_backward)
    src, tgt = "a cat sitting on a chair", "a tiger sitting on a chair"
    z0 = 0.8 * torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(4))
    inv.encode = lambda image: image.to("cuda").float()          # feed the latent directly: the VAE is tested elsewhere
    res = ed.edit(z0, src, tgt)
    ctx_s, ctx_t = inv.create_context(src).cpu(), inv.create_context(tgt).cpu()
    with torch.no_grad():
        o = oloop.DiffusionInversionOracle(oracle_unet(), S=S)
        z = o.sample(o.invert(z0, ctx_s), [ctx_t] if no_source_backward else [ctx_s, ctx_t])
    rel = lambda a, b: ((a.float().cpu() - b).norm() / b.norm()).item()
    if no_source_backward:
        assert set(res) == {"image", "latent"} and rel(res["latent"], z) < 3e-2
    else:
        assert rel(res["latent_inv"], z[:1]) < 3e-2 and rel(res["latent"], z[1:]) < 3e-2
    pipe.engine.close()


def test_dpm_solver_schedulers_vs_oracle():
    """The native DPM-Solver++(2M) pair (modules/schedulers.py, modules/inverse_schedulers/scheduling_dpmsolver_multistep_inverse.py; latent
    updates through etainv_lincomb3) vs the oracle's independent restatement, forward and backward, on a synthetic noise prediction."""
    from modules.schedulers import DDIMScheduler, DPMSolverMultistepScheduler
    from modules.inverse_schedulers import DPMSolverMultistepInverseScheduler
    from oracle import schedule as sch
    S = 12
    base = DDIMScheduler()
    bwd = DPMSolverMultistepScheduler.from_config(base.config)
    bwd.set_timesteps(S)
    fwd = DPMSolverMultistepInverseScheduler.from_scheduler(bwd)
    fwd.set_timesteps(S)
    assert fwd.timesteps.tolist() == sch.dpm_timesteps_forward(S).tolist() and bwd.timesteps.tolist() == sch.dpm_timesteps_backward(S).tolist()
    ac = sch.alphas_cumprod()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(1, 4, 16, 16, generator=g)
    model = lambda x, t: torch.tanh(0.7 * x) * (0.5 + t / 1000.0)                 # any deterministic eps(x, t)
    of, ob = sch.DpmStepper(ac, fwd.timesteps, 999), sch.DpmStepper(ac, bwd.timesteps, 0)
    xr, xn = x0.double(), x0.cuda()
    for i, t in enumerate(fwd.timesteps):
        xr = of.step(model(xr, int(t)), int(t), xr, i)
        xn = fwd.step(model(xn, int(t)), t, xn).prev_sample
        torch.testing.assert_close(xn.cpu().double(), xr, rtol=2e-5, atol=2e-5)
    for i, t in enumerate(bwd.timesteps):
        xr = ob.step(model(xr, int(t)), int(t), xr, i)
        xn = bwd.step(model(xn, int(t)), t, xn).prev_sample
        torch.testing.assert_close(xn.cpu().double(), xr, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("S,spacing", [(10, "leading"), (10, "linspace"), (50, "leading")])
@pytest.mark.parametrize("mode", ["samesame", "sameshift", "shiftshift"])
def test_dpm_inverse_scheduler_vs_reference_golden(golden, S, spacing, mode):
    """The native DPMSolverMultistepInverseScheduler (latent updates = etainv_lincomb3 on the device) vs the reference's own class
    (tests/golden/dpm_inverse.npz: modules/inverse_schedulers/scheduling_dpmsolver_multistep_inverse.py:10-159 run free over S steps), all three
    `inv_steps` modes incl. the negative first timestep of the shifted ones, fp32 latents."""
    from modules.schedulers import DDIMScheduler, DPMSolverMultistepScheduler
    from modules.inverse_schedulers import DPMSolverMultistepInverseScheduler
    from tests.test_oracle_golden import dpm_golden_run
    g = golden("dpm_inverse")
    bwd = DPMSolverMultistepScheduler.from_config({**DDIMScheduler().config, "timestep_spacing": spacing})

    def make(ts):
        fwd = DPMSolverMultistepInverseScheduler.from_scheduler(bwd, inv_steps=mode)
        fwd.set_timesteps(S)
        assert [int(t) for t in fwd.timesteps] == ts.tolist()
        return lambda eps, t, x: fwd.step(eps.cuda(), torch.tensor(t), x.cuda()).prev_sample

    assert dpm_golden_run(g, S, spacing, mode, make, dtype=torch.float32) < 2e-5


def test_diffinv_dpm_plugin_vs_oracle():
    """`diffinv --scheduler dpm` + simple editor through the plugin API vs the oracle loop driven by the oracle's DPM steppers"""
    from modules import load_diffusion_model, load_inverter, load_editor
    from oracle import loop as oloop, schedule as sch
    from tests.oracle_cache import oracle_unet      # one fp32 oracle UNet per session
    S, L = 4, 16
    pipe, _ = load_diffusion_model("CompVis/stable-diffusion-v1-4", "cuda", variant="fp16", latent_size=L, max_img=1)
    inv = load_inverter(type="diffinv", model=pipe, scheduler="dpm", num_inference_steps=S)
    with pytest.raises(NotImplementedError):
        load_inverter(type="etainv", model=pipe, scheduler="dpm", num_inference_steps=S)
    ed = load_editor(type="simple", inverter=inv)
    src, tgt = "a cat sitting on a chair", "a tiger sitting on a chair"
    z0 = 0.8 * torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(6))
    inv.encode = lambda image: image.to("cuda").float()
    res = ed.edit(z0, src, tgt)
    ctx_s, ctx_t = inv.create_context(src).cpu(), inv.create_context(tgt).cpu()
    ac = sch.alphas_cumprod()
    tf, tb = sch.dpm_timesteps_forward(S), sch.dpm_timesteps_backward(S)
    sf, sb = sch.DpmStepper(ac, tf, 999), sch.DpmStepper(ac, tb, 0)
    with torch.no_grad():
        o = oloop.DiffusionInversionOracle(oracle_unet(), S=S, step_fwd=sf.step, step_bwd=sb.step)
        o.t_fwd, o.t_bwd = tf, tb
        z = o.sample(o.invert(z0, ctx_s), [ctx_s, ctx_t])
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    assert rel(res["latent_inv"], z[:1]) < 3e-2 and rel(res["latent"], z[1:]) < 3e-2
    pipe.engine.close()
