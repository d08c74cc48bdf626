"""CPU oracle vs the golden vectors captured from the reference's own code (tests/golden/make_golden.py)."""
import json
import zlib

import numpy as np
import pytest
import torch

from oracle import schedule as sch
from oracle import ptp as optp
from oracle import loop as oloop
from tests.golden import recipes

GOLDEN_DIR = recipes.__file__.rsplit("/", 1)[0]


def test_schedule_tables(golden):
    g = golden("schedule")
    ac = sch.alphas_cumprod()
    np.testing.assert_array_equal(ac, g["alphas_cumprod"])          # bit-exact fp32 table
    assert float(g["final_alpha_cumprod"]) == float(ac[0])
    for S in (10, 50, 100):
        np.testing.assert_array_equal(sch.timesteps_backward(S), g[f"t_bwd_{S}"])
        np.testing.assert_array_equal(sch.timesteps_forward(S), g[f"t_fwd_{S}"])
        var = np.array([sch.variance(ac, int(t), S) for t in sch.timesteps_backward(S)])
        np.testing.assert_allclose(var, g[f"var_{S}"], rtol=5e-5, atol=1e-9)  # ref: fp32 scalars
    for key, eta in {"lin": (0.0, 0.4), "paper": [[0.6, 0], [1, 0.7]], "paper2": [[0.3, 0], [1, 0.2]],
                     "pow3": [[0.2, 0.1], [0.9, 0.8], 3], "const": 0.25}.items():
        np.testing.assert_allclose(sch.eta_table(eta), g[f"etas_{key}"], rtol=1e-12, atol=1e-15)


def test_ddim_inverse_known_answers(golden):
    g = golden("ddim_inverse")
    ac = sch.alphas_cumprod()
    x, e = torch.from_numpy(g["x"]), torch.from_numpy(g["eps"])
    n = 0
    for key in g.files:
        if not key.startswith("S"):
            continue
        S, mode, t = key.split("_")
        S, t = int(S[1:]), int(t[1:])
        a_from, a_to = sch.ddim_inverse_coeffs(ac, t, S, mode)
        out = sch.ddim_step(x, e, a_from, a_to)
        # reference scalars are fp32 0-dim tensors, oracle scalars float64: agree to fp32 rounding
        np.testing.assert_allclose(out.numpy(), g[key], rtol=5e-6, atol=5e-6)
        n += 1
    assert n == 16


@pytest.mark.parametrize("name", list(recipes.ETA_CASES))
def test_eta_step_known_answers(golden, name):
    g = golden("eta_step")
    eta_spec, t, fp16, use_mask = recipes.ETA_CASES[name]
    inp = recipes.eta_case_inputs(name)
    crcs = [recipes.crc(inp[k]) for k in ("latent", "unet_out", "src_prev", "mask_map", "noise")]
    assert crcs == list(g[f"{name}/crc"]), "seeded input recipe drifted (torch RNG changed?)"

    class U:
        def __call__(self, x, t, encoder_hidden_states=None):
            return {"sample": inp["unet_out"]}

        def set_ctrl(self, c):
            pass
    o = oloop.EtaInversionOracle(U(), S=50, eta=eta_spec, use_mask=use_mask)
    assert float(o.etas[t]) == float(g[f"{name}/eta"])
    ctx = torch.zeros(4, 77, 8, dtype=inp["latent"].dtype)
    new, eps, best, losses = o.step_backward(inp["latent"].clone(), t, ctx, inp["src_prev"], inp["noise"],
                                             inp["mask_map"], None)
    if fp16:
        # reference evaluates in fp16 (overflowing losses -> argmin 0, SURVEY E-7); oracle mirrors dtype
        assert int(g[f"{name}/best"]) == best
        np.testing.assert_allclose(new.float().numpy(), g[f"{name}/new"], rtol=2e-2, atol=5e-2)
        return
    gl = g[f"{name}/losses"]
    if np.isfinite(gl).all():
        np.testing.assert_allclose(losses.numpy(), gl, rtol=1e-4)
    assert best == int(g[f"{name}/best"])
    np.testing.assert_allclose(new.numpy(), g[f"{name}/new"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", list(recipes.ETA_MODE_CASES))
def test_eta_step_mask_modes(golden, name):
    """non-default eta-mask modes (gt / fwd maps, no threshold, pow) vs the reference's get_mask + predict_step_backward"""
    g = golden("eta_step_modes")
    mode = recipes.ETA_MODE_CASES[name]
    inp = recipes.eta_case_inputs(name)
    assert [recipes.crc(inp[k]) for k in ("latent", "unet_out", "src_prev", "mask_map", "noise")] == list(g[f"{name}/crc"])

    class U:
        def __call__(self, x, t, encoder_hidden_states=None):
            return {"sample": inp["unet_out"]}

        def set_ctrl(self, c):
            pass
    o = oloop.EtaInversionOracle(U(), S=50, eta=[[0.6, 0], [1, 0.7]], use_mask=True, thres=mode.get("thres", 0.2), mask_eta=mode["mask_eta"],
                                 mask_pow=mode.get("pow"))
    new, eps, best, losses = o.step_backward(inp["latent"].clone(), 980, torch.zeros(4, 77, 8), inp["src_prev"], inp["noise"], inp["mask_map"], None)
    np.testing.assert_allclose(new.numpy(), g[f"{name}/new"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", list(recipes.ETA_DIRINV_CASES))
def test_eta_step_target_dirinv(golden, name):
    """target_dirinv / mask_dirinv of the reference's predict_step_backward (eta_inversion.py:236-256)"""
    g = golden("eta_step_dirinv")
    mode = recipes.ETA_DIRINV_CASES[name]
    inp = recipes.eta_case_inputs(name)
    assert [recipes.crc(inp[k]) for k in ("latent", "unet_out", "src_prev", "mask_map", "noise")] == list(g[f"{name}/crc"])

    class U:
        def __call__(self, x, t, encoder_hidden_states=None):
            return {"sample": inp["unet_out"]}

        def set_ctrl(self, c):
            pass
    o = oloop.EtaInversionOracle(U(), S=50, eta=[[0.6, 0], [1, 0.7]], use_mask=True, thres=mode.get("thres", 0.2), mask_eta=mode["mask_eta"],
                                 mask_pow=mode.get("pow"), target_dirinv=mode["target_dirinv"], mask_dirinv=mode.get("mask_dirinv"))
    gt = recipes.dirinv_gt_mask(inp["mask_map"])
    src = {"gt": gt, "fwd": inp["mask_map"], "fwd_mean": inp["mask_map"]}
    new, eps, best, losses = o.step_backward(inp["latent"].clone(), 980, torch.zeros(4, 77, 8), inp["src_prev"], inp["noise"], src[mode["mask_eta"]], None,
                                             dirinv_map=src[mode["mask_dirinv"]] if mode.get("mask_dirinv") else None)
    np.testing.assert_allclose(new.numpy(), g[f"{name}/new"], rtol=1e-5, atol=2e-5)


def test_ptp_tables(golden):
    g = golden("ptp_tables")
    pairs = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))
    tok = optp.WordTokenizer()
    for i, (a, b) in enumerate(pairs):
        np.testing.assert_array_equal(np.array(tok.encode(a)), g[f"p{i}/ids_a"])
        mapper, alphas = optp.refinement_mapper(a, b, tok)
        np.testing.assert_array_equal(mapper, g[f"p{i}/mapper"][0])
        np.testing.assert_array_equal(alphas, g[f"p{i}/alphas"][0])
        for S in (10, 50):
            np.testing.assert_array_equal(optp.time_words_alpha([a, b], S, {"default_": 0.4}, tok),
                                          g[f"p{i}/ctw_{S}"].reshape(S + 1, 1, 77))
        w = b.split(" ")[1]
        np.testing.assert_array_equal(
            optp.time_words_alpha([a, b], 50, {"default_": 0.8, w: (0.1, 0.5)}, tok), g[f"p{i}/ctw_word"].reshape(51, 1, 77))
        np.testing.assert_array_equal(optp.equalizer(b, (w,), (2,), tok), g[f"p{i}/eq"][0])
        np.testing.assert_array_equal(optp.word_inds(b, w, tok), g[f"p{i}/inds_w1"])
        if len(b.split(" ")) > 2:
            np.testing.assert_array_equal(optp.word_inds(b, 2, tok), g[f"p{i}/inds_i2"])
        if f"p{i}/replace" in g.files:
            np.testing.assert_array_equal(optp.replacement_mapper(a, b, tok), g[f"p{i}/replace"][0])


PTP_VARIANTS = {
    "refine": dict(is_replace_controller=False, cross_replace_steps={"default_": .4}, self_replace_steps=.6,
                   blend_words=(("cat",), ("tiger",)), equilizer_params={"words": ("tiger",), "values": (2,)}),
    "replace": dict(is_replace_controller=True, cross_replace_steps={"default_": .8}, self_replace_steps=.4,
                    blend_words=None, equilizer_params=None),
}


@pytest.mark.parametrize("variant", list(PTP_VARIANTS))
def test_ptp_controller_algebra(golden, variant):
    g = golden("ptp_algebra")
    src, tgt = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))[0]
    ctrl = optp.make_edit_controller(src, tgt, 10, optp.WordTokenizer(), **PTP_VARIANTS[variant])
    np.testing.assert_array_equal(ctrl.cross_alpha.numpy().reshape(11, 77), g[f"{variant}/cross_alpha"].reshape(11, 77))
    out = recipes.drive_edit_controller(ctrl)
    keys = [k for k in g.files if k.startswith(variant + "/") and not k.endswith("cross_alpha")]
    assert len(keys) == len(out)
    for k in keys:
        np.testing.assert_allclose(out[k.split("/", 1)[1]].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_attention_store_maps(golden):
    g = golden("ptp_algebra")
    src, _ = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))[0]
    store = optp.AttentionStore()
    words = src.split(" ")
    maps = recipes.drive_store_controller(
        store, lambda: torch.stack([optp.attention_map(store, words.index(w) + 1, res=16, resize=64) for w in words]))
    np.testing.assert_allclose(maps.numpy(), g["store/maps"], rtol=1e-5, atol=1e-6)


def test_masactrl(golden):
    g = golden("masactrl")
    cases = sorted({k.split("/")[0] for k in g.files})
    assert len(cases) == 7
    for c in cases:
        step, layer = int(c.split("_")[0][1:]), int(c.split("_")[1][1:])
        m = oloop.MasaCtrl(4, 10)
        m.cur_step, m.cur_att_layer = step, layer
        q, k, v = (torch.from_numpy(g[f"{c}/{n}"]) for n in "qkv")
        out = m(layer % 2 == 1, layer, "up", q, k, v, q.shape[-1] ** -0.5, 8)
        np.testing.assert_allclose(out.numpy(), g[f"{c}/out"], rtol=1e-5, atol=1e-6, err_msg=c)
        # the fused-SDPA path used above fast_n tokens (768^2 images) is the same function: force it on the reference's cases
        m2 = oloop.MasaCtrl(4, 10, fast_n=0)
        m2.cur_step, m2.cur_att_layer = step, layer
        out2 = m2(layer % 2 == 1, layer, "up", q, k, v, q.shape[-1] ** -0.5, 8)
        np.testing.assert_allclose(out2.numpy(), g[f"{c}/out"], rtol=1e-5, atol=2e-6, err_msg=c + " (sdpa)")


# ----------------------------------------------------------------------------- end-to-end loop replay
@pytest.fixture(scope="module")
def toy_unet():
    from oracle.unet import build_unet
    return build_unet(0, block_out_channels=(32, 64, 128, 128))


E2E = {"simple": dict(eta=(0.0, 0.4)), "ptp": dict(eta=[[0.6, 0], [1, 0.7]]), "masactrl": dict(eta=(0.0, 0.4))}


@pytest.mark.parametrize("name", ["ptp", pytest.param("simple", marks=pytest.mark.slow),
                                  pytest.param("masactrl", marks=pytest.mark.slow)])
def test_e2e_loop_vs_reference(golden, toy_unet, name):
    """The reference's EtaInversion + editor ran on this same toy-width UNet at its hard-coded 64x64 latent
    (make_golden.gen_e2e); the oracle's loop restatement must land on the same latents."""
    g = golden("e2e_toy")
    S = int(g[f"{name}/S"])
    src, tgt = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))[0]
    noise = oloop.noise_table(S, 10, 64, seed=0)
    assert zlib.crc32(noise[0].numpy().tobytes()) == int(g["noise_crc"])
    z0 = torch.from_numpy(g["z0"])
    ctx_s, ctx_t = torch.from_numpy(g[f"{name}/ctx_src"]), torch.from_numpy(g[f"{name}/ctx_tgt"])
    with torch.no_grad():
        o = oloop.EtaInversionOracle(toy_unet, S=S, **E2E[name])
        inv = o.invert(z0, ctx_s, src)
        np.testing.assert_allclose(torch.cat(inv["latents"]).numpy(), g[f"{name}/inv_latents"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(torch.stack(inv["attn_maps_mean"]).numpy(), g[f"{name}/maps_mean"], rtol=1e-3, atol=1e-4)
        controller = masa = None
        if name == "ptp":
            controller = optp.make_edit_controller(src, tgt, S, optp.WordTokenizer(), **PTP_VARIANTS["refine"])
        elif name == "masactrl":
            masa = oloop.MasaCtrl(2, 10)
        trace = []
        z = o.sample(inv, ctx_s, ctx_t, noise, edit_word_idx=(1, 1), controller=controller, masactrl=masa, trace=trace)
    np.testing.assert_allclose(torch.stack([t["eps"] for t in trace]).numpy(), g[f"{name}/bwd_eps"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(torch.stack([t["latent"] for t in trace]).numpy(), g[f"{name}/bwd_latents"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(z[:1].numpy(), g[f"{name}/latent_inv"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(z[1:].numpy(), g[f"{name}/latent"], rtol=1e-3, atol=2e-4)


def test_dirinv_is_eta_zero(golden, toy_unet):
    """Reference DirectInversion + ptp editor on the toy UNet (make_golden.gen_e2e_dirinv) == the oracle loop with eta = 0 and no
    mask: pins the claim that dirinv is a special case of the same path."""
    g = golden("e2e_dirinv")
    S = int(g["S"])
    src, tgt = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))[0]
    z0 = torch.from_numpy(g["z0"])
    ctx_s, ctx_t = torch.from_numpy(g["ctx_src"]), torch.from_numpy(g["ctx_tgt"])
    with torch.no_grad():
        o = oloop.EtaInversionOracle(toy_unet, S=S, eta=(0.0, 0.0), noise_sample_count=1, use_mask=False)
        inv = o.invert(z0, ctx_s, src)
        np.testing.assert_allclose(torch.cat(inv["latents"]).numpy(), g["inv_latents"], rtol=1e-4, atol=2e-5)
        controller = optp.make_edit_controller(src, tgt, S, optp.WordTokenizer(), **PTP_VARIANTS["refine"])
        z = o.sample(inv, ctx_s, ctx_t, oloop.noise_table(S, 1, 64, seed=0), edit_word_idx=(1, 1), controller=controller)
    np.testing.assert_allclose(z[:1].numpy(), g["latent_inv"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(z[1:].numpy(), g["latent"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("mode", [pytest.param("bwd_source", marks=pytest.mark.slow), "bwd_source_target"])
def test_bwd_mask_sources_vs_reference(golden, toy_unet, mode):
    """eta mask from the backward-pass controller's maps (eta_inversion.py:176-183) through the reference's EtaInversion + ptp editor
    on the toy UNet (make_golden.gen_e2e_bwdmask)"""
    g = golden("e2e_bwdmask")
    S = int(g["S"])
    src, tgt = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))[0]
    tok = optp.WordTokenizer()
    from tests.golden.make_golden import text_embed            # the generator's stand-in text encoder (no reference code involved)
    emb = lambda p: text_embed(torch.tensor([tok.pad_ids(p)]))[0]
    ctx_s, ctx_t = torch.stack([emb(""), emb(src)]), torch.stack([emb(""), emb(tgt)])
    z0 = torch.from_numpy(g["z0"])
    with torch.no_grad():
        o = oloop.EtaInversionOracle(toy_unet, S=S, eta=(0.3, 0.6), use_mask=True, thres=0.15, mask_eta=mode)
        inv = o.invert(z0, ctx_s, src)
        controller = optp.make_edit_controller(src, tgt, S, tok, **PTP_VARIANTS["refine"])
        z = o.sample(inv, ctx_s, ctx_t, oloop.noise_table(S, 10, 64, seed=0), edit_word_idx=(1, 1), controller=controller)
    np.testing.assert_allclose(z[:1].numpy(), g[f"{mode}/latent_inv"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(z[1:].numpy(), g[f"{mode}/latent"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("case", ["res32", pytest.param("up_only", marks=pytest.mark.slow), "mid8", "down_bwd"])
def test_attn_res_and_from_where_vs_reference(golden, toy_unet, case):
    """non-default `attn_res` / `attn_from_where` of the eta mask (eta_inversion.py:161-162; aggregate_attention ptp.py:288-303 incl. its `res == 8 -> mid`
    rule) through the reference's EtaInversion + ptp editor on the toy UNet (make_golden.gen_e2e_attnres)"""
    from tests.golden.make_golden import ATTNRES_CASES, text_embed
    g = golden("e2e_attnres")
    S = int(g["S"])
    mm = ATTNRES_CASES[case]
    src, tgt = json.load(open(f"{GOLDEN_DIR}/prompt_pairs.json"))[0]
    tok = optp.WordTokenizer()
    emb = lambda p: text_embed(torch.tensor([tok.pad_ids(p)]))[0]
    ctx_s, ctx_t = torch.stack([emb(""), emb(src)]), torch.stack([emb(""), emb(tgt)])
    z0 = torch.from_numpy(g["z0"])
    with torch.no_grad():
        o = oloop.EtaInversionOracle(toy_unet, S=S, eta=(0.3, 0.6), use_mask=True, thres=mm.get("thres", 0.2), mask_eta=mm.get("mask_eta", "fwd_mean"),
                                     attn_res=mm.get("attn_res", 16), attn_from_where=mm.get("attn_from_where", ("up", "down")))
        inv = o.invert(z0, ctx_s, src)
        np.testing.assert_allclose(torch.stack(inv["attn_maps_mean"]).numpy(), g[f"{case}/maps_mean"], rtol=1e-3, atol=2e-5)
        controller = optp.make_edit_controller(src, tgt, S, tok, **PTP_VARIANTS["refine"])
        z = o.sample(inv, ctx_s, ctx_t, oloop.noise_table(S, 10, 64, seed=0), edit_word_idx=(1, 1), controller=controller)
    np.testing.assert_allclose(z[:1].numpy(), g[f"{case}/latent_inv"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(z[1:].numpy(), g[f"{case}/latent"], rtol=1e-3, atol=2e-4)


def test_clip_oracle_matches_transformers():
    """oracle/clip.py is pinned by the third-party implementation itself where it is importable: transformers'
    CLIPTextModel (ViT-L/14 text config) loaded with the oracle's seeded weights gives the same hidden states."""
    transformers = pytest.importorskip("transformers")
    from oracle.clip import build_clip
    cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                      num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu")
    hf = transformers.CLIPTextModel(cfg).eval()
    ora = build_clip(0)
    sd = {k: v for k, v in ora.state_dict().items()}
    if not any(k.startswith("text_model.") for k in hf.state_dict()):       # newer transformers dropped the prefix
        sd = {k[len("text_model."):]: v for k, v in sd.items()}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    ids = torch.randint(0, 49408, (2, 77), generator=torch.Generator().manual_seed(0))
    ids[:, 0], ids[:, 30:] = 49406, 49407
    with torch.no_grad():
        want = hf(ids)[0]
        got = ora(ids)[0]
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag,cin,cout,hw", [("up1r1", 2560, 1280, 8), ("same", 320, 320, 16)])
def test_resnet_block_matches_reference_restatement(golden, tag, cin, cout, hw):
    """oracle.unet.ResnetBlock2D vs the reference's own restatement of the [3P] diffusers forward (pnp_utils.py:136-185) run on the
    same parameters (fixture: tests/golden/make_golden.py gen_resnet_block): pins the residual block of the oracle UNet numerically."""
    from oracle.unet import ResnetBlock2D, synthetic_tensor
    g = golden("resnet_block")
    blk = ResnetBlock2D(cin, cout).eval()
    with torch.no_grad():
        for n, prm in blk.named_parameters():
            prm.copy_(synthetic_tensor(f"pin.{tag}.{n}", prm.shape, 3))
        gen = torch.Generator().manual_seed(cin + hw)
        x = torch.randn(2, cin, hw, hw, generator=gen)
        temb = torch.randn(2, 1280, generator=gen)
        assert np.array_equal(x.flatten()[:8].numpy(), g[f"{tag}_x_probe"])
        y = blk(x, temb)
    ref = torch.from_numpy(g[f"{tag}_y"])
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5), float((y - ref).abs().max())


def test_diffinv_vs_reference(golden, toy_unet):
    """`diffinv` (plain DDIM inversion + deterministic sampling, no source replay): oracle.loop.DiffusionInversionOracle vs the reference's
    DiffusionInversion + SimpleEditor on the toy UNet, with and without the source row in the backward pass (make_golden.gen_e2e_diffinv)"""
    g = golden("e2e_diffinv")
    S = int(g["S"])
    z0 = torch.from_numpy(g["z0"])
    ctx_s, ctx_t = torch.from_numpy(g["ctx_src"]), torch.from_numpy(g["ctx_tgt"])
    with torch.no_grad():
        o = oloop.DiffusionInversionOracle(toy_unet, S=S)
        inv = o.invert(z0, ctx_s)
        np.testing.assert_allclose(torch.cat(inv["latents"]).numpy(), g["inv_latents"], rtol=1e-4, atol=2e-5)
        z = o.sample(inv, [ctx_s, ctx_t])
        zt = o.sample(inv, [ctx_t])
    np.testing.assert_allclose(z[:1].numpy(), g["pair/latent_inv"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(z[1:].numpy(), g["pair/latent"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(zt.numpy(), g["target_only/latent"], rtol=1e-3, atol=2e-4)


DPM_CASES = [(10, "leading"), (10, "linspace"), (50, "leading")]


def dpm_golden_run(g, S, spacing, mode, make_step, dtype=torch.float64):
    """Regenerates the seeded x0 / eps of tests/golden/make_golden.py gen_dpm_inverse and runs `make_step(timesteps)` -> step(eps, t, x) over all S
    steps (free-running); returns the worst relative max-abs error over the stored steps."""
    key = f"S{S}_{spacing}_{mode}"
    ts = g[f"{key}_timesteps"]
    gen = torch.Generator().manual_seed(1000 * S + len(mode))
    x = torch.randn(1, 4, 8, 8, generator=gen, dtype=torch.float64)
    np.testing.assert_array_equal(x.flatten()[:4].numpy(), g[f"{key}_x0_probe"])
    step = make_step(ts)
    xs = []
    x = x.to(dtype)
    for t in ts:
        eps = torch.randn(1, 4, 8, 8, generator=gen, dtype=torch.float64)
        x = step(eps.to(dtype), int(t), x)
        xs.append(x)
    np.testing.assert_array_equal(eps.flatten()[:4].numpy(), g[f"{key}_eps_last_probe"])
    ref = g[f"{key}_x"]
    return max(float(np.abs(xs[j].detach().cpu().double().numpy() - ref[k]).max() / np.abs(ref[k]).max()) for k, j in enumerate(g[f"{key}_steps"]))


@pytest.mark.parametrize("S,spacing", DPM_CASES)
@pytest.mark.parametrize("mode", ["samesame", "sameshift", "shiftshift"])
def test_dpm_inverse_vs_reference_class(golden, S, spacing, mode):
    """oracle.schedule.DpmInverseStepper vs the reference's own DPMSolverMultistepInverseScheduler (run over a restated [3P] solver whose tables are
    fp32 tensors, hence 1e-5 and not 1e-12): timestep grids equal, every stored step of the free-running recursion equal."""
    g = golden("dpm_inverse")
    ac = sch.alphas_cumprod()

    def make(ts):
        st = sch.DpmInverseStepper(ac, S, mode, spacing)
        assert st.grid == ts.tolist()
        return st.step

    assert dpm_golden_run(g, S, spacing, mode, make) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_lowprec_emulation_matches_cpu_half(dtype):
    """oracle/lowprec.py (the reference's 16-bit execution emulated on the fp32 graph) vs torch's OWN 16-bit CPU execution of the same toy-width
    UNet (`.to(dtype)`): both lose the same amount against fp32 (within 25 %) -- the emulation neither flatters nor inflates the floor."""
    from oracle.unet import build_unet
    from oracle.lowprec import LowPrecisionUNet
    ch = (32, 64, 128, 128)
    g = torch.Generator().manual_seed(1)
    x, c = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 77, 768, generator=g)
    rel = lambda p, q: ((p - q).norm() / q.norm()).item()
    with torch.no_grad():
        ref = build_unet(0, block_out_channels=ch, groups=8)(x, torch.tensor(481), encoder_hidden_states=c)["sample"]
        emu = LowPrecisionUNet(build_unet(0, block_out_channels=ch, groups=8), dtype)(x, torch.tensor(481), encoder_hidden_states=c)["sample"]
        real = build_unet(0, block_out_channels=ch, groups=8).to(dtype)(x.to(dtype), torch.tensor(481), encoder_hidden_states=c.to(dtype))["sample"].float()
    e_emu, e_real = rel(emu, ref), rel(real, ref)
    assert 0.75 < e_emu / e_real < 1.25, (e_emu, e_real)
    assert (5e-4 < e_emu < 5e-3) if dtype == torch.float16 else (4e-3 < e_emu < 4e-2)
