"""CPU-only checks of the product's host side: prompt tables vs the reference goldens, schedule tables, the C-ABI
library (loads, exports every symbol include/etainv.h declares), the plugin registries, loud failure without a GPU,
and the world_size-2 shard / gather path on gloo."""
import ctypes
import json
import os
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
PAIRS = json.load(open(ROOT / "tests" / "golden" / "prompt_pairs.json"))


def test_prompt_tables_match_reference_goldens(golden):
    from modules.utils import seq_aligner, ptp_utils, ptp
    from modules.utils.tokenizer import WordLevelTokenizer
    g = golden("ptp_tables")
    tok = WordLevelTokenizer()

    class M:
        tokenizer = tok
    for i, (a, b) in enumerate(PAIRS):
        np.testing.assert_array_equal(np.array(tok.encode(b)), g[f"p{i}/ids_b"])
        mapper, alphas = seq_aligner.get_refinement_mapper([a, b], tok)
        np.testing.assert_array_equal(mapper.numpy(), g[f"p{i}/mapper"])
        np.testing.assert_array_equal(alphas.numpy(), g[f"p{i}/alphas"])
        for S in (10, 50):
            np.testing.assert_array_equal(ptp_utils.get_time_words_attention_alpha([a, b], S, {"default_": 0.4}, tok).numpy(), g[f"p{i}/ctw_{S}"])
        w = b.split(" ")[1]
        np.testing.assert_array_equal(
            ptp_utils.get_time_words_attention_alpha([a, b], 50, {"default_": 0.8, w: (0.1, 0.5)}, tok).numpy(), g[f"p{i}/ctw_word"])
        np.testing.assert_array_equal(ptp.get_equalizer(M, b, (w,), (2,)).numpy(), g[f"p{i}/eq"])
        np.testing.assert_array_equal(ptp_utils.get_word_inds(b, w, tok), g[f"p{i}/inds_w1"])
        if f"p{i}/replace" in g.files:
            np.testing.assert_array_equal(seq_aligner.get_replacement_mapper([a, b], tok).numpy(), g[f"p{i}/replace"])
    with pytest.raises(ValueError):
        seq_aligner.get_replacement_mapper(["a cat", "a very big cat"], tok)


def test_update_alpha_time_word_takes_the_numpy_word_inds_the_reference_passes():
    # reference ptp_utils.py:326-336 is called with get_word_inds' numpy array (and with a scalar tensor index); both must keep working
    from modules.utils import ptp_utils
    ref = torch.zeros(11, 1, 77)
    ref[1:5, 0, [2, 3]] = 1
    for inds in (np.array([2, 3]), torch.tensor([2, 3]), [2, 3]):
        np.testing.assert_array_equal(ptp_utils.update_alpha_time_word(torch.zeros(11, 1, 77), (0.1, 0.5), 0, inds).numpy(), ref.numpy())
    full = ptp_utils.update_alpha_time_word(torch.zeros(11, 1, 77), 0.4, 0)
    assert full[:4].eq(1).all() and full[4:].eq(0).all()


def test_schedule_tables_match_reference_goldens(golden):
    from etainv.pipeline import alphas_cumprod, eta_table, EtaLoop
    from modules.schedulers import DDIMScheduler
    from modules.inverse_schedulers import DDIMInverseScheduler
    g = golden("schedule")
    np.testing.assert_array_equal(alphas_cumprod().astype(np.float32), g["alphas_cumprod"])
    for key, eta in {"lin": (0.0, 0.4), "paper": [[0.6, 0], [1, 0.7]], "paper2": [[0.3, 0], [1, 0.2]],
                     "pow3": [[0.2, 0.1], [0.9, 0.8], 3], "const": 0.25}.items():
        np.testing.assert_allclose(eta_table(eta), g[f"etas_{key}"], rtol=1e-12, atol=1e-15)
    for S in (10, 50, 100):
        s = DDIMScheduler()
        s.set_timesteps(S)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"t_bwd_{S}"])
        inv = DDIMInverseScheduler.from_scheduler(s)
        inv.set_timesteps(S)
        np.testing.assert_array_equal(inv.timesteps.numpy(), g[f"t_fwd_{S}"])
        var = np.array([s._get_variance(int(t), int(t) - 1000 // S) for t in s.timesteps])
        np.testing.assert_allclose(var, g[f"var_{S}"], rtol=5e-5)

        class E:
            L = 64
            lib = None
        lp = EtaLoop(E(), S=S)
        np.testing.assert_array_equal(lp.t_bwd, g[f"t_bwd_{S}"])
        np.testing.assert_array_equal(lp.t_fwd, g[f"t_fwd_{S}"])
    assert float(DDIMScheduler().final_alpha_cumprod) == float(g["final_alpha_cumprod"])


def test_capi_exports_every_declared_symbol():
    from etainv import _capi
    header = (ROOT / "include" / "etainv.h").read_text()
    declared = set(re.findall(r"\b(etainv_[a-z0-9_]+)\s*\(", header))
    declared -= {"etainv_engine", "etainv_engine_config", "etainv_attn_ctrl"}
    assert len(declared) >= 29
    lib = _capi.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/etainv.h but not exported by libetainv_hip.so"
        assert name in _capi.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_capi.SIGNATURES) <= declared
    assert lib.etainv_abi_version() == 1
    # struct layouts agree with the header (sizes are part of the ABI)
    assert ctypes.sizeof(_capi.EngineConfig) == 8 * 4
    assert ctypes.sizeof(_capi.AttnCtrl) == 3 * 4 + 4 + 5 * 8 + 4 * 4 + 4 * 4


def test_argument_errors_are_reported_not_crashed():
    from etainv import _capi
    lib = _capi.load()
    assert lib.etainv_cfg_combine(None, None, 1.0, None, 4, _capi.F32, None) != 0
    assert b"null" in lib.etainv_last_error()
    with pytest.raises(_capi.EtainvError):
        _capi.check(lib.etainv_ddim_step(None, None, 0.5, 0.5, None, 4, _capi.F32, None))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    from etainv import _capi
    from etainv.engine import Engine
    with pytest.raises(_capi.EtainvError):
        Engine()


def test_registries_and_not_built_methods():
    import modules
    assert {"etainv", "diffinv"} <= set(modules.get_inversion_methods())
    assert {"simple", "ptp", "masactrl"} <= set(modules.get_edit_methods())
    with pytest.raises(NotImplementedError):
        modules.load_inverter("nti", model=None)
    with pytest.raises(NotImplementedError):
        modules.load_editor("pnp", inverter=None)
    modules.register_editor("custom", modules.SimpleEditor)
    assert "custom" in modules.get_edit_methods()
    assert modules.DiffusionInversion.get_available_schedulers() == ["ddim", "ddpm", "dpm"]


def test_shard_indices_cover_all_images():
    from etainv.shard import shard_indices
    for n, w in ((700, 8), (5, 2), (3, 4), (0, 2)):
        seen = sorted(i for r in range(w) for i in shard_indices(n, r, w))
        assert seen == list(range(n))
    assert len(shard_indices(700, 0, 8)) == 88 and len(shard_indices(700, 7, 8)) == 87


def _gloo_worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from etainv.shard import shard_indices, gather_latents
    idx = shard_indices(n_items, rank, world)
    local = torch.stack([torch.full((4, 8, 8), float(i)) for i in idx]) if idx else torch.zeros(0, 4, 8, 8)
    out = gather_latents(local, n_items, rank, world)
    q.put((rank, out[:, 0, 0, 0].tolist()))
    dist.destroy_process_group()


def test_gather_latents_world_size_2_gloo():
    import sys
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    for p in (str(ROOT), str(ROOT / "eta-inversion_amd")):
        os.environ["PYTHONPATH"] = p + os.pathsep + os.environ.get("PYTHONPATH", "")
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for _, vals in res:
        assert vals == [0.0, 1.0, 2.0, 3.0, 4.0]


def _toy_bpe(tmp_path):
    """a small byte-level BPE vocabulary trained on a toy corpus, written in the CLIP tokenizer's file formats"""
    import collections
    from modules.utils.tokenizer import _bytes_to_unicode
    b2u = _bytes_to_unicode()
    corpus = ("a photo of a cat sitting on a wooden chair . a round cake with orange frosting , the house near the frozen lake ! "
              "it's a dog's painting of 2 towers and 35 bridges café naïve").split()
    vocab = list(b2u.values()) + [v + "</w>" for v in b2u.values()]
    sym = lambda w: tuple([b2u[b] for b in w.encode()][:-1] + [b2u[w.encode()[-1]] + "</w>"])
    seqs, merges = collections.Counter(sym(w) for w in corpus), []
    for _ in range(80):
        pairs = collections.Counter()
        for s, c in seqs.items():
            for a, b in zip(s, s[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _c = max(sorted(pairs.items()), key=lambda kv: kv[1])
        merges.append((a, b))
        vocab.append(a + b)
        new = collections.Counter()
        for s, c in seqs.items():
            out, i = [], 0
            while i < len(s):
                if i + 1 < len(s) and s[i] == a and s[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(s[i])
                    i += 1
            new[tuple(out)] += c
        seqs = new
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    (tmp_path / "vocab.json").write_text(json.dumps({t: i for i, t in enumerate(vocab)}))
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    return str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt")


def test_clip_bpe_tokenizer_matches_transformers(tmp_path):
    """the native byte-level BPE (modules/utils/tokenizer.py) against the installed third-party implementation on the same files"""
    transformers = pytest.importorskip("transformers")
    from modules.utils.tokenizer import ClipBPETokenizer
    vocab, merges = _toy_bpe(tmp_path)
    ref, nat = transformers.CLIPTokenizer(vocab, merges), ClipBPETokenizer(vocab, merges)
    prompts = ["a cat sitting on a wooden chair", "A  photo of 35 Dogs!", "it's the frozen-lake's painting", "", "café naïve towers,bridges...",
               "a round cake with orange frosting", " leading and trailing  ", "x" * 200, "we'll they've i'm can't"]
    for p in prompts:
        assert nat.encode(p) == ref.encode(p), p
    a = nat(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    b = ref(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert a.shape == (len(prompts), 77) and torch.equal(a, b)
    assert (nat.bos_token_id, nat.eos_token_id) == (ref.bos_token_id, ref.eos_token_id)
    ids = nat.encode("a wooden chair")
    assert [nat.decode([i]) for i in ids[1:-1]] == [ref.decode([i]) for i in ids[1:-1]]      # ptp_utils.get_word_inds decodes single ids


def test_cv2_linear_resize_restatement():
    """modules/models/resize.py vs values worked out by hand from OpenCV's fixed-point INTER_LINEAR (11-bit coefficients, taps at pixel
    centres, edge clamping, no antialiasing) -- the reference preprocesses with cv2.resize (modules/models/__init__.py:64)."""
    from modules.models.resize import resize_linear_u8
    # 1 x 2 -> 1 x 4: f = (d + .5) / 2 - .5 = -.25, .25, .75, 1.25 -> clamp, (1536, 512), (512, 1536), clamp
    #   H = 255 * 512 = 130560; (2048 * (130560 >> 4)) >> 16 = 255; (255 + 2) >> 2 = 64;   255 * 1536 -> 765 -> 191
    out = resize_linear_u8(np.array([[0, 255]], np.uint8), (4, 1))
    assert out.tolist() == [[0, 64, 191, 255]]
    # down-scale 1 x 6 -> 1 x 2 (factor 3): f = 1.0 and 4.0 exactly -> single taps on pixels 1 and 4: NO averaging of the neighbours
    out = resize_linear_u8(np.array([[10, 200, 30, 40, 90, 60]], np.uint8), (2, 1))
    assert out.tolist() == [[200, 90]]
    # 4 x 4 -> 2 x 2 is the exact-2x case: box mean with rounding
    img = np.arange(16, dtype=np.uint8).reshape(4, 4) * 10
    assert resize_linear_u8(img, (2, 2)).tolist() == [[(0 + 10 + 40 + 50 + 2) >> 2, (20 + 30 + 60 + 70 + 2) >> 2],
                                                      [(80 + 90 + 120 + 130 + 2) >> 2, (100 + 110 + 140 + 150 + 2) >> 2]]
    # 2 x 2 -> 3 x 3, vertical and horizontal coefficients together: f = (d + .5) * 2/3 - .5 = -1/6, 1/2, 7/6 -> (0, 0), (0, .5), (1, 0)
    #   centre: H rows = (0*1024 + 100*1024, 200*1024 + 40*1024) = (102400, 245760); >> 4 = (6400, 15360); * 1024 >> 16 = (100, 240); (340 + 2) >> 2 = 85
    out = resize_linear_u8(np.array([[0, 100], [200, 40]], np.uint8), (3, 3))
    assert out.tolist() == [[0, 50, 100], [100, 85, 70], [200, 120, 40]]
    # multi-channel images keep their channel order; identity size is a copy
    rgb = np.random.default_rng(0).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    up = resize_linear_u8(rgb, (14, 10))
    assert up.shape == (10, 14, 3) and all(np.array_equal(up[..., c], resize_linear_u8(rgb[..., c], (14, 10))) for c in range(3))
    assert np.array_equal(resize_linear_u8(rgb, (7, 5)), rgb)
    # float bilinear (align_corners=False, no antialias) agrees within one grey level
    ref = torch.nn.functional.interpolate(torch.from_numpy(rgb).permute(2, 0, 1)[None].float(), size=(10, 14), mode="bilinear", align_corners=False)
    assert np.abs(up.astype(np.int32) - ref[0].permute(1, 2, 0).round().numpy().astype(np.int32)).max() <= 1


def test_dpm_solver_oracle_properties():
    """oracle.schedule DPM-Solver++ restatement ([3P] diffusers is absent: these are the checks that stand in for a pin): the first-order
    update IS the deterministic DDIM step; with a data prediction that is linear in lambda the 2M update is exact (it integrates
    alpha_t * int e^{-lambda} x0(lambda) dlambda with x0 linear), i.e. it beats first order by orders of magnitude; timestep grids."""
    from oracle import schedule as sch
    ac = sch.alphas_cumprod()
    tabs = sch.dpm_tables(ac)
    alpha, sigma, lam = tabs
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    for s, t in ((500, 480), (20, 0), (300, 320), (980, 999)):
        a = sch.dpm_first_order(x, sch.dpm_x0(x, eps, tabs, s), tabs, s, t)
        assert torch.allclose(a, sch.ddim_step(x, eps, float(ac[s]), float(ac[t])), rtol=0, atol=1e-12)
    # exact solution of dx/dlambda for x0(lambda) = A + B lambda:  x_t = (sigma_t/sigma_s) x_s + alpha_t [ (A + B lambda_t) - e^{-h} (A + B lambda_s) ] - alpha_t B (1 - e^{-h})
    A, B = torch.randn(4, generator=g, dtype=torch.float64), 0.3 * torch.randn(4, generator=g, dtype=torch.float64)
    x0 = lambda t: A + B * lam[t]
    s1, s0, t = 700, 680, 660
    xs = torch.randn(4, generator=g, dtype=torch.float64)
    h = lam[t] - lam[s0]
    exact = (sigma[t] / sigma[s0]) * xs + alpha[t] * ((x0(t) - B) - np.exp(-h) * (x0(s0) - B))
    e2 = (sch.dpm_second_order(xs, x0(s0), x0(s1), tabs, s1, s0, t) - exact).abs().max()
    e1 = (sch.dpm_first_order(xs, x0(s0), tabs, s0, t) - exact).abs().max()
    assert e2 < 0.05 * e1 and e2 < 1e-3
    assert sch.dpm_timesteps_backward(50).tolist()[:3] == [950, 931, 912] and sch.dpm_timesteps_backward(50)[-1] == 19
    assert sch.dpm_timesteps_forward(50).tolist()[:3] == [0, 19, 38] and len(sch.dpm_timesteps_forward(50)) == 50
    assert sch.dpm_timesteps_backward(10, "linspace").tolist() == [999, 899, 799, 699, 599, 500, 400, 300, 200, 100]


def test_backward_loop_row_economy_host_logic(monkeypatch):
    """Which UNet rows EtaLoop.sample issues per backward step (round 3; no GPU: the engine and the C ABI are replaced by recorders).  With the paper's eta
    schedule and the PIE prompt-to-prompt settings at S = 50: steps 0..19 (eta > 0, cross replacement live) the reference's four row groups; steps 20..29
    (eta == 0, no cross replacement, self-replace live) rows [u_t, c_t, c_s] with the cond source rows leaving after block 12; steps 30..49 after block 9.
    Without attention coupling the eta == 0 steps run [u_t, c_t]; MasaCtrl keeps all rows; skip_dead_source_rows=False restores 4 B rows everywhere."""
    from etainv import _capi, pipeline
    from etainv.pipeline import EtaLoop, PtpTables

    class FakeLib:
        def __getattr__(self, name):
            return lambda *a, **k: 0

    class FakeEngine:
        L, lib, map_div = 8, FakeLib(), 4

        def __init__(self):
            self.calls = []

        def maps_configure(self, div): self.map_div = div
        def word_maps_ex(self, *a, **k): pass

        def unet(self, latent, t, ctx, ctrl=None, out=None):
            c = ctrl.c if ctrl is not None else None
            self.calls.append(dict(rows=ctx.shape[0], n_lat=latent.shape[0], t=int(t), first_row=c.first_row if c else 0, exit=c.src_exit_block if c else 0,
                                   edit=bool(c.mapper) if c else False, self_on=bool(c.self_replace_active) if c else False))
            out.zero_()
            return out

        def maps_reset(self): pass
        def word_maps(self, *a, **k): pass
        def local_blend(self, x, *a, **k): return x

        import contextlib

        @contextlib.contextmanager
        def cached_context(self):
            yield
    monkeypatch.setattr(_capi, "ptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(_capi, "stream_ptr", lambda: None)
    monkeypatch.setattr(_capi, "check", lambda rc: None)
    monkeypatch.setattr(pipeline, "AttnControl", lambda **kw: type("C", (), {"c": _capi.AttnCtrl(
        mode=kw.get("mode", 0), n_img=kw.get("n_img", 1), mapper=1 if kw.get("mapper") is not None else None,
        self_replace_active=int(kw.get("self_replace_active", False)), masa_active=int(kw.get("masa_active", False)))})())
    S, B, L = 50, 2, 8
    ca = np.zeros((S + 1, B, 77), np.float32)
    ca[: int(0.4 * (S + 1))] = 1.0
    ptp = PtpTables(np.tile(np.arange(77, dtype=np.int32), (B, 1)), np.ones((B, 77), np.float32), ca, 0.6, S, equalizer=np.ones((B, 77), np.float32),
                    blend_alpha=np.zeros((B, 2, 77), np.float32), device="cpu")
    assert ptp.cross_active[:20].all() and not ptp.cross_active[20:].any() and (ptp.self_lo, ptp.self_hi) == (0, 30)
    inv = {"latents": torch.zeros(S + 1, B, 4, L, L), "maps_mean": torch.zeros(B, 3, L, L), "maps_steps": None}
    ctx = torch.zeros(B, 2, 77, 768)
    noise = torch.zeros(S, 10, 4, L, L)

    def run(**kw):
        eng = FakeEngine()
        loop = EtaLoop(eng, S=S, eta=[[0.6, 0], [1, 0.7]], **kw.pop("loop", {}))
        loop.sample(inv, ctx, ctx, noise, edit_word=torch.ones(B, dtype=torch.int64), **kw)
        return eng.calls, loop
    calls, loop = run(ptp=ptp)
    assert [c["rows"] for c in calls] == [4 * B] * 20 + [3 * B] * 30
    assert all(c["edit"] and c["first_row"] == 0 and c["exit"] == 0 and c["n_lat"] == 2 * B for c in calls[:20])
    assert all(not c["edit"] and c["first_row"] == B and c["exit"] == 12 and c["self_on"] and c["n_lat"] == 3 * B for c in calls[20:30])
    assert all(not c["edit"] and c["first_row"] == B and c["exit"] == 9 and not c["self_on"] for c in calls[30:])
    want = B * (20 * 4 + 10 * (2 + loop.SRC_EXIT_SHARE_12) + 20 * (2 + loop.SRC_EXIT_SHARE))
    assert abs(loop.rows_executed - want) < 1e-9
    from etainv.flops import exit_share, unet_macs                                              # the shares follow the latent size (layer walk)
    assert abs(unet_macs(64) / 1e9 - 401.64) < 0.01 and abs(unet_macs(96) / 1e9 - 1074.06) < 0.01
    assert abs(50 + 20 * 4 + 10 * (2 + exit_share(64, 12)) + 20 * (2 + exit_share(64, 9)) - 207.34) < 0.01   # per image at L = 64, + the S cond rows of the forward pass
    assert exit_share(96, 9) < exit_share(64, 9) and (loop.SRC_EXIT_SHARE, loop.SRC_EXIT_SHARE_12) == (exit_share(L, 9), exit_share(L, 12))
    calls, _ = run()                                                                            # no attention coupling (simple editor)
    assert [c["rows"] for c in calls] == [4 * B] * 20 + [2 * B] * 30 and all(c["n_lat"] == B for c in calls[20:])
    calls, _ = run(masactrl=(4, 10))                                                            # MasaCtrl couples u_t to u_s
    assert [c["rows"] for c in calls] == [4 * B] * 50
    calls, loop = run(ptp=ptp, loop=dict(skip_dead_source_rows=False))                          # the reference's row count
    assert [c["rows"] for c in calls] == [4 * B] * 50 and loop.rows_executed == 4 * B * 50
    assert [c["edit"] for c in calls] == [True] * 20 + [False] * 30                             # (the identity cross edit is still skipped)
