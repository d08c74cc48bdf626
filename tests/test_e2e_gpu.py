"""End-to-end etainv + {simple, ptp, masactrl} on MI355X (native loops, B = 2 image pairs per batch) vs the CPU
oracle (fp32, one pair at a time) with the same SD1.x-width synthetic UNet, contexts, noise table and edit tables.
Free-running S-step trajectories of a random-weight UNet amplify rounding, so tolerances are on relative L2."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
S, B = 4, 2


from tests.oracle_cache import oracle_leg, oracle_unet  # noqa: E402


@pytest.fixture(scope="module")
def setup():
    from etainv.engine import Engine
    engines = {}

    def get(L, dtype=torch.float16):
        if (L, dtype) not in engines:
            e = Engine(dtype=dtype, max_unet_batch=4 * B, latent_size=L, max_img=B)
            e.load_synthetic(0)
            engines[(L, dtype)] = e
        return engines[(L, dtype)]
    yield None, get
    for e in engines.values():
        e.close()


def _inputs(L):
    g = torch.Generator().manual_seed(77)
    pairs = json.load(open(__file__.rsplit("/", 1)[0] + "/golden/prompt_pairs.json"))
    pairs = [pairs[0], pairs[3]]
    z0 = 0.8 * torch.randn(B, 4, L, L, generator=g)
    ctx_src = torch.randn(B, 2, 77, 768, generator=g)
    ctx_tgt = torch.randn(B, 2, 77, 768, generator=g)
    ctx_tgt[:, 0] = ctx_src[:, 0]            # same "" uncond embedding for both prompts
    return pairs, z0, ctx_src, ctx_tgt


def relerr(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


PTP_CFG = dict(is_replace_controller=False, cross_replace_steps={"default_": .4}, self_replace_steps=.6)


CASES = [  # editor, L, dtype, use_mask, tolerance on the edited latent (rel L2): 2x the value measured on MI355X in round 3 (8.5e-3 ... 9.2e-3 fp16, 7.0e-2 bf16)
    ("simple", 16, torch.float16, True, 1.9e-2), ("ptp", 16, torch.float16, True, 1.8e-2), ("masactrl", 16, torch.float16, True, 1.9e-2),
    ("ptp_replace", 16, torch.float16, True, 1.8e-2), ("simple", 16, torch.float16, False, 1.9e-2),
    ("ptp", 16, torch.bfloat16, True, 1.4e-1),
    ("masactrl", 24, torch.float16, True, 1.7e-2),          # 768^2-style non-power-of-two token counts (N = 576 / 144 / 36 / 9)
]


def _edit_setup(editor, L):
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    replace = editor == "ptp_replace"
    if replace:
        editor = "ptp"
        pairs = [pairs[0], pairs[0]]     # AttentionReplace needs prompts of equal length (seq_aligner.py:161-163)
    eta = [[0.6, 0], [1, 0.7]] if editor == "ptp" else (0.0, 0.4)
    return editor, replace, pairs, z0, ctx_src, ctx_tgt, eta, [1, 1]


@oracle_leg(cases=sorted({(c[0], c[1], c[3]) for c in CASES}))
def leg_edit(editor, L, use_mask):
    """the fp32 oracle, one pair at a time"""
    from oracle import loop as oloop, ptp as optp
    editor, replace, pairs, z0, ctx_src, ctx_tgt, eta, edit_word = _edit_setup(editor, L)
    tok = optp.WordTokenizer()
    noise = oloop.noise_table(S, 10, L, seed=0)
    unet = oracle_unet()
    ref_inv, ref_out, ref_maps = [], [], []
    with torch.no_grad():
        for i, (src, tgt) in enumerate(pairs):
            o = oloop.EtaInversionOracle(unet, S=S, eta=eta, L=L, use_mask=use_mask)
            inv = o.invert(z0[i:i + 1], ctx_src[i], src)
            controller = masa = None
            if editor == "ptp":
                bw, tw = src.split(" ")[edit_word[i]], tgt.split(" ")[edit_word[i]]
                controller = optp.make_edit_controller(src, tgt, S, tok, blend_words=((bw,), (tw,)),
                                                       equilizer_params=None if replace else {"words": (tw,), "values": (2,)},
                                                       res=L // 4, thres_n=(L // 2) ** 2, **{**PTP_CFG, "is_replace_controller": replace})
            elif editor == "masactrl":
                masa = oloop.MasaCtrl(1, 10)
            z = o.sample(inv, ctx_src[i], ctx_tgt[i], noise, edit_word_idx=(edit_word[i], edit_word[i]), controller=controller, masactrl=masa)
            ref_inv.append(torch.cat(inv["latents"]))
            if use_mask:
                ref_maps.append(torch.stack(inv["attn_maps_mean"])[edit_word[i]])
            ref_out.append(z)
    ref_inv = torch.stack(ref_inv, 1)                     # (S+1, B, 4, L, L)
    ref_out = torch.cat([torch.stack([r[0] for r in ref_out]), torch.stack([r[1] for r in ref_out])])   # [src.., tgt..]
    return {"inv": ref_inv, "out": ref_out, "maps": torch.stack(ref_maps) if use_mask else None}


@pytest.mark.parametrize("editor,L,dtype,use_mask,tol", CASES)
def test_edit_vs_oracle(setup, editor, L, dtype, use_mask, tol):
    from oracle import loop as oloop, ptp as optp
    from etainv.pipeline import EtaLoop, PtpTables, noise_table
    _, get_engine = setup
    eng = get_engine(L, dtype)
    R = leg_edit(editor, L, use_mask)
    ref_inv, ref_out, ref_maps = R["inv"], R["out"], R["maps"]
    editor, replace, pairs, z0, ctx_src, ctx_tgt, eta, edit_word = _edit_setup(editor, L)
    tok = optp.WordTokenizer()
    noise = oloop.noise_table(S, 10, L, seed=0)

    # ---------------- native
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    for i, (src, _) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
    loop = EtaLoop(eng, S=S, eta=eta, use_mask=use_mask)
    inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
    ptp = masa = None
    if editor == "ptp":
        mp, al, eq, ba, ca, rm = [], [], [], [], [], []
        for i, (src, tgt) in enumerate(pairs):
            bw, tw = src.split(" ")[edit_word[i]], tgt.split(" ")[edit_word[i]]
            m, a = optp.refinement_mapper(src, tgt, tok)
            mp.append(m)
            al.append(a)
            eq.append(optp.equalizer(tgt, (tw,), (2,), tok))
            ba.append(optp.blend_alpha_layers([src, tgt], ((bw,), (tw,)), tok))
            ca.append(optp.time_words_alpha([src, tgt], S, {"default_": .4}, tok)[:, 0])
            if replace:
                rm.append(optp.replacement_mapper(src, tgt, tok))
        if replace:
            ptp = PtpTables(None, None, np.stack(ca, 1), 0.6, S, blend_alpha=np.stack(ba), replace_mat=np.stack(rm))
        else:
            ptp = PtpTables(np.stack(mp), np.stack(al), np.stack(ca, 1), 0.6, S, equalizer=np.stack(eq), blend_alpha=np.stack(ba))
    elif editor == "masactrl":
        masa = (1, 10)
    out = loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), noise.reshape(S, 10, 4, L, L).cuda(), edit_word=torch.tensor(edit_word),
                      ptp=ptp, masactrl=masa)
    torch.cuda.synchronize()

    e_inv = relerr(inv["latents"].cpu(), ref_inv)
    e_map = relerr(torch.stack([inv["maps_mean"][i, edit_word[i]] for i in range(B)]).cpu(), ref_maps[:, 0]) if use_mask else 0.0
    e_src = relerr(out[:B].cpu(), ref_out[:B])
    e_tgt = relerr(out[B:].cpu(), ref_out[B:])
    print(f"{editor} L={L} {dtype} mask={use_mask} replace={replace}: inversion traj {e_inv:.2e}, word map {e_map:.2e}, latent_inv {e_src:.2e}, latent {e_tgt:.2e}")
    bf = dtype == torch.bfloat16
    # measured (round 3): inversion trajectory 6.4e-4 / 5.0e-3, word map 2.3e-4 / 3.2e-3 (fp16 / bf16); bounds at 2x
    assert e_inv < (1e-2 if bf else 1.3e-3) and e_map < (6.4e-3 if bf else 5e-4)
    assert e_src < 1e-6                            # source row replays the stored inversion trajectory exactly (eta_inversion.py:247-249)
    assert e_tgt < tol


MASK_MODES = [dict(mask_eta="fwd", thres=0.2), dict(mask_eta="fwd_mean", thres=None, pow=2.0), dict(mask_eta="gt", thres=0.5)]


@oracle_leg(cases=[(m,) for m in MASK_MODES])
def leg_mask_modes(mode):
    from oracle import loop as oloop
    L = 16
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    noise = oloop.noise_table(S, 10, L, seed=0)
    gt = torch.rand(B, L, L, generator=torch.Generator().manual_seed(9))
    unet = oracle_unet()
    ref = []
    with torch.no_grad():
        for i, (src, tgt) in enumerate(pairs):
            o = oloop.EtaInversionOracle(unet, S=S, eta=(0.0, 0.4), L=L, use_mask=True, thres=mode.get("thres", 0.2), mask_eta=mode["mask_eta"],
                                         mask_pow=mode.get("pow"))
            inv = o.invert(z0[i:i + 1], ctx_src[i], src)
            ref.append(o.sample(inv, ctx_src[i], ctx_tgt[i], noise, edit_word_idx=(1, 1), gt_mask=gt[i:i + 1]))
    return {"ref": torch.cat([torch.stack([r[0] for r in ref]), torch.stack([r[1] for r in ref])])}


@pytest.mark.parametrize("mode", MASK_MODES)
def test_mask_modes_vs_oracle(setup, mode):
    """non-default eta-mask sources / shapes end to end (etainv + simple): per-timestep forward maps, soft mask with pow, given mask"""
    from oracle import loop as oloop
    from etainv.pipeline import EtaLoop
    unet, get_engine = setup
    L = 16
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    noise = oloop.noise_table(S, 10, L, seed=0)
    gt = torch.rand(B, L, L, generator=torch.Generator().manual_seed(9))
    ref = leg_mask_modes(mode)["ref"]
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    for i, (src, _) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
    loop = EtaLoop(eng, S=S, eta=(0.0, 0.4), use_mask=True, mask_thres=mode.get("thres", 0.2), mask_eta=mode["mask_eta"], mask_pow=mode.get("pow"))
    inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
    out = loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), noise.reshape(S, 10, 4, L, L).cuda(), edit_word=torch.tensor([1, 1]), gt_mask=gt)
    print(f"source row {relerr(out[:B].cpu(), ref[:B]):.2e}, edited latent {relerr(out[B:].cpu(), ref[B:]):.2e}")
    assert relerr(out[:B].cpu(), ref[:B]) < 1e-6 and relerr(out[B:].cpu(), ref[B:]) < 1.9e-2     # measured 8.5e-3 ... 9.6e-3 (round 3); bound at 2x


BWD_MODES = ["bwd_source", "bwd_target", "bwd_source_target"]


@oracle_leg(cases=[(m,) for m in BWD_MODES])
def leg_bwd_mask(mode):
    from oracle import loop as oloop, ptp as optp
    L, eta = 16, (0.3, 0.6)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tok = optp.WordTokenizer()
    noise = oloop.noise_table(S, 10, L, seed=0)
    unet = oracle_unet()
    ref = []
    with torch.no_grad():
        for i, (src, tgt) in enumerate(pairs):
            o = oloop.EtaInversionOracle(unet, S=S, eta=eta, L=L, use_mask=True, thres=0.15, mask_eta=mode)
            inv = o.invert(z0[i:i + 1], ctx_src[i], src)
            bw, tw = src.split(" ")[1], tgt.split(" ")[1]
            controller = optp.make_edit_controller(src, tgt, S, tok, blend_words=((bw,), (tw,)), equilizer_params={"words": (tw,), "values": (2,)},
                                                   res=L // 4, thres_n=(L // 2) ** 2, **PTP_CFG)
            ref.append(o.sample(inv, ctx_src[i], ctx_tgt[i], noise, edit_word_idx=(1, 1), controller=controller))
    return {"ref": torch.cat([torch.stack([r[0] for r in ref]), torch.stack([r[1] for r in ref])])}


@pytest.mark.parametrize("mode", BWD_MODES)
def test_bwd_mask_sources_vs_oracle(setup, mode):
    """eta mask from the backward-pass prompt-to-prompt store (running average over the steps done), etainv + ptp"""
    from oracle import loop as oloop, ptp as optp
    from etainv.pipeline import EtaLoop, PtpTables
    unet, get_engine = setup
    L = 16
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tok = optp.WordTokenizer()
    noise = oloop.noise_table(S, 10, L, seed=0)
    eta = (0.3, 0.6)
    ref = leg_bwd_mask(mode)["ref"]
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    mp, al, eq, ba, ca = [], [], [], [], []
    for i, (src, tgt) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
        bw, tw = src.split(" ")[1], tgt.split(" ")[1]
        m, a = optp.refinement_mapper(src, tgt, tok)
        mp.append(m); al.append(a)
        eq.append(optp.equalizer(tgt, (tw,), (2,), tok))
        ba.append(optp.blend_alpha_layers([src, tgt], ((bw,), (tw,)), tok))
        ca.append(optp.time_words_alpha([src, tgt], S, {"default_": .4}, tok)[:, 0])
    ptp = PtpTables(np.stack(mp), np.stack(al), np.stack(ca, 1), 0.6, S, equalizer=np.stack(eq), blend_alpha=np.stack(ba))
    loop = EtaLoop(eng, S=S, eta=eta, use_mask=True, mask_thres=0.15, mask_eta=mode)
    inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
    out = loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), noise.reshape(S, 10, 4, L, L).cuda(), edit_word=torch.tensor([1, 1]), ptp=ptp,
                      edit_word_tgt=torch.tensor([1, 1]))
    print(f"source row {relerr(out[:B].cpu(), ref[:B]):.2e}, edited latent {relerr(out[B:].cpu(), ref[B:]):.2e}")
    assert relerr(out[:B].cpu(), ref[:B]) < 1e-6 and relerr(out[B:].cpu(), ref[B:]) < 1.9e-2     # measured 8.5e-3 ... 9.6e-3 (round 3); bound at 2x


# (name, attn_res as a divisor of L, attn_from_where, mask_eta): the (L/2)^2 layers, the up / the down blocks only, the mid block's map (attn_res = L/8
# turns from_where into ["mid"], ptp.py:293-294), a from_where subset read from the backward-pass store
ATTN_LAYER_CASES = [("half", 2, ("up", "down"), "fwd_mean"), ("up", 4, ("up",), "fwd_mean"), ("mid", 8, ("up", "down"), "fwd_mean"),
                    ("down_bwd", 4, ("down",), "bwd_source_target")]


@oracle_leg(cases=[(c[0],) for c in ATTN_LAYER_CASES])
def leg_attn_layers(name):
    from oracle import loop as oloop, ptp as optp
    _, div, where, mask_eta = next(c for c in ATTN_LAYER_CASES if c[0] == name)
    L, eta = 16, (0.3, 0.6)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tok = optp.WordTokenizer()
    noise = oloop.noise_table(S, 10, L, seed=0)
    unet = oracle_unet()
    ref, maps = [], []
    with torch.no_grad():
        for i, (src, tgt) in enumerate(pairs):
            o = oloop.EtaInversionOracle(unet, S=S, eta=eta, L=L, use_mask=True, thres=None, mask_eta=mask_eta, attn_res=L // div, attn_from_where=where)   # soft mask: the map VALUES reach the latents
            inv = o.invert(z0[i:i + 1], ctx_src[i], src)
            maps.append(inv["attn_maps_mean"][1])
            bw, tw = src.split(" ")[1], tgt.split(" ")[1]
            controller = optp.make_edit_controller(src, tgt, S, tok, blend_words=((bw,), (tw,)), equilizer_params={"words": (tw,), "values": (2,)},
                                                   res=L // 4, thres_n=(L // 2) ** 2, **PTP_CFG)
            ref.append(o.sample(inv, ctx_src[i], ctx_tgt[i], noise, edit_word_idx=(1, 1), controller=controller))
    return {"ref": torch.cat([torch.stack([r[0] for r in ref]), torch.stack([r[1] for r in ref])]), "maps": torch.cat(maps)}


@pytest.mark.parametrize("case", ATTN_LAYER_CASES, ids=[c[0] for c in ATTN_LAYER_CASES])
def test_attn_res_and_from_where_vs_oracle(setup, case):
    """non-default `attn_res` / `attn_from_where` of the eta mask (eta_inversion.py:161-162): etainv_maps_configure + the layer mask of
    etainv_maps_word_maps_ex; the oracle side is pinned by the reference itself (tests/golden/e2e_attnres.npz)"""
    from oracle import loop as oloop, ptp as optp
    from etainv.pipeline import EtaLoop, PtpTables
    name, div, where, mask_eta = case
    unet, get_engine = setup
    L = 16
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tok = optp.WordTokenizer()
    noise = oloop.noise_table(S, 10, L, seed=0)
    leg = leg_attn_layers(name)
    ref = leg["ref"]
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    mp, al, eq, ba, ca = [], [], [], [], []
    for i, (src, tgt) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
        bw, tw = src.split(" ")[1], tgt.split(" ")[1]
        m, a = optp.refinement_mapper(src, tgt, tok)
        mp.append(m); al.append(a)
        eq.append(optp.equalizer(tgt, (tw,), (2,), tok))
        ba.append(optp.blend_alpha_layers([src, tgt], ((bw,), (tw,)), tok))
        ca.append(optp.time_words_alpha([src, tgt], S, {"default_": .4}, tok)[:, 0])
    ptp = PtpTables(np.stack(mp), np.stack(al), np.stack(ca, 1), 0.6, S, equalizer=np.stack(eq), blend_alpha=np.stack(ba))
    loop = EtaLoop(eng, S=S, eta=(0.3, 0.6), use_mask=True, mask_thres=None, mask_eta=mask_eta, attn_res=L // div, attn_from_where=where)
    try:
        inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
        assert eng.map_div == div
        m_err = relerr(inv["maps_mean"][:, 1].cpu(), leg["maps"])
        out = loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), noise.reshape(S, 10, 4, L, L).cuda(), edit_word=torch.tensor([1, 1]), ptp=ptp,
                          edit_word_tgt=torch.tensor([1, 1]))
        assert eng.map_div == 4                      # the backward pass went back to the (L/4)^2 store (LocalBlend)
    finally:
        eng.maps_configure(4)                        # (the engine is shared by the tests of this file)
    print(f"{name}: edit-word map {m_err:.2e}, source row {relerr(out[:B].cpu(), ref[:B]):.2e}, edited latent {relerr(out[B:].cpu(), ref[B:]):.2e}")
    assert m_err < 5e-3
    assert relerr(out[:B].cpu(), ref[:B]) < 1e-6 and relerr(out[B:].cpu(), ref[B:]) < 2e-2


def test_attn_layer_selection_errors(setup):
    """what the store cannot serve fails loudly, like the reference's own failure modes"""
    from etainv.pipeline import EtaLoop
    unet, get_engine = setup
    eng = get_engine(16, torch.float16)
    with pytest.raises(NotImplementedError):
        EtaLoop(eng, S=S, attn_res=16)                                                         # L itself: no cross layer has L^2 ... only L/2, L/4, L/8
    with pytest.raises(ValueError):
        EtaLoop(eng, S=S, attn_res=4, attn_from_where=("mid",))                                # the reference fails in torch.cat([])
    with pytest.raises(NotImplementedError):
        EtaLoop(eng, S=S, attn_res=8, mask_eta="bwd_source")                                   # backward store is (L/4)^2
    eng.maps_configure(2)
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    with pytest.raises(Exception, match="LocalBlend"):
        eng.local_blend(x, 1, torch.zeros(1, 2, 77, device="cuda"))
    eng.maps_configure(4)
    tok = torch.ones(1, 1, dtype=torch.int32, device="cuda")
    with pytest.raises(Exception, match="layer_mask"):
        eng.word_maps_ex(1, tok, 1, 0, 0x20, torch.zeros(1, 1, 16, 16, device="cuda"))


DIRINV = [("fwd_mean", "fwd_mean"), ("fwd_mean", "gt"), ("gt", "fwd")]


@oracle_leg(cases=DIRINV)
def leg_target_dirinv(mask_eta, mask_dirinv):
    from oracle import loop as oloop
    L = 16
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    noise = oloop.noise_table(S, 10, L, seed=0)
    gt = torch.rand(B, L, L, generator=torch.Generator().manual_seed(19))
    kw = dict(thres=0.3, mask_eta=mask_eta, target_dirinv=0.6, mask_dirinv=mask_dirinv)
    unet = oracle_unet()
    ref = []
    with torch.no_grad():
        for i, (src, tgt) in enumerate(pairs):
            o = oloop.EtaInversionOracle(unet, S=S, eta=(0.0, 0.4), L=L, use_mask=True, **kw)
            inv = o.invert(z0[i:i + 1], ctx_src[i], src)
            ref.append(o.sample(inv, ctx_src[i], ctx_tgt[i], noise, edit_word_idx=(1, 1), gt_mask=gt[i:i + 1]))
    return {"ref": torch.cat([torch.stack([r[0] for r in ref]), torch.stack([r[1] for r in ref])])}


@pytest.mark.parametrize("mask_eta,mask_dirinv", DIRINV)
def test_target_dirinv_vs_oracle(setup, mask_eta, mask_dirinv):
    """target_dirinv with a mask_dirinv (eta_inversion.py:234-256) end to end, etainv + simple; mask_dirinv may name a different map
    source than mask_eta"""
    from oracle import loop as oloop
    from etainv.pipeline import EtaLoop
    unet, get_engine = setup
    L = 16
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    noise = oloop.noise_table(S, 10, L, seed=0)
    gt = torch.rand(B, L, L, generator=torch.Generator().manual_seed(19))
    ref = leg_target_dirinv(mask_eta, mask_dirinv)["ref"]
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    for i, (src, _) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
    loop = EtaLoop(eng, S=S, eta=(0.0, 0.4), use_mask=True, mask_thres=0.3, mask_eta=mask_eta, target_dirinv=0.6, mask_dirinv=mask_dirinv)
    inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
    out = loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), noise.reshape(S, 10, 4, L, L).cuda(), edit_word=torch.tensor([1, 1]), gt_mask=gt)
    print(f"source row {relerr(out[:B].cpu(), ref[:B]):.2e}, edited latent {relerr(out[B:].cpu(), ref[B:]):.2e}")
    assert relerr(out[:B].cpu(), ref[:B]) < 1e-6 and relerr(out[B:].cpu(), ref[B:]) < 1.9e-2     # measured 8.5e-3 ... 9.6e-3 (round 3); bound at 2x


@oracle_leg()
def leg_forward_guidance_table():
    from oracle import loop as oloop
    L = 16
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    unet = oracle_unet()
    with torch.no_grad():
        return {"ref": torch.stack([torch.cat(oloop.EtaInversionOracle(unet, S=S, L=L, use_mask=False, guidance_scale_fwd=(1.0, 3.0))
                                              .invert(z0[i:i + 1], ctx_src[i], pairs[i][0])["latents"]) for i in range(B)], 1)}


def test_forward_guidance_table_vs_oracle(setup):
    """guidance_scale_fwd = (start, end): per-timestep CFG in the inversion pass (eta_inversion.py:108-110,325-326) -- the uncond half is run"""
    from oracle import loop as oloop
    from etainv.pipeline import EtaLoop
    unet, get_engine = setup
    L = 16
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tokens = torch.ones(B, 8, dtype=torch.int32)
    ref = leg_forward_guidance_table()["ref"]
    loop = EtaLoop(eng, S=S, use_mask=False, guidance_scale_fwd=(1.0, 3.0))
    assert not loop.skip_uncond_fwd
    inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
    print(f"inversion trajectory {relerr(inv['latents'].cpu(), ref):.2e}")
    assert relerr(inv["latents"].cpu(), ref) < 2.3e-3                                             # measured 1.13e-3 (round 3); bound at 2x


@pytest.mark.parametrize("editor", ["ptp", "simple"])
def test_dead_source_rows_are_skipped_not_changed(setup, editor):
    """EtaLoop.skip_dead_source_rows: backward steps with eta(t) == 0 run without eps(uncond source) (3 B rows with prompt-to-prompt, 2 B without an
    attention coupling).  Same edited latent as the reference's row count (the source row differs by at most one rounding: x_prev instead of
    x + (x_prev - x)), fewer UNet rows issued."""
    from oracle import ptp as optp
    from etainv.pipeline import EtaLoop, PtpTables, noise_table
    unet, get_engine = setup
    L, S_, eta = 16, 8, [[0.6, 0], [1, 0.7]]                                   # eta == 0 for t <= 580: 5 of the 8 backward steps
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tok = optp.WordTokenizer()
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    mp, al, eq, ba, ca = [], [], [], [], []
    for i, (src, tgt) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
        bw, tw = ws[1], tgt.split(" ")[1]
        m, a = optp.refinement_mapper(src, tgt, tok)
        mp.append(m); al.append(a)
        eq.append(optp.equalizer(tgt, (tw,), (2,), tok))
        ba.append(optp.blend_alpha_layers([src, tgt], ((bw,), (tw,)), tok))
        ca.append(optp.time_words_alpha([src, tgt], S_, {"default_": .4}, tok)[:, 0])
    ptp = PtpTables(np.stack(mp), np.stack(al), np.stack(ca, 1), 0.6, S_, equalizer=np.stack(eq), blend_alpha=np.stack(ba)) if editor == "ptp" else None
    nz = noise_table(S_, 10, L, seed=0)
    outs, rows = [], []
    import os
    for skip in (True, False):
        os.environ["ETAINV_NO_SRC_EXIT"] = "1"            # (the early exit of the cond source rows has its own test below)
        try:
            loop = EtaLoop(eng, S=S_, eta=eta, skip_dead_source_rows=skip)
        finally:
            del os.environ["ETAINV_NO_SRC_EXIT"]
        inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
        r0 = loop.rows_executed
        outs.append(loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), nz, edit_word=torch.tensor([1, 1]), ptp=ptp).clone())
        rows.append(loop.rows_executed - r0)
    n_dead = int(sum(1 for t in loop.t_bwd if loop.etas[int(t)] == 0.0))
    assert n_dead == 5 and rows[1] == 4 * B * S_ and rows[0] == rows[1] - n_dead * B * (1 if editor == "ptp" else 2)
    e_src, e_tgt = relerr(outs[0][:B], outs[1][:B]), relerr(outs[0][B:], outs[1][B:])
    print(f"{editor}: dead-row skipping vs the full row count: source row {e_src:.2e}, edited latent {e_tgt:.2e}; UNet rows {rows[0]} vs {rows[1]}")
    assert e_src < 1e-6 and e_tgt < 2e-3                                       # (fp16: tile / split-K choices follow the row count)


def test_cond_source_rows_exit_early_not_changed(setup):
    """etainv_attn_ctrl.src_exit_block: in eta == 0 backward steps after the cross replacement and the self-replace have ended, the cond source rows
    leave the UNet after transformer block 9 (their last stored attention layer).  Same edited latent, same LocalBlend masks (the store is what
    they still feed), fewer FLOPs.  ETAINV_NO_SRC_EXIT=1 is the A/B switch."""
    import os
    from oracle import ptp as optp
    from etainv.pipeline import EtaLoop, PtpTables, noise_table
    unet, get_engine = setup
    L, S_, eta = 16, 10, [[0.6, 0], [1, 0.7]]        # eta == 0 for t <= 500: steps 4..9; cross alpha zero from step 4; self-replace until step 6: exit after block 12 in steps 4, 5, after block 9 in 6..9
    eng = get_engine(L, torch.float16)
    pairs, z0, ctx_src, ctx_tgt = _inputs(L)
    tok = optp.WordTokenizer()
    W = max(len(s.split(" ")) for s, _ in pairs)
    tokens = torch.ones(B, W, dtype=torch.int32)
    mp, al, eq, ba, ca = [], [], [], [], []
    for i, (src, tgt) in enumerate(pairs):
        ws = src.split(" ")
        tokens[i, :len(ws)] = torch.tensor([ws.index(w) + 1 for w in ws], dtype=torch.int32)
        bw, tw = ws[1], tgt.split(" ")[1]
        m, a = optp.refinement_mapper(src, tgt, tok)
        mp.append(m); al.append(a)
        eq.append(optp.equalizer(tgt, (tw,), (2,), tok))
        ba.append(optp.blend_alpha_layers([src, tgt], ((bw,), (tw,)), tok))
        ca.append(optp.time_words_alpha([src, tgt], S_, {"default_": .4}, tok)[:, 0])
    ptp = PtpTables(np.stack(mp), np.stack(al), np.stack(ca, 1), 0.6, S_, equalizer=np.stack(eq), blend_alpha=np.stack(ba))
    nz = noise_table(S_, 10, L, seed=0)
    outs, rows = [], []
    for off in (False, True):
        if off:
            os.environ["ETAINV_NO_SRC_EXIT"] = "1"
        try:
            loop = EtaLoop(eng, S=S_, eta=eta)
        finally:
            os.environ.pop("ETAINV_NO_SRC_EXIT", None)
        inv = loop.invert(z0.cuda(), ctx_src.cuda(), tokens.cuda())
        r0 = loop.rows_executed
        outs.append(loop.sample(inv, ctx_src.cuda(), ctx_tgt.cuda(), nz, edit_word=torch.tensor([1, 1]), ptp=ptp).clone())
        rows.append(loop.rows_executed - r0)
    dead = [i for i, t in enumerate(loop.t_bwd) if loop.etas[int(t)] == 0.0 and not ptp.cross_active[i]]
    n12 = sum(1 for i in dead if ptp.self_lo <= i < ptp.self_hi)
    n_exit = len(dead) - n12
    assert (n_exit, n12) == (4, 2) and abs((rows[1] - rows[0]) - B * (n_exit * (1 - loop.SRC_EXIT_SHARE) + n12 * (1 - loop.SRC_EXIT_SHARE_12))) < 1e-6
    e_src, e_tgt = relerr(outs[0][:B], outs[1][:B]), relerr(outs[0][B:], outs[1][B:])
    print(f"cond source rows exit after block 9 in {n_exit} steps, after block 12 in {n12}: source row {e_src:.2e}, edited latent {e_tgt:.2e}; UNet row-equivalents {rows[0]:.1f} vs {rows[1]:.1f}")
    assert e_src == 0.0 and e_tgt < 2e-3
