"""The benchmark's own shapes inside the driver-run suite (VERDICT r4 "What's missing" 3): UNet calls with 128 / 96 rows at L = 64 -- where every level
runs the persistent ring kernels (K = 640 / 1280 rings, PATCH conv, phase-form upsamplers, head-major QKV planes) -- and the S = 50 loop at B = 32.

  * 128 identical rows against the committed fp32 oracle output of one sample (`leg_unet_bench_shape`, oracle/unet.py); rows bit-equal to each other;
  * the backward layouts of the benchmark at B = 32 -- 4 B rows [u_s, u_t, c_s, c_t], 3 B rows [u_t, c_s, c_t], 3 B rows [u_t, c_t, c_s] with the cond
    source rows leaving after block 12 / 9 -- with prompt-to-prompt controls and per-image inputs, every image against the B = 1 call of the same image
    (self-comparison across tilings: the oracle leg above anchors the numerics);
  * S = 50 free-running, pair 0 of tests/test_s50_gpu.py as image 0 of a 32-image batch: best-of-n 50 / 50 against the oracle trace, the edited latent
    no further from the oracle than 1.5 x the reference-precision floor (the B = 2 test's bar).
Reference being matched: modules/inversion/eta_inversion.py:207-294 (predict_step_backward / diffusion_backward), :321 (the UNet call)."""
import pytest
import torch

from tests.oracle_cache import CACHE_DIR, load
from tests.test_unet_gpu import leg_unet_bench_shape, relerr

pytestmark = pytest.mark.gpu
L = 64


@pytest.fixture(scope="module")
def big_engines():
    from etainv.engine import Engine
    made = {}

    def get(dtype):
        if dtype not in made:
            e = Engine(dtype=dtype, max_unet_batch=128, latent_size=L, max_img=32)
            e.load_synthetic(0)
            made[dtype] = e
        return made[dtype]
    yield get
    for e in made.values():
        e.close()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2.5e-2)])
def test_unet_rows128_vs_oracle(big_engines, dtype, tol):
    e = big_engines(dtype)
    g = torch.Generator().manual_seed(1)
    x1, c1 = torch.randn(1, 4, L, L, generator=g), torch.randn(1, 77, 768, generator=g)
    ref = leg_unet_bench_shape()["ref"]
    errs = {}
    for rows in (16, 128):
        out = torch.empty(rows, 4, L, L, device="cuda")
        e.unet(x1.repeat(rows, 1, 1, 1).cuda().contiguous(), 500, c1.repeat(rows, 1, 1).cuda().contiguous(), None, out=out)
        torch.cuda.synchronize()
        assert all(torch.equal(out[0], out[i]) for i in range(rows)), f"{rows} identical rows differ"
        errs[rows] = relerr(out[0].cpu(), ref)
    print(f"{dtype}: rel L2 vs the fp32 oracle: 16 rows {errs[16]:.2e}, 128 rows {errs[128]:.2e}")
    assert errs[128] < tol and errs[128] <= 1.5 * errs[16], errs


def _ptp_tables(B, g):
    """per-image prompt-to-prompt tables of a Refine + Reweight edit (random but valid: a permutation-like mapper with -1 holes, alphas in {0, 1},
    an equalizer with one boosted word, a cross_alpha row that keeps the first words)"""
    mapper = torch.stack([torch.randperm(77, generator=g) for _ in range(B)]).int()
    mapper[:, 60:] = -1
    alphas = (torch.rand(B, 77, generator=g) > 0.2).float()
    eq = torch.ones(B, 77)
    eq[torch.arange(B), torch.randint(1, 12, (B,), generator=g)] = 2.0
    ca = torch.zeros(B, 77)
    ca[:, :20] = 1.0
    return mapper.cuda(), alphas.cuda(), eq.cuda(), ca.cuda()


LAYOUTS = [("u_s,u_t,c_s,c_t", 0, 0, True), ("u_t,c_s,c_t", 1, 0, True), ("u_t,c_t,c_s exit 12", 1, 12, True), ("u_t,c_t,c_s exit 9", 1, 9, False)]


@pytest.mark.parametrize("name,skip_us,exit_block,self_on", LAYOUTS, ids=[l[0].replace(",", "-").replace(" ", "_") for l in LAYOUTS])
def test_backward_layouts_b32_vs_single_image_calls(big_engines, name, skip_us, exit_block, self_on):
    from etainv.engine import AttnControl
    from etainv import _capi
    e = big_engines(torch.float16)
    B = 32
    g = torch.Generator().manual_seed(77 + exit_block + skip_us)
    lat = (0.9 * torch.randn(2 * B, 4, L, L, generator=g)).cuda()            # [src x B, tgt x B]
    ctx = torch.randn(4 * B, 77, 768, generator=g).cuda()                    # [u_s, u_t, c_s, c_t] x B
    mapper, alphas, eq, ca = _ptp_tables(B, g)
    live = exit_block == 0                                                   # (an exit needs mapper == NULL: nothing injected from the source any more)

    def call(imgs):
        n = len(imgs)
        idx = torch.tensor(imgs, device="cuda")
        role = lambda r: ctx[r * B + idx]
        src, tgt = lat[idx], lat[B + idx]
        ctrl = AttnControl(mode=_capi.ATTN_PTP, n_img=n, store_maps=True, mapper=mapper[idx].contiguous() if live else None,
                           alphas=alphas[idx].contiguous(), equalizer=eq[idx].contiguous(), cross_alpha=ca[idx].contiguous(),
                           self_replace_active=self_on, self_max_tokens=(L // 2) ** 2)
        if not skip_us:
            x, c, rows_out = torch.cat([src, tgt]), torch.cat([role(0), role(1), role(2), role(3)]), 4 * n
        elif exit_block:
            ctrl.c.first_row, ctrl.c.src_exit_block = n, exit_block
            x, c, rows_out = torch.cat([tgt, tgt, src]), torch.cat([role(1), role(3), role(2)]), 2 * n      # the exited rows have no output
        else:
            ctrl.c.first_row = n
            x, c, rows_out = torch.cat([tgt, src]), torch.cat([role(1), role(2), role(3)]), 3 * n
        out = torch.empty(c.shape[0], 4, L, L, device="cuda")
        e.maps_reset()
        e.unet(x.contiguous(), 481, c.contiguous(), ctrl, out=out)
        torch.cuda.synchronize()
        return out[:rows_out].reshape(rows_out // n, n, 4, L, L)

    full = call(list(range(B)))
    assert torch.isfinite(full).all()
    worst = 0.0
    for b in (0, 13, 31):
        one = call([b])
        for r in range(one.shape[0]):
            worst = max(worst, relerr(full[r, b], one[r, 0]))
    print(f"{name}: B = 32 vs single-image calls, worst rel L2 over roles and images {worst:.2e}")
    assert worst < 3e-3                                                      # two fp16 executions of the same arithmetic on different tilings


def test_config5_rows_b8_masactrl_vs_single_image_calls():
    """BASELINE config 5 at its real batch (etainv + masactrl, 768^2: L = 96, B = 8 -> 32 UNet rows [u_s, u_t, c_s, c_t] x 8; the engine takes the row-skipping
    layouts with prompt-to-prompt only, and config 5 has eta > 0 at every step anyway), MasaCtrl active from block 10 on, per-image inputs: every probed image
    against the B = 1 call of the same image.  At these
    sizes the call runs the dual-M conv at 96^2 and 48^2 (an odd number of patches per image row: tile pairs straddle patch rows and images), the ring at
    24^2 / 12^2, head-major self-attention at N = 9216 with the K / V batch remap.  Reference: modules/utils/masactrl.py:56-72, eta_inversion.py:207-273."""
    from etainv.engine import AttnControl, Engine
    from etainv import _capi
    L5, B = 96, 8
    e = Engine(dtype=torch.float16, max_unet_batch=4 * B, latent_size=L5, max_img=B)
    e.load_synthetic(0)
    try:
        g = torch.Generator().manual_seed(905)
        lat = (0.9 * torch.randn(2 * B, 4, L5, L5, generator=g)).cuda()      # [src x B, tgt x B]
        ctx = torch.randn(4 * B, 77, 768, generator=g).cuda()                # [u_s, u_t, c_s, c_t] x B

        def call(imgs):
            n = len(imgs)
            idx = torch.tensor(imgs, device="cuda")
            role = lambda r: ctx[r * B + idx]
            src, tgt = lat[idx], lat[B + idx]
            ctrl = AttnControl(mode=_capi.ATTN_MASA, n_img=n, store_maps=True, masa_active=True, masa_first_block=10)
            x, c = torch.cat([src, tgt]), torch.cat([role(0), role(1), role(2), role(3)])
            out = torch.empty(c.shape[0], 4, L5, L5, device="cuda")
            e.maps_reset()
            e.unet(x.contiguous(), 481, c.contiguous(), ctrl, out=out)
            torch.cuda.synchronize()
            return out.reshape(c.shape[0] // n, n, 4, L5, L5)

        full = call(list(range(B)))
        assert torch.isfinite(full).all()
        worst = 0.0
        for b in (0, 5, 7):
            one = call([b])
            for r in range(one.shape[0]):
                worst = max(worst, relerr(full[r, b], one[r, 0]))
        print(f"config 5 rows: B = 8 vs single-image calls at L = 96, worst rel L2 over roles and images {worst:.2e}")
        assert worst < 3e-3                                                  # two fp16 executions of the same arithmetic on different tilings
    finally:
        e.close()


# two units in the last place of the compute dtype: a best-of-n choice may only differ from another execution's where the two candidates' losses are closer
# than that (the largest gap observed at a fork in round 5 was 1.8e-3, bf16)
NEAR_TIE = {"bf16": 2 * 2.0 ** -8, "fp16": 2 * 2.0 ** -10}


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
def test_s50_b32_batch_invariance(kind):
    """the benchmark step itself (etainv + ptp, S = 50, B = 32) with the oracle-traced pairs 0 and 1 as images 0 and 1, and EVERY image against the B = 2 call
    of the same two inputs.  The argmin over ten noise candidates (reference eta_inversion.py:330-375) is discontinuous: the B = 32 and B = 2 calls run on
    different tile forms (not bit-identical), so a choice may flip where two candidates nearly tie, and the edited latent then differs by O(1) from there on.
    Asserted: every first fork is such a near-tie (relative loss gap <= 2 ulp of the dtype, measured with the B = 2 run's own losses, and with the oracle's for
    images 0 / 1); images without a fork end within the reference-precision floor of their B = 2 run; the number of forked images is printed and bounded."""
    from etainv.engine import Engine
    from tests.parity_s50 import native_run, compare
    S, B = 50, 32
    refs = [load(CACHE_DIR / "s50_pair0.npz"), load(CACHE_DIR / "s50_pair1.npz")]
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[kind]
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    floor = refs[0]["floors"][kind]["final_edit_rel_l2"]      # (the emulated reference-precision run exists for pair 0 only: tests/test_s50_gpu.py uses it for both)
    eng = Engine(dtype=dt, max_unet_batch=4 * B, latent_size=L, max_img=B)
    eng.load_synthetic(0)
    try:
        big = native_run(dt, S, L, 2, batch=B, engine=eng)
        small = []
        for k in range(0, B, 2):
            small += native_run(dt, S, L, 2, batch=B, select=[k, k + 1], engine=eng)
    finally:
        eng.close()
    forked, worst_gap, worst_unforked = [], 0.0, 0.0
    for b in range(B):
        diff = (big[b]["best"] != small[b]["best"]).nonzero().flatten().tolist()
        if diff:
            i = diff[0]                                          # the first fork: everything after it is another trajectory
            ls = small[b]["losses"][i]
            gap = float(abs(ls[int(big[b]["best"][i])] - ls[int(small[b]["best"][i])]) / ls[int(small[b]["best"][i])])
            forked.append((b, i + 1, gap))
            worst_gap = max(worst_gap, gap)
        else:
            worst_unforked = max(worst_unforked, rel(big[b]["out"][1], small[b]["out"][1]))
        assert rel(big[b]["out"][0], small[b]["out"][0]) <= 1e-5   # the replayed source row
    print(f"{kind}: S = 50, B = 32 vs the B = 2 calls of the same inputs: {len(forked)} of {B} images fork (image, backward step, relative loss gap of the two "
          f"candidates in the B = 2 run): {[(b, st, f'{g:.1e}') for b, st, g in forked]}; images without a fork: edited latent within {worst_unforked:.2e} "
          f"of their B = 2 run (floor {floor:.2e})")
    assert worst_gap <= NEAR_TIE[kind], forked
    assert worst_unforked <= 1.5 * floor
    # How many images fork is a property of the precision, not of a kernel: measured in round 6 (MI355X), fp16 3 of 32 (gaps <= 4.6e-5), bf16 29 of 32 (gaps
    # <= 2.5e-3 = a third of one bf16 ulp; ten candidates' losses lie within ~1 % of each other, and two bf16 executions of the UNet differ by ~1.2 x the
    # reference-precision floor).  The fp16 count is bounded as a guard against a systematic difference; the bf16 count is reported.
    if kind == "fp16":
        assert len(forked) <= B // 4, forked
    # ---- images 0 and 1 against the fp32 oracle traces
    for b, ref in enumerate(refs):
        e_small, e_big = rel(small[b]["out"][1], ref["out"][1]), rel(big[b]["out"][1], ref["out"][1])
        cmp_big = compare(big[b], ref)
        flips = cmp_big["best_of_n_flips"]
        fl = floor
        print(f"    image {b} vs the fp32 oracle: B = 2 run {e_small:.2e}, B = 32 run {e_big:.2e} (floor {fl:.2e}); best-of-n {cmp_big['best_of_n_agree']}/{S} {flips}")
        first_fork = min([f["bwd_step"] for f in flips], default=S + 1)
        assert all(f["oracle_rel_loss_gap"] <= NEAR_TIE[kind] for f in flips if f["bwd_step"] == first_fork), flips
        # (checkpoints by STEP NUMBER: the committed trace keeps steps 1, 5, 10, 25, 50, the run more -- `compare`'s per_step rows pair them by position)
        before = {s_: rel(big[b]["bwd"][big[b]["steps"].index(s_)][1], ref["bwd"][list(ref["steps"]).index(s_)][1])
                  for s_ in ref["steps"] if s_ < first_fork and s_ in big[b]["steps"]}
        inv_err = {s_: rel(big[b]["inv"][big[b]["steps"].index(s_)], ref["inv"][list(ref["steps"]).index(s_)]) for s_ in ref["steps"] if s_ in big[b]["steps"]}
        assert before and all(v <= 1.5 * fl for v in before.values()), before
        assert all(v <= 1.5 * fl for v in inv_err.values()), inv_err      # the forward pass has no choices: it must track the oracle to the end
        if not flips:
            assert e_big <= 1.5 * fl and e_big <= 1.5 * max(e_small, 0.5 * fl)
        assert rel(big[b]["out"][0], ref["out"][0]) <= 1e-5
