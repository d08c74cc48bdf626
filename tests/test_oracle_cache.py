"""The committed oracle results (tests/golden/oracle_cache/, see tests/oracle_cache.py) cannot drift from oracle/: every leg the GPU tests read
has a file, one small leg is recomputed live in the default CPU suite, and all of them under ETAINV_SLOW=1 (about an hour on 8 cores)."""
import numpy as np
import pytest
import torch

from tests.golden.make_oracle_cache import import_legs

oc = import_legs()
ALL = [(name, args) for name, (fn, cases) in oc.LEGS.items() for args in cases]


def _leaves(obj, path="r"):
    if isinstance(obj, dict):
        for k, v in obj.items():
            yield from _leaves(v, f"{path}/{k}")
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _leaves(v, f"{path}/{i}")
    else:
        yield path, obj


def _compare(live, cached, tol):
    a, b = dict(_leaves(live)), dict(_leaves(cached))
    assert a.keys() == b.keys()
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, torch.Tensor):
            assert x.shape == y.shape and x.dtype == y.dtype, k
            if x.is_floating_point():
                # (NaN entries -- the unused slots of the optimiser legs' loss tables -- must sit at the same places; the norm runs over the rest)
                assert torch.equal(torch.isnan(x), torch.isnan(y)), f"{k}: NaN pattern differs"
                x, y = torch.nan_to_num(x.double(), nan=0.0), torch.nan_to_num(y.double(), nan=0.0)
                err = float((x - y).norm() / y.norm().clamp_min(1e-30))
                assert err <= tol, f"{k}: live oracle and committed result differ by {err:.2e}"
            else:
                assert torch.equal(x, y), k
        elif isinstance(x, float):
            assert abs(x - y) <= 0.05 * abs(y) + 1e-12, (k, x, y)          # floors are ratios of small differences
        else:
            assert x == y, (k, x, y)


def test_pack_round_trip(tmp_path):
    obj = {"a": torch.randn(3, 4), "i": torch.arange(5), "h": torch.randn(2).half(), "n": np.arange(3.0), "s": "x", "none": None,
           "l": [{"t": 3, "best": 7, "losses": torch.rand(10)}, (1.5, True)]}
    oc.save(tmp_path / "x.npz", obj)
    back = oc.load(tmp_path / "x.npz")
    assert back["s"] == "x" and back["none"] is None and back["l"][1] == (1.5, True) and back["l"][0]["best"] == 7
    assert torch.equal(back["a"], obj["a"]) and back["i"].dtype == torch.int64 and back["h"].dtype == torch.float16 and torch.equal(back["h"], obj["h"])
    assert np.array_equal(back["n"], obj["n"])


def test_every_leg_of_the_gpu_suite_has_a_committed_result():
    missing = [oc.key_of(n, a) for n, a in ALL if not (oc.CACHE_DIR / f"{oc.key_of(n, a)}.npz").exists()]
    assert not missing, f"run tests/golden/make_oracle_cache.py: {missing}"
    assert all((oc.CACHE_DIR / f"s50_pair{i}.npz").exists() for i in range(2))


def test_no_committed_result_is_stale():
    """MANIFEST.json records the fingerprint of oracle/*.py and of each leg's source its result was computed from: an edit of either makes the
    entry stale (the GPU suite would recompute it live) -- regenerate with tests/golden/make_oracle_cache.py"""
    manifest = oc.read_manifest()
    assert set(oc.key_of(n, a) for n, a in ALL) <= set(manifest), "entries without a fingerprint: make_oracle_cache.py --stamp-manifest after a live recheck"
    stale = {oc.key_of(n, a): why for n, a in ALL if (why := oc.stale_reason(oc.key_of(n, a), oc.LEGS[n][0]))}
    assert not stale, stale


# three cheap legs, one of each kind the GPU suite leans on -- a whole-UNet call, a prompt-to-prompt invert + edit loop, the attention-layer subsets of
# the map store -- recomputed live in the DEFAULT CPU suite: a stale loop / PtP leg fails here, not only under ETAINV_SLOW=1
LIVE = [("test_unet_gpu.leg_unet_forward", (16, 2, 500)), ("test_e2e_gpu.leg_edit", ("ptp", 16, True)), ("test_e2e_gpu.leg_attn_layers", ("mid",))]


@pytest.mark.parametrize("name,args", LIVE, ids=[oc.key_of(n, a) for n, a in LIVE])
def test_small_leg_recomputed_live_matches_the_committed_result(name, args):
    """SD1.x-width fp32 oracle at L = 16 (10 s to build the 860 M-parameter oracle once per process, seconds per leg)"""
    with torch.no_grad():
        live = oc.LEGS[name][0](*args)
    _compare(live, oc.load(oc.CACHE_DIR / f"{oc.key_of(name, args)}.npz"), 2e-5)


@pytest.mark.slow
@pytest.mark.parametrize("name,args", ALL, ids=[oc.key_of(n, a) for n, a in ALL])
def test_leg_recomputed_live_matches_the_committed_result(name, args):
    """thread count and blocking of the CPU kernels differ between hosts: fp32 summation order only (<= 1e-5 on the stored tensors; best-of-n
    indices must be equal)"""
    with torch.no_grad():
        live = oc.LEGS[name][0](*args)
    _compare(live, oc.load(oc.CACHE_DIR / f"{oc.key_of(name, args)}.npz"), 2e-5)
