"""Full-length parity in the driver-run suite: etainv + prompt-to-prompt, L = 64, S = 50, 2 pairs, the benchmark's eta [[0.6, 0], [1, 0.7]] /
n = 10 / cfg 7.5, FREE-RUNNING, against the fp32 CPU oracle's committed end states (reference modules/inversion/eta_inversion.py:207-294,
50 backward steps; the oracle run costs 300 UNet sample-forwards = 50 min per pair on the GPU box's host, so it is stored:
tests/golden/oracle_cache/s50_pair{0,1}.npz, written by `make_oracle_cache.py --from-parity-cache` from `tests/parity_s50.py`'s oracle workers).

This is the loop as the benchmark runs it: 30 of the 50 backward steps have eta(t) = 0, so the three-row layout [u_t, c_s, c_t], the exit of the
cond source rows after transformer block 12 / 9 and the shared context-independent prefix are all on the path that is compared with the oracle.

  * best-of-n: the native choice equals the oracle's at all 50 steps in fp32; in fp16 / bf16 a choice may differ only where the oracle's own two candidates
    nearly tie (relative loss gap <= 2 ulp of the dtype: the argmin is discontinuous, see the comment in the test), and up to such a fork ...
  * fp16 / bf16: ... the edited latent stays within 1.5 x the reference-precision floor -- what the reference's own 16-bit execution (emulated on the oracle,
    oracle/lowprec.py, stored next to the fp32 run) loses against fp32 on the same inputs: 4.0e-3 / 3.2e-2 at step 50;
  * fp32-operand engine: rel L2 <= 1e-5 and >= 99.9 % of the elements inside north_star's rtol 1e-3 / atol 1e-4."""
import pytest
import torch

from tests.oracle_cache import CACHE_DIR, load

pytestmark = pytest.mark.gpu
S, L, PAIRS = 50, 64, 2
DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}
NEAR_TIE = {"bf16": 2 * 2.0 ** -8, "fp16": 2 * 2.0 ** -10}     # 2 ulp of the compute dtype (the same bound as tests/test_bench_shape_gpu.py)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture(scope="module")
def refs():
    return [load(CACHE_DIR / f"s50_pair{i}.npz") for i in range(PAIRS)]


@pytest.mark.parametrize("kind", ["fp16", "bf16", "fp32"])
def test_s50_free_running_vs_cached_oracle(refs, kind):
    from tests.parity_s50 import native_run, keep_steps
    runs = native_run(DTYPES[kind], S, L, PAIRS)
    ks = keep_steps(S)
    floor = refs[0]["floors"].get(kind)                      # recorded for pair 0 (one 50-minute emulated run per precision)
    fails = []
    for i, (sub, ref) in enumerate(zip(runs, refs)):
        agree = int((sub["best"] == ref["best"]).sum())
        flips = [(s + 1, int(sub["best"][s]), int(ref["best"][s]),
                  float(abs(ref["losses"][s][int(sub["best"][s])] - ref["losses"][s][int(ref["best"][s])]) / ref["losses"][s][int(ref["best"][s])]))
                 for s in (sub["best"] != ref["best"]).nonzero().flatten().tolist()]
        a, b = sub["out"][1].double(), ref["out"][1].double()
        within = float(((a - b).abs() <= 1e-4 + 1e-3 * b.abs()).double().mean())
        e_fin, e_src = rel(a, b), rel(sub["out"][0], ref["out"][0])
        line = [f"{kind} pair {i}: best-of-n {agree}/{S}; final edited latent rel L2 {e_fin:.2e} (max abs {float((a - b).abs().max()):.2e} on |x| <= "
                f"{float(b.abs().max()):.1f}), share inside rtol 1e-3 / atol 1e-4 {within:.4f}; source row {e_src:.2e}; edit-word map {rel(sub['map'], ref['map']):.2e}"]
        # fp32: every choice must be the oracle's.  16-bit modes: the argmin over ten candidates whose losses lie within ~1 % of each other is discontinuous --
        # measured in round 6, two 16-bit executions of the same inputs on different tilings fork on 3 of 32 (fp16) / 29 of 32 (bf16) images
        # (tests/test_bench_shape_gpu.py::test_s50_b32_batch_invariance) -- so a choice may differ from the fp32 oracle's, but only where the ORACLE's own two
        # candidates nearly tie (relative loss gap <= 2 ulp of the dtype); everything up to the first such fork must stay inside the floor, and without a fork
        # so must the result.  (Rounds 3-5 happened to agree 50 / 50 on both pairs; the round-6 attention kernel rounds P against another reference maximum on
        # a few tiles and pair 1 takes the other candidate of a 8.5e-4 tie at step 16 in bf16.)
        first_fork = flips[0][0] if flips else S + 1
        for s in ref["steps"]:
            k_sub, k_ref = ks.index(s), ref["steps"].index(s)
            e_inv, e_edit = rel(sub["inv"][k_sub], ref["inv"][k_ref]), rel(sub["bwd"][k_sub][1], ref["bwd"][k_ref][1])
            fl = f" (floor {floor['edit_rel_l2'][str(s)]:.2e})" if floor else ""
            line.append(f"    step {s:2d}: inversion {e_inv:.2e}, edited {e_edit:.2e}{fl}" + (" [behind the fork]" if s >= first_fork else ""))
            if floor:
                if s < first_fork and e_edit > 1.5 * floor["edit_rel_l2"][str(s)]:
                    fails.append(f"pair {i} step {s}: edited latent {e_edit:.2e} > 1.5 x floor {floor['edit_rel_l2'][str(s)]:.2e}")
                if e_inv > max(1.5 * floor["inv_rel_l2"][str(s)], 1e-6):
                    fails.append(f"pair {i} step {s}: inversion latent {e_inv:.2e} > 1.5 x floor {floor['inv_rel_l2'][str(s)]:.2e}")
        print("\n".join(line))
        if agree != S and (kind == "fp32" or flips[0][3] > NEAR_TIE[kind]):
            fails.append(f"pair {i}: best-of-n differs at (step, native, oracle, oracle's relative loss gap) {flips}")
        if e_src > 1e-5:
            fails.append(f"pair {i}: source row {e_src:.2e} (the replay of the inversion trajectory is exact up to fp32 rounding)")
        if kind == "fp32":
            if e_fin > 1e-5:
                fails.append(f"pair {i}: fp32 edited latent {e_fin:.2e} > 1e-5")
            if within < 0.999:
                fails.append(f"pair {i}: {within:.5f} of the fp32 edited latent inside rtol 1e-3 / atol 1e-4")
        elif not flips:
            if e_fin > 1.5 * floor["final_edit_rel_l2"]:
                fails.append(f"pair {i}: final edited latent {e_fin:.2e} > 1.5 x floor {floor['final_edit_rel_l2']:.2e}")
    assert not fails, fails
