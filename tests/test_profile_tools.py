"""The scripts that turn rocprofv3 output into the figures of bench.py's `roofline` object and of profiles/ (tools/pmc_sq.py, pmc_traffic.py, launch_table.py):
fed synthetic counter files on CPU -- the kernel-family filter once missed a renamed kernel and halved `roofline.mfma_busy_frac` without a word."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
KERNELS = {  # mangled names as rocprofv3 prints them for this library's templates
    "_ZN6etainv12_GLOBAL__N_115pp_conv2_kernelIDF16bLb1EEEvNS_11IGemmParamsE": ("pp_conv2_kernel<bf16, true>", 600.0),
    "_ZN6etainv12_GLOBAL__N_114pp_conv_kernelIDF16bLb0EEEvNS_11IGemmParamsE": ("pp_conv_kernel<bf16, false>", 500.0),
    "_ZN6etainv12_GLOBAL__N_115pp_dualn_kernelIDF16bLi128ELi3ELb0ELi0EEEvNS_11IGemmParamsE": ("pp_dualn_kernel<bf16, 128, 3, false, 0>", 400.0),
    "_ZN6etainv12igemm_kernelIDF16bLi256ELi160ELi4ELi3ELi0ELi3ELb0EEEvNS_11IGemmParamsE": ("igemm_kernel<bf16, 256, 160, 4, 3, 0, 3, false>", 300.0),
    "_ZN6etainv18self_attn40_kernelIDF16bLi40ELb1ELi2ELi2EEEvPKT_PS1_iifiiiiii": ("self_attn40_kernel<bf16, 40, true, 2, 2>", 500.0),
}


def _counter_csv(path, counters):
    rows = ["Kernel_Name,Counter_Name,Counter_Value"]
    for name, (_, busy) in KERNELS.items():
        for c, v in counters(busy).items():
            rows.append(f'"{name}",{c},{v}')
    path.write_text("\n".join(rows) + "\n")


def test_pmc_sq_decodes_the_template_names_and_weights_the_whole_gemm_family(tmp_path):
    # every kernel: 1000 GPU cycles per XCD (x 8 XCDs summed), matrix pipes busy `busy` of 1000 cycles on each of the 1024 SIMDs
    _counter_csv(tmp_path / "p1.csv", lambda busy: {"GRBM_GUI_ACTIVE": 8000.0, "SQ_VALU_MFMA_BUSY_CYCLES": busy * 1024.0, "SQ_WAVE_CYCLES": 1e6, "SQ_BUSY_CU_CYCLES": 256000.0})
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_sq.py"), str(tmp_path / "p1.csv")], capture_output=True, text=True, check=True)
    d = json.loads(out.stdout)
    by = {e["kernel"]: e for e in d["kernels"]}
    for _, (readable, busy) in KERNELS.items():
        assert readable in by, (readable, sorted(by))
        assert abs(by[readable]["mfma_busy_frac"] - busy / 1000.0) < 1e-9
    fam = d["summary"]["igemm_family"]
    assert fam["kernels"] == 4                                   # both conv kernels, dual-N, the ring -- not the attention
    assert abs(fam["mfma_busy_frac"] - (0.6 + 0.5 + 0.4 + 0.3) / 4) < 1e-9
    assert abs(fam["share_of_gpu_cycles"] - 4 / 5) < 1e-9


def test_pmc_traffic_puts_both_conv_kernels_into_the_gemm_class(tmp_path):
    f, w = tmp_path / "fetch.csv", tmp_path / "write.csv"
    _counter_csv(f, lambda busy: {"FETCH_SIZE": 1000.0})
    _counter_csv(w, lambda busy: {"WRITE_SIZE": 500.0})
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_traffic.py"), str(f), str(w)], capture_output=True, text=True, check=True)
    d = json.loads(out.stdout)
    assert d["igemm"]["launches"] == 4 and d["self_attn"]["launches"] == 1


def test_launch_table_joins_the_trace_with_the_event_times(tmp_path):
    recs = [{"ms": 1.0, "flops": 2e12, "bytes": 1e9}, {"ms": 0.5, "flops": 5e11, "bytes": 1e9}, {"ms": 1.0, "flops": 2e12, "bytes": 1e9}]
    (tmp_path / "l.json").write_text(json.dumps({"rows": 128, "dtype": "bf16", "igemm": recs}))
    a = "igemm M=512 N=320 c1=320 c2=0 taps=9 stride=1 ups=0 H=64 W=64 geglu=0 ln=0 stat=1 res=1 rowvec=0 hm=0 route=ppconv"
    b = "igemm M=512 N=320 c1=320 c2=0 taps=1 stride=1 ups=0 H=1 W=512 geglu=0 ln=0 stat=0 res=1 rowvec=0 hm=0 route=dualn"
    # the trace holds the warm-up call's lines too: only the last len(recs) belong to the timed call
    (tmp_path / "t.txt").write_text("\n".join(["noise", b, b, b, a, b, a]) + "\n")
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "launch_table.py"), str(tmp_path / "l.json"), str(tmp_path / "t.txt")], capture_output=True, text=True, check=True)
    lines = out.stdout.splitlines()
    assert lines[0].startswith("3 launches, 2.50 ms")
    assert "x  2" in lines[1] and "2000.0 TF/s" in lines[1] and lines[1].endswith("route=ppconv")
    assert any(l.startswith("route ppconv") and "80.0%" in l for l in lines)
    assert any(l.startswith("route dualn") and "20.0%" in l for l in lines)


def test_counter_figures_are_reported_only_for_the_sources_they_were_taken_on(tmp_path):
    """bench.py's `roofline.traffic` / `mfma_busy_frac` are read from committed PMC summaries; a summary taken on other kernel sources must come back as
    null + a `stale:` note, never as a number (tools/srcstamp.py)."""
    sys.path.insert(0, str(ROOT / "tools"))
    from srcstamp import kernel_src_sha16, stamped_figure
    sha = kernel_src_sha16()
    assert len(sha) == 16 and sha == kernel_src_sha16()
    pick = lambda d: d.get("igemm", {}).get("hbm_bytes_per_launch")
    (tmp_path / "r01_pmc_traffic_rows128.json").write_text(json.dumps({"igemm": {"hbm_bytes_per_launch": 1.0}, "kernel_src_sha16": sha}))
    assert stamped_figure(tmp_path, "r*_pmc_traffic_rows128.json", sha, pick, "x")[0] == 1.0
    (tmp_path / "r02_pmc_traffic_rows128.json").write_text(json.dumps({"igemm": {"hbm_bytes_per_launch": 2.0}, "kernel_src_sha16": "0" * 16}))
    v, note = stamped_figure(tmp_path, "r*_pmc_traffic_rows128.json", sha, pick, "x")     # the NEWEST file decides: no silent fall-back to an older round's number
    assert v is None and note.startswith("stale: profiles/r02_pmc_traffic_rows128.json")
    (tmp_path / "r03_pmc_traffic_rows128.json").write_text(json.dumps({"igemm": {"hbm_bytes_per_launch": 3.0}}))
    v, note = stamped_figure(tmp_path, "r*_pmc_traffic_rows128.json", sha, pick, "x")
    assert v is None and "unstamped" in note
    assert stamped_figure(tmp_path, "nothing*.json", sha, pick, "x") == (None, None)
    # the aggregators stamp their output
    _counter_csv(tmp_path / "p1.csv", lambda busy: {"GRBM_GUI_ACTIVE": 8000.0, "SQ_VALU_MFMA_BUSY_CYCLES": busy * 1024.0})
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_sq.py"), str(tmp_path / "p1.csv")], capture_output=True, text=True, check=True)
    assert json.loads(out.stdout)["kernel_src_sha16"] == sha
