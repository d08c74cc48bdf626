#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc SQ passes per kernel instantiation: matrix-pipe busy share and where the waves' cycles go.

Input: counter_collection.csv files of one or more passes over the SAME command (tools/unet_call.py --rows 128), e.g.
    pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
    pass 2: SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f32_32x32x16, 16 per 16x16x32),
summed over the chip's 1024 SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    wait_any / wait_inst / active = share of SQ_WAVE_CYCLES (the three are disjoint and add up to ~1)
Only dispatches of the LAST `--last-frac` of the run are counted when given (the first UNet call is the warm-up).

    python tools/pmc_sq.py pass1/counter_collection.csv pass2/counter_collection.csv > profiles/rNN_pmc_sq_rows128.json
"""
import collections
import csv
import functools
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from srcstamp import kernel_src_sha16  # noqa: E402


@functools.lru_cache(maxsize=None)
def short(name):
    """readable kernel instantiation: rocprofv3 reports mangled names for templates, and neither c++filt nor the ROCm image's tools know the bf16 / f16
    manglings (DF16b / DF16_), so the template arguments of this library's kernels are decoded here"""
    m = re.match(r"^_ZN6etainv(?:12_GLOBAL__N_1)?(\d+)", name)
    if m:
        n = int(m.group(1))
        start = m.end()
        ident, rest = name[start:start + n], name[start + n:]
        args = []
        if rest.startswith("I"):
            for tok in re.finditer(r"DF16b|DF16_|f|Li(\d+)E|Lb([01])E", rest[1:rest.find("EEv") + 1 if "EEv" in rest else len(rest)]):
                t = tok.group(0)
                args.append("bf16" if t == "DF16b" else "f16" if t == "DF16_" else "float" if t == "f" else tok.group(1) if t.startswith("Li") else
                            ("true" if tok.group(2) == "1" else "false"))
        return ident + ("<" + ", ".join(args) + ">" if args else "")
    name = re.sub(r"^void ", "", name)
    name = name.replace("etainv::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16").replace("_Float16", "f16")


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    out = []
    tot_gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in per.values())
    for k, v in per.items():
        launches = max(n[k].values())
        e = {"kernel": k, "launches": launches}
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        if gui > 0:
            e["share_of_gpu_cycles"] = gui / tot_gui
            e["gpu_cycles_per_launch"] = gui / 8 / launches
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                e["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8.0)
            if "SQ_BUSY_CU_CYCLES" in v:
                e["cu_busy_frac"] = v["SQ_BUSY_CU_CYCLES"] / (256.0 * gui / 8.0)
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        # (SQ_WAVE_CYCLES may be collected in several passes: normalise the per-pass mean)
        wc_passes = max(1, n[k].get("SQ_WAVE_CYCLES", 0) // max(launches, 1))
        wc /= wc_passes
        if wc > 0:
            for c, nm in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                          ("SQ_WAIT_INST_LDS", "wait_inst_lds"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds"),
                          ("SQ_VALU_MFMA_COEXEC_CYCLES", "valu_mfma_coexec")):
                if c in v:
                    e[nm + "_of_wave_cycles"] = v[c] / wc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                # waves per SIMD x quad-cycles -> what share of the resident waves' lifetime the SIMD's matrix pipe was busy (2 waves per SIMD: x 2)
                e["mfma_busy_cycles_per_wave_quadcycle"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / wc
        if v.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
            e["lds_conflict_share"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
        e["raw"] = {c: x for c, x in sorted(v.items())}
        out.append(e)
    out.sort(key=lambda e: -e.get("share_of_gpu_cycles", 0.0))
    # the implicit-GEMM family (what bench.py's roofline line is about): matrix-pipe busy share weighted by GPU cycles
    fam = [e for e in out if any(t in e["kernel"] for t in ("igemm_kernel", "pp_conv_kernel", "pp_conv2_kernel", "pp_dualn_kernel", "pp_gemm_kernel")) and "mfma_busy_frac" in e]
    w = sum(e["share_of_gpu_cycles"] for e in fam)
    summary = {"igemm_family": {"share_of_gpu_cycles": w, "mfma_busy_frac": sum(e["mfma_busy_frac"] * e["share_of_gpu_cycles"] for e in fam) / w if w else None,
                                "kernels": len(fam)}}
    print(json.dumps({"kernel_src_sha16": kernel_src_sha16(), "summary": summary, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
