"""PIE-Bench loader mirror (eta-inversion_amd/dataset/pie_bench_data.py) vs records produced by the reference's own
dataset/pie_bench_data.py on a synthetic mapping file (tests/golden/pie_bench.json, masks bit-packed in pie_bench_masks.npz)."""
import json
import os
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden"


def test_records_and_masks(tmp_path):
    from dataset.pie_bench_data import PieBenchData, edit_image_name
    g = json.loads((GOLD / "pie_bench.json").read_text())
    masks = np.load(GOLD / "pie_bench_masks.npz")
    (tmp_path / "mapping_file.json").write_text(json.dumps(g["mapping"]))
    data = PieBenchData(str(tmp_path), skip_img_load=True)
    assert len(data) == len(g["records"])
    for i, want in enumerate(g["records"]):
        s = data[i]
        assert s["source_prompt"] == want["source_prompt"] and s["target_prompt"] == want["target_prompt"]
        assert os.path.relpath(s["image_file"], tmp_path) == want["image_rel"]
        assert s["edit_word_idx"] == want["edit_word_idx"]
        assert json.loads(json.dumps(s["edit"]["ptp"])) == want["ptp"]
        assert s["image"] is None
        m = s["mask"].numpy()
        assert m.shape == (512, 512) and m.dtype == np.float32
        assert np.array_equal(np.packbits(m.astype(np.uint8)), masks[f"mask{i}"])          # bit-exact
        assert float(m.sum()) == want["mask_sum"]
    # bracket stripping, missing blend word -> None, clipping of a run that passes the end of the image
    assert "[" not in data[0]["source_prompt"] and data[3]["edit_word_idx"][1] is None and data[2]["edit"]["ptp"]["blend_words"] is None
    assert edit_image_name(7, "a b", "a c") == "0007_a b_a c"


def test_limit_and_categories(tmp_path):
    from dataset.pie_bench_data import PieBenchData
    g = json.loads((GOLD / "pie_bench.json").read_text())
    (tmp_path / "mapping_file.json").write_text(json.dumps(g["mapping"]))
    assert len(PieBenchData(str(tmp_path), skip_img_load=True, limit=2)) == 2
    assert len(list(PieBenchData(str(tmp_path), skip_img_load=True, limit=3))) == 3
    assert list(PieBenchData.categories["9_change_style"]) == list(range(620, 700))
