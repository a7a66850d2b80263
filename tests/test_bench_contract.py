"""bench.py pieces that can be checked without a GPU: the PMC-traffic gate (a traffic figure
from another workload or another kernel variant must not reach the bench line) and the argument surface."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_traffic_is_taken_only_from_a_file_collected_on_the_same_workload(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # no file at all
    v, why = bench.load_pmc_traffic("k_conv_wino", 8, "panoptic")
    assert v is None and "no profiles" in why
    # a pre-round-3 file (no workload tag, flat keys): refused
    (prof / "r02_pmc_traffic.json").write_text(json.dumps({"conv_wino_bytes_per_launch": 1.3e8}))
    v, why = bench.load_pmc_traffic("k_conv_wino", 8, "panoptic")
    assert v is None and "refused" in why
    # a tagged file: accepted for its own workload and class, refused for another batch / config / class
    good = {"workload": {"config": "panoptic", "frames_per_step": 8}, "collected": "2026-09-27", "commit": "abc",
            "classes": {"k_conv_wino": {"bytes_per_launch": 1.9e8, "launches": 64,
                                        "variants": {"fvp::k_conv_wino<2, 4, 8, true, false>": {"bytes_per_launch": 2.0e8}}},
                        "k_project_triplane": {"bytes_per_launch": 4.6e8, "launches": 4, "variants": {}}}}
    (prof / "r03_pmc_traffic.json").write_text(json.dumps(good))
    v, src = bench.load_pmc_traffic("k_conv_wino", 8, "panoptic")
    assert v == 1.9e8 and "r03_pmc_traffic.json" in src and "k_conv_wino<2, 4, 8, true, false>" in src
    assert bench.load_pmc_traffic("k_conv_wino", 1, "panoptic")[0] is None
    assert bench.load_pmc_traffic("k_conv_wino", 8, "shelf")[0] is None
    assert bench.load_pmc_traffic("k_conv_dma", 8, "panoptic")[0] is None            # class absent
    assert bench.load_pmc_traffic("k_project_triplane", 8, "panoptic")[0] is None    # no variants listed


def test_committed_traffic_file_matches_the_headline_workload():
    v, src = bench.load_pmc_traffic("k_conv_wino", 8, "panoptic")
    assert v and v > 1e8, src


def test_argument_surface():
    a = bench.parse_args([])
    assert (a.gpus, a.batch, a.config, a.streams) == (1, 8, "panoptic", 4) and a.steps > 0 and a.warmup >= 0
    a = bench.parse_args(["--gpus", "8", "--config", "panoptic128", "--batch", "1", "--steps", "5", "--warmup", "2"])
    assert (a.gpus, a.config, a.batch, a.steps, a.warmup) == (8, "panoptic128", 1, 5, 2)
    assert bench.parse_args(["--backbone"]).backbone
