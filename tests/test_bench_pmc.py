"""bench.py's lookup of committed PMC summaries is by kernel INSTANTIATION (a substring match printed the PAIR
kernel's instruction count as C3's).  CPU test: reads profiles/*.json only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _committed(kernel, key):
    """the newest committed summary that holds the instantiation, read directly"""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        d = json.load(open(f))
        e = d.get("msc::" + kernel)
        if e and key in e:
            best = e[key].get("total", e[key].get("avg"))
    return best


def test_exact_instantiation_or_nothing():
    full = "k_score_tile_roles<false, false, false>"
    v, src = bench.pmc_entry(full, "SQ_INSTS_VALU")
    assert v is not None and v == _committed(full, "SQ_INSTS_VALU") and src.endswith("_pmc.json")
    # the PAIR instantiation is another kernel with other counters
    vp, _ = bench.pmc_entry("k_score_tile_roles<false, false, true>", "SQ_INSTS_VALU")
    assert vp is not None and vp != v
    # a prefix, a template-less name, a neighbour: nothing -- never somebody else's counters
    for name in ("k_score_tile", "k_score_tile_roles", "k_score_tile_roles<false, false", "k_score_niw"):
        assert bench.pmc_entry(name, "SQ_INSTS_VALU") == (None, None), name


def test_niw_f32_and_f64_kernels_are_told_apart():
    a, _ = bench.pmc_entry("k_score_niw<2, false, false>", "SQ_VALU_MFMA_BUSY_CYCLES")
    b, _ = bench.pmc_entry("k_score_niw64<2, 4, false, false>", "SQ_VALU_MFMA_BUSY_CYCLES")
    assert a is not None and b is not None and a != b
    assert a == _committed("k_score_niw<2, false, false>", "SQ_VALU_MFMA_BUSY_CYCLES")


def test_traffic_ratio_is_counter_bytes_over_algorithmic_bytes():
    t = bench.traffic_of("k_score_nich1<false, false, 4, false>", 1.028e9)
    assert t["traffic"] is not None and abs(t["traffic_ratio"] - t["traffic"] / 1.028e9) < 1e-12
    assert bench.traffic_of("k_no_such_kernel<1>", 1.0)["traffic"] is None


def test_weighted_vector_issue_prices_transcendentals_apart():
    """bench.valu_weighted: SQ_ACTIVE_INST_VALU counts a transcendental twice, so busy - insts of them at 8 cycles and the
    rest at the SIMD's 2 (tools/microbench/valu_rate.hip), against 256 x 4 SIMDs x 2.4 GHz"""
    import bench
    w = bench.valu_weighted(insts=1.0e9, busy_slots=1.2e9, ms=2.0)
    cyc = 0.8e9 * 2.0 + 0.2e9 * 8.0
    assert abs(w["weighted_simd_cycles_per_launch"] - cyc) < 1.0
    assert abs(w["frac_weighted"] - cyc / (2.0e-3 * 256 * 4 * 2.4e9)) < 1e-12
    # no transcendental at all: busy == insts
    assert abs(bench.valu_weighted(1.0e9, 1.0e9, 1.0)["frac_weighted"] - 2.0e9 / (1.0e-3 * 256 * 4 * 2.4e9)) < 1e-12
