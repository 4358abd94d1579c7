"""Host-side pieces of bench.py and of the resident path that need no GPU: the compact summary at the end of the line, the roofline
helpers, the telemetry sampler on a box without a card, the runs of a re-solve mask."""

import json

import numpy as np


def _line():
    rf = {"frac": 0.7812345, "avg_launch_ms": 2.61234, "mfma_probe_tflops": 77.1234, "mfma_probe_sustained_tflops": 77.9}
    leg = lambda v: {"value": v, "ms_per_stamp": 1e3 / v, "roofline": {"frac": 0.77}, "job_roofline_frac": 0.55}  # noqa: E731
    return {
        "value": 1662.123456, "ms_per_step": 154.0123, "n_gpus": 1, "roofline": rf, "step_ms": {"all": [1.0] * 20, "min": 153.1, "median": 154.0, "max": 158.2},
        "telemetry": {"sclk_mhz": {"min": 2250.0, "median": 2290.0, "max": 2320.0}, "power_w": {"median": 1270.0}, "temp_junction_c": {"max": 54.0}},
        "roofline_chol": {"frac": 0.54}, "roofline_build_A": {"frac": 0.94},
        "block": {"value": 1320.0, "ms_per_stamp": 0.757}, "eigen_block": {"value": 183.0, "ms_per_stamp": 5.4, "batches_run": [256]},
        "block_seam": {"value": 790.0, "host_threads": 16, "one_host_thread": {"value": 230.0}, "several_passes": {"value": 1000.0}},
        "kernel_seam": {"ms_per_stamp": 7.25, "ms_per_stamp_group4": 2.99},
        "configs": {"cfg1": leg(3650.0), "cfg4": leg(810.0), "cfg5": leg(208.0),
                    "cfg3": {"b32": {"ms_per_stamp": 7.3, "roofline": {"frac": 0.2}, "roofline_hbm": {"frac": 0.34}},
                             "b256": {"ms_per_stamp": 5.3, "roofline": {"frac": 0.28}, "roofline_hbm": {"frac": 0.5}}, "value": 188.0},
                    "paper4": dict(leg(75.0), stamps_repaired=128, batch=128, roofline_kernels={"frac": 0.74}, block={"value": 63.0, "seconds_per_block": 112.0},
                                   cpu_baseline={"value": 0.13})},
        "farm": {"value": 3400.0, "ms_per_stamp": 2.3, "makespan_s": 4.8, "blocks": 16, "ranks_seen": 8},
        "cpu_baseline": {"value": 3.2, "cores": 256, "one_thread": {"value": 1.6}, "processes": {"value": 19.0, "processes": 64}},
    }


def test_summary_is_compact_and_carries_every_leg():
    """A driver that keeps the last 2 000 characters of bench.py's line must find every leg in them: the summary is
    the LAST key and stays well below that size with every leg present; a leg that failed shows as an error, not as a gap."""
    import bench

    out = _line()
    sm = bench.summary_of(out)
    text = json.dumps(sm)
    assert len(text) < 1800, len(text)
    for key in ("headline", "block", "eigen_block", "block_seam", "kernel_seam", "cfg1", "cfg3", "cfg4", "cfg5", "paper4", "farm", "cpu"):
        assert key in sm, key
    assert sm["headline"]["v"] == 1662.0 and sm["headline"]["f"] == 0.7812 and sm["headline"]["sclk"] == [2250.0, 2290.0, 2320.0]
    assert sm["headline"]["chol"] == 0.54 and sm["headline"]["build_A"] == 0.94 and sm["paper4"]["stamps_repaired"] == 128
    assert sm["cfg3"]["b256"]["symv4_f"] == 0.5 and sm["paper4"]["block"]["s_per_block"] == 112.0 and sm["farm"]["ranks_seen"] == 8
    out["configs"]["cfg5"] = {"error": "ImcomError: out of memory " + "x" * 500}
    out["block"] = {"error": "boom"}
    sm = bench.summary_of(out)
    assert sm["cfg5"] == {"error": ("ImcomError: out of memory " + "x" * 500)[:60]} and sm["block"] == {"error": "boom"}
    out["summary"] = sm
    assert list(out)[-1] == "summary" and len(json.dumps(sm)) < 1800


def test_roofline_helpers_and_telemetry_without_a_card():
    import bench

    n = np.full(256, 2204.0)
    a = bench.roofline_build_A(n, 27.8)
    assert a["bound"] == "l2-gather" and abs(a["samples_per_s"] - 256 * 2204 * 2205 / 2 / 27.8e-3) < 1e-3 * a["samples_per_s"]
    assert abs(a["achieved"] - a["samples_per_s"] * 800 / 1e9) < 1e-6 * a["achieved"] and 0.9 < a["frac"] < 1.0
    c = bench.roofline_chol(n, 1, 19.7, 1.65, 36)
    assert c["bound"] == "mfma" and abs(c["achieved"] - 256 * 2204.0**3 / 3 / 21.35e-3 / 1e12) < 1e-6 * c["achieved"] and 0.5 < c["frac"] < 0.6
    assert bench.spread([3.0, 1.0, 2.0]) == {"min": 1.0, "median": 2.0, "max": 3.0} and bench.spread([]) is None
    t = bench.Telemetry(0)  # no GPU here: nothing to sample, never an error
    assert t.start() is t and t.stop()["source"] is None


def test_runs_of_a_mask():
    from pyimcom_amd.stamps import _runs

    assert _runs([False, True, True, False, True]) == [(1, 3), (4, 5)]
    assert _runs([True] * 4) == [(0, 4)] and _runs([False] * 3) == [] and _runs([]) == []
