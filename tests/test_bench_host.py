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


def _full_line(n_gpus=1):
    """A worst-case `out` of bench.main(): every leg present, long strings, long lists."""
    import bench

    out = _line()
    out.update({
        "metric": "postage-stamps/sec (and ms/stamp) for N~2k A-solve", "unit": "postage-stamps/s", "n_gpus": n_gpus, "ranks_seen": n_gpus, "steps": 20, "warmup": 5,
        "ms_per_stamp": 0.6012345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "per_rank_value": [1662.123456789] * n_gpus,
        "config": {"workload": "BASELINE configs[1] (cfg2): one batch of 256 48x48-output stamps (fade 0) per GPU per step, 6 exposures, analytic Roman-like PSF, "
                               "Cholesky kappa/C=0.0006, fp64" + " and more words" * 40, "stamps_per_step_per_gpu": 256, "N_mean": 2204.12345678, "m": 2304,
                   "parallelism": "block-farming x1 (no collective)"},
        "stage_ms_per_step": {f"stage{k}": 1.23456789 * k for k in range(12)},
        "telemetry": dict(out["telemetry"], source="sysfs hwmon of 0000:a7:00.0, every 50 ms", samples=47),
    })
    out["roofline"] = {"kernel": "solve_fwd_kernel+solve_bwd_kernel (blocked triangular solves: block-row updates and diagonal blocks, fp64 MFMA 16x16x4)" + "x" * 300,
                       "bound": "mfma", "achieved": 58.712345678, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.7812345, "traffic": 7920000000.123, "traffic_source": "profiles/r06_pmc_traffic.json",
                       "flops_per_launch": 159.2e9, "avg_launch_ms": 2.61234, "launches": 720, "mfma_probe_tflops": 77.1234, "mfma_probe_sustained_tflops": 77.9,
                       "mfma_probe_sustained_telemetry": {"samples": 24, "sclk_mhz": {"min": 2395.0, "median": 2395.0, "max": 2395.0}}}
    out["cpu_baseline"] = {"value": 3.212345678, "unit": "postage-stamps/s", "cores": 256, "kind": "port", "blas": "openblas 64 threads, openblas 64 threads",
                           "sample": "31 whole cfg2 stamps (N~2183, m=2304) through oracle/ on 256 cores (10.2 s): C interpolators with 32 OpenMP threads, scipy potrf/potrs on openblas" + "y" * 400,
                           "stage_ms_per_stamp": {"factor": 31.08, "tri_solve": 19.26, "build": 182.7, "solve": 121.1, "epilogue": 15.8},
                           "one_thread": {"value": 1.5882466977646428, "unit": "postage-stamps/s", "cores": 1, "stage_ms_per_stamp": {"factor": 48.0}},
                           "processes": {"value": 18.590319706435103, "unit": "postage-stamps/s", "processes": 64, "threads_per_process": 1, "cores": 64, "sample": "z" * 300}}
    out["configs"]["iter_default"] = {"value": 612.3456, "ms_per_stamp": 1.633, "roofline": {"frac": 0.5512, "bound": "hbm"}, "block": {"value": 480.0},
                                      "cpu_baseline": {"value": 2.2}, "image_rms_vs_cholesky": 1.2e-3, "cg_steps_mean": 29.6}
    out["farm"] = dict(out["farm"], per_rank_busy_s=[17.123456] * n_gpus, per_rank_wall_s=[30.0] * n_gpus, host_build_s=3.21, n_gpus=n_gpus)
    out["block"].update({"stage_ms_per_block": {f"s{k}": 1.0 * k for k in range(16)}, "workload": "w" * 500, "ms_per_block_all_reps": [1700.0, 1710.0, 1720.0]})
    out["summary"] = bench.summary_of(out)
    return out


def test_the_stdout_line_is_compact_and_parses():
    """the driver could not parse a 22 KB line.  The line bench.py prints LAST on stdout stays below 4 KB with every
    leg present and with strings / lists at their worst, carries the contract's keys, `roofline` (with frac) and `cpu_baseline`, and
    the verbose objects are not in it."""
    import bench

    for n_gpus in (1, 8):
        out = _full_line(n_gpus)
        assert len(json.dumps(out)) > 6000  # (the detail is what used to be printed)
        text = bench.compact_line(out, "bench_detail.json")
        assert len(text) < 4096 and "\n" not in text, len(text)
        line = json.loads(text)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                    "roofline", "cpu_baseline", "summary"):
            assert key in line, key
        assert line["roofline"]["frac"] == 0.78123 and line["roofline"]["bound"] == "mfma" and line["roofline"]["traffic"] > 0
        assert line["cpu_baseline"]["value"] == 3.2123 and line["cpu_baseline"]["cores"] == 256 and line["cpu_baseline"]["kind"] == "port"
        assert line["cpu_baseline"]["one_thread"] == 1.5882 and line["cpu_baseline"]["processes"]["processes"] == 64
        assert line["config"]["workload"].startswith("BASELINE configs[1]") and len(line["config"]["workload"]) <= 200
        assert line["n_gpus"] == n_gpus and line["value"] == 1662.123456 and line["detail"] == "bench_detail.json"
        for verbose in ("configs", "block", "eigen_block", "telemetry", "step_ms", "stage_ms_per_step"):
            assert verbose not in line
        assert "summary_truncated" not in line and line["summary"]["iter_default"]["v"] == 612.3
        if n_gpus > 1:
            assert len(line["per_rank_value"]) == n_gpus and line["farm"]["ranks_seen"] == 8
    # a line that would still be too long loses summary entries from the end, never the contract's keys
    out = _full_line()
    out["summary"] = {f"leg{k}": {"v": 1.0, "note": "n" * 300} for k in range(20)}
    line = json.loads(bench.compact_line(out, None))
    assert line["summary_truncated"] and "roofline" in line and "cpu_baseline" in line and len(json.dumps(line, separators=(",", ":"))) <= bench.COMPACT_LIMIT


def test_detail_goes_to_a_file_and_to_prefixed_stderr_lines(tmp_path, monkeypatch, capsys):
    import bench

    out = _full_line()
    monkeypatch.setenv("IMCOM_BENCH_DETAIL", str(tmp_path / "detail.json"))
    bench.write_detail(out)
    assert json.load(open(tmp_path / "detail.json"))["configs"]["cfg1"]["value"] == 3650.0
    cap = capsys.readouterr()
    assert cap.out == "" and all(ln.startswith("[bench detail] ") for ln in cap.err.splitlines())


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
