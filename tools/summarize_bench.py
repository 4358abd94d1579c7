"""Markdown summary of a bench.py JSON line (the table of DESIGN.md "Round 4"):  python tools/summarize_bench.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
st = d["stage_ms_per_step"]
print(f"* headline (BASELINE configs[1], 256 stamps per step, {d['steps']} steps): **{d['value']:.0f} stamps/s** ({d['ms_per_step']:.1f} ms per step, "
      f"{d['ms_per_stamp']:.3f} ms per stamp); solve family {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}** of {r['peak']} "
      f"(avg launch {r['avg_launch_ms']:.3f} ms, {r['launches']} launches; pure-MFMA probe {r['mfma_probe_tflops']:.1f}); traffic {r['traffic']} ({r.get('traffic_source')})")
print("  stages per step: " + ", ".join(f"{k} {v:.2f}" for k, v in st.items() if v > 0.005) + " ms")
if "block" in d:
    b = d["block"]
    print(f"* block leg (one 48 x 48 block, {b['psf_groups']} PSF groups, {b['tables']['computed']} tables computed, arena {b['tables']['arena']}): "
          f"**{b['value']:.0f} stamps/s** ({b['ms_per_block']:.0f} ms per block, median of {b.get('reps', 1)}: {[round(x) for x in b.get('ms_per_block_all_reps', [])]}); "
          + ", ".join(f"{k} {v:.0f}" for k, v in b["stage_ms_per_block"].items()) + f"; host + gaps {b['host_and_gaps_ms_per_block']:.0f} ms")
for k in ("kernel_seam", "block_seam"):
    if k in d:
        x = {kk: vv for kk, vv in d[k].items() if not kk.startswith("what")}
        print(f"* {k}: {json.dumps(x)}")
if "configs" in d:
    print("\n| configuration | batch | N mean | stamps/s | ms per stamp | roofline (kernel: achieved / peak = frac) | job frac |")
    print("|---|---|---|---|---|---|---|")
    for name in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        c = d["configs"].get(name)
        if not c:
            continue
        legs = [(k, v) for k, v in c.items() if isinstance(v, dict) and k.startswith("b") and k[1:].isdigit()] or [("", c)]
        for tag, leg in legs:
            ro = leg.get("roofline", {})
            extra = ""
            if "roofline_hbm" in leg:
                h = leg["roofline_hbm"]
                extra = f"; symv4 {h['achieved']:.0f} GB/s = {h['frac']:.2f} of HBM ({h['ms_per_step']:.0f} ms per step)"
            print(f"| {name} ({c.get('config', '')}) | {leg.get('batch', d['config']['stamps_per_step_per_gpu'])} | {leg.get('N_mean', d['config']['N_mean']):.0f} | "
                  f"{leg['value']:.0f} | {leg['ms_per_stamp']:.3f} | {ro.get('kernel', '')[:40]}: {ro.get('achieved', 0):.1f} / {ro.get('peak', 0)} = {ro.get('frac', 0):.3f}{extra} | "
                  f"{leg.get('job_roofline_frac', float('nan')):.2f} |")
if "eigen_block" in d:
    e = d["eigen_block"]
    print(f"\n* eigen_block ({e['workload'][:60]}...): **{e['value']:.0f} stamps/s** = {e['ms_per_stamp']:.2f} ms per stamp, batches {e['batches']}; "
          + ", ".join(f"{k} {v:.0f}" for k, v in e["stage_ms_per_block"].items()))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"* cpu_baseline ({c['kind']}, {c['cores']} cores): {c['value']:.2f} stamps/s; one thread {c.get('one_thread', {}).get('value', float('nan')):.2f}; "
          f"{c.get('processes', {}).get('processes')} single-threaded processes {c.get('processes', {}).get('value', float('nan')):.1f} stamps/s")
