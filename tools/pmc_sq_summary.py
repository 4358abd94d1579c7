"""Busy / stall summary of a rocprofv3 --pmc pass with SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE.   python pmc_sq_summary.py RESULTS.db OUT.txt "command" """
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, count(*), count(distinct dispatch_id), sum(counter_value), sum(duration) from pmc_events "
                   "group by name, counter_name").fetchall()
d = {}
for n, c, k, kd, v, t in rows:
    if "imcom::" not in n:
        continue
    sn = n.split("(")[0].replace("void ", "").replace("imcom::", "")
    e = d.setdefault(sn, {})
    e[c], e["launches"], e["ms"] = v, kd, t / 1e6 * kd / k
out = [f"# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
       f"SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- {sys.argv[3]}",
       "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) (1.0 = every SIMD's matrix pipe busy every cycle); "
       "wait_any / issue_stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES"]
for sn, v in sorted(d.items(), key=lambda kv: -kv[1]["ms"]):
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128 * max(v.get("GRBM_GUI_ACTIVE", 1), 1))
    out.append(f"{sn:28s} launches={v['launches']:4d} ms={v['ms']:8.2f} mfma_busy={mf:5.3f} wait_any={v.get('SQ_WAIT_ANY', 0) / wc:5.3f} "
               f"issue_stall={v.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f} active={v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f}")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[2:8]))
