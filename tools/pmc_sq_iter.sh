#!/bin/bash
# Matrix-pipe utilisation, wait shares and LDS conflicts of the blocked CG (iter_default, batch 64): two --pmc passes over tools/bench_iter.py,
# reduced by tools/pmc_sq_summary.py (GPU box, repo root): [IMCOM_ITER_SYM=0] tools/pmc_sq_iter.sh LABEL
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=${1:-iter}
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/pmc_sq_iter; rm -rf $O/sq $O/sq2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/sq -- python3 $ROOT/tools/bench_iter.py 64 1 0 > $O/run.log 2>&1 || { echo "pass failed"; tail -3 $O/run.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM -d $O/sq2 -- python3 $ROOT/tools/bench_iter.py 64 1 0 > $O/run2.log 2>&1 || { echo "pass 2 failed"; tail -3 $O/run2.log; }
cd $ROOT
S=$(find $O/sq -name '*.db' | head -1)
python tools/pmc_sq_summary.py "$S" $O/pmc_sq_iter_$L.txt "iter_default Iterative kernel, batch 64: python3 tools/bench_iter.py 64 1 0 ($L)"
S2=$(find $O/sq2 -name '*.db' | head -1)
if [ -n "$S2" ]; then python3 - "$S2" >> $O/pmc_sq_iter_$L.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, sum(counter_value) from pmc_events group by name, counter_name").fetchall()
d = {}
for n, c, v in rows:
    if "iter_block_cg" not in n: continue
    d.setdefault(n.split("(")[0].replace("void ", "").replace("imcom::", ""), {})[c] = v
for k, v in d.items():
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"# pass 2 {k}: " + " ".join(f"{c}/WAVE_CYCLES={x / wc:.3f}" for c, x in sorted(v.items()) if c != "SQ_WAVE_CYCLES"))
PY
fi
rm -rf $O/sq $O/sq2
cat $O/pmc_sq_iter_$L.txt
