# timing experiments on the skinny solve: kernel stats of the hinted repair on 168 paper4 stamps (results of skipped sweeps are wrong: timing only)
O=gpurun_out/s2; mkdir -p $O
for v in "$@"; do
  export IMCOM_SK_SKIP=$v
  IMCOM_LMIN_HINTED=4 IMCOM_LMIN_FINE=1 bash tools/prof_any.sh skexp$v tools/bench_repair_hinted.py 168 1 2>&1 | grep -E "skinny" | sed "s/^/skip=$v /"
done
