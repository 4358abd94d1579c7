set -e
O=gpurun_out/s2; mkdir -p $O
B=${1:-24}
IMCOM_LMIN_SKINNY=1 IMCOM_LMIN_DEBUG=1 timeout -k 10 400 python tools/bench_repair_hinted.py $B 2 check > $O/skinny1.json 2> $O/skinny1.err || { tail -30 $O/skinny1.err; exit 1; }
IMCOM_LMIN_SKINNY=0 IMCOM_LMIN_DEBUG=1 timeout -k 10 400 python tools/bench_repair_hinted.py $B 2 > $O/skinny0.json 2> $O/skinny0.err || { tail -30 $O/skinny0.err; exit 1; }
cat $O/skinny1.json $O/skinny0.json
grep -E "^\[lmin\] (round|[0-9]+ stamps)" $O/skinny1.err | tail -12
