# the bench line as the driver takes it (default flags), with its detail file and the wall time of the run; then the multi-rank rehearsal
# (two ranks sharing the one GPU) and the farm leg on one GPU.  GPU box, repo root:  bash tools/final_bench.sh LABEL
L=${1:-final}; O=gpurun_out/$L; mkdir -p $O
T0=$(date +%s)
IMCOM_BENCH_DETAIL=$O/bench_detail.json python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
T1=$(date +%s); echo "default bench.py: $((T1 - T0)) s wall" | tee $O/wall.txt
tail -1 $O/bench.json | wc -c
IMCOM_BENCH_DETAIL=$O/farm1_detail.json python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-block --no-configs --farm > $O/farm_1gpu.json 2> $O/farm_1gpu.err || { tail -20 $O/farm_1gpu.err; exit 1; }
IMCOM_BENCH_DETAIL=$O/rehearsal_detail.json python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-shared-gpu > $O/rehearsal_2ranks.json 2> $O/rehearsal.err || { tail -20 $O/rehearsal.err; exit 1; }
tail -1 $O/farm_1gpu.json | cut -c1-300; tail -1 $O/rehearsal_2ranks.json | cut -c1-300
