# session check of the partial-tile Cholesky and the chained block solves (run on the GPU box from the repo root)
set -e
O=gpurun_out/s2
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_stamps.py tests/test_gpu_kernels_block.py tests/test_gpu_paper4.py -x -q -k "resident or ragged or repair or seam or chol or kappa or golden" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python tools/bench_seam_paper4.py 4 > $O/seam_chain.json 2> $O/seam_chain.err || { tail -20 $O/seam_chain.err; exit 1; }
IMCOM_LMIN_CHAIN=0 timeout -k 10 200 python tools/bench_seam_paper4.py 4 > $O/seam_right.json 2> $O/seam_right.err || { tail -20 $O/seam_right.err; exit 1; }
head -1 $O/seam_chain.json; head -1 $O/seam_right.json
IMCOM_BENCH_DETAIL=$O/headline_detail.json timeout -k 10 300 python bench.py --no-cpu-baseline --no-block --no-configs --steps 10 --warmup 3 > $O/headline.json 2> $O/headline.err || { tail -20 $O/headline.err; exit 1; }
python -c "import json;d=json.loads(open('$O/headline.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['summary']['headline'])"
python -c "import json;d=json.load(open('$O/headline_detail.json'));print({k:round(v,2) for k,v in d.get('stage_ms_per_step',{}).items()})" || true
