set -e
O=gpurun_out/s2
mkdir -p $O
timeout -k 10 200 python tools/bench_seam_paper4.py 4 > $O/seam_chain.json 2> $O/seam_chain.err || { tail -20 $O/seam_chain.err; exit 1; }
head -1 $O/seam_chain.json
timeout -k 10 300 python -m pytest tests/test_gpu_paper4.py tests/test_gpu_kernels_block.py -x -q -k "seam or repair" > $O/tests_seam.log 2>&1 || { tail -30 $O/tests_seam.log; exit 1; }
tail -2 $O/tests_seam.log
