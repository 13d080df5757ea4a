set -x
python3 -m pytest tests/test_gpu_edges_fullsize.py tests/test_gpu_bench_line.py -q -x -p no:cacheprovider > gpurun_out/r6_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t1.log
python3 bench.py > gpurun_out/bench_r06_a.json 2> gpurun_out/bench_r06_a.err; echo "bench rc=$?"
tail -c 4500 gpurun_out/bench_r06_a.json
timeout 600 python3 tools/pageable_churn.py 3000 > gpurun_out/churn_on.log 2>&1; echo "churn_on rc=$?"
RR_LIB_PATH=$PWD/rustradio_amd/lib_stage_off/librustradio_amd.so timeout 600 python3 tools/pageable_churn.py 3000 > gpurun_out/churn_off.log 2>&1; echo "churn_off rc=$?"
tail -3 gpurun_out/churn_on.log gpurun_out/churn_off.log
tail -15 gpurun_out/r6_t1.log
