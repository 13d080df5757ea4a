timeout 2400 python3 tools/soak.py 10000 2000 test_fuzz_chain_nan_sets test_fuzz_nonfinite_sets > gpurun_out/soak_nf.log 2>&1; tail -4 gpurun_out/soak_nf.log
timeout 2400 python3 tools/soak.py 20000 500 > gpurun_out/soak_all.log 2>&1; tail -3 gpurun_out/soak_all.log
bash tools/abort_hunt.sh 12
