python3 -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider -k "chain_nan" 2>&1 | tail -30
