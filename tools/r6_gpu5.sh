python3 -m pytest tests/test_gpu_edges_fullsize.py -q -p no:cacheprovider -k "nan_sets or fused_fm or audio" 2>&1 | tail -30
