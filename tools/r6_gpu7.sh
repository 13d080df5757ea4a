python3 -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
