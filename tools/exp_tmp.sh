timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
