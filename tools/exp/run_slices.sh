timeout 900 python -m pytest tests/test_slices_gpu.py -x -q -m gpu 2>&1 | tail -15
