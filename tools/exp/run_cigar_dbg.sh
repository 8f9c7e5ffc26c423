timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cigar and flat" 2>&1 | tail -3
for g in 768 1536; do echo grid $g; SVX_FLAT_GRID=$g SVX_SCAN_MODE=flat ONLY=3 timeout 60 python tools/bench_cigar.py 2>&1 | tail -1; done
SVX_SCAN_MODE=flat ONLY=2 timeout 60 python tools/bench_cigar.py 2>&1 | tail -1
