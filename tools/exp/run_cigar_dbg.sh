for g in 768 1536 512 3072; do echo grid $g; SVX_FLAT_GRID=$g SVX_SCAN_MODE=flat ONLY=3 timeout 60 python tools/bench_cigar.py 2>&1 | tail -1; done
