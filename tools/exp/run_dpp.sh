timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_inflate.py -x -q -m gpu -k "cigar or block_type or zlib or damaged or crc or tiny" 2>&1 | tail -3
timeout 200 python tools/bench_cigar.py 2>&1 | tail -1
timeout 200 python tools/exp/lz_wave_bench.py 4 2>&1 | tail -7
