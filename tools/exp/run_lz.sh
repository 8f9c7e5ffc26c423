mkdir -p gpurun_out/r05b && python tools/exp/lz_wave_bench.py 4 > gpurun_out/r05b/lz_wave_bench.txt 2>&1; tail -20 gpurun_out/r05b/lz_wave_bench.txt
timeout 900 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu -k "block_type or zlib or damaged or crc or tiny or stream_of_its_own" 2>&1 | tail -15
