cd /root/repo
mkdir -p gpurun_out/r05f
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cigar" > gpurun_out/r05f/pytest_cigar.log 2>&1; tail -15 gpurun_out/r05f/pytest_cigar.log
python tools/bench_cigar.py > gpurun_out/r05f/bench_cigar_auto.json 2> gpurun_out/r05f/bench_cigar.err; cat gpurun_out/r05f/bench_cigar_auto.json
SVX_SCAN_MODE=groups python tools/bench_cigar.py > gpurun_out/r05f/bench_cigar_groups.json 2>> gpurun_out/r05f/bench_cigar.err; cat gpurun_out/r05f/bench_cigar_groups.json
SVX_SCAN_MODE=flat python tools/bench_cigar.py > gpurun_out/r05f/bench_cigar_flat.json 2>> gpurun_out/r05f/bench_cigar.err; cat gpurun_out/r05f/bench_cigar_flat.json
SVX_EXP_LIB=$(pwd)/svision_amd/fq16_libsvx.so SVX_SCAN_MODE=flat python tools/bench_cigar.py > gpurun_out/r05f/bench_cigar_flat16.json 2>> gpurun_out/r05f/bench_cigar.err; cat gpurun_out/r05f/bench_cigar_flat16.json
