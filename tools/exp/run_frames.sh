cd /root/repo
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cigar or scan" 2>&1 | tail -8
for mode in "" groups8 groups8s groups4 groups4s; do
  for i in 1 2; do
    echo "mode '$mode' run $i: $(SVX_SCAN_MODE=$mode REPS=50 timeout 200 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f' % (k, v['us']) for k, v in d.items()))")"
  done
done 2>&1 | tee gpurun_out/r05g/frames_${TAG:-a}.txt
