cd /root/repo
mkdir -p gpurun_out/r05g
for lib in $LIBS; do
  [ -f svision_amd/$lib.so ] || continue
  for lds in 0 32768 40960 54000 81920; do
    echo "$lib lds $lds: $(SVX_COUNT_LDS=$lds ONLY=${ONLY:-0,4,3} SVX_EXP_LIB=$(pwd)/svision_amd/$lib.so REPS=50 timeout 120 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f us' % (k, v['us']) for k, v in d.items()))")"
  done
done 2>&1 | tee gpurun_out/r05g/cgroup_${TAG:-d}.txt
