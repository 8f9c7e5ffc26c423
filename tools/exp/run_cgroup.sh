# count pass variants (lanes per alignment x quads in flight): tools/bench_cigar.py per library
cd /root/repo
mkdir -p gpurun_out/r05g
for lib in libsvx cg_g4q2_libsvx cg_g4q3_libsvx cg_g4q4_libsvx cg_g8q3_libsvx $EXTRA_LIBS; do
  [ -f svision_amd/$lib.so ] || continue
  for i in 1 2; do
    echo "$lib run $i: $(SVX_EXP_LIB=$(pwd)/svision_amd/$lib.so REPS=50 timeout 120 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f us' % (k, v['us']) for k, v in d.items()))")"
  done
done 2>&1 | tee gpurun_out/r05g/cgroup_${TAG:-a}.txt
