cd /root/repo
for lib in libsvx q3_libsvx q4_libsvx libsvx; do
  echo "$lib: $(SVX_EXP_LIB=$(pwd)/svision_amd/$lib.so ONLY=5,6,7,2,3 REPS=50 timeout 120 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f' % (k, v['us']) for k, v in d.items()))")"
done
