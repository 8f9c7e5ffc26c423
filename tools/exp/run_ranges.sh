cd /root/repo
for sh in 0 11 12 13 11; do
  echo "shift $sh: $(SVX_RANGE_SHIFT=$sh ONLY=3,2 REPS=50 timeout 120 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f' % (k, v['us']) for k, v in d.items()))")"
done
