R=$(pwd)
mkdir -p $R/gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp
for only in 3 2 4 0; do
rm -rf /tmp/rp_$only
ONLY=$only REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$only -- python $R/tools/bench_cigar.py > $R/gpurun_out/r05g/p$only.log 2>&1
f=$(find /tmp/rp_$only -name "*kernel_stats.csv" | head -1); echo "== size $only"; head -6 $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/(unsigned int const\*.*)"/"/' | cut -c1-150; cp $f $R/gpurun_out/r05g/size${only}_kernel_stats.csv
done
cd $R
for lds in 0 24000 32768 40960; do
  echo "lds $lds: $(SVX_COUNT_LDS=$lds ONLY=0,4,5 REPS=50 timeout 200 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f' % (k, v['us']) for k, v in d.items()))")"
done
