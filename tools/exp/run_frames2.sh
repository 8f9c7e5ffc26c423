cd /root/repo
mkdir -p gpurun_out/r05g
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cigar or scan" --timeout 60 2>&1 | tail -4
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for only in 3 4; do
rm -rf /tmp/rp_$only
ONLY=$only REPS=20 timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$only -- python $R/tools/bench_cigar.py > $R/gpurun_out/r05g/p$only.log 2>&1
f=$(find /tmp/rp_$only -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r05g/size${only}_kernel_stats.csv
echo "== size $only"; python - <<PY
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']; k='count' if 'count_kernel' in n else 'emit' if 'emit' in n else 'offsets' if 'offsets' in n else n[:20]
    print(k, r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))
PY
done
cd $R
for mode in "" groups8; do
    echo "mode '$mode': $(SVX_SCAN_MODE=$mode REPS=50 timeout 120 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f' % (k, v['us']) for k, v in d.items()))")"
done 2>&1 | tee gpurun_out/r05g/frames_${TAG:-b}.txt
