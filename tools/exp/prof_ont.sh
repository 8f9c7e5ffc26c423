R=$(pwd); cd /tmp && export TMPDIR=/tmp
for only in 3 2; do rm -rf /tmp/rp_$only
ONLY=$only REPS=20 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$only -- python $R/tools/bench_cigar.py > /dev/null 2>&1
f=$(find /tmp/rp_$only -name "*kernel_stats.csv" | head -1); echo "== size $only"; python - <<PY
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']; k='count' if 'count_kernel' in n else 'emit' if 'emit' in n else 'offsets' if 'offsets' in n else 'frames' if 'frames' in n else n[:20]
    print(k, r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))
PY
done
