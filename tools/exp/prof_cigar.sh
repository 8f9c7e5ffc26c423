R=$(pwd)
mkdir -p $R/gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp
for m in groups flat; do
rm -rf /tmp/rp_$m
SVX_SCAN_MODE=$m ONLY=3 REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$m -- python $R/tools/bench_cigar.py > $R/gpurun_out/r05g/$m.log 2>&1
f=$(find /tmp/rp_$m -name "*kernel_stats.csv" | head -1); echo "== $m"; head -8 $f | cut -d, -f1-8; cp $f $R/gpurun_out/r05g/ont_${m}_kernel_stats.csv
done
