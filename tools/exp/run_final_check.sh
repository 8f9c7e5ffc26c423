cd /root/repo
mkdir -p gpurun_out/r05h gpurun_out/r05scan
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/r05h/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r05h/pytest_all.log; tail -4 gpurun_out/r05h/pytest_all.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
REPS=50 timeout 200 python tools/bench_cigar.py > gpurun_out/r05scan/bench_cigar.json 2>/dev/null; cat gpurun_out/r05scan/bench_cigar.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for pair in "6 500" "7 800"; do set -- $pair; rm -rf /tmp/rp_$2; ONLY=$1 REPS=20 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$2 -- python $R/tools/bench_cigar.py > $R/gpurun_out/r05scan/$2.log 2>&1; f=$(find /tmp/rp_$2 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r05scan/cigar_$2_kernel_stats.csv; done
