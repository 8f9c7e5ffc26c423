# round 5, final form of svx_cigar_scan: kernel stats of the HiFi-sized, the ONT-shaped and the 50 k x 6,000 launch + the bench line
R=$(pwd)
mkdir -p $R/gpurun_out/r05scan
cd /tmp && export TMPDIR=/tmp
for pair in "4 hifi" "3 ont" "2 6000"; do
set -- $pair
rm -rf /tmp/rp_$2
ONLY=$1 REPS=20 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$2 -- python $R/tools/bench_cigar.py > $R/gpurun_out/r05scan/$2.log 2>&1
f=$(find /tmp/rp_$2 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r05scan/cigar_$2_kernel_stats.csv; head -7 $f | cut -c1-60,200-
done
cd $R
REPS=50 timeout 200 python tools/bench_cigar.py > gpurun_out/r05scan/bench_cigar.json 2>/dev/null; cat gpurun_out/r05scan/bench_cigar.json
SVX_SCAN_MODE=flat REPS=50 ONLY=3,2 timeout 200 python tools/bench_cigar.py > gpurun_out/r05scan/bench_cigar_flat.json 2>/dev/null; cat gpurun_out/r05scan/bench_cigar_flat.json
# PMC passes (separate runs; rocprofv3 --pmc without trace flags): bytes fetched / written and vector instructions per launch
cd /tmp
pmc() { tag=$1; shift; counters=$1; shift; rm -rf /tmp/rp_pmc_$tag; timeout 200 rocprofv3 --pmc $counters --output-format csv -d /tmp/rp_pmc_$tag -- "$@" > $R/gpurun_out/r05scan/pmc_$tag.log 2>&1; python $R/tools/pmc_summary.py /tmp/rp_pmc_$tag > $R/gpurun_out/r05scan/pmc_$tag.txt; }
ONLY=4 REPS=5 pmc cigar_hifi_fetch "FETCH_SIZE" python $R/tools/bench_cigar.py
ONLY=4 REPS=5 pmc cigar_hifi_write "WRITE_SIZE" python $R/tools/bench_cigar.py
ONLY=4 REPS=5 pmc cigar_hifi_sq "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" python $R/tools/bench_cigar.py
ONLY=3 REPS=5 pmc cigar_ont_fetch "FETCH_SIZE" python $R/tools/bench_cigar.py
ONLY=3 REPS=5 pmc cigar_ont_write "WRITE_SIZE" python $R/tools/bench_cigar.py
ONLY=3 REPS=5 pmc cigar_ont_sq "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" python $R/tools/bench_cigar.py
cd $R; tail -n 12 gpurun_out/r05scan/pmc_cigar_*_fetch.txt gpurun_out/r05scan/pmc_cigar_*_write.txt
