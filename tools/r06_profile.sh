#!/bin/bash
# Round-6 rocprofv3 evidence, run on the GPU box from the repo root:  bash tools/r06_profile.sh
# kernel-trace/stats and PMC counters are collected in SEPARATE runs (gpurun refuses mixed ones); every tool invocation runs under
# `timeout` and with --output-format csv.  Summaries land in gpurun_out/prof_r06/ and are copied into profiles/ by hand.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/rp_$name; timeout 400 "$@" > $OUT/$name.log 2>&1; }
stats() { f=$(find /tmp/rp_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$2; }
pmc() { tag=$1; shift; counters=$1; shift; rm -rf /tmp/rp_pmc_$tag; timeout 400 rocprofv3 --pmc $counters --output-format csv -d /tmp/rp_pmc_$tag -- "$@" > $OUT/pmc_$tag.log 2>&1; python $REPO/tools/pmc_summary.py /tmp/rp_pmc_$tag > $OUT/pmc_$tag.txt; }
# 1. the bench command itself (file-inclusive headline leg x3 + resident leg x3 + parity leg)
run bench_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench_trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg --detail $OUT/bench_trace.detail.json
stats bench_trace a_bench_kernel_stats.csv
# 2. eager device stage on the records of real candidate sites, per-kernel time
REAL=1 run stage_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stage_trace -- python $REPO/tools/prof_cnn.py 30
stats stage_trace b_device_stage_kernel_stats.csv
python $REPO/tools/kstats.py $OUT/b_device_stage_kernel_stats.csv > $OUT/b_device_stage_kernel_stats.txt
# 3. PMC of the device stage (roofline.traffic of the bench line) and its matrix-pipe utilisation
export REAL=1
pmc sq_prof_cnn "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32" python $REPO/tools/prof_cnn.py 4
pmc grbm_prof_cnn "GRBM_GUI_ACTIVE" python $REPO/tools/prof_cnn.py 4
pmc fetch_prof_cnn "FETCH_SIZE" python $REPO/tools/prof_cnn.py 4
pmc write_prof_cnn "WRITE_SIZE" python $REPO/tools/prof_cnn.py 4
unset REAL
# 4. the rasteriser (round 6: branch-free groups, one resident workgroup per CU): per-kernel time and what it writes
run raster_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_raster_trace -- $REPO/tools/exp/raster_bw 2048 11
stats raster_trace raster_bw_kernel_stats.csv
pmc write_raster "WRITE_SIZE" python $REPO/tools/microbench.py
pmc fetch_raster "FETCH_SIZE" python $REPO/tools/microbench.py
timeout 120 $REPO/tools/exp/raster_bw 2048 21 > $OUT/raster_bw.txt 2>&1
# 5. the file-inclusive leg kernel by kernel
run e2e_trace rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_e2e_trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg --detail $OUT/e2e_trace.detail.json
timeout 120 python $REPO/tools/e2e_kernel_timeline.py /tmp/rp_e2e_trace 7 > $OUT/e2e_kernel_timeline.txt
ls -la $OUT
