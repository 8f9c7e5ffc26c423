#!/bin/bash
# Round-3 rocprofv3 evidence, run on the GPU box from the repo root:  bash tools/r03_profile.sh
# kernel-trace/stats and PMC counters are collected in SEPARATE runs (gpurun refuses mixed ones); summaries land in
# gpurun_out/prof_r03/ and are copied into profiles/ by hand.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/rp_$name; timeout 600 "$@" > $OUT/$name.log 2>&1; }     # (a hung tool must not hold the box: gpurun counts that as a strike)
# 1. eager device stage on the records of real candidate sites, per-kernel time
REAL=1 run stage_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stage_trace -- python $REPO/tools/prof_cnn.py 30
cp $(find /tmp/rp_stage_trace -name "*kernel_stats.csv" | head -1) $OUT/stage_kernel_stats.csv
# 2. the bench command itself (graph replays of 256 images on 3 streams; its e2e legs hold the ingest kernels: svx_bgzf_inflate*, svx_bam_walk_*)
run bench_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench_trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-calibration
cp $(find /tmp/rp_bench_trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# 3. PMC: matrix-pipe utilisation and wave stall breakdown (SQ, <= 8 counters), clock (GRBM), HBM traffic (TCC; separate passes)
for tgt in "REAL=1 prof_cnn.py 4" "X=1 prof_conv.py"; do
  set -- $tgt; envv=$1; script=$2; arg=${3:-}
  tag=${script%.py}
  env $envv timeout 600 bash -c "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 --output-format csv -d /tmp/rp_pmc_sq_$tag -- python $REPO/tools/$script $arg" > $OUT/pmc_sq_$tag.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/rp_pmc_sq_$tag > $OUT/pmc_sq_$tag.txt
  env $envv timeout 600 bash -c "rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/rp_pmc_grbm_$tag -- python $REPO/tools/$script $arg" > $OUT/pmc_grbm_$tag.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/rp_pmc_grbm_$tag > $OUT/pmc_grbm_$tag.txt
  env $envv timeout 600 bash -c "rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_pmc_fetch_$tag -- python $REPO/tools/$script $arg" > $OUT/pmc_fetch_$tag.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/rp_pmc_fetch_$tag > $OUT/pmc_fetch_$tag.txt
  env $envv timeout 600 bash -c "rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_pmc_write_$tag -- python $REPO/tools/$script $arg" > $OUT/pmc_write_$tag.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/rp_pmc_write_$tag > $OUT/pmc_write_$tag.txt
done
# per-kernel durations of the same eager runs (for MFMA-busy / duration and clock = GRBM cycles / duration)
python $REPO/tools/kstats.py $OUT/stage_kernel_stats.csv > $OUT/stage_kernel_stats.txt
ls -la $OUT
