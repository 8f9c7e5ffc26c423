#!/bin/bash
# Round-5 rocprofv3 evidence, run on the GPU box from the repo root:  bash tools/r05_profile.sh
# kernel-trace/stats and PMC counters are collected in SEPARATE runs (gpurun refuses mixed ones); every tool invocation runs under
# `timeout` and with --output-format csv (a rocprofv3 run that writes its default database did not exit on this pool: 15 GPU-minutes).
# Summaries land in gpurun_out/prof_r05/ and are copied into profiles/ by hand.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/rp_$name; timeout 400 "$@" > $OUT/$name.log 2>&1; }
stats() { f=$(find /tmp/rp_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$2; }
# 1. the bench command itself (file-inclusive headline leg + resident leg)
run bench_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench_trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg
stats bench_trace a_bench_kernel_stats.csv
# 2. eager device stage on the records of real candidate sites, per-kernel time
REAL=1 run stage_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stage_trace -- python $REPO/tools/prof_cnn.py 30
stats stage_trace b_device_stage_kernel_stats.csv
# 3. the inflate kernels alone on the synthetic HiFi-like BAM: 85 k blocks (lane-per-block LZ) and 7 k blocks (wave-per-block LZ)
timeout 300 python $REPO/tools/exp/inflate_gpu_bench.py > $OUT/inflate_gpu_bench.log 2>&1
run inflate_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_inflate_trace -- python $REPO/tools/exp/inflate_once.py fast 3
stats inflate_trace c_inflate_kernel_stats.csv
run inflate_small_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_inflate_small_trace -- python $REPO/tools/exp/inflate_once.py fast-wave 0.25
stats inflate_small_trace c_inflate_small_kernel_stats.csv
pmc() { tag=$1; shift; counters=$1; shift; rm -rf /tmp/rp_pmc_$tag; timeout 400 rocprofv3 --pmc $counters --output-format csv -d /tmp/rp_pmc_$tag -- "$@" > $OUT/pmc_$tag.log 2>&1; python $REPO/tools/pmc_summary.py /tmp/rp_pmc_$tag > $OUT/pmc_$tag.txt; }
pmc fetch_inflate "FETCH_SIZE" python $REPO/tools/exp/inflate_once.py fast 3
pmc write_inflate "WRITE_SIZE" python $REPO/tools/exp/inflate_once.py fast 3
pmc fetch_inflate_small "FETCH_SIZE" python $REPO/tools/exp/inflate_once.py fast-wave 0.25
pmc write_inflate_small "WRITE_SIZE" python $REPO/tools/exp/inflate_once.py fast-wave 0.25
pmc sq_inflate "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" python $REPO/tools/exp/inflate_once.py fast 3
# 4. PMC of the device stage (roofline.traffic of the bench line)
export REAL=1
pmc sq_prof_cnn "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32" python $REPO/tools/prof_cnn.py 4
pmc grbm_prof_cnn "GRBM_GUI_ACTIVE" python $REPO/tools/prof_cnn.py 4
pmc fetch_prof_cnn "FETCH_SIZE" python $REPO/tools/prof_cnn.py 4
pmc write_prof_cnn "WRITE_SIZE" python $REPO/tools/prof_cnn.py 4
python $REPO/tools/kstats.py $OUT/b_device_stage_kernel_stats.csv > $OUT/b_device_stage_kernel_stats.txt
# 5. the file-inclusive leg kernel by kernel
run e2e_trace rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_e2e_trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg
timeout 120 python $REPO/tools/e2e_kernel_timeline.py /tmp/rp_e2e_trace 7 > $OUT/e2e_kernel_timeline.txt
# 6. the scan: tools/r05_profile_scan.sh (its own script: run after the scan's last kernel commit); the rasteriser export
run raster_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_raster_trace -- python $REPO/tools/microbench.py
stats raster_trace raster_microbench_kernel_stats.csv
# 7. what the section 8(f)2 / 8(f)3 kernels would cost or save inside -t N (the decision of DESIGN.md: retired there)
timeout 300 python $REPO/tools/f2f3_cost.py > $OUT/f2f3_cost.json 2> $OUT/f2f3_cost.err
ls -la $OUT
