#!/bin/bash
# Round-4 rocprofv3 evidence, run on the GPU box from the repo root:  bash tools/r04_profile.sh
# kernel-trace/stats and PMC counters are collected in SEPARATE runs (gpurun refuses mixed ones); summaries land in
# gpurun_out/prof_r04/ and are copied into profiles/ by hand.  Every tool invocation runs under `timeout`.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/rp_$name; timeout 400 "$@" > $OUT/$name.log 2>&1; }
# 1. the bench command itself (file-inclusive headline leg + resident leg; ingest kernels: bgzf_tokens_kernel, bgzf_lz_kernel, bgzf_crc32_kernel, bam_walk_*)
run bench_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench_trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration
cp $(find /tmp/rp_bench_trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# 2. eager device stage on the records of real candidate sites, per-kernel time
REAL=1 run stage_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stage_trace -- python $REPO/tools/prof_cnn.py 30
cp $(find /tmp/rp_stage_trace -name "*kernel_stats.csv" | head -1) $OUT/stage_kernel_stats.csv
# 3. the two-kernel inflate alone: 85 k blocks (the synthetic HiFi-like BAM of tools/exp/inflate_gpu_bench.py, three times over), CRC included
timeout 300 python $REPO/tools/exp/inflate_gpu_bench.py > $OUT/inflate_gpu_bench.log 2>&1
run inflate_trace rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_inflate_trace -- python $REPO/tools/exp/inflate_once.py fast 3
cp $(find /tmp/rp_inflate_trace -name "*kernel_stats.csv" | head -1) $OUT/inflate_kernel_stats.csv
pmc() { tag=$1; shift; counters=$1; shift; rm -rf /tmp/rp_pmc_$tag; timeout 400 rocprofv3 --pmc $counters --output-format csv -d /tmp/rp_pmc_$tag -- "$@" > $OUT/pmc_$tag.log 2>&1; python $REPO/tools/pmc_summary.py /tmp/rp_pmc_$tag > $OUT/pmc_$tag.txt; }
pmc sq_inflate "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" python $REPO/tools/exp/inflate_once.py fast 3
pmc sq2_inflate "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE" python $REPO/tools/exp/inflate_once.py fast 3
pmc fetch_inflate "FETCH_SIZE" python $REPO/tools/exp/inflate_once.py fast 3
pmc write_inflate "WRITE_SIZE" python $REPO/tools/exp/inflate_once.py fast 3
# 4. PMC of the device stage (kernels unchanged since round 2: re-measured for this round's roofline.traffic)
export REAL=1
pmc sq_prof_cnn "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32" python $REPO/tools/prof_cnn.py 4
pmc grbm_prof_cnn "GRBM_GUI_ACTIVE" python $REPO/tools/prof_cnn.py 4
pmc fetch_prof_cnn "FETCH_SIZE" python $REPO/tools/prof_cnn.py 4
pmc write_prof_cnn "WRITE_SIZE" python $REPO/tools/prof_cnn.py 4
python $REPO/tools/kstats.py $OUT/stage_kernel_stats.csv > $OUT/stage_kernel_stats.txt
# 5. the file-inclusive leg kernel by kernel (profiles/r04_e2e_timeline.md, third part)
run e2e_trace rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_e2e_trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine
python $REPO/tools/e2e_kernel_timeline.py /tmp/rp_e2e_trace 7 > $OUT/e2e_kernel_timeline.txt
# 6. the scan alone and the experiments behind section 5 of DESIGN.md ("Round 4, second half")
timeout 300 python $REPO/tools/bench_cigar.py > $OUT/bench_cigar.json 2> /dev/null
for p in -1 0 1; do timeout 200 python $REPO/tools/exp/queue_map.py $p 6; done > $OUT/queue_map.txt 2>&1
timeout 300 python $REPO/tools/exp/lz_corun.py $REPO/svision_amd/libsvx.so > $OUT/lz_corun.txt 2>&1
ls -la $OUT
