#!/bin/bash
# the whole-genome job (322 windows, ~62 GB of BAM) file-inclusive on one MI355X.  The box's disk (overlay, 79 GB)
# holds the file; the compressed segments wait there too (bench.py: spill_dir) and six processes simulate + compress (the cgroup
# allows 300 GiB of memory: a process holds a chromosome's records inflated and compressed).  `value` reads the file from the page
# cache it was written through, as every default run does.
out=gpurun_out/r06_wg_file; mkdir -p $out
( echo "memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"; df -h /tmp ) > $out/box.txt 2>&1
( while sleep 20; do echo "$(date +%T) mem $(cat /sys/fs/cgroup/memory.current 2>/dev/null) disk $(df --output=used /tmp | tail -1)"; done ) >> $out/box.txt 2>&1 &
mon=$!
timeout 2700 python bench.py --gpus 1 --steps 322 --e2e-windows 322 --warmup 5 --bam-dir /tmp --sim-procs 6 --no-cold-leg --no-other-engine --no-calibration \
    --detail $out/wg_file.detail.json > $out/wg_file.json 2> $out/wg_file.err
echo "rc=$?"; kill $mon; cat $out/wg_file.json; tail -c 1500 $out/wg_file.err
