#!/bin/bash
# the file-inclusive and the resident rate of the 20-window job with the rank pinned to K host CPUs.
out=gpurun_out/r06_hostcpus; mkdir -p $out
df -h /tmp /dev/shm . > $out/box.txt 2>&1; nproc >> $out/box.txt; cat /sys/fs/cgroup/cpu.max >> $out/box.txt 2>&1; free -g >> $out/box.txt
for K in ${KS:-16 8 4 2}; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --host-cpus $K --no-cpu-baseline --no-calibration --no-cold-leg --no-other-engine \
      --detail $out/k$K.detail.json > $out/k$K.json 2> $out/k$K.err
  echo "K=$K rc=$? $(cut -c1-400 $out/k$K.json)"
done
