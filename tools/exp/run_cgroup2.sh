cd /root/repo
mkdir -p gpurun_out/r05g
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cigar or scan" 2>&1 | tail -3
for lib in libsvx va_libsvx vb_libsvx vc_libsvx vd_libsvx ve_libsvx vf_libsvx vg_libsvx $EXTRA_LIBS; do
  [ -f svision_amd/$lib.so ] || continue
  for i in 1 2; do
    echo "$lib run $i: $(SVX_EXP_LIB=$(pwd)/svision_amd/$lib.so REPS=50 timeout 120 python tools/bench_cigar.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' | '.join('%s %.1f us' % (k, v['us']) for k, v in d.items()))")"
  done
done 2>&1 | tee gpurun_out/r05g/cgroup_${TAG:-b}.txt
