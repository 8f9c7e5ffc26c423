cd /root/repo
for i in 1 2; do
for lib in libsvx seg256_libsvx; do
python tools/exp/run_bench_with_lib.py svision_amd/$lib.so --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['e2e']['seconds'],3), 'res', round(d['config']['resident_sites_per_s']))"
done; done
for lib in libsvx seg256_libsvx; do
python tools/exp/run_bench_with_lib.py svision_amd/$lib.so --gpus 1 --workload cfg1 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1 $lib', round(d['value']), round(d['e2e']['seconds'],3), 'res', round(d['config']['resident_sites_per_s']))"
done
