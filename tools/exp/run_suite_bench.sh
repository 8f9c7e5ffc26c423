cd /root/repo
mkdir -p gpurun_out/r05h
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/r05h/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r05h/pytest_all.log; tail -6 gpurun_out/r05h/pytest_all.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05h/bench20.json 2> gpurun_out/r05h/bench20.err; python -c "
import json; d=json.loads(open('gpurun_out/r05h/bench20.json').read().strip().splitlines()[-1]); print('value', round(d['value']), 'res', round(d['config']['resident_sites_per_s']), 'frac', d['roofline']['frac'], d['roofline'].get('frac_stage_alone'))"
