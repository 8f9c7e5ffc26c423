set -u
O=gpurun_out/r05a
mkdir -p $O
python -m pytest tests/test_slices_gpu.py tests/test_gpu_inflate.py tests/test_htslike_bam.py -x -q -m gpu > $O/pytest_slices.log 2>&1; echo "pytest slices rc $?" >> $O/pytest_slices.log
tail -5 $O/pytest_slices.log
for w in cfg1 ont; do SVX_TIMING=1 python bench.py --gpus 1 --workload $w --no-cpu-baseline --no-calibration --no-other-engine > $O/$w.json 2> $O/$w.err; done
SVX_TIMING=1 python bench.py --gpus 1 --workload cfg1 --no-cpu-baseline --no-calibration --no-other-engine --e2e-sweep "SVX_SLICES=0;SVX_SLICES=1;SVX_SLICES=0;SVX_SLICES=1" > $O/cfg1_sweep.json 2> $O/cfg1_sweep.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine > $O/steps20.json 2> $O/steps20.err
for f in $O/*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d.get("e2e") or {}
    print(sys.argv[1].split("/")[-1], "value", round(d["value"]), "e2e_s", e.get("seconds") and round(e["seconds"],3), "resident", round(d["config"].get("resident_sites_per_s",0)), "ratio", d["config"].get("file_inclusive_over_resident") and round(d["config"]["file_inclusive_over_resident"],3), "frac", round(d["roofline"]["frac"],3), "sweep", [(s["env"], round(s["seconds"],3)) for s in d.get("e2e_sweep",[])], "slices", (e.get("rank0_feed") or {}).get("slices"), "replans", (e.get("rank0_feed") or {}).get("replans"), "first_ready", (e.get("rank0_feed") or {}).get("first_ready_s"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
done
python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc $?" >> $O/pytest_all.log
tail -5 $O/pytest_all.log
