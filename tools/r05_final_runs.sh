set -u
O=gpurun_out/${1:-r05final}
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/steps20_a.json 2> $O/steps20_a.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine > $O/steps20_b.json 2> $O/steps20_b.err
for w in cfg1 cfg2 ont contig; do python bench.py --gpus 1 --workload $w --no-calibration > $O/$w.json 2> $O/$w.err; done
python bench.py --gpus 1 --steps 100 --warmup 5 --no-calibration > $O/e2e_100windows.json 2> $O/e2e_100windows.err
for f in $O/*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d.get("e2e") or {}; c=d.get("e2e_cold_cache") or {}
    print(sys.argv[1].split("/")[-1], "value", round(d["value"]), "e2e_s", e.get("seconds") and round(e["seconds"],3), "resident", round(d["config"].get("resident_sites_per_s",0)), "ratio", d["config"].get("file_inclusive_over_resident") and round(d["config"]["file_inclusive_over_resident"],3), "frac", round(d["roofline"]["frac"],3), "alone", d["roofline"].get("frac_stage_alone"), "host_engine_s", (d.get("e2e_host_ingest") or {}).get("seconds"), "cold_s", c.get("seconds"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "replans", (e.get("rank0_feed") or {}).get("replans"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
done
