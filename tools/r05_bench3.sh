set -u
O=gpurun_out/${1:-r05c}
mkdir -p $O
for w in cfg1 ont; do SVX_TIMING=1 python bench.py --gpus 1 --workload $w --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg > $O/$w.json 2> $O/$w.err; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine > $O/steps20.json 2> $O/steps20.err
for f in $O/*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d.get("e2e") or {}
    c=d.get("e2e_cold_cache") or {}
    print(sys.argv[1].split("/")[-1], "value", round(d["value"]), "e2e_s", e.get("seconds") and round(e["seconds"],3), "resident", round(d["config"].get("resident_sites_per_s",0)), "ratio", d["config"].get("file_inclusive_over_resident") and round(d["config"]["file_inclusive_over_resident"],3), "frac", round(d["roofline"]["frac"],3), "alone", round(d["roofline"].get("frac_stage_alone") or 0,3), "slices", (e.get("rank0_feed") or {}).get("slices"), "replans", (e.get("rank0_feed") or {}).get("replans"), "first_ready", (e.get("rank0_feed") or {}).get("first_ready_s"), "gaps", e.get("cnn_gaps_ms_rank0"), "cold", c.get("seconds"), c.get("page_cache"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
done
