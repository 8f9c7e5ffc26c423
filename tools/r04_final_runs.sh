set -u
O=gpurun_out/final
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/steps20_a.json 2> $O/steps20_a.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/steps20_b.json 2> $O/steps20_b.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/steps20_c.json 2> $O/steps20_c.err
python bench.py --no-cpu-baseline > $O/default.json 2> $O/default.err
python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-calibration > $O/e2e_100windows.json 2> $O/e2e_100windows.err
for w in cfg1 ont contig; do python bench.py --gpus 1 --workload $w --no-cpu-baseline --no-calibration > $O/$w.json 2> $O/$w.err; done
SVX_DIST_BACKEND=gloo python bench.py --gpus 8 --steps 40 --no-cpu-baseline --no-calibration > $O/8ranks.json 2> $O/8ranks.err
python bench.py --resident --no-cpu-baseline --no-calibration > $O/wg_resident.json 2> $O/wg_resident.err
for f in $O/*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d.get("e2e") or {}
    print(sys.argv[1].split("/")[-1], "value", round(d["value"]), "e2e_s", e.get("seconds") and round(e["seconds"],3), "resident", round(d["config"].get("resident_sites_per_s",0)), "ratio", d["config"].get("file_inclusive_over_resident") and round(d["config"]["file_inclusive_over_resident"],3), "frac", round(d["roofline"]["frac"],3), "res_frac", d["roofline"].get("resident_leg",{}).get("frac_executed"), "host_engine_s", (d.get("e2e_host_ingest") or {}).get("seconds"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
done
