#!/bin/bash
# The round's lines: the driver's command, the other BASELINE configs (each with its parity leg: device path vs CPU port at
# bench size), the 100-window job.
set -u
O=gpurun_out/${1:-r06final}
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/steps20_a.detail.json > $O/steps20_a.json 2> $O/steps20_a.err
for w in ${WORKLOADS:-cfg1 cfg2 ont contig}; do python bench.py --gpus 1 --workload $w --no-calibration --detail $O/$w.detail.json > $O/$w.json 2> $O/$w.err; echo "$w rc=$?"; done
if [ "${BIG:-1}" = 1 ]; then python bench.py --gpus 1 --steps 100 --warmup 5 --no-calibration --detail $O/e2e_100windows.detail.json > $O/e2e_100windows.json 2> $O/e2e_100windows.err; fi
for f in $O/*.json; do case $f in *.detail.json) continue;; esac; python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]; p=d.get("parity_check") or {}
    print(sys.argv[1].split("/")[-1], "bytes", len(open(sys.argv[1]).read()), "value", round(d["value"]), d.get("value_repeats",{}).get("seconds"), "resident", round(c.get("resident_sites_per_s") or 0), "ratio", c.get("file_inclusive_over_resident") and round(c["file_inclusive_over_resident"],3),
          "frac", round(d["roofline"]["frac"],3), "alone", d["roofline"].get("frac_stage_alone") and round(d["roofline"]["frac_stage_alone"],3), "cpu", (d.get("cpu_baseline") or {}).get("value"),
          "parity", {k:p.get(k) for k in ("ok","windows","tsv_equal","sites_equal","images_compared","max_softmax_delta")})
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
done
