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
# multi-rank rehearsals on ONE GPU (gloo: RCCL refuses two ranks on a device): the N-rank code path of the line, not a measurement
if [ "${RANKS:-1}" = 1 ]; then
  SVX_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-calibration --no-cold-leg --no-other-engine --detail $O/2ranks.detail.json > $O/2ranks.json 2> $O/2ranks.err; echo "2 ranks rc=$?"; cut -c1-900 $O/2ranks.json
  SVX_DIST_BACKEND=gloo python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --no-calibration --no-cold-leg --no-other-engine --detail $O/8ranks.detail.json > $O/8ranks.json 2> $O/8ranks.err; echo "8 ranks rc=$?"; cut -c1-900 $O/8ranks.json
fi
