set -u
O=gpurun_out/${1:-r05d}
mkdir -p $O
SW="SVX_X=0;SVX_LARGE_GROUP_MB=400,SVX_PIPE_GROUP_MB=400;SVX_LARGE_GROUP_MB=400,SVX_PIPE_GROUP_MB=200;SVX_WAVE_LZ_BELOW=1;SVX_WAVE_LZ_BELOW=16000,SVX_LARGE_GROUP_MB=400,SVX_PIPE_GROUP_MB=400;SVX_X=0;SVX_LARGE_GROUP_MB=400,SVX_PIPE_GROUP_MB=400"
for w in cfg1 ont; do python bench.py --gpus 1 --workload $w --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg --e2e-sweep "$SW" > $O/$w.json 2> $O/$w.err; grep "e2e sweep" $O/$w.err; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-other-engine --no-cold-leg --e2e-sweep "$SW" > $O/steps20.json 2> $O/steps20.err; grep "e2e sweep" $O/steps20.err
for f in $O/*.json; do python - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e=d.get("e2e") or {}
print(sys.argv[1].split("/")[-1], "value", round(d["value"]), "e2e_s", round(e["seconds"],3), "resident", round(d["config"]["resident_sites_per_s"]), "ratio", round(d["config"]["file_inclusive_over_resident"],3), "replans", e["rank0_feed"].get("replans"), "gaps", e.get("cnn_gaps_ms_rank0"))
P
done
