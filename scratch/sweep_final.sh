set -e
mkdir -p gpurun_out/sweep
for cfg in "1 0" "16 0" "16 1" "64 1" "128 1" "256 0" "256 1" "512 1"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --cpu-scenes 0 --batch $1 --pipelined $2 > gpurun_out/sweep/b$1_p$2.json
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep/b$1_p$2.json"))
print(json.dumps({"scenes_per_step": $1, "pipelined": bool($2), "ms_per_step": d["ms_per_step"], "points_per_s": d["value"], "stack_hbm_frac": d["stack_hbm_frac"]}), flush=True)
PY
done
