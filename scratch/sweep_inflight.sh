set -e
mkdir -p gpurun_out/inflight
for cfg in "16 1" "16 2" "16 4" "64 2" "64 4" "256 2"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --cpu-scenes 0 --batch $1 --in-flight $2 --no-graph > gpurun_out/inflight/ng_b$1_l$2.json
  python - <<PY
import json
d=json.load(open("gpurun_out/inflight/ng_b$1_l$2.json"))
print("eager batch $1 in-flight $2:", d["ms_per_step"], "ms/step", round(d["value"]/1e6,1), "Mpts/s hbm", d["stack_hbm_frac"], flush=True)
PY
done
