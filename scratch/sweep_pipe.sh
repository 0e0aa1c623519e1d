set -e
mkdir -p gpurun_out/pipe
IFS=';' read -ra LIST <<< "${CFGS:-16 1;64 1;256 1}"
for cfg in "${LIST[@]}"; do
  IFS=' ' read -r B P <<< "$cfg"
  timeout -k 10 160 python bench.py --cpu-scenes 0 --batch $B --pipelined $P $EXTRA > gpurun_out/pipe/b${B}_p${P}$TAG.json
  python - <<PY
import json
d=json.load(open("gpurun_out/pipe/b${B}_p${P}$TAG.json"))
print("batch $B pipelined $P $EXTRA:", d["ms_per_step"], "ms/step", round(d["value"]/1e6,1), "Mpts/s hbm", d["stack_hbm_frac"], "dominant", d["roofline"]["kernel"], d["roofline"]["frac"], flush=True)
PY
done
