for v in "$@"; do
  EPNET_HIP_LIB=scratch/libs/lib_$v.so timeout -k 10 200 python bench.py --cpu-scenes 0 > gpurun_out/v_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/v_$v.json"))
print("variant $v: step", d["ms_per_step"], "group avg", d["kernels"]["group"]["avg_ms"], flush=True)
PY
done
timeout -k 10 200 python bench.py --cpu-scenes 0 > gpurun_out/v_base.json
python - <<PY
import json
d=json.load(open("gpurun_out/v_base.json"))
print("variant base: step", d["ms_per_step"], "group avg", d["kernels"]["group"]["avg_ms"], flush=True)
PY
