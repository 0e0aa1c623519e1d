for kb in 64 32 48 96 128; do
  EPNET_GATHER_LDS_KB=$kb timeout -k 10 200 python bench.py --cpu-scenes 0 > gpurun_out/g_$kb.json
  python - <<PY
import json
d=json.load(open("gpurun_out/g_$kb.json"))
print("lds_kb $kb: step", d["ms_per_step"], "group avg", d["kernels"]["group"]["avg_ms"], "GBps", d["kernels"]["group"]["GBps"], flush=True)
PY
done
