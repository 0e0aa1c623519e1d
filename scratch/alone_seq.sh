set -e
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/alone && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/alone
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/alone -- python3 bench.py --cpu-scenes 0 --steps 3 --warmup 1 --pipelined 0 --no-overlap --no-graph > gpurun_out/alone/bench.json
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/alone/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fps = [i for i, r in enumerate(rows) if "fps_indexed_kernel<8, 32" in r["Kernel_Name"]]
a, b = fps[-2], fps[-1]
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 0.02: print("%8.3f  %s" % (d, r["Kernel_Name"].replace("epnet::", "").split("(")[0][:60]))
PY
