set -eu
cd /tmp && export TMPDIR=/tmp
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
rm -rf "$ROOT/gpurun_out/trace" && mkdir -p "$ROOT/gpurun_out/trace"
cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --cpu-scenes 0 --batch ${B:-256} --steps 4 --warmup 2 ${EXTRA:-} > gpurun_out/trace/bench.json
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the timed steps: take the last 40% of FPS<8,32> launches from graph replays: just print a window around the 3rd-from-last big fps kernel
fps = [i for i, r in enumerate(rows) if "fps_indexed_kernel<8, 32" in r["Kernel_Name"] or "fps_pruned_kernel<8, 32" in r["Kernel_Name"]]
print("fps L1 launches", len(fps))
# steps: warmup eager 2 (capture warmup) + capture itself not executed + warmup 2 + steps 4 + instrumented 4
k = fps[-6]   # a replay in the timed region (before the 4 instrumented eager steps)
t0 = int(rows[k]["Start_Timestamp"])
k_end = fps[-5]
t_next = int(rows[k_end]["Start_Timestamp"])
print("step window %.3f ms" % ((t_next - t0) / 1e6))
out = open("gpurun_out/trace/step_timeline.txt", "w")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0 - 200000 or s > t_next + 200000: continue
    name = r["Kernel_Name"].replace("epnet::", "").split("(")[0][:60]
    line = "%9.3f %9.3f %8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), name)
    out.write(line + "\n")
out.close()
PY
