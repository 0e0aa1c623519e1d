"""kernel timeline of one steady pipelined step out of a rocprofv3 --kernel-trace CSV: python timeline.py <dir> <anchor kernel substring> <out>"""
import csv, glob, sys
d, anchor, outp = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
skip_tail = int(sys.argv[4]) if len(sys.argv) > 4 else 4     # anchors launched after the timed loop (instrumented eager steps)
k, k2 = hits[-(skip_tail + 2)], hits[-(skip_tail + 1)]
t0, t1 = int(rows[k]["Start_Timestamp"]), int(rows[k2]["Start_Timestamp"])
out = open(outp, "w")
out.write("anchor launches %d; step window %.3f ms\n" % (len(hits), (t1 - t0) / 1e6))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0 - 100000 or s > t1 + 100000:
        continue
    name = r["Kernel_Name"].replace("epnet::", "").split("(")[0][:70]
    out.write("%9.3f %9.3f %8.3f  q%-3s %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), name))
out.close()
print(open(outp).read()[:6000])
