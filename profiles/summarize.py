#!/usr/bin/env python3
"""Turns rocprofv3 output of `bench.py --pipelined 0 --no-overlap --no-graph` (every kernel alone on the device)
into per-op-family tables that can be held against bench.py's own numbers.

  summarize.py trace KERNEL_TRACE.csv OUT.json
      per family: launches, mean / median / min / max duration in ms -- the FIRST launch of every kernel symbol
      is left out (it carries the lazy code-object load and first-touch faults: 50 ms outliers)
  summarize.py pmc FETCH_counter_collection.csv WRITE_counter_collection.csv SCENES OUT.json
      per family: HBM bytes per launch from two separate PMC passes (FETCH_SIZE, WRITE_SIZE; KiB).
      /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of a
      streaming read, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; WRITE_SIZE is exact.

Family = one C-ABI call of the SA stack. The grouping call of a level (epnet_group_concat_multi, both MSG scales)
launches the centred-xyz kernel per scale and, when the level has features, the LDS-staged row gather serving both
scales: together that is one "group" launch.
"""
import collections
import csv
import json
import re
import statistics
import sys

# kernel-name prefixes (template argument lists may continue) -> op family
FPS = {"fps_indexed_kernel<8, 32": "fps N=16384 M=4096", "fps_indexed_kernel<4, 16": "fps N=4096 M=1024",
       "fps_pruned_kernel<8, 32": "fps N=16384 M=4096", "fps_pruned_kernel<4, 16": "fps N=4096 M=1024",
       "fps_wave_kernel<1, 16": "fps N=1024 M=256", "fps_wave_kernel<8, 2": "fps N=1024 M=256", "fps_wave_kernel<1, 4": "fps N=256 M=64",
       "fps_bigscene_kernel": "fps N=65536 M=16384"}


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def families(rows):
    """rows: dicts with 'name' in dispatch order -> list of (family, [row indices])"""
    out = []
    i = 0
    while i < len(rows):
        n = rows[i]["name"]
        fam = None
        idx = [i]
        for key, label in FPS.items():
            if key in n:
                fam = label
        if fam is None:
            if "group_xyz_centred" in n:
                # one grouping CALL (epnet_group_concat_multi: both scales of the level) = the run of centred-xyz and row-gather
                # kernels up to the next op: [xyz, xyz] without features, [xyz, xyz, lds2] with the rows staged once,
                # [xyz, rows, xyz, rows] when the launch is too small for that, [xyz, (transpose, rows) per chunk of scenes] for rows beyond LDS
                fam = "group"
                while idx[-1] + 1 < len(rows) and any(k in rows[idx[-1] + 1]["name"] for k in ("group_xyz_centred", "gather_rows", "transpose_cn")):
                    idx.append(idx[-1] + 1)
            elif "gather_rows" in n or "gather_centres" in n:
                fam = "gather"
            elif "bq_index_kernel" in n:
                fam = "scene_index"
            elif "bq_query_kernel" in n or "bq_query2_kernel" in n or "ball_query_kernel" in n:
                fam = "ball_query " + re.search(r"(bq_query2?_kernel<[\w, ]+>|ball_query_kernel<\d+>)", n).group(1)
            else:
                fam = "torch: " + n[:60]
        out.append((fam, idx))
        i = idx[-1] + 1
    return out


def cmd_trace(path, out):
    rows = [{"name": short(r["Kernel_Name"]), "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])}
            for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: r["start"])
    seen = set()
    for r in rows:
        r["first"] = r["name"] not in seen
        seen.add(r["name"])
    acc = collections.defaultdict(list)
    for fam, idx in families(rows):
        if any(rows[i]["first"] for i in idx):
            continue
        acc[fam].append(sum(rows[i]["end"] - rows[i]["start"] for i in idx) / 1e6)
    table = {f: {"launches": len(v), "mean_ms": round(statistics.fmean(v), 5), "median_ms": round(statistics.median(v), 5),
                 "min_ms": round(min(v), 5), "max_ms": round(max(v), 5), "total_ms": round(sum(v), 3)} for f, v in acc.items()}
    json.dump({"_doc": "per-family kernel durations from " + path + " (first launch of every kernel symbol excluded)",
               "families": dict(sorted(table.items(), key=lambda kv: -kv[1]["total_ms"]))}, open(out, "w"), indent=1)
    for f, v in sorted(table.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
        print("%-34s n=%4d mean %.4f median %.4f min %.4f max %.4f ms" % (f[:34], v["launches"], v["mean_ms"], v["median_ms"], v["min_ms"], v["max_ms"]))


def load_pmc(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [{"name": short(r["Kernel_Name"]), "v": float(r["Counter_Value"])} for r in rows]


def cmd_pmc(fetch_csv, write_csv, scenes, out):
    res = {}
    for counter, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        rows = load_pmc(path, counter)
        acc = collections.defaultdict(list)
        for fam, idx in families(rows):
            acc[fam].append(sum(rows[i]["v"] for i in idx))
        res[counter] = acc
    table = {}
    for fam in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
        f, w = res["FETCH_SIZE"].get(fam, [0.0]), res["WRITE_SIZE"].get(fam, [0.0])
        fk, wk = statistics.fmean(f), statistics.fmean(w)
        table[fam] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
                      "hbm_bytes_avg": int((2 * fk + wk) * 1024)}
    json.dump({"_doc": __doc__.strip(), "scenes_per_launch": int(scenes), "families": table}, open(out, "w"), indent=1)
    for f, v in sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes_avg"] * kv[1]["launches"])[:12]:
        print("%-34s n=%4d  %10.1f MB/launch" % (f[:34], v["launches"], v["hbm_bytes_avg"] / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        cmd_trace(sys.argv[2], sys.argv[3])
    else:
        cmd_pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
