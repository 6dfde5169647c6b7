#!/usr/bin/env python3
"""Assemble profiles/rNN_traffic_*.json from the counter CSVs of tools/profile_bench.sh.
usage: tools/traffic_json.py <outdir of profile_bench.sh> <kernel name substring> <out.json> <command string> [kt-substring]
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KB; the x2 on FETCH_SIZE is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md §HBM for wide coalesced reads), separate --pmc passes."""
import csv, glob, json, subprocess, sys

root, key, out, command = sys.argv[1:5]


def mean_counter(pass_dir, counter):
    vals = []
    for f in glob.glob(f"{root}/{pass_dir}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if key in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch, nf = mean_counter("pmc_fetch", "FETCH_SIZE")
write, nw = mean_counter("pmc_write", "WRITE_SIZE")
avg_ms = None
for f in glob.glob(f"{root}/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if key in row["Name"]:
            avg_ms = float(row["AverageNs"]) / 1e6
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
doc = {
    "kernel": key,
    "command": command,
    "commit": commit,
    "fetch_size_kb_raw_per_launch": fetch, "write_size_kb_raw_per_launch": write, "dispatches_averaged": [nf, nw],
    "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md §HBM: the counter tallies 128-B requests at 64 B), WRITE_SIZE exact",
    "fetch_bytes_per_launch": None if fetch is None else 2 * fetch * 1024,
    "write_bytes_per_launch": None if write is None else write * 1024,
    "hbm_bytes_per_launch": None if (fetch is None or write is None) else 2 * fetch * 1024 + write * 1024,
    "rocprof_avg_launch_ms": avg_ms,
    "note": "FETCH_SIZE counts L2 -> fabric requests: Infinity-Cache hits are included, so this is an upper bound of true HBM reads",
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc))
