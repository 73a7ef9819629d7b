#!/usr/bin/env python3
"""HBM-side bytes per batch of the cage-sample kernels (GPU box): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) over
tools/sample_bench.py on C3 (REF, DDGI) and C5 (DDGI: 3.2 GB of tiles) -> gpurun_out/profiles_out/<round>_<tag>_sample_traffic.json, the file
bench.py's sample.hbm_GBps replays (copy it to profiles/).  Bytes = FETCH_SIZE x 1024 x 2 (MI355X_MICROARCH.md: on gfx950 the counter tallies 128-byte
requests at 64) + WRITE_SIZE x 1024, per launch of the sample kernel proper (the grouping kernels are listed beside it)."""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rnd, tag = (sys.argv[1:] + ["r06", "x"])[:2]
out_dir = os.path.join(ROOT, "gpurun_out", "pmc_sample_traffic_" + tag)
os.makedirs(out_dir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles_out"), exist_ok=True)
from bench import WORKLOADS  # noqa: E402

records = []
for wl, mode in (("c3", 0), ("c3", 1), ("c5", 1)):
    env = dict(os.environ, SAMPLE_WORKLOAD=wl, SAMPLE_MODES=str(mode), TMPDIR="/tmp")
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        name = "%s_m%d_%s" % (wl, mode, counter)
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out_dir, "-o", name, "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "tools", "sample_bench.py")],
                       env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        agg = collections.defaultdict(list)
        try:
            with open(os.path.join(out_dir, name + "_counter_collection.csv")) as fh:
                for r in csv.DictReader(fh):
                    if "sample" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ddgi::", "").split("<")[0]].append(float(r["Counter_Value"]))
        except OSError as exc:
            print("#", name, "no data:", exc)
        for k, v in agg.items():
            per.setdefault(k, {})[counter] = (sum(v) / len(v), len(v))
    for k, c in per.items():
        f, w = c.get("FETCH_SIZE", (0.0, 0))[0], c.get("WRITE_SIZE", (0.0, 0))[0]
        records.append({"workload": WORKLOADS[wl]["name"], "mode": "ddgi" if mode else "ref", "kernel": k, "points": 1600 * 900, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                        "launches": c.get("FETCH_SIZE", (0, 0))[1], "hbm_bytes_per_batch": f * 1024 * 2 + w * 1024,
                        "note": "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024; rocprofv3 --pmc, one counter per pass, tools/sample_bench.py"})
path = os.path.join(ROOT, "gpurun_out", "profiles_out", "%s_%s_sample_traffic.json" % (rnd, tag))
with open(path, "w") as fh:
    json.dump(records, fh, indent=1)
for r in records:
    print("%-44s %-5s %-24s FETCH %12.0f KB  WRITE %10.0f KB  -> %.3f GB per batch" % (r["workload"][:44], r["mode"], r["kernel"], r["FETCH_SIZE_KB"], r["WRITE_SIZE_KB"], r["hbm_bytes_per_batch"] / 1e9))
