#!/bin/bash
# timeline of one rank's slab of an 8-way sharded C3 grid in DDGI mode at 8 frames in flight (GPU box): rocprofv3 --kernel-trace of
# tools/fif_timing.py restricted to that configuration; prints the last launches (start since the first, duration, gap to the previous end, stream)
OUT=/tmp/prof_slab; rm -rf $OUT; cd /tmp && export TMPDIR=/tmp
FIF_MODE=${FIF_MODE:-ddgi} FIF_WORLDS=${FIF_WORLDS:-8} FIF_FIFS=${FIF_FIFS:-8} FIF_UPDATES=24 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/fif_timing.py 2>/dev/null | grep world
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-70:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = {}
last_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end) / 1e3 if last_end is not None else 0.0
    print("%9.1f us  +%7.1f us  gap %6.1f  q %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r["Kernel_Name"].split("(")[0][:60]))
    last_end = max(last_end or e, e)
PY
