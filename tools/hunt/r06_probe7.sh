#!/bin/bash
# Round 6 (second session): the per-frame upload (threaded check + copy into a page-locked host copy) — its test, the bench line's host_buffers block, A/B against the library before
set -u
OUT=gpurun_out/r06_probe7; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_reconfigure.py tests/test_gpu_ray_tile.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.txt 2>&1
echo "tests rc $?: $(grep -E 'passed|failed' $OUT/tests.txt | tail -1)" | tee -a $OUT/summary.txt
grep -E "FAILED|Error" $OUT/tests.txt | head | tee -a $OUT/summary.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
d = json.load(open("gpurun_out/r06_probe7/bench.json"))
print("ms_per_step", d["ms_per_step"], "host_buffers", json.dumps(d.get("host_buffers"))[:700])
PY
python - <<'PY' 2>&1 | tee -a $OUT/summary.txt
# upload alone, timed: C3's 201 MB from a numpy array (pageable), 6 times
import time, numpy as np, sys
sys.path.insert(0, ".")
import ddgi_amd
from bench import WORKLOADS
w = WORKLOADS["c3"]
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
rays = eng.get_probe_rays()
for i in range(6):
    t = time.perf_counter(); eng.upload_probe_rays(rays); dt = time.perf_counter() - t
    print("upload %d: %.2f ms = %.1f GB/s" % (i, dt * 1e3, rays.nbytes / dt / 1e9))
PY
