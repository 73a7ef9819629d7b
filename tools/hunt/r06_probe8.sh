#!/bin/bash
# Round 6 (second session): the per-frame upload with unchanged chunks recognised — tests, the bench line's host_buffers block
set -u
OUT=gpurun_out/r06_probe8; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_reconfigure.py tests/test_gpu_ray_tile.py tests/test_gpu_frames_in_flight.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.txt 2>&1
echo "tests rc $?: $(grep -E 'passed|failed' $OUT/tests.txt | tail -1)" | tee -a $OUT/summary.txt
grep -E "FAILED|Error" $OUT/tests.txt | head | tee -a $OUT/summary.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
d = json.load(open("gpurun_out/r06_probe8/bench.json"))
print("ms_per_step", d["ms_per_step"], "parity", d["cpu_baseline"].get("parity_texels_differing"))
for k, v in d["host_buffers"].items():
    print(k, json.dumps(v)[:300])
PY
