#!/bin/bash
# Round 6, second session — the final build's evidence: profiles (tools/collect_r06.sh e), the per-frame upload before / after, fuzz + stress
set -u
OUT=gpurun_out/r06_final2; mkdir -p $OUT
t0=$(date +%s)
bash tools/collect_r06.sh e > $OUT/collect.log 2>&1
echo "collect_r06.sh e: $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
cat > /tmp/upload_ab.py <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ddgi_amd
from bench import WORKLOADS
w = WORKLOADS["c3"]
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
rays = eng.get_probe_rays()
ts = []
for i in range(8):
    t = time.perf_counter(); eng.upload_probe_rays(rays); ts.append(time.perf_counter() - t)
eng.probe_update(); eng.synchronize()
t = time.perf_counter()
for i in range(5):
    eng.upload_probe_rays(rays); eng.probe_update()
eng.synchronize()
step = (time.perf_counter() - t) / 5
print("%s: upload of %d MB: first %.1f ms, then %s ms (median %.2f ms = %.1f GB/s); upload + update per step %.2f ms = %.3f G rays/s" % (
    os.environ.get("DDGI_LIB", "libddgi_probe.so (this build)"), rays.nbytes >> 20, ts[0] * 1e3, " ".join("%.1f" % (x * 1e3) for x in ts[1:]), np.median(ts[1:]) * 1e3, rays.nbytes / np.median(ts[1:]) / 1e9, step * 1e3, len(rays) / step / 1e9))
PY
{ echo "# ddgi_upload_probe_rays of C3's ray buffer from a numpy array (pageable memory), 8 times, then 5 steps of upload + update (round 6, second session; one MI355X box)";
  DDGI_LIB=$GRAFT_REPO_ROOT/dynamic-diffuse-global-illumination-minecraft_amd/libddgi_probe_old.so python /tmp/upload_ab.py 2>/dev/null | sed 's|.*libddgi_probe_old.so|the library before (single-threaded check, vector::assign, copies from pageable memory)|';
  python /tmp/upload_ab.py 2>/dev/null; } > $OUT/host_buffers.txt
cat $OUT/host_buffers.txt | tee -a $OUT/summary.txt
t0=$(date +%s)
{ timeout 500 python tools/fuzz_parity.py 60 606; timeout 300 python tools/fuzz_sampler.py 2>&1 | tail -4; timeout 300 python tools/stress_queue_kernel.py 2>&1 | tail -6; } > $OUT/fuzz_and_stress.txt 2>&1
echo "fuzz + stress: $(( $(date +%s) - t0 )) s: $(tail -3 $OUT/fuzz_and_stress.txt | tr '\n' ' ' | cut -c1-400)" | tee -a $OUT/summary.txt
