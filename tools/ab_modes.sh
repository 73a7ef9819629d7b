D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for lib in libddgi_probe.so libddgi_probe_in1.so; do  # (in1: whatever A/B library was built last)
 for cfg in "--workload c3" "--workload c3 --mode ddgi" "--workload c4 --steps 6 --warmup 2" "--workload c5 --mode ddgi --steps 6 --warmup 3"; do
  echo -n "$lib $cfg: "
  DDGI_LIB=$D/$lib python bench.py $cfg --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); fm=d.get('fast_march') or {}
print('%.4f ms/step  kernel %.4f  fast %s  mw %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], fm.get('ms_per_step'), d.get('tuning',{}).get('march_waves')))"
 done
done
