D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for lib in libddgi_probe.so libddgi_probe_irr2.so; do
  echo "== $lib"
  DDGI_LIB=$D/$lib timeout 400 python bench.py --workload c5 --mode ddgi --steps 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d.get('blend')))"
done
