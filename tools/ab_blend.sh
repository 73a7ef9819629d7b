#!/bin/bash
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for lib in libddgi_probe.so $(cd $D; ls libddgi_probe_ng*.so); do
  echo -n "$lib : "; DDGI_LIB=$D/$lib python tools/ddgi_timing.py 2>/dev/null | grep "trace ms" | sed 's/.*blend ms//'
done
