"""Builds libddgi_probe.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_HERE, "libddgi_probe.so")
_SOURCES = ["ddgi_kernels.hip", "ddgi_trace_wf.hip", "ddgi_blend_sample.hip", "ddgi_device.h", "ddgi_oct.h", "ddgi_sampler.h", "ddgi_render.hip", "ddgi_host.cpp", "ddgi_engine.cpp", "ddgi_engine.h",
            "ddgi_exchange.cpp", "ddgi_visibility.hip", "ddgi_pinned_math.h", "ddgi_scene.h", "ddgi_types.h", "ddgi_host.h", "Makefile"]
_PROF_SO = os.path.join(_HERE, "libddgi_probe_prof.so")


def library_path():
    return _SO


def profiling_library_path():
    """The -DDDGI_PROFILING build (ablation switches, fault injection); never the default library."""
    return _PROF_SO


def _stale(so=_SO):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    deps = [os.path.join(_CSRC, s) for s in _SOURCES]
    deps.append(os.path.join(_HERE, "..", "include", "ddgi_probe.h"))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -shared -o libddgi_probe.so; returns the path."""
    jobs = str(min(8, os.cpu_count() or 1))
    if force or _stale():
        cmd = ["make", "-C", _CSRC, "-j", jobs] + (["-B"] if force else [])
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout)
        if res.returncode != 0:
            raise RuntimeError("building libddgi_probe.so failed:\n" + res.stdout)
    return _SO


def build_profiling_library(force=False):
    """make prof -> libddgi_probe_prof.so (selected by a test or tool with DDGI_LIB=<path>)."""
    if force or _stale(_PROF_SO):
        res = subprocess.run(["make", "-C", _CSRC, "-j", str(min(8, os.cpu_count() or 1)), "prof"] + (["-B"] if force else []),
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("building libddgi_probe_prof.so failed:\n" + res.stdout)
    return _PROF_SO
