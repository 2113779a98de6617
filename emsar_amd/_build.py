"""Build driver: compiles the HIP library (gfx950) and the C host in-tree.

    python -m emsar_amd._build            # everything
    python -m emsar_amd._build --force

Outputs (git-ignored, shipped to the GPU box by gpurun):
    emsar_amd/libemsar_hip.so     kernels + C ABI (include/emsar_hip.h)
    emsar_amd/libemsar_host.so    C host: rsh / alignment readers, model preparation, .fpkm writer
    emsar_amd/emsar-hip           C command-line driver linked against both
hipcc cross-compiles without a GPU; intermediates (.s with register usage) go to build/.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
BUILD = os.path.join(ROOT, "build")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

HIP_SO = os.path.join(PKG, "libemsar_hip.so")
HOST_SO = os.path.join(PKG, "libemsar_host.so")
CLI = os.path.join(PKG, "emsar-hip")

HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-std=c++17", "-fPIC", "-shared",
             "-Wall", "-Wextra", "-Wno-unused-value"]
HOST_SRC = ["rsh.c", "align.c", "model.c", "output.c", "pbgzf.c", "hostutil.c"]
C_FLAGS = ["-O2", "-std=c11", "-fPIC", "-Wall", "-Wextra", "-D_POSIX_C_SOURCE=200809L"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_hip(force=False):
    src = os.path.join(CSRC, "emsar_hip.hip")
    src2 = os.path.join(CSRC, "collapse.hip")
    deps = [src, src2, os.path.join(ROOT, "include", "emsar_hip.h")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))      # layouts, sets, kernels_*.hpp
    if force or _stale(HIP_SO, deps):
        os.makedirs(BUILD, exist_ok=True)
        # compile inside build/ so that -save-temps leaves the .s (register / LDS usage) there
        _run([HIPCC] + HIP_FLAGS + ["-save-temps", "-o", HIP_SO, src, src2], cwd=BUILD)
    return HIP_SO


def build_host(force=False):
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in HOST_SRC]
    if not all(os.path.exists(s) for s in srcs):
        return None
    deps = srcs + [os.path.join(hdir, "emsar_host.h"), os.path.join(ROOT, "include", "emsar_hip.h")]
    if force or _stale(HOST_SO, deps):
        _run(["gcc"] + C_FLAGS + ["-shared", "-o", HOST_SO] + srcs + ["-lm", "-lz", "-ldl"])
    main = os.path.join(hdir, "emsar_hip_main.c")
    if os.path.exists(main) and (force or _stale(CLI, deps + [main, HIP_SO])):
        _run(["gcc"] + C_FLAGS + ["-o", CLI, main, "-I" + os.path.join(ROOT, "include"), "-L" + PKG,
              "-lemsar_host", "-lemsar_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lz", "-lpthread", "-ldl"])
    return HOST_SO


def build_all(force=False):
    build_hip(force)
    build_host(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built:", ", ".join(p for p in (HIP_SO, HOST_SO, CLI) if os.path.exists(p)))
