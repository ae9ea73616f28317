"""Builds librrnet_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

  python rrnet_amd/csrc/build.py [--force]

One object per .hip/.cpp file so that per-file flags stay possible (softnms.hip needs
-ffp-contract=off for bit-exactness with the x86 reference)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INC = os.path.join(ROOT, "include")
OUT = os.path.join(os.path.dirname(HERE), "librrnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", INC, "-I", HERE,
          "-Wno-unused-result", "-fno-gpu-rdc"]
PER_FILE = {
    "softnms.hip": ["-ffp-contract=off"],
    "hardnms.hip": ["-ffp-contract=off"],
    "refine.hip": ["-ffp-contract=off"],
    "targets.hip": ["-ffp-contract=off"],
    "psroi.hip": ["-ffp-contract=off"],
}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip") or f.endswith(".cpp"))


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(HERE, src), os.path.join(INC, "rrnet_hip.h")] + \
           [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(HERE, "_build", src + ".o")
    if force or _stale(obj, src):
        cmd = [HIPCC] + COMMON + PER_FILE.get(src, []) + ["-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
