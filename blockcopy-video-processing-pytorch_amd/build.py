"""Build libblockcopy_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "blockcopy_hip.hip")]
HDR = [os.path.join(os.path.dirname(HERE), "include", "blockcopy_hip.h")]
OUT = os.path.join(HERE, "lib", "libblockcopy_hip.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SRC + HDR)


def build_hip_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-result", "-o", OUT + ".tmp"] + SRC
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build_hip_library(force="--force" in sys.argv, verbose="-v" in sys.argv))
