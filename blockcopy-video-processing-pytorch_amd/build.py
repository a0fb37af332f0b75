"""Build libblockcopy_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "blockcopy_hip.hip")]
EXTRA_SRC = [os.path.join(HERE, "csrc", "policy_net.hip")]      # translation units of their own (the RL policy's CNN: forward / backward / optimizer)
HDR = [os.path.join(os.path.dirname(HERE), "include", "blockcopy_hip.h")] + [os.path.join(HERE, "csrc", f) for f in
                                                                             ("conv3x3_mfma.inc", "conv3x3_v2.inc", "conv3x3_wino.inc", "conv3x3_wino32.inc", "conv3x3_wino4.inc", "stem7x7.inc", "head1x1.inc", "pred3x3.inc", "gemm1x1.inc", "spp.inc")]
OUT = os.path.join(HERE, "lib", "libblockcopy_hip.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
ARCH = "gfx950"
# the one source is compiled as 7 translation units (-DBC_PART=n, see ConvV2Args in csrc/blockcopy_hip.hip): part 0 = everything
# but the decompositions of the fused conv kernel, parts 1..6 = one (dtype, kernel size) slice of them each, part 7 = its Winograd form,
# part 8 = the wide-tile Winograd form, part 9 = the dilation-2 form of the direct kernel, part 10 = the Winograd F(4x4,3x3) form
PARTS = list(range(11)) + [12, 13]      # 12 / 13: the split (16-bit matrix pipe) form of the fp32 direct conv, kernel size 3 / 1


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SRC + EXTRA_SRC + HDR)


def build_hip_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result", "-Wno-unused-function"]
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    objs = [os.path.join(OBJ_DIR, f"part{n}.o") for n in PARTS] + [os.path.join(OBJ_DIR, os.path.basename(f)[:-4] + ".o") for f in EXTRA_SRC]

    def fresh(obj, deps):
        return not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps)

    def compile_part(n):
        # (an object is rebuilt when its own source -- or a header it includes -- is newer: editing policy_net.hip costs one unit, not twelve)
        if fresh(objs[n], ([EXTRA_SRC[n - len(PARTS)], HDR[0]] if n >= len(PARTS) else SRC + HDR) + [os.path.abspath(__file__)]):
            return
        if n >= len(PARTS):
            subprocess.check_call([hipcc()] + flags + ["-c", "-o", objs[n], EXTRA_SRC[n - len(PARTS)]])
        else:
            subprocess.check_call([hipcc()] + flags + [f"-DBC_PART={PARTS[n]}", "-c", "-o", objs[n]] + SRC)

    jobs = int(os.environ.get("BC_BUILD_JOBS", "0")) or max(1, min(len(objs), os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(compile_part, range(len(objs))))
    subprocess.check_call([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fvisibility=hidden", "-o", OUT + ".tmp"] + objs)
    os.replace(OUT + ".tmp", OUT)
    return OUT


ABI_DEMO_SRC = os.path.join(os.path.dirname(HERE), "tests", "abi_c", "abi_roundtrip.cpp")
ABI_DEMO = os.path.join(os.path.dirname(HERE), "tests", "abi_c", "abi_roundtrip")


def build_abi_consumer(force: bool = False) -> str:
    """tests/abi_c/abi_roundtrip: a C++/HIP program that uses the library through include/blockcopy_hip.h alone (no
    PyTorch) -- what a non-Python host of the reference's block ops would link."""
    lib = build_hip_library()
    if not force and os.path.exists(ABI_DEMO) and os.path.getmtime(ABI_DEMO) >= max(
            os.path.getmtime(f) for f in [ABI_DEMO_SRC, lib] + HDR):
        return ABI_DEMO
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", "-I", os.path.dirname(HDR[0]), ABI_DEMO_SRC,
           "-L", os.path.dirname(lib), "-lblockcopy_hip", "-Wl,-rpath," + os.path.dirname(lib),
           "-Wl,-rpath,$ORIGIN/../../blockcopy-video-processing-pytorch_amd/lib", "-o", ABI_DEMO + ".tmp"]
    subprocess.check_call(cmd)
    os.replace(ABI_DEMO + ".tmp", ABI_DEMO)
    return ABI_DEMO


if __name__ == "__main__":
    print(build_hip_library(force="--force" in sys.argv, verbose="-v" in sys.argv))
