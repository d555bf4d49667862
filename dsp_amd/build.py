"""Builds the native pieces in-tree (so the .so files travel with the repository snapshot).

    libmdc_parse.so   the product: gfx950 kernels + C ABI (hipcc)
    libmdc_gen.so     synthetic data generator (gcc)
    examples/xmltodb_loop   the C++ ingest loop over include/mdc_xmltodb.hpp (g++)
    examples/xmltodb_compat the reference's loop, call for call, over include/mdc_public_compat.h (g++)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wall", "-Wno-unused-function",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    out = "\n".join(l for l in r.stdout.splitlines() if "packed-fp32-ops" not in l)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + out)
    if out.strip():
        print(out, file=sys.stderr)


def build_parse(force=False):
    target = os.path.join(HERE, "libmdc_parse.so")
    srcs = [os.path.join(CSRC, f) for f in ("mdc_kernels.hip", "mdc_format.hip", "mdc_abi.cpp", "mdc_stream.cpp", "mdc_route.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("mdc_device.h", "mdc_numparse.h", "mdc_bignum.h")] + \
        [os.path.join(ROOT, "include", f) for f in ("mdc_parse.h", "mdc_format.h", "mdc_route.h")]
    if force or _newer(target, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc] + HIPCC_FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip"] + srcs +
             ["-o", target, "-lz", "-pthread"])
    return target


def build_diag(force=True, define="MDC_PHASE_TIMING", name="libmdc_parse_diag.so"):
    """Diagnostic variants (never used by tests or bench): MDC_PHASE_TIMING adds per-phase cycle
    counters, MDC_ABLATE lets MDC_DEBUG_SKIP switch phases off."""
    target = os.path.join(HERE, name)
    srcs = [os.path.join(CSRC, f) for f in ("mdc_kernels.hip", "mdc_format.hip", "mdc_abi.cpp", "mdc_stream.cpp", "mdc_route.cpp")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _run([hipcc] + HIPCC_FLAGS + ["-D" + define, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                                  "-x", "hip"] + srcs + ["-o", target, "-lz", "-pthread"])
    return target


def build_gen(force=False):
    target = os.path.join(HERE, "libmdc_gen.so")
    src = os.path.join(CSRC, "mdc_gen.c")
    if force or _newer(target, [src]):
        _run(["gcc", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-o", target, src])
    return target


def build_examples(force=False):
    target = os.path.join(ROOT, "examples", "xmltodb_loop")
    src = os.path.join(ROOT, "examples", "xmltodb_loop.cpp")
    deps = [src, os.path.join(ROOT, "include", "mdc_xmltodb.hpp"), os.path.join(ROOT, "include", "mdc_parse.h"),
            os.path.join(HERE, "libmdc_parse.so")]
    if force or _newer(target, deps):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src, "-o", target,
              "-L", HERE, "-lmdc_parse", "-Wl,-rpath,$ORIGIN/../dsp_amd"])
    return target


def build_compat_example(force=False):
    """The reference's ingest loop, call for call, over include/mdc_public_compat.h (g++)."""
    target = os.path.join(ROOT, "examples", "xmltodb_compat")
    src = os.path.join(ROOT, "examples", "xmltodb_compat.cpp")
    deps = [src, os.path.join(ROOT, "include", "mdc_public_compat.h"), os.path.join(ROOT, "include", "mdc_parse.h"),
            os.path.join(HERE, "libmdc_parse.so")]
    if force or _newer(target, deps):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src, "-o", target,
              "-L", HERE, "-lmdc_parse", "-Wl,-rpath,$ORIGIN/../dsp_amd"])
    return target


def build_all(force=False):
    return [build_parse(force), build_gen(force), build_examples(force), build_compat_example(force)]


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv)))
