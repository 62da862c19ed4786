"""In-tree build of libyolact_post.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libyolact_post.so")
SOURCES = ["api.hip", "score_filter.hip", "nms.hip", "mask.hip", "priors.hip", "rle.hip", "loss_grad.hip"]
HEADERS = ["common.hpp", os.path.join("..", "..", "include", "yolact_post.h")]
# -ffp-contract=off: YP-MATH writes every fma explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """`out`: experiment builds (extra -D flags) under another file name, loaded with YOLACT_POST_LIB=<path>."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    out = out or LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    # python -m yolact_amd.build [-o variant.so] [-DFLAG ...]
    argv = sys.argv[1:]
    out = None
    if "-o" in argv:
        i = argv.index("-o")
        out = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    print(build_lib(force=True, verbose=True, extra_flags=argv, out=out))
