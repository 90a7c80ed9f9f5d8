#!/usr/bin/env python3
"""AddressSanitizer build of libnerfloam_hip.so (host AND device code: hipcc -fsanitize=address -shared-libsan, gfx950:xnack+) into
ab_libs/libnerfloam_hip_asan.so - SURVEY section 5 "race detection / sanitizers".  Cross-compiles here; scripts/asan_run.sh runs the small
parity goldens through it on the GPU box (NL_LIB_PATH selects the library, HSA_XNACK=1, the ASan runtime preloaded).

    python scripts/asan_build.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_loam_amd import build as B                             # noqa: E402

OUT_DIR = os.path.join(ROOT, "ab_libs")
OUT = os.path.join(OUT_DIR, "libnerfloam_hip_asan.so")
FLAGS = ["--offload-arch=gfx950:xnack+", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-fsanitize=address", "-shared-libsan"]


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = os.path.join(OUT_DIR, "asan_obj")
    os.makedirs(tmp, exist_ok=True)
    procs, objs = [], []
    for src in B.SOURCES:
        obj = os.path.join(tmp, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(src)
            sys.stderr.write(f"--- {src}\n" + out.decode()[-3000:])
    if failed:
        raise SystemExit(f"ASan build failed for {failed}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950:xnack+", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan"] + objs + ["-o", OUT])
    print(OUT)


if __name__ == "__main__":
    main()
