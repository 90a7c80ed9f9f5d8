"""Build libnerfloam_hip.so (gfx950) in-tree with hipcc.  `python -m nerf_loam_amd.build`.

hipcc cross-compiles without a GPU.  Flags that matter for parity:
  -ffp-contract=off      the op order of nl_device_math.h is the contract with the oracle
  -munsafe-fp-atomics    hardware global_atomic_add_f32 for the embedding-gradient scatter
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnerfloam_hip.so")
SOURCES = ["nl_geometry.hip", "nl_decoder.hip", "nl_field.hip", "nl_optim.hip", "nl_select.hip", "nl_dist.hip", "nl_criterion.hip", "nl_mesh.hip", "nl_octree.cpp", "nl_iteration.cpp", "nl_exchange.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nerfloam_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


SVO_TORCH_SRC = os.path.join(HERE, "torch_ext", "svo_torch.cpp")
SVO_TORCH_OUT = os.path.join(HERE, "libnl_svo_torch.so")


def build_torch_ext(force=False):
    """libnl_svo_torch.so: TORCH_LIBRARY(svo, ...) - the reference's torch.classes.svo.Octree / torch.ops.svo.encode names -
    over the C ABI of libnerfloam_hip.so (host code only: g++ with the torch headers, linked against the library next to it)"""
    deps = [SVO_TORCH_SRC, OUT, os.path.join(HERE, "..", "include", "nerfloam_hip.h")]
    if not force and os.path.exists(SVO_TORCH_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(SVO_TORCH_OUT) for d in deps):
        return SVO_TORCH_OUT
    import torch
    from torch.utils import cpp_extension as ce
    inc = [f"-I{p}" for p in ce.include_paths()] + ["-I" + os.path.join(HERE, "..", "include")]
    libdir = ce.library_paths()[0]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, SVO_TORCH_SRC,
           "-L" + HERE, "-l:libnerfloam_hip.so", "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + libdir, "-o", SVO_TORCH_OUT]
    subprocess.check_call(cmd)
    return SVO_TORCH_OUT


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    tmp = os.path.join(HERE, "build")
    os.makedirs(tmp, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(tmp, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and out:
            sys.stderr.write(out.decode())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_ext(force="--force" in sys.argv))
