#!/usr/bin/env python3
"""Build the REFERENCE `grid` extension (third_party/sparse_voxels: svo_intersect, inverse_cdf_sampling, ...) for gfx950
into oracle/_ref/ - TEST INFRASTRUCTURE ONLY.

Purpose: pin oracle/nl_oracle.c's restatement of the two CUDA kernels (intersect_gpu.cu:193-272, sample_gpu.cu:133-239)
against the reference's own code, run on the GPU box (tests/test_gpu_reference_grid.py).  The product never loads these
modules: nothing under nerf_loam_amd/ may import oracle/, and the north_star rules out hipify / dual paths for the
product - this is the checker, built from the sources where they lie under /root/reference.

Recipe: torch.utils.cpp_extension.load (its hipify pass translates the .cu/.cpp on the fly) on the reference's six source
files.  hipify writes the translated files NEXT TO its inputs and /root/reference is read-only for us, so the files are
staged in a throw-away directory under $TMPDIR for the duration of the build and removed afterwards; nothing is copied
into the repository, only the two built modules land in oracle/_ref/ (git-ignored, ships to the GPU box):

    grid_ref.so      the reference's flags (setup.py: -O2; hipcc's default -ffp-contract=fast)
    grid_ref_nc.so   the same sources with -ffp-contract=off: separates "different algorithm" from "FMA contraction"
"""
import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NL_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "third_party", "sparse_voxels")
OUT = os.path.join(HERE, "_ref")
VARIANTS = {"grid_ref": [], "grid_ref_nc": ["-ffp-contract=off"]}


def build(force=False, verbose=False):
    if not os.path.isdir(SRC):
        print(f"reference not present at {REF} - skipping the grid_ref build")
        return []
    os.makedirs(OUT, exist_ok=True)
    todo = [n for n in VARIANTS if force or not os.path.exists(os.path.join(OUT, n + ".so"))]
    if not todo:
        return [os.path.join(OUT, n + ".so") for n in VARIANTS]
    os.environ["PYTORCH_ROCM_ARCH"] = "gfx950"
    from torch.utils.cpp_extension import load
    stage = tempfile.mkdtemp(prefix="nl_grid_ref_")
    try:
        shutil.copytree(os.path.join(SRC, "src"), os.path.join(stage, "src"))
        shutil.copytree(os.path.join(SRC, "include"), os.path.join(stage, "include"))
        srcs = sorted(glob.glob(os.path.join(stage, "src", "*.cpp")) + glob.glob(os.path.join(stage, "src", "*.cu")))
        for name in todo:
            bdir = os.path.join(stage, "build_" + name)
            os.makedirs(bdir)
            load(name=name, sources=srcs, extra_include_paths=[os.path.join(stage, "include")], with_cuda=True,
                 extra_cflags=["-O2", "-w"], extra_cuda_cflags=["-O2", "-w"] + VARIANTS[name], build_directory=bdir,
                 verbose=verbose, is_python_module=False)
            shutil.copy(os.path.join(bdir, name + ".so"), os.path.join(OUT, name + ".so"))
            print("built", os.path.join(OUT, name + ".so"))
    finally:
        shutil.rmtree(stage, ignore_errors=True)
    return [os.path.join(OUT, n + ".so") for n in VARIANTS]


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
