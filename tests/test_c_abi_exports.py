"""CPU: libnerfloam_hip.so loads without a GPU and exports every symbol include/nerfloam_hip.h declares
(no compute calls here); the product refuses to run without a device."""
import os
import re

import pytest

from nerf_loam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("nerfloam_hip.h", "nerfloam_hip_debug.h")      # the product surface; test / profiling / A-B aids (process-global switches)


def declared_symbols(headers=HEADERS):
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(nl_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_the_product_header_declares_no_process_global_switch():
    """the library is thread-safe given distinct streams - every nl_*_set_* (process-global state) lives in the debug header"""
    prod = declared_symbols(HEADERS[:1])
    assert not [s for s in prod if "_set_" in s or "_get_" in s or "selftest" in s], prod
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == sorted(HEADERS)


def test_header_symbols_exported_and_bound():
    lib = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nerfloam_hip.h but not exported"
    assert set(syms) == set(_lib.EXPORTED_SYMBOLS), set(syms) ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.nl_version() >= 100


def test_header_is_plain_c_and_every_entry_links_from_c(tmp_path):
    """the boundary is a C ABI: the header compiles as strict C99 (no C++, no torch types), and a C translation unit that takes
    the address of every declared entry point links against the library (and runs: nl_version, the host octree - no GPU calls)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    syms = declared_symbols()
    src = ['#include "nerfloam_hip.h"', '#include "nerfloam_hip_debug.h"', "#include <stdio.h>", "typedef void (*fn)(void);", "int main(void) {",
           "    fn table[] = {"] + [f"        (fn){s}," for s in syms] + ["    };",
           "    void* t = nl_octree_create(64);", "    int v[6] = {1, 2, 3, 1, 2, 4};",
           "    if (!t || nl_octree_insert(t, v, 2) != 0) return 2;",
           "    long long n = nl_octree_count_nodes(t), leaves = nl_octree_count_leaf_nodes(t);", "    nl_octree_destroy(t);",
           '    printf("%d %d %lld %lld\\n", (int)(sizeof table / sizeof table[0]), nl_version(), n, leaves);', "    return 0;", "}"]
    c = tmp_path / "abi.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe),
           "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    n_syms, version, nodes, leaves = out.stdout.split()
    assert int(n_syms) == len(syms) and int(version) >= 100 and int(leaves) == 2 and int(nodes) > 2


def test_iteration_descriptor_mirror_matches_the_c_struct(tmp_path):
    """nerf_loam_amd._lib.NlIterDesc (ctypes) against NlIterDesc of include/nerfloam_hip.h: same size, same offset for every field
    (a C program prints offsetof for each field name of the mirror)"""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = [n for n, _ in _lib.NlIterDesc._fields_]
    src = ['#include "nerfloam_hip.h"', "#include <stddef.h>", "#include <stdio.h>", "int main(void) {",
           '    printf("%zu", sizeof(NlIterDesc));'] + [f'    printf(" %zu", offsetof(NlIterDesc, {n}));' for n in names] + ["    return 0;", "}"]
    c = tmp_path / "desc.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "desc"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    vals = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(_lib.NlIterDesc)
    assert vals[1:] == [getattr(_lib.NlIterDesc, n).offset for n in names]
    hdr = open(os.path.join(ROOT, "include", "nerfloam_hip.h")).read()
    body = hdr[hdr.index("typedef struct NlIterDesc {"):hdr.index("} NlIterDesc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    declared = re.findall(r"[\*\s,]([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", body)
    assert declared == names, (declared, names)                   # every field of the C struct is mirrored, in order


def test_header_constants_match_binding():
    src = open(os.path.join(ROOT, "include", "nerfloam_hip.h")).read()
    consts = dict(re.findall(r"#define\s+(NL_[A-Z_]+)\s+(\d+)", src))
    assert int(consts["NL_MAX_HITS"]) == _lib.NL_MAX_HITS
    assert int(consts["NL_DEC_PARAMS"]) == _lib.NL_DEC_PARAMS == 16 * 256 + 256 + 256 * 256 + 256 + 256 + 1
    assert int(consts["NL_CNT_INTS"]) * 4 + int(consts["NL_CNT_DOUBLES"]) * 8 == _lib.NL_CNT_BYTES
    assert int(consts["NL_LOSS_SCALARS_BYTES"]) == _lib.NL_LOSS_SCALARS_BYTES


def test_launch_shape_table_is_one_table():
    """the ray-count thresholds that pick a launch shape live in csrc/nl_common.h only; the lanes-per-ray rule the host side follows is the
    library's (a pure host function: no device needed), and no source file repeats a threshold as a literal in a comparison"""
    lib = _lib.lib()
    table = dict(re.findall(r"#define\s+(NL_(?:RAYS|BLOCKS)_[A-Z0-9_]+)\s+(\d+)", open(os.path.join(ROOT, "nerf_loam_amd", "csrc", "nl_common.h")).read()))
    assert set(table) == {"NL_RAYS_ONE_WORKGROUP_SCAN", "NL_RAYS_SINGLE_LAUNCH_SCAN", "NL_RAYS_FUSED_SAMPLER", "NL_RAYS_ISECT_32_LANES",
                          "NL_RAYS_ISECT_16_LANES", "NL_BLOCKS_WIDE_MAP", "NL_RAYS_DECODER_SPLIT"}
    split = int(table["NL_RAYS_DECODER_SPLIT"])
    assert lib.nl_decoder_get_layout() == 0 and [lib.nl_decoder_layout_for(n) for n in (1, 2048, split, split + 1, 131072)] == [1, 1, 1, 0, 0]
    r32, r16, wide = int(table["NL_RAYS_ISECT_32_LANES"]), int(table["NL_RAYS_ISECT_16_LANES"]), int(table["NL_BLOCKS_WIDE_MAP"])
    for n, blocks, want in [(1, 0, 32), (r32, 0, 32), (r32 + 1, 0, 16), (r32 + 1, wide, 32), (r16, wide - 1, 16), (r16, wide, 32), (r16 + 1, wide, 8),
                            (131072, 10 * wide, 8)]:
        assert lib.nl_isect_lanes_for(n, blocks) == want, (n, blocks)
    for f in ("nl_geometry.hip", "nl_iteration.cpp"):
        src = open(os.path.join(ROOT, "nerf_loam_amd", "csrc", f)).read()
        code = re.sub(r"//[^\n]*|/\*.*?\*/", "", src, flags=re.S)
        assert not re.search(r"\bN\s*(<=|>|<|>=)\s*(4096|8192|16384|32768)\b", code), f
    assert "60_000" not in open(os.path.join(ROOT, "nerf_loam_amd", "pipeline.py")).read()


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.NerfLoamHipError):
        _lib.require_gpu()
    from nerf_loam_amd import pipeline
    with pytest.raises(_lib.NerfLoamHipError):
        pipeline.SdfEngine(max_rays=16)
    from nerf_loam_amd import grid
    with pytest.raises(RuntimeError):
        grid.svo_intersect(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.zeros(1, 1, 3), torch.zeros(1, 1, 9, dtype=torch.int32), 0.2, 20)


def test_rccl_binding_resolves_from_the_rccl_torch_loaded():
    """nl_comm_init_rccl (the communicator of the ray-sharded iteration) binds ncclAllGather / ncclAllReduce / ncclGroupStart / ncclGroupEnd
    from the RCCL already in the process - torch's own copy - instead of loading a second one (two RCCLs = two sets of globals).  No
    collective is issued here (no GPU); the world-1 RCCL run is tests/test_gpu_dist_rccl.py."""
    import ctypes
    import torch  # noqa: F401
    import torch.distributed  # noqa: F401
    lib = _lib.lib()
    comm = _lib.NlComm()
    dummy = ctypes.c_void_p(0x1000)
    assert lib.nl_comm_init_rccl(ctypes.byref(comm), dummy, 2, 1) == 0
    assert (comm.world, comm.rank, comm.ctx) == (2, 1, 0x1000)
    assert all(ctypes.cast(f, ctypes.c_void_p).value for f in (comm.all_gather, comm.all_reduce_sum, comm.group_begin, comm.group_end))
    assert lib.nl_comm_init_rccl(ctypes.byref(comm), dummy, 2, 2) != 0 and lib.nl_comm_init_rccl(ctypes.byref(comm), None, 1, 0) != 0
    paths = {line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line}
    assert len(paths) == 1, paths                                    # one RCCL in the process
    assert os.path.dirname(next(iter(paths))) == os.path.join(os.path.dirname(torch.__file__), "lib"), paths
