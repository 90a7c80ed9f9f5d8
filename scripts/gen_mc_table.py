"""Generates the 256-case marching-cubes triangle table used by csrc/nl_mesh.hip and oracle/mc_oracle.py.

The reference extracts its mesh with skimage.measure.marching_cubes (Lewiner) per voxel (src/utils/mesh_util.py:145-169); scikit-image is a third-party
dependency that is absent from this image, so its case tables cannot be consulted.  This script DERIVES a table from the cube's geometry instead of
restating one from memory:

  corner c = (x, y, z) bits (c & 1, c >> 1 & 1, c >> 2 & 1); "inside" = value < 0;
  edge   e = 4 * axis + (bit of the lower other axis) + 2 * (bit of the higher other axis), from the corner with bit[axis] = 0 to the one with 1;
  on every face the iso-curve is traced with the inside region on its left (seen from outside the cube); a face whose corners alternate gets two
  segments that cut each INSIDE corner off on its own (the same rule on both cells sharing the face, hence a watertight surface);
  the segments chain into closed loops around the inside regions, every loop is reversed (normals point to the positive side) and fanned into triangles.

Output: nerf_loam_amd/csrc/nl_mc_table.h (C arrays) and oracle/mc_table.json (the same numbers for the numpy oracle).  `python scripts/gen_mc_table.py --check`
regenerates and compares with the committed files."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def corner_pos(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def corner_of(p):
    return p[0] | (p[1] << 1) | (p[2] << 2)


def edge_id(c0, c1):
    p0, p1 = corner_pos(c0), corner_pos(c1)
    axis = [i for i in range(3) if p0[i] != p1[i]]
    assert len(axis) == 1
    a = axis[0]
    others = [i for i in range(3) if i != a]
    return 4 * a + p0[others[0]] + 2 * p0[others[1]]


def edge_corners(e):
    a, idx = e // 4, e % 4
    others = [i for i in range(3) if i != a]
    p = [0, 0, 0]
    p[others[0]] = idx & 1
    p[others[1]] = idx >> 1
    q = list(p)
    q[a] = 1
    return corner_of(p), corner_of(q)


def faces():
    """6 faces: the 4 corners counter-clockwise seen from outside"""
    out = []
    for a in range(3):
        b, c = (a + 1) % 3, (a + 2) % 3            # e_b x e_c = e_a
        for s in (0, 1):
            u, v = (b, c) if s == 1 else (c, b)     # u x v = outward normal
            cyc = []
            for (cu, cv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[a] = s
                p[u] = cu
                p[v] = cv
                cyc.append(corner_of(p))
            out.append(cyc)
    return out


FACES = faces()


def loops_of(cfg):
    inside = [(cfg >> c) & 1 for c in range(8)]
    nxt = {}
    for cyc in FACES:
        fl = [inside[c] for c in cyc]
        for i in range(4):
            # a run of inside corners starts where the counter-clockwise walk ENTERS the inside region: outside corner i-1 -> inside corner i
            if fl[i] and not fl[(i - 1) % 4]:
                e_in = edge_id(cyc[(i - 1) % 4], cyc[i])
                j = i
                while fl[(j + 1) % 4]:
                    j += 1
                    assert j < i + 4
                e_out = edge_id(cyc[j % 4], cyc[(j + 1) % 4])
                assert e_out not in nxt
                nxt[e_out] = e_in                  # the segment runs from where the walk leaves the inside region to where it entered it
    crossed = {e for e in range(12) if inside[edge_corners(e)[0]] != inside[edge_corners(e)[1]]}
    assert set(nxt.keys()) == crossed and set(nxt.values()) == crossed, (cfg, nxt, crossed)
    loops, seen = [], set()
    for e0 in sorted(crossed):
        if e0 in seen:
            continue
        loop, e = [], e0
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == e0 and len(loop) >= 3
        loops.append(loop[::-1])                   # reversed: right-hand normals point to the positive (outside) values
    return loops


def share_a_face(e1, e2):
    """both edges lie in one face of the cube"""
    for a in range(3):
        for s in (0, 1):
            if all((corner_pos(c)[a] == s) for e in (e1, e2) for c in edge_corners(e)):
                return True
    return False


def triangulations(poly):
    """all triangulations of a polygon given as a vertex list (orientation kept)"""
    if len(poly) == 3:
        yield [tuple(poly)]
        return
    # the triangle on the edge (poly[0], poly[-1]) has its apex at some k
    for k in range(1, len(poly) - 1):
        left = [poly[0:k + 1]] if k >= 2 else []
        right = [poly[k:]] if len(poly) - k >= 3 else []
        lts = list(triangulations(left[0])) if left else [[]]
        rts = list(triangulations(right[0])) if right else [[]]
        for lt in lts:
            for rt in rts:
                yield lt + [(poly[0], poly[k], poly[-1])] + rt


def best_triangulation(loop):
    """the triangulation with the fewest diagonals lying in a face of the cube (such a diagonal can coincide with a diagonal of the neighbouring cell: four
    triangles would meet along it); among those the first in enumeration order"""
    best, best_score = None, None
    for t in triangulations(loop):
        sides = {(loop[i], loop[(i + 1) % len(loop)]) for i in range(len(loop))}
        score = 0
        for tri in t:
            for i in range(3):
                a, b = tri[i], tri[(i + 1) % 3]
                if (a, b) not in sides and (b, a) not in sides and share_a_face(a, b):
                    score += 1
        if best is None or score < best_score:
            best, best_score = t, score
    return best, best_score


def table():
    tris, in_face = [], 0
    for cfg in range(256):
        t = []
        for loop in loops_of(cfg):
            tt, sc = best_triangulation(loop)
            in_face += sc
            # every triangle keeps the loop's orientation (its vertices appear in loop order)
            for tri in tt:
                idx = [loop.index(e) for e in tri]
                assert sum(idx[i] < idx[(i + 1) % 3] for i in range(3)) == 2, (loop, tri)
                t.append(tri)
        tris.append(t)
    print(f"diagonals lying in a cube face (counted twice each): {in_face}")
    return tris


def render_header(tris, max_t):
    lines = ["// GENERATED by scripts/gen_mc_table.py - do not edit.  Marching-cubes cases derived from the cube's geometry (see the script's docstring):",
             "// corner c = x | y << 1 | z << 2, inside = value < 0, edge e = 4 * axis + the other two axes' bits of its lower corner.",
             "#pragma once",
             f"#define NL_MC_MAX_TRIS {max_t}",
             "static __device__ const unsigned char NL_MC_NTRI[256] = {" + ", ".join(str(len(t)) for t in tris) + "};",
             f"static __device__ const signed char NL_MC_TRI[256][{3 * max_t}] = {{"]
    for cfg, t in enumerate(tris):
        flat = [e for tri in t for e in tri] + [-1] * (3 * (max_t - len(t)))
        lines.append("    {" + ", ".join(f"{e:2d}" for e in flat) + "}" + ("," if cfg < 255 else ""))
    lines.append("};")
    return "\n".join(lines) + "\n"


def main():
    tris = table()
    max_t = max(len(t) for t in tris)
    hdr = render_header(tris, max_t)
    js = json.dumps({"max_tris": max_t, "tris": [[list(tri) for tri in t] for t in tris]})
    ph = os.path.join(ROOT, "nerf_loam_amd", "csrc", "nl_mc_table.h")
    pj = os.path.join(ROOT, "oracle", "mc_table.json")
    if "--check" in sys.argv:
        ok = open(ph).read() == hdr and json.load(open(pj)) == json.loads(js)
        print("tables match" if ok else "TABLES DIFFER")
        sys.exit(0 if ok else 1)
    open(ph, "w").write(hdr)
    open(pj, "w").write(js + "\n")
    print(f"max triangles per cell {max_t}; total triangles over the 256 cases {sum(len(t) for t in tris)}")


if __name__ == "__main__":
    main()
