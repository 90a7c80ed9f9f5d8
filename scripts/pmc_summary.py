"""Condense the three rocprofv3 --pmc passes of scripts/gpu_pmc.sh into one per-kernel JSON (profiles/rNN_*_pmc_summary.json).

  python scripts/pmc_summary.py gpurun_out/<tag> profiles/r01_g_pmc_summary.json

Per kernel (averages per launch; for the full-scan bench launches only the LARGEST launches of a kernel are kept, the
pose-refine section of bench.py launches the same kernels on 2048 rays):
  xcd_cycles          GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
  mfma_busy_frac      SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x xcd_cycles)
  mfma_mops_f32/bf16  SQ_INSTS_VALU_MFMA_MOPS_*
  hbm_bytes_per_launch (2 x FETCH_SIZE + WRITE_SIZE) x 1 KB: FETCH_SIZE under-reports by 2x on gfx950 (MI355X_MICROARCH.md)
"""
import collections, csv, json, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def big(values, ref=None):
    """mean over the launches whose value is within 2x of the largest (drops the small pose-refine launches)"""
    if not values:
        return None
    top = max(values)
    keep = [v for v in values if v * 2 >= top] if top > 0 else values
    return sum(keep) / len(keep)


def main(prefix, out):
    mf, fe, wr = load(prefix + "_mfma_counters.csv"), load(prefix + "_fetch_counters.csv"), load(prefix + "_write_counters.csv")
    try:
        m16 = load(prefix + "_mfma16_counters.csv")               # (round 5: SQ_INSTS_VALU_MFMA_MOPS_F16, its own pass)
    except OSError:
        m16 = {}
    res = {}
    for k in sorted(mf):
        if not k.startswith("k_"):
            continue
        gui = mf[k]["GRBM_GUI_ACTIVE"]
        top = max(gui)
        idx = [i for i, v in enumerate(gui) if v * 2 >= top]
        sel = lambda arr: (sum(arr[i] for i in idx) / len(idx)) if arr and len(arr) == len(gui) else None
        xcd = sel(gui) / 8.0
        busy = sel(mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        f, w_ = big(fe[k].get("FETCH_SIZE", [])), big(wr[k].get("WRITE_SIZE", []))
        res[k] = {"launches_averaged": len(idx), "xcd_cycles": xcd,
                  "mfma_busy_frac": (busy / (1024.0 * xcd)) if busy is not None and xcd else None,
                  "mfma_mops_f32": sel(mf[k].get("SQ_INSTS_VALU_MFMA_MOPS_F32", [])),
                  "mfma_mops_bf16": sel(mf[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [])),
                  "mfma_mops_f16": big(m16.get(k, {}).get("SQ_INSTS_VALU_MFMA_MOPS_F16", [])) if m16 else None,
                  "fetch_kb": f, "write_kb": w_,
                  "hbm_bytes_per_launch": (2 * f + w_) * 1024.0 if f is not None and w_ is not None else None}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k.ljust(28), {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
