"""How fast the reference's bundle adjustment amplifies a rounding-level perturbation: the first BA call of tests/test_gpu_sequence.py (1024 rays,
20 iterations, embeddings + decoder trained) run with two partitions of the decoder-gradient sum (256 and 128 slabs: the same per-sample
values bit for bit, the weight gradient re-associated at 1e-8) - per-iteration distance of the decoder gradient and parameters between the runs,
and a repeat of the first run (bit-identical).  GPU only (diagnostic; profiles/experiments/README.md)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pytest.monkeypatch import MonkeyPatch
from nerf_loam_amd import _lib as L, pipeline as P
import test_gpu_sequence as T
L.require_gpu()
NCALLS = int(os.environ.get("NCALLS", "20"))


class Done(Exception):
    pass


def run(n_slabs):
    os.environ["NL_N_SLABS"] = str(n_slabs)
    from nerf_loam_amd import render_helpers as RH
    RH._ENGINES.clear()
    rec = []
    real = P.SdfEngine.run_bound

    def hooked(self, stages=3):
        real(self, stages)
        torch.cuda.synchronize()
        st = self.stats()
        rec.append(dict(P=st["P"], params=self._bound_dec.params.cpu().numpy().copy() if hasattr(self, "_bound_dec") else None,
                        grad=self._bound_dec.grad.cpu().numpy().copy() if hasattr(self, "_bound_dec") else None,
                        adam=int(self.adam_state[0].item())))
        if len(rec) >= NCALLS:
            raise Done()
    real_bind = P.SdfEngine.bind

    def bind(self, m, dec, *a, **k):
        self._bound_dec = dec
        return real_bind(self, m, dec, *a, **k)
    mp = MonkeyPatch()
    mp.setattr(P.SdfEngine, "run_bound", hooked); mp.setattr(P.SdfEngine, "bind", bind)
    try:
        T.test_five_scan_mapping_sequence_matches_the_oracle(mp)
    except Done:
        pass
    finally:
        mp.undo()
    return rec


a = run(256); b = run(128); a2 = run(256)
p0 = a[0]["params"]
for i, (x, y, z) in enumerate(zip(a, b, a2)):
    Tn = -(-x["P"] // 64)
    gn = np.linalg.norm(x["grad"].astype(np.float64))
    print(f"it {i:2d} P {x['P']:6d}/{y['P']:6d} tiles {Tn:4d} adam {x['adam']}/{y['adam']}: grad rel diff {np.linalg.norm((x['grad'] - y['grad']).astype(np.float64)) / gn:.3e} "
          f"(again {np.linalg.norm((x['grad'] - z['grad']).astype(np.float64)) / gn:.1e})  max |params diff| {np.abs(x['params'] - y['params']).max():.3e}  n differ {int((x['params'] != y['params']).sum())}")
