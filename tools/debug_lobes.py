"""Prints device-vs-golden rows of one surface lobe kind (debug aid for tests/test_gpu_parity.py)."""
import os, sys
import numpy as np
import torch  # noqa: F401  (first: its HIP runtime must be the one libyhair binds to)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python"))
import yhair_capi as yh
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = np.load(os.path.join(ROOT, "tests", "golden", "lobes.npz"))
ctx = yh.Context(0)
got = ctx.surface_lobe(kind, g["params"], g["normal"], g["wo"], g["wi"], g["rn"])
want = g[f"lobe_{kind}"]
rel = np.abs(got[:, :4] - want[:, :4]) / np.maximum(np.abs(want[:, :4]), 1e-7)
bad = np.flatnonzero(~(rel <= 1e-4).all(1))
print("kind", kind, "bad rows", len(bad), bad[:20])
for r in bad[:8]:
    print(r, "params", g["params"][r], "n.o", float(np.dot(g["normal"][r], g["wo"][r])), "\n   got ", got[r], "\n   want", want[r])
