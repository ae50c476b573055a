#!/usr/bin/env python3
"""Two images of tools/lab_render.py: relative RMSE, share of identical pixels, largest differences. usage: lab_compare.py A.npy B.npy"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]).astype(np.float64), np.load(sys.argv[2]).astype(np.float64)
d = a[..., :3] - b[..., :3]
same = (a == b).all(axis=2)
print(f"identical pixels {same.mean():.6f}, relRMSE {np.sqrt((d ** 2).mean()) / max(b[..., :3].mean(), 1e-30):.3e}, mean shift {d.mean() / max(b[..., :3].mean(), 1e-30):+.3e}, "
      f"alpha identical {bool(((a[..., 3] > 0) == (b[..., 3] > 0)).all())}")
