"""Writes a synthetic DATASET.mat (DataSet x-by-y-by-z, labels x-by-y) for command-line runs.
    make_mat_dataset.py OUT.mat X Y Z [single|double] [fraction labelled, default 1.0] [classes, default 13]"""
import sys

import numpy as np
import scipy.io as sio

out, x, y, z = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = np.float32 if len(sys.argv) > 5 and sys.argv[5] == "single" else np.float64
frac = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
classes = int(sys.argv[7]) if len(sys.argv) > 7 else 13
rng = np.random.default_rng(5)
ds = rng.random((x, y, z), dtype=np.float32).astype(dtype)
lab = ((np.arange(x)[:, None] // 8 * 3 + np.arange(y)[None, :] // 8) % classes + 1).astype(np.float64)
lab[rng.random((x, y)) >= frac] = 0
sio.savemat(out, {"DataSet": ds, "labels": lab}, do_compression=False)
print(out, ds.shape, ds.dtype, int((lab > 0).sum()), "labelled")
