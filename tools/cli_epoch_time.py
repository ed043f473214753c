"""End-to-end time of the `gcn` command line on a synthetic scene of a real scene's shape
(written as an uncompressed .mat under /tmp).  usage: cli_epoch_time.py [pavia_u|ksc|indian_pines] [variant] [epochs] [batch]"""
import os, subprocess, sys, tempfile, time
import numpy as np
import scipy.io as sio
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
name = sys.argv[1] if len(sys.argv) > 1 else "pavia_u"
variant = sys.argv[2] if len(sys.argv) > 2 else "mbgd_fp32"
epochs = sys.argv[3] if len(sys.argv) > 3 else "5"
batch = sys.argv[4] if len(sys.argv) > 4 else "128"
x, y, z, nl = synth.SCENES[name]
rng = np.random.default_rng(3)
labels = synth.make_labels(x, y, nl, rng)
dt = np.float32 if variant.endswith("32") else np.float64
scene = synth.make_scene(x, y, z, dt, rng)
tmp = tempfile.mkdtemp(dir="/tmp")
os.makedirs(os.path.join(tmp, "model"), exist_ok=True)
path = os.path.join(tmp, "DATASET.mat")
sio.savemat(path, {"DataSet": scene.reshape(z, y, x).transpose(2, 1, 0), "labels": labels.reshape(y, x).T})
args = [os.path.join(ROOT, "gcn_amd", "bin", "gcn"), path, "0", f"--variant={variant}", f"--epochs={epochs}",
        f"--batch={batch}", "--seed=5"]
t0 = time.perf_counter()
r = subprocess.run(args, cwd=tmp, capture_output=True, text=True, timeout=600)
wall = time.perf_counter() - t0
print(r.stdout[-1500:])
print(r.stderr[-500:])
print(f"{name} {variant} epochs={epochs} batch={batch}: process wall {wall:.2f} s, rc {r.returncode}")
