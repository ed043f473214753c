"""CPU: the CLI's own MAT-file (Level 5) reader/writer against scipy.io, through the
`gcn --mat-info / --mat-rewrite` utilities (they run before any device is touched)."""
import os
import subprocess

import numpy as np
import pytest
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GCN = os.path.join(ROOT, "gcn_amd", "bin", "gcn")

pytestmark = pytest.mark.skipif(not os.path.exists(GCN), reason="gcn CLI not built")


def info(path):
    r = subprocess.run([GCN, f"--mat-info={path}"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    out = {}
    for line in r.stdout.splitlines():
        parts = line.split()
        out[parts[0]] = (parts[1], tuple(int(v) for v in parts[2:-1]), float(parts[-1].split("=")[1]))
    return out


@pytest.mark.parametrize("compress", [False, True])
def test_reader_handles_classes_compression_and_3d(tmp_path, compress):
    rng = np.random.default_rng(0)
    ds64 = rng.random((7, 6, 5))
    ds32 = ds64.astype(np.float32)
    lab = (np.arange(42).reshape(7, 6) % 14).astype(np.float64)
    p = str(tmp_path / "a.mat")
    sio.savemat(p, {"DataSet": ds64, "single3d": ds32, "labels": lab, "lab_u8": lab.astype(np.uint8),
                    "lab_i16": lab.astype(np.int16), "abc": np.array([[1.5]]), "name_longer_than_8": lab[:2]},
                do_compression=compress)
    got = info(p)
    assert got["DataSet"][:2] == ("double", (7, 6, 5)) and abs(got["DataSet"][2] - ds64.sum()) < 1e-9
    assert got["single3d"][:2] == ("single", (7, 6, 5)) and abs(got["single3d"][2] - ds32.astype(np.float64).sum()) < 1e-3
    for k in ("labels", "lab_u8", "lab_i16"):
        assert got[k][1] == (7, 6) and got[k][2] == lab.sum(), k           # integer classes come back as doubles
    assert got["abc"][2] == 1.5 and got["name_longer_than_8"][1] == (2, 6)


def test_writer_round_trips_through_scipy(tmp_path):
    rng = np.random.default_rng(1)
    mats = {"filters": rng.normal(size=(171, 20)), "bias0": rng.normal(size=(20, 1)), "b": rng.normal(size=(3, 1)),
            "loss": rng.random((300, 1)), "omega1_long_name": rng.normal(size=(100, 40))}
    src, dst = str(tmp_path / "s.mat"), str(tmp_path / "d.mat")
    sio.savemat(src, mats)
    r = subprocess.run([GCN, f"--mat-rewrite={src}:{dst}"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    back = sio.loadmat(dst)
    for k, v in mats.items():
        assert np.array_equal(back[k], v), k


def test_reader_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "v73.mat"
    p.write_bytes(b"MATLAB 7.3 MAT-file, Platform: GLNXA64" + b" " * 200)
    r = subprocess.run([GCN, f"--mat-info={p}"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "v7.3" in r.stderr
    q = tmp_path / "junk.mat"
    q.write_bytes(b"not a mat file")
    r = subprocess.run([GCN, f"--mat-info={q}"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
    r = subprocess.run([GCN, f"--mat-info={tmp_path / 'absent.mat'}"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "cannot open" in r.stderr


def stream(path, var, chunk=None):
    spec = f"{path}:{var}" + (f":{chunk}" if chunk else "")
    r = subprocess.run([GCN, f"--mat-stream={spec}"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    parts = r.stdout.split()
    kv = dict(p.split("=") for p in parts if "=" in p)
    return parts[1], tuple(int(v) for v in parts[2:-2]), float(kv["sum"]), int(kv["peak_rss_kb"])


@pytest.mark.parametrize("compress", [False, True])
def test_streaming_reader_matches_scipy(tmp_path, compress):
    """The reader that hands a variable out a chunk at a time (what the trainer feeds the devices
    from): every class, compressed or not, chunk sizes that do and do not divide the array, a
    variable that is not the first in the file, names short enough for the small-element form."""
    rng = np.random.default_rng(3)
    ds64 = rng.random((9, 8, 7))
    lab = (np.arange(72).reshape(9, 8) % 14).astype(np.float64)
    p = str(tmp_path / "s.mat")
    sio.savemat(p, {"junk": rng.random((5, 5)), "DataSet": ds64, "ds32": ds64.astype(np.float32), "labels": lab,
                    "lu8": lab.astype(np.uint8), "li16": lab.astype(np.int16), "li32": lab.astype(np.int32),
                    "a_name_longer_than_eight": lab[:3]}, do_compression=compress)
    for chunk in (None, 1, 7, 100, 504, 1000):
        cls, dims, total, _ = stream(p, "DataSet", chunk)
        assert cls == "double" and dims == (9, 8, 7) and abs(total - ds64.sum()) < 1e-9, chunk
    cls, dims, total, _ = stream(p, "ds32", 50)
    assert cls == "single" and dims == (9, 8, 7) and abs(total - ds64.astype(np.float32).astype(np.float64).sum()) < 1e-9
    for k in ("labels", "lu8", "li16", "li32"):
        assert stream(p, k, 13)[1:3] == ((9, 8), lab.sum()), k
    assert stream(p, "a_name_longer_than_eight")[1] == (3, 8)
    r = subprocess.run([GCN, f"--mat-stream={p}:absent"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no variable" in r.stderr


def test_streaming_reader_keeps_host_memory_bounded(tmp_path):
    """A 96 MB single array, compressed: the streaming reader's peak resident set stays far below
    the array (the whole-file reader needs the file, the inflated element and the typed copy)."""
    rng = np.random.default_rng(4)
    big = rng.random((256, 256, 368), dtype=np.float32)
    p = str(tmp_path / "big.mat")
    sio.savemat(p, {"labels": np.ones((256, 256)), "DataSet": big}, do_compression=True)
    cls, dims, total, rss_kb = stream(p, "DataSet")
    assert cls == "single" and dims == (256, 256, 368)
    assert abs(total - big.astype(np.float64).sum()) < 1e-6 * big.size
    assert rss_kb * 1024 < 0.3 * big.nbytes, rss_kb


def test_reader_rejects_out_of_range_dimensions(tmp_path):
    """Dimensions come from the file: a negative one must be refused, not used to size a buffer."""
    p = str(tmp_path / "d.mat")
    sio.savemat(p, {"DataSet": np.zeros((4, 3, 2))})
    raw = bytearray(open(p, "rb").read())
    at = raw.find(np.array([4, 3, 2], dtype="<i4").tobytes())
    assert at > 0
    raw[at:at + 4] = np.array([-4], dtype="<i4").tobytes()
    q = tmp_path / "neg.mat"
    q.write_bytes(bytes(raw))
    for flag in (f"--mat-info={q}", f"--mat-stream={q}:DataSet"):
        r = subprocess.run([GCN, flag], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "dimensions" in r.stderr, (flag, r.stderr)


def test_streaming_reader_rejects_bad_chunk_counts(tmp_path):
    """CHUNK of --mat-stream: zero, negative or non-numeric is a usage error (not an endless loop or an
    allocation of 2^64 values); a count beyond the variable is clamped to it."""
    p = str(tmp_path / "c.mat")
    a = np.arange(24, dtype=np.float64).reshape(4, 3, 2)
    sio.savemat(p, {"DataSet": a})
    for bad in ("0", "-5", "abc", "7x"):
        r = subprocess.run([GCN, f"--mat-stream={p}:DataSet:{bad}"], capture_output=True, text=True, timeout=20)
        assert r.returncode == 1 and "CHUNK" in r.stderr, (bad, r.stderr)
    cls, dims, total, rss_kb = stream(p, "DataSet", 10 ** 15)
    assert dims == (4, 3, 2) and total == a.sum() and rss_kb < 200 * 1024
