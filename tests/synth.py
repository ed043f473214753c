"""Seeded synthetic hyperspectral scenes shaped like the reference's datasets
(test infrastructure; also used by bench.py).  Values uniform in [0,1), labels
laid out as axis-aligned rectangular patches of the 13 classes, 0 = unlabelled."""
import numpy as np

# name: (x, y, z, labelled pixels of the real scene)
SCENES = {
    "ksc": (512, 614, 176, 5211),
    "indian_pines": (145, 145, 200, 10249),
    "pavia_u": (610, 340, 103, 42776),
    "aviris": (2048, 2048, 224, None),      # fully labelled
}


def make_labels(x, y, n_labelled, rng, classes=13):
    """Column-major label image as a flat float64 array of x*y values (index r + c*x)."""
    lab = np.zeros((y, x), dtype=np.float64)          # lab[c, r]
    if n_labelled is None or n_labelled >= x * y:
        # fully labelled: class stripes of rectangles
        bx, by = max(x // 8, 1), max(y // 8, 1)
        cc, rr = np.meshgrid(np.arange(y) // by, np.arange(x) // bx, indexing="ij")
        return ((cc * 3 + rr) % classes + 1).astype(np.float64).ravel()
    placed = 0
    guard = 0
    while placed < n_labelled and guard < 100000:
        guard += 1
        h = int(rng.integers(2, max(3, min(24, y // 2))))
        w = int(rng.integers(2, max(3, min(24, x // 2))))
        c0 = int(rng.integers(0, y - h + 1))
        r0 = int(rng.integers(0, x - w + 1))
        patch = lab[c0:c0 + h, r0:r0 + w]
        free = patch == 0
        room = n_labelled - placed
        if free.sum() > room:
            # trim the patch so the count lands exactly
            flat = np.flatnonzero(free.ravel())[:room]
            m = np.zeros(patch.size, dtype=bool)
            m[flat] = True
            free = m.reshape(patch.shape)
        patch[free] = float(rng.integers(1, classes + 1))
        placed += int(free.sum())
    return lab.ravel()


def make_scene(x, y, z, dtype, rng):
    """Flat [z][y][x] array (x fastest)."""
    return rng.random(x * y * z, dtype=np.float64).astype(dtype)


def small_scene(x, y, z, seed, dtype=np.float64, border=True):
    """Small scene whose labelled set includes every border pixel and corner."""
    rng = np.random.default_rng(seed)
    lab = np.zeros((y, x), dtype=np.float64)
    if border:
        lab[0, :] = lab[-1, :] = 1
        lab[:, 0] = lab[:, -1] = 1
    inner = rng.random((y, x)) < 0.5
    lab[inner] = 1
    lab[lab > 0] = rng.integers(1, 14, size=int((lab > 0).sum())).astype(np.float64)
    return make_scene(x, y, z, dtype, rng), lab.ravel()
