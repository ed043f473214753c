"""CPU: the per-pixel algorithmic work `bench.py` divides by (its roofline numerators) against
the figures of the scope contract: training = sample_fwd_bwd + grad_reduce, inference =
conv_pool_fwd + fc_head_fwd, gather bytes per labelled pixel."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("z,infer,train,gather_f32", [
    (176, 702960, 1568520, None),      # KSC
    (200, 809040, 1804680, 14436),     # Indian Pines
    (103, 373880, 834360, 7452),       # Pavia U
    (224, 915120, 2040840, None),      # AVIRIS
])
def test_flops_and_bytes_per_pixel(z, infer, train, gather_f32):
    b = _bench()
    from gcn_amd import lib as L
    d = L.dims(z)
    w = b.kernel_work(d, z)
    assert w["conv_pool_fwd"][1] + w["fc_head_fwd"][1] == infer
    assert w["sample_fwd_bwd"][1] + w["grad_reduce"][1] == train
    assert all(w[k][0] == "mfma" for k in ("conv_pool_fwd", "fc_head_fwd", "sample_fwd_bwd", "grad_reduce"))
    if gather_f32 is not None:
        assert w["gather_tile"] == ("hbm", gather_f32)
