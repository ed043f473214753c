"""CPU: the C-ABI shared library loads and exports every symbol include/ccnn.h declares."""
import os
import re

import gcn_amd.lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ccnn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ccnn_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    decl = declared_symbols()
    assert len(decl) >= 20
    assert sorted(L.EXPORTS) == decl


def test_library_exports_every_declared_symbol():
    lib = L.load()
    for name in declared_symbols():
        assert getattr(lib, name) is not None


def test_host_only_entry_points_work_without_gpu():
    d = L.dims(176)
    assert (d.re, d.mre, d.pout, d.nparams) == (79, 40, 800, 84853)
    assert d.off_b2 + 13 == d.nparams
    assert "workspace" in L.error_string(-2)
    lib = L.load()
    assert lib.ccnn_mbgd_workspace_bytes(128, 200, 4) > 0
    assert lib.ccnn_forward_workspace_bytes(10, 103, 8) > 0
    assert lib.ccnn_sgd_workspace_bytes(176) > 0
    assert lib.ccnn_mbgd_workspace_bytes(128, 5, 4) == 0      # too few bands
    try:
        L.dims(5)
        raise AssertionError("expected CcnnError")
    except L.CcnnError as e:
        assert e.code == -1
