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


def test_step_path_choice_needs_no_gpu(monkeypatch):
    """ccnn_mbgd_path and the workspace size that follows it depend on the batch, the geometry and
    the element size only (never on a device), and CCNN_MBGD_PATH overrides the choice."""
    lib = L.load()
    monkeypatch.delenv("CCNN_MBGD_PATH", raising=False)
    assert lib.ccnn_mbgd_path(128, 103, 4) == 0            # the reference's batch sizes: one workgroup per sample
    assert lib.ccnn_mbgd_path(2048, 103, 4) == 1           # large batches: a tile of samples per workgroup
    assert lib.ccnn_mbgd_path(512, 103, 4) == 0 and lib.ccnn_mbgd_path(600, 103, 4) == 1
    assert lib.ccnn_mbgd_path(512, 200, 4) == 0 and lib.ccnn_mbgd_path(600, 200, 4) == 1
    assert lib.ccnn_mbgd_path(256, 176, 8) == 0 and lib.ccnn_mbgd_path(300, 176, 8) == 1
    assert lib.ccnn_mbgd_path(2048, 448, 8) == 2           # no tile fits the LDS: one launch per layer
    assert lib.ccnn_mbgd_path(0, 103, 4) < 0 and lib.ccnn_mbgd_path(128, 103, 2) < 0
    fused = lib.ccnn_mbgd_workspace_bytes(2048, 103, 4)
    monkeypatch.setenv("CCNN_MBGD_PATH", "fused")
    assert lib.ccnn_mbgd_path(2048, 103, 4) == 0
    assert lib.ccnn_mbgd_workspace_bytes(2048, 103, 4) > fused    # a conv-gradient row per sample, not per tile
    monkeypatch.setenv("CCNN_MBGD_PATH", "layers")
    assert lib.ccnn_mbgd_path(128, 103, 4) == 2


def test_tile_plans_fit_the_lds_for_every_geometry():
    """tile_plan (the host-side choice of waves, samples per tile and delta-image chunks of the
    sample-tile kernel) never asks for more LDS than a workgroup may have, covers the batch, and
    keeps one sample per wave - for every band count and both element sizes."""
    import ctypes as C

    class Plan(C.Structure):
        _fields_ = [("nwv", C.c_int), ("ms", C.c_int), ("nch", C.c_int), ("smem", C.c_size_t), ("tiles", C.c_int)]

    lib = L.load()
    fits = 0
    for es, sym in ((4, "_ZN4ccnn9tile_planIfEEbiRK9ccnn_dimsPNS_8TilePlanE"), (8, "_ZN4ccnn9tile_planIdEEbiRK9ccnn_dimsPNS_8TilePlanE")):
        fn = getattr(lib, sym)
        fn.restype = C.c_bool
        for z in range(21, 449):
            d = L.dims(z)
            for B in (1, 7, 256, 600, 2048, 5000):
                p = Plan()
                if not fn(B, C.byref(d), C.byref(p)):
                    assert lib.ccnn_mbgd_path(B, z, es) != 1      # no tile: the library must not pick that path
                    continue
                fits += 1
                assert p.smem <= 160 * 1024, (z, B, es, p.smem)
                assert p.nwv in (4, 8, 16) and 1 <= p.ms <= min(p.nwv, 8) and p.nch in (1, 2, 4, 8)
                assert p.tiles == (B + p.ms - 1) // p.ms
    assert fits > 3000


def test_integration_sketch_compiles_against_the_header(tmp_path):
    """Every ```cpp block of INTEGRATION.md that includes <ccnn.h> is compiled (-fsyntax-only) against
    include/ccnn.h: the document a maintainer copies from cannot drift from the boundary it describes."""
    import shutil
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```cpp\n(.*?)```", text, flags=re.S) if "#include <ccnn.h>" in b]
    assert blocks, "INTEGRATION.md lost its compilable sketch"
    cxx = shutil.which("g++") or "g++"     # host code only: the sketch is what a plain C++ trainer would hold
    for i, b in enumerate(blocks):
        src = tmp_path / f"sketch{i}.cpp"
        src.write_text(b)
        r = subprocess.run([cxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
                            "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
