#!/usr/bin/env python3
"""Builds an experimental copy of the library with extra compiler flags (diagnostics only):
    build_variant.py NAME -DFOO=1 ...   ->  tools/_variants/libccnn_NAME.so
Run a tool against it with CCNN_LIB_PATH=tools/_variants/libccnn_NAME.so."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_variants", f"libccnn_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(g.CSRC, s) for s in g.HIP_SOURCES]
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *g.HIPCC_FLAGS, *flags, "-o", out, *srcs])
print(out)
