"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/bvgraph_hip.h
declares (no compute calls: there is no GPU here); host-only entry points behave; without a device
the compute entry points fail loudly with BVG_E_HIP (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, CNR


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "bvgraph_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bvg_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(W):
    lib = C.CDLL(W.build())
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert lib.bvg_abi_version() == 3


def test_struct_layouts_match_header(W):
    assert C.sizeof(W.Params) == 56 and C.sizeof(W.ScanResult) == 72 and C.sizeof(W.Tuning) == 20


def test_host_side_entry_points(W, oracle):
    p = W.parse_properties(open(CNR + ".properties").read())
    assert (p.nodes, p.arcs, p.window_size, p.min_interval_length, p.residual_coding) == (325557, 3216152, 7, 3, W.ZETA)
    off = W.decode_offsets(open(CNR + ".offsets", "rb").read(), p.nodes, p.offset_coding)
    assert np.array_equal(off, oracle.Graph.load(CNR).offsets())
    assert W.arc_mix(12345, 678) == oracle.mix(12345, 678)
    with pytest.raises(W.IOException):
        W.parse_properties("graphclass=x.Y\nnodes=1\n")
    with pytest.raises(W.IOException):
        W.parse_properties("graphclass=it.unimi.dsi.big.webgraph.BVGraph\nnodes=1\ncompressionflags=BLOCKS_ZETA\n")
    with pytest.raises(W.EOFException):
        W.decode_offsets(b"\x00\x00", 5)


def test_compute_fails_loudly_without_a_gpu(W):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(W.DeviceError):
        W.BVGraph.load(CNR)
    with pytest.raises(W.IOException):
        W.BVGraph.load(os.path.join(ROOT, "tests", "golden", "does-not-exist"))


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "webgraph-big_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "bvg_oracle" not in src and "bvgo_" not in src and "from oracle" not in src, os.path.join(dirpath, f)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/bvgraph_hip.h must compile as C99 (no C++ in the signatures)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "%s"\nint main(void) { bvg_params p; bvg_default_params(&p); return 0; }\n' % os.path.join(ROOT, "include", "bvgraph_hip.h"))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "abi.o")])
