"""CPU: libsfnative.so builds for gfx950, loads, and exports every symbol include/sfnative.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from util import ROOT


def test_build_and_symbols():
    from streamingflow_amd import build, _lib
    path = build.build()
    assert os.path.exists(path)
    header = open(os.path.join(ROOT, "include", "sfnative.h")).read()
    declared = set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", header))
    declared -= {"sf_status", "sf_abi_check_header"}      # the enum tag; a static inline of the header
    h = ctypes.CDLL(path)
    missing = [n for n in sorted(declared) if not hasattr(h, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()
    assert lib.sf_version() >= 110 and lib.sf_abi_version() == _lib.SF_ABI_VERSION
    assert lib.sf_status_string(-2) == b"workspace too small"
    # workspace sizing is host arithmetic only
    assert lib.sf_dual_cell_ws_bytes(64, 1, 50, 50) >= 10 * 64 * 2500 * 4


def test_struct_sizes_match_header_layout():
    from streamingflow_amd import _lib
    assert ctypes.sizeof(_lib.ConvW) == 3 * 8 + 12 * 4 + 2 * 8  # 3 pointers + 11 int32 + 1 reserved + the optional split-bf16 and Winograd weight pointers
    assert ctypes.sizeof(_lib.DualW) == 13 * ctypes.sizeof(_lib.ConvW) + 16      # + gates1_x / gates1_s (round 3), tg7_h / tg7_r (round 6)
    assert ctypes.sizeof(_lib.GruW) == 3 * ctypes.sizeof(_lib.ConvW)
    assert ctypes.sizeof(_lib.ResW) == 3 * ctypes.sizeof(_lib.ConvW)


def _header_sizes(tmp_path):
    """sizeof of every public struct as gcc sees include/sfnative.h (SF_STRUCT_* order) + SF_ABI_VERSION."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "sfnative.h"\nint main(void) {\n'
                   '  printf("%d", SF_ABI_VERSION);\n'
                   + "".join(f'  printf(" %zu", sizeof({t}));\n' for t in
                             ("sf_conv_w", "sf_gru_w", "sf_dual_w", "sf_res_w", "sf_pmodel_w", "sf_encoder_w", "sf_decoder_w",
                              "sf_convnext_w", "sf_deeplab_w", "sf_bottleneck_w", "sf_bottle_w"))
                   + '  return SF_STRUCT_COUNT == 11 ? 0 : 1;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return int(out[0]), [int(v) for v in out[1:]]


def test_abi_guard_header_library_binding_agree(tmp_path):
    """The header (compiled by gcc here), the library and the ctypes binding agree on every public struct size, and
    the library refuses a stale caller (a 72-byte sf_conv_w would have been read past its end)."""
    import pytest
    from streamingflow_amd import _lib
    abi, sizes = _header_sizes(tmp_path)
    lib = _lib.lib()
    assert abi == _lib.SF_ABI_VERSION == lib.sf_abi_version()
    assert sizes == [lib.sf_abi_sizeof(i) for i in range(len(sizes))] == [ctypes.sizeof(t) for t in _lib.ABI_STRUCTS]
    assert lib.sf_abi_sizeof(len(sizes)) == 0
    ok = (ctypes.c_size_t * len(sizes))(*sizes)
    assert lib.sf_abi_check(abi, ok, len(sizes)) == 0
    assert lib.sf_abi_check(abi - 1, ok, len(sizes)) == -1
    stale = (ctypes.c_size_t * len(sizes))(*([72] + sizes[1:]))
    assert lib.sf_abi_check(abi, stale, len(sizes)) == -1
    # a binding with a stale struct is refused at load time
    real, _lib._LIB = _lib._LIB, None
    old = _lib.ABI_STRUCTS

    class StaleConvW(ctypes.Structure):
        _fields_ = _lib.ConvW._fields_[:-2]
    try:
        _lib.ABI_STRUCTS = (StaleConvW,) + old[1:]
        with pytest.raises(RuntimeError, match="ABI mismatch"):
            _lib.lib()
    finally:
        _lib.ABI_STRUCTS, _lib._LIB = old, real


def test_integration_md_binding_matches_header(tmp_path):
    """The ctypes stub printed in INTEGRATION.md is executed as written and its structs are compared with the header's."""
    from streamingflow_amd import build
    build.build()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "class ConvW" in b]
    assert len(stub) == 1
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)      # the stub loads "streamingflow_amd/libsfnative.so" relative to the checkout
    try:
        exec(compile(stub[0], "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    abi, sizes = _header_sizes(tmp_path)
    names = ("sf_conv_w", "sf_gru_w", "sf_dual_w", "sf_res_w", "sf_pmodel_w", "sf_encoder_w", "sf_decoder_w",
             "sf_convnext_w", "sf_deeplab_w", "sf_bottleneck_w", "sf_bottle_w")
    shown = {"ConvW": "sf_conv_w", "DualW": "sf_dual_w"}
    for cls, cname in shown.items():
        assert ctypes.sizeof(ns[cls]) == sizes[names.index(cname)], (cls, ctypes.sizeof(ns[cls]), sizes[names.index(cname)])
    # field by field against the product binding (names, types, order)
    from streamingflow_amd import _lib
    assert [(n, t) if not hasattr(t, "_fields_") else (n, "ConvW") for n, t in ns["ConvW"]._fields_] == \
           [(n, t) if not hasattr(t, "_fields_") else (n, "ConvW") for n, t in _lib.ConvW._fields_]
    assert [n for n, _ in ns["DualW"]._fields_] == [n for n, _ in _lib.DualW._fields_]
    assert ns["ABI_OK"] is True      # the stub's own handshake with the library ran


def test_state_dict_keys_match_reference_fixture():
    import json
    import streamingflow_amd as sfa
    from util import make_cfg
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys_c8.json")))
    net = sfa.FuturePredictionODE(8, 8, 4, make_cfg(8))
    got = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert got == want and len(got) == 297


def test_no_cpu_fallback():
    import pytest
    import torch
    import streamingflow_amd as sfa
    from util import make_cfg
    net = sfa.FuturePredictionODE(8, 8, 4, make_cfg(8)).eval()
    x = torch.zeros(1, 1, 8, 16, 16)
    ts = torch.zeros(1, 1, dtype=torch.float64)
    with pytest.raises(RuntimeError):
        net(x, x, None, ts, None, ts)


def test_unbuilt_options_fail_loudly():
    import pytest
    import streamingflow_amd as sfa
    from util import make_cfg
    cfg = make_cfg(8)
    cfg.MODEL.SMALL_ENCODER.SKIPCO = True
    with pytest.raises(NotImplementedError):
        sfa.FuturePredictionODE(8, 8, 4, cfg)
