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
    declared -= {"sf_status"}
    h = ctypes.CDLL(path)
    missing = [n for n in sorted(declared) if not hasattr(h, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()
    assert lib.sf_version() >= 100
    assert lib.sf_status_string(-2) == b"workspace too small"
    # workspace sizing is host arithmetic only
    assert lib.sf_dual_cell_ws_bytes(64, 1, 50, 50) >= 10 * 64 * 2500 * 4


def test_struct_sizes_match_header_layout():
    from streamingflow_amd import _lib
    assert ctypes.sizeof(_lib.ConvW) == 3 * 8 + 12 * 4 + 8      # 3 pointers + 11 int32 + 1 reserved + the optional split-bf16 weight pointer
    assert ctypes.sizeof(_lib.DualW) == 11 * ctypes.sizeof(_lib.ConvW) + 16
    assert ctypes.sizeof(_lib.GruW) == 3 * ctypes.sizeof(_lib.ConvW)
    assert ctypes.sizeof(_lib.ResW) == 3 * ctypes.sizeof(_lib.ConvW)


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
