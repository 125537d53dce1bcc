"""The C ABI is self-sufficient: a plain C program (gcc, no Python, no torch) links libsfnative.so, packs a reference-format
convolution + BatchNorm on the device with sf_pack_conv and runs sf_conv2d_fwd.  Also the Python
packer's torch-free path: sf_pack_conv's BatchNorm fold / ConvTranspose flip / duplicate-input fold / row interleave
against the same operations written with torch ops."""
import os
import shutil
import subprocess

import pytest
import torch

from util import ROOT

pytestmark = pytest.mark.gpu


def test_c_host_packs_and_runs_a_convolution(tmp_path):
    from streamingflow_amd import build
    lib = build.build()
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path / "test_cabi")
    cmd = [gcc, "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "test_cabi.c"), "-L" + os.path.dirname(lib), "-lsfnative", "-L/opt/rocm/lib", "-lamdhip64",
           "-lm", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max-abs error" in r.stdout


@pytest.mark.parametrize("mode", ["plain", "transposed", "fold_dup", "interleave"])
def test_pack_conv_matches_torch_ops(mode):
    from streamingflow_amd import _lib, packing
    torch.manual_seed(3)
    C, kh = 8, 3
    if mode == "transposed":
        w = torch.randn(C, 12, kh, kh, device="cuda")       # ConvTranspose2d [cin][cout][kh][kw]
        ref = w.permute(1, 0, 2, 3).flip(2, 3)
    elif mode == "fold_dup":
        w = torch.randn(12, 2 * C, kh, kh, device="cuda")
        ref = w[:, :C] + w[:, C:]
    else:
        w = torch.randn(16 if mode == "interleave" else 12, C, kh, kh, device="cuda")
        ref = w
    cout, cin = ref.shape[:2]
    bias = torch.randn(cout, device="cuda")
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w, cin, 0, bias=bias, transposed=mode == "transposed", fold_dup=mode == "fold_dup",
                        interleave=mode == "interleave")
    blob = pk.keep[0]
    n = cw.cout_pad * kh * kh * cw.cin_pad
    got = blob[:n].view(cw.cout_pad, kh, kh, cw.cin_pad)
    got_bias = blob[(n + 63) // 64 * 64 + (cw.cout_pad + 63) // 64 * 64:][:cw.cout_pad]
    want = torch.zeros_like(got)
    want_bias = torch.zeros(cw.cout_pad, device="cuda")
    rows = list(range(cout))
    if mode == "interleave":
        Ch = cout // 2
        rows = []
        for row in range(cw.cout_pad):
            T, g, r = row // 16, (row % 16) // 4, row % 4
            c = 8 * T + 2 * g + (r & 1)
            rows.append((c if r < 2 else Ch + c) if c < Ch else -1)
    for row, src in enumerate(rows):
        if src >= 0:
            want[row, :, :, :cin] = ref[src].permute(1, 2, 0)
            want_bias[row] = bias[src]
    assert torch.equal(got, want)
    assert torch.equal(got_bias, want_bias)
    assert cw.scale is None or cw.scale == 0
