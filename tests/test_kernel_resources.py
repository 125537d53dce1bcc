"""CPU: register / spill budget of every gfx950 kernel in the built libsfnative.so (round 3's one-kernel-per-step
spilled 244-383 VGPRs and 40-59 SGPRs, which confounded its measurement).  The code objects are cut out of the fat binary and
read with llvm-readelf --notes:
  * no kernel spills vector registers;
  * no kernel spills scalar registers, except the few listed below, whose scalar spills (v_writelane into a spare VGPR, no memory
    traffic) sit in cold prologue / epilogue code and are bounded here so that they cannot grow unnoticed;
  * the persistent flow kernel (SF_PERSIST=1) fits the 168-register cap of a 768-thread workgroup without scratch."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

from util import ROOT

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
# kernel-name fragment -> (max scalar spills, where they are)
SGPR_SPILL_ALLOWED = {
    "sp_flow_kernel": (20, "SE-gate prologue of the two SE-scaled phases of a step and the grouped scalar loads at the start of an item "
                           "(once per item, not in the K loop)"),
    "conv_wino_kernel": (8, "tile decode / epilogue address set-up"),
    "conv_wino5_kernel": (8, "tile decode / epilogue address set-up"),
    "conv_sp_kernelILi0ELb1ELi2": (1, "SE-gate prologue of the 32-pixel-tile AFFINE kernel (one v_writelane)"),
    # 64-pixel tiles with the Winograd block (round 6): the epilogue's scalars are requested in one batch before the loop and live across it
    # beside the SE gate's / the trusting gate's own; the overflow sits in v_writelane / v_readlane pairs outside the MFMA loop
    "conv_sp_kernelILi0ELb1ELi4ELb0": (12, "SE-scaled AFFINE, 64-pixel tiles: SE prologue + the epilogue's scalar batch"),
    "conv_sp_kernelILi4ELb1ELi4ELb0": (12, "SE-scaled SAMPLE, 64-pixel tiles: SE prologue + the epilogue's scalar batch"),
    "conv_sp_kernelILi3ELb0ELi4ELb0": (4, "TRUST, 64-pixel tiles: ten operand tensors' scalars held across the Winograd loop"),
    "conv_sp_kernelILi0ELb0ELi4ELb0": (4, "AFFINE, 64-pixel tiles: the epilogue's scalar batch held across the Winograd loop"),
    "conv_sp_kernelILi2ELb0ELi4ELb0": (30, "LayerNorm launch, 64-pixel tiles: the trusting gate's 7x7 as nine Winograd sub-kernels (tap-group cursor of the loaders, fused 1x1 "
                                           "layer's scalars) beside the direct form; outside the MFMA loop"),
    "dwconv7_ln_c64_kernel": (40, "row / column addresses kept in scalar registers by design (csrc/aux_kernels.hip)"),
}


def _code_objects(so):
    data = open(so, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        b = m.start()
        num = struct.unpack_from("<Q", data, b + 24)[0]
        off = b + 32
        for _ in range(num):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and s > 0:
                out.append(data[b + o:b + o + s])
    return out


def _kernels(so):
    res = []
    for co in _code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        cur = {}
        for ln in txt.split("\n"):
            m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", ln)
            if not m:
                continue
            k, v = m.groups()
            if k == "agpr_count" and cur:          # first key of a kernel's record
                if "name" in cur:
                    res.append(cur)
                cur = {}
            cur[k] = v.strip()
        if "name" in cur:
            res.append(cur)
    return res


def test_no_kernel_spills_vector_registers_and_scalar_spills_are_bounded():
    from streamingflow_amd import build
    if not os.path.exists(READELF) or not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("ROCm LLVM tools (llvm-readelf / hipcc) are not installed here")
    so = build.build()
    ks = [k for k in _kernels(so) if "vgpr_spill_count" in k]
    ours = [k for k in ks if k["name"].startswith("_ZN2sf")]
    assert len(ours) > 100, len(ours)            # the conv kernel families alone are well over a hundred instantiations
    bad = []
    for k in ks:
        vs, ss = int(k["vgpr_spill_count"]), int(k["sgpr_spill_count"])
        if vs:
            bad.append((k["name"], "vgpr_spill", vs))
        if ss:
            lim = next((v[0] for frag, v in SGPR_SPILL_ALLOWED.items() if frag in k["name"]), 0)
            if ss > lim:
                bad.append((k["name"], "sgpr_spill", ss))
    assert not bad, bad
    flow = [k for k in ours if "sp_flow_kernel" in k["name"]]
    assert len(flow) == 2                        # fp32 and bf16x3
    for k in flow:
        assert int(k["vgpr_count"]) <= 168 and int(k["private_segment_fixed_size"]) == 0, k
    # the Winograd kernel of the large launches: two workgroups of 8 waves per CU = 128 registers, no scratch
    wino = [k for k in ours if "conv_wino5_kernel" in k["name"]]
    assert len(wino) == 13                       # AFFINE / BLEND x (plain, concatenated images) x (32, 16 tiles) + the dilated AFFINE form + the 7x7 LayerNorm form and the sampling layer x (plain, concatenated)
    for k in wino:
        small = "Li2ELi1EEE" in k["name"]               # the 16-tile form: three workgroups of 8 waves per CU = 80 registers
        assert int(k["vgpr_count"]) <= (80 if small else 128) and int(k["private_segment_fixed_size"]) == 0, k
    assert sum("Li2ELi1EEE" in k["name"] for k in wino) == 4
    # the fused ConvNeXt MLP: two waves per SIMD (<= 256 registers between the vector and accumulator files), no scratch
    mlp = [k for k in ours if "convnext_mlp_kernel" in k["name"]]
    assert len(mlp) == 1
    assert int(mlp[0]["vgpr_count"]) + int(mlp[0]["agpr_count"]) <= 256 and int(mlp[0]["private_segment_fixed_size"]) == 0, mlp[0]
    # the small-P kernels of the default path (one launch per layer group) too: no scratch at all
    for k in ours:
        if "conv_sp_kernel" in k["name"]:
            assert int(k["private_segment_fixed_size"]) == 0, k["name"]
