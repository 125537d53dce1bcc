"""Stress of the cross-workgroup split-K hand-off: many launches that reuse the same slab and ticket addresses, with
fresh inputs every time, cache-thrashing work in between and a concurrent stream keeping some CUs busy (uneven load).  Prints
one line of 64-bit checksums over every output word; tests/test_gpu_splitk_stress.py runs it with the fence-free hand-off
(default), with SF_HANDOFF_FENCED=1 (release / acquire fences: the known-good form) and compares them bit for bit, and with
SF_SPLIT=0 (no hand-off at all) within a tolerance.
Usage: python3 tools/splitk_stress.py <iterations> [dump.pt]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from streamingflow_amd import _lib, packing, runtime  # noqa: E402


def checksum(t):
    v = t.contiguous().view(torch.int32).to(torch.int64)
    w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64)
    return int(((v.reshape(-1) * (w % 1000003)).sum() & 0x7FFFFFFFFFFFFFFF).item())


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dump = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(77)
    L = _lib.lib()
    pk = packing.Pack(None)
    # (n, H, W, c0, c1, cout, k): one latent on the small-P kernel (7x7 of the trusting gate, a 128 -> 128 layer) and the
    # same layers at 8 samples / a 200x200 latent slice on the LDS-DMA split-K tiles
    cases = [(1, 50, 50, 64, 64, 64, 7), (1, 50, 50, 128, 0, 128, 3), (8, 50, 50, 64, 64, 64, 7), (8, 50, 50, 128, 0, 128, 3), (1, 100, 100, 64, 64, 64, 7)]
    layers = []
    for (n, H, W, c0, c1, cout, k) in cases:
        w = torch.randn((cout, c0 + c1, k, k), device=dev, generator=g) * (1.0 / ((c0 + c1) * k * k)) ** 0.5
        layers.append(packing.conv_w(pk, w, c0, c1, act="lrelu"))
    ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), dev)
    side = torch.cuda.Stream(device=dev)
    junk = torch.empty(64 << 20, device=dev)       # 256 MB: sweeps the L2s and the Infinity Cache
    sums = [0] * len(cases)
    kept = []
    for it in range(iters):
        for ci, (n, H, W, c0, c1, cout, k) in enumerate(cases):
            a0 = torch.randn((n, H, W, c0), device=dev, generator=g)
            a1 = torch.randn((n, H, W, c1), device=dev, generator=g) if c1 else None
            out = torch.empty((n, H, W, cout), device=dev)
            if it % 3 == 0:
                with torch.cuda.stream(side):       # uneven load: a second stream occupies part of the chip
                    junk[: (8 << 20)].mul_(1.0001)
            _lib.check(L.sf_conv2d_ex_fwd(ctypes.byref(layers[ci]), runtime.ptr(a0), c0, runtime.ptr(a1), c1, None, cout, 0,
                                          ctypes.c_void_p(out.data_ptr()), cout, 0, n, H, W, 0, runtime.ptr(ws), ws.numel() * 4,
                                          runtime.stream_ptr(dev)), "conv2d_ex")
            if it % 5 == 1:
                junk.add_(1.0)                       # evict everything between two uses of the slab
            sums[ci] = (sums[ci] * 31 + checksum(out)) & 0x7FFFFFFFFFFFFFFF
            if dump and it in (0, iters - 1):
                kept.append(out.cpu())
    torch.cuda.synchronize()
    print("CHECKSUMS", " ".join(str(s) for s in sums), flush=True)
    if dump:
        torch.save(kept, dump)


if __name__ == "__main__":
    main()
