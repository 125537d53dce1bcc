"""Depthwise 7x7 + LayerNorm (ConvNeXt block front) timing at the shipped sizes.  Usage (GPU box): python tools/dwbench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import streamingflow_amd.layers.convolutions as Cv  # noqa: E402


def main():
    blk = Cv.Block(64).eval().cuda()
    for n in (7, 56, 224):
        x = torch.randn(n, 200, 200, 64, device="cuda")
        for _ in range(3):
            blk.forward_nhwc(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            blk.forward_nhwc(x)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"convnext block n={n} 200x200x64: {ms:.3f} ms  ({2 * x.numel() * 4 / ms / 1e6:.0f} GB/s on in+out of the block)")


if __name__ == "__main__":
    main()
