"""SparseEncoder on the shipped-size synthetic cloud, VERDICT r5 item 5: 38 % of the executed (site, tap) products have no input site, and a
skip at (16-row fragment, tap) granularity keeps 97 % of them in stored and in Morton order (profiles/r05_sparse_fragment_density.jsonl).
What is left to try is an order of the OUTPUT rows in which rows with the same neighbour mask sit together, so that a whole pixel tile of
the implicit-GEMM kernel (64 or 128 rows) can drop a tap.  For every convolution of the encoder this measures, at tap granularity:
  * the live (site, tap) share (the floor of any skip);
  * the share of (T-row tile, tap) pairs with at least one live row, T = 64 / 128, with the rows
      - as stored,
      - sorted by their neighbour mask read as a binary number whose most significant bits are the taps with the FEWEST live rows
        (rows of a tile then agree on the leading ~log2(n / T) taps of that order),
      - sorted by (number of live taps, mask);
and, weighted by executed FLOPs, what a tile-level tap skip could save per stage and over the encoder.
Usage: python3 tools/r06/sparse_mask_sort_density.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from streamingflow_amd.models.sparse_encoder import SparseEncoder   # noqa: E402
from streamingflow_amd.voxelize import Voxelization, voxelize   # noqa: E402
from workloads import hashfill, synthetic as cases   # noqa: E402
import voxelbench   # noqa: E402


def tile_live(lv, T):
    n, taps = lv.shape
    pad = (-n) % T
    return float(torch.cat([lv, torch.zeros((pad, taps), dtype=torch.bool, device=lv.device)], 0).view(-1, T, taps).any(1).float().mean())


def main():
    dev = torch.device("cuda", 0)
    cfg = dict(cases.SPARSE_SHIPPED)
    m = SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=cfg["base_channels"], output_channels=cfg["output_channels"],
                      encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"], block_type="basicblock").eval()
    m.load_state_dict(hashfill.fill_state_dict(m.state_dict(), seed=83, gain=1.6))
    m = m.to(dev)
    vs, rng, mp, mv = cases.VOXEL_SHIPPED
    vz = Voxelization(list(vs), list(rng), mp, (120000, mv)).eval()
    feats, coords, sizes = voxelize([voxelbench.cloud().to(dev)], vz)
    rows = []
    orig_conv = m._conv

    def spy_conv(w, f, nbr, n_out, add=None, act_after_add=False, mask=None):
        live = nbr[:n_out] >= 0
        taps = live.shape[1]
        cin = w.c0 + w.c1
        r = {"conv": len(rows), "cin": cin, "cout": w.cout, "output_sites": int(n_out), "taps": int(taps), "gflop_dense_taps": 2.0 * n_out * taps * cin * w.cout / 1e9,
             "site_tap_live": float(live.float().mean())}
        per_tap = live.float().mean(0)
        order_taps = torch.argsort(per_tap)                               # rarest tap first = most significant bit
        weights = (2 ** torch.arange(taps - 1, -1, -1, device=live.device, dtype=torch.float64))
        key = (live[:, order_taps].double() * weights).sum(1)
        by_mask = torch.argsort(key)
        by_count = torch.argsort(live.sum(1).double() * float(2 ** taps) + key)
        for T in (64, 128):
            r[f"stored_tile{T}"] = tile_live(live, T)
            r[f"mask_sorted_tile{T}"] = tile_live(live[by_mask], T)
            r[f"count_mask_sorted_tile{T}"] = tile_live(live[by_count], T)
        r["distinct_masks"] = int(torch.unique(key).numel())
        rows.append(r)
        return orig_conv(w, f, nbr, n_out, add, act_after_add, mask=mask)

    m._conv = spy_conv
    with torch.no_grad():
        m(feats, coords, 1, nhwc=True)
    tot = sum(r["gflop_dense_taps"] for r in rows)
    for r in rows:
        print(json.dumps(r))
    summ = {"summary": "share of the encoder's executed (dense-tap) FLOPs that remains with a (tile, tap) skip, by row order", "convolutions": len(rows),
            "gflop_dense_taps": tot, "useful_site_tap_share": sum(r["gflop_dense_taps"] * r["site_tap_live"] for r in rows) / tot}
    for name in ("stored", "mask_sorted", "count_mask_sorted"):
        for T in (64, 128):
            summ[f"{name}_tile{T}"] = sum(r["gflop_dense_taps"] * r[f"{name}_tile{T}"] for r in rows) / tot
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
