"""SparseEncoder on the shipped-size synthetic cloud: could a K loop that skips a (16-row MFMA fragment, tap) pair whose rows all lack a
neighbour drop work?  (the tile-level answer of round 4 — ~1.0 live at 64 / 128 rows — does not answer the 16-row
question.)  For every convolution of the encoder: the share of (output site, tap) pairs with an input site, and the share of (T consecutive
output rows, tap) pairs in which AT LEAST ONE row has one, T = 16 / 32 / 64, with the output rows in the order the product keeps them
(sorted by batch / coordinate) and in Morton (z-curve) order of their coordinates — the best a re-ordering could do for locality.
Weighted by each convolution's executed FLOPs: the share of the encoder's matrix work a fragment-level skip could drop.
Usage: python3 tools/r05/sparse_fragment_density.py"""
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


def morton(c):
    """c [n, 4] int32 (batch, x, y, z) -> int64 keys: batch major, then the interleaved bits of x, y, z (11 bits each)."""
    x, y, z = (c[:, k].to(torch.int64) for k in (1, 2, 3))
    key = torch.zeros_like(x)
    for b in range(11):
        key |= ((x >> b) & 1) << (3 * b + 2) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b)
    return key + (c[:, 0].to(torch.int64) << 40)


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
    rows, tables = [], {}
    orig_table, orig_conv = m._table, m._conv

    def spy_table(in_coords, out_coords, batch, shape, k, s, p, subm):
        t = orig_table(in_coords, out_coords, batch, shape, k, s, p, subm)
        tables[t.data_ptr()] = out_coords
        return t

    def spy_conv(w, f, nbr, n_out, add=None, act_after_add=False, mask=None):
        live = nbr[:n_out] >= 0
        oc = tables[nbr.data_ptr()][:n_out]
        taps = live.shape[1]
        cin = w.c0 + w.c1
        r = {"cin": cin, "cout": w.cout, "output_sites": int(n_out), "taps": int(taps), "flops_dense_taps": 2.0 * n_out * taps * cin * w.cout,
             "site_tap_live": float(live.float().mean())}
        order = torch.argsort(morton(oc))
        for name, lv in (("as_stored", live), ("morton", live[order])):
            for T in (16, 32, 64):
                pad = (-n_out) % T
                q = torch.cat([lv, torch.zeros((pad, taps), dtype=torch.bool, device=lv.device)], 0).view(-1, T, taps).any(1)
                r[f"{name}_frag{T}_tap_live"] = float(q.float().mean())
        rows.append(r)
        return orig_conv(w, f, nbr, n_out, add, act_after_add, mask=mask)

    m._table, m._conv = spy_table, spy_conv
    with torch.no_grad():
        m(feats, coords, 1, nhwc=True)
    tot = sum(r["flops_dense_taps"] for r in rows)
    for r in rows:
        print(json.dumps(r))
    summ = {"summary": "share of the encoder's executed (dense-tap) FLOPs that remains with a (T-row fragment, tap) skip", "convolutions": len(rows),
            "gflop_dense_taps": tot / 1e9, "useful_site_tap_share": sum(r["flops_dense_taps"] * r["site_tap_live"] for r in rows) / tot}
    for name in ("as_stored", "morton"):
        for T in (16, 32, 64):
            summ[f"{name}_frag{T}"] = sum(r["flops_dense_taps"] * r[f"{name}_frag{T}_tap_live"] for r in rows) / tot
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
