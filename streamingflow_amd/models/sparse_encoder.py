"""LiDAR ``SparseEncoder`` (mmdet3d/models/backbones/sparse_encoder.py:10-226) on the MI355X conv library —
SURVEY.md §8f N2, second half.

Same constructor arguments, ``forward(voxel_features, coors, batch_size)`` and ``state_dict`` keys as the
reference configuration StreamingFlow builds (streamingflow.py:111: ``block_type='basicblock'``, order
conv-norm-act; ``block_type='conv_module'`` — the class default — is built too): spconv weights ``[kx, ky, kz, Cin, Cout]``, ``BatchNorm1d(eps=1e-3)``; coordinates are
``(batch, x, y, z)`` in ``sparse_shape = (X, Y, Z)``.

Every sparse convolution is the implicit-GEMM kernel of the dense path reading a neighbour table
(``sf_sparse_conv_fwd``): output row j gathers, per kernel tap, the input row ``nbr[j][tap]`` — no
gather / GEMM / scatter-add passes, no atomics; BatchNorm, ReLU and the residual of ``SparseBasicBlock``
are fused in the epilogue.  Tables come from device-side sorts + binary searches
(``sf_sparse_table_fwd``), one per submanifold stage (shared by its convs) and one per strided conv
(whose output sites need one host read of their count).  ``dense()`` + permute/view is one scatter.
Evaluation mode only, CUDA tensors only.
"""
import math

import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..runtime import PackedModule, ptr

C = _lib.C


def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


class SparseConv3d(nn.Module):
    """Parameter container: spconv ``SparseConvolution`` (conv.py:52-112), weight [kx, ky, kz, Cin, Cout], no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, subm=False, indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.indice_key = subm, indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))      # conv.py:106-107


class SparseBasicBlock(nn.Module):
    """Parameter container with mmdet's BasicBlock names (sparse_block.py:61-107)."""

    def __init__(self, planes, eps, momentum):
        super().__init__()
        self.conv1 = SparseConv3d(planes, planes, 3, padding=1, subm=True)
        self.bn1 = nn.BatchNorm1d(planes, eps=eps, momentum=momentum)
        self.conv2 = SparseConv3d(planes, planes, 3, padding=1, subm=True)
        self.bn2 = nn.BatchNorm1d(planes, eps=eps, momentum=momentum)


def _convmodule(cin, cout, k, stride, padding, subm, key, eps, momentum):
    return nn.Sequential(SparseConv3d(cin, cout, k, stride, padding, subm, key), nn.BatchNorm1d(cout, eps=eps, momentum=momentum),
                         nn.ReLU(inplace=True))


class SparseEncoder(PackedModule):
    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"), norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01),
                 base_channels=16, output_channels=128, encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module"):
        super().__init__()
        if block_type not in ("basicblock", "conv_module"):
            raise AssertionError("block_type must be 'conv_module' or 'basicblock'")
        if tuple(order) != ("conv", "norm", "act"):
            raise NotImplementedError("only the post-activation order (conv, norm, act) is built")
        self.block_type = block_type
        self.sparse_shape = [int(v) for v in sparse_shape]
        self.in_channels, self.base_channels, self.output_channels = in_channels, base_channels, output_channels
        self.encoder_channels, self.encoder_paddings = encoder_channels, encoder_paddings
        self.stage_num = len(encoder_channels)
        eps, mom = norm_cfg.get("eps", 1e-3), norm_cfg.get("momentum", 0.01)
        self.conv_input = _convmodule(in_channels, base_channels, 3, 1, 1, True, "subm1", eps, mom)
        self.encoder_layers = nn.Sequential()
        cin = base_channels
        for i, blocks in enumerate(encoder_channels):
            stage = []
            for j, cout in enumerate(tuple(blocks)):
                pad = tuple(encoder_paddings[i])[j]
                if block_type == "conv_module":          # sparse_encoder.py:166-179, :203-213
                    if i != 0 and j == 0:
                        stage.append(_convmodule(cin, cout, 3, 2, pad, False, f"spconv{i + 1}", eps, mom))
                    else:
                        stage.append(_convmodule(cin, cout, 3, 1, pad, True, f"subm{i + 1}", eps, mom))
                elif j == len(blocks) - 1 and i != len(encoder_channels) - 1:
                    stage.append(_convmodule(cin, cout, 3, 2, pad, False, f"spconv{i + 1}", eps, mom))
                else:
                    if cin != cout:
                        raise ValueError("SparseBasicBlock needs in_channels == out_channels")
                    stage.append(SparseBasicBlock(cout, eps, mom))
                cin = cout
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", nn.Sequential(*stage))
        self.conv_out = _convmodule(cin, output_channels, (1, 1, 3), (1, 1, 2), 0, False, "spconv_down2", eps, mom)

    # ---- packing ----------------------------------------------------------------------------------
    def _pack(self):
        pk = packing.Pack({})

        def one(name, conv, bn, cin_pad_to=None):
            w = conv.weight.detach()
            kx, ky, kz, cin, cout = w.shape
            cp = cin_pad_to or cin
            w2 = torch.zeros((cout, cp, kx * ky * kz, 1), dtype=torch.float32, device=w.device)
            w2[:, :cin, :, 0] = w.permute(4, 3, 0, 1, 2).reshape(cout, cin, kx * ky * kz)
            sc = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
            bi = bn.bias.detach() - bn.running_mean.detach() * sc
            pk.struct[name] = packing.conv_w(pk, w2, cp, 0, sc, bi, "relu", pad=0)
        self._cin_pad = (self.in_channels + 3) // 4 * 4
        one("conv_input", self.conv_input[0], self.conv_input[1], self._cin_pad)
        for i, stage in enumerate(self.encoder_layers):
            for j, blk in enumerate(stage):
                if isinstance(blk, SparseBasicBlock):
                    one(f"{i}.{j}.c1", blk.conv1, blk.bn1)
                    one(f"{i}.{j}.c2", blk.conv2, blk.bn2)
                else:
                    one(f"{i}.{j}", blk[0], blk[1])
        one("conv_out", self.conv_out[0], self.conv_out[1])
        return pk

    # ---- device helpers -----------------------------------------------------------------------------
    @staticmethod
    def _i3(v):
        return (C.c_int32 * 3)(*[int(x) for x in v])

    def _table(self, in_coords, out_coords, batch, shape, k, s, p, subm):
        L = _lib.lib()
        dev = in_coords.device
        n_in, n_out = in_coords.shape[0], out_coords.shape[0]
        ntaps = k[0] * k[1] * k[2]
        nbr = torch.empty((max(n_out, 1), ntaps), dtype=torch.int32, device=dev)
        ws = runtime.workspace(L.sf_sparse_index_ws_bytes(n_in, ntaps), dev)
        _lib.check(L.sf_sparse_table_fwd(ptr(in_coords), n_in, ptr(out_coords), n_out, batch, self._i3(shape), self._i3(k), self._i3(s),
                                         self._i3(p), int(subm), ptr(nbr), ptr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "sparse_table")
        return nbr

    def _out_sites(self, coords, batch, shape, k, s, p):
        L = _lib.lib()
        dev = coords.device
        n_in = coords.shape[0]
        ntaps = k[0] * k[1] * k[2]
        oshape = [(shape[a] + 2 * p[a] - (k[a] - 1) - 1) // s[a] + 1 for a in range(3)]
        cap = min(n_in * ntaps, batch * oshape[0] * oshape[1] * oshape[2])
        out = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        cnt = torch.zeros((), dtype=torch.int32, device=dev)
        ws = runtime.workspace(L.sf_sparse_index_ws_bytes(n_in, ntaps), dev)
        _lib.check(L.sf_sparse_out_sites_fwd(ptr(coords), n_in, batch, self._i3(shape), self._i3(k), self._i3(s), self._i3(p), ptr(out), cap,
                                             ptr(cnt), ptr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "sparse_out_sites")
        return out[: int(cnt.item())], oshape

    @staticmethod
    def _conv(w, feats, nbr, n_out, add=None, act_after_add=False, mask=None):
        L = _lib.lib()
        dev = feats.device
        out = torch.empty((n_out, w.cout), dtype=torch.float32, device=dev)
        ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), dev)
        _lib.check(L.sf_sparse_conv_masked_fwd(C.byref(w), ptr(feats), feats.shape[1], feats.shape[0], ptr(nbr), ptr(mask), n_out, ptr(add), int(act_after_add),
                                               ptr(out), ptr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "sparse_conv")
        return out

    # ---- rows sorted by neighbour mask + per-tile tap masks (round 6) --------------------------------------------------------------
    # 38 % of the (site, tap) products of a shipped-size cloud have no input site.  In stored order a 64/128-row tile of the kernel still
    # needs every tap (some row of it always has one); with the rows of a stage sorted by their neighbour mask — read as a number whose
    # most significant bits are the rarest taps — the rows of a tile agree on those taps and a quarter of the (tile, tap) pairs drop out
    # (profiles/r06_s_sparse_mask_sort.jsonl).  The order of the rows is free: tables, features and coordinates are permuted together
    # and dense() scatters by coordinate.
    # MEASURED NO-GO, kept opt-in (SF_SPARSE_SORT=1; tests/test_sparse_encoder.py runs it): the kernels then walk 78 % of the dense-tap
    # products, bitwise the same results — and the encoder takes 15.5 ms instead of 12.1 (profiles/r06_t_sparse_mask_sort_ab.txt): rows
    # that are neighbours in the sorted order are not neighbours in space, so the 27 gathers of a tile no longer share input rows in L2
    # (in coordinate order each input row is read by up to 27 nearby output rows), and the sort / permutation passes cost time of their own.
    # In stored order the masks are all-ones in practice (a tile drops 0.3 % of its taps) and are not even computed.
    SORT = __import__("os").environ.get("SF_SPARSE_SORT", "0") != "0"

    @staticmethod
    def _mask_order(tab, n):
        live = tab[:n] >= 0
        taps = live.shape[1]
        order = torch.argsort(live.sum(0))                                             # rarest tap first
        w = (1 << torch.arange(taps - 1, -1, -1, device=tab.device, dtype=torch.int64))
        key = (live[:, order].to(torch.int64) * w).sum(1)
        return torch.argsort(key)

    @staticmethod
    def _permute_subm(tab, perm):
        """neighbour table of a submanifold stage after its rows (= its input rows) were permuted"""
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        t = tab[perm]
        return torch.where(t >= 0, inv[t.clamp_min(0).long()].to(torch.int32), t).contiguous()

    @staticmethod
    def _tile_mask(nbr, n):
        taps = nbr.shape[1]
        if n == 0 or taps > 31:
            return None
        live = nbr[:n] >= 0
        pad = (-n) % 64
        if pad:
            live = torch.cat([live, live.new_zeros((pad, taps))], 0)
        any_ = live.view(-1, 64, taps).any(1)
        w = (1 << torch.arange(taps, device=nbr.device, dtype=torch.int64))
        return (any_.to(torch.int64) * w).sum(1).to(torch.int32).contiguous()

    def forward(self, voxel_features, coors, batch_size, nhwc=False, **kwargs):
        """voxel_features [N, Cin] f32, coors [N, 4] int (batch, x, y, z) -> [B, C*D, H, W] (sparse_encoder.py:100-139)."""
        runtime.require_cuda(voxel_features, coors)
        if self.training:
            raise RuntimeError("streamingflow_amd.SparseEncoder is inference-only (BatchNorm statistics are folded): call .eval()")
        W = self.packed().struct
        B = int(batch_size)
        dev = voxel_features.device
        coords = coors.to(torch.int32).contiguous()
        n = coords.shape[0]
        shape = list(self.sparse_shape)
        x = torch.zeros((n, self._cin_pad), dtype=torch.float32, device=dev)
        x[:, : self.in_channels] = voxel_features
        k3, one3, zero3 = [3, 3, 3], [1, 1, 1], [0, 0, 0]
        tab = self._table(coords, coords, B, shape, k3, one3, zero3, True) if n else None
        tmask = None
        if n and self.SORT:      # the voxels in neighbour-mask order (features, coordinates and the table together)
            perm = self._mask_order(tab, n)
            x, coords, tab = x[perm].contiguous(), coords[perm].contiguous(), self._permute_subm(tab, perm)
        if n and self.SORT:
            tmask = self._tile_mask(tab, n)
        x = self._conv(W["conv_input"], x, tab, n, mask=tmask)
        for i, stage in enumerate(self.encoder_layers):
            for j, blk in enumerate(stage):
                if isinstance(blk, SparseBasicBlock):
                    if tab is None and n:
                        tab = self._table(coords, coords, B, shape, k3, one3, zero3, True)
                        tmask = self._tile_mask(tab, n) if self.SORT else None
                    y = self._conv(W[f"{i}.{j}.c1"], x, tab, n, mask=tmask)
                    x = self._conv(W[f"{i}.{j}.c2"], y, tab, n, add=x, act_after_add=True, mask=tmask)
                elif blk[0].subm:                       # conv_module: submanifold conv + BN + ReLU
                    if tab is None and n:
                        tab = self._table(coords, coords, B, shape, k3, one3, zero3, True)
                        tmask = self._tile_mask(tab, n) if self.SORT else None
                    x = self._conv(W[f"{i}.{j}"], x, tab, n, mask=tmask)
                else:
                    nxt = stage[j + 1] if j + 1 < len(stage) else (self.encoder_layers[i + 1][0] if i + 1 < len(self.encoder_layers) else None)
                    subm_next = nxt is not None and (isinstance(nxt, SparseBasicBlock) or nxt[0].subm)
                    x, coords, shape, n, tab, tmask = self._strided(W[f"{i}.{j}"], blk[0], x, coords, B, shape, n, sort_for_subm=subm_next and self.SORT)
        x, coords, shape, n, _, _ = self._strided(W["conv_out"], self.conv_out[0], x, coords, B, shape, n)
        Cc = x.shape[1] if n else self.output_channels
        out = torch.empty((B, shape[0], shape[1], Cc * shape[2]), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().sf_sparse_to_dense_fwd(ptr(x), ptr(coords), n, Cc, B, shape[0], shape[1], shape[2], ptr(out),
                                                     runtime.stream_ptr(dev)), "sparse_to_dense")
        return out if nhwc else runtime.to_nchw(out)

    def _strided(self, w, conv, x, coords, B, shape, n, sort_for_subm=False):
        """strided SparseConv3d; returns (features, coordinates, grid, sites, submanifold table of the new sites or None, its tile masks).
        sort_for_subm: submanifold convolutions follow on the new sites — their table is built here and the new sites are put in the order
        of ITS masks (the four 3x3x3 layers of a stage are 8x the strided layer's work) before the strided layer writes them."""
        k, s, p = conv.kernel_size, conv.stride, conv.padding
        oshape = [(shape[a] + 2 * p[a] - (k[a] - 1) - 1) // s[a] + 1 for a in range(3)]
        if n == 0:
            return x.new_zeros((0, w.cout)), coords, oshape, 0, None, None
        oc, oshape = self._out_sites(coords, B, shape, k, s, p)
        oc = oc.contiguous()
        m = oc.shape[0]
        tab = tmask = None
        if sort_for_subm and m:
            tab = self._table(oc, oc, B, oshape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
            perm = self._mask_order(tab, m)
            oc, tab = oc[perm].contiguous(), self._permute_subm(tab, perm)
            tmask = self._tile_mask(tab, m)
        nbr = self._table(coords, oc, B, shape, k, s, p, False)
        return self._conv(w, x, nbr, m, mask=self._tile_mask(nbr, m) if self.SORT else None), oc, oshape, m, tab, tmask
