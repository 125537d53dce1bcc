import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from streamingflow_amd.models import sparse_encoder as SE
from streamingflow_amd.voxelize import Voxelization, voxelize
from oracle import cases, sparse_encoder_ref as SR
from workloads import hashfill
import voxelbench
cfg = SR.default_cfg()
m = SE.SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=16, output_channels=128, encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"], block_type="basicblock").eval()
sd = hashfill.fill_state_dict({k: torch.empty(v) if v else torch.tensor(0) for k, v in SR.state_dict_shapes(cfg).items()}, seed=83, gain=1.6)
m.load_state_dict(sd); m = m.cuda()
vs, rng, mp, mv = cases.VOXEL_SHIPPED
vz = Voxelization(list(vs), list(rng), mp, (120000, mv)).eval()
feats, coords, sizes = voxelize([voxelbench.cloud().cuda()], vz)
orig_conv, orig_table, orig_sites = SE.SparseEncoder._conv, m._table, m._out_sites
def tconv(w, feats, nbr, n_out, add=None, act_after_add=False):
    torch.cuda.synchronize(); t=time.perf_counter(); r = orig_conv(w, feats, nbr, n_out, add, act_after_add); torch.cuda.synchronize()
    print(f"  conv n_out={n_out:7d} cin={w.c0:3d} cout={w.cout:3d} taps={w.kh:2d}: {(time.perf_counter()-t)*1e6:8.1f} us  useful {2.0*n_out*w.kh*w.c0*w.cout/1e9:6.2f} GF")
    return r
SE.SparseEncoder._conv = staticmethod(tconv)
def ttable(*a):
    torch.cuda.synchronize(); t=time.perf_counter(); r = orig_table(*a); torch.cuda.synchronize(); print(f"  table {(time.perf_counter()-t)*1e6:8.1f} us n_out={a[1].shape[0]}"); return r
m._table = ttable
def tsites(*a):
    torch.cuda.synchronize(); t=time.perf_counter(); r = orig_sites(*a); torch.cuda.synchronize(); print(f"  out_sites {(time.perf_counter()-t)*1e6:8.1f} us -> {r[0].shape[0]} sites, shape {r[1]}"); return r
m._out_sites = tsites
m(feats, coords, 1); print("---- second pass")
m(feats, coords, 1)
