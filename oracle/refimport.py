"""TEST INFRASTRUCTURE — import the *real* reference hot path from /root/reference (build container
only; the directory does not exist on the GPU box and nothing at run time there may need it).

The hot path imports three packages that are absent here and contribute no arithmetic
(SURVEY.md §8c): ``timm.models.layers.DropPath`` (identity at p=0), ``pyquaternion.Quaternion`` and
``nuscenes.utils.geometry_utils.transform_matrix`` (pulled in by an unused ``warp_features``
import).  They are replaced by inert stubs in ``sys.modules``; no reference file is copied.
"""
import os
import sys
import types
from types import SimpleNamespace

REF_ROOT = os.environ.get("SF_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "streamingflow"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


def install():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import torch

    class DropPath(torch.nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=DropPath)
    _stub("pyquaternion", Quaternion=object)
    _stub("nuscenes")
    _stub("nuscenes.utils")
    _stub("nuscenes.utils.geometry_utils", transform_matrix=lambda *a, **k: None)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # BEVerse modules: bypass package __init__ files that need mmcv/mmdet.
    base = os.path.join(REF_ROOT, "mmdet3d")
    for name, sub in [("mmdet3d", ""), ("mmdet3d.models", "models"),
                      ("mmdet3d.models.beverse", "models/beverse"),
                      ("mmdet3d.models.beverse.models", "models/beverse/models"),
                      ("mmdet3d.models.beverse.datasets", "models/beverse/datasets"),
                      ("mmdet3d.models.beverse.datasets.utils", "models/beverse/datasets/utils")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(base, sub)]
            sys.modules[name] = m


from workloads.synthetic import make_cfg       # noqa: E402,F401  (re-exported: tests and gen_golden use refimport.make_cfg)


def modules():
    """Return a namespace with the reference classes on the hot path."""
    install()
    from streamingflow.models.future_prediction_ode import FuturePredictionODE
    from streamingflow.layers import temporal_ode_bayes as tob
    from streamingflow.layers import temporal, convolutions, res_models
    from streamingflow.models import model_utils
    return SimpleNamespace(FuturePredictionODE=FuturePredictionODE, tob=tob, temporal=temporal,
                           convolutions=convolutions, res_models=res_models, model_utils=model_utils)


class patched_standard_normal:
    """Context manager: make ``Normal.rsample`` draw eps from ``source(shape, dtype, device)``.

    The reference samples through ``torch.distributions.Normal.rsample`` (model_utils.py:107-108),
    which calls ``torch.distributions.normal._standard_normal``; patching that name injects a
    reproducible eps stream without touching reference code.
    """

    def __init__(self, source):
        self.source = source

    def __enter__(self):
        import torch.distributions.normal as tdn
        self._mod, self._old = tdn, tdn._standard_normal
        tdn._standard_normal = lambda shape, dtype, device: self.source(tuple(shape), dtype, device)
        return self

    def __exit__(self, *exc):
        self._mod._standard_normal = self._old
        return False


def lift_splat_reference(kernel_forward):
    """Load the reference's lift-splat Python so that it can be *executed* here:
    ``mmdet3d/ops/bev_pool/bev_pool.py`` (ranks, argsort, interval bookkeeping, ``QuickCumsum``) and the
    ``streamingflow`` class of ``streamingflow/models/streamingflow.py`` (``bev_pool``,
    ``projection_to_birds_eye_view``, ``get_geometry``, ``create_frustum`` used as unbound functions
    on a stand-in ``self``).  The compiled CUDA extension ``bev_pool_ext`` cannot be built here (no
    nvcc); its one entry point used on the forward path is supplied by the caller
    (``kernel_forward`` = the oracle's restatement of bev_pool_cuda.cu:20-42).  Everything else the
    module imports but this path never calls (mmcv decorators, encoders, decoder, spconv ops) is
    replaced by inert stubs.  No reference file is copied."""
    install()
    import importlib

    def ident_factory(*a, **k):
        return lambda f: f

    _stub("mmcv")
    _stub("mmcv.runner", auto_fp16=ident_factory, force_fp32=ident_factory)
    tv = _stub("torchvision")      # utils/network.py:3,33 subclass a transform this path never uses
    tv.transforms = _stub("torchvision.transforms", Normalize=object)
    for name, attrs in [("streamingflow.models.encoder", {"Encoder": object}),
                        ("streamingflow.models.decoder", {"Decoder": object}),
                        ("streamingflow.models.planning_model", {"Planning": object}),
                        ("streamingflow.models.temporal_model", {"TemporalModelIdentity": object, "TemporalModel": object})]:
        _stub(name, **attrs)
    base = os.path.join(REF_ROOT, "mmdet3d", "ops")
    ops = sys.modules.get("mmdet3d.ops")
    if ops is None:
        ops = types.ModuleType("mmdet3d.ops")
        ops.__path__ = []          # do not run mmdet3d/ops/__init__.py (compiled extensions)
        sys.modules["mmdet3d.ops"] = ops
    pkg = types.ModuleType("mmdet3d.ops.bev_pool")
    pkg.__path__ = [os.path.join(base, "bev_pool")]
    sys.modules["mmdet3d.ops.bev_pool"] = pkg
    ext = types.ModuleType("mmdet3d.ops.bev_pool.bev_pool_ext")
    ext.bev_pool_forward = kernel_forward
    sys.modules["mmdet3d.ops.bev_pool.bev_pool_ext"] = ext
    pkg.bev_pool_ext = ext
    bp = importlib.import_module("mmdet3d.ops.bev_pool.bev_pool")
    ops.bev_pool = bp.bev_pool
    ops.Voxelization = object
    ops.DynamicScatter = object
    _stub("mmdet3d.models.builder", build_backbone=lambda *a, **k: None)
    sfm = importlib.import_module("streamingflow.models.streamingflow")
    from streamingflow.utils import geometry
    return SimpleNamespace(bev_pool_py=bp, model=sfm.streamingflow, geometry=geometry)


def voxel_reference():
    """The reference's ``Voxelization`` module (mmdet3d/ops/voxel/voxelize.py) running on the reference's
    own C++ CPU kernel, compiled from its sources by ``oracle/build_ref.py`` into oracle/_ref/.
    Returns a namespace with ``Voxelization`` and the raw extension, or None when the extension was
    never built (GPU box without the prebuilt file)."""
    import importlib
    from . import build_ref
    if available():
        build_ref.build()
    ext = build_ref.load_voxel_layer()
    if ext is None or not available():
        return SimpleNamespace(ext=ext, Voxelization=None) if ext is not None else None
    install()
    if "mmdet3d.ops" not in sys.modules:
        ops = types.ModuleType("mmdet3d.ops")
        ops.__path__ = []
        sys.modules["mmdet3d.ops"] = ops
    pkg = types.ModuleType("mmdet3d.ops.voxel")
    pkg.__path__ = [os.path.join(REF_ROOT, "mmdet3d", "ops", "voxel")]
    sys.modules["mmdet3d.ops.voxel"] = pkg
    sys.modules["mmdet3d.ops.voxel.voxel_layer"] = ext
    vz = importlib.import_module("mmdet3d.ops.voxel.voxelize")
    return SimpleNamespace(ext=ext, Voxelization=vz.Voxelization, voxelization=vz.voxelization)


def decoder_reference():
    """The reference's ``Decoder`` class (streamingflow/models/decoder.py) with the absent third-party
    ``torchvision.models.resnet.resnet18`` supplied by the oracle's restatement of torchvision's
    BasicBlock trunk (oracle/decoder_ref.py:tv_resnet18)."""
    import importlib
    install()
    from . import decoder_ref
    tv = _stub("torchvision")
    if not hasattr(tv, "transforms"):
        tv.transforms = _stub("torchvision.transforms", Normalize=object)
    models = _stub("torchvision.models")
    tv.models = models
    models.resnet = _stub("torchvision.models.resnet", resnet18=decoder_ref.tv_resnet18)
    sys.modules["torchvision.models.resnet"].resnet18 = decoder_ref.tv_resnet18
    sys.modules.pop("streamingflow.models.decoder", None)      # lift_splat_reference() may have stubbed it
    mod = importlib.import_module("streamingflow.models.decoder")
    return mod.Decoder


def temporal_model_reference():
    """The reference's ``TemporalModel`` (streamingflow/models/temporal_model.py) — importable as is."""
    import importlib
    install()
    sys.modules.pop("streamingflow.models.temporal_model", None)   # lift_splat_reference() may have stubbed it
    return importlib.import_module("streamingflow.models.temporal_model").TemporalModel


def eval_reference():
    """The reference's evaluation code: ``streamingflow/utils/instance.py`` (pure torch + scipy,
    imported as is) and ``IntersectionOverUnion`` / ``PanopticMetric`` of ``streamingflow/metrics.py``.  metrics.py
    needs ``pytorch_lightning.metrics`` (an API removed upstream; absent here): its ``Metric`` base class,
    ``stat_scores_multiple_classes`` and ``reduce`` are restated below from their published behaviour
    (pytorch-lightning 1.1 ``metrics/metric.py``, ``functional/classification.py``, ``functional/reduction.py``);
    ``skimage`` and ``streamingflow.utils.tools`` (plotting deps) are stubbed — these classes never call them."""
    import importlib
    import torch
    install()

    class Metric(torch.nn.Module):
        def __init__(self, compute_on_step=True, **kw):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None, persistent=False):
            self._defaults[name] = default.clone()
            setattr(self, name, default.clone())

        def forward(self, *a, **k):
            self.update(*a, **k)

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone())

    def stat_scores_multiple_classes(pred, target, num_classes=None, argmax_dim=1, reduction="none"):
        if pred.ndim == target.ndim + 1:
            pred = torch.argmax(pred, dim=argmax_dim)
        pred, target = pred.reshape(-1).long(), target.reshape(-1).long()
        tps = torch.zeros((num_classes + 1,), device=pred.device)
        fps, fns, sups = torch.zeros_like(tps), torch.zeros_like(tps), torch.zeros_like(tps)
        match_true = (pred == target).float()
        match_false = 1 - match_true
        tps.scatter_add_(0, pred, match_true)
        fps.scatter_add_(0, pred, match_false)
        fns.scatter_add_(0, target, match_false)
        tns = pred.size(0) - (tps + fps + fns)
        sups.scatter_add_(0, target, torch.ones_like(match_true))
        return (tps[:num_classes].float(), fps[:num_classes].float(), tns[:num_classes].float(), fns[:num_classes].float(),
                sups[:num_classes].float())

    def reduce(to_reduce, reduction):
        if reduction == "elementwise_mean":
            return torch.mean(to_reduce)
        if reduction == "none":
            return to_reduce
        if reduction == "sum":
            return torch.sum(to_reduce)
        raise ValueError("Reduction parameter unknown.")

    _stub("pytorch_lightning")
    _stub("pytorch_lightning.metrics")
    _stub("pytorch_lightning.metrics.metric", Metric=Metric)
    _stub("pytorch_lightning.metrics.functional")
    _stub("pytorch_lightning.metrics.functional.classification", stat_scores_multiple_classes=stat_scores_multiple_classes)
    _stub("pytorch_lightning.metrics.functional.reduction", reduce=reduce)
    _stub("skimage")
    _stub("skimage.draw", polygon=None)
    _stub("streamingflow.utils.tools", gen_dx_bx=None)
    inst = importlib.import_module("streamingflow.utils.instance")
    met = importlib.import_module("streamingflow.metrics")
    return SimpleNamespace(instance=inst, metrics=met)
