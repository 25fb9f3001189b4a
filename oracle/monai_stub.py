"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (generativemodels_amd/).

Minimal stand-in for the `monai` symbols that the unmodified reference (`/root/reference/generative`) imports, so the
reference itself can be executed on CPU in the build container as the live oracle.
MONAI (>=1.3.0, setup.py:21 of the reference) is a third-party dependency that is NOT vendored under /root/reference and
is not installed here; on the hot path it contributes no arithmetic of its own, only thin wrappers over torch:
Convolution -> nn.ConvNd / nn.ConvTransposeNd (+ ADN ordering), MLPBlock -> 2 nn.Linear + GELU/GEGLU, Pool -> nn.AvgPoolNd,
SpatialPad / CenterSpatialCrop -> symmetric pad / centre slice. Those published semantics are restated here.

Import this module BEFORE any `generative` import. It is only usable where /root/reference exists (this container);
nothing in the `-m gpu` tests, smoke() or bench.py depends on it.
"""
import abc
import contextlib
import enum
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
    return m


if "monai" not in sys.modules:
    for _n in [
        "monai", "monai.networks", "monai.networks.blocks", "monai.networks.blocks.mlp", "monai.networks.layers",
        "monai.networks.layers.factories", "monai.networks.layers.utils", "monai.utils", "monai.utils.misc",
        "monai.utils.enums", "monai.utils.type_conversion", "monai.config", "monai.inferers", "monai.data",
        "monai.transforms", "monai.engines", "monai.engines.trainer", "monai.engines.utils", "monai.metrics",
        "monai.metrics.metric", "monai.metrics.regression",
    ]:
        _mod(_n)
    M = sys.modules
    M["monai"].__version__ = "stub"

    class StrEnum(str, enum.Enum):
        def __str__(self):
            return self.value

        def __repr__(self):
            return self.value

    def ensure_tuple_rep(tup, dim):
        if isinstance(tup, torch.Tensor):
            tup = tup.tolist()
        if not isinstance(tup, (list, tuple)):
            return (tup,) * dim
        if len(tup) == dim:
            return tuple(tup)
        raise ValueError(f"Sequence must have length {dim}, got {len(tup)}.")

    def min_version(*a, **k):
        return True

    def optional_import(module, version="", version_checker=None, name="", as_type="default", **kw):
        import importlib

        try:
            m = importlib.import_module(module)
            return (getattr(m, name) if name else m), True
        except Exception:
            if as_type == "base":
                return type("_Lazy", (), {}), False
            return None, False

    class LossReduction(StrEnum):
        NONE = "none"
        MEAN = "mean"
        SUM = "sum"

    class MetricReduction(StrEnum):
        NONE = "none"
        MEAN = "mean"
        SUM = "sum"
        MEAN_BATCH = "mean_batch"
        SUM_BATCH = "sum_batch"
        MEAN_CHANNEL = "mean_channel"
        SUM_CHANNEL = "sum_channel"

    for _tgt in ("monai.utils", "monai.utils.misc", "monai.utils.enums"):
        for _k, _v in dict(StrEnum=StrEnum, ensure_tuple_rep=ensure_tuple_rep, min_version=min_version,
                           optional_import=optional_import, LossReduction=LossReduction,
                           MetricReduction=MetricReduction).items():
            setattr(M[_tgt], _k, _v)
    M["monai.utils"].convert_data_type = lambda *a, **k: (a[0], None, None)
    M["monai.utils.type_conversion"].convert_to_dst_type = lambda src, dst, **k: (src, None, None)

    class IgniteInfo:
        OPT_IMPORT_VERSION = "0.4.4"

    M["monai.config"].IgniteInfo = IgniteInfo

    ACT = {"relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "gelu": nn.GELU, "silu": nn.SiLU, "swish": nn.SiLU,
           "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "prelu": nn.PReLU, "elu": nn.ELU}

    class GEGLU(nn.Module):
        def forward(self, x):
            x, gate = x.chunk(2, dim=-1)
            return x * F.gelu(gate)

    ACT["geglu"] = GEGLU

    def get_act_layer(name):
        if name == "":
            return nn.Identity()
        args = {}
        if isinstance(name, (tuple, list)):
            name, args = name
        return ACT[str(name).lower()](**args)

    class _Act:
        RELU = "relu"
        LEAKYRELU = "leakyrelu"
        PRELU = "prelu"
        GELU = "gelu"
        SIGMOID = "sigmoid"
        TANH = "tanh"

        def __getitem__(self, k):
            return ACT[str(k).lower()]

    Act = _Act()

    class _Pool:
        AVG = "avg"
        MAX = "max"

        def __getitem__(self, key):
            kind, dim = key
            return {("avg", 1): nn.AvgPool1d, ("avg", 2): nn.AvgPool2d, ("avg", 3): nn.AvgPool3d,
                    ("max", 1): nn.MaxPool1d, ("max", 2): nn.MaxPool2d, ("max", 3): nn.MaxPool3d}[
                (str(kind).lower(), dim)]

    Pool = _Pool()

    def get_pool_layer(name, spatial_dims=1):
        args = {}
        if isinstance(name, (tuple, list)):
            name, args = name
        return Pool[name, spatial_dims](**args)

    M["monai.networks.layers"].Act = Act
    M["monai.networks.layers"].get_pool_layer = get_pool_layer
    M["monai.networks.layers.factories"].Pool = Pool
    M["monai.networks.layers.factories"].Act = Act
    M["monai.networks.layers.utils"].get_act_layer = get_act_layer

    class ADN(nn.Sequential):
        """Registers only the letters present in `ordering`, in that order, as children "N"/"D"/"A"."""

        def __init__(self, ordering="NDA", in_channels=None, act="RELU", norm=None, norm_dim=None, dropout=None,
                     dropout_dim=1):
            super().__init__()
            op = {"A": None, "D": None, "N": None}
            if norm is not None:
                nm = norm if isinstance(norm, str) else norm[0]
                nargs = {} if isinstance(norm, str) else dict(norm[1])
                # monai.networks.layers.utils.get_norm_layer: the layer gets the channel count plus the user's keyword arguments
                if nm.lower() == "instance":
                    op["N"] = [nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d][norm_dim - 1](in_channels, **nargs)
                elif nm.lower() == "batch":
                    op["N"] = [nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d][norm_dim - 1](in_channels, **nargs)
                elif nm.lower() == "group":
                    op["N"] = nn.GroupNorm(num_channels=in_channels, **nargs)
                else:
                    raise NotImplementedError(nm)
            if act is not None:
                op["A"] = get_act_layer(act)
            if dropout is not None:
                op["D"] = nn.Dropout(float(dropout))
            for item in ordering.upper():
                if op[item] is not None:
                    self.add_module(item, op[item])

    class Convolution(nn.Sequential):
        def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, adn_ordering="NDA",
                     act="PRELU", norm="INSTANCE", dropout=None, dropout_dim=1, dilation=1, groups=1, bias=True,
                     conv_only=False, is_transposed=False, padding=None, output_padding=None):
            super().__init__()
            if padding is None:
                padding = (kernel_size - 1) // 2 * dilation
            if is_transposed:
                if output_padding is None:
                    output_padding = strides - 1
                ct = [nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d][spatial_dims - 1]
                conv = ct(in_channels, out_channels, kernel_size=kernel_size, stride=strides, padding=padding,
                          output_padding=output_padding, groups=groups, bias=bias, dilation=dilation)
            else:
                ct = [nn.Conv1d, nn.Conv2d, nn.Conv3d][spatial_dims - 1]
                conv = ct(in_channels, out_channels, kernel_size=kernel_size, stride=strides, padding=padding,
                          dilation=dilation, groups=groups, bias=bias)
            self.add_module("conv", conv)
            if conv_only:
                return
            if act is None and norm is None and dropout is None:
                return
            self.add_module("adn", ADN(adn_ordering, out_channels, act, norm, spatial_dims, dropout, dropout_dim))

    class MLPBlock(nn.Module):
        def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0, act="GELU", dropout_mode="vit"):
            super().__init__()
            mlp_dim = mlp_dim or hidden_size
            self.linear1 = nn.Linear(hidden_size, mlp_dim) if act != "GEGLU" else nn.Linear(hidden_size, mlp_dim * 2)
            self.linear2 = nn.Linear(mlp_dim, hidden_size)
            self.fn = get_act_layer(act)
            self.drop1 = nn.Dropout(dropout_rate)
            self.drop2 = nn.Dropout(dropout_rate)

        def forward(self, x):
            return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))

    for _k, _v in dict(ADN=ADN, Convolution=Convolution, MLPBlock=MLPBlock).items():
        setattr(M["monai.networks.blocks"], _k, _v)
    M["monai.networks.blocks.mlp"].MLPBlock = MLPBlock

    @contextlib.contextmanager
    def eval_mode(*nets):
        tr = [n.training for n in nets]
        try:
            with torch.no_grad():
                yield [n.eval() for n in nets]
        finally:
            for n, t in zip(nets, tr):
                n.train(t)

    M["monai.networks"].eval_mode = eval_mode

    class Inferer(abc.ABC):
        @abc.abstractmethod
        def __call__(self, inputs, network, *a, **k):
            ...

    class SimpleInferer(Inferer):
        def __call__(self, inputs, network, *a, **k):
            return network(inputs, *a, **k)

    M["monai.inferers"].Inferer = Inferer
    M["monai.inferers"].SimpleInferer = SimpleInferer
    M["monai.data"].decollate_batch = lambda batch, *a, **k: list(torch.unbind(batch, 0))

    class SpatialPad:
        def __init__(self, spatial_size, **k):
            self.s = list(spatial_size)

        def __call__(self, img):  # img: C, spatial...
            pads = []
            for cur, tgt in zip(reversed(img.shape[1:]), reversed(self.s)):
                w = max(tgt - cur, 0)
                pads += [w // 2, w - w // 2]
            return F.pad(img, pads)

    class CenterSpatialCrop:
        def __init__(self, roi_size, **k):
            self.r = list(roi_size)

        def __call__(self, img):
            sl = [slice(None)]
            for cur, r in zip(img.shape[1:], self.r):
                if r <= 0 or r >= cur:
                    sl.append(slice(None))
                    continue
                st = cur // 2 - r // 2
                sl.append(slice(st, st + r))
            return img[tuple(sl)]

    class Transform:
        pass

    for _k, _v in dict(SpatialPad=SpatialPad, CenterSpatialCrop=CenterSpatialCrop, Transform=Transform).items():
        setattr(M["monai.transforms"], _k, _v)

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    M["monai.engines"].PrepareBatch = _Dummy
    M["monai.engines"].default_prepare_batch = lambda *a, **k: None
    M["monai.engines.trainer"].Trainer = _Dummy
    M["monai.engines.utils"].CommonKeys = _Dummy
    M["monai.engines.utils"].default_metric_cmp_fn = None
    M["monai.engines.utils"].default_prepare_batch = None
    M["monai.metrics.metric"].Metric = _Dummy
    M["monai.metrics.regression"].RegressionMetric = _Dummy
