"""Oracle (test infrastructure): patch transform + ResNet-50 forward_extract, CPU fp32.

Restates /root/reference/src/resnet.py:155-170 (``forward_extract``), :73-93
(``Bottleneck.forward``), :98-136 (topology) and the patch transform of
/root/reference/pre_processing/compute_features_hdf5.py:49-51,119-120 as pure
functions of a state_dict with torchvision's ``resnet50`` key names
(``conv1.weight``, ``bn1.*``, ``layer{1..4}.{i}.conv{1,2,3}.weight``,
``layer*.{i}.bn{1,2,3}.*``, ``layer*.0.downsample.{0,1}.*``).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5


def transform_patch_u8(img_u8_hwc):
    """compute_features_hdf5.py:119-120 + :49-51.

    uint8 [.., H, W, 3] -> f32 [.., 3, H, W]:  permute(2,0,1); ConvertImageDtype
    (x / 255 in fp32); Normalize((x - mean) / std) per channel, fp32.
    """
    x = torch.as_tensor(img_u8_hwc)
    x = x.movedim(-1, -3).to(torch.float32) / 255.0
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    return (x - mean) / std


def init_resnet50_state_dict(seed=99, perturb_bn=True):
    """He-init as resnet.py:113-119 (normal(0, sqrt(2/(k*k*out)))), BN gamma=1
    beta=0; with ``perturb_bn`` the affine and running statistics are made
    non-trivial so BN folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(name, cout, cin, k):
        n = k * k * cout
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / n)

    def bn(name, c):
        if perturb_bn:
            sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
            sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
            sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
            sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        else:
            sd[name + ".weight"] = torch.ones(c)
            sd[name + ".bias"] = torch.zeros(c)
            sd[name + ".running_mean"] = torch.zeros(c)
            sd[name + ".running_var"] = torch.ones(c)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, (nblocks, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nblocks):
            p = f"layer{li}.{b}"
            conv(p + ".conv1", planes, inplanes, 1)
            bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3)
            bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1)
            bn(p + ".bn3", planes * 4)
            if b == 0:
                conv(p + ".downsample.0", planes * 4, inplanes, 1)
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    return sd


def _bn(sd, name, x):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, BN_EPS)


def bottleneck(sd, p, x, stride):
    """resnet.py:73-93.  Stride lives on the 3x3 conv (resnet.py:64)."""
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    return F.relu(out + x)


def forward_extract(sd, x, return_intermediates=False):
    """resnet.py:155-170.  x: f32 [n, 3, H, W] -> f32 [n, 2048] (H=W=224 or 256)."""
    inter = OrderedDict()
    x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)))
    inter["conv1"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    inter["maxpool"] = x
    for li, nblocks in enumerate(LAYERS, start=1):
        for b in range(nblocks):
            x = bottleneck(sd, f"layer{li}.{b}", x, 2 if (b == 0 and li > 1) else 1)
        inter[f"layer{li}"] = x
    x = F.avg_pool2d(x, 7)            # nn.AvgPool2d(7): 8x8 map -> top-left 7x7 only
    x = x.reshape(x.shape[0], -1)
    return (x, inter) if return_intermediates else x


def embed_patches(sd, patches_u8, batch=1):
    """compute_features_hdf5.py:116-123: per-patch (batch=1, literal) or batched
    forward of uint8 HWC patches.  Returns f32 [n, 2048]."""
    outs = []
    with torch.no_grad():
        for i in range(0, len(patches_u8), batch):
            outs.append(forward_extract(sd, transform_patch_u8(patches_u8[i:i + batch])))
    return torch.cat(outs, 0)


def init_resnet50_state_dict_wide(seed, running_stats=None):
    """A weight set with the dynamic range of a trained network instead of He-init's flat one: every output channel's
    weight row is multiplied by m_c (log-uniform over 10^-1.5 .. 10^0.5) and BN gamma is log-uniform over 0.1 .. 10, beta =
    0.2 gamma N(0,1).  The running statistics are what BatchNorm would have tracked for these weights -- calibrated by
    ``calibrate_bn`` on a batch of patches and kept as data (``running_stats``: name -> tensor; tests/golden/
    resnet50_wide_bn.npz) -- so running_var spans ~4 decades (~ m_c^2) and the folded scale gamma / sqrt(var) ~5, while the
    activations stay in the range a trained network produces (per-channel std = gamma).  Without ``running_stats`` the
    statistics are mean 0 / var 1 placeholders."""
    g = torch.Generator().manual_seed(seed)
    sd = init_resnet50_state_dict(seed=seed + 1, perturb_bn=False)
    for k in list(sd):
        if k.endswith("running_mean"):
            name = k[:-len(".running_mean")]
            c = sd[k].numel()
            conv = name.replace("bn", "conv") if "downsample" not in name else name[:-1] + "0"
            m = torch.pow(10.0, -1.5 + 2.0 * torch.rand(c, generator=g))
            sd[conv + ".weight"] = sd[conv + ".weight"] * m.view(-1, 1, 1, 1)
            gamma = torch.pow(10.0, -1.0 + 2.0 * torch.rand(c, generator=g))
            sd[name + ".weight"] = gamma
            sd[name + ".bias"] = 0.2 * gamma * torch.randn(c, generator=g)
            if running_stats is not None:
                sd[name + ".running_mean"] = torch.as_tensor(running_stats[name + ".running_mean"]).float().clone()
                sd[name + ".running_var"] = torch.as_tensor(running_stats[name + ".running_var"]).float().clone()
    return sd


def calibrate_bn(sd, x):
    """Set every BN's running_mean / running_var to the batch statistics (biased variance over N, H, W) its input has when
    ``x`` (f32 [n, 3, H, W], normalised) flows through the network layer by layer -- what train-mode BN converges to.
    Returns the statistics as a dict of numpy arrays (the fixture)."""
    stats = {}

    def bn(name, y):
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        sd[name + ".running_mean"], sd[name + ".running_var"] = mean, var
        stats[name + ".running_mean"], stats[name + ".running_var"] = mean.numpy().copy(), var.numpy().copy()
        return _bn(sd, name, y)

    with torch.no_grad():
        x = F.max_pool2d(F.relu(bn("bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3))), 3, 2, 1)
        for li, nblocks in enumerate(LAYERS, start=1):
            for b in range(nblocks):
                p = f"layer{li}.{b}"
                s = 2 if (b == 0 and li > 1) else 1
                out = F.relu(bn(p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
                out = F.relu(bn(p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=s, padding=1)))
                out = bn(p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
                if (p + ".downsample.0.weight") in sd:
                    x = bn(p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=s))
                x = F.relu(out + x)
    return stats
