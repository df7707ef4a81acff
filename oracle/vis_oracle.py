"""Oracle (test infrastructure): ViS / SummaryMixing aggregator, CPU fp32.

Restates /root/reference/src/tformer_lin.py as pure functions of a
``state_dict`` with the reference's key names.  Nothing here is used by the
product path.

Key names (dumped from the reference module):
  pos_emb1D                                              [num_clusters, D]
  transformer.layers.{l}.0.mixers.{h}.local_norm.{weight,bias}   [f]
  transformer.layers.{l}.0.mixers.{h}.summary_norm.{weight,bias} [s]
  transformer.layers.{l}.0.mixers.{h}.{s,f}.{weight,bias}        [s|f, D]
  transformer.layers.{l}.0.mixers.{h}.c.{weight,bias}            [c, s+f]
  transformer.layers.{l}.0.projection.{weight,bias}              [D, H*c]
  transformer.layers.{l}.1.net.0.{weight,bias}   LayerNorm(D)
  transformer.layers.{l}.1.net.1.{weight,bias}   Linear(D, D)
  transformer.layers.{l}.1.net.3.{weight,bias}   Linear(D, D)
  linear_head.0.{weight,bias}  LayerNorm(D);  linear_head.1.{weight,bias} Linear(D, G)
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


def vis_config_from_state_dict(sd):
    """Recover (D, depth, nheads, f, s, c, G, num_clusters) from tensor shapes."""
    D = sd["pos_emb1D"].shape[1]
    num_clusters = sd["pos_emb1D"].shape[0]
    depth = 0
    while f"transformer.layers.{depth}.0.projection.weight" in sd:
        depth += 1
    nheads = 0
    while f"transformer.layers.0.0.mixers.{nheads}.f.weight" in sd:
        nheads += 1
    f = sd["transformer.layers.0.0.mixers.0.f.weight"].shape[0]
    s = sd["transformer.layers.0.0.mixers.0.s.weight"].shape[0]
    c = sd["transformer.layers.0.0.mixers.0.c.weight"].shape[0]
    G = sd["linear_head.1.weight"].shape[0]
    return dict(input_dim=D, depth=depth, nheads=nheads, dimensions_f=f,
                dimensions_s=s, dimensions_c=c, num_outputs=G,
                num_clusters=num_clusters)


def init_vis_state_dict(num_outputs, input_dim, depth, nheads, dimensions_f,
                        dimensions_s, dimensions_c, num_clusters=100, seed=0):
    """Seeded init with the same distributions as the reference's default torch
    init (tformer_lin.py:86-94: randn pos-emb, nn.Linear kaiming-uniform(a=sqrt5)
    + uniform bias, LayerNorm ones/zeros).  The draw ORDER is this oracle's own
    (it is a recipe for synthetic weights, not a claim of RNG parity with
    ``ViS.__init__``).
    """
    g = torch.Generator().manual_seed(seed)

    def linear(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
        return w, b

    sd = OrderedDict()
    sd["pos_emb1D"] = torch.randn(num_clusters, input_dim, generator=g)
    for l in range(depth):
        p = f"transformer.layers.{l}.0."
        for h in range(nheads):
            q = p + f"mixers.{h}."
            sd[q + "local_norm.weight"] = torch.ones(dimensions_f)
            sd[q + "local_norm.bias"] = torch.zeros(dimensions_f)
            sd[q + "summary_norm.weight"] = torch.ones(dimensions_s)
            sd[q + "summary_norm.bias"] = torch.zeros(dimensions_s)
            sd[q + "s.weight"], sd[q + "s.bias"] = linear(dimensions_s, input_dim)
            sd[q + "f.weight"], sd[q + "f.bias"] = linear(dimensions_f, input_dim)
            sd[q + "c.weight"], sd[q + "c.bias"] = linear(dimensions_c, dimensions_s + dimensions_f)
        sd[p + "projection.weight"], sd[p + "projection.bias"] = linear(input_dim, nheads * dimensions_c)
        p = f"transformer.layers.{l}.1.net."
        sd[p + "0.weight"] = torch.ones(input_dim)
        sd[p + "0.bias"] = torch.zeros(input_dim)
        sd[p + "1.weight"], sd[p + "1.bias"] = linear(input_dim, input_dim)
        sd[p + "3.weight"], sd[p + "3.bias"] = linear(input_dim, input_dim)
    sd["linear_head.0.weight"] = torch.ones(input_dim)
    sd["linear_head.0.bias"] = torch.zeros(input_dim)
    sd["linear_head.1.weight"], sd["linear_head.1.bias"] = linear(num_outputs, input_dim)
    return sd


def perturb_norm_params(sd, seed=1):
    """Make LayerNorm gains/biases non-trivial so parity tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if "norm" in k or k.endswith("net.0.weight") or k.endswith("net.0.bias") \
                or k.startswith("linear_head.0."):
            if k.endswith("weight"):
                sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
            else:
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return sd


def _gelu(x):
    # torch.nn.GELU() default = exact erf form (tformer_lin.py:20,22,24,57)
    return F.gelu(x)


def summary_mixing(sd, prefix, x):
    """tformer_lin.py:18-26 (SummaryMixing.forward) for one head."""
    f_dim = sd[prefix + "f.weight"].shape[0]
    s_dim = sd[prefix + "s.weight"].shape[0]
    local = F.linear(x, sd[prefix + "f.weight"], sd[prefix + "f.bias"])
    local = _gelu(F.layer_norm(local, (f_dim,), sd[prefix + "local_norm.weight"],
                               sd[prefix + "local_norm.bias"]))
    time = F.linear(x, sd[prefix + "s.weight"], sd[prefix + "s.bias"])
    time = _gelu(F.layer_norm(torch.mean(time, dim=1), (s_dim,),
                              sd[prefix + "summary_norm.weight"],
                              sd[prefix + "summary_norm.bias"]))
    time = time.unsqueeze(1).repeat(1, x.shape[1], 1)
    return _gelu(F.linear(torch.cat([local, time], dim=-1),
                          sd[prefix + "c.weight"], sd[prefix + "c.bias"]))


def multi_head_summary(sd, prefix, x, nheads):
    """tformer_lin.py:39-48 (MultiHeadSummary.forward)."""
    outs = [summary_mixing(sd, prefix + f"mixers.{h}.", x) for h in range(nheads)]
    outs = torch.cat(outs, dim=-1)
    return F.linear(outs, sd[prefix + "projection.weight"], sd[prefix + "projection.bias"])


def feed_forward(sd, prefix, x):
    """tformer_lin.py:51-61 (FeedForward)."""
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[prefix + "0.weight"], sd[prefix + "0.bias"])
    y = _gelu(F.linear(y, sd[prefix + "1.weight"], sd[prefix + "1.bias"]))
    return F.linear(y, sd[prefix + "3.weight"], sd[prefix + "3.bias"])


def vis_forward(sd, x, return_tokens=False):
    """tformer_lin.py:97-106 (ViS.forward) + :73-77 (SummaryTransformer.forward).

    x: f32 [B, 100, D] (any extra middle dims are flattened like
    ``rearrange('b ... d -> b (...) d')``).  Returns f32 [B, G].
    """
    cfg = vis_config_from_state_dict(sd)
    x = x.reshape(x.shape[0], -1, x.shape[-1]) + sd["pos_emb1D"]
    for l in range(cfg["depth"]):
        x = multi_head_summary(sd, f"transformer.layers.{l}.0.", x, cfg["nheads"]) + x
        x = feed_forward(sd, f"transformer.layers.{l}.1.net.", x) + x
    tokens = x
    x = x.mean(dim=1)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd["linear_head.0.weight"], sd["linear_head.0.bias"])
    out = F.linear(x, sd["linear_head.1.weight"], sd["linear_head.1.bias"])
    return (out, tokens) if return_tokens else out


def vis_loss_and_grads(sd, x, target):
    """MSELoss(mean) forward + autograd backward (vit.py:129,163-178)."""
    leaf = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    pred = vis_forward(leaf, x)
    loss = F.mse_loss(pred, target)
    loss.backward()
    grads = OrderedDict((k, v.grad) for k, v in leaf.items())
    return loss.detach(), pred.detach(), grads


def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999),
               eps=1e-8, weight_decay=0.0):
    """torch.optim.AdamW(amsgrad=False) single step, restated (main.py:180-183:
    ``AdamW(lr=args.lr, amsgrad=False, weight_decay=0.)``).  In-place on the
    dict-of-tensors arguments; ``step`` is the 1-based step count.
    """
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k in params:
        p, g = params[k], grads[k]
        if weight_decay != 0.0:
            p.mul_(1.0 - lr * weight_decay)
        exp_avg[k].mul_(b1).add_(g, alpha=1.0 - b1)
        exp_avg_sq[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (exp_avg_sq[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(exp_avg[k], denom, value=-lr / bc1)


# ---------------------------------------------------------------------------
# ViT baseline (src/vit.py:49-115) -- secondary model on the same loops
# ---------------------------------------------------------------------------
def vit_attention(sd, prefix, x, heads):
    """vit.py:62-74 (Attention.forward)."""
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
    qkv = F.linear(y, sd[prefix + "to_qkv.weight"]).chunk(3, dim=-1)
    B, N, inner = qkv[0].shape
    dh = inner // heads
    q, k, v = (t.reshape(B, N, heads, dh).permute(0, 2, 1, 3) for t in qkv)
    dots = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    attn = torch.softmax(dots, dim=-1)
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(B, N, inner)
    return F.linear(out, sd[prefix + "to_out.weight"])


def vit_forward(sd, x, heads):
    """vit.py:107-115 (ViT.forward)."""
    x = x.reshape(x.shape[0], -1, x.shape[-1]) + sd["pos_emb1D"]
    l = 0
    while f"transformer.layers.{l}.0.to_qkv.weight" in sd:
        x = vit_attention(sd, f"transformer.layers.{l}.0.", x, heads) + x
        x = feed_forward(sd, f"transformer.layers.{l}.1.net.", x) + x
        l += 1
    x = x.mean(dim=1)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd["linear_head.0.weight"], sd["linear_head.0.bias"])
    return F.linear(x, sd["linear_head.1.weight"], sd["linear_head.1.bias"])
