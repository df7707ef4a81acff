"""Oracle (test infrastructure): HE2RNA, the benchmark comparator of the reference -- /root/reference/src/he2rna.py:42-106,
built by pretrain_gtex.py:102-105 as ``HE2RNA(input_dim, layers=[256, 256], ks=[1, 2, 5, 10, 20, 50, 100], output_dim)``.

Pinned: tests/test_he2rna.py checks this restatement against tests/golden/he2rna.npz, produced by the reference's own
class (tests/golden/make_golden.py:gold_he2rna): eval forward, forward_fixed_k and autograd gradients.

The model (state-dict keys ``conv{i}.weight [out, in, 1]`` / ``conv{i}.bias``), for x [B, C, N] (channels x tiles):
  mask[b, n]  = 1 if max_c x[b, c, n] > 0 else 0                      (:94-95, over ALL C channels)
  h           = x[:, C - input_dim:]  -> ReLU(conv1x1) per hidden layer (dropout in training) -> last conv1x1   (:101-106)
  s[b, g, n]  = h[b, g, n] * mask[b, n]                               (:96)
  t           = top-k of s over the tiles, sorted descending          (:97)
  out_k[b, g] = sum_{j<k} t[b, g, j] * mask[b, j] / sum_{j<k} mask[b, j]      (:98 -- the mask of the FIRST k tile
                positions, not of the selected tiles; 0/0 = NaN when the first k tiles are all masked)
  eval: mean over ks of out_k, accumulated in list order (:88-91); training: one k drawn with np.random.choice (:85-86)."""
import torch
import torch.nn.functional as F


def tile_mask(x):
    return (x.max(dim=1, keepdim=True)[0] > 0).to(x.dtype)                      # [B, 1, N]


def scores(sd, x, input_dim):
    n_layers = len([k for k in sd if k.endswith(".weight")])
    h = x[:, x.shape[1] - input_dim:]
    for i in range(n_layers):
        h = F.conv1d(h, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"])
        if i + 1 < n_layers:
            h = F.relu(h)
    return h                                                                    # [B, G, N]


def forward_fixed_k(sd, x, k, input_dim):
    mask = tile_mask(x)
    s = scores(sd, x, input_dim) * mask
    t, _ = torch.topk(s, k, dim=2, largest=True, sorted=True)
    return torch.sum(t * mask[:, :, :k], dim=2) / torch.sum(mask[:, :, :k], dim=2)


def forward_eval(sd, x, ks, input_dim):
    pred = 0
    for k in ks:
        pred = pred + forward_fixed_k(sd, x, int(k), input_dim) / len(ks)
    return pred
