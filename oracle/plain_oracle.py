"""Oracle #2: the same path written out with plain matmul / softmax / layer-norm.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  No ``nn.Transformer*`` / ``nn.MultiheadAttention``
is used here, so this cross-checks oracle #1 and documents the arithmetic the HIP kernels reproduce.
It consumes a reference-layout ``state_dict`` (SURVEY.md section 8b key names).

Arithmetic sites restated (torch/ = site-packages/torch):
* packed in-projection          torch/nn/functional.py:5819-5852
* SDPA, scale 1/sqrt(dh)        torch/nn/functional.py:6386-6390, 6623-6631
* post-norm layer order         torch/nn/modules/transformer.py:1143-1153
* erf GELU, LN eps 1e-5         torch/nn/modules/transformer.py:1037, 1210-1211
* head / projection             /root/reference/models.py:37-44, 68-73
"""
import math

import torch


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mha(xq, xkv, w_in, b_in, w_out, b_out, nhead, causal):
    B, Tq, d = xq.shape
    Tk = xkv.shape[1]
    dh = d // nhead
    q = xq @ w_in[:d].T + b_in[:d]
    k = xkv @ w_in[d:2 * d].T + b_in[d:2 * d]
    v = xkv @ w_in[2 * d:].T + b_in[2 * d:]
    q = q.view(B, Tq, nhead, dh).transpose(1, 2)
    k = k.view(B, Tk, nhead, dh).transpose(1, 2)
    v = v.view(B, Tk, nhead, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf")), diagonal=1)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Tq, d)
    return o @ w_out.T + b_out


def positional_encoding(T, d):
    """/root/reference/models.py:10-14."""
    position = torch.arange(T).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
    pe = torch.zeros(T, d)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def project(sd, features):
    p = "feature_projection."
    h = features @ sd[p + "0.weight"].T + sd[p + "0.bias"]
    h = gelu(layer_norm(h, sd[p + "1.weight"], sd[p + "1.bias"]))
    h = h @ sd[p + "4.weight"].T + sd[p + "4.bias"]
    h = layer_norm(h, sd[p + "5.weight"], sd[p + "5.bias"])
    return h.unsqueeze(1) if h.dim() == 2 else h


def decode(sd, memory, captions, nhead, num_layers, collect=None):
    d = sd["embedding.weight"].shape[1]
    T = captions.shape[1]
    x = sd["embedding.weight"][captions] + positional_encoding(T, d)
    if collect is not None:
        collect["embed"] = x
    for i in range(num_layers):
        p = f"transformer_decoder.layers.{i}."
        sa = mha(x, x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                 sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], nhead, True)
        x = layer_norm(x + sa, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        ca = mha(x, memory, sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"],
                 sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"], nhead, False)
        x = layer_norm(x + ca, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        ff = gelu(x @ sd[p + "linear1.weight"].T + sd[p + "linear1.bias"])
        ff = ff @ sd[p + "linear2.weight"].T + sd[p + "linear2.bias"]
        x = layer_norm(x + ff, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
        if collect is not None:
            collect[f"layer{i}"] = x
    x = layer_norm(x, sd["transformer_decoder.norm.weight"], sd["transformer_decoder.norm.bias"])
    h = gelu(x @ sd["output_layer.0.weight"].T + sd["output_layer.0.bias"])
    return h @ sd["output_layer.3.weight"].T + sd["output_layer.3.bias"]


def forward(sd, features, captions, nhead, num_layers, collect=None):
    return decode(sd, project(sd, features), captions, nhead, num_layers, collect)


def label_smoothed_loss(logits, targets, eps=0.1):
    """Closed form of /root/reference/train.py:41-50: mean_{b,t}[lse - (1-eps) z_y - eps/V sum_v z_v]."""
    V = logits.shape[-1]
    lse = torch.logsumexp(logits, dim=-1)
    zy = logits.gather(-1, targets.unsqueeze(-1)).squeeze(-1)
    return (lse - (1 - eps) * zy - (eps / V) * logits.sum(-1)).mean()


def adamw_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update (torch/optim/adam.py _single_tensor_adam, decoupled wd)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v
