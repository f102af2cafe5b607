"""Attention (AttnProcessor2_0 semantics), FeedForward(geglu), AdaLayerNorm placeholder."""
import torch
import torch.nn.functional as F
from torch import nn


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, **kw):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.group_norm = None
        self.added_kv_proj_dim = None
        self.upcast_attention = upcast_attention
        self.scale = dim_head ** -0.5
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        b, n, _ = hidden_states.shape
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.to_q(hidden_states)
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        d = q.shape[-1] // self.heads
        q = q.view(b, -1, self.heads, d).transpose(1, 2)
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, self.heads * d).to(q.dtype)
        return self.to_out[1](self.to_out[0](o))


CrossAttention = Attention


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", **kw):
        super().__init__()
        assert activation_fn == "geglu"
        inner = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class AdaLayerNorm(nn.Module):  # never instantiated on this path (num_embeds_ada_norm=None)
    def __init__(self, *a, **k):
        raise NotImplementedError
