"""CLIP ViT-L/14 text encoder restated for the CPU oracle (test infrastructure).  Third-party network ([3P] transformers
`CLIPTextModel`, not vendored in the reference): called at modules/inversion/diffusion_inversion.py:230,241
(`text_encoder(input_ids)[0]` = last_hidden_state after the final LayerNorm).  Published architecture: 12 pre-LN layers,
hidden 768, 12 heads, MLP 3072 with quick_gelu, causal mask, learned 77 positions, vocab 49408.  Parameter names equal the
transformers state-dict keys.  Pinned: tests/test_oracle_golden.py::test_clip_oracle_matches_transformers
loads these weights into the installed transformers.CLIPTextModel and compares hidden states (1e-4)."""
import torch
import torch.nn as nn


class Layer(nn.Module):
    def __init__(self, d=768, heads=12, mlp=3072):
        super().__init__()
        self.heads = heads
        self.layer_norm1 = nn.LayerNorm(d)
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.layer_norm2 = nn.LayerNorm(d)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, mlp), nn.Linear(mlp, d)

    def forward(self, x):
        b, n, d = x.shape
        h = self.layer_norm1(x)
        sp = lambda t: t.reshape(b, n, self.heads, d // self.heads).transpose(1, 2)
        q, k, v = sp(self.self_attn.q_proj(h)), sp(self.self_attn.k_proj(h)), sp(self.self_attn.v_proj(h))
        s = q @ k.transpose(-1, -2) * (d // self.heads) ** -0.5
        s = s + torch.full((n, n), float("-inf")).triu(1)
        a = s.softmax(-1) @ v
        x = x + self.self_attn.out_proj(a.transpose(1, 2).reshape(b, n, d))
        h = self.mlp.fc1(self.layer_norm2(x))
        return x + self.mlp.fc2(h * torch.sigmoid(1.702 * h))


class CLIPTextModel(nn.Module):
    def __init__(self, vocab=49408, d=768, layers=12):
        super().__init__()
        tm = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(vocab, d)
        tm.embeddings.position_embedding = nn.Embedding(77, d)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([Layer(d) for _ in range(layers)])
        tm.final_layer_norm = nn.LayerNorm(d)
        self.text_model = tm

    def forward(self, ids):
        tm = self.text_model
        x = tm.embeddings.token_embedding(ids) + tm.embeddings.position_embedding.weight[None, : ids.shape[1]]
        for l in tm.encoder.layers:
            x = l(x)
        return (tm.final_layer_norm(x),)


def build_clip(seed=0):
    from .unet import synthetic_tensor
    with torch.device("meta"):
        m = CLIPTextModel()
    m = m.to_empty(device="cpu")
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "embedding" in name:
                import zlib
                g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF)
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(synthetic_tensor("clip." + name, p.shape, seed))
    return m.eval()
