"""Random-weight Llama-shaped models exposing the attribute paths the reference's runner
hard-codes (core/layer/model_graph.py:95-116, :196, :222-223; names documented in
models/configs/qwen.json):

    model.config.{vocab_size,num_attention_heads,num_key_value_heads,hidden_size,...}
    model.transformer.wte.weight            [V, hid]
    model.transformer.h[i].ln_1/.ln_2       .weight [hid], .eps
    model.transformer.h[i].attn.c_attn      .weight [(Hq+2Hkv)D, hid], .bias or None
    model.transformer.h[i].attn.c_proj      .weight [hid, Hq*D]
    model.transformer.h[i].mlp.w1 / .w2     .weight [I, hid]   (w1 = up, w2 = gate)
    model.transformer.h[i].mlp.c_proj       .weight [hid, I]
    model.transformer.ln_f, model.lm_head.weight [V, hid]

`config.intermediate_size` follows the Qwen-1 convention the reference relies on: it is
2*I (gate and up together); `config.ffn_hidden_size` is I.
"""
from dataclasses import dataclass

import torch


@dataclass
class SyntheticConfig:
    name: str
    vocab_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    ffn_hidden_size: int  # I
    layer_norm_epsilon: float = 1e-6
    max_position_embeddings: int = 8192
    rope_theta: float = 10000.0   # base of the rotary frequencies

    @property
    def intermediate_size(self) -> int:  # Qwen-1 meaning: gate + up
        return 2 * self.ffn_hidden_size

    def weight_bytes(self, tp: int = 1, head_sharded: bool = False) -> int:
        """bf16 bytes one decode step streams per GPU: sharded layers + the LM head (replicated as in
        the reference, or 1/tp of it when it is cut by vocabulary rows)."""
        h, d = self.hidden_size, self.head_dim
        per_layer = h * (self.num_attention_heads + 2 * self.num_key_value_heads) * d \
            + self.num_attention_heads * d * h + 3 * h * self.ffn_hidden_size
        head = h * self.vocab_size // (tp if head_sharded else 1)
        return 2 * (self.num_hidden_layers * per_layer // tp + head)

    def kv_bytes_per_token(self, tp: int = 1) -> int:
        """bf16 K+V bytes one cached token occupies per GPU over all layers."""
        return 2 * 2 * self.num_hidden_layers * (self.num_key_value_heads // tp) * self.head_dim


SHAPES = {
    # TinyLlama-1.1B
    "tinyllama-1.1b": SyntheticConfig("tinyllama-1.1b", 32000, 2048, 22, 32, 4, 64, 5632, 1e-5),
    # Llama-3-8B
    "llama3-8b": SyntheticConfig("llama3-8b", 128256, 4096, 32, 32, 8, 128, 14336, 1e-5),
    # Llama-3-70B
    "llama3-70b": SyntheticConfig("llama3-70b", 128256, 8192, 80, 64, 8, 128, 28672, 1e-5),
    # one rank's slice of Llama-3-70B under 8-way tensor parallelism (heads, kv heads and FFN columns
    # divided by 8; hidden size, vocabulary and layer count unchanged): per-GPU kernel work of that
    # configuration without the all-reduces
    "llama3-70b-tp8shard": SyntheticConfig("llama3-70b-tp8shard", 128256, 8192, 80, 8, 1, 128, 3584, 1e-5),
    # one rank's slice of Llama-3-8B under 2-, 4- and 8-way tensor parallelism (same convention)
    "llama3-8b-tp2shard": SyntheticConfig("llama3-8b-tp2shard", 128256, 4096, 32, 16, 4, 128, 7168, 1e-5),
    "llama3-8b-tp4shard": SyntheticConfig("llama3-8b-tp4shard", 128256, 4096, 32, 8, 2, 128, 3584, 1e-5),
    "llama3-8b-tp8shard": SyntheticConfig("llama3-8b-tp8shard", 128256, 4096, 32, 4, 1, 128, 1792, 1e-5),
    # small shapes for tests (all kernel paths, seconds on a CPU)
    "tiny-gqa": SyntheticConfig("tiny-gqa", 512, 256, 2, 8, 2, 64, 512, 1e-6, 2048),
    "tiny-mha": SyntheticConfig("tiny-mha", 512, 256, 2, 4, 4, 64, 384, 1e-6, 2048),
    "tiny-d128": SyntheticConfig("tiny-d128", 1024, 512, 2, 8, 2, 128, 1024, 1e-5, 2048),
    # query heads per kv head that are no power of two (the ratios of the Qwen2 family: 7, 6, 5)
    "tiny-gqa7": SyntheticConfig("tiny-gqa7", 512, 448, 2, 7, 1, 64, 512, 1e-6, 2048),
    "tiny-gqa6-d128": SyntheticConfig("tiny-gqa6-d128", 512, 768, 2, 6, 1, 128, 512, 1e-6, 2048),
    # eight kv heads: divisible by every tensor-parallel degree up to 8
    "tiny-kv8": SyntheticConfig("tiny-kv8", 512, 256, 2, 16, 8, 64, 512, 1e-6, 2048),
}


class _Param:
    """A bare parameter holder: `.weight` (and optionally `.bias`, `.eps`)."""

    def __init__(self, weight, bias=None, eps=None):
        self.weight, self.bias, self.eps = weight, bias, eps

    def to(self, device):
        self.weight = self.weight.to(device)
        if self.bias is not None:
            self.bias = self.bias.to(device)
        return self


class _Attn:
    def __init__(self, c_attn, c_proj):
        self.c_attn, self.c_proj = c_attn, c_proj


class _Mlp:
    def __init__(self, w1, w2, c_proj):
        self.w1, self.w2, self.c_proj = w1, w2, c_proj


class _Block:
    def __init__(self, ln_1, attn, ln_2, mlp):
        self.ln_1, self.attn, self.ln_2, self.mlp = ln_1, attn, ln_2, mlp


class _Transformer:
    def __init__(self, wte, h, ln_f):
        self.wte, self.h, self.ln_f = wte, h, ln_f


class SyntheticModel:
    def __init__(self, config, transformer, lm_head):
        self.config, self.transformer, self.lm_head = config, transformer, lm_head

    def eval(self):
        return self

    def parameters(self):
        t = self.transformer
        yield t.wte.weight
        for b in t.h:
            for p in (b.ln_1, b.attn.c_attn, b.attn.c_proj, b.ln_2, b.mlp.w1, b.mlp.w2, b.mlp.c_proj):
                if p.weight is not None:
                    yield p.weight
                if p.bias is not None:
                    yield p.bias
        yield t.ln_f.weight
        yield self.lm_head.weight

    def to(self, device):
        t = self.transformer
        t.wte.to(device)
        for b in t.h:
            for p in (b.ln_1, b.attn.c_attn, b.attn.c_proj, b.ln_2, b.mlp.w1, b.mlp.w2, b.mlp.c_proj):
                p.to(device)
        t.ln_f.to(device)
        self.lm_head.to(device)
        return self


def build_synthetic_model(cfg: SyntheticConfig, device="cpu", seed: int = 1234,
                          dtype=torch.bfloat16, qkv_bias: bool = False, std: float = 0.02,
                          norm_jitter: float = 0.0, num_layers: int = None) -> SyntheticModel:
    """Linear weights N(0, std^2), norm weights 1 (+ N(0, norm_jitter^2)), bias N(0, std^2) or
    absent.  Generation is seeded per device type, so every tensor-parallel rank that calls
    this with the same arguments holds the same full model before it slices its shard."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    h, d = cfg.hidden_size, cfg.head_dim
    hq, hkv, inter = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.ffn_hidden_size
    layers = cfg.num_hidden_layers if num_layers is None else num_layers

    def lin(n, k):
        w = torch.empty(n, k, dtype=torch.float32 if device == "cpu" else dtype, device=device)
        w.normal_(0.0, std, generator=gen)
        return w.to(dtype)

    def norm():
        w = torch.ones(h, dtype=torch.float32, device=device)
        if norm_jitter:
            w += torch.empty(h, dtype=torch.float32, device=device).normal_(0.0, norm_jitter, generator=gen)
        return _Param(w.to(dtype), eps=cfg.layer_norm_epsilon)

    wte = _Param(lin(cfg.vocab_size, h))
    blocks = []
    for _ in range(layers):
        bias = None
        if qkv_bias:
            bias = torch.empty((hq + 2 * hkv) * d, dtype=torch.float32, device=device).normal_(
                0.0, std, generator=gen).to(dtype)
        attn = _Attn(_Param(lin((hq + 2 * hkv) * d, h), bias=bias), _Param(lin(h, hq * d)))
        mlp = _Mlp(_Param(lin(inter, h)), _Param(lin(inter, h)), _Param(lin(h, inter)))
        blocks.append(_Block(norm(), attn, norm(), mlp))
    model = SyntheticModel(cfg, _Transformer(wte, blocks, norm()), _Param(lin(cfg.vocab_size, h)))
    if num_layers is not None:
        # a slice of a bigger shape: keep the config honest about what was built
        import dataclasses
        model.config = dataclasses.replace(cfg, num_hidden_layers=layers)
    return model
