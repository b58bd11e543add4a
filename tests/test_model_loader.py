"""Checkpoint loader (reference: core/model_loader.py:6, models/configs/qwen.json): a model written
out as safetensors in either naming scheme comes back with every parameter bit-identical under the
attribute paths the runners read, and every tensor-parallel rank cuts the same shards from it."""
import dataclasses
import sys, os

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from ckpt_util import write_checkpoint

import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.core.layer.model_graph import shard_block_weights
from micro_vllm_amd.core.model_loader import load_model, load_weights
from micro_vllm_amd.models.synthetic import SHAPES, build_synthetic_model

PARAMS = ("ln_1", "attn.c_attn", "attn.c_proj", "ln_2", "mlp.w1", "mlp.w2", "mlp.c_proj")


def _get(obj, path):
    for p in path.split("."):
        obj = getattr(obj, p)
    return obj


def _same_model(a, b):
    assert torch.equal(a.transformer.wte.weight, b.transformer.wte.weight)
    assert torch.equal(a.transformer.ln_f.weight, b.transformer.ln_f.weight)
    assert a.transformer.ln_f.eps == b.transformer.ln_f.eps
    assert torch.equal(a.lm_head.weight, b.lm_head.weight)
    assert len(a.transformer.h) == len(b.transformer.h)
    for x, y in zip(a.transformer.h, b.transformer.h):
        for path in PARAMS:
            px, py = _get(x, path), _get(y, path)
            assert torch.equal(px.weight, py.weight), path
            assert (px.bias is None) == (py.bias is None), path
            if px.bias is not None:
                assert torch.equal(px.bias, py.bias), path
        assert x.ln_1.eps == y.ln_1.eps and x.ln_2.eps == y.ln_2.eps


@pytest.mark.parametrize("scheme,shape,bias,shards", [
    ("qwen", "tiny-mha", True, 1), ("qwen", "tiny-mha", True, 3),
    ("llama", "tiny-gqa", False, 1), ("llama", "tiny-gqa", True, 2), ("llama", "tiny-d128", False, 4),
    ("llama", "tiny-gqa7", True, 2)])
def test_round_trip_is_bit_identical(tmp_path, scheme, shape, bias, shards):
    cfg = dataclasses.replace(SHAPES[shape], rope_theta=500000.0 if scheme == "llama" else 10000.0)
    src = build_synthetic_model(cfg, qkv_bias=bias, norm_jitter=0.1)
    write_checkpoint(src, str(tmp_path / "ckpt"), scheme, shards=shards)
    got, tok = load_model(str(tmp_path / "ckpt"), tokenizer=False)
    assert tok is None
    c = got.config
    for f in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
              "num_key_value_heads", "head_dim", "ffn_hidden_size", "layer_norm_epsilon",
              "max_position_embeddings", "rope_theta"):
        assert getattr(c, f) == getattr(cfg, f), f
    assert c.intermediate_size == 2 * cfg.ffn_hidden_size   # gate and up together, as the engine divides it
    _same_model(got, src)
    for world in (1, 2):
        for rank in range(world):
            for x, y in zip(got.transformer.h, src.transformer.h):
                a = shard_block_weights(x, c.num_attention_heads, c.num_key_value_heads, c.head_dim, rank, world)
                b = shard_block_weights(y, c.num_attention_heads, c.num_key_value_heads, c.head_dim, rank, world)
                for k in b:
                    assert (a[k] is None and b[k] is None) or torch.equal(a[k], b[k]), k


def test_tied_head_and_assignment(tmp_path):
    src = build_synthetic_model(SHAPES["tiny-gqa"])
    write_checkpoint(src, str(tmp_path / "c"), "llama", tie_head=True)
    got = load_weights(str(tmp_path / "c"))
    assert torch.equal(got.lm_head.weight, src.transformer.wte.weight)
    blk = got.transformer.h[0]
    blk.mlp.w1.weight = None              # what a runner does with a consumed source
    assert blk.mlp.w1.weight is None
    blk.ln_1.weight = blk.ln_1.weight.float()
    assert blk.ln_1.weight.dtype == torch.float32
    assert blk.attn.c_attn.bias is None


def test_other_dtypes_arrive_as_bf16(tmp_path):
    src = build_synthetic_model(SHAPES["tiny-gqa"])
    src.transformer.h[0].mlp.w1.weight = src.transformer.h[0].mlp.w1.weight.to(torch.float16)
    write_checkpoint(src, str(tmp_path / "c"), "llama")
    got = load_weights(str(tmp_path / "c"))
    w = got.transformer.h[0].mlp.w1.weight
    assert w.dtype == torch.bfloat16
    assert torch.equal(w, src.transformer.h[0].mlp.w1.weight.to(torch.bfloat16))


def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        load_weights(str(tmp_path))      # no config.json
    src = build_synthetic_model(SHAPES["tiny-gqa"])
    path = write_checkpoint(src, str(tmp_path / "c"), "llama")
    os.remove(os.path.join(path, "model.safetensors"))
    with pytest.raises(FileNotFoundError, match="safetensors"):
        load_weights(path)
    import json
    write_checkpoint(src, path, "llama")
    hf = json.load(open(os.path.join(path, "config.json")))
    hf["rope_scaling"] = {"rope_type": "llama3", "factor": 8.0}
    json.dump(hf, open(os.path.join(path, "config.json"), "w"))
    with pytest.raises(NotImplementedError, match="rope_scaling"):
        load_weights(path)


def test_load_model_returns_the_directory_s_tokenizer(tmp_path):
    """core/model_loader.py:7-26: tokenizer from local files only."""
    from ckpt_util import write_byte_tokenizer
    src = build_synthetic_model(SHAPES["tiny-gqa"])
    path = write_checkpoint(src, str(tmp_path / "c"), "llama")
    write_byte_tokenizer(path)
    model, tok = load_model(path)
    ids = tok.encode("héllo, wörld", add_special_tokens=True)
    assert tok.decode(ids, skip_special_tokens=True) == "héllo, wörld"
    assert max(ids) < model.config.vocab_size and tok.eos_token_id == 0
