"""Checkpoint directory -> the model object the runners consume, plus its tokenizer.

Same call as the reference's loader (core/model_loader.py:6: `load_model(model_path, device)` ->
`(model, tokenizer)`), but without instantiating a transformers model: the runners only ever read
parameter tensors through the attribute paths of models/configs/qwen.json
(`transformer.h[i].attn.c_attn.weight`, ...), so the loader maps the tensor names of a
safetensors checkpoint onto exactly those paths and nothing of the checkpoint's own Python code
is imported or run.

Two naming schemes are understood:

  * Qwen-1 (`model_type: "qwen"`): the names ARE the attribute paths (transformer.h.N.attn.c_attn
    fused q|k|v with bias, mlp.w1 = up, mlp.w2 = gate, mlp.c_proj = down; config
    `intermediate_size` counts gate and up together, `kv_channels` is the head size);
  * Llama family (`llama`, `mistral`, `qwen2`, ...: model.layers.N.self_attn.{q,k,v,o}_proj,
    mlp.{gate,up,down}_proj, input_layernorm / post_attention_layernorm, model.norm,
    model.embed_tokens, lm_head or tied embeddings): q, k and v (and their biases, where the
    checkpoint has them) are concatenated into c_attn on access.  These checkpoints already store
    q and k in the half-split rotary convention the attention kernels apply.

Tensors are read lazily, one parameter at a time, from memory-mapped files: the runner's
`prepare_weights` touches one block, cuts this rank's shard, relays it out on the device and
(with `release_source_weights`) drops the source, so host memory never holds more than a block,
and all ranks of a node share the page cache.
"""
import json
import logging
import os
from typing import Dict, List, Optional

import torch

from ..models.synthetic import SyntheticConfig, SyntheticModel, _Attn, _Block, _Mlp, _Transformer

logger = logging.getLogger(__name__)

_UNSET = object()


class CheckpointFiles:
    """Name -> tensor over the safetensors files of one directory (single file or sharded with a
    `*.index.json` weight map).  Files are opened on first use and stay memory-mapped."""

    def __init__(self, model_path: str):
        self.path = model_path
        self._where: Dict[str, str] = {}
        self._open = {}
        index = [f for f in sorted(os.listdir(model_path)) if f.endswith(".safetensors.index.json")]
        if index:
            with open(os.path.join(model_path, index[0])) as f:
                self._where = dict(json.load(f)["weight_map"])
        else:
            files = [f for f in sorted(os.listdir(model_path)) if f.endswith(".safetensors")]
            if not files:
                raise FileNotFoundError(
                    f"no .safetensors file in {model_path!r} (pickled .bin checkpoints are not read: "
                    "loading them executes code from the file)")
            for fn in files:
                for name in self._file(fn).keys():
                    self._where[name] = fn

    def _file(self, fn):
        h = self._open.get(fn)
        if h is None:
            from safetensors import safe_open
            h = safe_open(os.path.join(self.path, fn), framework="pt", device="cpu")
            self._open[fn] = h
        return h

    def __contains__(self, name):
        return name in self._where

    def names(self):
        return self._where.keys()

    def get(self, name: str, dtype=torch.bfloat16) -> torch.Tensor:
        if name not in self._where:
            raise KeyError(f"tensor {name!r} is not in the checkpoint at {self.path!r}")
        t = self._file(self._where[name]).get_tensor(name)
        return t if t.dtype == dtype else t.to(dtype)


class LazyParam:
    """`.weight` / `.bias` read from the checkpoint every time they are asked for (several names =
    concatenated along dim 0) until something is assigned - the runner assigns the device copy of
    small tensors and None to sources it has consumed."""

    def __init__(self, files: CheckpointFiles, weight: List[str], bias: Optional[List[str]] = None,
                 eps: Optional[float] = None, dtype=torch.bfloat16):
        self._files, self._w_names, self._b_names, self.eps, self._dtype = files, weight, bias, eps, dtype
        self._w = self._b = _UNSET

    def _read(self, names):
        parts = [self._files.get(n, self._dtype) for n in names]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)

    @property
    def weight(self):
        return self._read(self._w_names) if self._w is _UNSET else self._w

    @weight.setter
    def weight(self, value):
        self._w = value

    @property
    def bias(self):
        if self._b is not _UNSET:
            return self._b
        return None if not self._b_names else self._read(self._b_names)

    @bias.setter
    def bias(self, value):
        self._b = value

    def to(self, device):
        self.weight = self.weight.to(device)
        b = self.bias
        if b is not None:
            self.bias = b.to(device)
        return self


def _config_from_hf(hf: dict, name: str) -> SyntheticConfig:
    mt = hf.get("model_type", "")
    heads = hf["num_attention_heads"]
    if mt == "qwen":  # Qwen-1: intermediate_size = gate + up, head size in kv_channels, MHA
        return SyntheticConfig(
            name=name, vocab_size=hf["vocab_size"], hidden_size=hf["hidden_size"],
            num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=heads,
            num_key_value_heads=heads, head_dim=hf.get("kv_channels", hf["hidden_size"] // heads),
            ffn_hidden_size=hf["intermediate_size"] // 2,
            layer_norm_epsilon=hf.get("layer_norm_epsilon", 1e-6),
            max_position_embeddings=hf.get("seq_length", hf.get("max_position_embeddings", 8192)),
            rope_theta=float(hf.get("rotary_emb_base", 10000.0)))
    scaling = hf.get("rope_scaling")
    if scaling and scaling.get("rope_type", scaling.get("type")) not in (None, "default"):
        raise NotImplementedError(f"rope_scaling {scaling!r}: only plain rotary tables are built")
    return SyntheticConfig(
        name=name, vocab_size=hf["vocab_size"], hidden_size=hf["hidden_size"],
        num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=heads,
        num_key_value_heads=hf.get("num_key_value_heads") or heads,
        head_dim=hf.get("head_dim") or hf["hidden_size"] // heads,
        ffn_hidden_size=hf["intermediate_size"],
        layer_norm_epsilon=hf.get("rms_norm_eps", 1e-6),
        max_position_embeddings=hf.get("max_position_embeddings", 8192),
        rope_theta=float(hf.get("rope_theta", 10000.0)))


def _qwen1_blocks(files, cfg):
    eps = cfg.layer_norm_epsilon
    for i in range(cfg.num_hidden_layers):
        p = f"transformer.h.{i}."
        bias = [p + "attn.c_attn.bias"] if p + "attn.c_attn.bias" in files else None
        yield _Block(
            LazyParam(files, [p + "ln_1.weight"], eps=eps),
            _Attn(LazyParam(files, [p + "attn.c_attn.weight"], bias),
                  LazyParam(files, [p + "attn.c_proj.weight"])),
            LazyParam(files, [p + "ln_2.weight"], eps=eps),
            _Mlp(LazyParam(files, [p + "mlp.w1.weight"]), LazyParam(files, [p + "mlp.w2.weight"]),
                 LazyParam(files, [p + "mlp.c_proj.weight"])))


def _llama_blocks(files, cfg):
    eps = cfg.layer_norm_epsilon
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        qkv = [p + f"self_attn.{x}_proj.weight" for x in "qkv"]
        bias = [p + f"self_attn.{x}_proj.bias" for x in "qkv"]
        if not all(b in files for b in bias):
            bias = None
        yield _Block(
            LazyParam(files, [p + "input_layernorm.weight"], eps=eps),
            _Attn(LazyParam(files, qkv, bias), LazyParam(files, [p + "self_attn.o_proj.weight"])),
            LazyParam(files, [p + "post_attention_layernorm.weight"], eps=eps),
            # w1 is the factor that is NOT activated (up), w2 the one SiLU is applied to (gate)
            _Mlp(LazyParam(files, [p + "mlp.up_proj.weight"]), LazyParam(files, [p + "mlp.gate_proj.weight"]),
                 LazyParam(files, [p + "mlp.down_proj.weight"])))


def load_weights(model_path: str) -> SyntheticModel:
    """The model object alone (no tokenizer)."""
    with open(os.path.join(model_path, "config.json")) as f:
        hf = json.load(f)
    cfg = _config_from_hf(hf, os.path.basename(os.path.normpath(model_path)))
    files = CheckpointFiles(model_path)
    eps = cfg.layer_norm_epsilon
    if "transformer.wte.weight" in files:
        blocks = list(_qwen1_blocks(files, cfg))
        wte = LazyParam(files, ["transformer.wte.weight"])
        ln_f = LazyParam(files, ["transformer.ln_f.weight"], eps=eps)
        head = "lm_head.weight"
    elif "model.embed_tokens.weight" in files:
        blocks = list(_llama_blocks(files, cfg))
        wte = LazyParam(files, ["model.embed_tokens.weight"])
        ln_f = LazyParam(files, ["model.norm.weight"], eps=eps)
        head = "lm_head.weight" if "lm_head.weight" in files else "model.embed_tokens.weight"  # tied
    else:
        raise ValueError(f"{model_path!r}: neither Qwen-1 (transformer.*) nor Llama-family (model.*) "
                         "tensor names found")
    model = SyntheticModel(cfg, _Transformer(wte, blocks, ln_f), LazyParam(files, [head]))
    model.checkpoint = files
    logger.info("checkpoint %s: %d layers, %d tensors", model_path, cfg.num_hidden_layers, len(files.names()))
    return model


def load_tokenizer(model_path: str, trust_remote_code: bool = False):
    """Fast tokenizer, then the slow one, from local files only (core/model_loader.py:7-26).  The
    reference passes trust_remote_code=True (Qwen-1's tokenizer is Python shipped with the
    checkpoint); here that is the caller's explicit choice."""
    from transformers import AutoTokenizer
    try:
        return AutoTokenizer.from_pretrained(model_path, trust_remote_code=trust_remote_code,
                                             use_fast=True, local_files_only=True)
    except Exception as e:
        logger.warning("fast tokenizer failed (%s), trying the slow one", e)
        return AutoTokenizer.from_pretrained(model_path, trust_remote_code=trust_remote_code,
                                             use_fast=False, local_files_only=True)


def load_model(model_path: str, device=None, *, tokenizer: bool = True, trust_remote_code: bool = False):
    """-> (model, tokenizer).  `device` is accepted for the reference's signature; parameters stay
    in the memory-mapped files until a runner's `prepare_weights` moves this rank's shard."""
    model = load_weights(model_path)
    tok = load_tokenizer(model_path, trust_remote_code) if tokenizer else None
    return model, tok
