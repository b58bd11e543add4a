"""Writes a model object out as a safetensors checkpoint directory in one of the two tensor naming
schemes the loader reads (test infrastructure)."""
import json
import os

from safetensors.torch import save_file


def write_checkpoint(model, path, scheme, shards=1, tie_head=False):
    cfg = model.config
    hq, hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    t = {}
    tr = model.transformer
    if scheme == "qwen":
        assert hq == hkv, "Qwen-1 is multi-head attention"
        t["transformer.wte.weight"] = tr.wte.weight
        t["transformer.ln_f.weight"] = tr.ln_f.weight
        t["lm_head.weight"] = model.lm_head.weight
        for i, b in enumerate(tr.h):
            p = f"transformer.h.{i}."
            t[p + "ln_1.weight"], t[p + "ln_2.weight"] = b.ln_1.weight, b.ln_2.weight
            t[p + "attn.c_attn.weight"] = b.attn.c_attn.weight
            if b.attn.c_attn.bias is not None:
                t[p + "attn.c_attn.bias"] = b.attn.c_attn.bias
            t[p + "attn.c_proj.weight"] = b.attn.c_proj.weight
            t[p + "mlp.w1.weight"], t[p + "mlp.w2.weight"] = b.mlp.w1.weight, b.mlp.w2.weight
            t[p + "mlp.c_proj.weight"] = b.mlp.c_proj.weight
        hf = {"model_type": "qwen", "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size,
              "num_hidden_layers": cfg.num_hidden_layers, "num_attention_heads": hq, "kv_channels": d,
              "intermediate_size": 2 * cfg.ffn_hidden_size, "layer_norm_epsilon": cfg.layer_norm_epsilon,
              "seq_length": cfg.max_position_embeddings, "rotary_emb_base": cfg.rope_theta}
    else:
        t["model.embed_tokens.weight"] = tr.wte.weight
        t["model.norm.weight"] = tr.ln_f.weight
        if not tie_head:
            t["lm_head.weight"] = model.lm_head.weight
        for i, b in enumerate(tr.h):
            p = f"model.layers.{i}."
            t[p + "input_layernorm.weight"] = b.ln_1.weight
            t[p + "post_attention_layernorm.weight"] = b.ln_2.weight
            q, k, v = b.attn.c_attn.weight.split([hq * d, hkv * d, hkv * d], dim=0)
            t[p + "self_attn.q_proj.weight"], t[p + "self_attn.k_proj.weight"] = q, k
            t[p + "self_attn.v_proj.weight"] = v
            if b.attn.c_attn.bias is not None:
                bq, bk, bv = b.attn.c_attn.bias.split([hq * d, hkv * d, hkv * d], dim=0)
                t[p + "self_attn.q_proj.bias"], t[p + "self_attn.k_proj.bias"] = bq, bk
                t[p + "self_attn.v_proj.bias"] = bv
            t[p + "self_attn.o_proj.weight"] = b.attn.c_proj.weight
            t[p + "mlp.up_proj.weight"], t[p + "mlp.gate_proj.weight"] = b.mlp.w1.weight, b.mlp.w2.weight
            t[p + "mlp.down_proj.weight"] = b.mlp.c_proj.weight
        hf = {"model_type": "llama", "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size,
              "num_hidden_layers": cfg.num_hidden_layers, "num_attention_heads": hq,
              "num_key_value_heads": hkv, "head_dim": d, "intermediate_size": cfg.ffn_hidden_size,
              "rms_norm_eps": cfg.layer_norm_epsilon, "max_position_embeddings": cfg.max_position_embeddings,
              "rope_theta": cfg.rope_theta, "tie_word_embeddings": tie_head}
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf, f)
    t = {k: v.contiguous().cpu() for k, v in t.items()}
    names = sorted(t)
    if shards == 1:
        save_file(t, os.path.join(path, "model.safetensors"))
    else:
        weight_map = {}
        for s in range(shards):
            fn = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
            part = {n: t[n] for n in names[s::shards]}
            save_file(part, os.path.join(path, fn))
            weight_map.update({n: fn for n in part})
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": weight_map}, f)
    return path


def write_byte_tokenizer(path):
    """A byte-level tokenizer (256 byte tokens + one end-of-text token, no merges) saved the way
    `AutoTokenizer.from_pretrained(path, local_files_only=True)` reads it: any text round-trips, every
    id is below 257, nothing is downloaded."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<|endoftext|>": 0}
    for i, ch in enumerate(sorted(pre_tokenizers.ByteLevel.alphabet())):
        vocab[ch] = i + 1
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>").save_pretrained(path)
    return path
