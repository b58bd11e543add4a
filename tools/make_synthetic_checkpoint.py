"""Writes a random-weight checkpoint directory (safetensors shards + config.json + a byte-level
tokenizer) for one of the synthetic shapes, so that the loader, the engine and the server can be
run end to end without a trained model:
   python tools/make_synthetic_checkpoint.py tiny-d128 /tmp/ckpt [llama|qwen] [SHARDS=2]
   python -m micro_vllm_amd.api_server /tmp/ckpt --port 8000"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.models import SHAPES, build_synthetic_model
from ckpt_util import write_byte_tokenizer, write_checkpoint

shape, path = sys.argv[1], sys.argv[2]
scheme = sys.argv[3] if len(sys.argv) > 3 else "llama"
shards = int(sys.argv[4]) if len(sys.argv) > 4 else 2
model = build_synthetic_model(SHAPES[shape], "cpu", seed=1234, qkv_bias=(scheme == "qwen"))
write_checkpoint(model, path, scheme, shards=shards)
write_byte_tokenizer(path)
print(f"{shape} ({scheme} names, {shards} shard(s)) -> {path}")
