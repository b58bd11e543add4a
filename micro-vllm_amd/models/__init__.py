"""Model objects the runners consume (synthetic random-weight models with the attribute names
the reference's runner reads from a HuggingFace Qwen-1 checkpoint)."""
from .synthetic import SHAPES, SyntheticConfig, build_synthetic_model  # noqa: F401
