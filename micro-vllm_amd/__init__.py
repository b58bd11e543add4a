"""MI355X-native paged-KV decode path behind micro-vllm's Python API.

Sub-packages keep the reference's module names so callers can switch by import path
alone:  core.cache_manager / core.paged_attention / core.layer.model_graph /
kernel.rmsnorm_add / kernel.rmsnorm_residual / kernel.swiglu ...

The directory name contains a hyphen, so it is loaded under the module name
`micro_vllm_amd` (see micro_vllm_amd.py at the repository root).
`install_reference_names()` additionally publishes the sub-packages as top-level
`core` and `kernel`, which is what the reference's own scripts import.
"""
import sys

__version__ = "0.1.0"


def install_reference_names():
    """Make `import core...` / `import kernel...` resolve to this package."""
    from . import core, kernel  # noqa: F401

    pkg = sys.modules[__name__]
    for short in ("core", "kernel", "models"):
        full = f"{pkg.__name__}.{short}"
        __import__(full)
        sys.modules[short] = sys.modules[full]
        prefix = full + "."
        for name, mod in list(sys.modules.items()):
            if name.startswith(prefix):
                sys.modules[short + "." + name[len(prefix):]] = mod
