"""Drop-in ops with the reference's names (kernel/rmsnorm_add.py, kernel/rmsnorm_residual.py,
kernel/swiglu.py), each a thin shim over one C-ABI call into the gfx950 library."""
