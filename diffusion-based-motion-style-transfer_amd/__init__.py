"""MI355X-native denoising engine behind the reference's Python call surface.

Import as `mst_amd` (see ../mst_amd.py).  Sub-packages `diffusion`, `model` and `utils` mirror the
reference modules of the same names (the drop-in boundary); `engine` wraps the C-ABI
library built from `csrc/` (include/mst_engine.h)."""
__version__ = "0.1.0"
