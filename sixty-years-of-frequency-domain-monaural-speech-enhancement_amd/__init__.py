"""MI355X-native engine for the reference's STFT -> network -> iSTFT decode path.

Host side (Python) mirrors the reference's model classes and `enhance(args)`
drivers; all arithmetic runs in hand-written HIP kernels behind the C-ABI of
`include/se_engine.h` (libse_engine.so, built in-tree by `__graft_entry__.build()`).
There is no CPU / PyTorch fallback: creating an engine without the built
library, or without a gfx950 GPU, raises.
"""
from . import synth, schemas  # noqa: F401
