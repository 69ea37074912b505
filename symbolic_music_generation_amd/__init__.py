"""MI355X-native Transformer-XL / Reformer hot path for tokenized-MIDI language modelling.

Drop-in surface mirrors `musicnlp.models` / `musicnlp.trainer` of StefanHeng/Symbolic-Music-Generation; the
arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI declared in `include/musicxl.h`.
"""
__version__ = '0.1.0'
