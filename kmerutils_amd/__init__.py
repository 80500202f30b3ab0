"""kmerutils_amd -- MI355X-native k-mer generation + counting + sketching hot path of kmerutils.

Product code: hand-written gfx950 HIP kernels in csrc/ behind the C-ABI of include/kmu.h (libkmu.so), plus a thin
host-side mirror of the reference's SeqSketcherT / KmerCountT interfaces (sketching.py, kmercount.py).
No CPU fallback: the compute classes raise when the built HIP library or a HIP device is missing.
"""
from . import _abi  # noqa: F401
