"""Host-side Python plumbing for libppo_ba.so (tests and bench.py only; the product is the HIP library)."""
from . import binding, pose, synth  # noqa: F401
