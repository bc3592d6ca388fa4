"""yacht_amd — MI355X-native containment/ANI engine for YACHT's hot path.

Only what the path needs lives here: the HIP kernels and C ABI (csrc/, built into
lib/libyacht_hip.so), the ctypes binding (_lib), the RefDB handle (engine) and the host-side
mirror of the reference's Python interface for this path (utils, hypothesis_recovery_src, ...).
"""
__version__ = "0.1.0"
