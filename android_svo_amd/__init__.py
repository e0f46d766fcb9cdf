"""android_svo_amd -- MI355X-native SVO hot path (sparse image alignment, align2D,
depth-filter update) behind a C-ABI (include/svo_hip.h, csrc/).

The product path is the HIP library `csrc/libsvo_hip.so`; there is no CPU
fallback: importing `android_svo_amd.hip` raises if the library is missing.
"""
__all__ = ["synth"]
