"""MI355X-native V2VAE training hot path (HIP kernels behind a C ABI) with the reference's Python surface.

Importing the kernel layer requires the in-tree shared library (no CPU fallback); the pure-host
helpers in `synth` and `utility` import without it.
"""
__all__ = ["synth"]
