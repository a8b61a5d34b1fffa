"""raytracer_amd — MI355X-native per-ray hot path of tatref/raytracer.

The product is the HIP library `lib/librt_amd.so` behind the C ABI of
`include/rt_abi.h`; this package is the thin Python plumbing the tests, the
smoke test and bench.py use to reach it (the reference's own callers are Rust
and bind the same symbols through an `extern "C"` block).
"""
from .api import (Context, Scene2D, Progressive2D, Scene3D, World2D, ArrayWorld2D, Standalone3D, ArrayScene3D, RenderParams2D, RenderParams3D,  # noqa: F401
                  RtError, NBINS, write_png)
