"""MI355X-native hot path of astra-vision/rain-rendering.

The directory name carries a hyphen (it mirrors the reference repository name), so
import it with ``importlib.import_module("rain-rendering_amd")`` or through the alias
module ``rain_rendering_amd`` at the repository root.

Layout
  csrc/            HIP kernels + the C ABI (include/rainhip.h) -> csrc/librainhip.so
  hip_backend.py   ctypes binding of the C ABI, drop-table packing
  common/          host-side mirror of the reference's Python interface for this path
                   (Generator, DBManager, RainRenderer, colour conversions, solid angles)
  synthetic.py     seeded synthetic inputs in the reference's on-disk formats
  main.py          main.py-compatible command line
"""
__version__ = "0.1.0"
