"""MI355X-native ultrasound beamforming core behind the ogl_beamformer_lib C ABI.

The product is ogl_beamforming_amd/libogl_beamformer_lib.so (hand-written gfx950 HIP
kernels + C++ host, built from csrc/); this package is only the Python-side binding and
the synthetic acquisition generator used by tests and bench.py.  Importing it does not
load the library; `lib.library()` does and raises ImportError when the .so is missing.
"""
from . import params  # noqa: F401
