"""phones-las_amd: MI355X-native Listen-Attend-Spell training path (drop-in for the hot path of
sciforce/phones-las).  Import it as ``phones_las_amd`` (see the shim module at the repo root).

The compute path is liblas_hip.so (hand-written HIP for gfx950, C-ABI in include/las_hip.h).
There is no CPU fallback: ops raise if the library or a GPU is missing."""
__version__ = '0.1.0'
