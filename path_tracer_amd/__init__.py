"""MI355X-native sample loop of DryPerspective/Path-Tracer (gfx950 HIP kernels behind a C-ABI)."""
from ._capi import Device, PtError, load_library, LIB_PATH, SIGNATURES  # noqa: F401
