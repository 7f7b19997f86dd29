"""fast_limo_amd -- MI355X-native registration hot path of fast_LIMO (kNN -> plane fit ->
point-to-plane residual/Jacobian -> iterated ESKF update) behind the reference's
Localizer / Mapper API.  The compute path is hand-written HIP (gfx950) behind a C ABI
(``include/flimo_c.h``); this package only holds the ctypes binding used by tests/bench,
the build driver and the synthetic-scene generators."""
__version__ = "0.1.0"
