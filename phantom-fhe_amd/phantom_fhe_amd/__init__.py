"""phantom_fhe_amd -- MI355X-native RNS polynomial-arithmetic core for PhantomFHE.

Host-side mirror of the reference's hot-path interface over libphantom_amd.so (HIP, gfx950).
There is no CPU fallback: constructing a PhantomContext without the built library or without a
HIP device raises.
"""
from .core import (DBaseConverter, PhantomContext, PhantomRelinKey, coeff_modulus_create, fnwt_1d, inwt_1d, scheme_type,  # noqa: F401
                   has_tuning, set_strict, set_tuning, stream_copy_rate, stream_rate, to_device, to_host)
from .lib import EXP_LIB_PATH, EXPORTED, LIB_PATH, load  # noqa: F401
