"""phantom_fhe_amd -- MI355X-native RNS polynomial-arithmetic core for PhantomFHE.

Host-side mirror of the reference's hot-path interface over libphantom_amd.so (HIP, gfx950).
There is no CPU fallback: constructing a PhantomContext without the built library or without a
HIP device raises.
"""
from .core import (PhantomContext, PhantomRelinKey, coeff_modulus_create, scheme_type, set_tuning,  # noqa: F401
                   to_device, to_host)
from .lib import EXPORTED, LIB_PATH, load  # noqa: F401
