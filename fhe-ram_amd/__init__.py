"""fheram_amd — MI355X-native evaluator for the phantomzone-org/fhe-ram hot path
(Ram::read / read_prepare_write / write), behind a C ABI (include/fheram.h).

This package holds only what the path needs: `csrc/` (HIP kernels + the C-ABI library) and
the host-side mirror of the reference's public interface (Parameters, Address,
EvaluationKeysPrepared, Ram).  It never imports anything under oracle/.
"""
from .base import Base1D, Base2D, get_base_2d, reverse_bits_msb  # noqa: F401

try:  # the mirror of the reference API needs numpy + ctypes only; the .so is loaded lazily
    from .api import (Address, EvaluationKeysPrepared, FheRamError, FheUintPrepared, GLWESecret, GroupRam, Parameters, Ram,  # noqa: F401
                      cast_u8_to_signed, encode_coeff, expected_plain, galois_elements, library, library_path, noise_scale)
except ImportError as _e:  # pragma: no cover
    raise
