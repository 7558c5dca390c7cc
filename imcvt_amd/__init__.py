"""imcvt_amd — MI355X-native implementation of ImCvt's H.265 intra encode hot path.

Only what the path needs: csrc/ (HIP kernels + the C-ABI library) and hevc.py (host mirror of the
reference interface).  The oracle under /oracle is test infrastructure and is never imported from here.
"""
from .hevc import (DeviceEncoder, HEVCImageEncoder, HEVCImageEncoderBatch, load_library,  # noqa: F401
                   padded, stream_bound, writeHEVCImageFile)
