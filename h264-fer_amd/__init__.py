"""h264-fer_amd -- host-side mirror of the fer_h264 encoder interface over libferhip.

The product is the C-ABI shared library ``libferhip.so`` (HIP kernels for gfx950, see
``include/ferhip.h``).  This package only binds it with ctypes so that tests and the
benchmark can drive it the way the reference's GUI drives ``fer_h264::Starter``
(F/fer_h264.cpp:166-216): set parameters, feed pictures, collect the byte stream.
There is no CPU fallback: if the library or a GPU is missing, calls raise.
"""
from .ferhip import (FerHip, FerHipError, Decoder, DeviceBuffer, Y4MReader, decode_streams, unescape_nal, lib_path,  # noqa: F401
                     load_library, mb_unit, cavlc_blocks, mc_sub_mb_parts, MBU_QT, MBU_DEC4, MBU_DEC16, MBU_DECC, MBU_SKIP, TUNE_RESOLVE_WGS, TUNE_RESOLVE_GROUP, TUNE_SPECULATE, TUNE_OVERLAP_SORT)
from .synth import gen_frame, gen_frames, crop_to_mb  # noqa: F401
from .shard import gops_of_rank, merge_gop_streams, split_nals  # noqa: F401
