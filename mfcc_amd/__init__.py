"""mfcc_amd -- MI355X-native MFCC hot path (gfx950 HIP kernels behind a C ABI).

Drop-in for the per-frame math of lambdaconcept/mfcc's ``mfcc/core`` and nothing else:
``MFCC`` mirrors the core's constructor (mfcc/core/mfcc.py:20-21), ``mfcc_open / mfcc_convert /
mfcc_close / show_dir_content`` mirror the host driver (software/main.c).  All arithmetic runs
in ``libmfcc_hip.so``; importing this package needs the library to be built (no CPU fallback).
"""
from ._lib import MfccHipError, LIB_PATH, kernel_source_hash, load as load_library  # noqa: F401
from . import wire  # noqa: F401  (serial wire format + power gate, software/serial.c, cepstrum.c)
from .api import (MFCC, MfccStream, PAD_NOTEBOOK, PAD_STREAM, get_table, lift_file, lifter,  # noqa: F401
                  make_params, mfcc_close, mfcc_convert, mfcc_open, num_frames, show_dir_content)

__all__ = ["MFCC", "MfccStream", "lift_file", "mfcc_open", "mfcc_convert", "mfcc_close", "show_dir_content", "lifter",
           "num_frames", "get_table", "make_params", "MfccHipError", "load_library",
           "PAD_NOTEBOOK", "PAD_STREAM", "wire"]
