"""wavelets.jl_amd -- MI355X-native (gfx950, hand-written HIP) backend for the DWT hot path
of JuliaDSP/Wavelets.jl, with a host-side mirror of the reference's transform API.

    import wavelets_jl_amd as W            # (repo-root shim; the directory name has a dot)
    from wavelets_jl_amd import WT, wavelet, dwt, idwt
    x = W.to_device(np.random.randn(8192, 8192).astype(np.float32))
    y = dwt(x, wavelet(WT.db4))            # runs on the current HIP device/stream
    xr = idwt(y, wavelet(WT.db4))

The compute lives in libwavelets_mi355x.so (csrc/, C ABI in include/wavelets_mi355x.h).
"""
from . import wt as WT
from . import util as Util
from .wt import wavelet, OrthoFilter, GLS
from .util import (maxtransformlevels, sufficientpoweroftwo, detailindex, detailrange, detailn,
                   ndyadicscales, maketree, isvalidtree, iscube, isdyadic,
                   dyadicdetailindex, dyadicdetailrange, dyadicscalingrange, dyadicdetailn, maxdyadiclevel, tl2dyadiclevel,
                   dyadiclevel2tl, mirror, upsample, downsample, wcount, testfunction)
from .transforms import (dwt, idwt, dwt_, idwt_, dwt_oop_, idwt_oop_, dwtc, idwtc, dwtc_, idwtc_, wpt, iwpt, wpt_, iwpt_, dwt_batch, idwt_batch,
                         to_device, to_host, similar, julia_layout, is_julia_layout,
                         reserve_workspace, workspace_held, set_kernel_path, last_kernel, destroy_contexts, set_option, clear_options, options, set_arithmetic, get_arithmetic, arithmetic,
                         DimensionMismatch, ArgumentError, HIPError)
from .modwt import modwt, imodwt, maxmodwttransformlevels
from .threshold import (THType, HardTH, SoftTH, SemiSoftTH, SteinTH, BiggestTH, PosTH, NegTH, DEFAULT_TH, threshold, threshold_,
                        DNFT, VisuShrink, denoise, noisest, mad_, median, nspin2circ, circshift, DEFAULT_WAVELET)
from . import _lib

__all__ = [
    "WT", "Util", "wavelet", "OrthoFilter", "GLS",
    "dwt", "idwt", "dwt_", "idwt_", "dwt_oop_", "idwt_oop_", "dwtc", "idwtc", "dwtc_", "idwtc_", "wpt", "iwpt", "wpt_", "iwpt_", "dwt_batch", "idwt_batch",
    "maxtransformlevels", "sufficientpoweroftwo", "detailindex", "detailrange", "detailn",
    "ndyadicscales", "maketree", "isvalidtree", "iscube", "isdyadic",
    "dyadicdetailindex", "dyadicdetailrange", "dyadicscalingrange", "dyadicdetailn", "maxdyadiclevel", "tl2dyadiclevel",
    "dyadiclevel2tl", "mirror", "upsample", "downsample", "wcount", "testfunction",
    "to_device", "to_host", "similar", "julia_layout", "is_julia_layout",
    "reserve_workspace", "workspace_held", "set_kernel_path", "last_kernel", "destroy_contexts", "set_option", "clear_options", "options", "set_arithmetic", "get_arithmetic", "arithmetic",
    "DimensionMismatch", "ArgumentError", "HIPError",
    "modwt", "imodwt", "maxmodwttransformlevels",
    "THType", "HardTH", "SoftTH", "SemiSoftTH", "SteinTH", "BiggestTH", "PosTH", "NegTH", "DEFAULT_TH", "threshold", "threshold_",
    "DNFT", "VisuShrink", "denoise", "noisest", "mad_", "median", "nspin2circ", "circshift", "DEFAULT_WAVELET",
]
