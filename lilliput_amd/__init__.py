"""lilliput_amd -- MI355X-native hot path of discord/lilliput's ImageOps.Transform.

The product is ``liblilliput_hip.so`` (hand-written HIP for gfx950 behind the reference's cgo C ABI, see
``include/lilliput_hip.h``). This package is a thin ctypes front-end that mirrors the Go API names
(Decoder / ImageOps / ImageOptions) for tests, the smoke check and the benchmark. There is no CPU
fallback: loading fails loudly when the shared library has not been built.
"""
from .binding import (  # noqa: F401
    ERR_NAMES,
    Batch,
    BatchItemResult,
    Decoder,
    HostArena,
    ImageOps,
    ImageOptions,
    JpegProgressive,
    JpegQuality,
    PngCompression,
    WebpQuality,
    WebpMethod,
    WebpSegments,
    LilliputError,
    Node,
    ImageOpsFit,
    ImageOpsNoResize,
    ImageOpsResize,
    build,
    lib,
    lib_path,
    parse_raw_frames,
    service_sim,
    transform_one,
)
