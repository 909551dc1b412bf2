"""ctypes binding of include/lilliput_hip.h (Part B: batch extension, Part C: Go API mirror)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# lilliput.go:24-31 as LILLIPUT_* codes
ERR_NAMES = {
    0: "ok",
    1: "ErrInvalidImage",
    2: "ErrDecodingFailed",
    3: "ErrBufTooSmall",
    4: "unsupported by the device path",
    5: "device error (no GPU / HIP failure)",
    6: "ErrFrameBufNoPixels",
    7: "ErrEncodeTimeout",
    8: "io.EOF",
    9: "ErrSkipNotSupported",
    10: "ErrGifEncoderNeedsDecoder",
}
ImageOpsNoResize, ImageOpsFit, ImageOpsResize = 0, 1, 2  # ops.go:18-22
JpegQuality = 1  # opencv.go:44 (CV_IMWRITE_JPEG_QUALITY)
JpegProgressive = 2
PngCompression = 16  # opencv.go:45 (CV_IMWRITE_PNG_COMPRESSION)
WebpQuality = 64  # opencv.go:46 (CV_IMWRITE_WEBP_QUALITY); above 100 = lossless (webp.cpp:466-470)
WebpMethod, WebpFilterStrength, WebpFilterType, WebpAutofilter, WebpPartitions, WebpSegments, WebpPreprocessing, WebpThreadLevel, WebpPalette = range(1000, 1009)  # webp.hpp:13-23


class LilliputError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        detail = ""
        try:
            detail = lib().lilliput_hip_last_error().decode()
        except Exception:
            pass
        super().__init__("%s: %s %s" % (what, ERR_NAMES.get(code, str(code)), detail))


def lib_path():
    # LILLIPUT_HIP_LIB: another build of the same library (A/B measurements of kernel variants); the default is the in-tree one
    return os.environ.get("LILLIPUT_HIP_LIB") or os.path.join(_HERE, "liblilliput_hip.so")


def build(force=False):
    """Compile every HIP translation unit for gfx950 and link liblilliput_hip.so in-tree."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.run(cmd + ["clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(cmd + ["all"], check=True)
    return lib_path()


class _Item(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_size_t), ("dst", C.c_void_p), ("dst_cap", C.c_size_t),
                ("dst_len", C.c_size_t), ("status", C.c_int), ("out_width", C.c_int), ("out_height", C.c_int)]


class _BatchOptions(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("resize_method", C.c_int), ("normalize_orientation", C.c_int),
                ("jpeg_quality", C.c_int), ("chunk", C.c_int), ("jpeg_progressive", C.c_int)]


class _ImageOptions(C.Structure):
    _fields_ = [("file_type", C.c_char_p), ("width", C.c_int), ("height", C.c_int), ("resize_method", C.c_int),
                ("normalize_orientation", C.c_int), ("encode_options", C.POINTER(C.c_int)), ("encode_options_len", C.c_size_t),
                ("max_encode_frames", C.c_int), ("max_encode_duration_ns", C.c_int64), ("encode_timeout_ns", C.c_int64),
                ("disable_animated_output", C.c_int), ("force_sdr", C.c_int)]


def lib():
    """Load liblilliput_hip.so; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError("liblilliput_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    if not os.environ.get("LILLIPUT_HIP_KEEP_RUNTIME_ENV"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime initialises (see lp_engine.cpp: LpRuntimeEnv)
    L = C.CDLL(path)
    L.lilliput_hip_last_error.restype = C.c_char_p
    L.lilliput_hip_batch_create.restype = C.c_void_p
    L.lilliput_hip_batch_create.argtypes = [C.c_int]
    L.lilliput_hip_batch_destroy.argtypes = [C.c_void_p]
    L.lilliput_hip_batch_transform.argtypes = [C.c_void_p, C.POINTER(_Item), C.c_size_t, C.POINTER(_BatchOptions)]
    L.lilliput_hip_batch_upload.argtypes = [C.c_void_p, C.POINTER(_Item), C.c_size_t]
    L.lilliput_hip_batch_upload2.argtypes = [C.c_void_p, C.POINTER(_Item), C.c_size_t, C.c_int]
    L.lilliput_hip_batch_run.argtypes = [C.c_void_p, C.POINTER(_BatchOptions)]
    L.lilliput_hip_batch_download.argtypes = [C.c_void_p, C.POINTER(_Item), C.c_size_t]
    L.lilliput_hip_batch_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.lilliput_hip_batch_set_subsequence.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    L.lilliput_hip_batch_resident_round.argtypes = [C.c_void_p, C.c_size_t]
    L.lilliput_hip_batch_ingest_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.lilliput_hip_batch_ingest_stats.restype = None
    L.lilliput_hip_batch_ingest_stats2.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.lilliput_hip_batch_ingest_stats2.restype = None
    L.lilliput_hip_host_alloc.restype = C.c_void_p
    L.lilliput_hip_host_alloc.argtypes = [C.c_size_t, C.c_int]
    L.lilliput_hip_host_free.argtypes = [C.c_void_p]
    L.lilliput_hip_host_free.restype = None
    L.lilliput_hip_host_register.argtypes = [C.c_void_p, C.c_size_t]
    L.lilliput_hip_host_unregister.argtypes = [C.c_void_p]
    L.lilliput_hip_host_is_pinned.argtypes = [C.c_void_p, C.c_size_t]
    L.lilliput_hip_set_ingest_mode.argtypes = [C.c_char_p]
    L.lilliput_hip_engine_pool_stats.argtypes = [C.POINTER(C.c_size_t)]
    L.lilliput_hip_engine_pool_stats.restype = None
    L.lilliput_hip_mem_info.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.lilliput_hip_node_create.restype = C.c_void_p
    L.lilliput_hip_node_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.lilliput_hip_node_destroy.argtypes = [C.c_void_p]
    L.lilliput_hip_node_device_count.argtypes = [C.c_void_p]
    L.lilliput_hip_node_transform.argtypes = [C.c_void_p, C.POINTER(_Item), C.c_size_t, C.POINTER(_BatchOptions)]
    L.lilliput_hip_node_device_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.lilliput_hip_node_device_stats.restype = None
    L.lilliput_hip_node_queue_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.lilliput_hip_node_queue_stats.restype = None
    L.lilliput_hip_decode_jpeg.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4
    L.lilliput_hip_decode_jpeg_coefs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lilliput_hip_decode_jpeg_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lilliput_new_decoder.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.lilliput_decoder_close.argtypes = [C.c_void_p]
    L.lilliput_decoder_header.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 6
    L.lilliput_decoder_description.restype = C.c_char_p
    L.lilliput_decoder_description.argtypes = [C.c_void_p]
    L.lilliput_decoder_icc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.lilliput_decoder_animation_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.lilliput_new_image_ops.restype = C.c_void_p
    L.lilliput_new_image_ops.argtypes = [C.c_int]
    L.lilliput_image_ops_close.argtypes = [C.c_void_p]
    L.lilliput_image_ops_clear.argtypes = [C.c_void_p]
    L.lilliput_image_ops_transform.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_ImageOptions), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    # Part A handles are pointers
    for name in ("opencv_mat_create", "opencv_mat_create_from_data", "opencv_mat_create_empty_from_data", "opencv_mat_crop",
                 "opencv_decoder_create", "opencv_encoder_create", "opencv_mat_get_data"):
        getattr(L, name).restype = C.c_void_p
    L.opencv_mat_create_from_data.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.opencv_mat_create_empty_from_data.argtypes = [C.c_int, C.c_void_p]
    L.opencv_mat_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.opencv_mat_release.argtypes = [C.c_void_p]
    L.opencv_mat_crop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.opencv_mat_resize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.opencv_mat_orientation_transform.argtypes = [C.c_int, C.c_void_p]
    L.opencv_mat_get_width.argtypes = [C.c_void_p]
    L.opencv_mat_get_height.argtypes = [C.c_void_p]
    L.opencv_mat_get_data.argtypes = [C.c_void_p]
    L.opencv_mat_reset.argtypes = [C.c_void_p]
    L.opencv_mat_clear_to_transparent.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.opencv_copy_to_region.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.opencv_copy_to_region_with_alpha.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.opencv_decoder_create.argtypes = [C.c_void_p]
    L.opencv_decoder_release.argtypes = [C.c_void_p]
    L.opencv_decoder_read_header.argtypes = [C.c_void_p]
    L.opencv_decoder_read_header.restype = C.c_bool
    L.opencv_decoder_read_data.argtypes = [C.c_void_p, C.c_void_p]
    L.opencv_decoder_read_data.restype = C.c_bool
    for name in ("opencv_decoder_get_width", "opencv_decoder_get_height", "opencv_decoder_get_pixel_type", "opencv_decoder_get_orientation"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.opencv_encoder_create.argtypes = [C.c_char_p, C.c_void_p]
    L.opencv_encoder_release.argtypes = [C.c_void_p]
    L.opencv_encoder_write.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_size_t]
    L.opencv_encoder_write.restype = C.c_bool
    L.lilliput_hip_set_lazy_host.argtypes = [C.c_int]
    L.lilliput_hip_set_progressive_entropy.argtypes = [C.c_int]
    L.lilliput_hip_set_progressive_entropy.restype = None
    L.lilliput_hip_mat_sync_host.argtypes = [C.c_void_p]
    _LIB = L
    return L


class ImageOptions:
    """ops.go:26-65."""

    def __init__(self, FileType=".jpeg", Width=0, Height=0, ResizeMethod=ImageOpsFit, NormalizeOrientation=False, EncodeOptions=None,
                 MaxEncodeFrames=0, MaxEncodeDuration=0, EncodeTimeout=0, DisableAnimatedOutput=False, ForceSdr=False):
        self.FileType = FileType
        self.Width = Width
        self.Height = Height
        self.ResizeMethod = ResizeMethod
        self.NormalizeOrientation = NormalizeOrientation
        self.EncodeOptions = dict(EncodeOptions or {})
        self.MaxEncodeFrames = MaxEncodeFrames
        self.MaxEncodeDuration = MaxEncodeDuration
        self.EncodeTimeout = EncodeTimeout
        self.DisableAnimatedOutput = DisableAnimatedOutput
        self.ForceSdr = ForceSdr


class Decoder:
    """lilliput.NewDecoder (lilliput.go:129-164) for the sources this build serves (JPEG)."""

    def __init__(self, buf):
        self._buf = np.frombuffer(bytes(buf), dtype=np.uint8)  # the Go caller keeps its []byte alive; so do we
        h = C.c_void_p()
        rc = lib().lilliput_new_decoder(self._buf.ctypes.data_as(C.c_void_p), C.c_size_t(self._buf.size), C.byref(h))
        if rc:
            raise LilliputError(rc, "NewDecoder")
        self._h = h

    def Header(self):
        v = [C.c_int() for _ in range(6)]
        rc = lib().lilliput_decoder_header(self._h, *[C.byref(x) for x in v])
        if rc:
            raise LilliputError(rc, "Header")
        keys = ("width", "height", "pixel_type", "orientation", "num_frames", "content_length")
        return dict(zip(keys, [x.value for x in v]))

    def Description(self):
        return lib().lilliput_decoder_description(self._h).decode()

    def AnimationInfo(self):
        """GIF sources: (LoopCount, FrameCount, Duration in ms, BackgroundColor ARGB) of giflib.go:126-178."""
        v = (C.c_int * 4)()
        rc = lib().lilliput_decoder_animation_info(self._h, v)
        if rc:
            raise LilliputError(rc, "AnimationInfo")
        return v[0], v[1], v[2], v[3] & 0xFFFFFFFF

    def ICC(self):
        out = C.create_string_buffer(32768)  # ICCProfileBufferSize
        n = lib().lilliput_decoder_icc(self._h, out, 32768)
        return out.raw[:n]

    def Close(self):
        if self._h:
            lib().lilliput_decoder_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass


class ImageOps:
    """lilliput.NewImageOps / ImageOps.Transform (ops.go:83-91, 352-444)."""

    def __init__(self, maxSize):
        self._h = lib().lilliput_new_image_ops(int(maxSize))
        if not self._h:
            raise MemoryError("NewImageOps")

    def Transform(self, decoder, opt, dst_cap=8 << 20):
        dst = np.empty(dst_cap, dtype=np.uint8)
        flat = []
        for k, v in opt.EncodeOptions.items():
            flat += [int(k), int(v)]
        arr = (C.c_int * max(1, len(flat)))(*flat)
        o = _ImageOptions(opt.FileType.encode(), opt.Width, opt.Height, opt.ResizeMethod, int(bool(opt.NormalizeOrientation)),
                          C.cast(arr, C.POINTER(C.c_int)), len(flat), opt.MaxEncodeFrames, int(opt.MaxEncodeDuration), int(opt.EncodeTimeout),
                          int(bool(opt.DisableAnimatedOutput)), int(bool(opt.ForceSdr)))
        n = C.c_size_t()
        rc = lib().lilliput_image_ops_transform(self._h, decoder._h, C.byref(o), dst.ctypes.data_as(C.c_void_p), C.c_size_t(dst_cap), C.byref(n))
        if rc:
            raise LilliputError(rc, "Transform")
        return dst[: n.value].tobytes()

    def Clear(self):
        lib().lilliput_image_ops_clear(self._h)

    def Close(self):
        if self._h:
            lib().lilliput_image_ops_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass


def parse_raw_frames(blob):
    """Frames kept by the ".bgra-frames" test sink of ImageOps.Transform: [(HxWxC uint8 array, duration_ms)]."""
    out, i = [], 0
    while i + 16 <= len(blob):
        w, h, cn, ms = np.frombuffer(blob, dtype=np.uint32, count=4, offset=i)
        n = int(w) * int(h) * int(cn)
        out.append((np.frombuffer(blob, dtype=np.uint8, count=n, offset=i + 16).reshape(int(h), int(w), int(cn)).copy(), int(ms)))
        i += 16 + n
    return out


class HostArena:
    """lilliput_hip_host_alloc: one pinned, device-mapped block the caller places its encoded sources in (what a service that reads
    network bytes straight into pinned memory has); put() returns numpy views the batch calls read in place."""

    def __init__(self, nbytes, device=0):
        self._p = lib().lilliput_hip_host_alloc(int(nbytes), int(device))
        if not self._p:
            raise MemoryError("lilliput_hip_host_alloc(%d)" % nbytes)
        self.nbytes = int(nbytes)
        self._used = 0
        self._buf = (C.c_uint8 * self.nbytes).from_address(self._p)

    def put(self, data, align=64):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        off = (self._used + align - 1) // align * align
        if off + a.size > self.nbytes:
            raise MemoryError("host arena exhausted")
        view = np.frombuffer(self._buf, dtype=np.uint8, count=a.size, offset=off)
        view[:] = a
        self._used = off + a.size
        return view

    def close(self):
        if self._p:
            self._buf = None
            lib().lilliput_hip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchItemResult:
    __slots__ = ("status", "data", "width", "height")

    def __init__(self, status, data, width, height):
        self.status, self.data, self.width, self.height = status, data, width, height


class Batch:
    """The additive batched entry point (Part B of lilliput_hip.h): n independent JPEG -> JPEG transforms."""

    def __init__(self, device=0):
        self._h = lib().lilliput_hip_batch_create(int(device))
        if not self._h:
            raise LilliputError(5, "lilliput_hip_batch_create")
        self._items = None
        self._keep = None

    def resident_round(self, max_src_len):
        """Images per launch of a resident run (one full round of the entropy decoder's workgroups on the device)."""
        return int(lib().lilliput_hip_batch_resident_round(self._h, int(max_src_len)))

    def set_subsequence(self, S, C_):
        lib().lilliput_hip_batch_set_subsequence(self._h, int(S), int(C_))

    def _make_items(self, sources, dst_cap):
        n = len(sources)
        items = (_Item * n)()
        keep = []
        for i, s in enumerate(sources):
            a = s if isinstance(s, np.ndarray) else np.frombuffer(bytes(s), dtype=np.uint8)
            d = np.empty(dst_cap, dtype=np.uint8)
            keep.append((a, d))
            items[i].src = a.ctypes.data
            items[i].src_len = a.size
            items[i].dst = d.ctypes.data
            items[i].dst_cap = dst_cap
        return items, keep

    @staticmethod
    def _opts(width, height, method, normalize, quality, chunk, progressive=False):
        return _BatchOptions(int(width), int(height), int(method), int(bool(normalize)), int(quality), int(chunk), int(bool(progressive)))

    def _results(self):
        out = []
        for it, (_, d) in zip(self._items, self._keep):
            out.append(BatchItemResult(it.status, d[: it.dst_len].tobytes() if it.status == 0 else b"", it.out_width, it.out_height))
        return out

    def transform(self, sources, width, height, method=ImageOpsFit, normalize=False, quality=85, dst_cap=1 << 20, chunk=0, progressive=False):
        self._items, self._keep = self._make_items(sources, dst_cap)
        o = self._opts(width, height, method, normalize, quality, chunk, progressive)
        lib().lilliput_hip_batch_transform(self._h, self._items, len(sources), C.byref(o))
        return self._results()

    # the same call with the item array built beforehand (what a caller that already holds its buffers pays: nothing per call)
    def prepare(self, sources, dst_cap=1 << 20):
        self._items, self._keep = self._make_items(sources, dst_cap)

    def transform_prepared(self, width, height, method=ImageOpsFit, normalize=False, quality=85, chunk=0, progressive=False):
        o = self._opts(width, height, method, normalize, quality, chunk, progressive)
        return lib().lilliput_hip_batch_transform(self._h, self._items, len(self._items), C.byref(o))

    def results(self):
        return self._results()

    def ingest_stats(self):
        """Of the last transform(): entropy-coded bytes that reached the device, host ms of the ingest threads (summed), ms the compute
        threads waited for a chunk, wall ms of the call; bytes copied through pinned slots, bytes the DMA engine read from the caller's
        own pages, ms inside hipHostRegister, NUMA node of the ingest threads."""
        v = (C.c_double * 8)()
        lib().lilliput_hip_batch_ingest_stats2(self._h, v)
        return {"staged_bytes": int(v[0]), "stage_ms": v[1], "stall_ms": v[2], "wall_ms": v[3], "copied_bytes": int(v[4]), "direct_bytes": int(v[5]),
                "register_ms": v[6], "numa_node": int(v[7])}

    # staged form (bench): inputs resident in HBM before run()
    def upload(self, sources, dst_cap=1 << 20, streams=0):
        self._items, self._keep = self._make_items(sources, dst_cap)
        rc = lib().lilliput_hip_batch_upload2(self._h, self._items, len(sources), int(streams))  # streams: engines the batch is split over (0 = default)
        if rc:
            raise LilliputError(rc, "batch_upload")

    def run(self, width, height, method=ImageOpsFit, normalize=False, quality=85, chunk=0, progressive=False):
        o = self._opts(width, height, method, normalize, quality, chunk, progressive)
        rc = lib().lilliput_hip_batch_run(self._h, C.byref(o))
        if rc:
            raise LilliputError(rc, "batch_run")

    def download(self):
        lib().lilliput_hip_batch_download(self._h, self._items, len(self._items))
        return self._results()

    def timings(self):
        ms = (C.c_float * 10)()
        r = C.c_int()
        lib().lilliput_hip_batch_timings(self._h, ms, C.byref(r))
        keys = ("unstuff_ms", "huffman_ms", "idct_ms", "color_ms", "resize_ms", "encode_ms", "huff_spec_ms", "huff_verify_ms", "huff_scan_ms", "huff_write_ms")
        d = dict(zip(keys, [float(x) for x in ms]))
        d["verify_rounds"] = r.value
        return d

    # stage-level access (parity tests)
    def decode_jpeg(self, data):
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        w, h, ch, o = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        cap = 1 << 20
        while True:
            out = np.empty(cap, dtype=np.uint8)
            rc = lib().lilliput_hip_decode_jpeg(self._h, a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(w), C.byref(h), C.byref(ch), C.byref(o))
            if rc == 3 and cap < w.value * h.value * ch.value:
                cap = w.value * h.value * ch.value
                continue
            if rc:
                raise LilliputError(rc, "decode_jpeg")
            return out[: w.value * h.value * ch.value].reshape(h.value, w.value, ch.value).copy(), o.value

    def decode_jpeg_coefs(self, data, comp):
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        bw, bh = C.c_int(), C.c_int()
        cap = 1 << 18
        while True:
            out = np.empty(cap, dtype=np.int16)
            rc = lib().lilliput_hip_decode_jpeg_coefs(self._h, a.ctypes.data, a.size, comp, out.ctypes.data, cap, C.byref(bw), C.byref(bh))
            if rc == 3 and cap < bw.value * bh.value * 64:
                cap = bw.value * bh.value * 64
                continue
            if rc:
                raise LilliputError(rc, "decode_jpeg_coefs")
            return out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64).copy()

    def decode_jpeg_plane(self, data, comp):
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        pw, ph = C.c_int(), C.c_int()
        cap = 1 << 18
        while True:
            out = np.empty(cap, dtype=np.uint8)
            rc = lib().lilliput_hip_decode_jpeg_plane(self._h, a.ctypes.data, a.size, comp, out.ctypes.data, cap, C.byref(pw), C.byref(ph))
            if rc == 3 and cap < pw.value * ph.value:
                cap = pw.value * ph.value
                continue
            if rc:
                raise LilliputError(rc, "decode_jpeg_plane")
            return out[: pw.value * ph.value].reshape(ph.value, pw.value).copy()

    def close(self):
        if self._h:
            lib().lilliput_hip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Node(Batch):
    """lilliput_hip_node_*: one process, several GPUs sharing one chunk queue (devices=None: every visible GPU)."""

    def __init__(self, devices=None):
        if devices is None:
            h = lib().lilliput_hip_node_create(None, 0)
        else:
            arr = (C.c_int * len(devices))(*devices)
            h = lib().lilliput_hip_node_create(arr, len(devices))
        if not h:
            raise LilliputError(5, "lilliput_hip_node_create")
        self._n = h
        self._h = None
        self._items = None
        self._keep = None

    def device_count(self):
        return lib().lilliput_hip_node_device_count(self._n)

    def transform(self, sources, width, height, method=ImageOpsFit, normalize=False, quality=85, dst_cap=1 << 20, chunk=0, progressive=False):
        self._items, self._keep = self._make_items(sources, dst_cap)
        o = self._opts(width, height, method, normalize, quality, chunk, progressive)
        lib().lilliput_hip_node_transform(self._n, self._items, len(sources), C.byref(o))
        return self._results()

    def transform_prepared(self, width, height, method=ImageOpsFit, normalize=False, quality=85, chunk=0, progressive=False):
        o = self._opts(width, height, method, normalize, quality, chunk, progressive)
        return lib().lilliput_hip_node_transform(self._n, self._items, len(self._items), C.byref(o))

    def device_stats(self):
        out = []
        for k in range(self.device_count()):
            v = (C.c_double * 2)()
            lib().lilliput_hip_node_device_stats(self._n, k, v)
            out.append({"images": int(v[0]), "staged_bytes": int(v[1])})
        return out

    def queue_stats(self):
        v = (C.c_double * 2)()
        lib().lilliput_hip_node_queue_stats(self._n, v)
        return {"chunks": int(v[0]), "stolen": int(v[1])}

    def close(self):
        if self._n:
            lib().lilliput_hip_node_destroy(self._n)
            self._n = None


def service_sim(sources, threads, jobs, width=256, height=256, quality=85, resize_method=ImageOpsFit, max_size=8192, keep=True, file_type=".jpeg", encode_options=None,
                dst_cap=0, part="C"):
    """N OS threads, each with one ImageOps, each doing NewDecoder -> Header -> Transform -> Close per request through Part C of the
    C ABI (csrc/lp_service_sim.c: the Go service of /root/reference/README.md:82-85 in plain C, no interpreter between the calls).
    encode_options: {key: value} (default {JpegQuality: quality}). Returns {"seconds", "ok", "jobs", "first_error", "outputs" (first
    response per distinct source), "latency_ms" (numpy, per job)}."""
    lib()  # the library itself first: the simulator links it by soname next to itself
    S = C.CDLL(os.path.join(_HERE, "liblilliput_service_sim.so"))
    bufs = [np.frombuffer(bytes(d), dtype=np.uint8) if not isinstance(d, np.ndarray) else d for d in sources]
    n = len(bufs)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    keep_cap = max(1 << 16, width * height * 4 + 4096, dst_cap) if keep else 0
    keep_buf = np.zeros(max(1, n * keep_cap), dtype=np.uint8)
    keep_len = (C.c_long * n)(*([0] * n))
    lat = np.zeros(max(1, jobs), dtype=np.float32)
    secs, err = C.c_double(0.0), C.c_int(0)
    eo = encode_options if encode_options is not None else {JpegQuality: quality}
    flat = [int(x) for kv in eo.items() for x in kv]
    enc = (C.c_int * max(1, len(flat)))(*flat)
    S.lilliput_service_sim_run3.restype = C.c_long
    S.lilliput_service_sim_run3.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t,
                                            C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    # part "A": every request as the opencv_* call sequence of unchanged ops.go / opencv.go (lp_service_sim.c one_request_part_a)
    ok = S.lilliput_service_sim_run3(1 if str(part).upper() == "A" else 0, ptrs, lens, n, int(threads), int(jobs), int(width), int(height), file_type.encode(), enc, len(flat), int(resize_method), int(max_size), int(dst_cap),
                                     C.byref(secs), C.byref(err), keep_buf.ctypes.data if keep else None, keep_cap, keep_len if keep else None, lat.ctypes.data)
    outs = [keep_buf[k * keep_cap: k * keep_cap + keep_len[k]].tobytes() if keep and 0 < keep_len[k] <= keep_cap else None for k in range(n)]
    return {"seconds": secs.value, "ok": int(ok), "jobs": int(jobs), "first_error": err.value, "outputs": outs, "latency_ms": lat[:jobs]}


def transform_one(data, width, height, method=ImageOpsFit, normalize=False, quality=85, device=0, dst_cap=1 << 20, progressive=False):
    """lilliput_hip_transform_one: one image through the process-wide dispatchers (calls in flight at once share batch launches)."""
    L = lib()
    L.lilliput_hip_transform_one.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(_BatchOptions), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    dst = np.empty(dst_cap, dtype=np.uint8)
    o = _BatchOptions(width, height, method, int(bool(normalize)), quality, 0, int(bool(progressive)))
    n = C.c_size_t(0)
    rc = L.lilliput_hip_transform_one(int(device), src.ctypes.data, src.size, C.byref(o), dst.ctypes.data, dst_cap, C.byref(n))
    if rc:
        raise LilliputError(rc, "transform_one")
    return dst[: n.value].tobytes()
