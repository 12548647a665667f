"""ctypes binding of libocrs_amd.so (include/ocrs_amd.h).

The library is the product: there is NO Python/CPU fallback.  If the shared
object is missing or no GPU is visible, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCRS_AMD_LIB: an ablation build of the same library (ocrs_amd.build --variant), for A/B timing only
LIB_PATH = os.environ.get("OCRS_AMD_LIB") or os.path.join(_HERE, "libocrs_amd.so")

OCRS_OK = 0
STATUS_NAMES = {
    0: "OK", 1: "INVALID_ARGUMENT", 2: "MODEL_NOT_LOADED", 3: "MODEL_DIMS", 4: "RUN_FAILED", 5: "WRONG_OUTPUT",
    6: "IMAGE_SOURCE", 7: "DEVICE", 8: "IO", 9: "CAPACITY",
}

RUN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_float)),
                     C.POINTER(C.c_int64), C.POINTER(C.c_int))


class EngineParams(C.Structure):
    _fields_ = [("detection_model", C.c_void_p), ("recognition_model", C.c_void_p), ("debug", C.c_int),
                ("decode_method", C.c_int), ("beam_width", C.c_uint32), ("alphabet", C.c_char_p),
                ("allowed_chars", C.c_char_p), ("numerics", C.c_int), ("coalesce", C.c_int), ("coalesce_pages", C.c_int),
                ("coalesce_window_us", C.c_int), ("layout_threads", C.c_int), ("rec_max_pixels", C.c_int64)]


class GroupParams(C.Structure):
    _fields_ = [("detection_model", C.c_void_p), ("detection_model_len", C.c_size_t), ("recognition_model", C.c_void_p),
                ("recognition_model_len", C.c_size_t), ("devices", C.POINTER(C.c_int)), ("n_devices", C.c_size_t),
                ("debug", C.c_int), ("decode_method", C.c_int), ("beam_width", C.c_uint32), ("alphabet", C.c_char_p),
                ("allowed_chars", C.c_char_p), ("gather", C.c_int), ("numerics", C.c_int), ("coalesce", C.c_int),
                ("coalesce_pages", C.c_int), ("coalesce_window_us", C.c_int), ("layout_threads", C.c_int), ("rec_max_pixels", C.c_int64),
                ("min_block", C.c_int), ("shared_block", C.c_int)]


class RunOptions(C.Structure):
    _fields_ = [("timing", C.c_int)]


class TextCharC(C.Structure):
    _fields_ = [("ch", C.c_uint32), ("top", C.c_int32), ("left", C.c_int32), ("bottom", C.c_int32),
                ("right", C.c_int32)]


class OcrsError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status
        self.status_name = STATUS_NAMES.get(status, str(status))


# Every symbol include/ocrs_amd.h declares (tests check they are all exported).
DECLARED_SYMBOLS = [
    "ocrs_last_error", "ocrs_buffer_free", "ocrs_device_count", "ocrs_set_device", "ocrs_set_option", "ocrs_ctc_beam_search", "ocrs_model_load_file",
    "ocrs_model_load_bytes", "ocrs_model_from_callback", "ocrs_model_input_shape", "ocrs_model_run",
    "ocrs_model_flops", "ocrs_model_free", "ocrs_engine_new", "ocrs_engine_free", "ocrs_image_source_check_bytes",
    "ocrs_engine_prepare_input", "ocrs_engine_prepare_input_device", "ocrs_page_free", "ocrs_page_dims",
    "ocrs_page_image", "ocrs_engine_detect_words", "ocrs_engine_detect_words_batch",
    "ocrs_engine_detect_text_pixels", "ocrs_engine_detection_threshold", "ocrs_engine_find_text_lines",
    "ocrs_engine_find_text_lines_batch",
    "ocrs_engine_recognize_text", "ocrs_engine_recognize_text_batch", "ocrs_engine_recognize_tokens",
    "ocrs_engine_prepare_recognition_input", "ocrs_text_item_rotated_rect", "ocrs_rotated_rect_corners", "ocrs_engine_get_text", "ocrs_device_malloc", "ocrs_device_free",
    "ocrs_device_upload", "ocrs_device_synchronize", "ocrs_device_measure_peaks", "ocrs_engine_enable_timing", "ocrs_stage_count",
    "ocrs_stage_name", "ocrs_engine_stage_times", "ocrs_kernel_class_count", "ocrs_kernel_class_name",
    "ocrs_engine_kernel_stats", "ocrs_engine_set_kernel_timing_mask", "ocrs_gru_tile_plan", "ocrs_host_malloc", "ocrs_host_free", "ocrs_engine_prepare_input_batch",
    "ocrs_get_device", "ocrs_model_load_file_on_device", "ocrs_model_load_bytes_on_device", "ocrs_model_device", "ocrs_engine_device",
    "ocrs_engine_group_new", "ocrs_engine_group_free", "ocrs_engine_group_size", "ocrs_engine_group_member", "ocrs_group_deal",
    "ocrs_group_prepare_input_batch", "ocrs_group_prepare_input_device_batch", "ocrs_group_detect_words_batch",
    "ocrs_group_recognize_text_batch", "ocrs_group_gather", "ocrs_group_final_gather", "ocrs_group_worker_threads", "ocrs_group_last_gather", "ocrs_device_malloc_on", "ocrs_engine_coalesce_stats", "ocrs_coalescer_selftest", "ocrs_engine_kernel_mfma_flops", "ocrs_engine_prepare_input_jpeg", "ocrs_jpeg_decode_rgb", "ocrs_jpeg_info", "ocrs_jpeg_coefficients",
    "ocrs_group_member_stats", "ocrs_numa_parse_cpulist", "ocrs_numa_bind_selftest", "ocrs_engine_recognize_logits", "ocrs_engine_set_option", "ocrs_engine_get_option", "ocrs_option_name", "ocrs_device_pool_stats", "ocrs_device_pool_configure", "ocrs_device_pool_trim",
    "ocrs_device_set_isolation", "ocrs_device_isolation", "ocrs_group_set_replay", "ocrs_abi_version",
]

ABI_VERSION = 6   # include/ocrs_amd.h OCRS_ABI_VERSION

_lib = None


def lib():
    """Load the shared object (building it is __graft_entry__.build()'s job)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OcrsError(7, "libocrs_amd.so is not built: run `python -m ocrs_amd.build` "
                               "(there is no CPU fallback for the HIP engine)")
        L = C.CDLL(LIB_PATH)
        L.ocrs_abi_version.restype = C.c_uint32
        if L.ocrs_abi_version() != ABI_VERSION:   # EngineParams / GroupParams below mirror the structs of THIS version
            raise OcrsError(7, "libocrs_amd.so has ABI version %d, this binding was written for %d: rebuild (python -m ocrs_amd.build)"
                            % (L.ocrs_abi_version(), ABI_VERSION))
        L.ocrs_last_error.restype = C.c_char_p
        L.ocrs_engine_detection_threshold.restype = C.c_float
        L.ocrs_engine_detection_threshold.argtypes = [C.c_void_p]
        L.ocrs_stage_name.restype = C.c_char_p
        L.ocrs_kernel_class_name.restype = C.c_char_p
        L.ocrs_buffer_free.argtypes = [C.c_void_p]
        for name in ("ocrs_model_free", "ocrs_engine_free", "ocrs_page_free", "ocrs_engine_group_free"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        _lib = L
    return _lib


def check(status):
    if status != OCRS_OK:
        msg = lib().ocrs_last_error()
        raise OcrsError(status, msg.decode("utf-8", "replace") if msg else STATUS_NAMES.get(status, "error"))


def device_count():
    n = C.c_int(0)
    check(lib().ocrs_device_count(C.byref(n)))
    return n.value


def measure_peaks():
    """(fp32 MFMA TFLOP/s, HBM copy GB/s) this device sustains (micro-benchmarks in kernels_peaks.hip)."""
    a, b = C.c_double(0), C.c_double(0)
    check(lib().ocrs_device_measure_peaks(C.byref(a), C.byref(b)))
    return a.value, b.value


def set_option(name, value):
    """ocrs_set_option: the process DEFAULT of a tuning option — what engines created afterwards start with, and what a
    bare Model.run uses.  An existing engine keeps its own copy: OcrEngine.set_option."""
    check(lib().ocrs_set_option(name.encode(), C.c_long(int(value))))


def option_names():
    out, i = [], 0
    while True:
        n = C.c_char_p()
        check(lib().ocrs_option_name(i, C.byref(n)))
        if not n.value:
            return out
        out.append(n.value.decode())
        i += 1


POOL_FIELDS = ("device_live", "device_cached", "device_cap", "device_peak_live", "device_driver_allocs", "device_driver_frees",
               "pinned_live", "pinned_cached", "pinned_cap", "pinned_peak_live", "pinned_driver_allocs", "pinned_driver_frees")


def pool_stats(device=-1):
    """ocrs_device_pool_stats as a dict (bytes / counts)."""
    v = (C.c_uint64 * 12)()
    check(lib().ocrs_device_pool_stats(int(device), v))
    return dict(zip(POOL_FIELDS, (int(x) for x in v)))


def pool_trim(device=-1):
    check(lib().ocrs_device_pool_trim(int(device)))


def pool_configure(device=-1, device_cached_cap=0, pinned_cached_cap=0):
    check(lib().ocrs_device_pool_configure(int(device), C.c_uint64(int(device_cached_cap)), C.c_uint64(int(pinned_cached_cap))))


ISOLATION = {"auto": 0, "none": 1}


def set_isolation(policy="auto", device=-1):
    """ocrs_device_set_isolation: how kernels of engines with numerics != exact are kept away from other requests' kernels."""
    check(lib().ocrs_device_set_isolation(int(device), int(ISOLATION[policy])))


def isolation(device=-1):
    v = (C.c_int * 3)()
    check(lib().ocrs_device_isolation(int(device), v))
    return {"mode": ("free", "serial")[v[0]], "relaxed_engines": int(v[1]), "cus": int(v[2])}


def ctc_beam_search(logp, width, impl=0):
    """ocrs_ctc_beam_search on a [T, C] float32 matrix -> [(label, pos)]."""
    import numpy as np
    a = np.ascontiguousarray(logp, np.float32)
    t, c = a.shape
    lab, pos, n = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.c_size_t(0)
    check(lib().ocrs_ctc_beam_search(a.ctypes.data_as(C.POINTER(C.c_float)), t, c, C.c_uint32(width), int(impl),
                                     C.byref(lab), C.byref(pos), C.byref(n)))
    out = [(int(lab[i]), int(pos[i])) for i in range(n.value)]
    lib().ocrs_buffer_free(lab)
    lib().ocrs_buffer_free(pos)
    return out


def require_gpu():
    n = device_count()
    if n < 1:
        raise OcrsError(7, "no HIP device visible: the ocrs_amd engine has no CPU fallback")
    return n


def jpeg_coefficients(data):
    """ocrs_jpeg_coefficients (host only): (geom int32[28], quant uint16[256], coef int16[n_blocks, 64])."""
    import numpy as np
    geom = (C.c_int32 * 28)()
    quant = (C.c_uint16 * 256)()
    coef = C.POINTER(C.c_int16)()
    nb = C.c_size_t(0)
    buf = C.create_string_buffer(bytes(data), len(data))
    check(lib().ocrs_jpeg_coefficients(buf, C.c_size_t(len(data)), geom, quant, C.byref(coef), C.byref(nb)))
    a = np.ctypeslib.as_array(coef, shape=(max(nb.value, 1) * 64,))[: nb.value * 64].reshape(-1, 64).copy()
    lib().ocrs_buffer_free(coef)
    return np.array(geom[:], np.int32), np.array(quant[:], np.uint16), a


def jpeg_info(data):
    h, w, n, prog, nz = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_size_t(0)
    buf = C.create_string_buffer(bytes(data), len(data))
    check(lib().ocrs_jpeg_info(buf, C.c_size_t(len(data)), C.byref(h), C.byref(w), C.byref(n), C.byref(prog), C.byref(nz)))
    return {"height": h.value, "width": w.value, "components": n.value, "progressive": bool(prog.value), "nonzero": nz.value}


def jpeg_decode_rgb(data, device=-1):
    """ocrs_jpeg_decode_rgb: host entropy decode + GPU IDCT / upsampling / colour -> (RGB8 [H, W, 3], bytes that crossed PCIe)."""
    import numpy as np
    rgb = C.POINTER(C.c_uint8)()
    h, w, cb = C.c_int(0), C.c_int(0), C.c_size_t(0)
    buf = C.create_string_buffer(bytes(data), len(data))
    check(lib().ocrs_jpeg_decode_rgb(C.c_int(device), buf, C.c_size_t(len(data)), C.byref(rgb), C.byref(h), C.byref(w), C.byref(cb)))
    a = np.ctypeslib.as_array(rgb, shape=(h.value * w.value * 3,)).reshape(h.value, w.value, 3).copy()
    lib().ocrs_buffer_free(rgb)
    return a, cb.value
