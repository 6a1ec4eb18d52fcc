"""ctypes binding of libsiammask_hip.so (the C ABI in include/siammask_hip.h).

``import torch`` happens before ``ctypes.CDLL`` on purpose: torch bundles a HIP runtime with
the same SONAME (libamdhip64.so.7) as /opt/rocm's, and loading torch first makes the
dynamic linker resolve the library's dependency to the runtime torch already uses, so that
streams and device pointers are shared.

The product path has no CPU fallback: if the shared library is missing, loading raises.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMK_LIB=<path>: load another build of the library (measurement aid: A/B arms of compile-time choices, tools/measure/build_variant.sh)
LIB_PATH = os.environ.get("SMK_LIB") or os.path.join(_HERE, "libsiammask_hip.so")
CSRC = os.path.join(_HERE, "csrc")

DTYPE = {"f32": 0, "fp32": 0, "float32": 0, "f16": 1, "fp16": 1, "float16": 1, "half": 1,
         "f16x3": 2}       # split-operand fp16 (SMK_DTYPE_F16X3): fp32-grade cls / loc / box on the fp16 matrix pipe
VARIANT = {"rpn": 0, "base": 1, "sharp": 2}
TRACK_BOX, TRACK_MASK, TRACK_NO_MASK_HEAD = 0, 1, 2

# every symbol include/siammask_hip.h declares
SYMBOLS = (
    "smk_version", "smk_last_error", "smk_create", "smk_destroy", "smk_set_weight",
    "smk_finalize_weights", "smk_template", "smk_track", "smk_refine", "smk_set_decode_params", "smk_decode", "smk_step", "smk_set_graph_mode", "smk_seq_status", "smk_seq_sync_check", "smk_set_result_ring", "smk_result_ring_cursor", "smk_set_pipeline", "smk_pipeline_join", "smk_pipeline_observe",
    "smk_debug_read", "smk_debug_seq_inject", "smk_tune", "smk_tune_get", "smk_profile", "smk_profile_dump", "smk_op_conv2d_ex", "smk_op_conv2d", "smk_op_dw_xcorr",
    "smk_op_maxpool3x3s2", "smk_op_conv_seq", "smk_host_conv2d_ex", "smk_host_plan_conv", "smk_bench_conv", "smk_packed_size", "smk_export_packed",
    "smk_import_packed", "smk_crop_resize", "smk_paste_mask", "smk_paste_labels",
)


class ConvGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "B", "Cin", "H", "W", "Cout", "k", "stride", "pad", "dil", "relu", "res_mode",
        "win", "ups", "Hl", "Wl", "org_y", "org_x", "pos_mul", "pos_add", "cin_off", "cin_len")]


class SeqOp(ctypes.Structure):
    """smk_seq_op: one layer of an smk_op_conv_seq sequence"""
    _fields_ = [("g", ConvGeom), ("src", ctypes.c_int), ("res_src", ctypes.c_int), ("sync", ctypes.c_int),
                ("cfg", ctypes.c_int), ("kstag", ctypes.c_int), ("w_host", ctypes.c_void_p), ("b_host", ctypes.c_void_p),
                ("y_dev", ctypes.c_void_p)]


class SmkError(RuntimeError):
    code = 0      # the library's SMK_E_* return code (0: raised on the Python side)


E_SEQ = -6        # SMK_E_SEQ: the persistent sequence kernel reported a failure; the frame is to be re-submitted


def build_library(force=False, verbose=False):
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    out = subprocess.run(["make", "-C", CSRC, "-j", "4"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, universal_newlines=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise SmkError("building libsiammask_hip.so failed")
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmkError(
            "%s is missing: the HIP extension has not been built (run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `make -C siammask_amd/csrc`). There is no "
            "CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    L.smk_version.restype = ci
    L.smk_last_error.restype = ctypes.c_char_p
    L.smk_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci]
    L.smk_destroy.argtypes = [vp]
    L.smk_set_weight.argtypes = [vp, ctypes.c_char_p, fp, ctypes.POINTER(ctypes.c_int64), ci]
    L.smk_finalize_weights.argtypes = [vp]
    L.smk_template.argtypes = [vp, fp, ci, vp]
    L.smk_track.argtypes = [vp, fp, ci, ci, fp, fp, fp, vp]
    L.smk_refine.argtypes = [vp, vp, ci, ci, fp, vp]
    L.smk_set_decode_params.argtypes = [vp, fp, ci, ci, ctypes.c_double, ctypes.c_double]
    L.smk_decode.argtypes = [vp, fp, fp, ci, fp, vp, fp, vp]
    L.smk_step.argtypes = [vp, fp, ci, ci, fp, fp, fp, fp, fp, fp, vp]
    L.smk_set_graph_mode.argtypes = [vp, ci]
    L.smk_debug_seq_inject.argtypes = [vp, ci]
    L.smk_set_result_ring.argtypes = [vp, vp, vp, ci, ci]
    L.smk_set_pipeline.argtypes = [vp, ci]
    L.smk_pipeline_join.argtypes = [vp, vp]
    L.smk_pipeline_observe.argtypes = [vp, vp]
    L.smk_result_ring_cursor.argtypes = [vp, ctypes.POINTER(ci), ci, vp]
    L.smk_seq_sync_check.argtypes = [vp, vp, ctypes.POINTER(ci)]
    L.smk_tune.argtypes = [ctypes.c_char_p, ci]
    L.smk_tune_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci)]
    L.smk_profile.argtypes = [vp, ci]
    L.smk_profile_dump.argtypes = [vp, ctypes.c_char_p, ci]
    ip = ctypes.POINTER(ci)
    L.smk_debug_read.argtypes = [vp, ctypes.c_char_p, fp, ip, ip, ip, vp]
    gp = ctypes.POINTER(ConvGeom)
    L.smk_op_conv2d_ex.argtypes = [ci, ci, gp, fp, fp, fp, fp, vp, fp, vp]
    L.smk_op_conv2d.argtypes = [ci, ci, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, ci, ci, fp, fp, vp]
    L.smk_op_dw_xcorr.argtypes = [ci, fp, fp, ci, ci, ci, ci, ci, ci, fp, vp]
    L.smk_op_maxpool3x3s2.argtypes = [ci, fp, ci, ci, ci, ci, fp, vp]
    L.smk_host_conv2d_ex.argtypes = [gp, fp, fp, fp, fp, vp, fp]
    L.smk_host_plan_conv.argtypes = [gp, ci, ci, ip, ip, ip, ip]
    L.smk_packed_size.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.smk_export_packed.argtypes = [vp, vp, ctypes.c_uint64]
    L.smk_import_packed.argtypes = [vp, vp, ctypes.c_uint64]
    L.smk_crop_resize.argtypes = [vp, ctypes.c_int64, ci, ci, vp, vp, ci, ci, fp, vp]
    L.smk_paste_mask.argtypes = [fp, ci, vp, ci, ci, ci, ctypes.c_float, ctypes.c_float, vp, fp, vp]
    L.smk_paste_labels.argtypes = [fp, ci, vp, ci, ci, ci, ctypes.c_float, ctypes.c_float, vp, vp]
    L.smk_op_conv_seq.argtypes = [ctypes.POINTER(SeqOp), ci, fp, ci, ctypes.POINTER(ctypes.c_float), fp, ip, vp]
    L.smk_bench_conv.argtypes = [ci, ci, gp, ci, ci, ctypes.POINTER(ctypes.c_float), vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("smk_last_error",):
            fn.restype = ci
    _lib = L
    # SMK_TUNE="key=value,key=value": library tuning knobs from the environment (A/B runs of the test-suite)
    for kv in filter(None, os.environ.get("SMK_TUNE", "").split(",")):
        k, v = kv.split("=")
        check(L.smk_tune(k.strip().encode(), int(v)))
    return L


def check(rc):
    if rc != 0:
        msg = lib().smk_last_error()
        e = SmkError("libsiammask_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
        e.code = rc
        raise e


def current_stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def tune(**kw):
    """Set process-wide tuning knobs of the library (A/B measurements)."""
    for k, v in kw.items():
        check(lib().smk_tune(k.encode(), int(v)))


def tune_get(key):
    """Current value of a tuning knob (so that a test can put back what it changed)."""
    v = ctypes.c_int(0)
    check(lib().smk_tune_get(key.encode(), ctypes.byref(v)))
    return v.value
