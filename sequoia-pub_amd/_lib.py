"""ctypes binding of libsequoia_hip.so (the C ABI in include/sequoia_hip.h).

PyTorch is imported first on purpose: the library's NEEDED ``libamdhip64.so.7`` then
resolves to the HIP runtime PyTorch-ROCm already loaded, so device pointers and
streams are shared.  There is no fallback: a missing library or GPU raises.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SQ_HIP_LIB") or os.path.join(_HERE, "libsequoia_hip.so")      # SQ_HIP_LIB: another build of the library (same-box A/B runs)

SQ_F32 = 0
SQ_BF16 = 1
SQ_BF16X3 = 2       # split bf16 (hi + lo planes, three MFMAs per product): ResNet-50 embedder only
SQ_F16X3 = 3        # the same with fp16 planes (22 significant bits, range 65504): the fast parity mode
SQ_MAX_DEPTH = 16
HEAD_DIM = 64

DTYPES = {"fp32": SQ_F32, "f32": SQ_F32, "float32": SQ_F32, SQ_F32: SQ_F32,
          "bf16": SQ_BF16, "bfloat16": SQ_BF16, SQ_BF16: SQ_BF16,
          "bf16x3": SQ_BF16X3, "split-bf16": SQ_BF16X3, SQ_BF16X3: SQ_BF16X3,
          "f16x3": SQ_F16X3, "fp16x3": SQ_F16X3, "split-fp16": SQ_F16X3, SQ_F16X3: SQ_F16X3}


class VisConfig(ctypes.Structure):
    _fields_ = [("input_dim", ctypes.c_int32), ("depth", ctypes.c_int32), ("nheads", ctypes.c_int32),
                ("num_outputs", ctypes.c_int32), ("num_clusters", ctypes.c_int32)]


_LAYER_FIELDS = ["f_w", "f_b", "s_w", "s_b", "lnf_g", "lnf_b", "lns_g", "lns_b", "c_w", "c_b",
                 "proj_w", "proj_b", "ffln_g", "ffln_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b"]


class VisLayerOffsets(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in _LAYER_FIELDS]


class VisLayout(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_int64), ("head_ln_g", ctypes.c_int64), ("head_ln_b", ctypes.c_int64),
                ("head_w", ctypes.c_int64), ("head_b", ctypes.c_int64), ("total", ctypes.c_int64),
                ("layer", VisLayerOffsets * SQ_MAX_DEPTH)]


class VitConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("dim", "depth", "heads", "mlp_dim", "num_outputs", "num_clusters")]


_VIT_LAYER_FIELDS = ["ln1_g", "ln1_b", "qkv_w", "out_w", "ln2_g", "ln2_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b"]


class VitLayerOffsets(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in _VIT_LAYER_FIELDS]


class VitLayout(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_int64), ("head_ln_g", ctypes.c_int64), ("head_ln_b", ctypes.c_int64),
                ("head_w", ctypes.c_int64), ("head_b", ctypes.c_int64), ("total", ctypes.c_int64),
                ("layer", VitLayerOffsets * SQ_MAX_DEPTH)]


class ConvDesc(ctypes.Structure):
    _fields_ = [("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("k", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("k_padded", ctypes.c_int32)]


class ResNet50Layout(ctypes.Structure):
    _fields_ = [("conv", ConvDesc * 53), ("w_total", ctypes.c_int64), ("b_total", ctypes.c_int64)]


class SequoiaHipError(RuntimeError):
    pass


_lib = None


def _declare(lib):
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.sq_last_error.restype = ctypes.c_char_p
    lib.sq_last_error.argtypes = []
    lib.sq_version.restype = i32
    lib.sq_device_ok.restype = i32
    lib.sq_vis_layout_init.restype = i32
    lib.sq_vis_layout_init.argtypes = [ctypes.POINTER(VisConfig), ctypes.POINTER(VisLayout)]
    lib.sq_vis_workspace_bytes.restype = sz
    lib.sq_vis_workspace_bytes.argtypes = [ctypes.POINTER(VisConfig), i32, i32, i32]
    lib.sq_vis_forward.restype = i32
    lib.sq_vis_forward.argtypes = [ctypes.POINTER(VisConfig), i32, vp, vp, vp, vp, i32, i32, vp, sz, vp]
    lib.sq_vis_forward_ex.restype = i32
    lib.sq_vis_forward_ex.argtypes = [ctypes.POINTER(VisConfig), i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, sz, vp]
    lib.sq_vis_forward_tiles.restype = i32
    lib.sq_vis_forward_tiles.argtypes = [ctypes.POINTER(VisConfig), i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, sz, vp]
    lib.sq_vis_backward_workspace_bytes.restype = sz
    lib.sq_vis_backward_workspace_bytes.argtypes = [ctypes.POINTER(VisConfig), i32, i32]
    lib.sq_vis_backward.restype = i32
    lib.sq_vis_backward.argtypes = [ctypes.POINTER(VisConfig), i32, vp, vp, vp, vp, vp, i32, vp, sz, vp, sz, vp]
    lib.sq_gene_eval_workspace_bytes.restype = sz
    lib.sq_gene_eval_workspace_bytes.argtypes = [i32, i32]
    lib.sq_gene_eval_stats.restype = i32
    lib.sq_gene_eval_stats.argtypes = [vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.sq_window_vote.restype = i32
    lib.sq_window_vote.argtypes = [vp, i32, i32, vp, i32, i32, i32, ctypes.c_float, vp, vp]
    lib.sq_vis_backward_buckets.restype = i32
    lib.sq_vis_backward_buckets.argtypes = [ctypes.POINTER(VisConfig), i32, vp, vp, vp, vp, vp, i32, vp, sz, vp, sz, vp,
                                            ctypes.POINTER(vp), i32]
    lib.sq_vis_grad_buckets.restype = i32
    lib.sq_vis_grad_buckets.argtypes = [ctypes.POINTER(VisConfig), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), i32]
    lib.sq_train_scratch_bytes.restype = sz
    lib.sq_train_scratch_bytes.argtypes = [i32]
    f32 = ctypes.c_float
    lib.sq_mse_loss_grad.restype = i32
    lib.sq_mse_loss_grad.argtypes = [vp, vp, sz, f32, vp, vp, vp, vp]
    lib.sq_adamw_step.restype = i32
    lib.sq_adamw_step.argtypes = [vp, vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, f32, vp]
    lib.sq_batch_metrics.restype = i32
    lib.sq_batch_metrics.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.sq_kmeans_workspace_bytes.restype = sz
    lib.sq_kmeans_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.sq_kmeans_fit.restype = i32
    lib.sq_kmeans_fit.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32, i32, ctypes.c_double, vp, vp, vp, vp, vp, sz, vp]
    lib.sq_resnet50_layout_init.restype = i32
    lib.sq_resnet50_layout_init.argtypes = [ctypes.POINTER(ResNet50Layout)]
    lib.sq_resnet50_workspace_bytes.restype = sz
    lib.sq_resnet50_workspace_bytes.argtypes = [i32, i32, i32]
    lib.sq_resnet50_extract.restype = i32
    lib.sq_resnet50_extract.argtypes = [i32, vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.sq_resnet50_extract_checked.restype = i32
    lib.sq_resnet50_extract_checked.argtypes = [i32, vp, vp, vp, vp, i32, i32, vp, vp, sz, vp, vp]
    lib.sq_vit_layout_init.restype = i32
    lib.sq_vit_layout_init.argtypes = [ctypes.POINTER(VitConfig), ctypes.POINTER(VitLayout)]
    lib.sq_vit_workspace_bytes.restype = sz
    lib.sq_vit_workspace_bytes.argtypes = [ctypes.POINTER(VitConfig), i32, i32, i32]
    lib.sq_vit_forward.restype = i32
    lib.sq_vit_forward.argtypes = [ctypes.POINTER(VitConfig), i32, vp, vp, vp, vp, i32, i32, vp, sz, vp]
    lib.sq_vit_backward_workspace_bytes.restype = sz
    lib.sq_vit_backward_workspace_bytes.argtypes = [ctypes.POINTER(VitConfig), i32, i32]
    lib.sq_vit_backward.restype = i32
    lib.sq_vit_backward.argtypes = [ctypes.POINTER(VitConfig), i32, vp, vp, vp, vp, vp, i32, vp, sz, vp, sz, vp]
    lib.sq_prof_enable.restype = i32
    lib.sq_prof_enable.argtypes = [i32]
    lib.sq_prof_report.restype = i32
    lib.sq_prof_report.argtypes = [ctypes.c_char_p, sz]
    lib.sq_prof_marker_names.restype = i32
    lib.sq_prof_marker_names.argtypes = [ctypes.c_char_p, sz]
    lib.sq_linear.restype = i32
    lib.sq_linear.argtypes = [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.sq_he2rna_tile_mask.restype = i32
    lib.sq_he2rna_tile_mask.argtypes = [vp, i32, i32, vp, vp]
    lib.sq_he2rna_topk_mean.restype = i32
    lib.sq_he2rna_topk_mean.argtypes = [vp, i32, vp, vp, i32, f32, vp, i32, i32, i32, vp]
    lib.sq_he2rna_topk_mean_bwd.restype = i32
    lib.sq_he2rna_topk_mean_bwd.argtypes = [vp, i32, vp, vp, i32, f32, vp, vp, i32, i32, i32, i32, vp]
    lib.sq_linear_weight_grad.restype = i32
    lib.sq_linear_weight_grad.argtypes = [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, sz, vp]
    lib.sq_linear_weight_grad_group.restype = i32
    lib.sq_linear_weight_grad_group.argtypes = [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.sq_linear_x3.restype = i32
    lib.sq_linear_x3.argtypes = [i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.sq_cast_f32_to_bf16.restype = i32
    lib.sq_cast_f32_to_bf16.argtypes = [vp, vp, sz, vp]
    lib.sq_cast_bf16_to_f32.restype = i32
    lib.sq_cast_bf16_to_f32.argtypes = [vp, vp, sz, vp]
    for name, (res, args) in _OPTIONAL.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args


_OPTIONAL = {}


def register_signature(name, restype, argtypes):
    """Let sibling modules declare the entry points they bind (kept next to their use)."""
    _OPTIONAL[name] = (restype, argtypes)
    if _lib is not None and hasattr(_lib, name):
        fn = getattr(_lib, name)
        fn.restype = restype
        fn.argtypes = argtypes


def lib():
    """Load (once) and return the shared library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SequoiaHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or sequoia-pub_amd/csrc/build.sh).  sequoia-pub_amd has no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(rc):
    if rc != 0:
        raise SequoiaHipError(f"libsequoia_hip error {rc}: {lib().sq_last_error().decode()}")


def require_gpu(device=None):
    """The product path runs on MI355X only: fail loudly otherwise."""
    if not torch.cuda.is_available():
        raise SequoiaHipError("no ROCm GPU visible to PyTorch: the HIP path cannot run (and there is no CPU fallback)")
    if not lib().sq_device_ok():
        raise SequoiaHipError("the visible GPU is not gfx950 (MI355X); libsequoia_hip is built for gfx950 only")


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def prof_enable(on, markers=False):
    """HIP-event timing of the library's launches on / off; markers=True: sq_prof_enable(2), marker launches around every one."""
    check(lib().sq_prof_enable(2 if (on and markers) else int(bool(on))))


def prof_marker_names():
    import json
    buf = ctypes.create_string_buffer(1 << 18)
    check(lib().sq_prof_marker_names(buf, len(buf)))
    return json.loads(buf.value.decode())


def prof_report():
    """Per-kernel HIP-event timings recorded since prof_enable(True) (synchronises first)."""
    import json
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 20)
    check(lib().sq_prof_report(buf, len(buf)))
    return json.loads(buf.value.decode())
