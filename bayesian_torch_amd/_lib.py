"""ctypes binding of libbtx.so (include/btx.h).  The HIP library is THE product path: there is no fallback —
`lib()` raises if it is missing, and every non-zero return code raises `BtxError`."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KIND_REPARAM, KIND_FLIPOUT = 0, 1
ACT_F32, ACT_BF16 = 0, 1
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
PREC_CODE = {"f32": 0, "bf16": 1, "bf16x3": 2}
FLAG_TRANSPOSED, FLAG_KL_ACCUM, FLAG_ROWFUSE, FLAG_OUT_F32, FLAG_OUT_BF16, FLAG_GATHER, FLAG_SWAP_SIGNS, FLAG_CONCURRENT = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_REVERSE = 256
E_UNSUPPORTED = -3
STREAM_EPS_W, STREAM_EPS_B, STREAM_SIGN_IN, STREAM_SIGN_OUT = 0, 1, 2, 3
ABI_VERSION = 8
FLAG_LANES_SHIFT = 16
SAMPLE_SKIP_MU = 1


class BtxError(RuntimeError):
    pass


class Geom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "NB", "D", "H", "W", "C", "N", "KD", "KH", "KW", "sd", "sh", "sw", "pd", "ph", "pw",
        "dd", "dh", "dw", "od", "oh", "ow", "groups")]


class Rng(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("sample_idx", ctypes.c_uint32), ("layer_id", ctypes.c_uint32),
                ("sample_idx_dev", ctypes.c_void_p)]


class BnFuse(ctypes.Structure):
    _fields_ = [("residual", ctypes.c_void_p), ("relu", ctypes.c_int32), ("mask", ctypes.c_void_p), ("dres", ctypes.c_void_p)]


class Epilogue(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("relu", ctypes.c_int32), ("pool", ctypes.c_int32)]


class Lanes(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("x_stride", ctypes.c_int64), ("out_stride", ctypes.c_int64),
                ("res_stride", ctypes.c_int64)]


class Noise(ctypes.Structure):
    _fields_ = [("eps_w", ctypes.c_void_p), ("eps_b", ctypes.c_void_p),
                ("sign_in", ctypes.c_void_p), ("sign_out", ctypes.c_void_p), ("sampled_w", ctypes.c_void_p)]


class SampleItem(ctypes.Structure):
    _fields_ = [("geom", ctypes.POINTER(Geom)), ("mu_w", ctypes.c_void_p), ("rho_w", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("kind", ctypes.c_int32), ("layer_id", ctypes.c_uint32),
                ("src_KW", ctypes.c_int32), ("src_C", ctypes.c_int32)]


class KlItem(ctypes.Structure):
    _fields_ = [("mu", ctypes.c_void_p), ("rho", ctypes.c_void_p), ("prior_mu_t", ctypes.c_void_p),
                ("prior_sigma_t", ctypes.c_void_p), ("dmu", ctypes.c_void_p), ("drho", ctypes.c_void_p),
                ("prior_mu", ctypes.c_float), ("prior_sigma", ctypes.c_float), ("n", ctypes.c_size_t)]


EXPORTS = ("btx_abi_version", "btx_strerror", "btx_kl_workspace_bytes", "btx_kl_gauss", "btx_kl_model_workspace_bytes",
           "btx_kl_gauss_model", "btx_kl_gauss_model_bwd", "btx_contract_wgrad",
           "btx_contract_workspace_bytes", "btx_contract_fwd", "btx_contract_fwd_ex", "btx_contract_fwd_lanes", "btx_contract_pool_shape", "btx_out_shape", "btx_fill_eps", "btx_fill_sign", "btx_rho_grad",
           "btx_mc_packed_floats", "btx_mc_accumulate", "btx_mc_accumulate_lanes", "btx_sampled_w_bytes", "btx_sample_weights", "btx_sampled_w_bytes_lanes", "btx_sample_weights_lanes", "btx_rowfuse_pack", "btx_maxpool2d_cl", "btx_avgpool_global_cl",
           "btx_bn_workspace_bytes", "btx_bn_train_fwd", "btx_bn_train_bwd", "btx_dgrad_weights",
           "btx_wgrad_workspace_bytes", "btx_contract_wgrad_ws", "btx_maxpool2d_cl_train", "btx_maxpool2d_cl_bwd")


def lib_path():
    # BTX_LIB: alternative build of the same ABI (A/B measurements of kernel variants)
    return os.environ.get("BTX_LIB") or os.path.join(_HERE, "libbtx.so")


def lib():
    """Load libbtx.so once.  Raises BtxError (loudly) when the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise BtxError("libbtx.so not found at %s — build it with `python __graft_entry__.py build` "
                       "(hipcc --offload-arch=gfx950); there is no fallback path." % path)
    L = ctypes.CDLL(path)
    vp, sz, u32, i32, f32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.c_float
    L.btx_abi_version.restype = i32
    L.btx_abi_version.argtypes = []
    L.btx_strerror.restype = ctypes.c_char_p
    L.btx_strerror.argtypes = [i32]
    L.btx_kl_workspace_bytes.restype = sz
    L.btx_kl_workspace_bytes.argtypes = [sz]
    L.btx_kl_gauss.restype = i32
    L.btx_kl_gauss.argtypes = [vp, vp, sz, vp, vp, f32, f32, vp, u32, vp, sz, vp]
    L.btx_kl_model_workspace_bytes.restype = sz
    L.btx_kl_model_workspace_bytes.argtypes = [i32]
    L.btx_kl_gauss_model.restype = i32
    L.btx_kl_gauss_model.argtypes = [ctypes.POINTER(KlItem), i32, vp, vp, sz, vp]
    L.btx_kl_gauss_model_bwd.restype = i32
    L.btx_kl_gauss_model_bwd.argtypes = [ctypes.POINTER(KlItem), i32, vp, vp]
    L.btx_contract_wgrad.restype = i32
    L.btx_contract_wgrad.argtypes = [i32, ctypes.POINTER(Geom), vp, vp, vp, vp, vp, vp, ctypes.POINTER(Rng),
                                     ctypes.POINTER(Noise), i32, u32, vp]
    L.btx_contract_wgrad_ws.restype = i32
    L.btx_contract_wgrad_ws.argtypes = [i32, ctypes.POINTER(Geom), vp, vp, vp, vp, vp, vp, ctypes.POINTER(Rng),
                                        ctypes.POINTER(Noise), i32, u32, vp, sz, vp, vp, vp]
    L.btx_wgrad_workspace_bytes.restype = sz
    L.btx_wgrad_workspace_bytes.argtypes = [i32, ctypes.POINTER(Geom), i32, u32]
    L.btx_contract_workspace_bytes.restype = sz
    L.btx_contract_workspace_bytes.argtypes = [ctypes.POINTER(Geom), i32, i32, i32, u32]
    L.btx_contract_fwd.restype = i32
    L.btx_contract_fwd.argtypes = [i32, ctypes.POINTER(Geom), vp, vp, vp, vp, vp, vp, ctypes.POINTER(Rng),
                                   ctypes.POINTER(Noise), i32, i32, u32, vp, sz, vp]
    L.btx_contract_fwd_ex.restype = i32
    L.btx_contract_fwd_ex.argtypes = L.btx_contract_fwd.argtypes + [ctypes.POINTER(Epilogue)]
    L.btx_contract_fwd_lanes.restype = i32
    L.btx_contract_fwd_lanes.argtypes = L.btx_contract_fwd_ex.argtypes + [ctypes.POINTER(Lanes)]
    L.btx_contract_pool_shape.restype = i32
    L.btx_contract_pool_shape.argtypes = [ctypes.POINTER(Geom), i32, i32, u32] + [ctypes.POINTER(ctypes.c_int32)] * 2
    L.btx_out_shape.restype = i32
    L.btx_out_shape.argtypes = [ctypes.POINTER(Geom), u32] + [ctypes.POINTER(ctypes.c_int32)] * 3
    L.btx_fill_eps.restype = i32
    L.btx_fill_eps.argtypes = [vp, sz, ctypes.POINTER(Rng), u32, vp]
    L.btx_rho_grad.restype = i32
    L.btx_rho_grad.argtypes = [vp, vp, vp, sz, ctypes.POINTER(Rng), u32, vp]
    L.btx_fill_sign.restype = i32
    L.btx_fill_sign.argtypes = [vp, sz, ctypes.POINTER(Rng), u32, vp]
    L.btx_mc_packed_floats.restype = sz
    L.btx_mc_packed_floats.argtypes = [i32, i32]
    L.btx_sampled_w_bytes.restype = sz
    L.btx_sampled_w_bytes.argtypes = [ctypes.POINTER(Geom), i32, i32]
    L.btx_sample_weights.restype = i32
    L.btx_sample_weights.argtypes = [ctypes.POINTER(SampleItem), i32, ctypes.POINTER(Rng), i32, vp]
    L.btx_sampled_w_bytes_lanes.restype = sz
    L.btx_sampled_w_bytes_lanes.argtypes = [ctypes.POINTER(Geom), i32, i32, i32]
    L.btx_sample_weights_lanes.restype = i32
    L.btx_sample_weights_lanes.argtypes = [ctypes.POINTER(SampleItem), i32, ctypes.POINTER(Rng), i32, vp, i32, u32]
    L.btx_rowfuse_pack.restype = i32
    L.btx_rowfuse_pack.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int64), i32, i32, i32, i32, vp, i32, i32, i32, i32, i32,
                                   i32, vp]
    L.btx_maxpool2d_cl.restype = i32
    L.btx_maxpool2d_cl.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.btx_maxpool2d_cl_train.restype = i32
    L.btx_maxpool2d_cl_train.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.btx_maxpool2d_cl_bwd.restype = i32
    L.btx_maxpool2d_cl_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.btx_avgpool_global_cl.restype = i32
    L.btx_avgpool_global_cl.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.btx_mc_accumulate.restype = i32
    L.btx_mc_accumulate.argtypes = [vp, i32, i32, i32, f32, vp, vp]
    L.btx_mc_accumulate_lanes.restype = i32
    L.btx_mc_accumulate_lanes.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp]
    i64 = ctypes.c_longlong
    L.btx_bn_workspace_bytes.restype = sz
    L.btx_bn_workspace_bytes.argtypes = [i64, i32]
    L.btx_bn_train_fwd.restype = i32
    L.btx_bn_train_fwd.argtypes = [vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, ctypes.POINTER(BnFuse), vp, sz, vp]
    L.btx_bn_train_bwd.restype = i32
    L.btx_bn_train_bwd.argtypes = [vp, vp, vp, i32, i64, i32, vp, i32, vp, vp, vp, vp, ctypes.POINTER(BnFuse), vp, sz, vp]
    L.btx_dgrad_weights.restype = i32
    L.btx_dgrad_weights.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, ctypes.POINTER(Rng), vp]
    if L.btx_abi_version() != ABI_VERSION:
        raise BtxError("libbtx.so ABI %d != expected %d" % (L.btx_abi_version(), ABI_VERSION))
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise BtxError("libbtx: %s (code %d)" % (lib().btx_strerror(rc).decode(), rc))
