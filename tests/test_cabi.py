"""The C-ABI face: libbtx.so loads without a GPU and exports exactly what include/btx.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "btx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(btx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from bayesian_torch_amd import _lib
    names = _declared()
    assert len(names) >= 11
    assert set(names) == set(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.lib_path())
    for n in names:
        assert hasattr(L, n), n


def test_abi_version_and_argument_errors_without_gpu():
    """pure host-side calls: version, strerror, shape arithmetic, argument validation (no kernel is launched)"""
    from bayesian_torch_amd import _lib
    L = _lib.lib()
    assert L.btx_abi_version() == 8
    assert b"NULL" in L.btx_strerror(-1)
    g = _lib.Geom()
    g.NB, g.D, g.H, g.W, g.C, g.N = 64, 1, 56, 56, 64, 128
    g.KD, g.KH, g.KW = 1, 3, 3
    g.sd, g.sh, g.sw = 1, 2, 2
    g.pd, g.ph, g.pw = 0, 1, 1
    g.dd = g.dh = g.dw = 1
    g.groups = 1
    d, h, w = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert L.btx_out_shape(ctypes.byref(g), 0, ctypes.byref(d), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (d.value, h.value, w.value) == (1, 28, 28)
    assert L.btx_out_shape(ctypes.byref(g), 1, ctypes.byref(d), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (h.value, w.value) == (111, 111)
    g.groups = 3
    assert L.btx_out_shape(ctypes.byref(g), 0, ctypes.byref(d), ctypes.byref(h), ctypes.byref(w)) == -2
    g.groups = 1
    assert L.btx_contract_fwd(1, ctypes.byref(g), None, None, None, None, None, None, None, None, 0, 0, 0, None, 0,
                              None) == -1
    assert L.btx_kl_gauss(None, None, 10, None, None, 0.0, 1.0, None, 0, None, 0, None) == -1
    assert L.btx_mc_packed_floats(64, 1000) == 2 * 64 * 1000 + 64 + 2
    # training-mode BatchNorm (ABI 7): workspace arithmetic and argument validation
    assert L.btx_bn_workspace_bytes(802816, 64) == (512 * 2 * 64 + 5 * 64) * 4 and L.btx_bn_workspace_bytes(100, 12) == 0
    assert L.btx_bn_train_fwd(None, None, 1, 100, 64, None, None, None, None, 0, 0.1, 1e-5, None, None, None, None, None, 0, None) == -1
    # workspace = split-K partials (small-M ResNet18 layer4 shape) + the weight tiles sampled once per launch:
    # the big-M layer1 shape needs only the latter, 2 arrays (mu, delta) x 64 channels x K=576 x 2 bytes
    g.H = g.W = 7
    g.C = g.N = 512
    g.sh = g.sw = 1
    assert L.btx_contract_workspace_bytes(ctypes.byref(g), 1, 1, 1, 0) > 2 * 512 * 4608 * 2
    g.H = g.W = 56
    g.C = g.N = 64
    # (+ 4 KiB, 256-byte aligned: the image-group queues of the persistent form of the tap-unrolled 3x3 kernel)
    assert L.btx_contract_workspace_bytes(ctypes.byref(g), 1, 1, 1, 0) == 2 * 64 * 576 * 2 + 4096
    # BTX_PREC_BF16X3 (split-bf16): the f32 mode's kernel family, tiles of [hi | lo] bf16 = 4 bytes per weight
    assert L.btx_contract_workspace_bytes(ctypes.byref(g), 1, 0, 2, 0) == L.btx_contract_workspace_bytes(ctypes.byref(g), 1, 0, 0, 0)
    assert L.btx_sampled_w_bytes(ctypes.byref(g), 1, 2) == L.btx_sampled_w_bytes(ctypes.byref(g), 1, 0) > 0
    assert L.btx_contract_workspace_bytes(ctypes.byref(g), 1, 0, 3, 0) == 0  # unknown precision code
    assert L.btx_mc_accumulate_lanes(None, 2, 4, 10, 0, 0.0, None, None) == -1
    # weight-gradient slabs (ABI 8), ResNet18 layer1 at batch 64: the all-taps kernel splits the 3136 64-pixel steps of its one
    # 64 x 64 tile into 242 chunks of 13 (one workgroup per CU), the tap-per-workgroup kernel into 56 for its 9 taps (504 workgroups: one round); mean + delta
    assert L.btx_wgrad_workspace_bytes(1, ctypes.byref(g), 1, 0) == 242 * 2 * 64 * 576 * 4
    assert L.btx_wgrad_workspace_bytes(1, ctypes.byref(g), 0, 0) == 56 * 2 * 64 * 576 * 4   # f32 activations: no all-taps kernel
    assert L.btx_wgrad_workspace_bytes(0, ctypes.byref(g), 0, 0) == 56 * 64 * 576 * 4
    assert L.btx_contract_wgrad_ws(1, ctypes.byref(g), None, None, None, None, None, None, None, None, 1, 0, None, 0, None, None, None) == -4


def test_argument_errors_of_the_sampling_and_format_entry_points():
    """host-side validation of btx_sample_weights / btx_sampled_w_bytes / btx_rowfuse_pack / btx_maxpool2d_cl /
    btx_avgpool_global_cl: every call below returns before anything is launched"""
    from bayesian_torch_amd import _lib
    L = _lib.lib()
    g = _lib.Geom()
    g.NB, g.D, g.H, g.W, g.C, g.N = 1, 1, 3, 3, 64, 128
    g.KD, g.KH, g.KW = 1, 3, 3
    g.sd = g.sh = g.sw = 1
    g.dd = g.dh = g.dw = 1
    g.groups = 1
    # tile image of a 128 x (3*3*64) weight matrix: 2 n-tiles x 64 channels x K elements.  Flipout: mu tiles + delta tiles
    # (one set per MC sample lane) + the f32 sigma cache BTX_SAMPLE_SKIP_MU reads; Reparameterization: W tiles per lane
    w = 128 * 576
    assert L.btx_sampled_w_bytes(ctypes.byref(g), 1, 1) == 2 * w * 2 + w * 4
    assert L.btx_sampled_w_bytes(ctypes.byref(g), 0, 1) == w * 2
    assert L.btx_sampled_w_bytes(ctypes.byref(g), 1, 0) == 2 * w * 4 + w * 4
    assert L.btx_sampled_w_bytes_lanes(ctypes.byref(g), 1, 1, 4) == (1 + 4) * w * 2 + w * 4
    assert L.btx_sampled_w_bytes_lanes(ctypes.byref(g), 0, 1, 4) == 4 * w * 2
    assert L.btx_sampled_w_bytes_lanes(ctypes.byref(g), 1, 1, 0) == 0
    g.N = 100  # ragged n-tile: padded to 128 channels
    assert L.btx_sampled_w_bytes(ctypes.byref(g), 0, 1) == 128 * 576 * 2
    r = _lib.Rng(1, 2, 3, None)
    assert L.btx_sample_weights(None, 1, ctypes.byref(r), 1, None) == -1
    items = (_lib.SampleItem * 1)()
    assert L.btx_sample_weights(items, 1, None, 1, None) == -1
    assert L.btx_sample_weights(items, 0, ctypes.byref(r), 1, None) == 0
    assert L.btx_sample_weights(items, 1, ctypes.byref(r), 7, None) == -5       # unknown precision
    assert L.btx_sample_weights(items, 1, ctypes.byref(r), 1, None) == -1       # NULL geom / pointers inside the item
    st = (ctypes.c_int64 * 4)(1, 1, 1, 1)
    one = ctypes.c_void_p(16)
    assert L.btx_rowfuse_pack(None, 1, st, 1, 3, 8, 8, one, 1, 8, 8, 4, 0, 0, None) == -1
    assert L.btx_rowfuse_pack(one, 1, st, 1, 3, 8, 8, one, 1, 7, 8, 4, 0, 0, None) == -2   # Hp < H
    assert L.btx_rowfuse_pack(one, 1, st, 1, 5, 8, 8, one, 1, 8, 8, 4, 0, 0, None) == -3   # C > cp
    assert L.btx_rowfuse_pack(one, 9, st, 1, 3, 8, 8, one, 1, 8, 8, 4, 0, 0, None) == -5   # dtype
    assert L.btx_maxpool2d_cl(one, one, 1, 1, 8, 8, 12, 3, 2, 1, None) == -3               # C % 8
    assert L.btx_maxpool2d_cl(one, one, 1, 1, 8, 8, 16, 3, 2, 2, None) == -2               # 2*pad > k
    assert L.btx_maxpool2d_cl(ctypes.c_void_p(8), one, 1, 1, 8, 8, 16, 3, 2, 1, None) == -6  # alignment
    assert L.btx_avgpool_global_cl(one, one, 1, 1, 49, 12, None) == -3
    assert L.btx_avgpool_global_cl(None, one, 1, 1, 49, 16, None) == -1


def test_pool_shape_query_without_gpu():
    """btx_contract_pool_shape (host-side plan only): the ResNet stem on its row-fused geometry takes BtxEpilogue.pool,
    f32 activations / non-row-fused / oversized geometries do not"""
    from bayesian_torch_amd import _lib
    L = _lib.lib()
    g = _lib.Geom()
    g.NB, g.D, g.H, g.W, g.C, g.N = 64, 1, 230, 230, 4, 64          # 224^2 + 2*3 padding, 3 -> 4 channels
    g.KD, g.KH, g.KW = 1, 7, 8                                        # 7 -> 8 taps per kernel row
    g.sd, g.sh, g.sw = 1, 2, 2
    g.pd = g.ph = g.pw = 0
    g.dd = g.dh = g.dw = 1
    g.groups = 1
    hq, wq = ctypes.c_int32(), ctypes.c_int32()
    q = lambda act, prec, flags: L.btx_contract_pool_shape(ctypes.byref(g), act, prec, flags, ctypes.byref(hq), ctypes.byref(wq))  # noqa: E731
    assert q(_lib.ACT_BF16, _lib.PREC_BF16, _lib.FLAG_ROWFUSE) == 1 and (hq.value, wq.value) == (56, 56)
    assert q(_lib.ACT_F32, _lib.PREC_F32, _lib.FLAG_ROWFUSE) == 0      # bf16 only
    assert q(_lib.ACT_BF16, _lib.PREC_BF16, 0) == 0                    # the row-fused stem path only
    g.H = g.W = 518                                                   # 256 output columns: two conv rows exceed a half tile
    assert q(_lib.ACT_BF16, _lib.PREC_BF16, _lib.FLAG_ROWFUSE) == 0
    g.H = g.W = 230
    g.N = 96                                                          # whole 64-channel tiles only
    assert q(_lib.ACT_BF16, _lib.PREC_BF16, _lib.FLAG_ROWFUSE) == 0
