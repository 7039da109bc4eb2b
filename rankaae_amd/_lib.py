"""ctypes binding of ``librankaae_hip.so`` (C ABI declared in ``include/rankaae_hip.h``).

There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librankaae_hip.so")

RAAE_MAX_PARTS = 512
ABI_VERSION = 16
IN_NONE, IN_PRELU_BN_DROP, IN_PRELU_DROP = 0, 1, 2
OUT_RAW, OUT_STATS_PRELU, OUT_STATS_RAW, OUT_SOFTPLUS, OUT_RELU = 0, 1, 2, 3, 4
G_DIRECT, G_SOFTPLUS, G_PRELU_BN, G_PRELU, G_RELU = 0, 1, 2, 3, 4


class BnT(C.Structure):
    """``raae_bn_t``"""
    _fields_ = [("partials", C.c_void_p), ("nparts", C.c_int), ("count", C.c_float),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("momentum", C.c_float), ("eps", C.c_float), ("update_running", C.c_int)]


class MaskGenT(C.Structure):
    """``raae_maskgen_t``"""
    _fields_ = [("keys", C.c_void_p), ("offset", C.c_uint), ("thr", C.c_uint), ("inv", C.c_float)]


class DenseFwdT(C.Structure):
    """``raae_dense_fwd_t``"""
    _fields_ = [("x", C.c_void_p), ("B", C.c_int), ("K", C.c_int), ("in_kind", C.c_int), ("slope", C.c_void_p),
                ("has_bn", C.c_int), ("bn", BnT), ("mask", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("N", C.c_int), ("z", C.c_void_p), ("out_kind", C.c_int), ("out_slope", C.c_void_p),
                ("out_partials", C.c_void_p), ("storage", C.c_int), ("mask_scale", C.c_float), ("gen", MaskGenT)]


class DenseBwdT(C.Structure):
    """``raae_dense_bwd_t``"""
    _fields_ = [("g", C.c_void_p), ("g_kind", C.c_int), ("g_partials", C.c_void_p), ("g_nparts", C.c_int),
                ("zout", C.c_void_p), ("out_slope", C.c_void_p), ("has_out_bn", C.c_int), ("out_bn", BnT),
                ("B", C.c_int), ("N", C.c_int), ("x", C.c_void_p), ("K", C.c_int), ("in_kind", C.c_int),
                ("slope", C.c_void_p), ("has_bn", C.c_int), ("bn", BnT), ("mask", C.c_void_p), ("w", C.c_void_p),
                ("dw", C.c_void_p), ("db", C.c_void_p), ("dslope", C.c_void_p), ("slab_stride", C.c_long),
                ("dx", C.c_void_p), ("dx_partials", C.c_void_p), ("storage", C.c_int), ("mask_scale", C.c_float),
                ("gen", MaskGenT)]


ST_X, ST_MASK, ST_Z = 1, 2, 4        # RAAE_ST_*: bf16 storage bits of the dense kernels


class StepBeginT(C.Structure):
    """``raae_step_begin_t``"""
    _fields_ = [("steps", C.c_void_p), ("nsteps", C.c_int), ("step_mask", C.c_uint), ("rng_state", C.c_void_p),
                ("cursor", C.c_void_p), ("stride", C.c_int), ("ticket", C.c_void_p), ("spec", C.c_void_p),
                ("aux", C.c_void_p), ("idx", C.c_void_p), ("B", C.c_int), ("L", C.c_int), ("n_aux", C.c_int),
                ("spec_noise", C.c_float), ("noise_tape", C.c_void_p), ("noise_goff", C.c_long),
                ("spec_out", C.c_void_p), ("aux_out", C.c_void_p), ("tape", C.c_void_p), ("seg_desc", C.c_void_p),
                ("seg_scale", C.c_void_p), ("nseg", C.c_int), ("total", C.c_long)]


class DiscFusedT(C.Structure):
    """``raae_disc_fused_t``"""
    _fields_ = [("z_real", C.c_void_p), ("styles", C.c_void_p), ("noise", C.c_void_p), ("sigma", C.c_float),
                ("mask1", C.c_void_p), ("mask2", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("s1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("s2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p), ("alpha", C.c_void_p),
                ("n_real", C.c_int), ("n_fake", C.c_int), ("ns", C.c_int), ("hidden", C.c_int),
                ("dw1", C.c_void_p), ("db1", C.c_void_p), ("ds1", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p),
                ("ds2", C.c_void_p), ("dw3", C.c_void_p), ("db3", C.c_void_p), ("slab_stride", C.c_long),
                ("dstyles", C.c_void_p), ("partial", C.c_void_p), ("ticket", C.c_void_p), ("loss", C.c_void_p)]


class LossFinT(C.Structure):
    """``raae_loss_fin_t``"""
    _fields_ = [("scale", C.c_float), ("out", C.c_void_p), ("slot", C.c_int), ("acc_slot", C.c_int),
                ("ticket", C.c_void_p)]


class ViewT(C.Structure):
    """``raae_view_t``"""
    _fields_ = [("raw", C.c_void_p), ("slope", C.c_void_p), ("bn", BnT), ("has_bn", C.c_int), ("mask", C.c_void_p)]


class GradT(C.Structure):
    """``raae_grad_t``"""
    _fields_ = [("g", C.c_void_p), ("g_partials", C.c_void_p), ("g_nparts", C.c_int), ("u", C.c_void_p),
                ("bn", BnT), ("has_bn", C.c_int), ("raw", C.c_void_p), ("slope", C.c_void_p), ("act", C.c_int)]


class ConvT(C.Structure):
    """``raae_conv_t``"""
    _fields_ = [(n, C.c_int) for n in ("Cin", "Lin", "Cout", "Lout", "K", "stride", "pad", "pad_replicate",
                                       "groups", "transposed")]


class BlockFwdAT(C.Structure):
    """``raae_block_fwd_a_t``"""
    _fields_ = ([("inp", ViewT), ("mask", C.c_void_p), ("B", C.c_int)] +
                [(n, C.c_int) for n in ("Cin", "Cout", "Lin", "L1", "Lout", "E")] +
                [("cv1", ConvT), ("cvs", ConvT), ("has_short", C.c_int)] +
                [(n, C.c_void_p) for n in ("w1", "b1", "slope1", "ws", "bs", "wf1", "bf1", "se1", "wf2", "bf2", "se2",
                                           "T1", "Sh", "E1", "E2", "pT1", "pE2")] +
                [(n, C.c_int) for n in ("S", "ngroups", "sh_lin", "sh_l1", "sh_lout", "sh_e", "halo")])


class BlockFwdBT(C.Structure):
    """``raae_block_fwd_b_t``"""
    _fields_ = ([("vT1", ViewT), ("vE2", ViewT), ("vR", ViewT)] +
                [(n, C.c_int) for n in ("B", "Cin", "Cout", "L1", "Lout")] +
                [("cv2", ConvT), ("cve", ConvT), ("has_short", C.c_int), ("has_excit", C.c_int)] +
                [(n, C.c_void_p) for n in ("w2", "b2", "slope2", "we", "be", "se3", "Sh", "ss", "T2", "E3", "Y", "pY")] +
                [(n, C.c_int) for n in ("S", "ngroups", "sh_l1", "sh_lout", "halo2")])


class BlockBwdBT(C.Structure):
    """``raae_block_bwd_b_t``"""
    _fields_ = ([("gy", GradT), ("vT1", ViewT), ("vE2", ViewT)] +
                [(n, C.c_int) for n in ("B", "Cin", "Cout", "L1", "Lout")] +
                [("cv2", ConvT), ("cve", ConvT), ("has_short", C.c_int), ("has_excit", C.c_int)] +
                [(n, C.c_void_p) for n in ("w2", "we", "slope2", "ss", "se", "T2", "Sh", "Ex", "dT2", "dSh", "dEx",
                                           "dBn2", "dBnE", "pdBn2", "pdBnE", "dslope2", "dslope_s", "dslope_e")] +
                [("slab_stride", C.c_long)] + [(n, C.c_int) for n in ("S", "ngroups", "sh_l1", "sh_lout")])


class BlockBwdAT(C.Structure):
    """``raae_block_bwd_a_t``"""
    _fields_ = ([("g1", GradT), ("ge", GradT), ("inp", ViewT), ("mask", C.c_void_p)] +
                [(n, C.c_int) for n in ("B", "Cin", "Cout", "Lin", "L1", "Lout", "E")] +
                [("cv1", ConvT), ("cvs", ConvT), ("has_short", C.c_int), ("has_excit", C.c_int)] +
                [(n, C.c_void_p) for n in ("w1", "ws", "wf1", "wf2", "se1", "E1", "dSh", "dT1", "dE2", "dE1", "dR",
                                           "pdR", "dslope1", "dslope_e2", "dslope_e1")] +
                [("slab_stride", C.c_long)] +
                [(n, C.c_int) for n in ("S", "ngroups", "sh_lin", "sh_l1", "sh_lout", "sh_e")])


class WgradConvT(C.Structure):
    _fields_ = [("go", GradT), ("cv", ConvT), ("inp", ViewT), ("dw", C.c_void_p), ("dbias", C.c_void_p)]


class WgradLinT(C.Structure):
    _fields_ = [("go", GradT), ("C", C.c_int), ("E", C.c_int), ("Lin", C.c_int), ("inp", ViewT), ("dw", C.c_void_p),
                ("dbias", C.c_void_p)]


class BlockWgradT(C.Structure):
    """``raae_block_wgrad_t``"""
    _fields_ = [("n_conv", C.c_int), ("n_lin", C.c_int), ("B", C.c_int), ("slab_stride", C.c_long),
                ("conv", WgradConvT * 4), ("lin", WgradLinT * 2)]


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
_PI = C.POINTER(C.c_int)
_PB = C.POINTER(BnT)
_PV, _PG, _PC = C.POINTER(ViewT), C.POINTER(GradT), C.POINTER(ConvT)

# name -> (restype, argtypes); every symbol include/rankaae_hip.h declares
SIGNATURES = {
    "raae_dense_fwd": (_I, [_P, _I, _I, _I, _P, _PB, _P, _P, _P, _I, _P, _I, _P, _P, _PI, _P]),
    "raae_dense_bwd": (_I, [_P, _I, _P, _I, _P, _P, _PB, _I, _I, _P, _I, _I, _P, _PB, _P, _P,
                            _P, _P, _P, _L, _PI, _P, _P, _P]),
    "raae_dense_bwd_st": (_I, [_P, _I, _P, _I, _P, _P, _PB, _I, _I, _P, _I, _I, _P, _PB, _P, _P,
                               _P, _P, _P, _L, _PI, _P, _P, _I, _P]),
    "raae_dense_bwd_s": (_I, [C.POINTER(DenseBwdT), _PI, _P]),
    "raae_stat_collapse2": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _P]),
    "raae_style_bn_fwd": (_I, [_P, _I, _I, _PB, _P, _P]),
    "raae_style_bn_bwd": (_I, [_P, _P, _I, _I, _PB, _F, _P, _P]),
    "raae_rank_loss_work_bytes": (_L, [_I, _I]),
    "raae_rank_loss_fwd_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "raae_rank_rows_pairs": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "raae_rank_rows_finish": (_I, [_P, _I, _I, _I, _I, C.c_float, _P, _P, _P, _I, _P]),
    "raae_style_metrics": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "raae_group_mean": (_I, [_P, _I, _I, _I, _P, _P]),
    "raae_recon_loss_fwd_bwd": (_I, [_P, _P, _I, _I, _I, _P, _PI, _P, C.POINTER(LossFinT), _P]),
    "raae_smooth_loss_fwd_bwd": (_I, [_P, _I, _I, C.POINTER(C.c_float), _I, _P, _PI, _P, C.POINTER(LossFinT), _P]),
    "raae_mse_fwd_bwd": (_I, [_P, _P, _L, _P, _PI, _P, C.POINTER(LossFinT), _P]),
    "raae_bce_pair_fwd_bwd": (_I, [_P, _I, _I, _P, _P, _P]),
    "raae_disc_input": (_I, [_P, _P, _P, _F, _I, _I, _I, _P, _P]),
    "raae_scale_by_dev": (_I, [_P, _P, _F, _L, _P, _P]),
    "raae_loss_finalize": (_I, [_P, _I, _F, _P, _I, _I, _P]),
    "raae_gather_batch": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P]),
    "raae_adam_step": (_I, [_P, _P, _P, _P, _L, _P, _L, _P, _P, _I, _I, _P]),
    "raae_conv_fwd": (_I, [_PV, _I, _PC, _P, _P, _P, _I, _P, _P, _PI, _I, _P]),
    "raae_conv_bwd_data": (_I, [_PG, _I, _PC, _P, _PV, _P, _I, _P, _PI, _P]),
    "raae_conv_bwd_weight": (_I, [_PG, _I, _PC, _PV, _P, _P, _P, _L, _PI, _P]),
    "raae_head_bwd_supported": (_I, [_PG, _I, _PC, _PV]),
    "raae_head_bwd": (_I, [_PG, _I, _PC, _P, _PV, _P, _P, _PI, _P, _P, _L, _PI, _P]),
    "raae_lenlin_fwd": (_I, [_PV, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P, _PI, _P]),
    "raae_lenlin_bwd_data": (_I, [_PG, _I, _I, _I, _P, _PV, _I, _P, _I, _P, _PI, _P]),
    "raae_lenlin_bwd_weight": (_I, [_PG, _I, _I, _I, _PV, _I, _P, _P, _P, _L, _PI, _P]),
    "raae_sum3_fwd": (_I, [_PV, _PV, _PV, _I, _I, _I, _P, _P, _PI, _P]),
    "raae_grad_materialize": (_I, [_PG, _I, _I, _I, _P, _I, _P, _L, _PI, _P]),
    "raae_block_fwd_a": (_I, [C.POINTER(BlockFwdAT), _PI, _P]),
    "raae_block_fwd_b": (_I, [C.POINTER(BlockFwdBT), _PI, _P]),
    "raae_disc_fused": (_I, [C.POINTER(DiscFusedT), _PI, _P]),
    "raae_dense_fwd2": (_I, [C.POINTER(DenseFwdT), C.POINTER(DenseFwdT), _PI, _PI, _P]),
    "raae_dense_fwd_s": (_I, [C.POINTER(DenseFwdT), _PI, _P]),
    "raae_block_fwd_a2": (_I, [C.POINTER(BlockFwdAT), C.POINTER(BlockFwdAT), _PI, _PI, _P]),
    "raae_block_fwd_b2": (_I, [C.POINTER(BlockFwdBT), C.POINTER(BlockFwdBT), _PI, _PI, _P]),
    "raae_block_bwd_b": (_I, [C.POINTER(BlockBwdBT), _PI, _P]),
    "raae_block_bwd_a": (_I, [C.POINTER(BlockBwdAT), _PI, _P]),
    "raae_block_wgrad": (_I, [C.POINTER(BlockWgradT), _PI, _P]),
    "raae_block_bwd_b_wgrad": (_I, [C.POINTER(BlockBwdBT), C.POINTER(BlockWgradT), _PI, _PI, _P]),
    "raae_slab_reduce": (_I, [_P, _L, _P, _L, _P, _I, _P]),
    "raae_step_tick": (_I, [_P, _I, C.c_uint, _P, _P, _I, _P]),
    "raae_step_begin": (_I, [C.POINTER(StepBeginT), _P]),
    "raae_rng_fill": (_I, [_P, _P, _P, _I, _L, C.c_ulonglong, _P, _P]),
    "raae_record_begin": (_I, []),
    "raae_record_end": (_I, [C.POINTER(C.c_void_p), _PI]),
    "raae_record_free": (_I, [_P]),
    "raae_multi_build": (_I, [C.POINTER(C.c_void_p), _I, C.POINTER(C.c_void_p)]),
    "raae_multi_launch": (_I, [_P, _P]),
    "raae_multi_count": (_I, [_P]),
    "raae_multi_free": (_I, [_P]),
    "raae_tile_hint": (_I, [_I]),
    "raae_tail_prepare": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "raae_graph_begin": (_I, [_P]),
    "raae_graph_end": (_I, [_P, C.POINTER(C.c_void_p)]),
    "raae_graph_launch": (_I, [_P, _P]),
    "raae_graph_destroy": (_I, [_P]),
    "raae_event_create": (_I, [C.POINTER(C.c_void_p)]),
    "raae_event_record": (_I, [_P, _P]),
    "raae_event_elapsed_ms": (_I, [_P, _P, C.POINTER(C.c_float)]),
    "raae_event_destroy": (_I, [_P]),
    "raae_stream_sync": (_I, [_P]),
    "raae_error_string": (C.c_char_p, [_I]),
    "raae_device_info": (_I, [_PI, _PI, C.c_char_p, _I]),
    "raae_abi_version": (_I, []),
    "raae_source_digest": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with rankaae_amd/csrc/build.sh (or __graft_entry__.build()). "
            "rankaae_amd has no CPU or PyTorch fallback for the training path.")
    # PyTorch ships its own libamdhip64; if this library were loaded first it would pull in the system copy and
    # the process would hold two HIP runtimes (kernels registered with one, the device owned by the other:
    # "no ROCm-capable device is detected").  Importing torch first makes both resolve to the same runtime.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.raae_abi_version() != ABI_VERSION:
        raise HipLibraryMissing(f"{LIB_PATH} has ABI version {lib.raae_abi_version()}, this binding needs "
                                f"{ABI_VERSION}: rebuild with rankaae_amd/csrc/build.sh")
    want = source_digest()
    got = lib.raae_source_digest().decode()
    if want is not None and got != want:
        raise HipLibraryMissing(f"{LIB_PATH} is stale (built from sources {got}, tree has {want}): "
                                "rebuild with rankaae_amd/csrc/build.sh")
    _lib = lib
    return lib


def source_digest():
    """sha256 prefix over the header and kernel sources, in build.sh's order; None when the sources are
    not shipped next to the library."""
    import glob
    import hashlib
    src = os.path.join(_HERE, "csrc")
    hdr = os.path.join(_HERE, "..", "include", "rankaae_hip.h")
    if not os.path.isdir(src) or not os.path.exists(hdr):
        return None
    files = [hdr]
    for pat in ("raae_*.h", "raae_*.inc", "raae_*.hip"):
        files += sorted(glob.glob(os.path.join(src, pat)))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(code, what=""):
    if code != 0:
        msg = load().raae_error_string(int(code))
        raise HipCallError(f"{what}: error {code}: {msg.decode() if msg else '?'}")
