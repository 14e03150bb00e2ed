"""ctypes binding of libspegnet_hip.so (the C ABI declared in include/spegnet_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  The product path never
imports the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPG_LIBRARY=<path>: tools/ point this at libspegnet_hip_dev.so (python spegnet_amd/build.py --dev), the build that still contains the
# superseded kernel families and their ablation switches; the product library is the default and the only one tests / bench load
LIB_PATH = os.environ.get("SPG_LIBRARY") or os.path.join(_HERE, "libspegnet_hip.so")

SPG_F32, SPG_BF16 = 0, 1
ABI_VERSION = 309   # = SPG_ABI_VERSION of include/spegnet_hip.h that SIGNATURES below was written for
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
ACT_GELU_SAVE_GRAD, ACT_MUL_H = 3, 4   # bf16: C2 = gelu'(pre) saved by the forward GEMM | C = acc * gelu_h in the backward GEMM

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argument type string (p pointer, i int, l long, f float); every function returns int
SIGNATURES = {
    "spg_gemm_nt": "ipppppppiiiiiiiiiiiip",
    "spg_gemm_tn": "ipppppliiiiiiiiiiiip",
    "spg_gemm_tn_group": "ii" "pppp" "i" "ppppp" "plp" "i" "p",
    "spg_gemm_tn_group_reduce_batch": "ippp",
    "spg_gemm_tn_blocks": "ii" "pppp" "i" "ppppp" "ip" "i" "p",
    "spg_pack_matrix": "ippiiip",
    "spg_pack_batch": "ipiip",
    "spg_pack_conv3x3": "ipppiip",
    "spg_unpack_conv3x3_grad": "ppiip",
    "spg_layernorm_fwd": "ipppppp" "iifp",
    "spg_layernorm_bwd": "ippppppppp" "ii" "plp" "p",
    "spg_layernorm_param_grads_batch": "ii" "pppppp" "ppp" "plp" "p",
    "spg_attn_fwd": "ippppp" "iiiiiip",
    "spg_attn_bwd": "ippppppppp" "p" "iiiiiip",
    "spg_maxpool2_fwd": "ippp" "iiiiiip",
    "spg_maxpool2_bwd": "ippp" "iiiiiip",
    "spg_patch_im2col": "ipp" "iiiip",
    "spg_preprocess_image": "pp" "iiii" "ppp",
    "spg_preprocess_batch": "pppp" "p" "iii" "ppp",
    "spg_colsum": "ipp" "iiii" "plp" "p",
    "spg_gap_sum": "ipp" "ili" "plp" "p",
    "spg_chan_prod_sum": "ippp" "ili" "plp" "p",
    "spg_add": "ippp" "lp",
    "spg_cast_bf16": "pp" "lip",
    "spg_cast_bf16_sq": "pp" "lpip",
    "spg_copy_channels": "ipp" "liiiiiip",
    "spg_bn_stats": "ipp" "li" "plp" "p",
    "spg_bn_stats_finalize": "ipp" "ppppppp" "liff" "plp" "p",
    "spg_bn_stats_finalize4": "ipp" "ppppppp" "liff" "plp" "p",
    "spg_bn_stats_finalize_part": "plp" "ppppppp" "liff" "plp" "p",
    "spg_conv3x3_fwd_stats": "ippppp" "l" "iiiiii" "p",
    "spg_conv3x3_wgrad": "ipppp" "pl" "iiiiiii" "p",
    "spg_bn_finalize": "ppppppp" "liffip",
    "spg_bn_apply": "ippp" "liip",
    "spg_bn_bwd_reduce": "ippppp" "lii" "plp" "p",
    "spg_bn_bwd_apply": "ippppppppp" "liip",
    "spg_upsample_bilinear": "ipp" "iiiiiiiip",
    "spg_upsample_bilinear_bwd": "ipp" "iiiiiiiiip",
    "spg_se_fc": "ppppp" "iiifp",
    "spg_se_fc_bwd": "ppppppppp" "iiif" "plp" "p",
    "spg_pack_cols2": "ipipi" "piip",
    "spg_add_cols_batch": "ipppppp" "p",
    "spg_chan_scale": "ippp" "ilip",
    "spg_chan_scale_bwd": "ipppp" "ilip",
    "spg_dwconv3x3": "ippp" "iiiiiip",
    "spg_dwconv3x3_wgrad": "ippp" "iiiii" "plp" "p",
    "spg_easpp_fuse": "ippppppp" "ilip",
    "spg_easpp_fuse_bwd": "ipppppppppppp" "p" "ili" "plp" "p",
    "spg_head1x1": "ipppp" "lip",
    "spg_head1x1_bwd": "ipppppp" "lii" "plp" "p",
    "spg_dwconv4": "ipppp" "iiii" "p",
    "spg_dwconv4_dgrad": "ippppp" "iiii" "p",
    "spg_dwconv4_wgrad": "ipppp" "iiii" "plp" "p",
    "spg_easpp_fuse_bn": "ippppp" "ili" "p",
    "spg_easpp_fuse_bn_bwd": "ippppppppppp" "ili" "plp" "p",
    "spg_easpp_global_fwd": "pppppppppppp" "iilffi" "p",
    "spg_easpp_global_bwd": "ppppppppppppp" "iili" "p",
    "spg_cfi_combine": "ipppp" "iiiiiiii" "p",
    "spg_bn_apply_head": "ippppp" "p" "lii" "p",
    "spg_ped_gather": "ipp" "iii" "p" "iii" "p" "iii" "p",
    "spg_ped_gather_bwd": "ipp" "iiiiiiiii" "p",
    "spg_bn_bwd_head": "ippppppp" "pppppp" "li" "plp" "p",
    "spg_loss_weight_map": "pppp" "iif" "plp" "p",
    "spg_loss_reduce": "ippppp" "iiiiiff" "plp" "p",
    "spg_loss_finalize": "pppp" "iiffffffp",
    "spg_loss_grad": "ippppppp" "iiiiifffff" "pp",
    "spg_prefetch_hint": "pl",
    "spg_loss_reduce_all": "ippp" "pppppp" "iiff" "plp" "p",
    "spg_loss_grad_all": "ippppp" "ppppppp" "iiffff" "pp",
    "spg_sumsq": "pp" "l" "plp" "p",
    "spg_sumsq_fold": "ppii" "pp" "p" "plp" "p",
    "spg_adamw": "ppppppppp" "fffff" "ilp",
    "spg_adamw_pack": "i" "ppppppppp" "fffff" "i" "pii" "p",
}
# name -> (return type, argument types): host-side size queries
QUERIES = {
    "spg_gemm_tn_workspace_bytes": ("l", "iiii"),
    "spg_conv3x3_stats_rows": ("l", "iiiiiii"),
    "spg_conv3x3_wgrad_workspace_bytes": ("l", "iiiiiii"),
    "spg_gemm_tn_group_workspace_bytes": ("l", ""),
    "spg_gemm_tn_group_desc_bytes": ("l", ""),
    "spg_gemm_tn_blocks_count": ("l", "iipp"),
    "spg_num_cus": ("i", "i"),
    "spg_reduce_workspace_floats": ("l", "iii"),
    "spg_reduce_counters": ("i", "iii"),
    "spg_layernorm_param_grads_batch_workspace_floats": ("l", "ip"),
    "spg_loss_workspace_floats": ("l", "ii"),
    "spg_loss_reduce_all_workspace_floats": ("l", "i"),
    "spg_head1x1_bwd_workspace_floats": ("l", "i"),
    "spg_bn_bwd_head_workspace_floats": ("l", "ii"),
    "spg_easpp_fuse_bn_bwd_workspace_floats": ("l", "ii"),
    "spg_easpp_fuse_bn_bwd_counters": ("i", "ii"),
    "spg_bn_bwd_head_counters": ("i", "ii"),
}
# dev library only (python spegnet_amd/build.py --dev; csrc/dev/nt_chain_host.inc): the chained-launch experiment of tools/chain_bench.py
_OPTIONAL = {"spg_nt_chain": "ii" "p" "pl" "p" "i" "p"}
_OPTIONAL_QUERIES = {"spg_nt_chain_counter_words": ("l", "ip")}


class ChainPhase(ctypes.Structure):
    """spg_chain_phase_t of include/spegnet_hip.h (one phase of spg_nt_chain)"""
    _fields_ = [("kind", _I), ("act", _I), ("depends", _I), ("M", _I), ("N", _I), ("K", _I),
                ("x", _P), ("w", _P), ("c", _P), ("c2", _P), ("bias", _P), ("residual", _P), ("gelu_h", _P)]


CHAIN_GEMM = 0
CHAIN_MAX_PHASES = 6
_CT = {"p": _P, "i": _I, "l": _L, "f": _F}

_lib = None


def load() -> ctypes.CDLL:
    """Loads the shared library once; raises if it is absent (build with `python spegnet_amd/build.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  Loaded after this library, the process would hold two
    # runtimes -- this library bound to the system one, which then finds no device ("no ROCm-capable device is detected" on the first
    # launch; seen on the GPU box when build() and smoke() ran in one process)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the SPEGNet HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.spg_last_error.restype = ctypes.c_char_p
    lib.spg_version.restype = _I
    have = int(lib.spg_version())
    if have != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} is ABI revision {have}, this binding was written for {ABI_VERSION}: a stale library would receive "
                           "shifted arguments. Rebuild it (`python spegnet_amd/build.py --force`, or `--dev` for the tools/ library).")
    for name, (res, sig) in QUERIES.items():   # size queries: return a count, not a status
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _CT[res], [_CT[c] for c in sig]
    for name, (res, sig) in _OPTIONAL_QUERIES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = _CT[res], [_CT[c] for c in sig]
    for table, required in ((SIGNATURES, True), (_OPTIONAL, False)):
        for name, sig in table.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if required:
                    raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild the extension")
                continue
            fn.restype = _I
            fn.argtypes = [_CT[c] for c in sig]
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.spg_last_error().decode()}")
