"""
ctypes binding of ``libgadfly_hip.so`` (C-ABI declared in ``include/gadfly_hip.h``).

This is the ONLY compute backend of the package: there is no CPU fallback.  Loading
fails loudly if the shared library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``gadfly_amd._lib.build()``), and every compute
call requires a HIP device (``torch.cuda.is_available()``).  PyTorch is used only for
device buffers and streams; the arguments that cross the boundary are raw device
pointers and sizes.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libgadfly_hip.so")
if os.environ.get("GADFLY_SO"):                 # another build of the same library (A/B measurements)
    SO_PATH = os.path.abspath(os.environ["GADFLY_SO"])
SOURCES = [os.path.join(CSRC, "gadfly_hip.hip"), os.path.join(CSRC, "gadfly_dense.hip")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "gadfly_hip.h")

GF_SOLVE_LOWER, GF_SOLVE_UPPER, GF_MATMUL_LOWER = 0, 1, 2
GF_SWEEP_AUTO, GF_SWEEP_COLUMN, GF_SWEEP_TILED = 0, 1, 2
GF_SWEEP_ZERO_START = 0x100
GF_MAX_WIDTH = 256

_lib = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/gadfly_hip.h declares
SIGNATURES = {
    "gf_version": (_int, []),
    "gf_last_error": (ctypes.c_char_p, []),
    "gf_leading_dim": (_int, [_int]),
    "gf_build_matrices": (_int, [_int, _i64, _i64, _int, _int, _int] + [_vp] * 7
                          + [_vp, _i64, _vp, _i64] + [_vp] * 4 + [_vp]),
    "gf_state_size": (_i64, [_int]),
    "gf_state_cols": (_int, [_int]),
    "gf_factor": (_int, [_int, _i64, _i64, _int, _int] + [_vp] * 4 + [_vp, _i64]
                  + [_vp] * 6 + [_vp]),
    "gf_scaled_supported": (_int, [_int]),
    "gf_scaled_wide_supported": (_int, [_int]),
    "gf_build_scaled": (_int, [_int, _i64, _i64, _int, _int, _int] + [_vp] * 8 + [_int]
                        + [_vp, _i64, _vp, _i64] + [_vp] * 4 + [_vp]),
    "gf_factor_scaled": (_int, [_int, _i64, _i64, _int, _int] + [_vp] * 5 + [_vp, _i64]
                         + [_vp] * 5 + [_vp]),
    "gf_fused_supported": (_int, [_int, _int]),
    "gf_fused_state_size": (_i64, [_int, _int]),
    "gf_loglike_fused": (_int, [_int, _i64, _i64, _int, _int, _int, _int, _int] + [_vp] * 8
                         + [_vp, _i64, _vp, _i64, _vp, _i64] + [_vp] * 5 + [_vp]),
    "gf_chunk_sweep": (_int, [_int, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int] + [_vp] * 8
                       + [_vp, _i64, _vp, _i64, _vp, _i64] + [_vp] * 9 + [_vp]),
    "gf_fused_row_stride": (_int, [_int, _int]),
    "gf_scaled_propagator": (_int, [_int, _i64, _int, _int, _vp, _vp, _vp, _vp]),
    "gf_chunk_linear": (_int, [_int, _int, _i64, _i64, _int, _int, _int, _int, _int]
                        + [_vp] * 8 + [_vp]),
    "gf_chunk_linear_combine": (_int, [_int, _int, _i64, _i64, _int, _int, _int] + [_vp] * 5
                                + [_vp]),
    "gf_dense_solve": (_int, [_int, _int, _int] + [_vp] * 2 + [_vp]),
    "gf_dense_solve_logdet": (_int, [_int, _int, _int] + [_vp] * 3 + [_vp]),
    "gf_dense_width": (_int, [_int]),
    "gf_wide_combine_work": (_i64, [_int, _int, _int]),
    "gf_wide_combine": (_int, [_int, _i64, _i64, _int, _int] + [_vp] * 7 + [_vp]),
    "gf_chunk_corrections_work": (_i64, [_int, _int]),
    "gf_chunk_corrections": (_int, [_int, _int, _int, _int, _int] + [_vp] * 6 + [_vp]),
    "gf_wide_gram": (_int, [_int, _i64, _i64, _int, _int, _int, _int, _int] + [_vp] * 5 + [_vp]),
    "gf_lft_tree_work": (_i64, [_int, _int, _int]),
    "gf_lft_tree_scan": (_int, [_int, _int, _int] + [_vp] * 8 + [_vp]),
    "gf_bgemm": (_int, [_int, _int, _int, _int, _int, _int, _vp, _int, _i64, _vp, _int, _i64,
                        _vp, _int, _i64, _vp, _int, _i64, _vp]),
    "gf_chunk_segment_transitions": (_int, [_int, _int, _int] + [_vp] * 4 + [_vp]),
    "gf_chunk_linear_combine_seg": (_int, [_int, _int, _int, _int, _int] + [_vp] * 6 + [_vp]),
    "gf_chunk_transition": (_int, [_int, _i64, _i64, _int, _int, _int, _int, _int, _int] + [_vp] * 9 + [_vp]),
    "gf_chunk_transition_wide": (_int, [_int, _i64, _i64, _int, _int, _int, _int] + [_vp] * 7 + [_vp]),
    "gf_chunk_combine": (_int, [_int, _int, _int] + [_vp] * 5 + [_vp]),
    "gf_chunk_combine_tree_work": (_i64, [_int, _int, _int]),
    "gf_chunk_combine_tree": (_int, [_int, _int, _int, _int] + [_vp] * 8 + [_vp]),
    "gf_reduce_work": (_i64, [_i64]),
    "gf_reduce_tile": (_int, [_int, _i64] + [_vp] * 4 + [_int, _vp]),
    "gf_loglike_finish": (_int, [_int, _i64] + [_vp] * 4 + [_vp]),
    "gf_solve": (_int, [_int, _int, _i64, _int, _int, _int] + [_vp] * 6 + [_vp]),
    "gf_solve_chunk": (_int, [_int, _int, _i64, _i64, _int, _int, _int] + [_vp] * 7 + [_int, _vp]),
    "gf_solve_chunk_rhs": (_int, [_int, _int, _i64, _i64, _int, _int, _int, _int] + [_vp] * 7 + [_int, _vp]),
    "gf_chunk_diag_scan": (_int, [_int, _int, _int, _int, _vp, _vp, _vp]),
    "gf_cross_covariance": (_int, [_int, _i64, _int, _int, _int] + [_vp] * 6 + [_vp, _i64, _vp, _i64, _vp, _vp]),
    "gf_interp_work": (_i64, [_i64]),
    "gf_interp_plan": (_int, [_i64, _vp, _vp, _dbl, _vp, _vp, _vp]),
    "gf_interp_fill": (_int, [_i64, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp]),
    "gf_psd_power": (_int, [_int, _i64, _i64, _dbl, _vp, _vp, _vp]),
    "gf_psd_bin": (_int, [_int, _i64, _int, _vp, _vp, _vp, _dbl, _vp, _vp, _vp]),
    "gf_general_matmul_work": (_i64, [_int, _i64, _i64, _int]),
    "gf_general_matmul": (_int, [_int, _i64, _i64, _int, _int, _vp,
                                 _vp, _i64, _vp, _vp,
                                 _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp]),
}


class GadflyHipError(RuntimeError):
    """Raised when the native library is missing or reports an argument/launch error."""


HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"]
_DEPS = [HEADER] + [os.path.join(CSRC, h) for h in ("fastmath.h", "gf_internal.h", "gf_wave.h")]


def hipcc_command(out=SO_PATH):
    """The one-shot form of the build (what `build()` does per translation unit, then links)."""
    return ["hipcc"] + HIPCC_FLAGS + ["-shared", "-o", out] + SOURCES


def _source_hash(src):
    """SHA-256 over a translation unit, the headers it includes and the compiler flags: what its object is
    a function of (file times say nothing after a checkout or a copy to another machine)."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for path in [src] + _DEPS:
        with open(path, "rb") as fh:
            h.update(b"\0" + os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (cross-compiles without a GPU): one object per
    translation unit, rebuilt when the hash of its source + shared headers + flags differs from the one
    recorded beside the object (``<unit>.o.srchash``), then the link (recorded the same way beside the library)."""
    objs, hashes = [], []
    for src in SOURCES:
        obj = os.path.splitext(src)[0] + ".o"
        tag = obj + ".srchash"
        want = _source_hash(src)
        objs.append(obj)
        hashes.append(want)
        have = open(tag).read().strip() if os.path.exists(tag) else None
        if force or not os.path.exists(obj) or have != want:
            cmd = ["hipcc"] + HIPCC_FLAGS + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            with open(tag, "w") as fh:
                fh.write(want + "\n")
    tag = SO_PATH + ".srchash"
    want = " ".join(hashes)
    have = open(tag).read().strip() if os.path.exists(tag) else None
    if force or not os.path.exists(SO_PATH) or have != want:
        cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", SO_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(tag, "w") as fh:
            fh.write(want + "\n")
    elif verbose:
        print(f"{SO_PATH}: up to date with its sources (hash {hashes[0][:12]} ...)")
    return SO_PATH


def load():
    """Load the shared library and bind every declared symbol (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise GadflyHipError(
            f"{SO_PATH} not found: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "gadfly_amd has no CPU fallback."
        )
    # The library links against libamdhip64; PyTorch ships its own copy of that runtime.  Whichever is
    # loaded first serves both, and device buffers made by one runtime are unknown to the other (kernel
    # launches then fail with "no ROCm-capable device"): load torch's runtime first, always.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().gf_last_error().decode()


def check(status, what):
    if status != 0:
        raise GadflyHipError(f"{what} failed (status {status}): {last_error()}")


def require_device():
    """The product path needs a HIP device; refuse to run anything without one."""
    import torch
    if not torch.cuda.is_available():
        raise GadflyHipError(
            "no HIP device available: gadfly_amd computes only on an MI355X-class GPU "
            "(there is deliberately no CPU fallback)."
        )
    load()
    return torch


def ptr(tensor):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if tensor is None else tensor.data_ptr()


def stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream
