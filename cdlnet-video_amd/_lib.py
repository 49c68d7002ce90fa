"""ctypes binding of libcdlnet_hip.so (C ABI declared in include/cdlnet_hip.h).

The library is the product: there is no fallback.  `lib()` raises
`HipLibraryMissing` when the shared object has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C cdlnet-video_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDLNET_HIP_LIB points at an alternative build of the same library (kernel experiments); default in-tree
LIB_PATH = os.environ.get("CDLNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libcdlnet_hip.so")

CDL_EINVAL = -10001
CDL_EUNSUPPORTED = -10002


class HipLibraryMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


class Geom(ctypes.Structure):
    """Mirror of `cdl_geom` (include/cdlnet_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in
                ("N", "C", "M", "D", "H", "W", "Pd", "Ph", "Pw", "pd", "ph", "pw", "sd", "sh", "sw")]


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_G = ctypes.POINTER(Geom)
_IP = ctypes.POINTER(ctypes.c_int)

# name -> argtypes; every function returns int except cdl_version
SIGNATURES = {
    "cdl_preprocess": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _IP, _P],
    "cdl_postprocess": [_P, _P, _P, _I, _I, _I, _I, _I, _IP, _P],
    "cdl_postprocess_bwd": [_P, _P, _I, _I, _I, _I, _I, _IP, _P],
    "cdl_thresholds": [_P, _P, _P, _I, _I, _I, _P],
    "cdl_shrink": [_P, _P, _P, _I, ctypes.c_size_t, _P],
    "cdl_analysis": [_G, _P, _P, _F, _P, _P, _P, _P, _P],
    "cdl_synthesis": [_G, _P, _P, _P, _F, _P, _P, _P, _P],
    "cdl_synthesis_ws": [_G, _P, _P, _P, _F, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "cdl_wgrad": [_G, _P, _P, _P, _F, _P, _P, ctypes.c_size_t, _P],
    "cdl_wgrad_pair": [_G, _P, _P, _F, _P, _P, _P, _F, _P, _P, ctypes.c_size_t, _P],
    "cdl_tau_grad": [_G, _P, _P, _P, _P, _P, _P, _P],
    "cdl_tau_grad_gate": [_G, _P, _P, _P, _P, _P, _P, _P],
    "cdl_analysis_ws": [_G, _P, _P, _F, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "cdl_analysis_rev_ws": [_G, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "cdl_analysis_prox_ws": [_G, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "cdl_analysis_prox": [_G, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "cdl_ista_forward": [_G, _I] + [_P] * 14 + [ctypes.c_size_t, _P],
    "cdl_ista_backward": [_G, _I] + [_P] * 26 + [ctypes.c_size_t, _P],
    "cdl_nle_mad": [_P, _P, _P, ctypes.c_size_t, _I, _I, _I, _I, _P],
    "cdl_residual_forward": [_G, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "cdl_residual_backward": [_G] + [_P] * 11 + [ctypes.c_size_t, _P],
    "cdl_prox_csr": [_G, _P, _P, _P, _P, _P, _P, _P, _P],
    "cdl_prox_csr_bwd": [_G] + [_P] * 15 + [ctypes.c_size_t, _P],
    "cdl_project_filters": [_P, _I, _I, _P],
    "cdl_project_filter_banks": [_P, _I, _I, _I, _P],
    "cdl_gabor_filters": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cdl_gabor_filters_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cdl_options_reload": [],
    "cdl_set_exact_fp32": [_I],
    "cdl_gabor_filter_banks": [_I, _P, _P, _P, _P, _IP, _P, _I, _I, _I, _I, _P],
    "cdl_gabor_filter_banks_bwd": [_I, _P, _P, _P, _P, _IP, _P, _P, _I, _I, _I, _I, _P],
    "cdl_fused2d_supported": [_G],
    "cdl_fused2d_prep": [_P, _P, _P, _I, _I, _P],
    "cdl_fused2d_iter_fwd": [_G, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P],
    "cdl_fused2d_support_map": [_G, _P, _P, _P],
    "cdl_fused2d_timing": [_I],
    "cdl_fused2d_timing_read": [ctypes.POINTER(ctypes.c_double), _IP],
    "cdl_fused2d_assemble": [_G, _P, _P, _P, _F, _P, _P],
    "cdl_fused2d_stage_bwd": [_G, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "cdl_fused2d_stage_bwd_da": [_G, _P, _P, _P, _P, _P, _P, _P, _I, _P, _F, _P, _P, _I, _P],
    "cdl_fused2d_dtau_reduce": [_G, _P, _P, _P, _P, _P],
    "cdl_fused2d_wgrad": [_G, _P, _P, _F, _P, _P, _P, _F, _P, _P, _I, _P],
    "cdl_fused2d_forward": [_G, _I] + [_P] * 11 + [_I, _P],
    "cdl_fused2d_backward": [_G, _I] + [_P] * 20 + [_I, _P],
    "cdl_fusedg_supported": [_G],
    "cdl_fusedg_code_layout": [_G, _I],
    "cdl_fusedg_set_timeline": [_P],
    "cdl_fusedg_prep": [_G, _P, _P, _P, _P],
    "cdl_fusedg_iter_fwd": [_G, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P],
    "cdl_fusedg_stage_bwd": [_G, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "cdl_fusedg_assemble": [_G, _P, _P, _P, _F, _P, _P],
    "cdl_fusedg_dtau_reduce": [_G, _P, _P, _P, _P, _P],
    "cdl_fusedg_forward": [_G, _I] + [_P] * 11 + [_I, _P],
    "cdl_fusedg_backward": [_G, _I] + [_P] * 20 + [ctypes.c_size_t, _I, _P],
}
SIZE_T_FUNCS = {"cdl_fusedg_code_floats": [_G, _I], "cdl_fusedg_frag_bytes": [_G], "cdl_fusedg_patch_floats": [_G], "cdl_fusedg_tiles": [_G],
                "cdl_fusedg_map_words": [_G], "cdl_fused2d_frag_bytes": [_I], "cdl_fused2d_patch_floats": [_G], "cdl_fused2d_code_bytes": [_G, _I],
                "cdl_fused2d_tiles": [_G], "cdl_fused2d_map_words": [_G], "cdl_fused2d_wgrad_workspace_floats": [_G],
                "cdl_wgrad_workspace_floats": [_G], "cdl_prox_csr_scratch_floats": [_G],
                "cdl_synthesis_workspace_floats": [_G], "cdl_ista_scratch_floats": [_G], "cdl_analysis_workspace_floats": [_G], "cdl_analysis_rev_workspace_floats": [_G],
                "cdl_nle_mad_scratch_floats": [_I, _I, _I, _I], "cdl_residual_scratch_floats": [_G]}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: the HIP kernels are the only compute path of this package; "
                "build them with `make -C cdlnet-video_amd/csrc` (hipcc --offload-arch=gfx950).")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = header / library mismatch
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        for name, argtypes in SIZE_T_FUNCS.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_size_t
        handle.cdl_version.restype = ctypes.c_char_p
        handle.cdl_version.argtypes = []
        _lib = handle
    return _lib


def reload_options():
    """Make the library re-read the CDL_* environment switches (it snapshots them at first use)."""
    check(lib().cdl_options_reload(), "cdl_options_reload")


def available():
    return os.path.exists(LIB_PATH)


def check(rc, name):
    if rc == 0:
        return
    if rc == CDL_EINVAL:
        raise HipKernelError(f"{name}: invalid argument (CDL_EINVAL)")
    if rc == CDL_EUNSUPPORTED:
        raise HipKernelError(f"{name}: shape not supported by this kernel (CDL_EUNSUPPORTED)")
    raise HipKernelError(f"{name}: HIP runtime error {-rc}")
