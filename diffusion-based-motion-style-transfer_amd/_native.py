"""ctypes binding of libmst_engine.so (include/mst_engine.h).

The library is the product path: if it is missing or does not load, every entry point raises --
there is no Python/CPU fallback.  `import torch` happens first so that the HIP runtime the library
binds to (SONAME libamdhip64.so.7) is the one PyTorch already loaded; streams and device pointers
are then shared with torch tensors."""
import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library must share)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MST_ENGINE_LIB") or os.path.join(_HERE, "libmst_engine.so")   # env: A/B builds
_lib = None


class MstConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "feats", "max_frames", "max_rows", "latent_dim", "num_heads", "ff_size", "num_layers",
        "clip_dim", "pe_len", "device")]


class MstLoopArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("frames", C.c_int32), ("cfg", C.c_int32), ("sampler", C.c_int32),
        ("mask_noise", C.c_int32), ("clip_denoised", C.c_int32), ("noise_mode", C.c_int32),
        ("t_start", C.c_int32), ("t_end", C.c_int32), ("eta", C.c_float), ("seed", C.c_uint64),
        ("scale_dev", C.c_void_p), ("inpainting_mask_dev", C.c_void_p),
        ("inpainted_motion_dev", C.c_void_p), ("noise_dev", C.c_void_p), ("x_dev", C.c_void_p),
        ("xstart_dump_dev", C.c_void_p)]


# name -> (restype, argtypes); must list every function include/mst_engine.h declares
SIGNATURES = {
    "mst_last_error": (C.c_char_p, []),
    "mst_version": (C.c_int, []),
    "mst_engine_create": (C.c_int, [C.POINTER(MstConfig), C.POINTER(C.c_void_p)]),
    "mst_engine_destroy": (None, [C.c_void_p]),
    "mst_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]),
    "mst_weights_complete": (C.c_int, [C.c_void_p]),
    "mst_schedule_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "mst_schedule_destroy": (None, [C.c_void_p]),
    "mst_set_text": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mst_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                              C.c_void_p, C.c_void_p]),
    "mst_sample_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MstLoopArgs), C.c_void_p]),
    "mst_loop_slices": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mst_q_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                               C.c_void_p, C.c_void_p]),
    "mst_step_epilogue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_philox_normal": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p]),
    "mst_train_tape_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "mst_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mst_train_model_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_train_model_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                           C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mst_dropout_mask": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mst_adamw_workspace_bytes": (C.c_int64, [C.c_int32, C.POINTER(C.c_int64)]),
    "mst_adamw_step": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mst_recover_from_ric": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "mst_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "mst_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                   C.c_int32]),
    "mst_debug_stop_after": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mst_debug_copy": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]),
}


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and "MST_ENGINE_LIB" not in os.environ:
            _build_in_place()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The HIP library is the only implementation of the denoising path; there is no fallback.")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _build_in_place():
    """Fresh checkout (the .so is git-ignored): compile the library with hipcc where it belongs.
    Same command as __graft_entry__.build(); failures surface as the 'missing' error above."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return
    try:
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB_PATH,
                        "mst_engine.hip"], cwd=os.path.join(_HERE, "csrc"), check=True)
    except (subprocess.CalledProcessError, OSError):
        if os.path.exists(LIB_PATH):
            os.remove(LIB_PATH)


def check(rc):
    if rc != 0:
        raise RuntimeError("mst_engine: " + lib().mst_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
