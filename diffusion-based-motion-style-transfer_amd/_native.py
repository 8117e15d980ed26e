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
    "mst_source_hash": (C.c_char_p, []),
    "mst_engine_create": (C.c_int, [C.POINTER(MstConfig), C.POINTER(C.c_void_p)]),
    "mst_engine_destroy": (None, [C.c_void_p]),
    "mst_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]),
    "mst_load_layers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mst_weights_complete": (C.c_int, [C.c_void_p]),
    "mst_schedule_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "mst_schedule_destroy": (None, [C.c_void_p]),
    "mst_set_text": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mst_set_text_dropped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mst_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                              C.c_void_p, C.c_void_p]),
    "mst_sample_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MstLoopArgs), C.c_void_p]),
    "mst_loop_slices": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "mst_q_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                               C.c_void_p, C.c_void_p]),
    "mst_step_epilogue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_step_epilogue_mt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_step_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64,
                                    C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_masked_l2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_text_cosine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_philox_normal": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p]),
    "mst_train_tape_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "mst_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mst_train_model_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mst_train_model_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                           C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mst_motion_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                             C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mst_motion_encoder_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                              C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mst_train_wait_layer_grads": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mst_dropout_mask": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mst_adamw_workspace_bytes": (C.c_int64, [C.c_int32, C.POINTER(C.c_int64)]),
    "mst_adamw_step": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "mst_recover_from_ric": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "mst_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "mst_profile_event_overhead_us": (C.c_float, [C.c_void_p]),
    "mst_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                   C.c_int32]),
    "mst_set_precise": (C.c_int, [C.c_void_p, C.c_int32]),
    "mst_set_trunk_groups": (C.c_int, [C.c_void_p, C.c_int32]),
    "mst_trunk_check": (C.c_int, [C.c_void_p]),
    "mst_debug_stop_after": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mst_debug_copy": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]),
}


# The one compile recipe.  No -D but the source hash is ever passed: the kernels' tunables are constants in the sources, and the
# diagnostic hooks (phase stamps) exist only in the probes under csrc/probes/, which define MST_PROBE_BUILD themselves.
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               # no SLP packing: hipcc turns pairs of scalar f32 ops into v_pk_fma_f32 / v_pk_mul_f32 + v_mov shuffles, which buy nothing
               # on CDNA4 (a packed op issues at the rate of its two scalar ones) and cost the shuffles: +0.4 % clips/s, same-box A/B
               "-fno-slp-vectorize")


def source_files():
    """Everything the library is compiled from: csrc/*.hip, csrc/*.h and the public header."""
    import glob
    csrc = os.path.join(_HERE, "csrc")
    return sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + \
        [os.path.join(os.path.dirname(_HERE), "include", "mst_engine.h")]


def source_hash():
    """Hash of the sources AND of the compile flags: a library built with other flags is stale, not silently different."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode() + b"\0")
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_hash():
    """Source hash the in-tree library was built from (side file written by the build), or None."""
    try:
        with open(LIB_PATH + ".srchash") as fh:
            return fh.read().strip()
    except OSError:
        return None


def is_stale():
    return not os.path.exists(LIB_PATH) or built_hash() != source_hash()


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it cannot be loaded."""
    global _lib
    if _lib is None:
        custom = "MST_ENGINE_LIB" in os.environ
        if not custom and is_stale():          # fresh checkout (the .so is git-ignored) or a kernel source edited since the build
            build_in_place()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The HIP library is the only implementation of the denoising path; there is no fallback.")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if not custom:
                    raise
                continue        # an OLDER build named by MST_ENGINE_LIB (same-box A/B, tools/lib_ab.sh): entry points added since are simply absent
            fn.restype = res
            fn.argtypes = args
        if not custom and l.mst_source_hash().decode() != source_hash():
            raise RuntimeError(f"{LIB_PATH} was built from other sources than the ones in csrc/ (hipcc missing or the build "
                               "failed?): rebuild it with `python -c 'import __graft_entry__ as g; g.build(force=True)'`")
        _lib = l
    return _lib


def build_in_place(check=False, force=False):
    """Compile the library with hipcc where it belongs (the one build recipe: __graft_entry__.build() calls this too).
    The source hash (sources + flags) is compiled in (mst_source_hash) and written beside the .so, so a library that is older
    than an edit of ANY csrc/ file is rebuilt instead of silently used.

    Safe when several ranks import the package at once (bench.py --gpus N, torchrun): the build runs under an exclusive lock on
    LIB_PATH + '.lock', staleness is re-checked once the lock is held (the winner's library is then simply used), the compiler
    writes to a temporary file that is renamed over the library, and an existing library is never deleted -- a rank whose own
    compile fails reports it (check=True) or leaves whatever is there (check=False) for lib() to judge by its hash."""
    import fcntl
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if check:
            raise RuntimeError("hipcc not found")
        return
    if any("MST_PROBE_BUILD" in f for f in HIPCC_FLAGS):
        raise RuntimeError("MST_PROBE_BUILD is for csrc/probes/ only: the product library is never built with diagnostic hooks")
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return                                  # another process built it while we waited
            h = source_hash()
            tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
            try:
                subprocess.run([hipcc, *HIPCC_FLAGS, f'-DMST_SRC_HASH="{h}"', "-o", tmp, "mst_engine.hip"],
                               cwd=os.path.join(_HERE, "csrc"), check=True)
                os.replace(tmp, LIB_PATH)
                with open(LIB_PATH + ".srchash.tmp", "w") as fh:
                    fh.write(h + "\n")
                os.replace(LIB_PATH + ".srchash.tmp", LIB_PATH + ".srchash")
            except (subprocess.CalledProcessError, OSError):
                if os.path.exists(tmp):
                    os.remove(tmp)
                if check:
                    raise
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def check(rc):
    if rc != 0:
        raise RuntimeError("mst_engine: " + lib().mst_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
