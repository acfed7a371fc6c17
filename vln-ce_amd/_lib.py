"""ctypes binding of libvlnce_hip.so (C ABI in include/vlnce_hip.h).

`HipLib` exposes one method per C entry point, taking torch tensors where the
C function takes device pointers (None -> NULL) and launching on torch's
current HIP stream.  There is NO fallback: if the shared library is missing
or a tensor is not on a GPU the call raises.  (tests/hostsim.py swaps in a
tensor-level simulator of this class to exercise the *host* logic on CPU; it
lives under tests/ and is never importable from the product.)
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLNCE_HIP_LIB: load another build of the library (bisection / tuning builds, scripts/build_variants.sh)
LIB_PATH = os.environ.get("VLNCE_HIP_LIB") or os.path.join(_HERE, "libvlnce_hip.so")

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_long


def _buffer_format(explicit, *bufs):
    """plane format of a launch's weight buffers: the tag ops.split_weights / pack_weights /
    stem7_pack_weights leave on them (`_vlnce_fmt`); an explicit value that contradicts a tag -- the
    kernel would read fp16 planes as bf16 or the reverse, silently -- raises."""
    tags = {getattr(b, "_vlnce_fmt", None) for b in bufs if b is not None} - {None}
    if explicit:
        tags.add(int(explicit))
    if len(tags) > 1:
        raise RuntimeError(f"weight buffers / w_format disagree on the plane format: {sorted(tags)}")
    return tags.pop() if tags else 1


class ConvDesc(C.Structure):
    _fields_ = [(n, _I) for n in
                ("N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "Ho", "Wo", "ldx", "ldy")]


class WeightJob(C.Structure):   # vlnce_weight_job
    _fields_ = [("w_oihw", _P), ("dst", _P), ("Cout", _I), ("Cin", _I), ("T", _I), ("kind", _I),
                ("transposed", _I), ("format", _I)]


class WeightPrepPlan:
    """the device-side job table of one vlnce_conv2d_prepare_weights launch (+ the tensors it
    points into, kept alive)"""
    __slots__ = ("table", "first_item", "njobs", "total", "keep")


class Prologue(C.Structure):
    _fields_ = [("in_scale", _P), ("in_shift", _P), ("in_center", _P), ("in_relu", _I),
                ("x2", _P), ("in2_scale", _P), ("in2_shift", _P), ("in2_center", _P),
                ("side_out", _P), ("w_split", _P), ("w_frag", _P), ("options", _P), ("w_format", _I)]


class Frames(C.Structure):
    _fields_ = [("x", _P), ("x2", _P), ("mask2", _P), ("dtype", _I), ("N", _I), ("F", _I),
                ("Hs", _I), ("Ws", _I), ("C", _I), ("y0", _I), ("x0", _I), ("H", _I), ("W", _I)]


class BnSums(C.Structure):
    _fields_ = [("acc", _P), ("workspace", _P), ("workspace_bytes", _L)]


class Epilogue(C.Structure):
    _fields_ = [("scale", _P), ("shift", _P), ("residual", _P), ("ldr", _I), ("act", _I),
                ("accumulate", _I), ("stat_partial", _P), ("bn", C.POINTER(BnSums))]


_SIGNATURES = {
    "vlnce_version": (_I, []),
    "vlnce_last_error": (C.c_char_p, []),
    "vlnce_set_option": (_I, [C.c_char_p, _I]),
    "vlnce_get_option": (_I, [C.c_char_p, C.POINTER(_I)]),
    "vlnce_option_default": (_I, [C.c_char_p, C.POINTER(_I)]),
    "vlnce_conv2d_split_weights": (_I, [_P, _P, C.c_long, _I, _P]),
    "vlnce_conv2d_last_path": (_I, []),
    "vlnce_embedding_bwd": (_I, [_P, _P, _P, _L, _I, _L, _L, _P]),
    "vlnce_conv2d_pack_bytes": (C.c_long, [C.POINTER(ConvDesc)]),
    "vlnce_conv2d_pack_weights": (_I, [_P, _P, C.POINTER(ConvDesc), _I, _P]),
    "vlnce_weight_job_items": (C.c_long, [C.POINTER(WeightJob)]),
    "vlnce_conv2d_prepare_weights": (_I, [_P, _P, _I, C.c_long, _P]),
    "vlnce_conv2d_tiles_m": (_I, [C.POINTER(ConvDesc)]),
    "vlnce_conv2d_tile_rows": (_I, [C.POINTER(ConvDesc)]),
    "vlnce_conv2d_bn_workspace_bytes": (C.c_long, [C.POINTER(ConvDesc)]),
    "vlnce_bn_finalize_sums": (_I, [_P, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "vlnce_conv2d_fwd": (_I, [_P, _P, _P, C.POINTER(ConvDesc), C.POINTER(Prologue),
                              C.POINTER(Epilogue), _P]),
    "vlnce_gemm": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, C.POINTER(Epilogue), _P]),
    "vlnce_colsum": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "vlnce_bn_finalize_workspace_bytes": (C.c_size_t, [_I, _I]),
    "vlnce_bn_finalize": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P,
                               C.c_size_t, _P]),
    "vlnce_scale_shift_act": (_I, [_P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P]),
    "vlnce_gn_chunks": (_I, [_I]),
    "vlnce_gn_partial": (_I, [_P, _I, _I, _I, _P, _P]),
    "vlnce_gn_finalize": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _P]),
    "vlnce_gn_finalize_tiles": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _P]),
    "vlnce_maxpool3x3s2": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "vlnce_scale_shift_add_act": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "vlnce_avgpool2x2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "vlnce_ragged_pad_rows": (_I, [_P, _I, _P, _I, _I, _L, _F, _P, _P]),
    "vlnce_ragged_pad_rows_i64": (_I, [_P, _P, _I, _I, _L, _L, _P, _P]),
    "vlnce_dagger_targets": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "vlnce_ppo_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _F, _F, _I,
                            _P, _P, _P]),
    "vlnce_ppo_returns": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _P]),
    "vlnce_space_to_depth2": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "vlnce_frames_s2d": (_I, [C.POINTER(Frames), _P, _I, _I, _P, _P, _P]),
    "vlnce_frames_avgpool2": (_I, [C.POINTER(Frames), _P, _P]),
    "vlnce_stem7_fwd": (_I, [C.POINTER(Frames), _P, _P, _P, _I, _P, _I, C.POINTER(Epilogue), _P]),
    "vlnce_frames_f32": (_I, [C.POINTER(Frames), _P, _P, _P, _P]),
    "vlnce_frames_gather": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vlnce_frames_resize_area": (_I, [_P] + [_I] * 11 + [_P, _P]),
    "vlnce_adaptive_avgpool": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vlnce_attn_fwd": (_I, [_P, _P, _I, _P, _I, _P, _I, _F, _P, _P, _I, _I, _I, _I, _P]),
    "vlnce_attn_bwd": (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _F, _P, _P, _P, _I, _P, _I,
                            _I, _I, _I, _I, _P]),
    "vlnce_rowzero_mask": (_I, [_P, _I, _L, _I, _P, _P]),
    "vlnce_attn_fwd_shared": (_I, [_P, _P, _I, _P, _I, _P, _I, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vlnce_attn_bwd_shared": (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _F, _P, _P, _P, _P, _I, _P, _I,
                                   _I, _I, _I, _I, _P]),
    "vlnce_segment_sum": (_I, [_P, _P, _I, _I, _L, _P, _P]),
    "vlnce_action_head_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "vlnce_action_head_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "vlnce_gru_gates_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vlnce_gru_gates_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vlnce_lstm_gates_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vlnce_lstm_gates_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vlnce_mean_rows": (_I, [_P, _P, _I, _I, _I, _P]),
    "vlnce_mask_rows": (_I, [_P, _P, _P, _I, _I, _P]),
    "vlnce_conv2d_wgrad": (_I, [_P, _P, _P, C.POINTER(ConvDesc), _P, _I, _I, _P]),
    "vlnce_bn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "vlnce_bn_bwd_workspace_floats": (C.c_size_t, [_L, _I]),
    "vlnce_gn_bwd_workspace_floats": (C.c_size_t, [_I, _I, _I, _I]),
    "vlnce_gn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "vlnce_maxpool3x3s2_argmax": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vlnce_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vlnce_adaptive_avgpool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vlnce_rnn_seq_supported": (_I, [_I, _I]),
    "vlnce_rnn_seq_fwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_rnn_seq_bwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_option_count": (_I, []),
    "vlnce_option_index": (_I, [C.c_char_p]),
    "vlnce_linear_rows_supported": (_I, [_I, _I, _I]),
    "vlnce_linear_rows_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "vlnce_linear_rows_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_rnn_seq_fwd2": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_rnn_seq_bwd2": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_rnn_seq_wgrad": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_group_norm_small": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _P, _I, _P, _P]),
    "vlnce_gru_rollout_supported": (_I, [_I, _I]),
    "vlnce_gru_rollout_workspace_bytes": (C.c_long, [_I, _I]),
    "vlnce_gru_rollout_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_gru_rollout_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vlnce_rnn_step_supported": (_I, [_I, _I, _I]),
    "vlnce_rnn_step_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vlnce_rnn_step_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vlnce_select_rows": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "vlnce_act_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def load_cdll(path=LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "The VLN-CE policy path has no CPU fallback."
        )
    dll = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(dll, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return dll


class _CallState(threading.local):
    """device of the tensors of the call being assembled + the device to switch back to"""
    dev = None
    restore = None


_CALL = _CallState()


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libvlnce_hip: tensor is not on a GPU (no CPU fallback)")
    idx = t.device.index
    if _CALL.dev is None:
        _CALL.dev = idx
    elif _CALL.dev != idx:
        first, _CALL.dev = _CALL.dev, None
        raise RuntimeError(f"libvlnce_hip: tensors of one call live on cuda:{first} and "
                           f"cuda:{idx}")
    return t.data_ptr()


def _stream():
    """HIP stream for the call whose pointer arguments were just converted by _ptr(): torch's
    current stream OF THE TENSORS' DEVICE.  The reference trainers put the policy on
    torch.device("cuda", TORCH_GPU_ID) without ever calling torch.cuda.set_device
    (base_il_trainer.py:58-66), so the process's current device need not be the policy's; a
    kernel must be launched with its own device current, so the device is switched here and
    switched back by HipLib._check right after the launch."""
    dev, _CALL.dev = _CALL.dev, None
    if dev is None:
        return torch.cuda.current_stream().cuda_stream
    cur = torch.cuda.current_device()
    if dev != cur:
        _CALL.restore = cur
        torch.cuda.set_device(dev)
    return torch.cuda.current_stream(dev).cuda_stream


def _leave_device():
    if _CALL.restore is not None:
        cur, _CALL.restore = _CALL.restore, None
        torch.cuda.set_device(cur)


class HipLib:
    """Tensor-level view of the C ABI."""

    name = "hip"

    ABI = 143  # include/vlnce_hip.h

    def __init__(self, path=LIB_PATH):
        self.dll = load_cdll(path)
        have = int(self.dll.vlnce_version())
        if have != self.ABI:  # struct layouts / signatures moved: a stale .so would corrupt memory
            raise RuntimeError(f"{path} has ABI {have}, this package binds ABI {self.ABI}: "
                               "rebuild it (python __graft_entry__.py)")

        self._conv_math = None
        #  the `with lib.options(...)` scope is per THREAD, like the library's own
        # thread-local per-launch options: autograd worker threads and side-stream helpers must not
        # see, or restore, another thread's overrides
        self._tls = threading.local()
        self._options_from_env()

    # ---- dispatch options (vlnce_set_option): the library itself never reads the environment;
    # the VLNCE_* variables of INTEGRATION.md section 8 are translated here, once, at load
    OPTION_NAMES = ("conv_math", "p3", "p3_tile", "s3", "u3", "u3_waves", "x3_tile", "igemm_tile",
                    "igemm_nobuf", "igemm_no_splitk", "wgrad_tile", "rollout_one_xcd", "m3")

    def _options_from_env(self):
        for name in self.OPTION_NAMES:
            v = os.environ.get("VLNCE_" + name.upper())
            if v is None:
                continue
            if name == "conv_math":   # f32 | 0: fp32 MFMA; bf16 | 1: three bf16 planes; f16 | 2: fp16 planes
                v = {"f32": 0, "0": 0, "bf16": 1, "1": 1, "f16": 2, "2": 2}.get(v.strip().lower(), v)
            elif name in ("igemm_nobuf", "igemm_no_splitk"):
                v = 0 if v in ("", "0") else 1
            try:
                v = int(v)
            except ValueError:   #  a malformed tuning variable must not break the load
                import warnings
                warnings.warn(f"VLNCE_{name.upper()}={v!r} is not an integer: ignored "
                              f"(the option keeps its default)")
                continue
            self.set_option(name, v)

    def set_option(self, name, value):
        self._check(self.dll.vlnce_set_option(name.encode(), int(value)), "vlnce_set_option")
        if name == "conv_math":
            self._conv_math = int(value)

    def plane_format(self, options=None):
        """the plane format (1 = three bf16 planes, 2 = fp16 planes; include/vlnce_hip.h) in which the
        weights of a launch are packed = the effective "conv_math" of that launch: its own options,
        else the enclosing `with lib.options(...)`, else the process value (0 = fp32 MFMA: the planes
        are not read; packed as 1)."""
        v = None
        if options:
            v = options.get("conv_math")
        if v is None:
            sc = getattr(self._tls, "scoped", None)
            if sc:
                v = sc.get("conv_math")
        if v is None or v < 0:
            v = self._conv_math
            if v is None:
                v = self._conv_math = self.get_option("conv_math")
        return 2 if v == 2 else 1

    def get_option(self, name):
        v = _I(0)
        self._check(self.dll.vlnce_get_option(name.encode(), C.byref(v)), "vlnce_get_option")
        return v.value

    def option_default(self, name):
        v = _I(0)
        self._check(self.dll.vlnce_option_default(name.encode(), C.byref(v)), "vlnce_option_default")
        return v.value

    # options that choose among the CONVOLUTION kernels travel with each launch
    # (vlnce_prologue.options); the diagnostic switches of the GEMM / rollout entry points, which
    # take no prologue, stay process values
    PER_LAUNCH = ("conv_math", "p3", "p3_tile", "s3", "u3", "u3_waves", "x3_tile", "m3")

    def _launch_options(self, overrides):
        """ctypes int array for vlnce_prologue.options from {name: value} (None = no overrides)"""
        if not overrides:
            return None
        arr = (C.c_int * int(self.dll.vlnce_option_count()))(*([-1] * int(self.dll.vlnce_option_count())))
        for name, v in overrides.items():
            i = int(self.dll.vlnce_option_index(name.encode()))
            if i < 0:
                raise RuntimeError(f"unknown dispatch option {name!r}")
            arr[i] = int(v)
        return arr

    def options(self, **kw):
        """`with lib.options(u3=2, s3=0): ...` -- convolution-kernel options (PER_LAUNCH) are handed
        to every conv2d_fwd of the block through its prologue: the library's process state is not
        touched; the GEMM / rollout diagnostics are set, then restored."""
        lib = self
        unknown = [k for k in kw if k not in self.OPTION_NAMES]
        if unknown:
            raise RuntimeError(f"unknown dispatch option(s) {unknown}")

        class _Scope:
            def __enter__(self_inner):
                self_inner.prev = getattr(lib._tls, "scoped", None)
                scoped = dict(self_inner.prev or {})
                scoped.update({k: v for k, v in kw.items() if k in lib.PER_LAUNCH})
                lib._tls.scoped = scoped or None
                self_inner.old = {k: lib.get_option(k) for k in kw if k not in lib.PER_LAUNCH}
                for k in self_inner.old:
                    lib.set_option(k, kw[k])
                return lib

            def __exit__(self_inner, *exc):
                lib._tls.scoped = self_inner.prev
                for k, v in self_inner.old.items():
                    lib.set_option(k, v)
                return False

        return _Scope()

    def _check(self, rc, what):
        _leave_device()
        if rc != 0:
            raise RuntimeError(f"{what} failed (rc={rc}): {self.dll.vlnce_last_error().decode()}")

    # ---- conv / gemm
    @staticmethod
    def _desc(g):
        return ConvDesc(*[int(g[k]) for k, _ in ConvDesc._fields_])

    def conv2d_tiles(self, g):
        d = self._desc(g)
        return self.dll.vlnce_conv2d_tiles_m(C.byref(d)), self.dll.vlnce_conv2d_tile_rows(C.byref(d))

    def conv2d_fwd(self, x, w, y, g, in_scale=None, in_shift=None, in_relu=0, scale=None,
                   shift=None, residual=None, ldr=0, act=0, accumulate=0, stat_partial=None,
                   in_center=None, x2=None, in2_scale=None, in2_shift=None, in2_center=None,
                   side_out=None, w_split=None, w_frag=None, bn=None, options=None, w_format=None):
        """bn: train-mode BatchNorm statistics added by the launch (vlnce_bn_sums): (acc
        [16, C, 2] f64, workspace uint8) -- finish them with bn_finalize_sums().
        options: {name: value} dispatch options of this launch (default: those of the enclosing
        `with lib.options(...)` block, else the process values).
        w_format: plane format of w_split / w_frag; default: the format the buffers were made with
        (ops.split_weights / pack_weights tag them), which any explicit value must agree with."""
        d = self._desc(g)
        w_format = _buffer_format(w_format, w_split, w_frag)
        opts = self._launch_options(options if options is not None
                                    else getattr(self._tls, "scoped", None))
        pro = Prologue(_ptr(in_scale), _ptr(in_shift), _ptr(in_center), int(in_relu), _ptr(x2),
                       _ptr(in2_scale), _ptr(in2_shift), _ptr(in2_center), _ptr(side_out),
                       _ptr(w_split), _ptr(w_frag),
                       C.cast(opts, C.c_void_p) if opts is not None else None, int(w_format))
        bnp = None
        if bn is not None:
            acc, ws = bn
            bnp = C.pointer(BnSums(_ptr(acc), _ptr(ws), ws.numel() * ws.element_size()))
        epi = Epilogue(_ptr(scale), _ptr(shift), _ptr(residual), int(ldr), int(act),
                       int(accumulate), _ptr(stat_partial), bnp)
        self._check(self.dll.vlnce_conv2d_fwd(_ptr(x), _ptr(w), _ptr(y), C.byref(d), C.byref(pro),
                                              C.byref(epi), _stream()), "vlnce_conv2d_fwd")

    def bn_finalize_sums(self, acc, M, gamma, beta, eps, momentum, running_mean, running_var,
                         scale_out, mean_out, shift_out=None, rstd_out=None):
        Cc = scale_out.numel()
        self._check(self.dll.vlnce_bn_finalize_sums(
            _ptr(acc), int(M), Cc, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
            _ptr(running_mean), _ptr(running_var), _ptr(scale_out), _ptr(shift_out),
            _ptr(mean_out), _ptr(rstd_out), _stream()), "vlnce_bn_finalize_sums")

    def conv2d_bn_workspace_bytes(self, g):
        d = self._desc(g)
        return int(self.dll.vlnce_conv2d_bn_workspace_bytes(C.byref(d)))

    def conv2d_split_weights(self, w, planes, fmt=1):
        self._check(self.dll.vlnce_conv2d_split_weights(_ptr(w), _ptr(planes), w.numel(), int(fmt),
                                                        _stream()), "vlnce_conv2d_split_weights")

    def conv2d_last_path(self):
        return int(self.dll.vlnce_conv2d_last_path())

    def conv2d_pack_bytes(self, g):
        d = self._desc(g)
        return int(self.dll.vlnce_conv2d_pack_bytes(C.byref(d)))

    def conv2d_pack_weights(self, w, frag, g, fmt=1):
        d = self._desc(g)
        self._check(self.dll.vlnce_conv2d_pack_weights(_ptr(w), _ptr(frag), C.byref(d), int(fmt),
                                                       _stream()), "vlnce_conv2d_pack_weights")

    WP_F32, WP_PLANES, WP_FRAGMENTS = 0, 1, 2

    def weight_prep_plan(self, jobs):
        """jobs: [(w_oihw parameter tensor, dst tensor, kind, transposed, format)] -> WeightPrepPlan
        (the job table uploaded once; raises for a job the kernel does not take)"""
        arr = (WeightJob * len(jobs))()
        first = [0]
        for j, (w, dst, kind, transposed, fmt) in zip(arr, jobs):
            assert w.is_cuda and w.is_contiguous() and w.dtype == torch.float32 and w.dim() == 4
            j.w_oihw, j.dst = w.data_ptr(), dst.data_ptr()
            j.Cout, j.Cin, j.T = w.size(0), w.size(1), w.size(2) * w.size(3)
            j.kind, j.transposed, j.format = int(kind), int(bool(transposed)), int(fmt or 0)
            items = int(self.dll.vlnce_weight_job_items(C.byref(j)))
            if items <= 0:
                raise ValueError(f"weight_prep_plan: job not eligible: {tuple(w.shape)} kind={kind} "
                                 f"transposed={transposed} format={fmt}")
            first.append(first[-1] + items)
        plan = WeightPrepPlan()
        dev = jobs[0][1].device
        plan.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        plan.first_item = torch.tensor(first, dtype=torch.int64).to(dev)
        plan.njobs, plan.total = len(jobs), first[-1]
        plan.keep = [(w, dst) for w, dst, *_ in jobs]
        return plan

    def conv2d_prepare_weights(self, plan):
        self._check(self.dll.vlnce_conv2d_prepare_weights(_ptr(plan.table), _ptr(plan.first_item),
                                                          plan.njobs, plan.total, _stream()),
                    "vlnce_conv2d_prepare_weights")

    def gemm(self, A, lda, transA, B, ldb, transB, Cm, ldc, M, N, K, scale=None, shift=None,
             residual=None, ldr=0, act=0, accumulate=0):
        epi = Epilogue(_ptr(scale), _ptr(shift), _ptr(residual), int(ldr), int(act),
                       int(accumulate), None)
        self._check(self.dll.vlnce_gemm(_ptr(A), lda, transA, _ptr(B), ldb, transB, _ptr(Cm), ldc,
                                        M, N, K, C.byref(epi), _stream()), "vlnce_gemm")

    def colsum(self, x, ldx, M, N, out, accumulate=0):
        self._check(self.dll.vlnce_colsum(_ptr(x), ldx, M, N, _ptr(out), accumulate, _stream()),
                    "vlnce_colsum")

    # ---- norms
    def bn_finalize_workspace_bytes(self, tiles_m, Cc):
        return int(self.dll.vlnce_bn_finalize_workspace_bytes(tiles_m, Cc))

    def bn_finalize(self, partial, tiles_m, tile_rows, M, Cc, gamma, beta, eps, momentum,
                    running_mean, running_var, scale_out, shift_out, mean_out=None, rstd_out=None,
                    workspace=None):
        wb = workspace.numel() * workspace.element_size() if workspace is not None else 0
        self._check(self.dll.vlnce_bn_finalize(
            _ptr(partial), tiles_m, tile_rows, M, Cc, _ptr(gamma), _ptr(beta), eps, momentum,
            _ptr(running_mean), _ptr(running_var), _ptr(scale_out), _ptr(shift_out),
            _ptr(mean_out), _ptr(rstd_out), _ptr(workspace), wb, _stream()), "vlnce_bn_finalize")

    def scale_shift_act(self, x, scale, shift, rows_per_sample, residual, y, M, Cc, act,
                        center=None):
        self._check(self.dll.vlnce_scale_shift_act(
            _ptr(x), _ptr(scale), _ptr(shift), _ptr(center), rows_per_sample, _ptr(residual),
            _ptr(y), M, Cc, act, _stream()), "vlnce_scale_shift_act")

    def gn_chunks(self, HW):
        return self.dll.vlnce_gn_chunks(HW)

    def gn_partial(self, x, Nimg, HW, Cc, partial):
        self._check(self.dll.vlnce_gn_partial(_ptr(x), Nimg, HW, Cc, _ptr(partial), _stream()),
                    "vlnce_gn_partial")

    def gn_finalize(self, partial, Nimg, HW, Cc, groups, gamma, beta, eps, scale_out, shift_out,
                    mean_out=None, rstd_out=None, center_out=None):
        self._check(self.dll.vlnce_gn_finalize(
            _ptr(partial), Nimg, HW, Cc, groups, _ptr(gamma), _ptr(beta), eps, _ptr(scale_out),
            _ptr(shift_out), _ptr(center_out), _ptr(mean_out), _ptr(rstd_out), _stream()),
            "vlnce_gn_finalize")

    # ---- pools
    def gn_finalize_tiles(self, partial, tile_rows, Nimg, HW, Cc, groups, gamma, beta, eps,
                          scale_out, shift_out, mean_out=None, rstd_out=None, center_out=None):
        self._check(self.dll.vlnce_gn_finalize_tiles(
            _ptr(partial), tile_rows, Nimg, HW, Cc, groups, _ptr(gamma), _ptr(beta), eps,
            _ptr(scale_out), _ptr(shift_out), _ptr(center_out), _ptr(mean_out), _ptr(rstd_out),
            _stream()), "vlnce_gn_finalize_tiles")

    def maxpool3x3s2(self, x, y, N, H, W, Cc, Ho, Wo, in_scale=None, in_shift=None, in_relu=0,
                     in_center=None):
        self._check(self.dll.vlnce_maxpool3x3s2(_ptr(x), _ptr(y), N, H, W, Cc, Ho, Wo,
                                                _ptr(in_scale), _ptr(in_shift), _ptr(in_center),
                                                int(in_relu), _stream()), "vlnce_maxpool3x3s2")

    def scale_shift_add_act(self, x1, s1, t1, x2, s2, t2, y, M, Cc, act, c1=None, c2=None):
        self._check(self.dll.vlnce_scale_shift_add_act(
            _ptr(x1), _ptr(s1), _ptr(t1), _ptr(c1), _ptr(x2), _ptr(s2), _ptr(t2), _ptr(c2),
            _ptr(y), M, Cc, act, _stream()), "vlnce_scale_shift_add_act")

    # ---- cached-feature DAgger data path
    _SRC = {torch.float32: 0, torch.float16: 1, torch.int64: 2}

    def ragged_pad_rows(self, src, offsets, B, Tmax, D, fill, dst):
        assert offsets.dtype == torch.int32 and dst.dtype == torch.float32
        self._check(self.dll.vlnce_ragged_pad_rows(
            _ptr(src), self._SRC[src.dtype], _ptr(offsets), B, Tmax, D, float(fill),
            _ptr(dst), _stream()), "vlnce_ragged_pad_rows")

    def ragged_pad_rows_i64(self, src, offsets, B, Tmax, D, fill, dst):
        assert src.dtype == torch.int64 and dst.dtype == torch.int64
        self._check(self.dll.vlnce_ragged_pad_rows_i64(
            _ptr(src), _ptr(offsets), B, Tmax, D, int(fill), _ptr(dst), _stream()),
            "vlnce_ragged_pad_rows_i64")

    def dagger_targets(self, oracle, offsets, B, Tmax, coef, corrected, weights, masks):
        self._check(self.dll.vlnce_dagger_targets(
            _ptr(oracle), _ptr(offsets), B, Tmax, float(coef), _ptr(corrected),
            _ptr(weights), _ptr(masks), _stream()), "vlnce_dagger_targets")

    def ppo_loss(self, values, returns, value_preds, logp, old_logp, adv, ent_pano, ent_offset,
                 ent_distance, radians, B, clip, value_coef, entropy_coef, pano_coef, offset_coef,
                 distance_coef, reg_coef, use_clipped, stats, grads):
        self._check(self.dll.vlnce_ppo_loss(
            _ptr(values), _ptr(returns), _ptr(value_preds), _ptr(logp), _ptr(old_logp), _ptr(adv),
            _ptr(ent_pano), _ptr(ent_offset), _ptr(ent_distance), _ptr(radians), B, clip, value_coef,
            entropy_coef, pano_coef, offset_coef, distance_coef, reg_coef, int(use_clipped),
            _ptr(stats), _ptr(grads), _stream()), "vlnce_ppo_loss")

    def ppo_returns(self, rewards, value_preds, masks, next_value, returns, T, N, gamma, tau,
                    use_gae):
        self._check(self.dll.vlnce_ppo_returns(_ptr(rewards), _ptr(value_preds), _ptr(masks),
                                               _ptr(next_value), _ptr(returns), T, N, float(gamma),
                                               float(tau), int(use_gae), _stream()),
                    "vlnce_ppo_returns")

    def space_to_depth2(self, x, y, N, H, W, Cc, pad_lo, pad_hi, scale=None, shift=None):
        self._check(self.dll.vlnce_space_to_depth2(_ptr(x), _ptr(y), N, H, W, Cc, pad_lo, pad_hi,
                                                   _ptr(scale), _ptr(shift), _stream()),
                    "vlnce_space_to_depth2")

    # ---- observation ingest
    _DT = {torch.float32: 0, torch.uint8: 1}

    def _frames(self, fr):
        """fr: dict(x, x2, mask2, N, F, Hs, Ws, C, y0, x0, H, W) with tensors for x / x2 / mask2"""
        x = fr["x"]
        if fr.get("x2") is not None and fr["x2"].dtype != x.dtype:
            raise RuntimeError("libvlnce_hip: frames and the extra frame differ in dtype")
        return Frames(_ptr(x), _ptr(fr.get("x2")), _ptr(fr.get("mask2")), self._DT[x.dtype],
                      *[int(fr[k]) for k in ("N", "F", "Hs", "Ws", "C", "y0", "x0", "H", "W")])

    def frames_s2d(self, fr, y, pad_lo, pad_hi, scale=None, shift=None):
        d = self._frames(fr)
        self._check(self.dll.vlnce_frames_s2d(C.byref(d), _ptr(y), pad_lo, pad_hi, _ptr(scale),
                                              _ptr(shift), _stream()), "vlnce_frames_s2d")

    def stem7_fwd(self, fr, in_scale, in_shift, w_frag, y, scale=None, shift=None, act=0, bn=None,
                  w_format=None):
        """RGB stem (7x7 / stride 2 / pad 3) straight from the frame descriptor; epilogue =
        scale / shift / act, or bn = the column-sum accumulator of vlnce_bn_sums."""
        d = self._frames(fr)
        w_format = _buffer_format(w_format, w_frag)
        bnp = None
        if bn is not None:
            bnp = C.pointer(BnSums(_ptr(bn), None, 0))
        epi = Epilogue(_ptr(scale), _ptr(shift), None, 0, int(act), 0, None, bnp)
        self._check(self.dll.vlnce_stem7_fwd(C.byref(d), _ptr(in_scale), _ptr(in_shift),
                                             _ptr(w_frag), int(w_format), _ptr(y), y.size(-1),
                                             C.byref(epi), _stream()), "vlnce_stem7_fwd")

    def frames_avgpool2(self, fr, y):
        d = self._frames(fr)
        self._check(self.dll.vlnce_frames_avgpool2(C.byref(d), _ptr(y), _stream()),
                    "vlnce_frames_avgpool2")

    def frames_f32(self, fr, y, scale=None, shift=None):
        d = self._frames(fr)
        self._check(self.dll.vlnce_frames_f32(C.byref(d), _ptr(y), _ptr(scale), _ptr(shift),
                                              _stream()), "vlnce_frames_f32")

    def frames_gather(self, srcs, elem_bytes, N, Hs, Ws, Cc, y0, x0, H, W, out):
        arr = (C.c_void_p * len(srcs))(*[_ptr(t) for t in srcs])
        self._check(self.dll.vlnce_frames_gather(arr, len(srcs), elem_bytes, N, Hs, Ws, Cc, y0, x0,
                                                 H, W, _ptr(out), _stream()), "vlnce_frames_gather")

    def frames_resize_area(self, x, is_u8, NF, Hs, Ws, Cc, OH, OW, y0, x0, H, W, out):
        self._check(self.dll.vlnce_frames_resize_area(_ptr(x), 1 if is_u8 else 0, NF, Hs, Ws, Cc, OH,
                                                      OW, y0, x0, H, W, _ptr(out), _stream()),
                    "vlnce_frames_resize_area")

    def avgpool2x2(self, x, y, N, H, W, Cc):
        self._check(self.dll.vlnce_avgpool2x2(_ptr(x), _ptr(y), N, H, W, Cc, _stream()),
                    "vlnce_avgpool2x2")

    def adaptive_avgpool(self, x, y, N, H, W, Cc, OH, OW, ldy):
        self._check(self.dll.vlnce_adaptive_avgpool(_ptr(x), _ptr(y), N, H, W, Cc, OH, OW, ldy,
                                                    _stream()), "vlnce_adaptive_avgpool")

    def mean_rows(self, x, y, B, P, Cc):
        self._check(self.dll.vlnce_mean_rows(_ptr(x), _ptr(y), B, P, Cc, _stream()),
                    "vlnce_mean_rows")

    # ---- categorical action head
    def action_head_fwd(self, x, ldx, w, b, M, K, A, logits_out, nan_count=None):
        self._check(self.dll.vlnce_action_head_fwd(_ptr(x), ldx, _ptr(w), _ptr(b), M, K, A,
                                                   _ptr(logits_out), _ptr(nan_count), _stream()),
                    "vlnce_action_head_fwd")

    def action_head_bwd(self, x, ldx, w, logits, dlogits, M, K, A, dx=None, dw=None, db=None):
        self._check(self.dll.vlnce_action_head_bwd(_ptr(x), ldx, _ptr(w), _ptr(logits),
                                                   _ptr(dlogits), M, K, A, _ptr(dx), _ptr(dw),
                                                   _ptr(db), _stream()), "vlnce_action_head_bwd")

    # ---- attention
    def attn_fwd(self, q, K, ldk, V, ldv, mask, mask_mode, scale, out, attn_out, B, P, Dk, Dv,
                 kv_index=None):
        if kv_index is not None:
            self._check(self.dll.vlnce_attn_fwd_shared(
                _ptr(q), _ptr(K), ldk, _ptr(V), ldv, _ptr(mask), mask_mode, scale, _ptr(kv_index),
                _ptr(out), _ptr(attn_out), B, P, Dk, Dv, _stream()), "vlnce_attn_fwd_shared")
            return
        self._check(self.dll.vlnce_attn_fwd(_ptr(q), _ptr(K), ldk, _ptr(V), ldv, _ptr(mask),
                                            mask_mode, scale, _ptr(out), _ptr(attn_out), B, P, Dk,
                                            Dv, _stream()), "vlnce_attn_fwd")

    def segment_sum(self, x, index, B, U, row_elems, out):
        self._check(self.dll.vlnce_segment_sum(_ptr(x), _ptr(index), B, U, row_elems, _ptr(out),
                                               _stream()), "vlnce_segment_sum")

    def attn_bwd(self, dout, q, K, ldk, V, ldv, mask, mask_mode, scale, attn, dq, dK, lddk, dV,
                 lddv, B, P, Dk, Dv, kv_index=None):
        if kv_index is not None:
            self._check(self.dll.vlnce_attn_bwd_shared(
                _ptr(dout), _ptr(q), _ptr(K), ldk, _ptr(V), ldv, _ptr(mask), mask_mode, scale,
                _ptr(kv_index), _ptr(attn), _ptr(dq), _ptr(dK), lddk, _ptr(dV), lddv, B, P, Dk, Dv,
                _stream()), "vlnce_attn_bwd_shared")
            return
        self._check(self.dll.vlnce_attn_bwd(_ptr(dout), _ptr(q), _ptr(K), ldk, _ptr(V), ldv,
                                            _ptr(mask), mask_mode, scale, _ptr(attn), _ptr(dq),
                                            _ptr(dK), lddk, _ptr(dV), lddv, B, P, Dk, Dv,
                                            _stream()), "vlnce_attn_bwd")

    def rowzero_mask(self, x, ld, rows, Cc, mask):
        self._check(self.dll.vlnce_rowzero_mask(_ptr(x), ld, rows, Cc, _ptr(mask), _stream()),
                    "vlnce_rowzero_mask")

    # ---- recurrent cells
    def gru_gates_fwd(self, gi, gh, h_prev, mask, h_out, gates_out, hn_out, B, H):
        self._check(self.dll.vlnce_gru_gates_fwd(_ptr(gi), _ptr(gh), _ptr(h_prev), _ptr(mask),
                                                 _ptr(h_out), _ptr(gates_out), _ptr(hn_out), B, H,
                                                 _stream()), "vlnce_gru_gates_fwd")

    def gru_gates_bwd(self, dh_out, gates, hn, h_prev, mask, dgi, dgh, dh_prev, B, H):
        self._check(self.dll.vlnce_gru_gates_bwd(_ptr(dh_out), _ptr(gates), _ptr(hn), _ptr(h_prev),
                                                 _ptr(mask), _ptr(dgi), _ptr(dgh), _ptr(dh_prev),
                                                 B, H, _stream()), "vlnce_gru_gates_bwd")

    def lstm_gates_fwd(self, gi, gh, c_prev, mask, h_out, c_out, gates_out, B, H):
        self._check(self.dll.vlnce_lstm_gates_fwd(_ptr(gi), _ptr(gh), _ptr(c_prev), _ptr(mask),
                                                  _ptr(h_out), _ptr(c_out), _ptr(gates_out), B, H,
                                                  _stream()), "vlnce_lstm_gates_fwd")

    def lstm_gates_bwd(self, dh_out, dc_out, gates, c_prev, c_out, mask, dgates, dc_prev, B, H):
        self._check(self.dll.vlnce_lstm_gates_bwd(_ptr(dh_out), _ptr(dc_out), _ptr(gates),
                                                  _ptr(c_prev), _ptr(c_out), _ptr(mask),
                                                  _ptr(dgates), _ptr(dc_prev), B, H, _stream()),
                    "vlnce_lstm_gates_bwd")

    def mask_rows(self, x, mask, out, B, H):
        self._check(self.dll.vlnce_mask_rows(_ptr(x), _ptr(mask), _ptr(out), B, H, _stream()),
                    "vlnce_mask_rows")

    # ---- backward of the visual trunks
    def conv2d_wgrad(self, x, dy, dw, g, dy_pow2=None, accumulate=False):
        """dy_pow2: the [2, P] power-of-two buffer bn_bwd / gn_bwd filled for this dy, or None;
        accumulate: dw += (the caller zeroed it)"""
        d = self._desc(g)
        self._check(self.dll.vlnce_conv2d_wgrad(_ptr(x), _ptr(dy), _ptr(dw), C.byref(d),
                                                _ptr(dy_pow2),
                                                dy_pow2.size(1) if dy_pow2 is not None else 0,
                                                int(bool(accumulate)), _stream()),
                    "vlnce_conv2d_wgrad")

    def bn_bwd_workspace_floats(self, M, Cc):
        return int(self.dll.vlnce_bn_bwd_workspace_floats(M, Cc))

    def bn_bwd(self, dy, y, x, mean, rstd, gamma, M, Cc, relu, use_batch_stats, dx, dres, dgamma,
               dbeta, workspace=None, pow2=None):
        """pow2: [2, P] floats or None -- on return P copies of 2^k, then P of 2^-k, the power of
        two at which dx fits the fp16 planes (include/vlnce_hip.h)."""
        self._check(self.dll.vlnce_bn_bwd(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd),
                                          _ptr(gamma), M, Cc, int(relu), int(use_batch_stats),
                                          _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta),
                                          _ptr(workspace), _ptr(pow2),
                                          pow2.size(1) if pow2 is not None else 0, _stream()),
                    "vlnce_bn_bwd")

    def gn_bwd_workspace_floats(self, Nimg, HW, Cc, groups):
        return int(self.dll.vlnce_gn_bwd_workspace_floats(Nimg, HW, Cc, groups))

    def gn_bwd(self, dy, y, x, mean, rstd, gamma, Nimg, HW, Cc, groups, relu, dx, dres, dgamma,
               dbeta, workspace, pow2=None):
        self._check(self.dll.vlnce_gn_bwd(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd),
                                          _ptr(gamma), Nimg, HW, Cc, groups, int(relu), _ptr(dx),
                                          _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(workspace),
                                          _ptr(pow2), pow2.size(1) if pow2 is not None else 0,
                                          _stream()), "vlnce_gn_bwd")

    def maxpool3x3s2_argmax(self, x, y, argmax, N, H, W, Cc, Ho, Wo):
        self._check(self.dll.vlnce_maxpool3x3s2_argmax(_ptr(x), _ptr(y), _ptr(argmax), N, H, W, Cc,
                                                       Ho, Wo, _stream()),
                    "vlnce_maxpool3x3s2_argmax")

    def maxpool3x3s2_bwd(self, dy, argmax, dx, N, H, W, Cc, Ho, Wo):
        self._check(self.dll.vlnce_maxpool3x3s2_bwd(_ptr(dy), _ptr(argmax), _ptr(dx), N, H, W, Cc,
                                                    Ho, Wo, _stream()), "vlnce_maxpool3x3s2_bwd")

    def adaptive_avgpool_bwd(self, dy, dx, N, H, W, Cc, OH, OW):
        self._check(self.dll.vlnce_adaptive_avgpool_bwd(_ptr(dy), _ptr(dx), N, H, W, Cc, OH, OW,
                                                        _stream()), "vlnce_adaptive_avgpool_bwd")

    @staticmethod
    def _parr(ts, n):
        """host array of `n` device pointers (NULL array when ts is None)."""
        if ts is None:
            return None
        arr = (C.c_void_p * 2)()
        for i in range(n):
            arr[i] = _ptr(ts[i])
        return arr

    def rnn_seq_supported(self, kind, H):
        return bool(self.dll.vlnce_rnn_seq_supported(kind, H))

    def rnn_seq_fwd(self, kind, dirs, gi, w_hh, b_hh, lengths, out, h_final, gates_save, aux_save,
                    B, Lm, H):
        pa = self._parr
        self._check(self.dll.vlnce_rnn_seq_fwd(
            kind, dirs, pa(gi, dirs), pa(w_hh, dirs), pa(b_hh, dirs), _ptr(lengths), pa(out, dirs),
            pa(h_final, dirs), pa(gates_save, dirs), pa(aux_save, dirs), B, Lm, H, _stream()),
            "vlnce_rnn_seq_fwd")

    def rnn_seq_bwd(self, kind, dirs, w_hh_t, lengths, out, gates_save, aux_save, dout, dh_final,
                    dgi, dgh, B, Lm, H):
        pa = self._parr
        self._check(self.dll.vlnce_rnn_seq_bwd(
            kind, dirs, pa(w_hh_t, dirs), _ptr(lengths), pa(out, dirs), pa(gates_save, dirs),
            pa(aux_save, dirs), pa(dout, dirs), pa(dh_final, dirs), pa(dgi, dirs), pa(dgh, dirs),
            B, Lm, H, _stream()), "vlnce_rnn_seq_bwd")

    def linear_rows_supported(self, M, N, K):
        return bool(self.dll.vlnce_linear_rows_supported(M, N, K))

    def linear_rows_fwd(self, x, ldx, w, ldw, bias, act, y, ldy, M, N, K):
        self._check(self.dll.vlnce_linear_rows_fwd(_ptr(x), ldx, _ptr(w), ldw, _ptr(bias), act, _ptr(y),
                                                   ldy, M, N, K, _stream()), "vlnce_linear_rows_fwd")

    def linear_rows_bwd(self, x, ldx, w, ldw, dy, lddy, y, ldy, act, dx, dw, db, M, N, K):
        self._check(self.dll.vlnce_linear_rows_bwd(_ptr(x), ldx, _ptr(w), ldw, _ptr(dy), lddy, _ptr(y),
                                                   ldy, act, _ptr(dx), _ptr(dw), _ptr(db), M, N, K,
                                                   _stream()), "vlnce_linear_rows_bwd")

    def rnn_seq_fwd2(self, kind, dirs, gi, w_hh, b_hh, lengths, out_tm, seq, seq_st, seq_sb, h_final,
                     gates_save, aux_save, B, Lm, H):
        pa = self._parr
        self._check(self.dll.vlnce_rnn_seq_fwd2(
            kind, dirs, pa(gi, dirs), pa(w_hh, dirs), pa(b_hh, dirs), _ptr(lengths), pa(out_tm, dirs),
            _ptr(seq), seq_st, seq_sb, pa(h_final, dirs), pa(gates_save, dirs), pa(aux_save, dirs),
            B, Lm, H, _stream()), "vlnce_rnn_seq_fwd2")

    def rnn_seq_bwd2(self, kind, dirs, w_hh, lengths, out_tm, gates_save, aux_save, dseq, dseq_st,
                     dseq_sb, dout_ws, dh_final, dgi, dgh, B, Lm, H):
        pa = self._parr
        self._check(self.dll.vlnce_rnn_seq_bwd2(
            kind, dirs, pa(w_hh, dirs), _ptr(lengths), pa(out_tm, dirs), pa(gates_save, dirs),
            pa(aux_save, dirs), _ptr(dseq), dseq_st, dseq_sb, _ptr(dout_ws), pa(dh_final, dirs),
            pa(dgi, dirs), pa(dgh, dirs), B, Lm, H, _stream()), "vlnce_rnn_seq_bwd2")

    def rnn_seq_wgrad(self, kind, dirs, dgi, dgh, out_tm, x_tm, ldx, E, w_ih, dw_ih, dw_hh, db_ih,
                      db_hh, dx_tm, B, Lm, H, first_dir=0):
        pa = self._parr
        self._check(self.dll.vlnce_rnn_seq_wgrad(
            kind, dirs, first_dir, pa(dgi, dirs), pa(dgh, dirs), pa(out_tm, dirs), _ptr(x_tm), ldx, E,
            pa(w_ih, dirs), pa(dw_ih, dirs), pa(dw_hh, dirs), pa(db_ih, dirs), pa(db_hh, dirs),
            _ptr(dx_tm), B, Lm, H, _stream()), "vlnce_rnn_seq_wgrad")

    def group_norm_small(self, x, N, HW, Cc, groups, gamma, beta, eps, residual, act, y):
        self._check(self.dll.vlnce_group_norm_small(
            _ptr(x), N, HW, Cc, groups, _ptr(gamma), _ptr(beta), eps, _ptr(residual), act, _ptr(y),
            _stream()), "vlnce_group_norm_small")

    def gru_rollout_supported(self, N, H):
        if os.environ.get("VLNCE_GRU_ROLLOUT", "1") == "0":  # A/B switch (scripts/bench_data_path.py)
            return False
        return bool(self.dll.vlnce_gru_rollout_supported(N, H))

    def gru_rollout_workspace_bytes(self, N, H):
        return int(self.dll.vlnce_gru_rollout_workspace_bytes(N, H))

    def gru_rollout_fwd(self, gi, h0, mask, w_hh, b_hh, hp, out, gates, aux, workspace, T, N, H):
        self._check(self.dll.vlnce_gru_rollout_fwd(
            _ptr(gi), _ptr(h0), _ptr(mask), _ptr(w_hh), _ptr(b_hh), _ptr(hp), _ptr(out),
            _ptr(gates), _ptr(aux), _ptr(workspace), T, N, H, _stream()), "vlnce_gru_rollout_fwd")

    def gru_rollout_bwd(self, dout, dh_final, gates, aux, hp, mask, w_hh_t, dgi, dgh, dh0,
                        workspace, T, N, H):
        self._check(self.dll.vlnce_gru_rollout_bwd(
            _ptr(dout), _ptr(dh_final), _ptr(gates), _ptr(aux), _ptr(hp), _ptr(mask),
            _ptr(w_hh_t), _ptr(dgi), _ptr(dgh), _ptr(dh0), _ptr(workspace), T, N, H, _stream()),
            "vlnce_gru_rollout_bwd")

    def rnn_step_supported(self, N, H, lstm):
        if os.environ.get("VLNCE_RNN_STEP_FUSED", "1") == "0":  # A/B switch (scripts/bench_data_path.py)
            return False
        return bool(self.dll.vlnce_rnn_step_supported(N, H, int(lstm)))

    def rnn_step_fwd(self, lstm, gi, h_prev, c_prev, mask, w_hh, b_hh, hp_out, h_out, aux_out,
                     gates_out, N, H):
        self._check(self.dll.vlnce_rnn_step_fwd(
            int(lstm), _ptr(gi), _ptr(h_prev), _ptr(c_prev), _ptr(mask), _ptr(w_hh), _ptr(b_hh),
            _ptr(hp_out), _ptr(h_out), _ptr(aux_out), _ptr(gates_out), N, H, _stream()),
            "vlnce_rnn_step_fwd")

    def rnn_step_bwd(self, lstm, dout, carry, dc, gates, aux, hp, c_prev, mask, w_hh_t, dgi, dgh,
                     acc0, dc_prev, N, H):
        self._check(self.dll.vlnce_rnn_step_bwd(
            int(lstm), _ptr(dout), _ptr(carry), _ptr(dc), _ptr(gates), _ptr(aux), _ptr(hp),
            _ptr(c_prev), _ptr(mask), _ptr(w_hh_t), _ptr(dgi), _ptr(dgh), _ptr(acc0),
            _ptr(dc_prev), N, H, _stream()), "vlnce_rnn_step_bwd")

    def select_rows(self, mask, a, b, out, B, H):
        self._check(self.dll.vlnce_select_rows(_ptr(mask), _ptr(a), _ptr(b), _ptr(out), B, H,
                                               _stream()), "vlnce_select_rows")

    def embedding_bwd(self, tokens, grad_rows, grad_weight, padding_idx):
        self._check(self.dll.vlnce_embedding_bwd(_ptr(tokens), _ptr(grad_rows), _ptr(grad_weight),
                                                 tokens.numel(), grad_weight.size(1),
                                                 -1 if padding_idx is None else int(padding_idx),
                                                 grad_weight.size(0), _stream()),
                    "vlnce_embedding_bwd")

    def act_bwd(self, dy, y, dz, n, act):
        self._check(self.dll.vlnce_act_bwd(_ptr(dy), _ptr(y), _ptr(dz), n, act, _stream()),
                    "vlnce_act_bwd")


_LIB = None


def get_lib():
    global _LIB
    if _LIB is None:
        _LIB = HipLib()
    return _LIB
