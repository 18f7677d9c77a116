"""ctypes binding of libwavenet_hip.so (include/wavenet_hip.h).

The HIP library is THE compute path: there is no CPU or PyTorch fallback.
If the shared object is missing or a call fails, this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WN_LIB_PATH') or os.path.join(
    os.path.dirname(_HERE), 'libwavenet_hip.so')   # WN_LIB_PATH: A/B builds

c_int, c_long, c_float = ctypes.c_int, ctypes.c_long, ctypes.c_float
c_void_p, c_u64 = ctypes.c_void_p, ctypes.c_uint64
P = c_void_p  # every device / host pointer crosses as a raw address

# name -> (restype, argtypes); mirrors include/wavenet_hip.h one to one
SIGNATURES = {
    'wn_version': (c_int, []),
    'wn_error_string': (ctypes.c_char_p, [c_int]),
    'wn_mu_law_thresholds_host': (c_int, [c_int, P]),
    'wn_mu_law_decode_table_host': (c_int, [c_int, P]),
    'wn_mu_law_encode': (c_int, [P, P, c_long, P, c_int, P]),
    'wn_mu_law_decode': (c_int, [P, P, c_long, P, c_int, P]),
    'wn_causal_gather': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int,
                                 P]),
    'wn_scalar_causal_fwd': (c_int, [P, P, c_int, P, c_int, c_int, c_int, P]),
    'wn_scalar_causal_wgrad': (c_int, [P, P, P, c_int, c_int, c_int, c_int,
                                       P]),
    'wn_causal_wgrad_slabs': (c_int, [c_long]),
    'wn_causal_wgrad': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'wn_layer_fwd': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int,
                             c_int, c_int, P]),
    'wn_stack_flag_count': (c_long, [c_int, c_int, c_int]),
    'wn_stack_wimg_floats': (c_int, []),
    'wn_stack_tile_rows': (c_int, [c_int, c_int, c_int]),
    'wn_stack_pack': (c_int, [P, c_long, P, P, c_int, P]),
    'wn_stack_fwd': (c_int, [P, P, P, P, P, c_long, c_int, P, P, P, P,
                             c_int, c_int, c_int, c_int, c_int, P]),
    'wn_stack_fwd_skip': (c_int, [P, P, P, P, P, c_long, c_int, P, P, P, P,
                                  c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    'wn_stack_fwd_skip_ok': (c_int, [c_int, c_int, c_int, c_int]),
    'wn_stack_skip_img_floats': (c_long, [c_int]),
    'wn_stack_skip_pack': (c_int, [P, c_int, P, P]),
    'wn_stack_bwd_slabs': (c_int, [c_int, c_int, c_int]),
    'wn_stack_bwd': (c_int, [P, P, P, P, P, c_long, P, P, P, c_long, P, P, P,
                             P, P, c_int, c_int, c_int, c_int, P]),
    'wn_layer_wgrad_slab_floats': (c_int, []),
    'wn_dense_planes': (c_int, [P, c_long, P, P, P, c_long, P, c_long, c_long, c_int,
                                P]),
    'wn_dense_planes_gate': (c_int, [P, c_long, P, P, c_long, P, P, c_long, P, P,
                                     c_long, c_long, c_int, P]),
    'wn_layer_fwd_k': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, P]),
    'wn_layer_bwd_k': (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int,
                               c_int, c_int, c_int, c_int, c_int, c_long, P]),
    'wn_layer_wgrad_k': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_long, P]),
    'wn_layer_fwd_blk': (c_int, [P, c_long, c_int, P, P, P, P, P, c_int, P, P,
                                 c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, P, P, c_long, c_int, c_int, c_int,
                                 c_long, P]),
    'wn_layer_bwd_blk': (c_int, [P, P, c_long, c_int, P, P, P, P, c_int,
                                 c_long, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_long, P]),
    'wn_layer_bwd2_slabs': (c_int, [c_int, c_int]),
    'wn_layer_bwd2_wimg_floats': (c_int, []),
    'wn_layer_bwd2_pack': (c_int, [P, c_long, P, c_int, P]),
    'wn_layer_bwd2': (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int,
                              c_int, P]),
    'wn_gemm_nn': (c_int, [P, c_long, c_int, c_long, P, c_int, P, P, c_long,
                           P, c_long, P, c_long, c_int, c_long, P, c_long,
                           c_int, c_int, c_int, P]),
    'wn_gemm_split_w_bytes': (c_long, [c_int, c_int]),
    'wn_gemm_nn_split': (c_int, [P, c_long, c_int, c_long, P, c_int, P, P,
                                 c_long, P, c_long, P, c_long, c_int, c_long,
                                 P, c_long, c_int, c_int, c_int, P, c_int, P]),
    'wn_gemm_tn_tail_rows': (c_int, [c_int, c_int]),
    'wn_gemm_tn_slab_floats': (c_long, [c_int, c_int]),
    'wn_gemm_tn_splits': (c_int, [c_long, c_int, c_int, c_int]),
    'wn_gemm_tn_split': (c_int, [P, c_long, c_int, c_long, P, c_long, P, c_int,
                                 c_long, c_int, c_int, c_int, c_int, P]),
    'wn_gemm_tn': (c_int, [P, c_long, c_int, c_long, P, c_int, c_int, P,
                           c_long, P, c_int, c_long, c_int, c_int, c_int, P]),
    'wn_reduce_slabs': (c_int, [P, c_int, c_long, c_int, c_long, c_long,
                                c_long, P, c_long, c_int, c_long, P]),
    'wn_reduce_slabs_mt': (c_int, [P, c_int, c_long, c_long, P, c_long, P, c_int,
                                   c_long, c_int, P]),
    'wn_reduce_pair_slabs': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P,
                                     c_int, c_long, c_int, c_int, P]),
    'wn_transpose': (c_int, [P, c_int, c_int, c_long, P, c_long, P]),
    'wn_xent_partials': (c_int, [c_long]),
    'wn_xent': (c_int, [P, c_long, P, P, P, c_int, c_int, c_int, c_int, P]),
    'wn_softmax64_row': (c_int, [P, c_int, P, P]),
    'wn_adam': (c_int, [P, P, P, P, c_long, c_float, c_float, c_float,
                        c_float, c_float, c_float, P, P]),
    'wn_momentum': (c_int, [P, P, P, c_long, c_float, c_float, c_float,
                            c_float, P, P]),
    'wn_rmsprop': (c_int, [P, P, P, P, c_long, c_float, c_float, c_float,
                           c_float, c_float, c_float, P, P]),
    'wn_l2_partials_count': (c_int, []),
    'wn_l2_partials': (c_int, [P, c_long, P, P, P]),
    'wn_gc_bias': (c_int, [P, c_long, c_long, c_long, c_int, P, c_int, P, P,
                           c_int, c_int, c_int, P]),
    'wn_colsum_clip_chunks': (c_int, [c_int]),
    'wn_colsum_clip': (c_int, [P, P, c_int, c_int, P, P, P]),
    'wn_gc_grad': (c_int, [P, c_long, c_long, c_int, P, c_int, P, P, c_int,
                           c_int, P, P, P, c_int, P]),
    'wn_causal_conv': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int,
                               c_int, P]),
    'wn_time_to_batch': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'wn_batch_to_time': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'wn_diag_mfma_peak': (c_int, [P, c_int, c_int, P]),
    'wn_axpy': (c_int, [P, P, c_float, P, c_long, P]),
    'wn_fill': (c_int, [P, c_long, c_float, P]),
    'wn_sum_rows': (c_int, [P, c_int, c_int, P, P]),
    'wn_fastgen_state_floats': (c_long, [P, c_int]),
    'wn_fastgen_init': (c_int, [P, c_long, P, c_int, P]),
    'wn_fastgen_run': (c_int, [P, P, c_long, P, P, P, P, P, P, P, P, c_int,
                               c_int, c_int, P, P, P, c_int, c_int, c_float,
                               c_u64, P, c_int, c_int, c_int, P]),
    'wn_fastgen_run_wide': (c_int, [P, P, c_long, P, P, P, P, P, P, P, P,
                                    c_int, c_int, c_int, c_int, P, P, P, c_int,
                                    c_int, c_float, c_u64, P, c_int, c_int,
                                    c_int, P, P]),
    'wn_fastgen_wide_coop_bytes': (c_long, [c_int, c_int, c_int, c_int]),
    'wn_fastgen_step': (c_int, [P, P, c_long, P, P, P, P, P, P, P, P, c_int,
                                c_int, c_int, P, P, P, P, P, c_int, P, P, P,
                                P, P, P, P]),
    'wn_fastgen_persist_workgroups': (c_int, [c_int, c_int, c_int]),
    'wn_fastgen_persist_role': (c_int, [c_int, c_int, c_int]),
    'wn_fastgen_persist_ll_words': (c_long, [c_int, c_int, c_int]),
    'wn_fastgen_persist': (c_int, [P, P, c_long, P, P, P, P, P, P, P, P, c_int,
                                   c_int, c_int, P, P, P, P, P, c_int, P, P, P,
                                   P, P, P, P, P, c_int, P]),
    'wn_fastgen_pre': (c_int, [P, c_long, P, P, c_int, P, P, P, P]),
    'wn_fastgen_finish': (c_int, [c_int, P, P, P, P, P, P]),
    'wn_fastgen_pack': (c_int, [P, c_long, P, c_int, P]),
}



_lib = None


class WaveNetHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WaveNetHipError(
            'libwavenet_hip.so not found at %s: build it with '
            '`python __graft_entry__.py build` (hipcc --offload-arch=gfx950). '
            'There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        lib = load()
        msg = lib.wn_error_string(int(code)).decode()
        raise WaveNetHipError('%s failed: %s (code %d)' % (what, msg, code))


# ---- launch plans --------------------------------------------------------
# A training step is ~235 kernel launches whose arguments (raw device
# addresses and ints) do not change from step to step.  Re-deriving them in
# Python every step (tensor views, data_ptr(), current_stream()) costs about as
# much host time as the GPU needs for the step, so the host records the
# (function, args) sequence once per workspace and replays it.
_rec = None


class record(object):
    """Context manager: every `call` inside is executed AND appended to
    `self.plan` as (ctypes function, args, name, flops-or-None)."""

    def __enter__(self):
        global _rec
        self._outer = _rec
        self.plan = []
        _rec = self.plan
        return self

    def __exit__(self, *exc):
        global _rec
        _rec = self._outer
        return False


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero code."""
    lib = load()
    fn = getattr(lib, name)
    if _rec is not None:
        _rec.append((fn, args, name, None))
    code = fn(*args)
    if code != 0:
        check(code, name)


def call_py(fn):
    """Run a host-side action that belongs INTO the launch sequence (a
    cross-stream event record / wait) and, while recording, append it to the
    plan so that a replay repeats it at the same position."""
    def wrapped():
        fn()
        return 0
    if _rec is not None:
        _rec.append((wrapped, (), 'py', None))
    wrapped()


def call_timed(name, args, flops, events):
    """`call` for the GEMMs; with `events` (a list) the launch is bracketed by
    HIP events on torch's current stream and (start, end, flops) is appended
    (bench.py's live roofline measurement)."""
    lib = load()
    fn = getattr(lib, name)
    if _rec is not None:
        _rec.append((fn, args, name, flops))
    _timed(fn, args, name, flops, events)


def _timed(fn, args, name, flops, events):
    if events is None:
        code = fn(*args)
    else:
        import torch
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        code = fn(*args)
        e.record()
        events.append((s, e, flops, name))
    if code != 0:
        check(code, name)


def replay(plan, events=None):
    for fn, args, name, flops in plan:
        if flops is not None and events is not None:
            _timed(fn, args, name, flops, events)
        else:
            code = fn(*args)
            if code != 0:
                check(code, name)


def ptr(t):
    """Raw address of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stack_variant(rows=0, waves=0):
    """WN_STACK_VARIANT of include/wavenet_hip.h: the explicit variant word of
    the stack launches (0 = the library's choice for the shape)."""
    return (rows & 0x3f) | ((waves & 0xf) << 8)


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise WaveNetHipError(
            'no MI355X/ROCm device visible: the wavenet HIP path needs a GPU '
            '(there is no CPU fallback)')
