"""ctypes binding of libreconvat_hip.so (include/reconvat_hip.h).

The product path has NO fallback: if the shared library is missing, importing a symbol raises, and every
op raises when its inputs are not on a HIP device.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RECONVAT_HIP_LIB: A/B another build of the same ABI (kernel experiments); default = the in-tree library
LIB_PATH = os.environ.get('RECONVAT_HIP_LIB') or os.path.join(_HERE, 'libreconvat_hip.so')

P = ctypes.c_void_p
U = ctypes.c_uint
I = ctypes.c_int
L = ctypes.c_long
F = ctypes.c_float

# name -> (restype, argtypes); mirrors include/reconvat_hip.h one to one
SIGNATURES = {
    'rv_abi_version': (I, []),
    'rv_last_error': (ctypes.c_char_p, []),
    'rv_source_digest': (ctypes.c_char_p, []),
    'rv_melspec_lognorm_fwd': (I, [P, L, I, I, P, P, P, P, P, I, I, I, I, I, P, I, P, P]),
    'rv_packed_weight_floats': (L, [I, I, I]),
    'rv_pack_weights': (I, [P, P, I, I, I, L, L, I, I, I, P]),
    'rv_pack_table_entry_bytes': (L, []),
    'rv_pack_table_fill': (L, [P, I, P, P, I, I, I, L, L, I, I, I]),
    'rv_pack_table_run': (I, [P, I, L, P]),
    'rv_conv_fwd': (I, [I, P, I, I, I, I, I, P, I, I, I, I, P, P, I, I, P, P, I, P, F, P]),
    'rv_conv_wgrad_workspace_bytes': (L, [I, I, I, I, I]),
    'rv_conv_wgrad_set_plan': (I, [I, I, I, I, I, I, I]),
    'rv_conv_wgrad': (I, [I, P, I, I, I, I, P, I, I, I, I, I, P, L, L, I, P, I, P, L, P]),
    'rv_conv_wgrad_deferred': (L, [I, P, I, I, I, I, P, I, I, I, I, I, P, L, L, I, P, P, L, P, P]),
    'rv_conv_wgrad_seg': (I, [I, I, P, P, I, I, I, I, I, I, I, I, I, P, L, L, I, P, I, P, L, P]),
    'rv_conv_wgrad_deferred_seg': (L, [I, I, P, P, I, I, I, I, I, I, I, I, I, P, L, L, I, P, P, L, P, P]),
    'rv_wgrad_table_entry_bytes': (L, []),
    'rv_wgrad_table_finalize': (L, [P, I]),
    'rv_wgrad_reduce_table': (I, [P, I, L, P]),
    'rv_bn_workspace_bytes': (L, [I]),
    'rv_bn_lrelu_fwd': (I, [P, I, L, I, P, P, P, P, P, F, F, I, F, P, I, P, I, P, P, I, P]),
    'rv_bn_lrelu_fwd_skip': (I, [P, I, L, I, P, P, P, P, P, F, F, I, F, P, I, I, P, P, I, P, I, P, P, I, P]),
    'rv_bn_running_update': (I, [P, P, P, P, I, F, P]),
    'rv_bn_running_update_table': (I, [P, I, F, P]),
    'rv_bn_lrelu_bwd': (I, [P, I, P, I, L, I, P, F, I, P, I, P, P, I, P, I, P]),
    'rv_gemm_splitk_workspace_bytes': (L, [I, I, I, I]),
    'rv_gemm_splitk_ticket_bytes': (L, [I, I, I, I]),
    'rv_gemm': (I, [P, L, L, P, L, L, P, L, L, P, L, L, P, I, I, I, I, I, I, I, L, L, L, P, P, P, P]),
    'rv_gemm_table_entry_bytes': (L, []),
    'rv_gemm_table_fill': (L, [P, P, L, L, P, L, L, P, L, L, P, I, I, I, I, I, L, L, L, P, P]),
    'rv_gemm_table_finalize': (L, [P, I, P]),
    'rv_gemm_table_run': (I, [P, I, L, L, I, P]),
    'rv_sigmoid_bwd': (I, [P, I, P, I, P, I, P, I, L, I, P]),
    'rv_colsum': (I, [P, I, L, I, P, I, P]),
    'rv_colsum_ordered_workspace_bytes': (L, [L, I]),
    'rv_colsum_ordered': (I, [P, I, L, I, P, I, P, P]),
    'rv_sums_fold': (I, [P, I, I, I, P, I, P]),
    'rv_local_attn_fwd': (I, [P, P, P, L, P, P, P, I, I, I, I, P]),
    'rv_local_attn_bwd': (I, [P, P, P, P, L, P, P, P, P, P, L, P, I, I, I, I, P]),
    'rv_vat_perturb_fwd': (I, [P, P, L, I, F, F, P, P, P, P, P]),
    'rv_vat_perturb_bwd': (I, [P, P, P, L, I, F, F, P, P]),
    'rv_reduce_workspace_bytes': (L, [L]),
    'rv_reduce_mean': (I, [I, P, P, L, P, P, P, P]),
    'rv_loss_bwd': (I, [I, P, P, L, P, P, P]),
    'rv_adam_step': (I, [P, P, P, P, L, P, F, L, F, F, F, F, F, P, P]),
    'rv_counter_add': (I, [P, L, P, P]),
    'rv_clip_scale': (I, [P, L, P, F, P]),
    'rv_crop_segments': (I, [P, P, P, P, P, I, L, I, I, P, P, P, P, P, P]),
    'rv_lstm_flag_bytes': (L, [I]),
    'rv_lstm_fwd': (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    'rv_lstm_bwd': (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    'rv_maxpool_w2_dropout_fwd': (I, [P, P, P, L, I, I, F, U, P, P]),
    'rv_maxpool_w2_dropout_bwd': (I, [P, P, P, L, I, I, F, P]),
    'rv_dropout': (I, [P, P, P, P, L, F, U, P, P]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP extension has not been built. '
            'Run `python -c "import __graft_entry__ as g; g.build()"` (or `python reconvat_amd/build.py`). '
            'There is no CPU/PyTorch fallback for the product path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the library is git-ignored and ships prebuilt next to its sources: refuse one that was built from OTHER sources
    # (an explicitly named experiment library -- RECONVAT_HIP_LIB: ablation / A-B builds of the tools -- or RV_SKIP_DIGEST_CHECK=1 is
    # exempt; the in-tree default, i.e. everything the product, the tests, smoke() and bench.py load, never is)
    built_from, tree = lib.rv_source_digest().decode(), source_digest()
    if built_from != tree and not os.environ.get('RECONVAT_HIP_LIB') and os.environ.get('RV_SKIP_DIGEST_CHECK') != '1':
        raise RuntimeError(f'{LIB_PATH} was built from sources with digest {built_from}, the sources in reconvat_amd/csrc have {tree}: '
                           'rebuild it (`python reconvat_amd/build.py`)')
    _lib = lib
    return lib


def source_digest():
    """Digest of the kernel sources in the tree (reconvat_amd/build.py::source_digest); load() compares it with rv_source_digest()."""
    from . import build
    return build.source_digest()


def last_error():
    return load().rv_last_error().decode()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('reconvat_amd ops run on a HIP device only (got a CPU tensor); there is no CPU fallback')


# Measurement hook (bench.py / tools): HOOK[0](name, args, fn) -> status is called INSTEAD of fn(*args) for every launch the
# host side makes through invoke() / call() -- it records the launch (or brackets it with HIP events) and calls through.
HOOK = [None]


def invoke(name, *args):
    """Invoke an entry point and return its status (no exception)."""
    fn = getattr(load(), name)
    hook = HOOK[0]
    return hook(name, args, fn) if hook is not None else fn(*args)


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = invoke(name, *args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib.rv_last_error().decode()}')
