"""ctypes binding of libmi_maml.so (include/mi_maml.h).  There is no CPU fallback: if the HIP library is missing or a call
fails, this raises."""

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(CSRC, 'libmi_maml.so')

_lib = None


class MiModelDesc(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('in_channels', C.c_int32), ('in_h', C.c_int32), ('in_w', C.c_int32),
                ('hidden', C.c_int32), ('max_pool', C.c_int32), ('ways', C.c_int32), ('head_mean_pool', C.c_int32)]


class MiPolicyDesc(C.Structure):
    _fields_ = [('state_size', C.c_int32), ('action_size', C.c_int32), ('hidden1', C.c_int32), ('hidden2', C.c_int32),
                ('activation', C.c_int32)]


class MiBnTangentArgs(C.Structure):
    _fields_ = [('z', C.c_void_p), ('zd', C.c_void_p), ('mu', C.c_void_p), ('rstd', C.c_void_p), ('m1', C.c_void_p),
                ('m2', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('pstride', C.c_size_t),
                ('gammad', C.c_void_p), ('betad', C.c_void_p), ('vstride', C.c_size_t), ('dgamma', C.c_void_p),
                ('dbeta', C.c_void_p), ('gstride', C.c_size_t), ('dp', C.c_void_p), ('dpd', C.c_void_p),
                ('tasks', C.c_int32), ('n', C.c_int32), ('ho', C.c_int32), ('wo', C.c_int32), ('c', C.c_int32),
                ('pool', C.c_int32)]


class MiBlock1Args(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('wd', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('pstride', C.c_size_t), ('gammad', C.c_void_p), ('betad', C.c_void_p), ('vstride', C.c_size_t),
                ('mu', C.c_void_p), ('rstd', C.c_void_p), ('m1', C.c_void_p), ('m2', C.c_void_p), ('dgamma', C.c_void_p),
                ('dbeta', C.c_void_p), ('gstride', C.c_size_t), ('rdgamma', C.c_void_p), ('rdbeta', C.c_void_p),
                ('hstride', C.c_size_t), ('dp', C.c_void_p), ('dpd', C.c_void_p), ('arg_in', C.c_void_p), ('zh_in', C.c_void_p),
                ('tasks', C.c_int32), ('n', C.c_int32), ('h', C.c_int32), ('w_', C.c_int32), ('ci', C.c_int32), ('co', C.c_int32)]


class MiError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', CSRC, '-j4'], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise MiError('building libmi_maml.so failed')
    return LIB_PATH


_SIGS = {
    'mi_engine_create': (C.c_int, [C.POINTER(MiModelDesc), C.c_int, C.POINTER(C.c_void_p)]),
    'mi_engine_destroy': (None, [C.c_void_p]),
    'mi_last_error': (C.c_char_p, [C.c_void_p]),
    'mi_version': (C.c_char_p, []),
    'mi_engine_set_fused_block1': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_engine_set_overlap': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_engine_set_graph': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_engine_set_bn_export': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_engine_set_fused_finalize': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_engine_set_fused_block1_reduce': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_debug_plan_offsets': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_param_count': (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    'mi_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_meta_batch_maml': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t]),
    'mi_meta_batch_maml_tv': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_engine_set_fused_tail': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_engine_set_fused_last_block': (C.c_int, [C.c_void_p, C.c_int]),
    'mi_debug_tail_stamps': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mi_anil_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_meta_batch_anil': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t]),
    'mi_forward_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_forward_logits': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_size_t]),
    'mi_learner_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_learner_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_learner_hvp_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_learner_hvp': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_cg_update': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double]),
    'mi_cg_update_checked': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double]),
    'mi_cg_init': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_trpo_scale_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    'mi_gae_max_rows': (C.c_int, [C.c_int]),
    'mi_upload_i32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'mi_copy_segments': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'mi_gae_advantages': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    'mi_adam_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_float,
                               C.c_float, C.c_float, C.c_float, C.c_float]),
    'mi_prepare_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_input_gram_scratch_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'mi_input_gram': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mi_gram_bn_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_stream_copy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_sample_tasks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p]),
    'mi_conv3x3_bn_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_bn_relu_pool': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mi_bn_relu_pool_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_conv3x3_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    'mi_head_fwd_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    'mi_kernel_scratch_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'mi_conv3x3_tangent': (C.c_int, [C.c_void_p] * 5 + [C.c_size_t] + [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p] * 4 + [C.c_size_t]),
    'mi_conv3x3_bwd2': (C.c_int, [C.c_void_p] * 7 + [C.c_size_t] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    'mi_bn_tangent_fwd': (C.c_int, [C.c_void_p, C.POINTER(MiBnTangentArgs), C.c_void_p]),
    'mi_bn_tangent_bwd': (C.c_int, [C.c_void_p, C.POINTER(MiBnTangentArgs), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_void_p, C.c_size_t]),
    'mi_block1_scratch_bytes': (C.c_size_t, [C.c_int] * 6),
    'mi_block1_run': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(MiBlock1Args), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    'mi_pooled_reduce': (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    'mi_block1_wgrad_gram': (C.c_int, [C.c_void_p, C.POINTER(MiBlock1Args), C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t]),
    'mi_debug_conv_tiles_per_wave': (C.c_int, [C.c_int] * 5),
    'mi_debug_set_trace': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_policy_create': (C.c_int, [C.POINTER(MiPolicyDesc), C.c_int, C.POINTER(C.c_void_p)]),
    'mi_policy_destroy': (None, [C.c_void_p]),
    'mi_policy_last_error': (C.c_char_p, [C.c_void_p]),
    'mi_policy_param_count': (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    'mi_trpo_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_policy_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_size_t]),
    'mi_policy_adapt': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_trpo_surrogate': (C.c_int, [C.c_void_p] * 13 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_void_p, C.c_size_t]),
    'mi_trpo_fvp': (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_size_t]),
    'mi_policy_set_fused_fvp': (C.c_int, [C.c_int]),
    'mi_debug_policy_sweep_stamps': (C.c_int, [C.c_void_p]),
    'mi_debug_conv_stamps': (C.c_int, [C.c_void_p]),
    'mi_conv_set_split_bf16': (C.c_int, [C.c_int]),
    'mi_conv_get_split_bf16': (C.c_int, [C.POINTER(C.c_uint)]),
    'mi_conv_set_b16': (C.c_int, [C.c_int]),
    'mi_block1_set_split_bf16': (C.c_int, [C.c_int]),
    'mi_sparse_wgrad_set_split_bf16': (C.c_int, [C.c_int]),
    'mi_trpo_general_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_trpo_kl_prepare': (C.c_int, [C.c_void_p] * 10 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_trpo_fvp_general': (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                           C.c_size_t]),
    'mi_policy_meta_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_policy_meta_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 +
                             [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_policy_meta_batch_dones': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10 +
                                   [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_trpo_steps_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'mi_trpo_surrogate_steps': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10 +
                                [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_trpo_fvp_steps': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 +
                          [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'mi_profile_enable': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'mi_profile_kinds': (C.c_int, []),
    'mi_profile_op_name': (C.c_char_p, [C.c_int]),
    'mi_profile_collect': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
}

EXPORTS = tuple(_SIGS)


def load():
    """Load libmi_maml.so (built in-tree by ``build()`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('MI_MAML_LIB', LIB_PATH)          # A/B runs of two builds of the same sources on one box
    if not os.path.exists(path):
        raise MiError(f'{path} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                      '(hipcc --offload-arch=gfx950). There is no CPU fallback for the MAML hot path.')
    import torch  # noqa: F401  -- torch's bundled HIP runtime must be the one in the process before ours is resolved
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != 0:
        msg = load().mi_last_error(handle)
        raise MiError(f'libmi_maml error {rc}: {msg.decode() if msg else "?"}')
