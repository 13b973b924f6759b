"""ctypes binding of ``libtonic_hip.so`` (the C ABI declared in ``include/tonic_hip.h``).

The product path has NO fallback: if the shared library is missing or a call fails, a
``TonicHipError`` is raised.  Build the library with ``python __graft_entry__.py build`` or
``make -C tonic_amd/csrc`` (hipcc cross-compiles gfx950 without a GPU).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, 'libtonic_hip.so')

c_float_p = ctypes.c_void_p   # device pointers travel as plain integers
c_i32, c_i64, c_f64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

class QOptimizer(ctypes.Structure):
    """tonic_q_optimizer of include/tonic_hip.h."""
    _fields_ = [('d_grad_sums', c_vp), ('d_exp_avg', c_vp), ('d_exp_avg_sq', c_vp),
                ('d_state', c_vp), ('d_info_row', c_vp), ('d_step_constants', c_vp),
                ('lr', c_f64), ('beta1', c_f64), ('beta2', c_f64), ('eps', c_f64)]


class QStore(ctypes.Structure):
    """tonic_q_store_t of include/tonic_hip.h (field for field)."""
    _fields_ = [('d_buf_observations', c_vp), ('d_buf_actions', c_vp), ('d_buf_next_observations', c_vp),
                ('d_buf_rewards', c_vp), ('d_buf_resets', c_vp), ('d_buf_terminations', c_vp),
                ('d_buf_discounts', c_vp), ('d_observations', c_vp), ('d_norm_acc', c_vp),
                ('row', c_i64), ('discount_factor', c_f64)]


class QIteration(ctypes.Structure):
    """tonic_q_iteration_t of include/tonic_hip.h (field for field)."""
    _fields_ = [('kind', c_i32), ('actor_due', c_i32),
                ('B', c_i32), ('O', c_i32), ('H', c_i32), ('A', c_i32),
                ('global_batch', c_i64),
                ('d_actor', c_vp), ('d_critics', c_vp), ('d_target_actor', c_vp),
                ('d_target_critics', c_vp),
                ('d_norm_mean', c_vp), ('d_norm_std', c_vp), ('norm_clip', c_f64),
                ('d_observations', c_vp), ('d_actions', c_vp), ('d_next_observations', c_vp),
                ('d_rewards', c_vp), ('d_discounts', c_vp),
                ('d_eps_critic', c_vp), ('d_eps_actor', c_vp),
                ('critic_entropy_coeff', c_f64), ('actor_entropy_coeff', c_f64),
                ('noise_scale', c_f64), ('noise_clip', c_f64), ('target_coeff', c_f64),
                ('critic', QOptimizer), ('actor', QOptimizer),
                ('d_workspace', c_vp), ('workspace_bytes', c_i64), ('phase', c_i32),
                ('refresh_images', c_i32), ('stage', c_i32), ('slot', c_i32), ('ahead', ctypes.c_void_p)]


# name -> (restype, argtypes); mirrors include/tonic_hip.h one to one.
SIGNATURES = {
    'tonic_last_error': (ctypes.c_char_p, []),
    'tonic_abi_version': (c_i32, []),
    'tonic_target_arch': (ctypes.c_char_p, []),
    'tonic_set_tuning': (ctypes.c_int, [ctypes.c_char_p, c_i32]),
    'tonic_get_tuning': (ctypes.c_int, [ctypes.c_char_p, c_vp]),
    'tonic_ppo_actor_param_count': (c_i64, [c_i32, c_i32]),
    'tonic_v_critic_param_count': (c_i64, [c_i32]),
    'tonic_gae_workspace_bytes': (c_i64, [c_i64, c_i64, c_i32]),
    'tonic_gae_lambda_returns': (ctypes.c_int, [c_vp] * 9 + [c_i64, c_i64, c_f64, c_f64, c_i32,
                                                              c_vp, c_i64, c_vp]),
    'tonic_advantage_stats_from_moments': (ctypes.c_int, [c_vp, c_vp, c_vp]),
    'tonic_ppo_act': (ctypes.c_int, [c_vp] * 5 + [c_i64, c_i32, c_i32, c_vp]),
    'tonic_value_forward': (ctypes.c_int, [c_vp] * 3 + [c_f64] + [c_vp] * 2 + [c_i64, c_i32, c_vp]),
    'tonic_mlp64_grad_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'tonic_ppo_workspace_bytes': (c_i64, [c_i64, c_i32, c_i32, c_i32]),
    'tonic_ppo_act_wide': (ctypes.c_int, [c_vp] * 5 + [c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    'tonic_value_forward_wide': (ctypes.c_int, [c_vp] * 3 + [c_f64] + [c_vp] * 2 +
                                 [c_i64, c_i32, c_vp, c_i64, c_vp]),
    'tonic_ppo_actor_grad': (ctypes.c_int, [c_vp] * 7 + [c_i64, c_i32, c_i32, c_f64, c_f64,
                                                          c_vp, c_i32, c_vp, c_i64, c_vp]),
    'tonic_value_regression_grad': (ctypes.c_int, [c_vp] * 3 + [c_f64] + [c_vp] * 3 +
                                    [c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    'tonic_ppo_torso_param_count': (c_i64, [c_i32, c_i32, c_i32, c_i32, c_vp]),
    'tonic_ppo_torso_workspace_bytes': (c_i64, [c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'tonic_ppo_act_torso': (ctypes.c_int, [c_i32, c_vp, c_i32] + [c_vp] * 5 + [c_i64, c_i32, c_i32, c_vp,
                                                                           c_i64, c_vp]),
    'tonic_value_forward_torso': (ctypes.c_int, [c_i32, c_vp, c_i32] + [c_vp] * 3 + [c_f64] + [c_vp] * 2 +
                                  [c_i64, c_i32, c_vp, c_i64, c_vp]),
    'tonic_ppo_actor_grad_torso': (ctypes.c_int, [c_i32, c_vp, c_i32] + [c_vp] * 7 +
                                   [c_i64, c_i32, c_i32, c_f64, c_f64, c_vp, c_vp, c_i64, c_vp]),
    'tonic_value_regression_grad_torso': (ctypes.c_int, [c_i32, c_vp, c_i32] + [c_vp] * 3 + [c_f64] +
                                          [c_vp] * 3 + [c_i64, c_i32, c_vp, c_i64, c_vp]),
    'tonic_adam_step': (ctypes.c_int, [c_vp] * 5 + [c_i64, c_f64, c_f64, c_f64, c_f64, c_f64,
                                                     c_i32, c_f64, c_f64, c_vp, c_vp, c_vp, c_vp]),
    'tonic_clip_workspace_bytes': (c_i64, [c_i64]),
    'tonic_clip_grad_norm': (ctypes.c_int, [c_vp, c_i64, c_f64, c_f64, c_vp, c_vp, c_i64, c_vp]),
    'tonic_adam_step_pair': (ctypes.c_int,
                             [c_vp] * 5 + [c_i64, c_f64, c_i32, c_f64, c_f64, c_vp, c_vp, c_vp] +
                             [c_vp] * 5 + [c_i64, c_f64, c_i32, c_vp] + [c_f64] * 4 + [c_vp]),
    'tonic_segment_store': (ctypes.c_int, [c_vp] * 15 + [c_i64, c_i64, c_i32, c_i32, c_vp]),
    'tonic_meanstd_record': (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    'tonic_segment_gather': (ctypes.c_int, [c_vp] * 11 + [c_i64, c_i64, c_i32, c_i32, c_vp]),
    'tonic_ppo_collect_step': (ctypes.c_int, [c_vp] * 16 + [c_i64, c_i64, c_i32, c_i32, c_vp]),
    'tonic_ppo_packed_actor_floats': (c_i64, [c_i32, c_i32]),
    'tonic_ppo_pack_actor': (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    'tonic_ppo_collect_step_packed': (ctypes.c_int, [c_vp] * 16 + [c_i64, c_i64, c_i32, c_i32, c_vp]),
    'tonic_ppo_collect_steps_packed': (ctypes.c_int, [c_vp] * 14 + [c_i64, c_i64, c_i64, c_i32, c_i32,
                                                                    c_vp]),
    'tonic_q_iteration_workspace_bytes': (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    'tonic_q_iteration_supported': (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32]),
    'tonic_q_iteration_ahead_supported': (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]),
    'tonic_q_iteration': (ctypes.c_int, [c_vp, c_vp]),
    'tonic_polyak_update': (ctypes.c_int, [c_vp, c_vp, c_i64, c_f64, c_vp]),
    'tonic_adam_polyak_step': (ctypes.c_int, [c_vp] * 5 + [c_i64] * 3 + [c_f64] * 5 + [c_i32, c_vp, c_vp,
                                                                               c_f64, c_vp]),
    'tonic_buffer_accumulate_n_steps': (ctypes.c_int, [c_vp] * 7 + [c_i64] * 4 + [c_i32, c_i32, c_f64,
                                                                      c_vp]),
    'tonic_offpolicy_workspace_bytes': (c_i64, [c_i32] * 4),
    'tonic_mlp_weight_stride': (c_i32, [c_i32]),
    'tonic_mlp_hidden': (c_i32, [c_i32, c_i32, c_i32]),
    'tonic_mlp_actor_param_count': (c_i64, [c_i32] * 4),
    'tonic_q_critic_param_count': (c_i64, [c_i32] * 3),
    'tonic_buffer_store': (ctypes.c_int, [c_vp] * 14 + [c_i64, c_i64, c_i32, c_i32, c_f64, c_vp]),
    'tonic_buffer_gather': (ctypes.c_int, [c_vp] * 11 + [c_i64, c_i32, c_i32, c_i32, c_vp]),
    'tonic_policy_forward': (ctypes.c_int, [c_vp] * 4 + [c_i32] * 5 + [c_vp, c_i64, c_vp]),
    'tonic_twin_q_grad': (ctypes.c_int, [c_i32] + [c_vp] * 5 + [c_f64] + [c_vp] * 7 + [c_i32] * 4 +
                          [c_f64] * 3 +
                          [c_vp, c_i64, c_vp]),
    'tonic_actor_q_grad': (ctypes.c_int, [c_i32] + [c_vp] * 4 + [c_f64] + [c_vp] * 3 + [c_i32] * 4 +
                           [c_f64] +
                           [c_vp, c_i64, c_vp]),
    'tonic_distributional_workspace_bytes': (c_i64, [c_i32] * 5),
    'tonic_distributional_q_grad': (ctypes.c_int, [c_vp] * 5 + [c_f64] + [c_vp] * 7 + [c_i32] * 5 +
                                    [c_vp, c_i64, c_vp]),
    'tonic_distributional_actor_grad': (ctypes.c_int, [c_vp] * 4 + [c_f64] + [c_vp] * 3 +
                                        [c_i32] * 5 + [c_vp, c_i64, c_vp]),
    'tonic_mpo_workspace_bytes': (c_i64, [c_i32] * 5),
    'tonic_expected_sarsa_grad': (ctypes.c_int, [c_vp] * 5 + [c_f64] + [c_vp] * 7 + [c_i32] * 5 +
                                  [c_vp, c_i64, c_vp]),
    'tonic_mpo_actor_grad': (ctypes.c_int, [c_vp] * 4 + [c_f64] + [c_vp] * 2 + [c_f64] + [c_vp] * 5 +
                             [c_i32] * 5 + [c_f64] * 4 + [c_i32] + [c_vp, c_i64, c_vp]),
    'tonic_mpo_actor_grad_shard': (ctypes.c_int, [c_vp] * 4 + [c_f64] + [c_vp] * 2 + [c_f64] + [c_vp] * 4 +
                                   [c_i32] * 6 + [c_vp, c_i64, c_vp]),
    'tonic_mpo_dual_step': (ctypes.c_int, [c_vp] * 2 + [c_f64] + [c_vp] * 3 + [c_i32] * 4 + [c_f64] * 4 +
                            [c_i32, c_vp]),
    'tonic_collector_block_bytes': (c_i64, [c_i64, c_i32, c_i32]),
    'tonic_collector_block_init': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_i32]),
    'tonic_collector_block_offset': (c_i64, [c_vp, c_i32]),
    'tonic_collector_synthetic_step': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32]),
    'tonic_collector_ring': (ctypes.c_int, [c_vp]),
    'tonic_collector_block_carry_over': (ctypes.c_int, [c_vp, c_i32]),
    'tonic_collector_arm': (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32]),
    'tonic_collector_q_act': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'tonic_mlp_actor_image_bytes': (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    'tonic_collector_claim': (ctypes.c_int, [c_vp]),
    'tonic_collector_worker_wait': (c_i64, [c_vp, c_i64, c_f64]),
    'tonic_collector_worker_done': (ctypes.c_int, [c_vp]),
    'tonic_collector_submit_actions': (ctypes.c_int, [c_vp]),
    'tonic_collector_wait_obs': (ctypes.c_int, [c_vp, c_f64]),
    'tonic_collector_shutdown': (ctypes.c_int, [c_vp]),
    'tonic_collector_create': (ctypes.c_int, [ctypes.POINTER(c_vp), c_vp, c_i32]),
    'tonic_host_device_pointer': (c_vp, [c_vp]),
    'tonic_collector_destroy': (ctypes.c_int, [c_vp]),
    'tonic_collector_stream': (c_vp, [c_vp]),
    'tonic_collector_transport': (c_i32, [c_vp]),
    'tonic_collector_bind_segment': (ctypes.c_int, [c_vp] * 9 + [c_i64]),
    'tonic_collector_begin_rollout': (ctypes.c_int, [c_vp, c_vp, c_vp]),
    'tonic_collector_ppo_step': (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32]),
    'tonic_collector_wait_actions': (ctypes.c_int, [c_vp, c_f64]),
    'tonic_collector_end_rollout': (ctypes.c_int, [c_vp, c_i64, c_vp]),
    'tonic_comm_handle_bytes': (c_i64, []),
    'tonic_comm_init': (ctypes.c_int, [ctypes.POINTER(c_vp), c_i32, c_i32, c_i64]),
    'tonic_comm_export': (ctypes.c_int, [c_vp, c_vp]),
    'tonic_comm_connect': (ctypes.c_int, [c_vp, c_vp]),
    'tonic_allreduce_f32': (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    'tonic_comm_status': (ctypes.c_int, [c_vp]),
    'tonic_comm_set_timeout': (ctypes.c_int, [c_vp, c_f64]),
    'tonic_comm_can_access_peer': (ctypes.c_int, [c_i32, c_i32]),
    'tonic_stream_gate': (ctypes.c_int, [c_vp, ctypes.c_uint32, c_f64, c_vp]),
    'tonic_comm_destroy': (ctypes.c_int, [c_vp]),
    'tonic_debug_grad16_phases': (ctypes.c_int, [c_vp] * 6 + [c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'tonic_debug_forward_stamps': (ctypes.c_int, [c_vp]),
    'tonic_debug_occupy': (ctypes.c_int, [c_i32, c_f64, c_vp]),
    'tonic_gemm_f32': (ctypes.c_int, [ctypes.c_char_p] + [c_vp] * 6 + [c_i32] * 8 + [c_f64, c_vp]),
}


ABI_VERSION = 11       # include/tonic_hip.h: tonic_abi_version()


class TonicHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the shared library (once) and declares every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise TonicHipError(
            f'{LIBRARY_PATH} not found: the HIP extension is required (no CPU fallback). '
            'Build it with `python __graft_entry__.py build` or `make -C tonic_amd/csrc`.')
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype, fn.argtypes = restype, argtypes
    if lib.tonic_abi_version() != ABI_VERSION:
        raise TonicHipError(f'{LIBRARY_PATH} has ABI {lib.tonic_abi_version()}, this package needs '
                            f'{ABI_VERSION}: rebuild it (`make -C tonic_amd/csrc`)')
    # developer switch: TONIC_AMD_TUNING="key=value,key=value" -> tonic_set_tuning at load
    for item in filter(None, os.environ.get('TONIC_AMD_TUNING', '').split(',')):
        key, _, value = item.partition('=')
        if lib.tonic_set_tuning(key.strip().encode(), int(value)) != 0:
            raise TonicHipError(f'TONIC_AMD_TUNING: {lib.tonic_last_error().decode()}')
    _lib = lib
    return lib


_fast = False          # False: not looked for yet; None: absent or switched off


def hot(name):
    """The binding of a per-environment-step entry point: the vectorcall shim tonic_amd/_fastcall*.so
    (csrc/fastcall.c: ~0.1 us per call) when it has been built, else the ctypes prototype (0.35 - 0.6 us).
    Same C entry, same arguments, same status codes either way; TONIC_AMD_FASTCALL=0 forces ctypes."""
    global _fast
    lib = load()
    if _fast is False:
        _fast = None
        if os.environ.get('TONIC_AMD_FASTCALL', '1') != '0':
            try:
                from tonic_amd import _fastcall
                _fastcall.bind(LIBRARY_PATH)
                _fast = _fastcall
            except ImportError:
                pass
            except Exception as error:       # a stale shim (other ABI, missing entry): ctypes serves, loudly
                import warnings
                warnings.warn(f'tonic_amd._fastcall not used ({error}); the per-step entries go through ctypes '
                              '(`make -C tonic_amd/csrc fast` rebuilds the shim)')
    return getattr(_fast, name) if _fast is not None else getattr(lib, name)


def check(status, what):
    if status != 0:
        msg = load().tonic_last_error().decode()
        raise TonicHipError(f'{what} failed with status {status}: {msg}')


def ptr(tensor):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if tensor is None:
        return None
    if not tensor.is_contiguous():
        raise TonicHipError('non-contiguous tensor passed to the HIP engine')
    return tensor.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


import contextlib
import gc


@contextlib.contextmanager
def capturing(graph, **kwargs):
    """``torch.cuda.graph(graph)`` with Python's cyclic garbage collector held off while the stream captures.  A
    collection that starts inside the capture may finalize device objects that earlier work left behind — tensors,
    events, an older captured graph: hipFree / hipGraphExecDestroy while a stream is capturing is an error, raised
    from a destructor, i.e. an abort (seen as a rare `Fatal Python error: Aborted ... Garbage-collecting` of a whole
    test session, never with the collector off).  Default error mode `thread_local`: calls of OTHER threads (a
    collective's watchdog, the noise helper) do not invalidate this thread's capture."""
    import torch
    kwargs.setdefault('capture_error_mode', 'thread_local')
    enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kwargs):
            yield
    finally:
        if enabled:
            gc.enable()

