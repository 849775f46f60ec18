"""ctypes binding of include/pcgc.h — the ONLY way the Python host reaches compute.

There is no fallback: if libpcgc_hip.so / libpcgc_host.so are missing or a call
fails, an exception is raised (the product never routes through oracle/ or a
torch implementation of the hot path).
"""
import ctypes
import os


_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")

c_int, c_i64, c_f32, c_vp, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/pcgc.h declares
HIP_API = {
    "pcgc_version": (c_int, []),
    "pcgc_last_error": (ctypes.c_char_p, []),
    "pcgc_conv3d_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "pcgc_vrn_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "pcgc_net_param_count": (c_int, [c_int]),
    "pcgc_net_create": (c_int, [c_int, c_vp, c_int, c_vp, c_vp]),
    "pcgc_net_destroy": (None, [c_vp]),
    "pcgc_net_set_algo": (c_int, [c_vp, c_int]),
    "pcgc_net_set_skip_counter": (c_int, [c_vp, c_vp]),
    "pcgc_rowocc": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "pcgc_net_set_profiling": (c_int, [c_vp, c_int]),
    "pcgc_net_profile_report": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "pcgc_net_workspace_bytes": (c_sz, [c_vp, c_int, c_int]),
    "pcgc_net_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_vp, c_sz, c_vp]),
    "pcgc_repro_eval": (c_int, [c_int, c_vp, c_vp, c_i64, c_vp]),
    "pcgc_round_minmax": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "pcgc_round_minmax_i16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "pcgc_symbols_to_values": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp]),
    "pcgc_symbols_to_values_seg": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "pcgc_laplace_likelihood": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp]),
    "pcgc_laplace_cdf": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "pcgc_factorized_likelihood": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_vp]),
    "pcgc_factorized_pmf": (c_int, [c_vp, c_int, c_int, c_int, c_f32, c_vp, c_vp]),
    "pcgc_topk_threshold": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_topk_workspace_bytes": (c_sz, [c_int, c_i64]),
    "pcgc_bce_sums": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_bce_workspace_bytes": (c_sz, [c_i64]),
    "pcgc_classify_workspace_bytes": (c_sz, []),
    "pcgc_classify_sums": (c_int, [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_confusion_matrix": (c_int, [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "pcgc_focal_workspace_bytes": (c_sz, []),
    "pcgc_focal_loss": (c_int, [c_vp, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_focal_loss_bwd": (c_int, [c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp, c_vp]),
    "pcgc_voxelize": (c_int, [c_vp, c_i64, c_int, c_vp, c_int, c_vp]),
    "pcgc_voxelize_points": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "pcgc_d1_workspace_bytes": (c_sz, [c_int]),
    "pcgc_d1_mse": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_d2_workspace_bytes": (c_sz, [c_int, c_i64]),
    "pcgc_d2_transfer_normals": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_d2_mse": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_conv3d_bwd_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "pcgc_conv3d_bwd_data": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "pcgc_conv3d_bwd_data_fused": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "pcgc_conv3d_bwd_weight": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "pcgc_relu_bwd": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_i64, c_int, c_vp]),
    "pcgc_vrn_merge": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "pcgc_add_inplace": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "pcgc_vrn_bwd_split": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pcgc_vrn_fwd_train_signs_supported": (c_int, [c_int, c_int]),
    "pcgc_vrn_fwd_train_signs": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_fwd_train_q4": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_layout_q4": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "pcgc_train_plan_set_layout": (c_int, [c_vp, c_int, c_int, c_int]),
    "pcgc_vrn_bwd_split_signs": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pcgc_vrn_bwd_tail_supported": (c_int, [c_int, c_int]),
    "pcgc_vrn_bwd_tail_split_supported": (c_int, [c_int, c_int]),
    "pcgc_vrn_bwd_tail": (c_int, [c_vp] * 11 + [c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_bwd_tail_split": (c_int, [c_vp] * 13 + [c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_bwd_tail_split_q4": (c_int, [c_vp] * 13 + [c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_bwd_input_supported": (c_int, [c_int, c_int]),
    "pcgc_vrn_bwd_input": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_bwd_input_q4": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_vrn_fwd_train_supported": (c_int, [c_int, c_int]),
    "pcgc_vrn_fwd_train": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_train_plan_create": (c_int, [c_vp, c_int, c_vp]),
    "pcgc_train_plan_destroy": (None, [c_vp]),
    "pcgc_train_plan_layers": (c_int, [c_vp]),
    "pcgc_train_plan_prepare": (c_int, [c_vp, c_vp]),
    "pcgc_train_conv_fwd": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_train_conv_bwd_data": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "pcgc_train_conv_bwd_weight": (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp]),
    "pcgc_train_conv_bwd_weight_pair": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "pcgc_train_conv_fwd_pair": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "pcgc_train_conv_bwd_data_pair": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "pcgc_train_conv_bwd_data_chain": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "pcgc_train_conv_fwd_merge": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_train_plan_finish_weights": (c_int, [c_vp, c_vp]),
    "pcgc_train_plan_defer_small": (c_int, [c_vp, c_int]),
    "pcgc_abs_max": (c_int, [c_vp, c_f32, c_vp, c_vp, c_i64, c_vp]),
    "pcgc_laplace_likelihood_bwd": (c_int, [c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pcgc_factorized_bwd_workspace_bytes": (c_sz, [c_int]),
    "pcgc_factorized_likelihood_bwd": (c_int, [c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_i64, c_int, c_vp, c_sz, c_vp]),
    "pcgc_bce_bwd": (c_int, [c_vp, c_vp, c_f32, c_f32, c_vp, c_i64, c_vp]),
    "pcgc_bce_bwd_dev": (c_int, [c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, c_vp, c_i64, c_vp]),
    "pcgc_laplace_likelihood_bwd_dev": (c_int, [c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, c_vp, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pcgc_factorized_likelihood_bwd_dev": (c_int, [c_vp, c_vp, ctypes.c_double, ctypes.c_double, c_vp, c_f32, c_vp, c_vp, c_i64, c_int, c_vp, c_sz, c_vp]),
    "pcgc_sum_log_workspace_bytes": (c_sz, []),
    "pcgc_sum_log": (c_int, [c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_train_loss_sums_workspace_bytes": (c_sz, [c_i64]),
    "pcgc_train_loss_sums": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcgc_adam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "pcgc_adam_step_guarded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp]),
}
HOST_API = {
    "pcgc_host_last_error": (ctypes.c_char_p, []),
    "pcgc_pmf_to_quantized_cdf": (c_int, [c_vp, c_i64, c_int, c_int, c_vp]),
    "pcgc_range_encode": (c_int, [c_vp, c_i64, c_int, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp]),
    "pcgc_range_encode_values": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp]),
    "pcgc_range_decode": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_int, c_int, c_int, c_vp]),
    "pcgc_range_decode_progress": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "pcgc_range_encode_lohi_batch": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_i64, c_vp, c_int]),
    "pcgc_range_decode_u16_batch": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_int]),
    "pcgc_partition": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "pcgc_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, c_vp, c_i64]),
    "pcgc_host_repro_eval": (c_int, [c_int, c_vp, c_vp, c_i64]),
    "pcgc_format_points_int": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "pcgc_parse_ply_points": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int]),
}

_hip = None
_host = None


class PcgcError(RuntimeError):
    pass


def _load(name, api):
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise PcgcError("%s is not built: run `python -m pcgcv1_amd.build` (hipcc --offload-arch=gfx950). "
                        "There is no CPU fallback for the hot path." % path)
    lib = ctypes.CDLL(path)
    for fn, (res, args) in api.items():
        f = getattr(lib, fn)          # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    return lib


def _hip_runtime_global():
    """libpcgc_hip.so carries no DT_NEEDED for the HIP runtime: bind it to the ONE runtime this process
    uses — PyTorch's bundled libamdhip64.so when torch is importable (so torch streams / allocations and
    our launches share a runtime), else /opt/rocm's."""
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    candidates += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"), "libamdhip64.so"]
    for c in candidates:
        try:
            return ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    raise PcgcError("no HIP runtime (libamdhip64.so) found for libpcgc_hip.so")


def hip():
    global _hip
    if _hip is None:
        _hip_runtime_global()
        _hip = _load("libpcgc_hip.so", HIP_API)
    return _hip


def host():
    global _host
    if _host is None:
        _host = _load("libpcgc_host.so", HOST_API)
    return _host


def check(rc, what="pcgc call"):
    if rc != 0:
        raise PcgcError("%s failed (%d): %s" % (what, rc, hip().pcgc_last_error().decode()))


def check_host(rc, what="pcgc host call"):
    if rc != 0:
        raise PcgcError("%s failed (%d): %s" % (what, rc, host().pcgc_host_last_error().decode()))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise PcgcError("no HIP device visible: the pcgcv1_amd hot path only runs on an MI355X (gfx950)")
    return torch.device("cuda", torch.cuda.current_device())


_DEBUG_DEVICE = os.environ.get("PCGC_DEBUG_DEVICE", "0") == "1"


def dptr(t):
    """Raw device/host pointer of a contiguous torch tensor (None -> NULL).  PCGC_DEBUG_DEVICE=1: a device tensor must live
    on the calling thread's CURRENT device — the library launches on the stream handed to it, and a worker thread that
    never called torch.cuda.set_device would launch on device 0's runtime state with another device's pointers (a
    one-GPU box cannot show that; the GPU suite runs once with the check on)."""
    if t is None:
        return None
    assert t.is_contiguous(), "pcgc kernels need contiguous tensors"
    if _DEBUG_DEVICE and t.is_cuda:
        import torch
        cur = torch.cuda.current_device()
        assert t.device.index == cur, "tensor on cuda:%s handed to a launch from a thread whose current device is cuda:%d" % (
            t.device.index, cur)
    return ctypes.c_void_p(t.data_ptr())


def bind_device(dev):
    """Make `dev` (torch.device / index) the calling thread's current device.  Every thread that launches kernels calls this
    first: the current device is per-thread state in HIP, a fresh worker thread starts on device 0."""
    import torch
    idx = dev.index if hasattr(dev, "index") else int(dev)
    if idx is not None and torch.cuda.current_device() != idx:
        torch.cuda.set_device(idx)


def nptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_pool = [None, None]


def workers(kind="job"):
    """Persistent Python worker threads (a fresh thread per call costs 0.1-0.4 ms before it does anything), two pools:
    "pipe" runs the host pipeline bodies of transform._run_pipes — they rendezvous on a barrier and WAIT for jobs they
    submit — and "job" runs those nested jobs (the z string coder, the progressive z decoder).  Kept apart so that pipeline
    bodies of concurrent calls can never occupy every worker their own nested jobs need (a half-started set of pipelines
    waiting for jobs that have no thread left would hang the process)."""
    i = 0 if kind == "job" else 1
    if _pool[i] is None:
        with _pool_lock:
            if _pool[i] is None:
                from concurrent.futures import ThreadPoolExecutor
                _pool[i] = ThreadPoolExecutor(max_workers=48 if i == 0 else 64, thread_name_prefix="pcgc-%s" % kind)
    return _pool[i]


_pool_lock = __import__("threading").Lock()
_streams = {}


def side_stream(role, parent):
    """The `role` stream that belongs to the stream `parent` — one per (role, parent) for the whole PROCESS, whatever codec
    (checkpoint) asks: HIP streams are multiplexed onto a fixed number of hardware queues in creation order, and a fresh
    set of streams per checkpoint would land the later checkpoints' pipeline and entropy streams on shared queues
    (measured: the third checkpoint of a bench run 8 % slower than the first)."""
    import torch
    key = (role, int(parent.cuda_stream), parent.device.index)
    st = _streams.get(key)
    if st is None:
        with _pool_lock:
            st = _streams.get(key)
            if st is None:
                st = _streams[key] = torch.cuda.Stream(device=parent.device)
    return st
_trace = None


def mark(label):
    """Timeline hook (tools/timeline2.py sets _trace): a no-op in the product."""
    if _trace is not None:
        _trace(label)


def host_threads():
    """Threads for the per-cube range coder streams.  Each stream is ~0.3 ms of work, so more than a few
    dozen threads only adds start-up cost (measured on the 256-core GPU box: 16-64 threads 1.9 ms for 205
    cubes, 256 threads 6.2 ms).  64 rather than 32: the batches on the critical path are the 50-cube first slices of
    the decoder pipelines, which 64 threads decode in one round instead of two (-0.7 ms per step, profiles/r03_vG_handover_timeline.txt)."""
    n = os.environ.get("PCGC_HOST_THREADS")
    if n:
        return max(1, int(n))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # one process per GPU: the ranks of a node share its cores (two host pipelines per rank, each coding with this pool)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    return max(1, min(64, avail // max(1, local)))
