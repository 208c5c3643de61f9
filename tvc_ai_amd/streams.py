"""The few HIP streams this package uses, ONE of each kind per device and process.

ROCm multiplexes HIP streams onto a small number of hardware queues (4 by default, GPU_MAX_HW_QUEUES), and `torch.cuda.Stream()`
hands out a different pooled stream on every call (32 per priority).  Measured (tools/stream_count.py, 4 096 envs): with the
default 4 queues, a HIGH-priority learner stream and four or more other streams of one priority touched by anything in the process,
the two-stream train step collapses from 0.71 to 3.5 - 3.6 ms (bench.py's fifth trainer of a process hit it; a normal-priority
learner stream, or GPU_MAX_HW_QUEUES >= 8, never did).  So:
  * every stream is created once and reused (a process that builds many trainers / captures does not walk through the pool);
  * bench.py (and this package's __init__, when the HIP runtime is not up yet) ask for 16 hardware queues (the partitioned schedule adds CU-masked streams);
  * VecTrainer.tune_learner_stream() times a few steps with the high- and the normal-priority learner stream and keeps the faster
    (bench.py calls it at warm-up): whatever else shares the process, the collapse cannot survive into the measured steps."""
import os

import torch

_cache = {}


def _key(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def learner_stream(device, priority=None) -> "torch.cuda.Stream":
    """stream the SAC update runs on beside the acting pass (VecTrainer); high priority unless TVC_SIDE_PRIORITY / `priority` says
    otherwise; one per (device, priority)"""
    prio = int(os.environ.get("TVC_SIDE_PRIORITY", "-1")) if priority is None else int(priority)
    k = ("learner", _key(device), prio)
    if k not in _cache:
        _cache[k] = torch.cuda.Stream(torch.device("cuda", k[1]), priority=prio)
    return _cache[k]


def masked_stream(device, lo: int, hi: int) -> "torch.cuda.Stream":
    """A stream whose kernels run only on the compute units of mask bits [lo, hi) (hipExtStreamCreateWithCUMask).  On MI355X a
    CONTIGUOUS range of n bits owns n / 8 CUs in each of the 8 XCDs (bits 0-31 = 4 CUs per XCD); non-contiguous masks are ignored
    by the runtime (probed with tvc_debug_hwid, profiles/r01_h_learner_update_timeline.md).  One stream per (device, lo, hi)."""
    import ctypes
    k = ("masked", _key(device), int(lo), int(hi))
    if k in _cache:
        return _cache[k]
    dev = torch.device("cuda", k[1])
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    if not (0 <= lo < hi <= n_cu):
        raise ValueError(f"CU mask bits [{lo}, {hi}) outside [0, {n_cu})")
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime PyTorch already loaded
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    words = (ctypes.c_uint32 * ((n_cu + 31) // 32))()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), words)
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    _cache[k] = torch.cuda.ExternalStream(h.value, device=dev)
    return _cache[k]


def capture_stream(device) -> "torch.cuda.Stream":
    """stream hipGraph captures are recorded on (VecTrainer.capture / capture_segments, bench.graph_time_us)"""
    k = ("capture", _key(device))
    if k not in _cache:
        _cache[k] = torch.cuda.Stream(torch.device("cuda", k[1]))
    return _cache[k]
