"""Does a private slice of the chip for the learner pay at SMALL shards?  (At 65 536 envs it does not: profiles/r01_h_*.md.)
At 4 096 / 8 192 envs the step is bounded by the update (~535 us alone, ~700 us beside the acting kernel); the acting kernel needs
only 256 / 512 workgroups.  Streams are created with hipExtStreamCreateWithCUMask (a contiguous range of n mask bits = n/8 CUs in
every XCD) and handed to torch as ExternalStreams.

    python tools/cumask_shard.py [n_envs ...]
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from tvc_ai_amd.trainer import VecTrainer

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device="cuda:0")


def run(n, main_mask, side_mask, steps=300, segments=False):
    tr = VecTrainer(n, device="cuda:0", family=0, batch_size=256, replay_capacity=1_000_000, seed=42, defer_join=True,
                    distinct_window=1000)
    tr.prefill_env(1000)
    if side_mask is not None:
        tr._side = masked_stream(*side_mask)
    main = masked_stream(*main_mask) if main_mask is not None else torch.cuda.current_stream()
    with torch.cuda.stream(main):
        fn = tr.capture_segments() if segments else (lambda: tr.step(True))
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


if __name__ == "__main__":
    # one configuration per process (streams created by earlier cases would share hardware queues with later ones):
    #   cumask_shard.py <n_envs> <segments 0|1> <main lo> <main hi> <side lo> <side hi>     (-1 -1 = unmasked)
    n, seg = int(sys.argv[1]), bool(int(sys.argv[2]))
    m = [int(a) for a in sys.argv[3:7]]
    mm = None if m[0] < 0 else (m[0], m[1])
    sm = None if m[2] < 0 else (m[2], m[3])
    ms = run(n, mm, sm, segments=seg)
    print(json.dumps({"n": n, "segments": seg, "main_mask": mm, "side_mask": sm, "ms_per_step": round(ms, 4)}), flush=True)
