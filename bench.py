#!/usr/bin/env python3
"""bench.py -- env-steps/s (+ SAC updates/s) of the MI355X-native rocket-TVC hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run, one
rank per GPU).  W untimed warm-up steps, then exactly K timed steps between barrier +
torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input:
  workload "physics" : one vector env step (N_env envs/GPU, pre-generated U(-1,1)^2 actions, auto-reset)
  workload "train"   : policy act on N_env observations -> vector env step -> replay insert ->
                       one SAC update at batch 256 (available once the learner kernels are built)
Envs shard over ranks with no data-path collective in "physics"; "train" adds the critic/actor gradient
all-reduce (RCCL) of the update.  scaling = weak (envs per GPU fixed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
# Algorithmic bytes of one env-step through the single-step kernel (DESIGN.md "K1 traffic"):
# reads  dyn 13 + prev_action 2 + aux 2 + episode 1 + reward window 10 + action 2        = 30 words
# writes dyn 13 + prev_action 2 + aux 2 + episode 1 + window slot 1 + obs 10 + reward 1  = 30 words + 2 flag bytes
# (the 16-byte-cell layout actually moves 136 B in + 142 B out = 278 B; PMC-confirmed, profiles/r01_b_*)
ENV_STEP_BYTES = 30 * 4 + 30 * 4 + 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=65536,
                    help="64k parallel envs (the configuration BASELINE.json's metric is quoted on; fits one GPU). "
                         "8192 = the per-GPU shard of configs[4] (65 536 envs on 8 GPUs)")
    ap.add_argument("--workload", choices=["physics", "train", "auto"], default="auto")
    ap.add_argument("--family", type=int, default=0, help="SAC network family: 0 = reference shapes, 1 = 256x256 MLP")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per K steps")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample length (0 = skip)")
    ap.add_argument("--roofline-envs", type=int, default=1 << 22, help="bandwidth-regime size for the extra roofline point")
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        local = 0
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    return world, rank, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world, device):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def kernel_event_times(fn_step, n_launch, device):
    """Average duration of ONE launch, HIP events on the stream the kernel is launched on."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launch)]
    torch.cuda.synchronize(device)
    for k in range(n_launch):
        evs[k][0].record()
        fn_step(k)
        evs[k][1].record()
    torch.cuda.synchronize(device)
    ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3  # us
    return float(np.mean(ts)), float(np.median(ts)), float(np.min(ts))


def cpu_baseline(seconds):
    """The fp64 oracle (oracle/tvc_oracle.c, kind 'port') stepping envs on the host cores, same step
    semantics (contact, auto-reset, 10-entry diversity window), pre-sampled U(-1,1)^2 actions."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import envoracle as eo
    eo.lib()
    cores = max(1, min(os.cpu_count() or 1, 16))
    n_env, T = 64, 2000
    rng = np.random.default_rng(123)
    acts = rng.uniform(-1, 1, (T, n_env, 2)).astype(np.float32)
    vecs = [eo.OracleVec(n_env, contact=1, auto_reset=1, distinct_window=10) for _ in range(cores)]
    deadline = time.perf_counter() + seconds

    def work(v):
        done = 0
        while time.perf_counter() < deadline:
            s, _ = v.run(acts)  # ctypes releases the GIL for the C loop
            done += s
        return done
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, vecs))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{total} env-steps: {cores} threads x 64 envs, oracle/tvc_oracle.c fp64, "
                      f"contact+auto-reset, {dt:.1f} s"}


def main():
    args = parse()
    world, rank, local = dist_setup(args)
    device = torch.device("cuda", local)
    from tvc_ai_amd import VecRocketTVCEnv
    workload = args.workload
    have_agent = False
    if workload in ("auto", "train"):
        try:
            from tvc_ai_amd import trainer as _trainer  # noqa: F401
            have_agent = True
        except ImportError:
            have_agent = False
        if workload == "train" and not have_agent:
            raise SystemExit("workload 'train' needs tvc_ai_amd.trainer")
        workload = "train" if have_agent else "physics"

    n = args.envs_per_gpu
    K, W = args.steps, args.warmup
    extra = {}
    if workload == "train":
        from tvc_ai_amd import trainer
        result = trainer.bench_train(args, world, rank, device)
        step_fn, sync_extra = result["step_fn"], result
        extra.update(result.get("extra", {}))
        env = result["env"]
    else:
        env = VecRocketTVCEnv(n, device=device, seed=42, env_id_offset=rank * n)
        env.reset()
        g = torch.Generator(device=device).manual_seed(1000 + rank)
        n_act = 64
        acts = (torch.rand((n_act, n, 2), device=device, generator=g) * 2 - 1).contiguous()

        def step_fn(k):
            env.step(acts[k % n_act])

    for k in range(W):
        step_fn(k)
    torch.cuda.synchronize(device)

    graph = None
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for k in range(K):
                        step_fn(k)
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
        except Exception as e:  # e.g. a collective that cannot be captured: fall back to eager launches
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize(device)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier(world)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ev0.record()  # HIP events on the stream the kernels are launched on (torch's current stream)
    if graph is not None:
        graph.replay()
    else:
        for k in range(K):
            step_fn(k)
    ev1.record()
    torch.cuda.synchronize(device)
    barrier(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, device)
    dev_us_per_step = ev0.elapsed_time(ev1) * 1e3 / K

    env_steps = float(n) * world * K
    out = {
        "metric": "env-steps/sec + SAC updates/sec at 64k parallel envs, 1/2/4/8 MI355X",
        "value": env_steps / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{workload}: {n} envs/GPU x {world} GPU(s) = {n * world} envs "
                               f"(64k parallel envs per GPU, weak scaling), contact + auto-reset, "
                               f"{'hipGraph of K steps' if graph is not None else 'eager launches'}",
                   "envs_per_gpu": n, "total_envs": n * world},
    }
    if workload == "train":
        out["sac_updates_per_s"] = extra.pop("updates_per_step", 1.0) * K * 1.0 / dt
    out.update(extra)

    if rank == 0:
        # ---- roofline of the dominant kernel: env_step_kernel (HBM-bound: ~2.7 flop/B)
        if workload == "physics":
            # the timed region holds exactly K launches of this one kernel: average launch duration =
            # HIP-event time of the region / K (agrees with rocprofv3's per-dispatch average, profiles/)
            mean_us = med_us = dev_us_per_step
        else:
            mean_us, med_us, _ = kernel_event_times(extra_env_only(env, device), 200, device)
        ach = ENV_STEP_BYTES * n / (mean_us * 1e-6) / 1e9
        pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            pmc = pj.get(str(n), {}).get("hbm_bytes_per_launch")
        except Exception:
            pmc = None
        out["roofline"] = {"bound": "hbm", "kernel": "env_step_kernel<W10>", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc,
                           "launch_us_mean": mean_us, "launch_us_median": med_us,
                           "algorithmic_bytes_per_launch": ENV_STEP_BYTES * n}
        # the same kernel in its bandwidth regime (H3: at 8 192 envs one launch moves 2 MB and is latency-bound)
        try:
            nb = args.roofline_envs
            big = VecRocketTVCEnv(nb, device=device, seed=7)
            big.reset()
            ab = (torch.rand((4, nb, 2), device=device) * 2 - 1).contiguous()
            for k in range(10):
                big.step(ab[k % 4])
            m2, md2, mn2 = kernel_event_times(lambda k: big.step(ab[k % 4]), 50, device)
            ach2 = ENV_STEP_BYTES * nb / (m2 * 1e-6) / 1e9
            out["roofline_large_n"] = {"envs": nb, "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach2 / HBM_PEAK_GBS, "launch_us_mean": m2,
                                       "env_steps_per_s": nb / (m2 * 1e-6)}
            big.close()
            del big, ab
        except Exception as e:  # never lose the headline line to the optional point
            out["roofline_large_n"] = {"error": str(e)}
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    barrier(world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def extra_env_only(env, device):
    n = env.num_envs
    acts = (torch.rand((8, n, 2), device=device) * 2 - 1).contiguous()
    return lambda k: env.step(acts[k % 8])


if __name__ == "__main__":
    main()
