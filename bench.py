#!/usr/bin/env python3
"""bench.py -- env-steps/s (+ SAC updates/s) of the MI355X-native rocket-TVC hot path.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N>1: one rank per GPU over RCCL, either launched by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or -- when started directly without WORLD_SIZE --
bench.py launches its own N ranks as a child `python -m torch.distributed.run` BEFORE anything in this process touches the
GPU, relays rank 0's JSON line and exits with the children's code.  It exits non-zero when the ranks that joined differ from
--gpus.  W untimed warm-up steps, then exactly K timed steps between barrier + torch.cuda.synchronize() on both sides, MAX
over ranks, rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input:
  workload "physics" : one vector env step (N_env envs/GPU, pre-generated U(-1,1)^2 actions, auto-reset)
  workload "train"   : policy act on N_env observations -> vector env step -> replay insert ->
                       one SAC update at batch 256 (available once the learner kernels are built)
Envs shard over ranks with no data-path collective in "physics"; "train" adds the critic/actor gradient
all-reduce (RCCL) of the update.  Default scaling = weak (--envs-per-gpu fixed, 65 536 = the metric's "64k parallel envs" on
every GPU); `--total-envs T` fixes the job size instead (strong scaling: T / N envs per GPU, e.g. 65 536 -> 8 192 per GPU at
N = 8 = BASELINE configs[4]).  With N > 1 the weak line also carries a `strong_scaling` object measured in the same run.
The default train workload is BASELINE configs[4]'s environment: full domain randomisation at curriculum stage 5 with the
curriculum driver attached (`--dr-stage 0` = the shipped, un-randomised env).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Hardware queues: the train step needs the stream it is issued from, the learner's stream and (data parallel) RCCL's stream on
# DIFFERENT hardware queues; ROCm's default of 4 is shared with every other stream the process touches (tvc_ai_amd/streams.py probes
# the learner's stream; RCCL's is PyTorch's to choose).  Read by the HIP runtime when it initialises, hence before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
import torch  # noqa: E402



def _capture_stream(device):
    from tvc_ai_amd.streams import capture_stream  # one capture stream per process and device (see streams.py: hardware queues)
    return capture_stream(device)


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak (= fp32 vector peak)
# Algorithmic bytes of one env-step through the single-step kernel (DESIGN.md "K1 traffic"):
# reads  dyn 13 + prev_action 2 + aux 2 + episode 1 + reward window 10 + action 2        = 30 words
# writes dyn 13 + prev_action 2 + aux 2 + episode 1 + window slot 1 + obs 10 + reward 1  = 30 words + 2 flag bytes
# (the cell layout actually moves 128 B in + 134 B out = 262 B; PMC: profiles/pmc_traffic.json)
ENV_STEP_BYTES = 30 * 4 + 30 * 4 + 2
# domain-randomised instantiation: + the six per-episode parameters (mass / thrust scale, cg offset, wind xyz) READ each step
# (they only change at a reset, so their write-back is not algorithmic); + 8 B when the episode statistics are on (running return)
ENV_STEP_BYTES_DR = ENV_STEP_BYTES + 6 * 4


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
    ap.add_argument("--total-envs", type=int, default=0, help="strong scaling: total envs of the job, split evenly over the ranks "
                                                              "(overrides --envs-per-gpu; 65536 = BASELINE configs[4])")
    ap.add_argument("--dr-stage", type=int, default=5, help="train: domain randomisation at curriculum stage 0-5 (config.yaml:236-286, "
                                                            "340-349) with the curriculum driver attached; 0 = shipped env, no DR")
    ap.add_argument("--updates-per-step", type=int, default=1, help="train: SAC updates per vector step (they run beside the "
                                                                      "acting pass on the second stream)")
    ap.add_argument("--shipped-acting", action="store_true", help="train: act like the reference under its shipped config.yaml "
                                                                  "(hierarchical goal policy + safety layer + curiosity bonus)")
    ap.add_argument("--reward-window", type=int, default=0, choices=[0, 10, 1000],
                    help="reward-history window of MultiObjectiveReward's diversity test: 1000 = the reference's whole deque "
                         "(env/...:221; 4 KB of state per env), 10 = the approximate throughput mode; 0 = default: 1000 for the train "
                         "workload (reference semantics), 10 for the physics-only lines")
    ap.add_argument("--exact-reward", action="store_true", help="alias of --reward-window 1000")
    ap.add_argument("--share-cus", choices=["auto", "on", "off"], default="auto",
                    help="train: one acting workgroup per CU so that the update runs beside the acting pass (auto: on with the two-stream schedule)")
    ap.add_argument("--share-rows", type=int, default=-1, help="train: rows the acting kernel handles in its CU-sharing form; -1 = "
                                                               "chosen at warm-up from measured step times (max over ranks)")
    ap.add_argument("--cu-split", default="auto", help="train: CU partition of the step's two streams: auto = chosen at warm-up from "
                                                       "measured step times below 12 288 envs per GPU; 0 = none; k = mask bits [0, k) "
                                                       "for acting / env / replay, the rest for the update")
    ap.add_argument("--segments", choices=["auto", "on", "off"], default="auto",
                    help="train: replay the step as segment graphs between the update's collectives (auto: on with --gpus > 1, where "
                         "a whole-step graph cannot hold the RCCL calls)")
    ap.add_argument("--acting-dropout", action="store_true", help="train: act in train mode like the reference's get_action "
                                                                  "(Dropout(0.1) live in the policy, agent/...:765)")
    ap.add_argument("--acting-x3", action="store_true",
                    help="train: the one-launch acting kernel on the bf16 matrix pipe with three-term split operands (fp32-exact, "
                         "tvc_actor_x3.h) instead of the f32-input MFMA; without the flag the default run reports it as the extra "
                         "`acting_x3` leg")
    ap.add_argument("--no-x3-leg", action="store_true", help="skip the extra split-operand leg of the default run")
    ap.add_argument("--prefill-steps", type=int, default=1000,
                    help="train: random-action env steps before the warm-up so that the timed steps see the long-run env state (reward "
                         "histories full: from step 1000 of a run the step kernel scans a whole 1000-entry ring per env)")
    ap.add_argument("--no-shard-sizes", action="store_true", help="skip the extra 4 096 / 8 192-env legs (BASELINE's per-GPU shards)")
    ap.add_argument("--no-overlap", action="store_true", help="train: run the update after the acting pass instead of beside it")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per K steps")
    ap.add_argument("--graph", action="store_true", help="train: capture the K steps in one hipGraph (default: eager; the host "
                                                         "stays ahead of the device, both measure the same)")
    ap.add_argument("--steps-per-launch", type=int, default=1, help="physics: T env steps per kernel launch with pre-supplied actions "
                                                                    "(tvc_env_step_many, the tests/benchmark.py:40-60 procedure); --steps must be a multiple")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample length (0 = skip)")
    ap.add_argument("--loop-only", action="store_true", help="skip the roofline / diagnostic legs and the CPU baseline (profiling runs: "
                                                             "the kernel statistics then cover the timed loop only)")
    ap.add_argument("--roofline-envs", type=int, default=1 << 22, help="bandwidth-regime size for the extra roofline point")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks as a child torch.distributed.run.  Nothing in this process has touched
    the GPU (importing torch and counting devices does not initialise HIP on this image) and nothing will: it only waits."""
    import socket
    import subprocess
    forced = os.environ.get("TVC_FORCE_DEVICE") is not None  # rehearsal on a 1-GPU box: all ranks on one card over gloo
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not forced:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible; refusing to run fewer ranks than asked "
              f"(set TVC_FORCE_DEVICE=0 TVC_DIST_BACKEND=gloo to rehearse on one card)", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TVC_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks for a 1-GPU box: TVC_FORCE_DEVICE=0 puts every rank on one card, TVC_DIST_BACKEND=gloo swaps
    # RCCL (which refuses two ranks per GPU) for gloo; the launch contract (nccl, one rank per GPU) is the default
    if os.environ.get("TVC_FORCE_DEVICE") is not None:
        local = int(os.environ["TVC_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        backend = os.environ.get("TVC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local = 0
    if args.gpus != world:  # never report a line for a job other than the one asked for
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined (WORLD_SIZE={world}); aborting", file=sys.stderr)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        sys.exit(3)
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


def reward_window(args, workload):
    """the reward-history window of this run: --reward-window / --exact-reward, else 1000 (reference) for train, 10 for physics"""
    if getattr(args, "exact_reward", False):
        return 1000
    w = int(getattr(args, "reward_window", 0) or 0)
    return w if w else (1000 if workload == "train" else 10)


def cpu_baseline(seconds, n_envs=0, window=10):
    """The fp64 oracle (oracle/tvc_oracle.c, kind 'port') stepping envs on the host cores, same step
    semantics (contact, auto-reset, the same reward-history window as the GPU line), pre-sampled U(-1,1)^2 actions."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import envoracle as eo
    eo.lib()
    cores = max(1, min(os.cpu_count() or 1, 16))
    n_env, T = 64, 2000
    rng = np.random.default_rng(123)
    acts = rng.uniform(-1, 1, (T, n_env, 2)).astype(np.float32)
    vecs = [eo.OracleVec(n_env, contact=1, auto_reset=1, distinct_window=window) for _ in range(cores)]
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
    out = {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"{total} env-steps: {cores} threads x 64 envs, oracle/tvc_oracle.c fp64, "
                     f"contact+auto-reset, {dt:.1f} s (env half only; learner half below)"}
    # learner half: the eager-PyTorch restatement of _update_sac (oracle/sac_torch.py, reference shapes, B = 256)
    try:
        from oracle import sac_torch as st
        torch.set_num_threads(cores)
        g = torch.Generator().manual_seed(0)
        shapes = {"input_embedding.weight": (256, 10), "input_embedding.bias": (256,), "feature_norm.weight": (256,),
                  "feature_norm.bias": (256,), "policy_head.0.weight": (512, 256), "policy_head.0.bias": (512,),
                  "policy_head.2.weight": (512,), "policy_head.2.bias": (512,), "policy_head.4.weight": (512, 512),
                  "policy_head.4.bias": (512,), "policy_head.6.weight": (512,), "policy_head.6.bias": (512,),
                  "policy_head.8.weight": (4, 512), "policy_head.8.bias": (4,)}
        for l in range(4):
            p = f"transformer_encoder.layers.{l}."
            shapes.update({p + "self_attn.in_proj_weight": (768, 256), p + "self_attn.in_proj_bias": (768,),
                           p + "self_attn.out_proj.weight": (256, 256), p + "self_attn.out_proj.bias": (256,),
                           p + "linear1.weight": (512, 256), p + "linear1.bias": (512,), p + "linear2.weight": (256, 512),
                           p + "linear2.bias": (256,), p + "norm1.weight": (256,), p + "norm1.bias": (256,),
                           p + "norm2.weight": (256,), p + "norm2.bias": (256,)})
        mk = lambda sh: (torch.randn(sh, generator=g) / (sh[1] ** 0.5 if len(sh) > 1 else 10.0)) + (1.0 if len(sh) == 1 else 0.0)
        P = {k: mk(v) for k, v in shapes.items()}
        qs = {"0.weight": (512, 12), "0.bias": (512,), "2.weight": (512,), "2.bias": (512,), "4.weight": (256, 512),
              "4.bias": (256,), "6.weight": (256,), "6.bias": (256,), "8.weight": (1, 256), "8.bias": (1,)}
        Q1, Q2 = {k: mk(v) for k, v in qs.items()}, {k: mk(v) for k, v in qs.items()}
        orc = st.SacOracle(P, Q1, Q2)
        B = 256
        batch = (torch.randn(B, 10), torch.rand(B, 2) * 2 - 1, torch.randn(B), torch.randn(B, 10), torch.zeros(B))
        e1, e2 = torch.randn(B, 2), torch.randn(B, 2)
        for _ in range(2):
            orc.update(*batch, e1, e2)
        t0 = time.perf_counter()
        n_up = 0
        while time.perf_counter() - t0 < max(2.0, seconds * 0.4):
            orc.update(*batch, e1, e2)
            n_up += 1
        out["sac_updates_per_s"] = n_up / (time.perf_counter() - t0)
        out["sac_sample"] = f"{n_up} updates, eager PyTorch fp32 restatement, {cores} threads, B=256, reference shapes"
        # acting half: policy forward on a bounded sample of rows, then the same loop the GPU runs
        # (act on n envs -> n env-steps -> one update) assembled from the three measured CPU rates
        rows = 4096
        obs = torch.randn(rows, 10)
        with torch.no_grad():
            st.actor_forward(P, obs)
            t0 = time.perf_counter()
            n_fw = 0
            while time.perf_counter() - t0 < max(1.0, seconds * 0.2):
                st.actor_forward(P, obs)
                n_fw += 1
        act_rows_per_s = n_fw * rows / (time.perf_counter() - t0)
        out["act_rows_per_s"] = act_rows_per_s
        if n_envs:
            t_step = n_envs / out["value"] + n_envs / act_rows_per_s + 1.0 / out["sac_updates_per_s"]
            out["env_only_value"] = out["value"]
            out["value"] = n_envs / t_step
            out["sample"] += (f"; train-loop value = {n_envs} envs / (env {n_envs / out['env_only_value'] * 1e3:.1f} ms + act "
                              f"{n_envs / act_rows_per_s * 1e3:.1f} ms + update {1e3 / out['sac_updates_per_s']:.1f} ms) per vector step, "
                              f"policy forward timed on {rows} rows")
    except Exception as e:
        out["sac_updates_per_s"] = None
        out["sac_sample"] = f"failed: {e}"
    return out


def timed_steps(step_fn, K, W, world, device, use_graph):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides; max over ranks.
    -> (seconds, device microseconds per step from HIP events on the launch stream, captured?)"""
    for k in range(W):
        step_fn(k)
    torch.cuda.synchronize(device)
    graph = None
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            side = _capture_stream(device)
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
    return dt, ev0.elapsed_time(ev1) * 1e3 / K, graph is not None


def allreduce_report(tr, world, device):
    """What the data-parallel update exchanges, and how long the two collectives take on their own (measured here,
    back to back on the gradient buffers themselves, max over ranks)."""
    sac = tr.sac
    sizes = [int((sac.grads.numel() - sac.n_policy) * 4), int(sac.n_policy * 4)]
    rep = {"calls_per_update": 2, "bytes_per_update": sum(sizes), "critic_bytes": sizes[0], "actor_bytes": sizes[1],
           "op": "all_reduce(sum) over the flat fp32 gradient slices; 1/world folded into the Adam kernel"}
    if world == 1:
        rep["measured_us_per_update"] = None
        rep["note"] = "single rank: no collective is issued"
        return rep
    import torch.distributed as dist
    bufs = [torch.zeros_like(sac.grads[sac.n_policy:]), torch.zeros_like(sac.grads[:sac.n_policy])]
    for _ in range(5):
        for b in bufs:
            dist.all_reduce(b)
    torch.cuda.synchronize(device)
    dist.barrier()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        for b in bufs:
            dist.all_reduce(b)
    torch.cuda.synchronize(device)
    us = max_over_ranks((time.perf_counter() - t0) / reps * 1e6, world, device)
    rep["measured_us_per_update"] = us
    rep["bus_gbps"] = 2.0 * (world - 1) / world * sum(sizes) / (us * 1e-6) / 1e9
    rep["calls_counted_in_timed_region"] = tr.sync.calls if tr.sync is not None else 0
    return rep


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world, rank, local = dist_setup(args)
    device = torch.device("cuda", local)
    from tvc_ai_amd import VecRocketTVCEnv
    workload = args.workload
    if workload in ("auto", "train"):
        from tvc_ai_amd import trainer  # noqa: F401  (fails loudly: the HIP library is the product)
        workload = "train"

    strong = args.total_envs > 0
    if strong and args.total_envs % world:
        raise SystemExit(f"--total-envs {args.total_envs} does not divide over {world} ranks")
    n = args.total_envs // world if strong else args.envs_per_gpu
    K, W = args.steps, args.warmup
    backend = (os.environ.get("TVC_DIST_BACKEND", "nccl") if world > 1 else None)
    extra = {}
    tr = None
    if workload == "train":
        result = trainer.bench_train(args, world, rank, device, n_envs=n)
        step_fn, tr = result["step_fn"], result["trainer"]
        extra.update(result.get("extra", {}))
    else:
        env = VecRocketTVCEnv(n, device=device, seed=42, env_id_offset=rank * n)
        env.reset()
        g = torch.Generator(device=device).manual_seed(1000 + rank)
        n_act = 64
        acts = (torch.rand((n_act, n, 2), device=device, generator=g) * 2 - 1).contiguous()

        T = max(1, args.steps_per_launch)
        if T > 1:  # state stays in registers across T steps of one launch; outputs of all T steps are written
            if K % T or W % T:
                raise SystemExit("--steps and --warmup must be multiples of --steps-per-launch")
            outs = (torch.empty((T, n, 10), device=device), torch.empty((T, n), device=device),
                    torch.empty((T, n), dtype=torch.uint8, device=device), torch.empty((T, n), dtype=torch.uint8, device=device))
            acts_t = acts[:T].contiguous() if T <= n_act else (torch.rand((T, n, 2), device=device, generator=g) * 2 - 1).contiguous()

            def step_fn(k):
                if k % T == 0:
                    env.step_many(acts_t, out=outs)
            extra["steps_per_launch"] = T
        else:
            def step_fn(k):
                env.step(acts[k % n_act])

    # physics: one kernel per step -> capture the K launches in one hipGraph (removes the per-launch host cost).
    # train: ~140 launches per step; eager keeps the graph size independent of K (a 1000-step graph would hold 140k
    # nodes) and measures the same as a captured loop because the host stays ahead of the device.
    # multi-GPU train: the two gradient all-reduces (RCCL) sit between kernel phases; they are issued eagerly
    # rather than captured (collective capture is not something a 1-GPU gpurun box can validate)
    use_graph = (not args.no_graph) and (workload == "physics" or (args.graph and world == 1))
    launch_mode = "hipGraph of K steps" if use_graph else "eager launches"
    if workload == "train" and not use_graph and (args.segments == "on" or (args.segments == "auto" and world > 1)) \
            and tr.updates_per_step == 1:
        seg = tr.capture_segments()  # 4 graph launches + the 2 collectives per step instead of ~130 kernel launches
        step_fn = lambda k: seg()
        launch_mode = "segment graphs between the update's collectives (4 graph launches per step)"
    if workload == "train" and use_graph and n >= 32768 and tr.overlap:
        # the two streams' branches of ONE graph do not overlap at this size (3.8 ms per step at 32 768 envs against 1.10 eager,
        # DESIGN.md 10.11): the whole-step graph is a small-shard mode; what is captured here is the sequential schedule
        tr.overlap = False
        launch_mode += ", sequential schedule (>= 32 768 envs)"
    dt, dev_us_per_step, captured = timed_steps(step_fn, K, W, world, device, use_graph)

    env_steps = float(n) * world * K
    mode = (f"strong scaling: {args.total_envs} envs in total = {n} per GPU" if strong
            else f"weak scaling: {n} envs per GPU (the metric's 64k parallel envs on every GPU)")
    out = {
        "metric": "env-steps/sec + SAC updates/sec at 64k parallel envs, 1/2/4/8 MI355X",
        "value": env_steps / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32" if not getattr(args, "acting_x3", False) else "f32 (acting pass: fp32 operands split in three bf16 terms, six products on the "
                                                                       "bf16 matrix pipe, fp32 accumulate; update: f32-input MFMA)",
        "data": "synthetic",
        "world_size": world,
        "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else "none (single rank)",
        "config": {"workload": f"{workload}: {n} envs/GPU x {world} GPU(s) = {n * world} envs ({mode}), contact + auto-reset, "
                               f"reward-history window {reward_window(args, workload)}"
                               f"{' (the reference deque, env/...:221)' if reward_window(args, workload) == 1000 else ' (approximate throughput mode)'}, "
                               f"{launch_mode if (captured or not use_graph) else 'eager launches'}",
                   "envs_per_gpu": n, "total_envs": n * world, "reward_window": reward_window(args, workload)},
    }
    if workload == "train":
        out["sac_updates_per_s"] = extra.pop("updates_per_step", 1.0) * K * 1.0 / dt
        out["allreduce"] = allreduce_report(tr, world, device)
        if tr.curriculum is not None:
            extra["curriculum"] = {"stage_index": tr.curriculum.current_stage_idx,
                                   "stage": getattr(tr.curriculum.get_current_stage(), "name", None),
                                   "evaluations_in_run": len(tr.curriculum_log),
                                   "last_evaluation": tr.curriculum_log[-1] if tr.curriculum_log else None}
    out.update(extra)

    # weak line at N > 1: also measure the fixed-job-size variant of BASELINE configs[3-4] (65 536 envs over the N GPUs)
    if workload == "train" and world > 1 and not strong and (n * world) != 65536 and 65536 % world == 0:
        try:
            n2 = 65536 // world
            tr.close()
            res2 = trainer.bench_train(args, world, rank, device, n_envs=n2)
            dt2, _, _ = timed_steps(res2["step_fn"], K, W, world, device, False)
            out["strong_scaling"] = {"total_envs": 65536, "envs_per_gpu": n2, "value": 65536.0 * K / dt2, "unit": "env-steps/s",
                                     "ms_per_step": dt2 / K * 1e3, "sac_updates_per_s": res2["trainer"].updates_per_step * K / dt2,
                                     "note": "same run, same K / W, job size fixed at 65 536 envs (BASELINE configs[4])"}
            tr = res2["trainer"]
        except Exception as e:
            out["strong_scaling"] = {"error": f"{type(e).__name__}: {e}"}

    if workload == "train" and world == 1 and not args.loop_only and not args.no_shard_sizes:
        # BASELINE configs[2-4] put 4 096 - 8 192 envs on a GPU: the same loop at those sizes (they bound the strong-scaling curve)
        try:
            tr.close()
            tr = None
            out["shard_sizes"] = shard_sizes(args, device)
        except Exception as e:
            out["shard_sizes"] = {"error": f"{type(e).__name__}: {e}"}

    if workload == "train" and world == 1 and not args.loop_only and not args.no_x3_leg and not args.acting_x3 \
            and args.family == 0 and n >= 16384 and not args.acting_dropout and not args.shipped_acting:
        # the same loop with the acting pass on the bf16 matrix pipe (split operands, fp32-exact): reported NEXT TO the headline, which
        # keeps the f32-input MFMA kernel
        try:
            if tr is not None:
                tr.close()
                tr = None
            out["acting_x3"] = x3_leg(args, device, n, K, W)
        except Exception as e:
            out["acting_x3"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        try:
            if not args.loop_only:
                out.update(roofline_report(args, workload, n, step_fn if workload == "physics" else None, dev_us_per_step, device,
                                           extras=world == 1))
        except Exception as e:  # the headline line must survive a failure of the diagnostic legs
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if args.cpu_seconds > 0 and world == 1 and not args.loop_only:  # rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, n if workload == "train" else 0, window=reward_window(args, workload))
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    barrier(world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def x3_leg(args, device, n, K, W):
    """The headline workload again with VecTrainer(acting_x3=True): every Linear of the acting pass on v_mfma_f32_16x16x32_bf16 with both
    operands split in three bf16 terms and six products accumulated in fp32 (tvc_actor_x3.h).  Same K / W, same tuning steps."""
    import copy
    from tvc_ai_amd import trainer
    a2 = copy.copy(args)
    a2.acting_x3 = True
    res = trainer.bench_train(a2, 1, 0, device, n_envs=n)
    t = res["trainer"]
    dt, _, _ = timed_steps(res["step_fn"], K, W, 1, device, False)
    rep = {"value": n * K / dt, "unit": "env-steps/s", "ms_per_step": dt / K * 1e3, "sac_updates_per_s": t.updates_per_step * K / dt,
           "steps": K, "warmup": W, "share_rows": t.share_rows,
           "arithmetic": "v_mfma_f32_16x16x32_bf16, operands x = x_h + x_m + x_l (bf16 each, 24 mantissa bits in all), products hh hm mh hl lh mm, "
                         "fp32 accumulate; error against an fp64 evaluation equal to the f32-input MFMA kernel's (tests/test_acting_x3_gpu.py)",
           "kernel": x3_kernel_roofline(n, device)}
    t.close()
    # ... and acting in TRAIN mode like the reference's get_action (Dropout live, attention unfolded: 1.9x the MFMA work) with the same
    # arithmetic (actor_x3_kernel<true>), next to the f32 train-mode kernel (actor_split_kernel<true>): the reference-semantics figures
    try:
        tm = {}
        for name, x3 in (("f32_mfma", False), ("split_operands", True)):
            a3 = copy.copy(args)
            a3.acting_x3, a3.acting_dropout = x3, True
            r3 = trainer.bench_train(a3, 1, 0, device, n_envs=n)
            dt3, _, _ = timed_steps(r3["step_fn"], min(K, 200), min(W, 30), 1, device, False)
            k3 = min(K, 200)
            tm[name] = {"value": n * k3 / dt3, "ms_per_step": dt3 / k3 * 1e3, "sac_updates_per_s": r3["trainer"].updates_per_step * k3 / dt3}
            r3["trainer"].close()
        rep["train_mode_acting"] = {**tm, "unit": "env-steps/s", "note": "VecTrainer(acting_dropout=True): tvc_sac_act flags bit 3 (+ bit 4), "
                                                                        "masks element for element those of oracle/sac_torch.py: DropMasks"}
    except Exception as e:
        rep["train_mode_acting"] = {"error": f"{type(e).__name__}: {e}"}
    return rep


BF16_MFMA_PEAK_TF = 2500.0  # dense, guide


def x3_kernel_roofline(n, device):
    """isolated launch of actor_x3_kernel: algorithmic flops (3.29 MFLOP per row, as for the f32 kernel) / time against the roof of THIS
    arithmetic, the bf16 dense peak / 6 (every algorithmic multiply-add is six bf16 ones); also the executed bf16 rate"""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    sacr = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n), device=device, seed=2)
    ob, ep = torch.randn(n, 10, device=device), torch.randn(n, 2, device=device)
    outs = tuple(torch.empty(n, 2, device=device) for _ in range(3))
    us = graph_time_us(lambda k: sacr.act(ob, ep, out=outs, x3=True), 5, device)
    sacr.close()
    macs = 16 * 256 + 3 * 256 * 256 + 4 * (2 * 256 * 512) + 256 * 512 + 512 * 512 + 512 * 4
    flops = 2.0 * macs * n
    tf = flops / (us * 1e-6) / 1e12
    traffic = pmc_traffic(((n + 63) // 64) * 256, "tvcnn::actor_x3_kernel")
    t_stale = STALE.get("traffic") if traffic else None
    loop = in_loop_us("tvcnn::actor_x3_kernel", f"{n}_x3")
    return {"bound": "mfma", "kernel": "tvcnn::actor_x3_kernel", "achieved": tf, "peak": BF16_MFMA_PEAK_TF / 6.0, "unit": "TFLOP/s",
            "frac": tf / (BF16_MFMA_PEAK_TF / 6.0), "traffic": traffic,
            "traffic_source": f"{PMC_FILE} (committed rocprofv3 --pmc pass, not measured in this run)" if traffic else None,
            "traffic_stale": t_stale, "in_loop_us": loop, "in_loop_stale": STALE.get("in_loop") if loop else None,
            "in_loop_frac": (flops / (loop * 1e-6) / 1e12 / (BF16_MFMA_PEAK_TF / 6.0)) if loop else None,
            "launch_us": us, "flops_per_launch": flops, "flops_per_row": 2.0 * macs,
            "executed_bf16_tflops": 6.0 * tf, "bf16_dense_peak_tflops": BF16_MFMA_PEAK_TF,
            "times_the_f32_mfma_peak": tf / MFMA_F32_PEAK_TF,
            "note": "achieved = algorithmic flops / time; peak = bf16 dense peak / 6 products; launch time live (hipGraph of 5 isolated launches)"}


def shard_sizes(args, device, sizes=(4096, 8192), K=200, W=30):
    """ms per step / SAC updates per s of the train workload at BASELINE's per-GPU shard sizes, single rank, in the launch mode a
    multi-rank job uses (segment graphs) and as eager launches."""
    from tvc_ai_amd import trainer
    rep = {}
    for n in sizes:
        ent = {}
        for mode in ("segment_graphs", "eager"):
            res = trainer.bench_train(args, 1, 0, device, n_envs=n)
            t = res["trainer"]
            fn = res["step_fn"]
            if mode == "segment_graphs":
                seg = t.capture_segments()
                fn = lambda k: seg()
            dt = min(timed_steps(fn, K, W, 1, device, False)[0] for _ in range(2))  # (a latency-bound loop of tiny kernels can catch
            ent[mode] = {"ms_per_step": dt / K * 1e3, "sac_updates_per_s": K / dt, "env_steps_per_s": n * K / dt}  # the chip in a low clock state: best of two)
            if t.share_tuning is not None:
                ent["share_rows"] = t.share_rows
            ent["main_stream_cu_mask_bits"] = t.cu_split
            t.close()
        best = min(("segment_graphs", "eager"), key=lambda m: ent[m]["ms_per_step"])
        rep[str(n)] = {"ms_per_step": ent[best]["ms_per_step"], "sac_updates_per_s": ent[best]["sac_updates_per_s"],
                       "env_steps_per_s": ent[best]["env_steps_per_s"], "mode": best, **ent}
    rep["note"] = f"same workload as the headline line (DR stage, reward window, update in the loop), {K} timed steps after {W}"
    return rep


def graph_time_us(fn, reps, device, rounds=5):
    """Average duration of one launch: `reps` back-to-back launches captured in a hipGraph, HIP events on the
    stream they run on, best of `rounds` (what rocprofv3's per-dispatch average agrees with, profiles/)."""
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    side = _capture_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(reps):
                fn(k)
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize(device)
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


PMC_FILE = "profiles/pmc_traffic.json"
LOOP_STATS_FILE = "profiles/r03_train_loop_kernel_stats.json"
STALE = {}  # what pmc_traffic() / in_loop_us() last answered -> was it taken at another build of the library than the one running


def _lib_sha():
    from tvc_ai_amd.build import sources_sha256
    try:
        return sources_sha256()
    except Exception:
        return None


def pmc_traffic(n, kernel="env_step_kernel"):
    """HBM bytes per launch from the COMMITTED rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, both in
    KiB; tools/pmc_to_json.py), keyed by kernel and grid size in threads; None when that size was not profiled.
    Not measured in this run: the JSON line labels it with `traffic_source`, and with `traffic_stale` when the pass was taken at
    other library sources than the ones running (sha256 stored by the tool)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            ent = json.load(f).get(kernel, {}).get(str(n), {})
        STALE["traffic"] = ent.get("lib_sources_sha256") != _lib_sha() if ent else None
        return ent.get("hbm_bytes_per_launch")
    except Exception:
        STALE["traffic"] = None
        return None


def in_loop_us(kernel, envs):
    """rocprofv3 --kernel-trace --stats average of `kernel` INSIDE the train loop (committed summary of this same command,
    tools/loop_stats_to_json.py), next to the isolated launch time measured live; None when not profiled at this size;
    `in_loop_stale` says whether it was taken at other library sources than the ones running."""
    try:
        with open(os.path.join(ROOT, LOOP_STATS_FILE)) as f:
            d = json.load(f).get(str(envs), {})
        STALE["in_loop"] = d.get("_lib_sources_sha256") != _lib_sha() if d else None
        return d.get(kernel)
    except Exception:
        STALE["in_loop"] = None
        return None


def integrator_roofline(n, device, us=None, dr_stage=None, stats=False, window=10):
    """HBM roofline of env_step_kernel at n envs (fresh env, pre-generated actions); dr_stage > 0 = the domain-randomised
    instantiation the default train loop runs; window = 1000: the reference-exact reward history (the step then reads the env's
    whole 1000-entry ring once to keep the distinct count: 990 more words of algorithmic traffic per env-step)."""
    from tvc_ai_amd import VecRocketTVCEnv
    dr = dr_stage is not None and dr_stage > 0
    if us is None:
        over = {}
        if dr:
            from tvc_ai_amd.env import dr_from_yaml
            over = dr_from_yaml({}, dr_stage)
        if window == 1000:
            over["distinct_window"] = 1000
        env = VecRocketTVCEnv(n, device=device, seed=7, **over)
        if stats:
            env.enable_episode_stats()
        env.reset()
        acts = (torch.rand((8, n, 2), device=device) * 2 - 1).contiguous()
        for k in range(40):  # spread the envs over episode phases
            env.step(acts[k % 8])
        us = graph_time_us(lambda k: env.step(acts[k % 8]), 50, device)
        env.close()
    per_env = (ENV_STEP_BYTES_DR if dr else ENV_STEP_BYTES) + (8 if stats else 0) + (990 * 4 if window == 1000 else 0)
    ach = per_env * n / (us * 1e-6) / 1e9
    kname = ("env_step_kernel<W1000," if window == 1000 else "env_step_kernel<W10,") + ("DR>" if dr else "noDR>")
    traffic = pmc_traffic(n, ("env_step_kernel_w1000_dr" if dr else None) if window == 1000
                          else ("env_step_kernel_dr" if dr else "env_step_kernel"))
    note = {}
    if window == 1000:  # 4 KB of ring per env: 268 MB at 65 536 envs, re-read by back-to-back launches mostly from the Infinity Cache
        note = {"note": "exact reward history: the 1000-entry ring of every env is read once per step; measured with back-to-back "
                        "launches, where the ring (4 KB x envs) is largely served by the 256 MB Infinity Cache, so `achieved` can "
                        "approach or exceed what HBM alone delivers; HBM traffic of this instantiation was not profiled"}
    return {"bound": "hbm", "kernel": kname, "envs": n, "achieved": ach, "peak": HBM_PEAK_GBS, **note,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": f"{PMC_FILE} (committed rocprofv3 --pmc pass, not measured in this run)" if traffic else None,
            "traffic_stale": STALE.get("traffic") if traffic else None,
            "launch_us": us, "algorithmic_bytes_per_env_step": per_env,
            "algorithmic_bytes_per_launch": per_env * n, "env_steps_per_s": n / (us * 1e-6)}


def _layer_kernel_roofline(L, n, device):
    """Row counts below the one-launch kernel's threshold (or the MLP family): the dominant kernel is the fused Linear +
    residual + LayerNorm of the per-layer acting pass (K = 512 -> N = 256, M = envs); < 6144 rows the unfused 64x64 kernel."""
    rep = {}
    M, N, K = n, 256, 512
    X = torch.randn(M, K, device=device)
    W = torch.randn(N, K, device=device) / 16
    b = torch.zeros(N, device=device)
    R = torch.randn(M, N, device=device)
    gam, bet = torch.ones(N, device=device), torch.zeros(N, device=device)
    Y = torch.empty(M, N, device=device)
    fused = M >= 6144
    if fused:
        kname = "tvcnn::gemm_rowln_kernel<4, 32> (Linear 512->256 + residual + LayerNorm, M = envs)"
        us = graph_time_us(lambda k: L.tvc_nn_linear_ln_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(),
                                                                gam.data_ptr(), bet.data_ptr(), Y.data_ptr(), M, N, K, 0,
                                                                torch.cuda.current_stream(device).cuda_stream), 20, device)
        traffic = pmc_traffic(((M + 31) // 32) * 256, "tvcnn::gemm_rowln_kernel<4, 32>")
    else:
        kname = "tvcnn::gemm_kernel<true,true> (Linear 512->256, M = envs)"
        us = graph_time_us(lambda k: L.tvc_nn_linear_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, N, K, 0,
                                                             0, torch.cuda.current_stream(device).cuda_stream), 20, device)
        traffic = pmc_traffic((N // 64) * ((M + 63) // 64) * 256, "tvcnn::gemm_kernel")
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
    loop = in_loop_us(kname.split(" (")[0], n)
    rep["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": tf,
                       "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                       "traffic": traffic,
                       "traffic_source": f"{PMC_FILE} (committed rocprofv3 --pmc pass, not measured in this run)" if traffic else None,
                       "traffic_stale": STALE.get("traffic") if traffic else None,
                       "launch_us": us, "launch_us_source": "live: hipGraph of 20 isolated launches, HIP events on their stream",
                       "in_loop_us": loop, "in_loop_stale": STALE.get("in_loop") if loop else None,
                       "in_loop_source": f"{LOOP_STATS_FILE} (committed rocprofv3 --kernel-trace --stats of the train loop)" if loop else None,
                       "in_loop_frac": (2.0 * M * N * K / (loop * 1e-6) / 1e12 / MFMA_F32_PEAK_TF) if loop else None,
                       "flops_per_launch": 2.0 * M * N * K, "dtype": "f32 in / f32 acc MFMA"}
    return rep


def roofline_report(args, workload, n, physics_step_fn, dev_us_per_step, device, extras=True):
    """`roofline` of the dominant kernel (+ the integrator's); extras = the diagnostic legs (learner alone, acting alone, hierarchical
    path, N = 1 plumbing, integrator at 4 M envs): rank 0 of a single-rank run only, the other ranks of a multi-GPU job would wait."""
    rep = {}
    if workload == "physics":
        # the timed region holds exactly K launches of this one kernel: launch duration = HIP-event time / K
        rep["roofline"] = integrator_roofline(n, device, us=dev_us_per_step)
    else:
        from tvc_ai_amd import _native as nat
        L = nat.load()
        # the whole acting pass as one launch: from 16 384 rows the row-owner kernel (csrc/tvc_actor_rows.h: 64 rows per workgroup),
        # from 1 024 rows its split sibling (csrc/tvc_actor_split.h: 16 rows per workgroup, the four waves split every Linear)
        rows_kernel = args.family == 0 and n >= 1024
        if rows_kernel and getattr(args, "acting_x3", False) and n >= 16384:
            rep["roofline"] = x3_kernel_roofline(n, device)
        elif rows_kernel:
            # dominant kernel of the train loop = the WHOLE acting pass as one launch: every Linear of the policy on
            # v_mfma_f32_16x16x4_f32 (dense f32 peak 157.3 TFLOP/s), activations in registers, weights streamed through LDS.
            # flops per launch = 2 x MACs per row (as executed: embedding folded into layer 0, attention folded to one 256x256
            # Linear per layer) x rows
            from tvc_ai_amd.agent import NativeSAC, sac_cfg
            sacr = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n), device=device, seed=2)
            ob, ep = torch.randn(n, 10, device=device), torch.randn(n, 2, device=device)
            outs = tuple(torch.empty(n, 2, device=device) for _ in range(3))
            us = graph_time_us(lambda k: sacr.act(ob, ep, out=outs), 5, device)
            macs = 16 * 256 + 3 * 256 * 256 + 4 * (2 * 256 * 512) + 256 * 512 + 512 * 512 + 512 * 4
            flops = 2.0 * macs * n
            kshort = "tvcnn::actor_rows_kernel" if n >= 16384 else "tvcnn::actor_split_kernel"
            kname = kshort + " (policy forward of all rows in one launch: 4 encoder layers + head + sample)"
            import ctypes as C
            clk = (C.c_double * 3)()
            clock = None
            if n >= 16384 and L.tvc_debug_rows_clock(sacr._h, ob.data_ptr(), n, 20, clk, torch.cuda.current_stream(device).cuda_stream) == 0:
                clock = {"in_kernel_mhz": clk[0], "median_workgroup_lifetime_us": clk[1], "workgroups": int(clk[2]),
                         "f32_mfma_peak_at_that_clock_tflops": clk[0] * 1e6 * 64 * 4 * 256 / 1e12}
            traffic = pmc_traffic(((n + 63) // 64) * 256 if n >= 16384 else ((n + 15) // 16) * 256, kshort)
            sacr.close()
            del ob, ep, outs
            tf = flops / (us * 1e-6) / 1e12
            loop = in_loop_us(kshort, n)
            rep["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": tf / MFMA_F32_PEAK_TF, "traffic": traffic,
                               "traffic_source": f"{PMC_FILE} (committed rocprofv3 --pmc pass, not measured in this run)" if traffic else None,
                               "traffic_stale": STALE.get("traffic") if traffic else None,
                               "launch_us": us, "launch_us_source": "live: hipGraph of 5 isolated launches, HIP events on their stream",
                               "in_loop_us": loop, "in_loop_stale": STALE.get("in_loop") if loop else None,
                               "in_loop_source": (f"{LOOP_STATS_FILE} (committed rocprofv3 --kernel-trace --stats of the train loop; per-step "
                                                  "total of the loop's TWO launches of n/2 rows: one beside the learner at 1 workgroup "
                                                  "per CU, one alone)") if loop else None,
                               "in_loop_frac": (flops / (loop * 1e-6) / 1e12 / MFMA_F32_PEAK_TF) if loop else None,
                               "flops_per_launch": flops, "flops_per_row": 2.0 * macs, "clock": clock, "dtype": "f32 in / f32 acc MFMA"}
        else:
            rep.update(_layer_kernel_roofline(L, n, device))
        rep["roofline_integrator"] = integrator_roofline(n, device, dr_stage=args.dr_stage, stats=args.dr_stage > 0,
                                                         window=reward_window(args, workload))
        if not extras:
            return rep
        # learner alone (no env stepping): back-to-back SAC updates at B = 256 on a fixed batch
        try:
            from tvc_ai_amd.agent import NativeSAC, sac_cfg
            sac = NativeSAC(sac_cfg(args.family, batch_size=256, max_act_rows=256, dropout_p=0.1 if args.family == 0 else 0.0),
                            device=device, seed=1)
            B = 256
            bt = (torch.randn(B, 10, device=device), torch.rand(B, 2, device=device) * 2 - 1, torch.randn(B, device=device),
                  torch.randn(B, 10, device=device), torch.zeros(B, device=device), torch.randn(B, 2, device=device),
                  torch.randn(B, 2, device=device))
            us_up = graph_time_us(lambda k: sac.update(*bt), 20, device)
            flops = 4.88e9 if args.family == 0 else 0.565e9  # SURVEY 8d algorithmic minimum per update
            rep["sac_learner_only"] = {"updates_per_s": 1e6 / us_up, "us_per_update": us_up, "batch": B,
                                       "mfma_tflops": flops / (us_up * 1e-6) / 1e12,
                                       "dropout_in_update": 0.1 if args.family == 0 else 0.0,
                                       "note": "latency-bound at batch 256 (H5): ~100 dependent launches per update"}
            sac.close()
        except Exception as e:
            rep["sac_learner_only"] = {"error": str(e)}
        # the whole acting pass alone (policy forward on all N observations + Gaussian sample): MFMA utilisation of the
        # 12 GEMMs together.  MACs per row as executed: 10*256 (embedding folded with layer 0's attention), 3 x 256*256 (folded
        # attention of layers 1-3), 4 x (256*512 + 512*256), head 256*512 + 512*512 + 512*4 (family 1: 10*256 + 256*256 + 256*4)
        try:
            from tvc_ai_amd.agent import NativeSAC, sac_cfg
            sac = NativeSAC(sac_cfg(args.family, batch_size=256, max_act_rows=n), device=device, seed=2)
            ob, ep = torch.randn(n, 10, device=device), torch.randn(n, 2, device=device)
            outs = tuple(torch.empty(n, 2, device=device) for _ in range(3))
            us_a = graph_time_us(lambda k: sac.act(ob, ep, out=outs), 5, device)
            macs = (16 * 256 + 3 * 256 * 256 + 4 * (2 * 256 * 512) + 256 * 512 + 512 * 512 + 512 * 4) if args.family == 0 \
                else (10 * 256 + 256 * 256 + 256 * 4)
            tf_a = 2.0 * macs * n / (us_a * 1e-6) / 1e12
            rep["acting_pass_only"] = {"us_per_call": us_a, "rows": n, "rows_per_s": n / (us_a * 1e-6), "mfma_tflops": tf_a,
                                       "frac_of_f32_mfma_peak": tf_a / MFMA_F32_PEAK_TF, "flops_per_row": 2.0 * macs}
            sac.close()
            del ob, ep, outs
        except Exception as e:
            rep["acting_pass_only"] = {"error": str(e)}
        # the reference's default acting path (hierarchical_rl.enabled, agent/...:751-754) for all N envs per call
        try:
            from tvc_ai_amd.hierarchical import HierarchicalPolicy
            hp = HierarchicalPolicy(10, 2, device=device, max_rows=n, seed=3)
            ob = torch.randn(n, 10, device=device)
            ep, uu = torch.randn(n, 2, device=device), torch.rand(n, device=device)
            us_h = graph_time_us(lambda k: hp.act(ob, ep, uu), 5, device)
            rep["hierarchical_acting"] = {"rows_per_s": n / (us_h * 1e-6), "us_per_call": us_h, "rows": n,
                                          "note": "goal policy + categorical draw + goal-conditioned low-level policy (with SE block)"}
            hp.close()
            del ob, ep, uu
        except Exception as e:
            rep["hierarchical_acting"] = {"error": str(e)}
        # BASELINE configs[0] through the drop-in surface: ONE env (EnhancedRocketTVCEnv wrapper) + MultiAlgorithmAgent,
        # the loop of scripts/train.py:535-620 (get_action -> step -> B=1 update with a BoolTensor `dones`), numpy in/out
        try:
            from tvc_ai_amd import EnhancedRocketTVCEnv
            from tvc_ai_amd.agent import MultiAlgorithmAgent
            env1 = EnhancedRocketTVCEnv(enable_curiosity=True, device=device)
            ag = MultiAlgorithmAgent(10, 2, {"tvc_native": {"batch_size": 1, "max_act_rows": 16},
                                             "physics_informed": {"enabled": True}, "hierarchical_rl": {"enabled": True},
                                             "safety": {"safety_layer": {"enabled": True}}}, device=device)
            def loop(algorithm, budget):
                obs, _ = env1.reset()
                n1 = skipped = 0
                t0 = time.perf_counter()
                while n1 < 200 and time.perf_counter() - t0 < budget:
                    a, _ = ag.get_action(torch.from_numpy(obs).unsqueeze(0), algorithm=algorithm)
                    nobs, r, term, trunc, _ = env1.step(a.flatten())
                    out = ag.update({"states": torch.from_numpy(obs).unsqueeze(0), "actions": torch.from_numpy(a),
                                     "rewards": torch.tensor([r]), "next_states": torch.from_numpy(nobs).unsqueeze(0),
                                     "dones": torch.BoolTensor([term or trunc])}, algorithm=algorithm)
                    skipped += 0 if out else 1  # like the reference, update() logs and swallows an exception (agent/...:908-912)
                    obs = env1.reset()[0] if (term or trunc) else nobs
                    n1 += 1
                return n1 / (time.perf_counter() - t0), n1, skipped
            import logging
            ag.logger.setLevel(logging.CRITICAL)  # the simplified B = 1 PPO update (agent/...:914-948) can diverge; counted below instead
            sps_ref, n_ref, skip_ref = loop(None, 6.0)   # scripts/train.py: select_algorithm() -> 'ppo' (eager pass-through update)
            sps_sac, n_sac, skip_sac = loop("sac", 6.0)  # the same loop pinned to the accelerated learner
            rep["reference_plumbing_n1"] = {"steps_per_s": sps_sac, "steps": n_sac, "updates_skipped": skip_sac,
                                            "steps_per_s_default_selection_ppo_passthrough": sps_ref,
                                            "updates_skipped_default_selection": skip_ref,
                                            "config": "config.yaml defaults: hierarchical goal policy + safety layer + curiosity + physics-informed loss on",
                                            "note": "1 env + 1 online update per step through the reference's Python surface (host round "
                                                    "trips every call); steps_per_s = algorithm 'sac' (HIP learner), the other figure = the "
                                                    "reference's own default selection, 'ppo', through the eager pass-through; reference "
                                                    "docs imply 35-93 steps/s (BASELINE.md)"}
            env1.close()
        except Exception as e:
            rep["reference_plumbing_n1"] = {"error": str(e)}
    try:  # the integrator in its bandwidth regime (H3: small batches are launch/latency-bound)
        rep["roofline_integrator_large_n"] = integrator_roofline(args.roofline_envs, device)
        if workload == "train" and args.dr_stage > 0:
            rep["roofline_integrator_large_n_dr"] = integrator_roofline(args.roofline_envs, device, dr_stage=args.dr_stage, stats=True)
    except Exception as e:  # never lose the headline line to the optional point
        rep["roofline_integrator_large_n"] = {"error": str(e)}
    return rep


if __name__ == "__main__":
    main()
