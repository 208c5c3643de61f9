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
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak (= fp32 vector peak)
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
    ap.add_argument("--dr-stage", type=int, default=None, help="train: domain randomisation at curriculum stage 0-5 (default: off)")
    ap.add_argument("--updates-per-step", type=int, default=1, help="train: SAC updates per vector step (they run beside the "
                                                                      "acting pass on the second stream)")
    ap.add_argument("--shipped-acting", action="store_true", help="train: act like the reference under its shipped config.yaml "
                                                                  "(hierarchical goal policy + safety layer + curiosity bonus)")
    ap.add_argument("--no-overlap", action="store_true", help="train: run the update after the acting pass instead of beside it")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per K steps")
    ap.add_argument("--graph", action="store_true", help="train: capture the K steps in one hipGraph (default: eager; the host "
                                                         "stays ahead of the device, both measure the same)")
    ap.add_argument("--steps-per-launch", type=int, default=1, help="physics: T env steps per kernel launch with pre-supplied actions "
                                                                    "(tvc_env_step_many, the tests/benchmark.py:40-60 procedure); --steps must be a multiple")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample length (0 = skip)")
    ap.add_argument("--roofline-envs", type=int, default=1 << 22, help="bandwidth-regime size for the extra roofline point")
    return ap.parse_args()


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


def cpu_baseline(seconds, n_envs=0):
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

    for k in range(W):
        step_fn(k)
    torch.cuda.synchronize(device)

    graph = None
    # physics: one kernel per step -> capture the K launches in one hipGraph (removes the per-launch host cost).
    # train: ~140 launches per step; eager keeps the graph size independent of K (a 1000-step graph would hold 140k
    # nodes) and measures the same as a captured loop because the host stays ahead of the device
    use_graph = (not args.no_graph) and (workload == "physics" or (args.graph and world == 1))
    # multi-GPU train: the two gradient all-reduces (RCCL) sit between kernel phases; they are issued eagerly
    # rather than captured (collective capture is not something a 1-GPU gpurun box can validate)
    if use_graph:
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
        try:
            out.update(roofline_report(args, workload, n, step_fn if workload == "physics" else None, dev_us_per_step, device))
        except Exception as e:  # the headline line must survive a failure of the diagnostic legs
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if args.cpu_seconds > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, n if workload == "train" else 0)
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    barrier(world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def graph_time_us(fn, reps, device, rounds=5):
    """Average duration of one launch: `reps` back-to-back launches captured in a hipGraph, HIP events on the
    stream they run on, best of `rounds` (what rocprofv3's per-dispatch average agrees with, profiles/)."""
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device)
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


def pmc_traffic(n, kernel="env_step_kernel"):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, both in
    KiB; tools/pmc_to_json.py), keyed by kernel and grid size in threads; None when that size was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(kernel, {}).get(str(n), {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def integrator_roofline(n, device, us=None):
    """HBM roofline of env_step_kernel at n envs (fresh env, pre-generated actions)."""
    from tvc_ai_amd import VecRocketTVCEnv
    if us is None:
        env = VecRocketTVCEnv(n, device=device, seed=7)
        env.reset()
        acts = (torch.rand((8, n, 2), device=device) * 2 - 1).contiguous()
        for k in range(40):  # spread the envs over episode phases
            env.step(acts[k % 8])
        us = graph_time_us(lambda k: env.step(acts[k % 8]), 50, device)
        env.close()
    ach = ENV_STEP_BYTES * n / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": "env_step_kernel<W10,noDR>", "envs": n, "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(n), "launch_us": us,
            "algorithmic_bytes_per_launch": ENV_STEP_BYTES * n, "env_steps_per_s": n / (us * 1e-6)}


def roofline_report(args, workload, n, physics_step_fn, dev_us_per_step, device):
    rep = {}
    if workload == "physics":
        # the timed region holds exactly K launches of this one kernel: launch duration = HIP-event time / K
        rep["roofline"] = integrator_roofline(n, device, us=dev_us_per_step)
    else:
        # dominant kernel of the train loop = the fused Linear + residual + LayerNorm of the acting pass (M = envs;
        # the K = 512 -> N = 256 FFN-out layer is the largest single line of the kernel trace, profiles/), fp32-input
        # MFMA (v_mfma_f32_16x16x4_f32), dense peak 157.3 TFLOP/s.  Small N (< 6144 rows) runs the unfused 64x64 kernel.
        from tvc_ai_amd import _native as nat
        L = nat.load()
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
        rep["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": tf,
                           "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                           "traffic": traffic, "launch_us": us, "flops_per_launch": 2.0 * M * N * K,
                           "dtype": "f32 in / f32 acc MFMA"}
        del R, gam, bet
        del X, W, b, Y
        rep["roofline_integrator"] = integrator_roofline(n, device)
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
            macs = (10 * 256 + 3 * 256 * 256 + 4 * (2 * 256 * 512) + 256 * 512 + 512 * 512 + 512 * 4) if args.family == 0 \
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
            obs, _ = env1.reset()
            n1 = 0
            t0 = time.perf_counter()
            while n1 < 300 and time.perf_counter() - t0 < 10.0:
                a, _ = ag.get_action(torch.from_numpy(obs).unsqueeze(0))
                nobs, r, term, trunc, _ = env1.step(a.flatten())
                ag.update({"states": torch.from_numpy(obs).unsqueeze(0), "actions": torch.from_numpy(a),
                           "rewards": torch.tensor([r]), "next_states": torch.from_numpy(nobs).unsqueeze(0),
                           "dones": torch.BoolTensor([term or trunc])})
                obs = env1.reset()[0] if (term or trunc) else nobs
                n1 += 1
            rep["reference_plumbing_n1"] = {"steps_per_s": n1 / (time.perf_counter() - t0), "steps": n1,
                                            "config": "config.yaml defaults: hierarchical goal policy + safety layer + curiosity + physics-informed loss on",
                                            "note": "1 env + 1 online SAC update per step through the reference's Python surface "
                                                    "(host round trips every call); reference docs imply 35-93 steps/s (BASELINE.md)"}
            env1.close()
        except Exception as e:
            rep["reference_plumbing_n1"] = {"error": str(e)}
    try:  # the integrator in its bandwidth regime (H3: small batches are launch/latency-bound)
        rep["roofline_integrator_large_n"] = integrator_roofline(args.roofline_envs, device)
    except Exception as e:  # never lose the headline line to the optional point
        rep["roofline_integrator_large_n"] = {"error": str(e)}
    return rep


if __name__ == "__main__":
    main()
