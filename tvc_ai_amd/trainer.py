"""Vectorised training loop over the HIP env and SAC kernels: the counterpart of
StateOfTheArtTrainer.run_episode (scripts/train.py:535-620) for N envs per GPU.

One ``step()`` = policy act on N observations -> vector env step -> replay insert (true next observation,
done = terminated|truncated) -> [uniform sample of B rows -> SAC update] x updates_per_step.  Everything is enqueued
without host synchronisation (the update on a second HIP stream beside the acting pass), so K steps can also be
captured in one hipGraph.  Options mirror the reference's training env / agent switches: curiosity bonus, the
hierarchical acting path, the safety layer, train-mode dropout in the update; ``save_checkpoint`` / ``load_checkpoint``
resume a run exactly.
"""
import time
from typing import Optional

import os

import torch

from .streams import capture_stream, learner_stream, masked_stream

from .agent import NativeSAC, ReplayBuffer, dropout_seed_of, sac_cfg
from .env import VecRocketTVCEnv
from .parallel import GradSync, broadcast_parameters


class VecTrainer:
    def __init__(self, num_envs: int, device="cuda:0", config: Optional[dict] = None, family: int = 0, batch_size: int = 256,
                 replay_capacity: int = 1_000_000, seed: int = 42, rank: int = 0, world: int = 1, updates_per_step: int = 1,
                 max_episode_steps: int = 1000, enable_curiosity: bool = False, overlap: bool = True,
                 enable_hierarchical: bool = False, enable_safety: bool = False, dropout_p: Optional[float] = None,
                 share_cus: Optional[bool] = None, defer_join: bool = False, share_rows: Optional[int] = None,
                 acting_dropout: bool = False, acting_x3: bool = False, **env_over):
        self.device = torch.device(device)
        self.n, self.B, self.world, self.rank = num_envs, batch_size, world, rank
        self.updates_per_step = updates_per_step
        self.env = VecRocketTVCEnv(num_envs, device=self.device, config=config, max_episode_steps=max_episode_steps,
                                   seed=seed, env_id_offset=rank * num_envs, want_final_obs=True, **env_over)
        # the reference's update runs in train mode (Dropout(0.1) active in the policy and the critics); family 1 has none
        self.dropout_p = (0.1 if family == 0 else 0.0) if dropout_p is None else float(dropout_p)
        self.sac = NativeSAC(sac_cfg(family, batch_size=batch_size, max_act_rows=num_envs, dropout_p=self.dropout_p,
                                     dropout_seed=dropout_seed_of(seed, rank)),  # replicas draw different masks on their batches
                             device=self.device, seed=seed)
        broadcast_parameters(self.sac.params)  # identical replicas (rank 0's initialisation)
        self.sac.sync_derived()
        self.rb = ReplayBuffer(replay_capacity, 10, 2, device=self.device, seed=seed * 1000003 + rank)
        self.sync = GradSync() if world > 1 else None
        self.overlap = overlap
        # defer_join = True: a step returns without waiting for its update on the learner's stream (the next step waits before it
        # takes its snapshot); whoever reads learner state in between synchronises the device first (state_dict() does)
        self.defer_join = bool(defer_join) and bool(overlap)
        # acting kernel placement (tvc_sac_act flags bit 2): alone on the chip it is fastest with two workgroups per CU, which
        # own every register of the CU -- the update then runs AFTER the acting pass.  One workgroup per CU is ~25 % slower
        # alone but lets the update's kernels run BESIDE it, which wins at every update count (65 536 envs: 2.49 vs 2.58 ms per
        # step at 1 update per step, 2.74 vs 3.40 at 2, 4.18 vs 4.92 at 4; profiles/r02_f_bench_matrix.md).
        self.share_cus = bool(overlap) if share_cus is None else bool(share_cus)
        # ... and only for as many rows as the update needs company: with ONE update per step the first half of the rows is
        # acted on in the sharing form (2 rounds of 16 384 rows, 1.18 ms at 65 536 envs: the update runs beside them), the second
        # half in the exclusive form (one round of 512 workgroups, 0.95 ms): 2.13 ms instead of 2.36 (all shared) or 1.89 + the
        # update (all exclusive).  From two updates per step on, every row is shared.
        self.share_rows = num_envs
        if self.share_cus and updates_per_step <= 1 and num_envs >= 32768 and (num_envs // 2) % 64 == 0:
            self.share_rows = num_envs // 2
        if acting_x3 and family == 0 and updates_per_step <= 1 and num_envs >= 16384:
            # the split-operand kernel at one workgroup per CU (one wave per SIMD) is in its bad regime: by default every row takes the
            # exclusive form and the update follows it (65 536 envs: 1.69 ms with 0 shared rows, 1.66 with 16 384, 2.02 with 32 768)
            self.share_rows = 0
        if share_rows is not None:  # explicit split (bench: chosen at warm-up by tune_share_rows, identically on every rank)
            self.share_rows = max(0, min(int(share_rows), num_envs))
        self.share_tuning = None
        self.stream_tuning = None
        # the learner's stream: high HIP priority + raised wave priority inside its kernels (TVC_LEARNER_PRIO): -4 % on the step at
        # 2 and 4 updates per step, the update ends ~0.2 ms earlier at 1 (tools/ab_prio.sh)
        self._side = self._learner = learner_stream(self.device)  # (one per process and device: streams.py)
        self._main = None        # set_cu_split(): the step's own (CU-masked) stream instead of the caller's current stream
        self.cu_split = 0
        self.cu_tuning = None
        self._fork = torch.cuda.Event()
        # the host may not run more than this many steps ahead of the device (0 = unbounded): a loop that never reads anything back
        # would otherwise fill the launch queues, and a full queue is waited on far less efficiently than an event
        self.max_steps_in_flight = int(os.environ.get("TVC_STEPS_IN_FLIGHT", "4"))
        self._inflight = [torch.cuda.Event() for _ in range(max(0, self.max_steps_in_flight))]
        self._side_done = None  # timing event at the end of the learner's stream (tune_share_rows)
        # acting_dropout = True: act in train mode like the reference's get_action (agent/...:765): Dropout live in the policy
        # (attention not folded; one launch from 1 024 rows: actor_split_kernel<true>); False = the deterministic folded net
        self.acting_dropout = bool(acting_dropout) and self.dropout_p > 0.0
        # acting_x3 = True: the one-launch acting kernel (>= 16 384 rows) runs on the bf16 matrix pipe with three-term split operands,
        # fp32-exact (tvc_actor_x3.h, tvc_sac_act flags bit 4); its weight stream is packed now, before any update can run beside it
        # (with acting_dropout: the train-mode instantiation of the same kernel on the stream of the net as trained, flags bits 3 + 4)
        self.acting_x3 = bool(acting_x3) and family == 0
        if self.acting_x3 and not self.acting_dropout:
            self.sac.enable_x3()
        # intrinsic curiosity bonus of the reference's training env (scripts/train.py:318, env/...:496-502)
        self.curiosity = None
        if enable_curiosity:
            from .curiosity import VecCuriosity
            self.curiosity = VecCuriosity(device=self.device, max_rows=num_envs, seed=seed)
        # acting path of the shipped config.yaml (agent/...:751-754, 785-789): the never-trained hierarchical policy picks
        # the action, the safety layer corrects it; the SAC policy is still what the updates train
        self.hier = self.safety = None
        if enable_hierarchical:
            from .hierarchical import HierarchicalPolicy
            self.hier = HierarchicalPolicy(10, 2, device=self.device, max_rows=num_envs, seed=seed + 11)
            self.u_goal = torch.empty((num_envs,), device=self.device)
        if enable_safety:
            from .curiosity import SafetyLayer
            self.safety = SafetyLayer(device=self.device, max_rows=num_envs, seed=seed + 12)
            self.act_raw = torch.empty((num_envs, 2), device=self.device)
        d, n, B = self.device, num_envs, batch_size
        self.obs = [torch.empty((n, 10), device=d), torch.empty((n, 10), device=d)]
        self.cur = 0
        self.act = torch.empty((n, 2), device=d)
        if self.acting_x3 and self.acting_dropout and n >= 16384:
            # pack the train-mode streams now, before an update can run beside the first acting call (one throw-away call on zeros)
            z = torch.zeros((n, 10), device=d)
            self.sac.act(z, None, train_mode=True, x3=True)
            del z
        self.mean = torch.empty((n, 2), device=d)
        self.ls = torch.empty((n, 2), device=d)
        self.eps_act = torch.empty((n, 2), device=d)
        K = max(1, updates_per_step)
        # two sets of update inputs (batches + noise), alternating by step parity: the overlapped schedule draws the inputs of step
        # t + 1 while the update of step t may still be reading its own
        self._sets = [{"eps1": [torch.empty((B, 2), device=d) for _ in range(K)], "eps2": [torch.empty((B, 2), device=d) for _ in range(K)],
                       "batches": [(torch.empty((B, 10), device=d), torch.empty((B, 2), device=d), torch.empty((B,), device=d),
                                    torch.empty((B, 10), device=d), torch.empty((B,), device=d)) for _ in range(K)]} for _ in range(2)]
        self.eps1, self.eps2, self.batches = self._sets[0]["eps1"], self._sets[0]["eps2"], self._sets[0]["batches"]
        self.batch = self.batches[0]
        self._snapshot = False  # collect() acts with the policy snapshot (overlapped schedule)
        self.curriculum = None
        self.prev_done = torch.ones((n,), dtype=torch.uint8, device=d)  # 1 = next step is the first of an episode
        torch.manual_seed(seed + rank)  # default CUDA generator: hipGraph-capturable normal draws
        o, _ = self.env.reset()
        self.obs[0].copy_(o)
        self.steps = 0

    def close(self):
        torch.cuda.synchronize(self.device)  # the learner's stream may still be running the last update
        self.env.close()
        self.sac.close()
        self.rb.close()
        if self.curiosity is not None:
            self.curiosity.close()
        if self.hier is not None:
            self.hier.close()
        if self.safety is not None:
            self.safety.close()

    # -- curriculum (scripts/train.py:458-460 calls the manager once per episode on the host and drops its answer, SURVEY
    #    F11; here the driver reads device-side episode statistics every `every` vector steps WITHOUT stalling the loop: the
    #    counters are copied to pinned memory behind an event and consumed one period later)
    def attach_curriculum(self, driver, every: int = 100, min_episodes: int = 50):
        self.curriculum = driver
        driver.env = self.env
        driver._apply()
        self._cur_every, self._cur_min_eps = int(every), int(min_episodes)
        self.env.enable_episode_stats()
        self._cur_host = torch.zeros(4, dtype=torch.float64).pin_memory()
        self._cur_event = torch.cuda.Event()
        self._cur_pending = False
        self._cur_req_step, self._cur_lag = 0, max(1, min(8, int(every) // 2))
        self._cur_last = [0.0, 0.0, 0.0, 0.0]
        self._cur_next = (self.steps // self._cur_every + 1) * self._cur_every  # next request: when `steps` reaches this multiple
        self.curriculum_log = []

    def _curriculum_tick(self):
        # Read-back of a request made `lag` steps ago: a FIXED step distance (not "whenever the copy has landed"), so that with data
        # parallelism every rank evaluates, and changes stage, at the same step; the copy landed long ago, the wait costs nothing.
        if self._cur_pending and self.steps >= self._cur_req_step + self._cur_lag:
            self._cur_event.synchronize()
            tot = self._cur_host.tolist()
            d = [a - b for a, b in zip(tot, self._cur_last)]
            if d[0] >= self._cur_min_eps:  # enough finished episodes for an evaluation (the reference evaluates 50, :success_criteria)
                self._cur_last = tot
                metrics = {"eval_success_rate": d[1] / d[0], "eval_reward_mean": d[2] / d[0], "eval_length_mean": d[3] / d[0]}
                before = self.curriculum.current_stage_idx
                self.curriculum.update(self.steps * self.n * self.world, metrics)
                self.curriculum_log.append({"step": self.steps, "episodes": d[0], **metrics, "stage_before": before,
                                            "stage_after": self.curriculum.current_stage_idx})
            self._cur_pending = False
        # a request is due when `steps` has CROSSED a multiple of `every` (a replayed graph advances it by several steps at once)
        if not self._cur_pending and self.steps >= self._cur_next:
            self._cur_next = (self.steps // self._cur_every + 1) * self._cur_every
            tot = self.env.episode_stats_tensor()
            if self.world > 1:  # every rank's driver sees the whole job's episodes, so all ranks change stage together
                import torch.distributed as dist
                dist.all_reduce(tot)
            self._cur_host.copy_(tot, non_blocking=True)
            self._cur_event.record()
            self._cur_pending = True
            self._cur_req_step = self.steps

    def prefill_env(self, steps: int = 1000, seed: int = 1234):
        """Bring the env population to its long-run state before a measurement: `steps` env steps with uniform random actions
        (no acting pass, no learner, nothing inserted), which spreads the envs over episode phases and FILLS the 1000-entry
        reward histories (they persist across episodes, so from step 1000 of any run every step scans a full ring).  The current
        observation buffer ends up holding the envs' current observations."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        acts = (torch.rand((8, self.n, 2), device=self.device, generator=g) * 2 - 1).contiguous()
        cur = self.obs[self.cur]
        for k in range(int(steps)):
            _, _, term, trunc, _ = self.env.step(acts[k % 8], out_obs=cur)
        if steps > 0:
            torch.bitwise_or(term, trunc, out=self.prev_done)

    def collect(self):
        """act + env step + replay insert"""
        cur, nxt = self.obs[self.cur], self.obs[1 - self.cur]
        self.eps_act.normal_()
        raw = self.act_raw if self.safety is not None else self.act
        # (with a CU partition the acting kernel has its slice of the chip to itself: exclusive form)
        share = self.share_cus and self._snapshot and self.share_rows > 0 and self._main is None
        if self.hier is not None:
            self.u_goal.uniform_()
            # (the never-trained hierarchy has no snapshot to read; the update runs beside it)
            a, _, _, _ = self.hier.act(cur, self.eps_act, self.u_goal, clamp=self.safety is None,
                                       share_rows=self.share_rows if share else 0, x3=self.acting_x3)
            raw.copy_(a)
        else:
            k = self.share_rows if share else self.n
            if 0 < k < self.n:  # two launches over row blocks: [0, k) leaves room for the update, [k, n) takes the whole chip
                for lo, hi, sh in ((0, k, True), (k, self.n, False)):
                    self.sac.act(cur[lo:hi], self.eps_act[lo:hi], out=(raw[lo:hi], self.mean[lo:hi], self.ls[lo:hi]),
                                 clamp=self.safety is None, snapshot=self._snapshot, share_cus=sh, train_mode=self.acting_dropout,
                                 x3=self.acting_x3)
            else:
                self.sac.act(cur, self.eps_act, out=(raw, self.mean, self.ls), clamp=self.safety is None, snapshot=self._snapshot,
                             share_cus=share, train_mode=self.acting_dropout, x3=self.acting_x3)
        if self.safety is not None:  # sees the unclamped sample; clamps its result
            self.safety.apply(cur, raw, out=self.act)
        o, rew, term, trunc, info = self.env.step(self.act, out_obs=nxt)
        if self.curiosity is not None:  # added after the env's clip, skipped on the first step of an episode
            self.curiosity.add_intrinsic_reward(cur, self.act, info["final_observation"], rew, self.prev_done)
            torch.bitwise_or(term, trunc, out=self.prev_done)
        self.rb.insert(cur, self.act, rew, info["final_observation"], term, trunc)
        self.cur = 1 - self.cur

    def learn(self, k: int = 0):
        self.rb.sample(self.B, out=self.batches[k])
        self.eps1[k].normal_()
        self.eps2[k].normal_()
        s, a, r, s2, d = self.batches[k]
        gs = self.sync.grad_scale if self.sync is not None else 1.0
        return self.sac.update(s, a, r, s2, d, self.eps1[k], self.eps2[k], all_reduce=self.sync, grad_scale=gs)

    def _throttle(self):
        """wait (on the host) for the step issued max_steps_in_flight steps ago; call _throttle_mark() at the end of the step"""
        if self._inflight and self.steps >= len(self._inflight) and not torch.cuda.is_current_stream_capturing():
            self._inflight[self.steps % len(self._inflight)].synchronize()

    def _throttle_mark(self):
        if self._inflight and not torch.cuda.is_current_stream_capturing():
            self._inflight[self.steps % len(self._inflight)].record(torch.cuda.current_stream(self.device))

    def set_cu_split(self, main_bits: int):
        """Partition the chip between the two streams of the step: acting pass / env step / replay on the compute units of mask bits
        [0, main_bits), the SAC update on the rest (0 = no partition: the caller's stream and the learner stream, whole chip).
        At BASELINE's per-GPU shard sizes the step is bound by the update, whose ~90 latency-bound kernels take 1.3x their solo
        time beside the acting kernel's 256 - 512 long-lived workgroups; with a slice of its own the update runs undisturbed and
        the acting kernel still finishes inside it (4 096 envs: 0.72 -> 0.65 ms per step with 96 / 160, 8 192: 0.79 -> 0.68 with
        128 / 128; tools/cumask_shard.py, profiles/r03_h_cu_split.md).  At 65 536 envs it loses (round 1).  Call between steps."""
        torch.cuda.synchronize(self.device)
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        k = int(main_bits)
        if k <= 0:
            self._main, self.cu_split = None, 0
            self._side = self._learner
            return
        if k % 8 or not (8 <= k <= n_cu - 8):
            raise ValueError(f"set_cu_split: a multiple of 8 in [8, {n_cu - 8}] (8 mask bits = one CU per XCD)")
        self._main = masked_stream(self.device, 0, k)
        self._side = masked_stream(self.device, k, n_cu)
        self.cu_split = k

    def _on_main(self, fn):
        """run fn() on the step's own stream (when partitioned), ordered after / before the caller's stream"""
        if self._main is None or torch.cuda.is_current_stream_capturing():
            return fn()
        caller = torch.cuda.current_stream(self.device)
        # hipExtStreamCreateWithCUMask makes BLOCKING streams: the legacy default stream already orders itself against them (and an
        # explicit wait enqueued on it would drag the learner's stream into that ordering every step: 1.6 instead of 0.65 ms per
        # step at 4 096 envs).  Only a caller on a stream of its own needs the explicit edges.
        explicit = caller.cuda_stream != 0
        if explicit:
            self._main.wait_stream(caller)
        with torch.cuda.stream(self._main):
            out = fn()
        if explicit:
            caller.wait_stream(self._main)
        return out

    def step(self, learn: bool = True):
        self._throttle()
        self._on_main(lambda: self._step(learn))

    def _step(self, learn: bool = True):
        if learn and self.steps > 0:
            self._step_pipelined(two_streams=self.overlap)
        else:  # first step (empty replay) / pure collection
            if self.defer_join:
                torch.cuda.current_stream(self.device).wait_stream(self._side)
            self.collect()
            if learn:
                for k in range(self.updates_per_step):
                    self.learn(k)
        self._throttle_mark()
        self.steps += 1
        if self.curriculum is not None and not torch.cuda.is_current_stream_capturing():
            self._curriculum_tick()  # host-side bookkeeping (event query, pinned read-back): not part of a captured graph

    def _draw_update_inputs(self, cur):
        for k in range(self.updates_per_step):
            self.rb.sample(self.B, out=cur["batches"][k])
            cur["eps1"][k].normal_()
            cur["eps2"][k].normal_()

    def _step_pipelined(self, two_streams: bool = True):
        """One train step in the order every schedule shares: draw the step's batches and noise, act + env step + replay insert,
        updates_per_step SAC updates on the batches drawn FIRST (no read of a row that is being overwritten).
        two_streams: the acting pass (large GEMMs over all envs) reads a SNAPSHOT of the policy taken at the start of the step,
        so the whole update -- gradient phases (hundreds of latency-bound batch-256 kernels), the RCCL all-reduces when data
        parallel, and both Adam steps -- runs beside it on the side stream; the two streams meet once per step.  The policy that
        acts during step t is the one left by step t-1 either way, so the sequential form (two_streams = False: live parameters,
        everything on the current stream) does the same arithmetic on the same inputs; tests compare the two."""
        main = torch.cuda.current_stream(self.device)
        sac = self.sac
        cur = self._sets[self.steps & 1]  # this step's update inputs; the previous update may still be reading the other set
        batches, eps1, eps2 = cur["batches"], cur["eps1"], cur["eps2"]
        self._draw_update_inputs(cur)
        gs = self.sync.grad_scale if self.sync is not None else 1.0
        if not two_streams:
            self.collect()
            for k in range(self.updates_per_step):
                s, a, r, s2, d = batches[k]
                sac.update(s, a, r, s2, d, eps1[k], eps2[k], all_reduce=self.sync, grad_scale=gs)
            return
        side = self._side
        # defer_join: the streams meet HERE, not at the end of the previous step: the tail of the previous update (it needs a little
        # longer than the acting launch it runs beside) overlaps the env step, the replay insert and the draws above
        main.wait_stream(side)
        sac.snapshot_policy()
        self._fork.record(main)
        # the acting pass is enqueued FIRST: the host needs hundreds of microseconds to enqueue the ~100 learner launches
        # of an update, and the GPU would otherwise sit idle on the main stream for that long at small env counts
        self._snapshot = True
        try:
            self.collect()
        finally:
            self._snapshot = False
        side.wait_event(self._fork)
        with torch.cuda.stream(side):
            for k in range(self.updates_per_step):
                s, a, r, s2, d = batches[k]
                sac.update(s, a, r, s2, d, eps1[k], eps2[k], all_reduce=self.sync, grad_scale=gs)
            if self._side_done is not None:
                self._side_done.record(side)
        if not self.defer_join or torch.cuda.is_current_stream_capturing():
            main.wait_stream(side)  # (a captured graph must end with every forked stream joined)

    def _check_capturable(self):
        if self.curriculum is not None and not int(self.env.cfg.dr_enabled):
            # a stage change reaches replayed launches through the device-resident DR record (tvc_env_set_dr_async), but the
            # <DR> instantiation of the step kernel is chosen at launch: the driver always applies dr_enabled = 1
            raise RuntimeError("capture with a curriculum attached needs the domain-randomised env (dr_enabled = 1) at capture time")

    def capture(self, steps_per_replay: int = 2):
        """Capture `steps_per_replay` whole train steps (both streams, RNG draws included) in ONE hipGraph and return a function that
        replays it: at small env counts the ~130 launches of a step cost the host more than the device (4 096 envs: 0.77 ms per step
        replayed, 0.80 - 0.93 ms launched eagerly, profiles/r02_f_bench_matrix.md).  Single-rank only (a data-parallel update has
        two collectives inside: capture_segments()).  With a curriculum attached the driver's host-side bookkeeping runs after
        every replay; a stage change reaches the replayed step kernels through the device-resident DR record."""
        if self.world > 1:
            raise RuntimeError("VecTrainer.capture: single-rank only (use capture_segments() with data parallelism)")
        if int(steps_per_replay) % 2:
            raise ValueError("VecTrainer.capture: an even number of steps per replay (the observation double buffer flips every step)")
        self._check_capturable()
        if self.steps < 2:  # the pipelined schedule starts at the second step; warm every kernel up before capturing
            for _ in range(2 - self.steps):
                self.step(True)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        cap = capture_stream(self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        steps0 = self.steps
        # From 32 768 envs the two streams' branches of ONE graph do not overlap the way two live streams do (measured: 3.8 ms per
        # step at 32 768 envs, 4.9 at 65 536, against 1.10 / 1.94 eager; 0.87 at 16 384 is fine): there the SEQUENTIAL schedule is
        # captured -- the same arithmetic on the same inputs (1.41 ms at 32 768 envs); eager steps and capture_segments() keep overlapping
        sequential = self.overlap and self.n >= 32768
        if sequential:
            self.overlap = False
        try:
            with torch.cuda.stream(cap):
                with torch.cuda.graph(graph, stream=cap):
                    for _ in range(int(steps_per_replay)):
                        self.step(True)
        finally:
            if sequential:
                self.overlap = True
        torch.cuda.current_stream(self.device).wait_stream(cap)
        self.steps = steps0  # capturing executed nothing

        def replay():
            graph.replay()
            self.steps += int(steps_per_replay)
            if self.curriculum is not None:
                self._curriculum_tick()

        replay.graph = graph
        return replay

    def capture_segments(self):
        """The step as SEGMENT graphs between the collectives of a data-parallel update, valid at any world size:
            main stream : A = [policy snapshot, act, env step, replay insert, draw the NEXT step's batches and noise]
            side stream : B1 = [critic_grads]  -> all_reduce(critic slice) ->  B2 = [critic_apply, actor_grads]
                          -> all_reduce(actor slice) ->  B3 = [actor_apply (Adam, fold, pack, Polyak)]
        replayed around the two eager all_reduce calls (none at world 1): 4 graph launches + 2 collectives per step instead of ~130
        kernel launches.  Drawing the next step's batches at the END of A is the same arithmetic as drawing them at the start of
        the next step (both sit between the two steps' replay inserts), and lets the side stream start the update the moment the
        step begins.  Two graph sets alternate (observation double buffer, double-buffered update inputs).  Returns step_fn()."""
        if self.updates_per_step != 1:
            raise ValueError("capture_segments: one update per step")
        self._check_capturable()
        while self.steps < 2:
            self.step(True)
        torch.cuda.synchronize(self.device)
        sac, dev = self.sac, self.device
        gs = self.sync.grad_scale if self.sync is not None else 1.0
        # the eager loop has drawn nothing ahead: draw the inputs of the next step now, as graph A will from here on
        self._draw_update_inputs(self._sets[self.steps & 1])
        torch.cuda.synchronize(dev)
        cap = capture_stream(dev)
        sets = []
        steps0, cur0 = self.steps, self.cur
        for par in range(2):
            st = steps0 + par            # parity of the step this set serves
            upd = self._sets[st & 1]       # inputs of THIS step's update (drawn at the end of the previous A)
            nxt = self._sets[(st + 1) & 1]
            s, a, r, s2, d = upd["batches"][0]
            e1, e2 = upd["eps1"][0], upd["eps2"][0]
            graphs = {}

            def cap_graph(fn):
                g = torch.cuda.CUDAGraph()
                cap.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cap):
                    with torch.cuda.graph(g, stream=cap):
                        fn()
                torch.cuda.current_stream(dev).wait_stream(cap)
                return g

            def seg_a():
                sac.snapshot_policy()
                self._snapshot = True
                try:
                    self.collect()
                finally:
                    self._snapshot = False
                self._draw_update_inputs(nxt)
            graphs["A"] = cap_graph(seg_a)
            graphs["B1"] = cap_graph(lambda: sac.critic_grads(s, a, r, s2, d, e1))
            graphs["B2"] = cap_graph(lambda: (sac.critic_apply(gs), sac.actor_grads(s, e2)))
            graphs["B3"] = cap_graph(lambda: sac.actor_apply(gs))
            sets.append(graphs)
        self.cur = cur0  # capturing executed nothing (collect() flipped the double buffer twice: back where it was)
        assert self.steps == steps0
        torch.cuda.synchronize(dev)
        sync = self.sync
        n_pol = sac.n_policy

        def seg_step():
            main = torch.cuda.current_stream(dev)
            side = self._side             # (read per step: set_cu_split / tune_learner_stream may have replaced it since the capture)
            g = sets[(self.steps - steps0) & 1]
            main.wait_stream(side)        # previous update done: its parameters are what A snapshots
            self._fork.record(main)       # ... and the batches drawn by the previous A are complete
            g["A"].replay()
            side.wait_event(self._fork)
            with torch.cuda.stream(side):
                g["B1"].replay()
                if sync is not None:
                    sync(sac.grads[n_pol:])
                g["B2"].replay()
                if sync is not None:
                    sync(sac.grads[:n_pol])
                g["B3"].replay()
                if self._side_done is not None:
                    self._side_done.record(side)
            if not self.defer_join:
                main.wait_stream(side)
            self._throttle_mark()
            self.cur = 1 - self.cur
            self.steps += 1
            if self.curriculum is not None:
                self._curriculum_tick()

        def step_fn():
            self._throttle()
            self._on_main(seg_step)

        step_fn.graphs = sets
        return step_fn

    def uses_rows_kernel(self) -> bool:
        """does the acting pass of this trainer go through the one-launch kernels (whose CU-sharing form the split chooses)?"""
        return int(self.sac.cfg.family) == 0 and self.n >= 1024

    def tune_share_rows(self, candidates=None, steps: int = 20):
        """Choose how many rows the acting kernel handles in its CU-sharing form (the rest run in the exclusive form) from MEASURED
        step times: each candidate split runs `steps` real train steps (3 more to settle) in the regime the loop runs in (join
        deferred to the next step), timed with HIP events around the whole run; the time is maximised over the ranks of a
        data-parallel job (so it includes the collectives, and every rank picks the same split); the fastest wins.  Also records
        the update's end-of-stream slack at the chosen split: how long before the end of the step's main-stream work the learner's
        stream went idle (negative: the update is the critical path)."""
        n = self.n
        if not (self.share_cus and self.overlap) or self.hier is not None:
            return None
        if candidates is None:
            if n >= 16384:  # 64-row workgroups: the split decides how many rows run one workgroup per CU
                candidates = sorted({(n * k // 8) // 64 * 64 for k in (0, 2, 3, 4, 5, 6, 8)})
            else:           # 16-row workgroups: all rows at one workgroup per CU, or none
                candidates = [0, n]
        while self.steps < 2:
            self.step(True)
        results = []
        dj = self.defer_join
        self.defer_join = True
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            for k in candidates:
                self.share_rows = int(k)
                for _ in range(3):
                    self.step(True)
                torch.cuda.synchronize(self.device)
                e0.record()
                for _ in range(steps):
                    self.step(True)
                torch.cuda.current_stream(self.device).wait_stream(self._side)
                e1.record()
                torch.cuda.synchronize(self.device)
                us = e0.elapsed_time(e1) * 1e3 / steps
                if self.world > 1:
                    import torch.distributed as dist
                    t = torch.tensor([us], dtype=torch.float64, device=self.device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    us = float(t.item())
                results.append((int(k), us))
            best = min(results, key=lambda kv: kv[1])
            self.share_rows = best[0]
            # slack of the learner's stream at the chosen split
            self._side_done = torch.cuda.Event(enable_timing=True)
            m_end = torch.cuda.Event(enable_timing=True)
            slacks = []
            for _ in range(5):
                self.step(True)
                m_end.record()
                torch.cuda.synchronize(self.device)
                slacks.append(self._side_done.elapsed_time(m_end) * 1e3)  # > 0: the update ended before the main stream did
            self._side_done = None
        finally:
            self.defer_join = dj
        self.share_tuning = {"chosen_share_rows": best[0], "us_per_step": best[1],
                             "candidates": [{"share_rows": k, "us_per_step": us} for k, us in results],
                             "update_end_slack_us": sorted(slacks)[len(slacks) // 2],
                             "note": "slack = main-stream end minus learner-stream end of a step (median of 5), measured after the "
                                     "choice; times are max over ranks"}
        return self.share_tuning

    def tune_cu_split(self, candidates=(0, 96, 128, 160), steps: int = 30):
        """Choose the CU partition of the two streams (set_cu_split) from measured step times, like tune_share_rows (real train
        steps in the deferred-join regime, max over ranks, every rank takes the same decision).  Only up to 16 384 envs, where the
        acting kernel's 256 - 512 workgroups fit a slice of the chip at two per CU and the update is the critical path."""
        if not self.overlap or self.hier is not None or self.n > 16384 or not self.uses_rows_kernel():
            return None
        while self.steps < 2:
            self.step(True)
        for _ in range(10):  # (the first candidate would otherwise pay for whatever the previous phase left cold)
            self.step(True)
        dj = self.defer_join
        self.defer_join = True
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        seen = {}
        try:
            for k in list(candidates) * 2:  # two passes, the minimum counts: a one-off stall (seen: ~20 ms) must not decide
                self.set_cu_split(int(k))
                for _ in range(6):
                    self.step(True)
                torch.cuda.synchronize(self.device)
                e0.record()
                for _ in range(steps):
                    self.step(True)
                torch.cuda.current_stream(self.device).wait_stream(self._side)
                e1.record()
                torch.cuda.synchronize(self.device)
                us = e0.elapsed_time(e1) * 1e3 / steps
                if self.world > 1:
                    import torch.distributed as dist
                    t = torch.tensor([us], dtype=torch.float64, device=self.device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    us = float(t.item())
                seen[int(k)] = min(us, seen.get(int(k), float("inf")))
            results = [(int(k), seen[int(k)]) for k in candidates]
            best = min(results, key=lambda kv: kv[1])
            if best[0] != 0 and best[1] > 0.97 * dict(results).get(0, float("inf")):
                best = (0, dict(results)[0])  # (a partition has to pay clearly: it also takes the learner's stream priority away)
            self.set_cu_split(best[0])
        finally:
            self.defer_join = dj
        self.cu_tuning = {"main_stream_cu_mask_bits": best[0], "us_per_step": best[1],
                          "candidates": [{"main_bits": k, "us_per_step": us} for k, us in results],
                          "note": "acting / env / replay on mask bits [0, k), the update on [k, n_cu); 0 = unpartitioned"}
        return self.cu_tuning

    def tune_learner_stream(self, steps: int = 10, margin: float = 0.9):
        """Keep the high-priority learner stream unless a NORMAL-priority one is clearly faster (step time below `margin` x): with
        the hardware queues oversubscribed by other streams of the process a high-priority stream can cost 5x (streams.py).  Timed
        like tune_share_rows (real train steps, max over ranks, every rank takes the same decision)."""
        if not self.overlap:
            return None
        while self.steps < 2:
            self.step(True)
        cands = [("high", learner_stream(self.device, -1)), ("normal", learner_stream(self.device, 0))]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = {}
        for name, st in cands:
            torch.cuda.synchronize(self.device)
            self._side = st
            for _ in range(3):
                self.step(True)
            torch.cuda.synchronize(self.device)
            e0.record()
            for _ in range(steps):
                self.step(True)
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            e1.record()
            torch.cuda.synchronize(self.device)
            us = e0.elapsed_time(e1) * 1e3 / steps
            if self.world > 1:
                import torch.distributed as dist
                t = torch.tensor([us], dtype=torch.float64, device=self.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                us = float(t.item())
            res[name] = us
        pick = "normal" if res["normal"] < margin * res["high"] else "high"
        torch.cuda.synchronize(self.device)
        self._side = self._learner = dict(cands)[pick]
        self.stream_tuning = {"learner_stream_priority": pick, "us_per_step": res}
        return self.stream_tuning

    # -- true resume (the reference's --resume is a stub, scripts/train.py:904-907): learner, env SoA state, current
    #    observations, replay contents and counters, and the device RNG state
    def state_dict(self):
        torch.cuda.synchronize(self.device)
        rows, meta = self.rb.export()
        return {
            "steps": self.steps, "cur": self.cur,
            "sac": {"params": self.sac.params.cpu(), "adam_m": self.sac.adam_m.cpu(), "adam_v": self.sac.adam_v.cpu(),
                    "adam_steps": self.sac.adam_steps()},
            "env": {k: v.cpu() for k, v in self.env.export_state().items()},
            "obs": self.obs[self.cur].cpu(), "prev_done": self.prev_done.cpu(),
            "replay": {"rows": rows.cpu(), "meta": meta, "seed": int(self.rb.seed)},
            "env_seed": int(self.env.cfg.seed),
            "rng": torch.cuda.get_rng_state(self.device),
            # the never-trained acting-path nets are seeded by the constructor: carry them, so that a trainer built with another
            # seed resumes with the SAME curiosity bonus / goal policy / safety correction
            "aux_nets": {k: v.params.cpu() for k, v in (("curiosity", self.curiosity), ("safety", self.safety),
                                                        ("hier_high", self.hier.high if self.hier is not None else None),
                                                        ("hier_low", self.hier.low if self.hier is not None else None))
                         if v is not None},
            # device-side episode statistics (running return per env + totals) and the curriculum driver's position: without them
            # the first episode finished after a resume carries a partial return and the driver restarts from its constructor stage
            "episode_stats": self.env.episode_stats_state(),
            "curriculum": None if self.curriculum is None else {
                "stage_idx": int(self.curriculum.current_stage_idx), "step": int(self.curriculum.current_step),
                "transitions": [int(x) for x in self.curriculum.stage_transition_steps],
                "completed": [bool(st.completed) for st in self.curriculum.stages],
                "last_totals": [float(x) for x in self._cur_last]},
            "act_counters": {"sac": self.sac.act_counter(),
                             "hier_low": self.hier.low.act_counter() if self.hier is not None else 0},
        }

    def load_state_dict(self, sd):
        self.steps, self.cur = int(sd["steps"]), int(sd["cur"])
        self.sac.params.copy_(sd["sac"]["params"])
        self.sac.adam_m.copy_(sd["sac"]["adam_m"])
        self.sac.adam_v.copy_(sd["sac"]["adam_v"])
        self.sac.set_adam_steps(sd["sac"]["adam_steps"])
        self.sac.sync_derived()
        self.env.import_state(**sd["env"])
        self.obs[self.cur].copy_(sd["obs"])
        self.prev_done.copy_(sd["prev_done"])
        self.rb.import_(sd["replay"]["rows"], sd["replay"]["meta"])
        self.rb.seed = int(sd["replay"]["seed"])  # Philox key of the batch draws
        if int(sd["env_seed"]) != int(self.env.cfg.seed) and int(self.env.cfg.dr_enabled):
            raise ValueError("domain-randomised envs draw from (seed, env id, episode): build the trainer with the checkpoint's "
                             f"seed {int(sd['env_seed'])} to resume")
        mine = {"curiosity": self.curiosity, "safety": self.safety,
                "hier_high": self.hier.high if self.hier is not None else None,
                "hier_low": self.hier.low if self.hier is not None else None}
        saved = sd.get("aux_nets", {})
        for k, net in mine.items():
            if (net is not None) != (k in saved):
                raise ValueError(f"checkpoint {'has' if k in saved else 'lacks'} the {k} net but this trainer was built "
                                 f"{'without' if net is None else 'with'} it")
            if net is not None:
                net.params.copy_(saved[k])
        if self.hier is not None:
            self.hier.low.sync_derived()
        self.env.load_episode_stats_state(sd.get("episode_stats") or {})
        cs = sd.get("curriculum")
        if cs is not None and self.curriculum is not None:
            drv = self.curriculum
            drv.current_stage_idx, drv.current_step = int(cs["stage_idx"]), int(cs["step"])
            drv.stage_transition_steps = list(cs["transitions"])
            for st, done in zip(drv.stages, cs["completed"]):
                st.completed = bool(done)
            drv._apply()
            self._cur_last = [float(x) for x in cs["last_totals"]]
            self._cur_pending = False
            self._cur_next = (self.steps // self._cur_every + 1) * self._cur_every
        ac = sd.get("act_counters") or {}
        self.sac.set_act_counter(int(ac.get("sac", 0)))
        if self.hier is not None:
            self.hier.low.set_act_counter(int(ac.get("hier_low", 0)))
        torch.cuda.set_rng_state(sd["rng"], self.device)
        torch.cuda.synchronize(self.device)

    def save_checkpoint(self, path: str):
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path: str):
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    def stats(self):
        return {"env_steps": self.steps * self.n, "updates": self.steps * self.updates_per_step,
                "losses": self.sac.losses.cpu().tolist()}


def bench_train(args, world, rank, device, n_envs=None):
    """bench.py workload 'train': returns the step function, the trainer and the description of what runs."""
    family = getattr(args, "family", 0)
    n = int(n_envs if n_envs is not None else args.envs_per_gpu)
    env_over = {}
    stage = getattr(args, "dr_stage", None)
    stage = None if stage is None or stage <= 0 else int(stage)
    if stage is not None:  # BASELINE configs[4]: full domain randomisation at a curriculum stage (config.yaml:236-286, 340-349)
        from .env import dr_from_yaml
        env_over = dr_from_yaml({}, stage)
    # reward-history window: the reference's whole 1000-entry deque (env/...:221) unless the approximate 10-entry mode is asked for
    win = 1000 if getattr(args, "exact_reward", False) else int(getattr(args, "reward_window", 0) or 0)
    env_over["distinct_window"] = win if win else 1000
    shipped = bool(getattr(args, "shipped_acting", False))
    utd = max(1, int(getattr(args, "updates_per_step", 1)))
    tr = VecTrainer(n, device=device, family=family, batch_size=256, replay_capacity=1_000_000, seed=42,
                    rank=rank, world=world, updates_per_step=utd, overlap=not getattr(args, "no_overlap", False),
                    share_cus={"auto": None, "on": True, "off": False}[getattr(args, "share_cus", "auto")],
                    defer_join=True,  # the bench synchronises the device around its timed region
                    share_rows=None if int(getattr(args, "share_rows", -1)) < 0 else int(args.share_rows),
                    acting_dropout=bool(getattr(args, "acting_dropout", False)),
                    acting_x3=bool(getattr(args, "acting_x3", False)),
                    enable_hierarchical=shipped, enable_safety=shipped, enable_curiosity=shipped, **env_over)
    if stage is not None:  # the curriculum driver reads device-side episode statistics and owns the stage from here on
        from .curriculum import CurriculumDriver
        from .env import default_curriculum_config
        drv = CurriculumDriver(default_curriculum_config())
        drv.current_stage_idx = stage - 1
        drv.current_step = sum(s.duration_steps for s in drv.stages[:stage - 1])
        tr.attach_curriculum(drv, every=50, min_episodes=50)
    # steady state of a long run: every env's 1000-entry reward history is full (from step 1000 on the step kernel scans the whole
    # ring every step -- the first few hundred steps of a run are up to 0.4 ms per step cheaper at 65 536 envs and are NOT what
    # this bench reports)
    prefill = int(getattr(args, "prefill_steps", 1000))
    tr.prefill_env(prefill)
    stream_tuning = tr.tune_learner_stream() if tr.overlap and not shipped and os.environ.get("TVC_TUNE_STREAM", "1") != "0" else None
    tuning = None
    if int(getattr(args, "share_rows", -1)) < 0 and utd == 1 and not shipped and tr.share_cus and tr.overlap and tr.uses_rows_kernel():
        tuning = tr.tune_share_rows()  # measured split, identical on every rank (times are maximised over the ranks)
    cu_tuning = None  # (after the split: the unpartitioned candidate is then the best unpartitioned schedule)
    if getattr(args, "cu_split", "auto") == "auto":
        cu_tuning = tr.tune_cu_split() if not shipped else None
    elif int(args.cu_split) > 0:
        tr.set_cu_split(int(args.cu_split))
    fam = "reference shapes (seq-len-1 transformer actor 2.26M trainable params, 512/256 GELU+LN critics)" if family == 0 \
        else "256x256 ReLU MLP actor/critics"
    return {"step_fn": lambda k: tr.step(True), "env": tr.env, "trainer": tr,
            "extra": {"updates_per_step": float(utd), "sac": {"family": fam, "batch": 256, "replay_capacity": 1_000_000,
                                                      "utd": f"{utd} update(s) per vector step", "dtype": "f32 MFMA",
                                                      "dropout_in_update": tr.dropout_p,
                                                      "acting_kernel_shares_cus": tr.share_cus,
                                                      "acting_rows_in_sharing_form": tr.share_rows if tr.share_cus else 0,
                                                      "share_rows_tuning": tuning,
                                                      "acting_dropout": tr.acting_dropout,
                                                      "acting_arithmetic": "bf16 matrix pipe, operands split in three bf16 terms, six products, fp32 accumulate "
                                                                           "(fp32-exact: tests/test_acting_x3_gpu.py)" if tr.acting_x3
                                                      else "f32-input MFMA (v_mfma_f32_16x16x4_f32)",
                                                      "acting": "hierarchical goal policy + safety layer + curiosity bonus (shipped config.yaml)"
                                                      if shipped else "SAC policy",
                                                      "reward_history_window": int(tr.env.cfg.distinct_window),
                                                      "env_prefill_steps": prefill,
                                                      "cu_split": {"main_stream_cu_mask_bits": tr.cu_split, "tuning": cu_tuning},
                                                      "learner_stream": stream_tuning,
                                                      "domain_randomisation": "off (shipped env)" if stage is None
                                                      else f"curriculum stage {stage} with the curriculum driver attached "
                                                           f"(device-side episode statistics): {env_over}"}}}


def smoke():
    """tiny end-to-end train loop on cuda:0 (called from __graft_entry__.smoke)"""
    tr = VecTrainer(256, device="cuda:0", family=0, batch_size=64, replay_capacity=4096, seed=1, enable_curiosity=True)
    for _ in range(6):
        tr.step(True)
    torch.cuda.synchronize()
    st = tr.stats()
    assert all(torch.isfinite(torch.tensor(st["losses"]))), st
    tr.close()
    return st
