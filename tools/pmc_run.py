"""Workload for rocprofv3 --pmc passes.
  pmc_run.py env 8192 65536 4194304   -> eager env_step launches at those sizes
  pmc_run.py gemm 65536 256 256       -> the acting-pass Linear kernel at M N K (tvc_nn_linear_forward)
  pmc_run.py rowln 65536 256 512      -> the fused Linear + residual + LayerNorm kernel (tvc_nn_linear_ln_forward)
  pmc_run.py act 65536                -> the one-launch acting kernel (tvc_sac_act, actor_rows_kernel)
  pmc_run.py envdr 65536 4194304      -> env_step with full domain randomisation (stage 5) and episode statistics on
  pmc_run.py envdr1000 65536          -> the same with the 1000-entry reward history (the train loop's instantiation)
  pmc_run.py act 4096 8192            -> below 16 384 rows tvc_sac_act launches actor_split_kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "env"
if mode == "rowln":  # pmc_run.py rowln M N K -> the fused Linear + residual + LayerNorm acting kernel
    from tvc_ai_amd import _native as nat
    L = nat.load()
    M, N, K = [int(x) for x in sys.argv[2:5]]
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.zeros(N, device="cuda")
    R = torch.randn(M, N, device="cuda"); g = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    for _ in range(20):
        nat.check(L.tvc_nn_linear_ln_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), g.data_ptr(), be.data_ptr(),
                                             Y.data_ptr(), M, N, K, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
elif mode == "act":
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    for n in [int(x) for x in sys.argv[2:]]:
        sac = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n), device="cuda:0", seed=2)
        ob, ep = torch.randn(n, 10, device="cuda"), torch.randn(n, 2, device="cuda")
        outs = tuple(torch.empty(n, 2, device="cuda") for _ in range(3))
        for _ in range(10):
            sac.act(ob, ep, out=outs, x3=os.environ.get("TVC_ACT_X3", "0") == "1")
        torch.cuda.synchronize()
        sac.close()
elif mode in ("envdr", "envdr1000"):  # envdr1000: the train loop's instantiation <W1000, DR> (reference-exact 1000-entry reward history)
    from tvc_ai_amd import VecRocketTVCEnv
    from tvc_ai_amd.env import dr_from_yaml
    for n in [int(x) for x in sys.argv[2:]]:
        over = dr_from_yaml({}, 5)
        if mode == "envdr1000":
            over["distinct_window"] = 1000
        env = VecRocketTVCEnv(n, **over); env.enable_episode_stats(); env.reset()
        if mode == "envdr1000":  # fill every env's 1000-entry history first: the step then scans the whole ring, as in a long run
            acts0 = (torch.rand((4, n, 2), device="cuda") * 2 - 1).contiguous()
            for k in range(1010):
                env.step(acts0[k % 4])
        acts = (torch.rand((4, n, 2), device="cuda") * 2 - 1).contiguous()
        for k in range(40):
            env.step(acts[k % 4])
        torch.cuda.synchronize()
        env.close()
elif mode == "gemm":
    from tvc_ai_amd import _native as nat
    L = nat.load()
    M, N, K = [int(x) for x in sys.argv[2:5]]
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    for _ in range(20):
        nat.check(L.tvc_nn_linear_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, N, K, 0, 0,
                                          torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
else:
    from tvc_ai_amd import VecRocketTVCEnv
    sizes = [int(x) for x in (sys.argv[2:] or ["8192", "4194304"])]
    for n in sizes:
        env = VecRocketTVCEnv(n); env.reset()
        acts = (torch.rand((4, n, 2), device="cuda") * 2 - 1).contiguous()
        for k in range(40):
            env.step(acts[k % 4])
        torch.cuda.synchronize()
        env.close()
