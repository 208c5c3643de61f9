"""Acting pass alone (tvc_sac_act: policy forward + Gaussian sample) at several row counts; compares against the eager fp32
restatement.  TVC_ROWS_MIN=1000000000 forces the per-layer kernels, TVC_ROWS_MIN=1 the one-launch row-owner kernel.
usage: act_bench.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd.agent import NativeSAC, sac_cfg

rows = [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384, 65536]
dev = torch.device("cuda:0")
SHARE = os.environ.get("TVC_ACT_SHARE", "0") == "1"  # tvc_sac_act flags bit 2: one workgroup per CU
TRAIN = os.environ.get("TVC_ACT_TRAIN", "0") == "1"   # tvc_sac_act flags bit 3: the net as trained, Dropout live (1.9x the MFMA work: nothing folds)
X3 = os.environ.get("TVC_ACT_X3", "0") == "1"        # tvc_sac_act flags bit 4: split-operand kernel (bf16 matrix pipe, fp32-exact)
for n in rows:
    sac = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n, dropout_p=0.1 if TRAIN else 0.0), device=dev, seed=2)
    ob, ep = torch.randn(n, 10, device=dev), torch.randn(n, 2, device=dev)
    outs = tuple(torch.empty(n, 2, device=dev) for _ in range(3))
    for _ in range(3):
        sac.act(ob, ep, out=outs, share_cus=SHARE, x3=X3, train_mode=TRAIN)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(5):
                sac.act(ob, ep, out=outs, share_cus=SHARE, x3=X3, train_mode=TRAIN)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    macs = 10 * 256 + 3 * 256 * 256 + 4 * (2 * 256 * 512) + 256 * 512 + 512 * 512 + 512 * 4
    print(f"rows {n:6d}: {best:8.1f} us  {2.0 * macs * n / best / 1e6:6.1f} TFLOP/s  (TVC_ROWS_MIN={os.environ.get('TVC_ROWS_MIN', 'default')}, share_cus={SHARE}, x3={X3}, train_mode={TRAIN})", flush=True)
    if n >= 12288 or os.environ.get("TVC_ROWS_MIN") == "1":
        import ctypes as C
        out = (C.c_double * 3)()
        rc = sac.L.tvc_debug_rows_clock(sac._h, ob.data_ptr(), n, 40, out, torch.cuda.current_stream().cuda_stream)
        if rc == 0:
            print(f"             in-kernel clock {out[0]:7.1f} MHz, median workgroup lifetime {out[1]:8.1f} us, {int(out[2])} workgroups "
                  f"-> f32-MFMA peak at that clock {out[0] * 1e6 * 64 * 4 * 256 / 1e12:6.1f} TFLOP/s", flush=True)
    sac.close()
