"""CPU-side checks of the drop-in boundary: the library builds, loads, exports every symbol that
include/tvc_native.h declares, the ctypes mirror matches the C struct, and the product path fails
loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tvc_native.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(tvc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from tvc_ai_amd import _native as nat
    L = nat.load()
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/tvc_native.h but not exported"
    for n in names:
        assert n in nat.SIGNATURES, f"{n} has no ctypes signature in tvc_ai_amd/_native.py"
    assert L.tvc_abi_version() == 2


def _c_layout(tmp_path, struct, fields):
    """sizeof + offsetof of every field, from gcc and the header itself"""
    src = tmp_path / f"{struct}.c"
    body = "".join(f'printf("%zu\\n", offsetof({struct}, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tvc_native.h"\n'
                   f'int main(){{printf("%zu\\n", sizeof({struct}));{body}return 0;}}\n')
    exe = tmp_path / struct
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = list(map(int, subprocess.check_output([str(exe)]).split()))
    return vals[0], dict(zip(fields, vals[1:]))


@pytest.mark.parametrize("struct,mirror", [("tvc_env_cfg", "EnvCfg"), ("tvc_sac_cfg", "SacCfg")])
def test_ctypes_struct_matches_header(tmp_path, struct, mirror):
    """size and EVERY field offset of the ctypes mirrors against the C structs of include/tvc_native.h; also that the
    mirror lists the header's fields in the header's order (a renamed / reordered field is caught, not only a resized one)."""
    from tvc_ai_amd import _native as nat
    cls = getattr(nat, mirror)
    fields = [f[0] for f in cls._fields_]
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, flags=re.S).group(1)
    declared = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if stmt:
            declared += [re.sub(r"\[.*\]", "", n.strip()) for n in re.sub(r"^[a-z0-9_]+\s+", "", stmt).split(",")]
    assert declared == fields, (declared, fields)
    size, offs = _c_layout(tmp_path, struct, fields)
    assert size == C.sizeof(cls)
    for f in fields:
        assert offs[f] == getattr(cls, f).offset, f


def test_dropin_tree_resolves_the_imports_of_train_py():
    """scripts/train.py:44-45 with <repo>/dropin on sys.path: module paths, class names, MissionPhase order (index / 7 is obs[8])."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from env.enhanced_rocket_tvc_env import EnhancedRocketTVCEnv, MissionPhase\n"
            "from agent.multi_algorithm_agent import MultiAlgorithmAgent\n"
            "import env, agent\n"
            "assert [p.value for p in MissionPhase] == ['boost', 'coast', 'landing', 'touchdown', 'hover', 'complete', 'failed']\n"
            "assert env.EnhancedRocketTVCEnv is EnhancedRocketTVCEnv and agent.MultiAlgorithmAgent is MultiAlgorithmAgent\n"
            "for n in ('make_training_env', 'make_evaluation_env', 'make_debug_env', 'MissionPhase'): assert hasattr(env, n), n\n"
            "for n in ('reset', 'step', 'close', 'render'): assert hasattr(EnhancedRocketTVCEnv, n), n\n"
            "for n in ('select_algorithm', 'get_action', 'update', 'update_performance', 'save_checkpoint', 'load_checkpoint', 'to'):\n"
            "    assert hasattr(MultiAlgorithmAgent, n), n\n"
            "import inspect\n"
            "assert list(inspect.signature(EnhancedRocketTVCEnv.__init__).parameters)[1:8] == ['config', 'max_episode_steps', "
            "'render_mode', 'enable_hierarchical', 'enable_curiosity', 'enable_physics_informed', 'debug']\n"
            "assert list(inspect.signature(MultiAlgorithmAgent.__init__).parameters)[1:4] == ['obs_dim', 'action_dim', 'config']\n"
            "print('ok')\n") % (os.path.join(ROOT, "dropin"), ROOT)
    out = subprocess.check_output([sys.executable, "-c", code], cwd="/tmp")
    assert out.strip().endswith(b"ok")


def test_default_cfg_matches_reference_constants():
    # env/enhanced_rocket_tvc_env.py:409-464, :324-352
    from tvc_ai_amd import _native as nat
    cfg = nat.EnvCfg()
    nat.load().tvc_env_default_cfg(C.byref(cfg))
    assert cfg.mass == 2.0 and cfg.thrust == 35.0
    assert cfg.inertia_xx == (1 / 12) * 2.0 * (3 * 0.05 ** 2 + 1.0 ** 2)
    assert cfg.inertia_zz == (1 / 2) * 2.0 * 0.05 ** 2
    assert cfg.lin_damp == 0.01 and cfg.ang_damp == 0.02 and cfg.gravity == 9.81
    assert cfg.n_sub == 4 and cfg.dt_sub == 0.02 / 4 and cfg.max_episode_steps == 1000
    assert list(cfg.init_pos) == [0, 0, 1.0] and list(cfg.init_quat) == [0, 0, 0, 1.0]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    import tvc_ai_amd
    with pytest.raises(tvc_ai_amd.TvcError, match="no HIP device|no CPU fallback|GPU"):
        tvc_ai_amd.VecRocketTVCEnv(4)
    with pytest.raises(tvc_ai_amd.TvcError):
        tvc_ai_amd.VecRocketTVCEnv(4, device="cpu")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tvc_ai_amd/ may reference it."""
    pkg = os.path.join(ROOT, "tvc_ai_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "tvc_oracle.h" not in txt or "same model as oracle" in txt, f
                assert "libtvc_oracle" not in txt, f


def test_integer_distinct_test_equals_float_test():
    # kernel uses 5*distinct > 4*len in place of distinct > len*0.8 (env/...:221); identical for len <= 1000
    for n in range(0, 1001):
        thr = n * 0.8
        for d in (int(thr) - 1, int(thr), int(thr) + 1):
            if 0 <= d <= n:
                assert (d > thr) == (5 * d > 4 * n), (n, d)
