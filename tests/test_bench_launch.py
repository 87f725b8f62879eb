"""bench.py launch plan (CPU): `--gpus N` must never print a 1-GPU number under an N-GPU name."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_single_gpu_is_one_rank():
    b = _bench()
    assert b.launch_plan(1, {}, 1, []) == {'mode': 'rank', 'world': 1, 'rank': 0, 'local_rank': 0}
    assert b.launch_plan(1, {}, None, [])['mode'] == 'rank'


def test_self_launch_spawns_one_rank_per_gpu():
    b = _bench()
    plan = b.launch_plan(8, {}, 8, ['--gpus', '8', '--steps', '20', '--warmup', '5'], port=29511)
    assert plan['mode'] == 'spawn' and plan['world'] == 8
    cmd = plan['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
    assert os.path.basename(cmd[-7]) == 'bench.py'


def test_too_few_devices_fails_loudly():
    b = _bench()
    with pytest.raises(SystemExit) as e:
        b.launch_plan(2, {}, 1, ['--gpus', '2'])
    assert 'only 1 GPU' in str(e.value)
    with pytest.raises(SystemExit):
        b.launch_plan(8, {}, 0, [])


def test_inside_launcher_world_must_match():
    b = _bench()
    env = {'WORLD_SIZE': '4', 'RANK': '3', 'LOCAL_RANK': '3'}
    assert b.launch_plan(4, env, None, []) == {'mode': 'rank', 'world': 4, 'rank': 3, 'local_rank': 3}
    with pytest.raises(SystemExit):
        b.launch_plan(8, env, None, [])
    with pytest.raises(SystemExit):
        b.launch_plan(1, env, None, [])
    with pytest.raises(SystemExit):          # a rank whose device does not exist
        b.launch_plan(4, env, 2, [])


def test_parent_of_self_launch_never_imports_torch_or_touches_gpu():
    """The spawn branch returns before `import torch` / Engine(): checked on the source text."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.index("plan['mode'] == 'spawn'") < main.index('import torch') < main.index('pkg.Engine(')
