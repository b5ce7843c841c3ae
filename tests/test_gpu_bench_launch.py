"""bench.py launch forms (VERDICT r1 "next" #1): the N > 1 path has to start from a bare shell and its collective path
has to have run on hardware before an 8-GPU node sees it.

The GPU box has ONE MI355X.  RCCL refuses two ranks on one device ("Duplicate GPU detected", log in
profiles/r02_rccl_duplicate_gpu.log), so the two halves are covered separately:
  * two self-launched ranks sharing the GPU, gather over gloo (host-staged) - self-launch, shard, GatherPipe, MAX-reduce;
  * one rank with --force-pg over nccl - init_process_group('nccl'), the RCCL gather / barrier / all_reduce kernels.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env=None, timeout=600):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(env or {}))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, cwd=ROOT, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_from_a_bare_shell():
    r = _run(['--gpus', '2', '--batch', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--backend', 'gloo'],
             env={'SE_BENCH_SHARE_GPU': '1'})
    assert r['n_gpus'] == 2 and r['config']['global_batch'] == 16 and r['value'] > 0
    assert r['steps'] == 2 and r['warmup'] == 1 and r['scaling'] == 'weak'


def test_bench_self_launches_eight_ranks_from_a_bare_shell():
    """The launch shape of the driver's last scaling point (N = 8) on the one GPU this box has: eight self-launched ranks share
    the device, three timed steps so that both GatherPipe slots wrap, gather over gloo (RCCL refuses ranks that share a device)."""
    r = _run(['--gpus', '8', '--batch', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-profile', '--backend', 'gloo'],
             env={'SE_BENCH_SHARE_GPU': '1'}, timeout=900)
    assert r['n_gpus'] == 8 and r['config']['global_batch'] == 16 and r['value'] > 0
    assert r['collective'] == {'backend': 'gloo', 'ranks': 8, 'distinct_devices': 1} and r['rccl_ranks'] == 0


def test_bench_single_rank_gather_over_rccl():
    r = _run(['--gpus', '1', '--batch', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--force-pg'])
    assert r['n_gpus'] == 1 and r['value'] > 0
    assert r['rccl_ranks'] == 1 and r['collective']['backend'] == 'nccl' and r['collective']['distinct_devices'] == 1


def test_bench_rejects_more_ranks_than_gpus():
    e = dict(os.environ)
    e.pop('SE_BENCH_SHARE_GPU', None)
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--batch', '2', '--steps', '1',
                        '--warmup', '0', '--no-cpu-baseline'], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'GPU(s) visible' in (p.stdout + p.stderr)
