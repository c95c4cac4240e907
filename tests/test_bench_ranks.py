"""bench.py's N-rank path, end to end, without a GPU: `bench.py --gpus 2` with no launcher must start its two ranks itself, the
ranks must rendezvous, shard the tasks, exchange through the communicator and print ONE line that says who took part."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_launches_its_ranks_and_describes_them():
    from tests import devlib
    devlib.emu_library()                       # built once here, not by two ranks at the same time
    env = dict(os.environ, PROMP_EMU_CUS='2', PROMP_EMU_DEVICES='2', PYTHONPATH=ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'bench_emu_driver.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
           '--repeats', '1', '--test-shape', '4,2,12,5,3,32', '--no-roofline', '--no-cpu-baseline', '--no-plugin-path']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and 'test_shape' in d
    assert d['config']['meta_batch_size'] == 4 and d['config']['tasks_per_gpu'] == 2
    r = d['rccl']
    assert r['nranks'] == 2 and r['nranks_reported_by_every_rank'] == [2, 2] and r['ranks'] == [0, 1]
    assert r['distinct_devices'] == 2 and len(r['devices']) == 2
    assert len(r['per_rank_ms_per_step']) == 2 and all(x > 0 for x in r['per_rank_ms_per_step'])
    assert r['exchanges_per_step'] == 6 and r['allreduce_us'] > 0        # E = 5 epochs + the statistics pass
    assert r['replicas_equal'] is True and r['replica_checksums']['adam_t'][0] == r['replica_checksums']['adam_t'][1] > 0
    assert d['weak_batch']['meta_batch_size'] == 8 and d['weak_batch']['tasks_per_gpu'] == 4
    assert d['value'] > 0 and d['ms_per_step'] > 0


def test_bench_under_torch_distributed_run_with_two_ranks():
    """the launch line the round-end driver uses for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ... -- the ranks take RANK / WORLD_SIZE / MASTER_* from the launcher (bench.py must
    not start ranks of its own), rendezvous beside the launcher's store (MASTER_PORT + 1) and rank 0 alone prints the line"""
    from tests import devlib
    devlib.emu_library()
    env = dict(os.environ, PROMP_EMU_CUS='2', PROMP_EMU_DEVICES='2', PYTHONPATH=ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    import socket
    with socket.socket() as s:                 # a free port whose successor is free too (the id exchange listens on MASTER_PORT + 1)
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'bench_emu_driver.py'), '--gpus', '2', '--steps', '1',
           '--warmup', '0', '--repeats', '1', '--test-shape', '4,2,12,5,3,32', '--no-roofline', '--no-cpu-baseline', '--no-plugin-path']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['tasks_per_gpu'] == 2
    r = d['rccl']
    assert r['nranks'] == 2 and r['nranks_reported_by_every_rank'] == [2, 2] and r['replicas_equal'] is True
    assert d['value'] > 0 and d['ms_per_step'] > 0
