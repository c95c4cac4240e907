"""N>1 path on CPU: world_size-2 gloo processes.  RCCL cannot run without GPUs, so what is covered here is
everything around the one collective of the path -- the task sharding (task i -> rank i % world), the layout and
meaning of the all-reduced buffer [grad sums | loss sum | inner-KL sums | outer-KL sum], the 1/M_global scaling and
the replicated Adam step -- with torch.distributed(gloo) standing in for ncclAllReduce and the oracle for the
kernels.  The result must equal the single-process meta-update."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
from oracle import policy as op, promp as pm
from tests import helpers
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
M, P, T, O, A, hidden, K = 4, 2, 20, 5, 3, (8, 8), 1
theta, all_slabs, _ = helpers.make_promp_case(77, M, P, T, O, A, hidden, K)
spec = op.PolicySpec(O, A, hidden)
alpha, eta = np.full(spec.n_params, 0.1), np.array([5e-4])
th = theta.astype(np.float64)
adam = pm.AdamState(spec.n_params)
mine = [i for i in range(M) if i %% world == rank]            # same sharding rule as bench.py / DeviceSession
for epoch in range(3):
    r = pm.meta_objective_and_grad(spec, th, all_slabs, alpha, eta, 0.3, tasks=mine, n_tasks_total=1)   # local SUMS
    # the fused buffer libpromp_hip all-reduces: [grad | J | inner_kl[K] | outer_kl]  (k_reduce_final)
    J_sum = r['loss'] - float(np.mean(eta * r['inner_kl']))
    buf = torch.tensor(np.concatenate([r['grad'], [J_sum], r['inner_kl'], [r['outer_kl']]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    buf = buf.numpy() / M                                      # k_mean_adam: 1 / n_tasks_global
    grad = buf[:spec.n_params]
    th = pm.adam_step(th, grad, adam, 1e-3)
if rank == 0:
    np.save(os.environ['OUT'], th)
dist.destroy_process_group()
'''


def test_two_rank_sharded_meta_update_equals_single_process(tmp_path):
    from oracle import policy as op, promp as pm
    from tests import helpers
    out = str(tmp_path / 'theta.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, OUT=out, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29631', str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    M, P, T, O, A, hidden, K = 4, 2, 20, 5, 3, (8, 8), 1
    theta, all_slabs, _ = helpers.make_promp_case(77, M, P, T, O, A, hidden, K)
    spec = op.PolicySpec(O, A, hidden)
    th, adam = theta.astype(np.float64), pm.AdamState(spec.n_params)
    for _ in range(3):
        r = pm.meta_objective_and_grad(spec, th, all_slabs, np.full(spec.n_params, 0.1), np.array([5e-4]), 0.3)
        th = pm.adam_step(th, r['grad'], adam, 1e-3)
    np.testing.assert_allclose(np.load(out), th, rtol=1e-10, atol=1e-12)


def test_rendezvous_socket_fallback_two_processes(tmp_path):
    """promp_amd.comm.exchange_unique_id without torch: rank 0 serves the 128-byte id over a socket."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
from promp_amd import comm
rank = int(sys.argv[1])
uid = comm._exchange_socket(rank, 2, bytes(range(128)) if rank == 0 else None, '127.0.0.1', 29733, 60)
assert uid == bytes(range(128))
''' % ROOT
    ps = [subprocess.Popen([sys.executable, '-c', code, str(r)]) for r in (0, 1)]
    assert [p.wait(timeout=120) for p in ps] == [0, 0]


def test_rendezvous_under_torchrun_agent_store(tmp_path):
    """The launch the driver uses for N>1: python -m torch.distributed.run --master-addr/--master-port ... bench.py.
    promp_amd.comm.exchange_unique_id must hand rank 0's 128-byte id to every rank through the agent's TCPStore."""
    script = tmp_path / 'rdzv.py'
    script.write_text(r'''
import os, sys
sys.path.insert(0, %r)
from promp_amd import comm
rank, world, local = comm.env_world()
uid = comm.exchange_unique_id(rank, world, lambda: bytes((7 * i + 3) %% 256 for i in range(128)))
assert uid == bytes((7 * i + 3) %% 256 for i in range(128)), uid[:8]
uid2 = comm.exchange_unique_id(rank, world, lambda: bytes((5 * i + 1) %% 256 for i in range(128)))   # a second communicator
assert uid2 == bytes((5 * i + 1) %% 256 for i in range(128)), uid2[:8]
assert local == rank
open(os.path.join(%r, 'ok%%d' %% rank), 'w').write('1')
''' % (ROOT, str(tmp_path)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29641', str(script)]
    subprocess.run(cmd, check=True, timeout=600, cwd=ROOT)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
