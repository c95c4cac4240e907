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


LIB_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
from promp_amd import _lib
from tests import devlib, helpers
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
os.environ['PROMP_EMU_CUS'] = '2'
lib = devlib.emu_library()          # the library's own kernel + host sources (SIMT interpreter build): no GPU in this container
M, P, T, O, A, hidden, K = 4, 1, 24, 5, 3, (32, 32), 1
theta, all_slabs, all_paths = helpers.make_promp_case(78, M, P, T, O, A, hidden, K)
mine = [i for i in range(M) if i %% world == rank]            # task i -> rank i %% world (bench.py, DeviceSession)
sub_paths = [type(p)((j, p[i]) for j, i in enumerate(mine)) for p in all_paths]
sub_slabs = [[s[i] for i in mine] for s in all_slabs]
R = max(sum(len(q['rewards']) for pl in p.values() for q in pl) for p in sub_paths)
ctx = _lib.Context(len(mine), O, A, hidden, K, max_rows=R, max_paths=len(mine) * P, lib=lib, n_tasks_global=M)
helpers.upload_slabs(ctx, sub_paths, sub_slabs)
ctx.set_theta(theta)
ctx.set_step_sizes(np.full(theta.size, 0.1, np.float32))
eta = np.array([5e-4], np.float32)
for epoch in range(2):
    ctx.meta_grad(0.3, eta)                                    # local SUMS stay in the exchange buffer (no communicator attached)
    buf = torch.from_numpy(ctx.reduced_get())                  # [grad | J | inner_kl[K] | outer_kl]
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)                 # gloo stands in for ncclAllReduce
    ctx.reduced_set(buf.numpy())
    ctx.adam_step(1e-3)                                        # k_mean_adam: 1 / n_tasks_global, replicated Adam
th = ctx.get_theta()
ths = [torch.zeros(theta.size) for _ in range(world)]
dist.all_gather(ths, torch.from_numpy(th))
assert all(torch.equal(ths[0], t) for t in ths), 'replicated parameters diverged'
if rank == 0:
    np.save(os.environ['OUT'], th)
dist.destroy_process_group()
'''


def test_two_rank_library_kernels_with_external_collective(tmp_path):
    """The library itself on two gloo ranks: each rank runs libpromp's kernels (SIMT-interpreter build, no GPU here) on
    its shard of the tasks, the [Theta+K+2] exchange buffer crosses the ranks through promp_reduced_get / _set, and
    promp_adam_step applies the replicated update.  Must equal the one-process run over all tasks."""
    from promp_amd import _lib
    from tests import devlib, helpers
    out = str(tmp_path / 'theta.npy')
    script = tmp_path / 'worker.py'
    script.write_text(LIB_WORKER % dict(root=ROOT))
    devlib.emu_library()               # build once, before two processes race for it
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29651', str(script)]
    subprocess.run(cmd, check=True, env=dict(os.environ, OUT=out, MASTER_ADDR='127.0.0.1'), timeout=900, cwd=ROOT)
    os.environ['PROMP_EMU_CUS'] = '2'
    try:
        lib = devlib.emu_library()
        M, P, T, O, A, hidden, K = 4, 1, 24, 5, 3, (32, 32), 1
        theta, all_slabs, all_paths = helpers.make_promp_case(78, M, P, T, O, A, hidden, K)
        R = max(sum(len(q['rewards']) for pl in p.values() for q in pl) for p in all_paths)
        ctx = _lib.Context(M, O, A, hidden, K, max_rows=R, max_paths=M * P, lib=lib)
        helpers.upload_slabs(ctx, all_paths, all_slabs)
        ctx.set_theta(theta)
        ctx.set_step_sizes(np.full(theta.size, 0.1, np.float32))
        ctx.optimize(2, 1e-3, 0.3, np.array([5e-4], np.float32))
        ref = ctx.get_theta()
        ctx.close()
    finally:
        del os.environ['PROMP_EMU_CUS']
    got = np.load(out)
    # float32 sums in a different task order (rank-major instead of task-major): Adam steps are ~lr per element
    assert np.max(np.abs(got - ref)) < 2e-5 and np.mean(np.abs(got - ref)) < 2e-6
    assert np.max(np.abs(ref - theta)) > 5e-4


def test_launcher_and_socket_rendezvous_without_torch(tmp_path):
    """python -m promp_amd.launch (no PyTorch anywhere): three ranks, two exchanges each through rank 0's socket server;
    torch must not have been imported by the package; a failing rank takes the job down with its exit code."""
    script = tmp_path / 'rdzv.py'
    script.write_text(r'''
import os, sys
sys.path.insert(0, %r)
from promp_amd import comm
rank, world, local = comm.env_world()
assert world == 3 and local == rank
for k in (3, 11):
    uid = comm.exchange_unique_id(rank, world, lambda: bytes((k * i + rank) %% 256 for i in range(128)))
    assert uid == bytes((k * i) %% 256 for i in range(128)), uid[:8]
assert 'torch' not in sys.modules
open(os.path.join(%r, 'ok%%d' %% rank), 'w').write('1')
if len(sys.argv) > 1 and rank == 1:
    sys.exit(7)
''' % (ROOT, str(tmp_path)))
    cmd = [sys.executable, '-m', 'promp_amd.launch', '--nproc', '3', '--master-port', '29741', str(script)]
    assert subprocess.run(cmd, timeout=300, cwd=ROOT).returncode == 0
    assert all(os.path.exists(tmp_path / ('ok%d' % r)) for r in range(3))
    assert subprocess.run(cmd[:-1] + ['--master-port', '29751', str(script), 'fail'], timeout=300, cwd=ROOT).returncode == 7


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
    promp_amd.comm.exchange_unique_id must hand rank 0's 128-byte id to every rank: by default over its own socket next to
    the agent's store (MASTER_PORT + 1), and -- PROMP_RDZV=torch -- through the agent's TCPStore."""
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
    os.remove(tmp_path / 'ok0'), os.remove(tmp_path / 'ok1')
    cmd[cmd.index('29641')] = '29645'
    subprocess.run(cmd, check=True, timeout=600, cwd=ROOT, env=dict(os.environ, PROMP_RDZV='torch'))
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


STATS_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
from collections import OrderedDict
import numpy as np
import torch, torch.distributed as dist
from promp_amd import _lib, session
from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
from promp_amd.samplers.dice_sample_processor import DiceMetaSampleProcessor
from promp_amd.utils import logger
from tests import devlib, helpers
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
os.environ['PROMP_EMU_CUS'] = '2'
_lib.set_library_for_testing(devlib.emu_library())
meta, paths, g = helpers.load_sample_proc('ragged')
M = len(paths)
mine = [i for i in range(M) if i %% world == rank]            # task i -> rank i %% world
keys = list(paths.keys())
sub = OrderedDict((j, paths[keys[i]]) for j, i in enumerate(mine))
first = sub[0][0]
O, A = first['observations'].shape[1], first['actions'].shape[1]

def gloo(values, op):                                          # stands in for promp_allreduce_f64 (RCCL needs GPUs)
    t = torch.from_numpy(values)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 'max' else dist.ReduceOp.SUM)
    return t.numpy()

sess = session.DeviceSession(len(mine), O, A, (32, 32), 1, n_tasks_global=M, rank=rank, world=world)
sess.collective = gloo
logged = {}
logger.logkv = lambda k, v: logged.__setitem__(k, v)
import promp_amd.samplers.base as sb
sb.logger.logkv = logger.logkv
proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), **meta['kwargs'])
out = proc.process_samples(sub, log='all', log_prefix='Step_0-')
T = max(len(p['rewards']) for pl in paths.values() for p in pl)
dproc = DiceMetaSampleProcessor(baseline=LinearFeatureBaseline(), max_path_length=T, discount=0.99)
dlogged = {}
sb.logger.logkv = lambda k, v: dlogged.__setitem__(k, v)
dout = dproc.process_samples(OrderedDict((j, [dict(p) for p in paths[keys[i]]]) for j, i in enumerate(mine)), log='all', log_prefix='D-')
np.savez(os.environ['OUT'] + '.%%d.npz' %% rank, tasks=np.array(mine),
         adj=np.concatenate([sd['adj_avg_rewards'] for sd in out]),
         dice_adj=np.concatenate([sd['adj_avg_rewards'].reshape(-1) for sd in dout]),
         stat_keys=np.array(sorted(logged)), stat_vals=np.array([logged[k] for k in sorted(logged)], dtype=np.float64),
         dstat_keys=np.array(sorted(dlogged)), dstat_vals=np.array([dlogged[k] for k in sorted(dlogged)], dtype=np.float64))
dist.destroy_process_group()
'''


def test_two_rank_sample_statistics_span_the_whole_meta_batch(tmp_path):
    """SURVEY K6 / K7 on two gloo ranks: each rank processes its shard of the tasks (the library's kernels, SIMT-interpreter
    build); adj_avg_rewards (E-MAML's weight, meta_sample_processor.py:40-44) and the six logged path statistics
    (samplers/base.py:135-149) must be those of the WHOLE meta-batch -- equal to the reference's outputs in the golden
    fixture and to a one-process run -- for MetaSampleProcessor and DiceMetaSampleProcessor alike."""
    from tests import devlib, helpers
    out = str(tmp_path / 'stats')
    script = tmp_path / 'worker.py'
    script.write_text(STATS_WORKER % dict(root=ROOT))
    devlib.emu_library()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29661', str(script)]
    subprocess.run(cmd, check=True, env=dict(os.environ, OUT=out, MASTER_ADDR='127.0.0.1'), timeout=900, cwd=ROOT)
    meta, paths, g = helpers.load_sample_proc('ragged')
    keys = list(paths.keys())
    n_rows = [sum(len(p['rewards']) for p in paths[k]) for k in keys]
    offs = np.concatenate([[0], np.cumsum(n_rows)])
    # the one-process reference for the logged statistics and the DiCE processor
    und = np.array([np.sum(np.asarray(p['rewards'], dtype=np.float64)) for k in keys for p in paths[k]])
    disc = np.array([np.sum(np.asarray(p['rewards'], dtype=np.float64) * 0.99 ** np.arange(len(p['rewards']))) for k in keys for p in paths[k]])
    T = max(len(p['rewards']) for pl in paths.values() for p in pl)
    padded = np.concatenate([np.pad(np.asarray(p['rewards'], dtype=np.float64), (0, T - len(p['rewards']))) for k in keys for p in paths[k]])
    pads_per_task = [len(paths[k]) * T for k in keys]
    poffs = np.concatenate([[0], np.cumsum(pads_per_task)])
    dice_ref = (padded - padded.mean()) / (padded.std() + 1e-8)
    for rank in (0, 1):
        r = np.load(out + '.%d.npz' % rank)
        want = np.concatenate([g['adj_avg_rewards'][offs[i]:offs[i + 1]] for i in r['tasks']])
        np.testing.assert_allclose(r['adj'], want, rtol=1e-4, atol=1e-5)      # (the one-process tolerance: float32 rewards)
        stats = dict(zip(r['stat_keys'], r['stat_vals']))
        assert stats['Step_0-NumTrajs'] == len(und)
        np.testing.assert_allclose(stats['Step_0-AverageReturn'], und.mean(), rtol=1e-10)
        np.testing.assert_allclose(stats['Step_0-StdReturn'], und.std(), rtol=1e-8)
        np.testing.assert_allclose(stats['Step_0-MaxReturn'], und.max(), rtol=1e-12)
        np.testing.assert_allclose(stats['Step_0-MinReturn'], und.min(), rtol=1e-12)
        disc_ret = np.array([np.sum(np.asarray(p['rewards'], dtype=np.float64) * meta['kwargs'].get('discount', 0.99) ** np.arange(len(p['rewards']))) for k in keys for p in paths[k]])
        np.testing.assert_allclose(stats['Step_0-AverageDiscountedReturn'], disc_ret.mean(), rtol=1e-9)
        dwant = np.concatenate([dice_ref[poffs[i]:poffs[i + 1]] for i in r['tasks']])
        np.testing.assert_allclose(r['dice_adj'], dwant, rtol=1e-9, atol=1e-10)
        dstats = dict(zip(r['dstat_keys'], r['dstat_vals']))
        np.testing.assert_allclose(dstats['D-AverageReturn'], und.mean(), rtol=1e-6)      # (path sums of float32 rewards, as the reference's)
        np.testing.assert_allclose(dstats['D-AverageDiscountedReturn'], disc.mean(), rtol=1e-6)
        assert dstats['D-NumTrajs'] == len(und)


OPT_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
from promp_amd import _lib, synthetic
from tests import devlib
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
os.environ['PROMP_EMU_CUS'] = '2'
_lib.set_library_for_testing(devlib.emu_library())
from tests.test_distributed import run_promp_iteration

def gloo(values, op):                                          # stands in for the RCCL exchange (no GPUs here)
    t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64))
    if op == 'max':
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.numpy()
    # the fixed-order form (promp_comm_fixed_order's: all-gather, then every rank adds the vectors in rank order)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = parts[0].clone()
    for p in parts[1:]:
        out += p
    return out.numpy()

M = 4
mine = [i for i in range(M) if i %% world == rank]
theta, stats = run_promp_iteration(mine, M, rank, world, gloo)
ths = [torch.zeros(theta.size) for _ in range(world)]
dist.all_gather(ths, torch.from_numpy(theta))
assert all(torch.equal(ths[0], t) for t in ths), 'replicated parameters diverged'
np.savez(os.environ['OUT'] + '.%%d.npz' %% rank, theta=theta, loss_before=stats['loss_before'], loss_after=stats['loss_after'],
         inner_kl=stats['inner_kl'])
dist.destroy_process_group()
'''


def run_promp_iteration(task_ids, M_global, rank, world, collective, epochs=2):
    """one ProMP iteration through the plugin classes (process_samples x2, _adapt, optimize_policy) on the tasks `task_ids`
    of an M_global-task meta-batch; every task's paths come from its own seed, so a shard is exactly its part of the batch"""
    from promp_amd import synthetic
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.meta_algos.pro_mp import ProMP
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    O, A, hidden, P, T = 5, 3, (32, 32), 2, 24
    np.random.seed(11)                                   # identical initial parameters on every rank
    policy = MetaGaussianMLPPolicy(name='meta-policy', obs_dim=O, action_dim=A, meta_batch_size=len(task_ids), hidden_sizes=hidden,
                                   n_tasks_global=M_global, rank=rank, world=world, device_id=0)
    policy.session.collective = collective
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=0.99, gae_lambda=1, normalize_adv=True)
    algo = ProMP(policy=policy, inner_lr=0.1, meta_batch_size=len(task_ids), num_inner_grad_steps=1, learning_rate=1e-3,
                 num_ppo_steps=epochs, clip_eps=0.3, target_inner_step=0.01, init_inner_kl_penalty=5e-4, adaptive_inner_kl_penalty=False)
    policy.switch_to_pre_update()
    samples = []
    for step in range(2):
        th = np.stack([np.concatenate([v.reshape(-1) for v in d.values()]) for d in policy.policies_params_vals])
        paths = synthetic.make_paths_for_tasks(900 + step, task_ids, th, P, T, O, A, hidden)
        samples.append(proc.process_samples(paths, log=False))
        if step == 0:
            algo._adapt(samples[0])
    algo.optimize_policy(samples, log=False)
    theta = np.concatenate([v.reshape(-1) for v in policy.get_param_values().values()]).astype(np.float32)
    return theta, algo.last_stats


def test_two_rank_promp_optimize_policy_through_the_collective_hook(tmp_path, monkeypatch):
    """ProMP.optimize_policy end to end on two gloo ranks whose session exchanges through `collective` (no communicator in
    the contexts): the Adam epochs must use the WHOLE meta-batch's gradient -- equal to the one-process run over all four
    tasks, identical on both ranks.  (Before the fix the ranks applied their local sums and drifted apart silently; the
    library now also refuses promp_optimize on a sharded context without a communicator.)"""
    from promp_amd import _lib
    from tests import devlib
    out = str(tmp_path / 'opt')
    script = tmp_path / 'worker.py'
    script.write_text(OPT_WORKER % dict(root=ROOT))
    devlib.emu_library()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29671', str(script)]
    subprocess.run(cmd, check=True, env=dict(os.environ, OUT=out, MASTER_ADDR='127.0.0.1'), timeout=900, cwd=ROOT)
    monkeypatch.setenv('PROMP_EMU_CUS', '2')
    _lib.set_library_for_testing(devlib.emu_library())
    try:
        ref_theta, ref_stats = run_promp_iteration([0, 1, 2, 3], 4, 0, 1, None)
        # a sharded context without a communicator refuses the calls that would use a mean it does not have
        ctx = _lib.Context(2, 5, 3, (32, 32), 1, max_rows=64, max_paths=4, lib=devlib.emu_library(), n_tasks_global=4)
        for call in (lambda: ctx.optimize(1, 1e-3, 0.3, np.array([5e-4], np.float32)),
                     lambda: ctx.constraint_hvp(np.zeros(ctx.n_params, np.float32))):
            try:
                call()
                raise AssertionError('a sharded context without a communicator must refuse')
            except _lib.PrompError as e:
                assert 'no communicator' in str(e)
        ctx.close()
    finally:
        _lib.set_library_for_testing(None)
    r0, r1 = np.load(out + '.0.npz'), np.load(out + '.1.npz')
    assert np.array_equal(r0['theta'], r1['theta'])
    # float32 sums in another task order (rank-major): Adam steps are ~lr per element
    assert np.max(np.abs(r0['theta'] - ref_theta)) < 2e-5 and np.mean(np.abs(r0['theta'] - ref_theta)) < 2e-6
    np.testing.assert_allclose(r0['loss_before'], ref_stats['loss_before'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r0['loss_after'], ref_stats['loss_after'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r0['inner_kl'], ref_stats['inner_kl'], rtol=1e-3, atol=1e-7)
