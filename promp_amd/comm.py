"""Rendezvous for the RCCL communicator: one process per GPU, launched by
``python -m torch.distributed.run`` (or any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_ADDR / MASTER_PORT).  Only the 128-byte ncclUniqueId travels over this side channel; the data path
is ncclAllReduce inside libpromp_hip.so (RCCL over xGMI).

Under torchrun the elastic agent already hosts a TCPStore on MASTER_ADDR:MASTER_PORT, so ranks join it as
clients (torch here is launcher plumbing only).  Without torch, rank 0 serves the id on a plain socket.
"""
import os
import socket
import time


def env_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def _exchange_socket(rank, world, uid, addr, port, timeout):
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        for _ in range(world - 1):
            c, _ = srv.accept()
            c.sendall(uid)
            c.close()
        srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.2)
    buf = b''
    while len(buf) < 128:
        chunk = c.recv(128 - len(buf))
        if not chunk:
            raise RuntimeError('rendezvous: connection closed early')
        buf += chunk
    c.close()
    return buf


_store = None     # one TCPStore client per process, reused by later exchanges
_n_exchanges = 0  # every rank exchanges in the same order, so the counter names the exchange


def exchange_unique_id(rank, world, make_id, timeout=300):
    """rank 0 calls make_id() -> bytes[128]; every rank returns the same bytes.  May be called several times per
    process (one communicator per context): each call uses its own key."""
    global _store, _n_exchanges
    if world == 1:
        return make_id()
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500'))
    uid = make_id() if rank == 0 else None
    seq = _n_exchanges
    _n_exchanges += 1
    try:
        from datetime import timedelta
        from torch.distributed import TCPStore
    except ImportError:
        return _exchange_socket(rank, world, uid, addr, port + 1, timeout)
    if _store is None:
        agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '') == 'True'
        _store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent_store), timeout=timedelta(seconds=timeout),
                          wait_for_workers=False)
    key = 'promp_amd/nccl_uid/%s/%d' % (os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), seq)
    if rank == 0:
        _store.set(key, uid)
    return bytes(_store.get(key))
