"""Rendezvous for the RCCL communicator: one process per GPU, started by any launcher that sets RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT -- ``python -m promp_amd.launch`` (this package, no PyTorch) or
``python -m torch.distributed.run``.  Only the 128-byte ncclUniqueId travels over this side channel; the data path is
ncclAllReduce inside libpromp_hip.so (RCCL over xGMI).

Default: rank 0 serves the ids on a plain TCP socket (MASTER_ADDR, PROMP_RDZV_PORT or MASTER_PORT + 1 -- under torchrun
MASTER_PORT itself belongs to the elastic agent's store), one small server thread per process that answers every later exchange
too (one communicator per context).  PyTorch is not imported.  PROMP_RDZV=torch opts into the launcher's TCPStore instead.
"""
import os
import socket
import struct
import threading
import time


def env_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def _recv_exact(c, n):
    buf = b''
    while len(buf) < n:
        chunk = c.recv(n - len(buf))
        if not chunk:
            raise RuntimeError('rendezvous: connection closed early')
        buf += chunk
    return buf


class _IdServer(object):
    """rank 0: answers "which id belongs to exchange number seq" for every rank, as often as they ask"""

    def __init__(self, addr, port):
        self.ids, self.served, self.cond = {}, {}, threading.Condition()
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind((addr, port))
        self.sock.listen(64)
        t = threading.Thread(target=self._serve, daemon=True)
        t.start()

    def publish(self, seq, uid, n_peers, timeout):
        """make the id of exchange `seq` known and wait until every peer has fetched it (rank 0 must not run ahead -- or
        exit -- with ranks still on their way)"""
        deadline = time.time() + timeout
        with self.cond:
            self.ids[seq] = bytes(uid)
            self.cond.notify_all()
            while self.served.get(seq, 0) < n_peers:
                if time.time() > deadline:
                    raise RuntimeError('rendezvous: %d of %d ranks fetched the id of exchange %d within %d s'
                                       % (self.served.get(seq, 0), n_peers, seq, timeout))
                self.cond.wait(1.0)

    def _serve(self):
        while True:
            try:
                c, _ = self.sock.accept()
            except OSError:
                return
            threading.Thread(target=self._answer, args=(c,), daemon=True).start()

    def _answer(self, c):
        try:
            (seq,) = struct.unpack('<i', _recv_exact(c, 4))
            with self.cond:
                while seq not in self.ids:
                    self.cond.wait(1.0)
                uid = self.ids[seq]
            c.sendall(uid)
            with self.cond:
                self.served[seq] = self.served.get(seq, 0) + 1
                self.cond.notify_all()
        except Exception:
            pass
        finally:
            c.close()


_server = None


def _exchange_socket(rank, world, uid, addr, port, timeout, seq=0):
    global _server
    if rank == 0:
        if _server is None:
            _server = _IdServer(addr, port)
        _server.publish(seq, uid, world - 1, timeout)
        return bytes(uid)
    deadline = time.time() + timeout
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.2)
    c.settimeout(timeout)
    c.sendall(struct.pack('<i', seq))
    buf = _recv_exact(c, 128)
    c.close()
    return buf


_store = None     # PROMP_RDZV=torch: one TCPStore client per process, reused by later exchanges
_n_exchanges = 0  # every rank exchanges in the same order, so the counter names the exchange


def exchange_unique_id(rank, world, make_id, timeout=300):
    """rank 0 calls make_id() -> bytes[128]; every rank returns the same bytes.  May be called several times per
    process (one communicator per context): each call is its own exchange."""
    global _store, _n_exchanges
    if world == 1:
        return make_id()
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500'))
    uid = make_id() if rank == 0 else None
    seq = _n_exchanges
    _n_exchanges += 1
    if os.environ.get('PROMP_RDZV', 'socket') != 'torch':
        return _exchange_socket(rank, world, uid, addr, int(os.environ.get('PROMP_RDZV_PORT', port + 1)), timeout, seq)
    from datetime import timedelta
    from torch.distributed import TCPStore        # explicit opt-in only: the product does not need PyTorch
    if _store is None:
        agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '') == 'True'
        _store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent_store), timeout=timedelta(seconds=timeout),
                          wait_for_workers=False)
    key = 'promp_amd/nccl_uid/%s/%d' % (os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), seq)
    if rank == 0:
        _store.set(key, uid)
    return bytes(_store.get(key))
