"""python -m promp_amd.launch --nproc N [--master-addr A] [--master-port P] script.py [args ...]

One process per GPU on this node, no PyTorch: sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT the way
torch.distributed.run does (promp_amd.comm.env_world reads them; the ncclUniqueId crosses through promp_amd.comm's socket on
MASTER_PORT + 1), waits for all ranks, and takes the others down when one fails.  bench.py runs under either launcher."""
import argparse
import os
import subprocess
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m promp_amd.launch')
    ap.add_argument('--nproc', type=int, required=True)
    ap.add_argument('--master-addr', default='127.0.0.1')
    ap.add_argument('--master-port', type=int, default=29500)
    ap.add_argument('script')
    ap.add_argument('args', nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    procs = []
    for r in range(a.nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.nproc), LOCAL_WORLD_SIZE=str(a.nproc),
                   MASTER_ADDR=a.master_addr, MASTER_PORT=str(a.master_port))
        procs.append(subprocess.Popen([sys.executable, a.script] + a.args, env=env))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in alive:            # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            p.terminate()
        rc = 130
    return rc


if __name__ == '__main__':
    sys.exit(main())
