"""ctypes front end of oracle/promp_cpu.c (TEST / MEASUREMENT INFRASTRUCTURE ONLY, see the C file's header).

build() compiles the C restatement with gcc -O3 -march=native -fopenmp into oracle/_build/ (kept out of history);
CpuPort runs one hot-path step on synthetic fixed-length batches the way bench.py's GPU iteration does."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'promp_cpu.c')
OUT = os.path.join(HERE, '_build', 'libpromp_cpu.so')
KIND = dict(ratio=0, clip=1, loglik=2, kl=3)


class Dims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('M', 'P', 'T', 'O', 'A', 'H1', 'H2')]


def build(force=False):
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(['gcc', '-O3', '-march=native', '-fopenmp', '-shared', '-fPIC', SRC, '-o', OUT, '-lm'])
    return OUT


_F = C.POINTER(C.c_float)


def _f(a):
    return a.ctypes.data_as(_F)


class CpuPort(object):
    def __init__(self, M, P, T, O, A, hidden):
        self.lib = C.CDLL(build())
        self.lib.pc_threads.restype = C.c_int
        self.d = Dims(M, P, T, O, A, hidden[0], hidden[1])
        self.n_params = O * hidden[0] + hidden[0] + hidden[0] * hidden[1] + hidden[1] + hidden[1] * A + 2 * A
        self.M, self.N = M, P * T

    def threads(self):
        return int(self.lib.pc_threads())

    def set_threads(self, n):
        self.lib.pc_set_threads(int(n))

    def process_samples(self, obs, rew, discount=0.99, gae_lambda=1.0, reg=1e-5, normalize_adv=True):
        obs, rew = np.ascontiguousarray(obs, np.float32), np.ascontiguousarray(rew, np.float32)
        adv = np.empty(self.M * self.N, np.float32)
        ret = np.empty(self.M * self.N, np.float64)
        self.lib.pc_process_samples(C.byref(self.d), _f(obs), _f(rew), C.c_double(discount), C.c_double(gae_lambda), C.c_double(reg),
                                    int(normalize_adv), _f(adv), ret.ctypes.data_as(C.POINTER(C.c_double)))
        return adv, ret

    @staticmethod
    def _slab(s):
        return [np.ascontiguousarray(s[k], np.float32) for k in ('obs', 'act', 'adv', 'old_mean', 'old_log_std')]

    def adapt(self, theta, alpha, slab, inner='ratio'):
        theta = np.ascontiguousarray(theta, np.float32)
        per_task = int(theta.ndim == 2)
        out = np.empty((self.M, self.n_params), np.float32)
        a = self._slab(slab)
        self.lib.pc_adapt(C.byref(self.d), _f(theta), per_task, _f(np.ascontiguousarray(alpha, np.float32)), KIND[inner],
                          *[_f(x) for x in a], _f(out))
        return out

    def meta_grad(self, theta, alpha, eta, clip_eps, slab0, slab1, inner='ratio', outer='clip', want_grad=True):
        theta = np.ascontiguousarray(theta, np.float32)
        grad, stats = np.zeros(self.n_params, np.float32), np.zeros(3, np.float32)
        a0, a1 = self._slab(slab0), self._slab(slab1)
        self.lib.pc_meta_grad(C.byref(self.d), _f(theta), _f(np.ascontiguousarray(alpha, np.float32)), C.c_float(eta),
                              C.c_float(clip_eps), KIND[inner], KIND[outer], int(want_grad), *[_f(x) for x in a0 + a1],
                              _f(grad), _f(stats))
        return grad, dict(loss=float(stats[0]), inner_kl=float(stats[1]), outer_kl=float(stats[2]))

    def adam(self, theta, m, v, t, grad, lr):
        tt = C.c_longlong(t)
        self.lib.pc_adam(self.n_params, _f(theta), _f(m), _f(v), C.byref(tt), _f(np.ascontiguousarray(grad, np.float32)), C.c_float(lr))
        return int(tt.value)

    def promp_step(self, theta, alpha, eta, clip_eps, lr, epochs, raw0, raw1, opts):
        """one bench.py iteration: process_samples x2, _adapt, E x (meta-gradient, Adam), stats"""
        theta = np.array(theta, np.float32)
        adv0, _ = self.process_samples(raw0['obs'], raw0['rew'], **opts)
        s0 = dict(raw0, adv=adv0)
        self.adapt(theta, alpha, s0)
        adv1, _ = self.process_samples(raw1['obs'], raw1['rew'], **opts)
        s1 = dict(raw1, adv=adv1)
        m, v, t = np.zeros_like(theta), np.zeros_like(theta), 0
        first = None
        for _ in range(epochs):
            g, st = self.meta_grad(theta, alpha, eta, clip_eps, s0, s1)
            first = st if first is None else first
            t = self.adam(theta, m, v, t, g, lr)
        _, st = self.meta_grad(theta, alpha, eta, clip_eps, s0, s1, want_grad=False)
        return theta, dict(loss_before=first['loss'] if first else st['loss'], loss_after=st['loss'], inner_kl=st['inner_kl'],
                           outer_kl=st['outer_kl'])
