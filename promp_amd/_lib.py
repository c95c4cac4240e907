"""ctypes binding of libpromp_hip.so (include/promp_hip.h).

The product binds exactly one library: ``promp_amd/libpromp_hip.so`` built in-tree by
``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU fallback: if the library is missing or no
HIP device is usable, loading / context creation raises.

``Library(path)`` with an explicit path exists for the test-suite, which also runs the same C ABI
compiled against the SIMT interpreter in tests/emu (kernel-source verification in a GPU-less container).
"""
import ctypes as C
import os

import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, 'libpromp_hip.so')

BASELINE_ZERO, BASELINE_LINEAR_FEATURE, BASELINE_LINEAR_TIME = 0, 1, 2
INNER_RATIO, INNER_LOGLIK, INNER_DICE = 0, 1, 2
OUTER_CLIP, OUTER_RATIO, OUTER_KL, OUTER_LOGLIK = 0, 1, 2, 3
LOSS_RATIO, LOSS_CLIP, LOSS_LOGLIK, LOSS_KL = 0, 1, 2, 3
KERNEL_FWD_BWD, KERNEL_HVP, KERNEL_GRAM, KERNEL_FWD, KERNEL_EXCHANGE = 0, 1, 2, 3, 4


class PrompError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [('n_tasks', C.c_int32), ('n_tasks_global', C.c_int32), ('obs_dim', C.c_int32), ('act_dim', C.c_int32),
                ('hidden1', C.c_int32), ('hidden2', C.c_int32), ('num_inner_steps', C.c_int32),
                ('max_rows', C.c_int32), ('max_paths', C.c_int32),
                ('n_hidden', C.c_int32), ('hidden3', C.c_int32), ('hidden4', C.c_int32), ('hidden_act', C.c_int32)]


class ProcOpts(C.Structure):
    _fields_ = [('discount', C.c_double), ('gae_lambda', C.c_double), ('reg_coeff', C.c_double),
                ('normalize_adv', C.c_int32), ('positive_adv', C.c_int32), ('baseline_kind', C.c_int32),
                ('reserved', C.c_int32)]


class PointEnvOpts(C.Structure):
    _fields_ = [('normalization_scale', C.c_double), ('max_step', C.c_double), ('sparse_radius', C.c_double),
                ('reward_type', C.c_int32), ('clip_infos', C.c_int32), ('seed', C.c_uint64)]


POINT_REWARD = dict(dense=0, dense_squared=1, sparse=2)

_F = C.POINTER(C.c_float)
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)
_P = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/promp_hip.h
SIGNATURES = {
    'promp_ctx_create': (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(Dims)]),
    'promp_ctx_destroy': (None, [_P]),
    'promp_last_error': (C.c_char_p, []),
    'promp_abi_version': (C.c_int, []),
    'promp_param_count': (C.c_int, [C.POINTER(Dims)]),
    'promp_feature_dim': (C.c_int, [C.POINTER(Dims), C.c_int]),
    'promp_sync': (C.c_int, [_P]),
    'promp_upload_step': (C.c_int, [_P, C.c_int, C.c_int, _I, _I, _F, _F, _F, _F, _F, C.c_int]),
    'promp_process_samples': (C.c_int, [_P, C.c_int, C.POINTER(ProcOpts)]),
    'promp_download_processed': (C.c_int, [_P, C.c_int, _F, _F, _D, _D, _D, _D]),
    'promp_download_raw': (C.c_int, [_P, C.c_int, _D, _D]),
    'promp_set_coeffs': (C.c_int, [_P, C.c_int, C.c_int, _D]),
    'promp_predict_baseline': (C.c_int, [_P, C.c_int, C.c_int, _D]),
    'promp_set_advantages': (C.c_int, [_P, C.c_int, _F]),
    'promp_set_theta': (C.c_int, [_P, _F]),
    'promp_get_theta': (C.c_int, [_P, _F]),
    'promp_set_step_sizes': (C.c_int, [_P, _F]),
    'promp_set_learn_std': (C.c_int, [_P, C.c_int]),
    'promp_set_primal_cache': (C.c_int, [_P, C.c_int]),
    'promp_set_reuse_adapt': (C.c_int, [_P, C.c_int]),
    'promp_adapt_passes_skipped': (C.c_longlong, [_P]),
    'promp_constraint_hvp_cached_passes': (C.c_longlong, [_P]),
    'promp_state_version': (C.c_longlong, [_P]),
    'promp_set_min_std': (C.c_int, [_P, C.c_float]),
    'promp_set_schedule': (C.c_int, [_P, C.c_int, C.c_int]),
    'promp_set_rewards_f64': (C.c_int, [_P, C.c_int, _D]),
    'promp_stage_step': (C.c_int, [_P, C.c_int, C.c_int, _I, _I, _F, _F, _F, _F, _F, C.c_int]),
    'promp_commit_step': (C.c_int, [_P, C.c_int]),
    'promp_stage_wait': (C.c_int, [_P]),
    'promp_host_alloc': (C.c_void_p, [C.c_size_t]),
    'promp_host_free': (None, [C.c_void_p]),
    'promp_constraint_hvp': (C.c_int, [_P, C.c_int, _F, C.c_int, _F]),
    'promp_cg_solve': (C.c_int, [_P, C.c_int, _F, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, _F, C.POINTER(C.c_double)]),
    'promp_set_adam_state': (C.c_int, [_P, _F, _F, C.c_int64]),
    'promp_get_adam_state': (C.c_int, [_P, _F, _F, C.POINTER(C.c_int64)]),
    'promp_switch_to_pre_update': (C.c_int, [_P]),
    'promp_set_task_thetas': (C.c_int, [_P, _F]),
    'promp_get_task_thetas': (C.c_int, [_P, _F]),
    'promp_inner_adapt': (C.c_int, [_P, C.c_int, C.c_int]),
    'promp_policy_forward': (C.c_int, [_P, _F, C.c_int, _F]),
    'promp_rollout_point_env': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _D, _D, _F, C.POINTER(PointEnvOpts)]),
    'promp_begin_rollout': (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    'promp_begin_collection': (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    'promp_end_collection': (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _F]),
    'promp_policy_step': (C.c_int, [_P, C.c_int, C.c_int, _F, C.c_uint64, C.c_int, _F]),
    'promp_set_rewards': (C.c_int, [_P, C.c_int, _F]),
    'promp_download_step': (C.c_int, [_P, C.c_int, _F, _F, _F, _F, _F]),
    'promp_meta_grad': (C.c_int, [_P, C.c_float, _F, C.c_int, C.c_int, _F, _F]),
    'promp_adam_step': (C.c_int, [_P, C.c_float]),
    'promp_optimize': (C.c_int, [_P, C.c_int, C.c_float, C.c_float, _F, C.c_int, C.c_int, _F, _F]),
    'promp_set_dice_rewards': (C.c_int, [_P, C.c_int, _F]),
    'promp_optimize_begin': (C.c_int, [_P, C.c_int, C.c_float, C.c_float, _F, C.c_int, C.c_int]),
    'promp_optimize_end': (C.c_int, [_P, _F, _F]),
    'promp_comm_unique_id': (C.c_int, [_P, C.c_size_t]),
    'promp_comm_init': (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    'promp_comm_move': (C.c_int, [_P, _P]),
    'promp_comm_split_path': (C.c_int, [_P, C.c_int]),
    'promp_comm_info': (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]),
    'promp_comm_fixed_order': (C.c_int, [_P, C.c_int]),
    'promp_reduced_get': (C.c_int, [_P, _F]),
    'promp_reduced_set': (C.c_int, [_P, _F]),
    'promp_allreduce_f64': (C.c_int, [_P, _D, C.c_int, C.c_int]),
    'promp_eval_loss_grad': (C.c_int, [_P, C.c_int, C.c_int, C.c_float, C.c_int, _F, _F, _F]),
    'promp_eval_hvp': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, _F, _F]),
    'promp_prof_enable': (C.c_int, [_P, C.c_int]),
    'promp_prof_read': (C.c_int, [_P, C.c_int, _D, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'promp_split_events': (C.c_int, [_P, C.POINTER(C.c_int64)]),
    'promp_device_info': (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}


class Library:
    def __init__(self, path=None):
        self.path = path or DEFAULT_LIBRARY
        if not os.path.exists(self.path):
            raise PrompError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                             '(hipcc --offload-arch=gfx950).  promp_amd has no CPU fallback.' % self.path)
        self.cdll = C.CDLL(self.path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)   # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args

    def check(self, rc):
        if rc < 0:
            raise PrompError(self.cdll.promp_last_error().decode())
        return rc


_default = None


def get_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


def set_library_for_testing(lib):
    """Test hook: make ``lib`` (a ``Library``) the one new contexts bind.  Used by tests/ to run the plugin
    classes against tests/emu/libpromp_emu.so; never called by the package itself."""
    global _default
    _default = lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def pinned_empty(lib, shape, dtype=np.float32):
    """an uninitialised ndarray in page-locked host memory (promp_host_alloc): DMA source of Context.stage_step"""
    import weakref
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    count = int(np.prod(shape))
    nbytes = max(count * np.dtype(dtype).itemsize, 1)
    ptr = lib.cdll.promp_host_alloc(C.c_size_t(nbytes))
    if not ptr:
        raise PrompError(lib.cdll.promp_last_error().decode())
    raw = (C.c_char * nbytes).from_address(ptr)
    weakref.finalize(raw, lib.cdll.promp_host_free, C.c_void_p(ptr))     # every view keeps `raw` alive through .base
    return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)


class _HostPool(object):
    """Large host arrays for the plugin path's uploads and downloads, recycled once nobody else references them.

    The plugin API hands NumPy arrays to the caller (flattened observations, returns, advantages ...) and the caller may keep
    them, so a buffer is reused only when its reference count says that no view of it is alive any more.  Fresh allocations
    of this size (3 - 13 MB) cost more in page faults than the copy they receive; with a library at hand the buffers are
    page-locked (promp_host_alloc): the copies to and from the device then run at PCIe speed instead of through the
    runtime's bounce buffers.

    Requests are rounded up to a SIZE CLASS (1.25x geometric steps from 64 KB), and any free owner of the class serves them:
    environments that end episodes early (Ant, Walker, Hopper, Humanoid) ask for a different row count every iteration, and
    with exact sizes every iteration would leave eight more page-locked multi-MB buffers behind.  The pool also keeps at most
    `max_bytes` of buffers; beyond that the least recently used FREE owners are dropped (their memory goes back through the
    finalizer once the last view dies).  The per-class `keep` is deliberately generous: a training loop holds two generations of
    a step's buffers (this batch's SamplesData while the previous batch's are being released), and dropping / re-allocating
    page-locked memory every step costs milliseconds (measured: 19.7 vs 11 ms per plugin step with keep = 8).  Thread-safe: one lock around the bookkeeping (LinearBaseline's fit thread and the
    sampler's thread both draw from it)."""
    _MIN_CLASS = 1 << 16

    def __init__(self, keep=64, max_bytes=None):
        import os, threading
        if max_bytes is None:        # PROMP_HOST_POOL_MB: the cap per process (N ranks per node hold N pools of page-locked memory)
            max_bytes = int(float(os.environ.get('PROMP_HOST_POOL_MB', '1024')) * (1 << 20))
        self._owners = {}            # (class bytes, pinned) -> [owner, ...], most recently used last
        self._used = {}              # id(owner) -> tick of its last hand-out (eviction order over ALL classes)
        self._tick = 0
        self._keep, self._max_bytes, self._bytes = keep, int(max_bytes), 0
        self._lock = threading.Lock()

    @classmethod
    def size_class(cls, nbytes):
        c = cls._MIN_CLASS
        while c < nbytes:
            c = (c + (c >> 2) + 4095) & ~4095
        return c

    @staticmethod
    def _free(lst, i):
        import sys
        cand = lst[i]
        return sys.getrefcount(cand) == 3         # the pool's list, `cand`, getrefcount's argument: no view outside (CPython)

    def retained_bytes(self):
        return self._bytes

    def retained_owners(self):
        return sum(len(v) for v in self._owners.values())

    def get(self, shape, dtype, lib=None):
        shape = tuple(int(x) for x in shape)
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        cls_bytes = self.size_class(max(count * dt.itemsize, 1))
        key = (cls_bytes, lib is not None)
        with self._lock:
            lst = self._owners.setdefault(key, [])
            owner = None
            for i in range(len(lst)):
                if self._free(lst, i):
                    owner = lst.pop(i)
                    break
            if owner is None:
                owner = self._new_owner(cls_bytes, lib)
                self._bytes += cls_bytes
            lst.append(owner)                    # most recently used last
            self._tick += 1
            self._used[id(owner)] = self._tick
            if len(lst) > self._keep:
                self._drop_free(lst, cls_bytes, len(lst) - self._keep, skip=owner)
            if self._bytes > self._max_bytes:
                self._evict(skip=owner)
        return owner[:count * dt.itemsize].view(dt).reshape(shape)

    def _drop_free(self, lst, cls_bytes, n, skip):
        i = 0
        while n > 0 and i < len(lst):
            if lst[i] is not skip and self._free(lst, i):
                self._used.pop(id(lst[i]), None)
                del lst[i]
                self._bytes -= cls_bytes
                n -= 1
            else:
                i += 1

    def _evict(self, skip):
        # over the cap: the FREE owners go, least recently handed out first, whatever their size class (owners somebody still
        # references stay -- their memory is the caller's until the last view dies)
        order = sorted(((self._used.get(id(o), 0), key, o) for key, lst in self._owners.items() for o in lst), key=lambda e: e[0])
        for _, key, owner in order:
            if self._bytes <= self._max_bytes:
                break
            lst = self._owners[key]
            for i in range(len(lst)):
                if lst[i] is owner:
                    if owner is not skip and self._free_at(lst, i):
                        self._used.pop(id(owner), None)
                        del lst[i]
                        self._bytes -= key[0]
                    break
        for key in [k for k, v in self._owners.items() if not v]:
            del self._owners[key]

    @staticmethod
    def _free_at(lst, i):
        import sys
        # (the pool's list, _evict's `order` entry, its loop variable, getrefcount's argument: no view outside)
        return sys.getrefcount(lst[i]) == 4

    @staticmethod
    def _new_owner(nbytes, lib):
        if lib is not None:
            try:
                import weakref
                ptr = lib.cdll.promp_host_alloc(C.c_size_t(nbytes))
                if ptr:
                    raw = (C.c_char * nbytes).from_address(ptr)
                    weakref.finalize(raw, lib.cdll.promp_host_free, C.c_void_p(ptr))
                    return np.frombuffer(raw, dtype=np.uint8, count=nbytes)      # views collapse their .base onto this array
            except Exception:
                pass
        return np.empty(nbytes, np.uint8)


host_pool = _HostPool()


class LazyResults(object):
    """The per-row results of one promp_process_samples call (float32 advantages, float64 returns and raw advantages) as they sit
    on the device, fetched over PCIe on FIRST USE.  The plugin classes hand these out instead of downloading 3.2 MB per
    sampling step that the algorithms never read (they work on the device-resident copy).  A context materialises every live
    LazyResults of a step right before anything overwrites that step's arrays (Context._pre_write), so a holder can never
    return data of a later batch: either it fetched in time, or it fetches now from an unchanged step."""
    __slots__ = ('ctx', 'step', '_data', '__weakref__')
    fetch_count = 0        # downloads so far, process-wide (tests, benchmarks)

    def __init__(self, ctx, step):
        self.ctx, self.step, self._data = ctx, int(step), None

    def materialize(self):
        if self._data is None:
            ctx = self.ctx
            if not ctx._h:
                raise PrompError('the device context of these results is closed')
            R = ctx.step_rows[self.step]
            adv = host_pool.get((R,), np.float32, ctx.lib)
            ctx._call('promp_download_processed', self.step, None, _ptr(adv, C.c_float), None, None, None, None)
            ret, raw = ctx.download_raw(self.step)
            self._data = dict(advantages=adv, returns=ret, raw_advantages=raw)
            self.ctx = None
            LazyResults.fetch_count += 1
        return self._data

    @property
    def fetched(self):
        return self._data is not None


class LazyCompute(object):
    """LazyResults' protocol for host-side expressions: fn() -> dict(field -> array), evaluated on first use"""
    __slots__ = ('_fn', '_data')

    def __init__(self, fn):
        self._fn, self._data = fn, None

    def materialize(self):
        if self._data is None:
            self._data, self._fn = self._fn(), None
        return self._data

    @property
    def fetched(self):
        return self._data is not None


class LazyRows(np.lib.mixins.NDArrayOperatorsMixin):
    """rows [a, b) of one field of a LazyResults: converts to the ndarray on first use (np.asarray, arithmetic, indexing, any
    ndarray attribute); shape / dtype / len are known without fetching"""
    __slots__ = ('_res', '_field', '_a', '_b', '_dtype')
    _DTYPES = dict(advantages=np.dtype(np.float32), returns=np.dtype(np.float64), raw_advantages=np.dtype(np.float64))
    __array_priority__ = 100

    def __init__(self, res, field, a, b, dtype=None):
        self._res, self._field, self._a, self._b = res, field, int(a), int(b)
        self._dtype = np.dtype(dtype) if dtype is not None else self._DTYPES[field]

    def _arr(self):
        return self._res.materialize()[self._field][self._a:self._b]

    def __array__(self, dtype=None, copy=None):
        a = self._arr()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x._arr() if isinstance(x, LazyRows) else x for x in inputs)
        if 'out' in kwargs:
            kwargs['out'] = tuple(x._arr() if isinstance(x, LazyRows) else x for x in kwargs['out'])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __getitem__(self, k):
        return self._arr()[k]

    def __len__(self):
        return self._b - self._a

    def __iter__(self):
        return iter(self._arr())

    shape = property(lambda self: (self._b - self._a,))
    dtype = property(lambda self: self._dtype)
    ndim = 1
    size = property(lambda self: self._b - self._a)

    def __getattr__(self, name):            # mean(), reshape(), astype(), ... : the ndarray's
        if name.startswith('__'):
            raise AttributeError(name)
        return getattr(self._arr(), name)

    def __reduce__(self):                   # pickle / joblib / copy: the rows themselves (a LazyRows holds a device context)
        return (np.array, (self._arr(),))

    def __repr__(self):
        return 'LazyRows(%s[%d:%d], %s)' % (self._field, self._a, self._b, 'fetched' if self._res.fetched else 'on the device')


HIDDEN_ACTS = dict(tanh=0, relu=1, identity=2)


def hidden_act_id(act):
    """the reference's hidden_nonlinearity argument (policies/base.py:31, mlp.py:47: a TF function or None) -> promp_dims.hidden_act:
    'tanh' / a callable named tanh -> 0, 'relu' / a callable named relu -> 1, None / 'identity' / 'linear' -> 2 (linear hidden
    layers, as tf.layers.dense(activation=None) builds them); anything else is refused -- never silently replaced"""
    if act is None:
        return HIDDEN_ACTS['identity']
    name = act if isinstance(act, str) else getattr(act, '__name__', '')
    name = {'linear': 'identity'}.get(name, name)
    if name not in HIDDEN_ACTS:
        raise PrompError('hidden nonlinearity %r unsupported: tanh, relu or None (identity)' % (act,))
    return HIDDEN_ACTS[name]


OUTPUT_ACTS = dict(identity=0, tanh=1, relu=2)      # PROMP_OUT_ACT_*: bits 8.. of promp_dims.hidden_act
OUT_ACT_SHIFT = 8


def output_act_id(act):
    """the reference's output_nonlinearity argument (policies/networks/mlp.py:53-60, 114-117: a TF function applied to the mean
    network's last layer, or None -- what every run script passes) -> PROMP_OUT_ACT_*; anything else is refused by name"""
    if act is None:
        return OUTPUT_ACTS['identity']
    name = act if isinstance(act, str) else getattr(act, '__name__', '')
    name = {'linear': 'identity'}.get(name, name)
    if name not in OUTPUT_ACTS:
        raise PrompError('output nonlinearity %r unsupported: None (linear), tanh or relu' % (act,))
    return OUTPUT_ACTS[name]


class Context:
    """One promp_ctx (one GPU).  Thin, NumPy-in / NumPy-out."""

    def __init__(self, n_tasks, obs_dim, act_dim, hidden_sizes, num_inner_steps=1, max_rows=0, max_paths=0,
                 n_tasks_global=None, device_id=0, lib=None, hidden_act='tanh', output_act=None):
        self.lib = lib or get_library()
        hs = tuple(int(h) for h in hidden_sizes)
        if not 1 <= len(hs) <= 4:
            raise PrompError('hidden_sizes %r unsupported: 1 to 4 hidden layers' % (hidden_sizes,))
        h4 = hs + (0,) * (4 - len(hs))
        self.dims = Dims(int(n_tasks), int(n_tasks_global or n_tasks), int(obs_dim), int(act_dim), h4[0], h4[1],
                         int(num_inner_steps), int(max_rows), int(max_paths), len(hs), h4[2], h4[3],
                         hidden_act_id(hidden_act) | (output_act_id(output_act) << OUT_ACT_SHIFT))
        self._h = _P()
        self.lib.check(self.lib.cdll.promp_ctx_create(C.byref(self._h), int(device_id), C.byref(self.dims)))
        self.n_params = self.lib.cdll.promp_param_count(C.byref(self.dims))
        self.n_tasks, self.K = int(n_tasks), int(num_inner_steps)
        self.step_rows = {}
        self._staged, self._live_refs = {}, {}
        self._upload_refs = {}       # step -> [(wait epoch, sources)] of promp_upload_step calls whose copies may be in flight
        self._wait_epoch = 0
        self.step_ls_rows = {}
        self.step_paths = {}
        self._lazy = {}              # step -> WeakSet of LazyResults that have not fetched yet

    def close(self):
        if self._h:
            for step in list(self._lazy):
                self._pre_write(step)
            self.lib.cdll.promp_ctx_destroy(self._h)
            self._h = _P()

    def lazy_results(self, step):
        """the processed rows of `step` as a holder that downloads on first use (see LazyResults)"""
        res = LazyResults(self, step)
        self._lazy.setdefault(int(step), weakref.WeakSet()).add(res)
        return res

    def _pre_write(self, step):
        """called by everything that overwrites a step's arrays: results handed out lazily and still alive fetch now"""
        live = self._lazy.pop(int(step), None)
        if live:
            for res in list(live):
                res.materialize()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # entry points that return only after the work enqueued before them has completed (the compute stream waits for every upload
    # enqueued before it, so an upload's sources are free once one of these has returned)
    _WAITS = frozenset(('promp_sync', 'promp_download_processed', 'promp_download_raw', 'promp_get_theta', 'promp_stage_wait'))

    def _call(self, name, *args):
        rc = self.lib.check(getattr(self.lib.cdll, name)(self._h, *args))
        if name in self._WAITS:
            self._wait_epoch += 1
        return rc

    def _retain_upload(self, step, refs):
        """Keep the sources of an enqueued upload referenced until its copies have completed: entries older than the last call that
        waited for the device are dropped; a third upload of a step with no such call in between waits itself (enqueueing later
        uploads proves nothing about earlier DMAs -- ADVICE r5)."""
        lst = self._upload_refs.setdefault(int(step), [])
        lst[:] = [e for e in lst if e[0] == self._wait_epoch]
        if len(lst) >= 2:
            self._call('promp_sync')
            del lst[:]
        lst.append((self._wait_epoch, refs))

    # ---- trajectories ----
    def upload_step(self, step, task_path_offsets, path_row_offsets, obs, rew, act=None, old_mean=None,
                    old_log_std=None):
        self._pre_write(step)
        tpo = np.ascontiguousarray(task_path_offsets, dtype=np.int32)
        pro = np.ascontiguousarray(path_row_offsets, dtype=np.int32)
        rew64 = np.ascontiguousarray(rew, dtype=np.float64) if np.asarray(rew).dtype == np.float64 else None
        obs, rew = _f32(obs), _f32(rew)
        per_row = 0
        if act is not None:
            act, old_mean, old_log_std = _f32(act), _f32(old_mean), _f32(old_log_std)
            old_log_std = old_log_std.reshape(-1, act.shape[1])
            if old_log_std.shape[0] == obs.shape[0]:
                per_row = 1          # the reference's layout: agent_infos['log_std'] is [rows, A]
            elif old_log_std.shape[0] == self.n_tasks:
                per_row = 0          # compact: one row per task
            else:
                raise PrompError('old_log_std must be [rows, A] or [n_tasks, A]')
        n_paths = len(pro) - 1
        self._call('promp_upload_step', int(step), int(n_paths), _ptr(tpo, C.c_int32), _ptr(pro, C.c_int32),
                   _ptr(obs, C.c_float), _ptr(act, C.c_float), _ptr(rew, C.c_float), _ptr(old_mean, C.c_float),
                   _ptr(old_log_std, C.c_float), per_row)
        # promp_upload_step returns with its copies enqueued: page-locked sources (host_pool's buffers) must not be recycled while the
        # DMA may still read them.  The context keeps them referenced -- the pool only reuses buffers nobody references -- until a
        # call that waited for the device has returned since (_retain_upload).
        self._retain_upload(step, (tpo, pro, obs, act, rew, old_mean, old_log_std, rew64))
        self.step_rows[step] = int(pro[-1])
        self.step_paths[step] = int(n_paths)
        self.step_ls_rows[step] = int(pro[-1]) if per_row else self.n_tasks
        if rew64 is not None:        # float64 rewards stay float64 on the device (the reference scans the env's float64)
            self._call('promp_set_rewards_f64', int(step), _ptr(rew64, C.c_double))

    def stage_step(self, step, task_path_offsets, path_row_offsets, obs, rew, act, old_mean, old_log_std):
        """promp_stage_step: upload_step's arguments, copied on the copy stream into the step's second slab set; call
        commit_step(step) to make it current.  The arrays are read after the call returns when they are pinned
        (pinned_empty): they are kept referenced here, and must not be written until stage_wait() or the commit's first use."""
        self._pre_write(step)
        tpo = np.ascontiguousarray(task_path_offsets, dtype=np.int32)
        pro = np.ascontiguousarray(path_row_offsets, dtype=np.int32)
        obs, rew, act, old_mean, old_log_std = _f32(obs), _f32(rew), _f32(act), _f32(old_mean), _f32(old_log_std)
        old_log_std = old_log_std.reshape(-1, act.shape[1])
        if old_log_std.shape[0] not in (obs.shape[0], self.n_tasks):
            raise PrompError('old_log_std must be [rows, A] or [n_tasks, A]')
        per_row = 1 if old_log_std.shape[0] == obs.shape[0] and obs.shape[0] != self.n_tasks else 0
        n_paths = len(pro) - 1
        self._call('promp_stage_step', int(step), int(n_paths), _ptr(tpo, C.c_int32), _ptr(pro, C.c_int32),
                   _ptr(obs, C.c_float), _ptr(act, C.c_float), _ptr(rew, C.c_float), _ptr(old_mean, C.c_float),
                   _ptr(old_log_std, C.c_float), per_row)
        self._staged[step] = dict(rows=int(pro[-1]), paths=int(n_paths), ls_rows=int(pro[-1]) if per_row else self.n_tasks,
                                  refs=(obs, rew, act, old_mean, old_log_std))

    def commit_step(self, step):
        self._pre_write(step)
        st = self._staged.pop(step)
        self._call('promp_commit_step', int(step))
        self.step_rows[step], self.step_paths[step], self.step_ls_rows[step] = st['rows'], st['paths'], st['ls_rows']
        self._live_refs[step] = st['refs']       # until the next commit: the copies may still be in flight

    def stage_wait(self):
        self._call('promp_stage_wait')

    def process_samples(self, step, discount=0.99, gae_lambda=1.0, normalize_adv=False, positive_adv=False,
                        baseline_kind=BASELINE_LINEAR_FEATURE, reg_coeff=1e-5):
        self._pre_write(step)
        o = ProcOpts(float(discount), float(gae_lambda), float(reg_coeff), int(bool(normalize_adv)),
                     int(bool(positive_adv)), int(baseline_kind), 0)
        self._call('promp_process_samples', int(step), C.byref(o))
        self._last_kind = int(baseline_kind)

    # The downloads in two halves: the destination arrays first (page-locked, recycled), the blocking copies into them later -- a
    # caller with host work that needs the arrays but not yet their contents (views per path and per task: samplers/base.py)
    # does it in between, while the device is still busy with the upload and the sample-processing kernels.
    def alloc_processed(self, step, baseline_kind=None, want_returns32=True, want_advantages=True):
        R, P = self.step_rows[step], self.step_paths[step]
        kind = self._last_kind if baseline_kind is None else baseline_kind
        D = self.lib.cdll.promp_feature_dim(C.byref(self.dims), int(kind))
        return dict(returns=host_pool.get((R,), np.float32, self.lib) if want_returns32 else None,
                    advantages=host_pool.get((R,), np.float32, self.lib) if want_advantages else None,
                    coeffs=np.zeros((self.n_tasks, D), np.float64), path_returns0=np.empty(P, np.float64),
                    path_undiscounted=np.empty(P, np.float64), path_reward_sumsq=np.empty(P, np.float64))

    def fetch_processed(self, step, out):
        D = out['coeffs'].shape[1]
        self._call('promp_download_processed', int(step), _ptr(out['returns'], C.c_float),
                   _ptr(out['advantages'], C.c_float), _ptr(out['coeffs'], C.c_double) if D else None,
                   _ptr(out['path_returns0'], C.c_double), _ptr(out['path_undiscounted'], C.c_double),
                   _ptr(out['path_reward_sumsq'], C.c_double))
        return out

    def download_processed(self, step, baseline_kind=None, want_returns32=True, want_advantages=True):
        return self.fetch_processed(step, self.alloc_processed(step, baseline_kind, want_returns32, want_advantages))

    def alloc_raw(self, step):
        R = self.step_rows[step]
        return host_pool.get((R,), np.float64, self.lib), host_pool.get((R,), np.float64, self.lib)

    def fetch_raw(self, step, ret, adv):
        self._call('promp_download_raw', int(step), _ptr(ret, C.c_double), _ptr(adv, C.c_double))
        return ret, adv

    def download_raw(self, step):
        return self.fetch_raw(step, *self.alloc_raw(step))

    def set_coeffs(self, step, baseline_kind, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        self._call('promp_set_coeffs', int(step), int(baseline_kind), _ptr(coeffs, C.c_double))

    def predict_baseline(self, step, baseline_kind):
        out = np.empty(self.step_rows[step], np.float64)
        self._call('promp_predict_baseline', int(step), int(baseline_kind), _ptr(out, C.c_double))
        return out

    def set_advantages(self, step, adv):
        self._pre_write(step)
        adv = _f32(adv)
        assert adv.shape == (self.step_rows[step],)
        self._call('promp_set_advantages', int(step), _ptr(adv, C.c_float))

    def set_dice_rewards(self, step, rw):
        """promp_set_dice_rewards: per-row DiCE rewards (valid rows, pre-scaled); the device derives the gradient weights"""
        self._pre_write(step)
        rw = _f32(rw)
        assert rw.shape == (self.step_rows[step],)
        self._call('promp_set_dice_rewards', int(step), _ptr(rw, C.c_float))

    # ---- parameters ----
    def set_theta(self, theta):
        theta = _f32(theta)
        assert theta.shape == (self.n_params,)
        self._call('promp_set_theta', _ptr(theta, C.c_float))

    def get_theta(self):
        out = np.empty(self.n_params, np.float32)
        self._call('promp_get_theta', _ptr(out, C.c_float))
        return out

    def set_step_sizes(self, s):
        s = _f32(np.broadcast_to(np.asarray(s, dtype=np.float32), (self.n_params,)))
        self._call('promp_set_step_sizes', _ptr(s, C.c_float))

    def set_adam_state(self, m, v, t):
        m, v = _f32(m), _f32(v)
        self._call('promp_set_adam_state', _ptr(m, C.c_float), _ptr(v, C.c_float), int(t))

    def get_adam_state(self):
        m, v, t = np.empty(self.n_params, np.float32), np.empty(self.n_params, np.float32), C.c_int64(0)
        self._call('promp_get_adam_state', _ptr(m, C.c_float), _ptr(v, C.c_float), C.byref(t))
        return m, v, int(t.value)

    def switch_to_pre_update(self):
        self._call('promp_switch_to_pre_update')

    def set_task_thetas(self, th):
        th = _f32(th)
        assert th.shape == (self.n_tasks, self.n_params)
        self._call('promp_set_task_thetas', _ptr(th, C.c_float))

    def get_task_thetas(self):
        out = np.empty((self.n_tasks, self.n_params), np.float32)
        self._call('promp_get_task_thetas', _ptr(out, C.c_float))
        return out

    def policy_forward(self, obs):
        """obs [n_tasks, B, O] -> means [n_tasks, B, A] under every task's current parameters"""
        obs = _f32(obs)
        assert obs.ndim == 3 and obs.shape[0] == self.n_tasks
        out = np.empty((self.n_tasks, obs.shape[1], self.dims.act_dim), np.float32)
        self._call('promp_policy_forward', _ptr(obs, C.c_float), int(obs.shape[1]), _ptr(out, C.c_float))
        return out

    def rollout_point_env(self, step, goals, start, noise=None, clip_infos=True, reward_type='dense', normalization_scale=0.0,
                          max_step=0.2, sparse_radius=0.5, seed=0, path_length=None):
        """Device rollout of the 2-D point-mass meta-environment (normalize(MetaPointEnvCorner()) with normalization_scale=10; 0 = the bare environment):
        goals [M,2], start [M,B,2] (float64), noise [M,B,T,2] standard normals or None (drawn on the device from `seed`;
        then path_length is required) -> fills step `step`'s slab with M*B paths of length T (see promp_rollout_point_env)."""
        self._pre_write(step)
        goals = np.ascontiguousarray(goals, dtype=np.float64)
        start = np.ascontiguousarray(start, dtype=np.float64)
        M, B = self.n_tasks, start.shape[1]
        if noise is not None:
            noise = _f32(noise)
            T = noise.shape[2]
            assert noise.shape == (M, B, T, 2)
        else:
            T = int(path_length)
        assert goals.shape == (M, 2) and start.shape == (M, B, 2)
        opts = PointEnvOpts(float(normalization_scale), float(max_step), float(sparse_radius), POINT_REWARD[reward_type],
                            int(bool(clip_infos)), int(seed))
        self._call('promp_rollout_point_env', int(step), int(B), int(T), _ptr(goals, C.c_double), _ptr(start, C.c_double),
                   _ptr(noise, C.c_float) if noise is not None else None, C.byref(opts))
        self.step_rows[step], self.step_paths[step] = M * B * T, M * B
        self.step_ls_rows[step] = M

    def begin_rollout(self, step, envs_per_task, path_length):
        """lay step `step` out as n_tasks * envs_per_task fixed-length paths for policy_step to fill"""
        self._pre_write(step)
        self._call('promp_begin_rollout', int(step), int(envs_per_task), int(path_length))
        self._rollout_shape = (int(envs_per_task), int(path_length))
        self.step_rows[step] = self.n_tasks * envs_per_task * path_length
        self.step_paths[step] = self.n_tasks * envs_per_task
        self.step_ls_rows[step] = self.n_tasks

    def begin_collection(self, step, envs_per_task, max_steps):
        """ragged collection: policy_step files vectorised step s under (s, environment) in a staging area until
        end_collection copies the finished episodes into the slab"""
        self._pre_write(step)
        self._call('promp_begin_collection', int(step), int(envs_per_task), int(max_steps))
        self._rollout_shape = (int(envs_per_task), int(max_steps))

    def end_collection(self, step, task_path_offsets, path_env, path_start, path_len, rewards):
        """the finished episodes, in path order: path p = steps [path_start[p], path_start[p] + path_len[p]) of environment
        path_env[p]; rewards [rows] in path order"""
        self._pre_write(step)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        tpo, env, start, ln = i32(task_path_offsets), i32(path_env), i32(path_start), i32(path_len)
        rew = _f32(rewards).reshape(-1)
        assert tpo.shape == (self.n_tasks + 1,) and env.shape == start.shape == ln.shape and rew.size == int(ln.sum())
        self._call('promp_end_collection', int(step), int(env.size), tpo.ctypes.data_as(C.c_void_p), env.ctypes.data_as(C.c_void_p),
                   start.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), _ptr(rew, C.c_float))
        self.step_rows[step], self.step_paths[step], self.step_ls_rows[step] = int(ln.sum()), int(env.size), self.n_tasks

    def policy_step(self, step, t, obs, seed=0, clip_infos=True):
        """obs [M, B, O] -> actions [M, B, A]; observation, action and mean land in the slab at row (task, env, t)"""
        B = self._rollout_shape[0]
        obs = _f32(obs).reshape(self.n_tasks, B, self.dims.obs_dim)
        out = np.empty((self.n_tasks, B, self.dims.act_dim), np.float32)
        self._call('promp_policy_step', int(step), int(t), _ptr(obs, C.c_float), int(seed), int(bool(clip_infos)),
                   _ptr(out, C.c_float))
        return out

    def set_rewards(self, step, rewards):
        """float64 rewards stay float64 on the device (the returns / GAE scans read them), anything else goes as float32"""
        self._pre_write(step)
        if np.asarray(rewards).dtype == np.float64:
            r = np.ascontiguousarray(rewards, dtype=np.float64).reshape(-1)
            assert r.size == self.step_rows[step]
            self._call('promp_set_rewards_f64', int(step), _ptr(r, C.c_double))
            return
        r = _f32(rewards).reshape(-1)
        assert r.size == self.step_rows[step]
        self._call('promp_set_rewards', int(step), _ptr(r, C.c_float))

    def download_step(self, step):
        """the slab of a sampling step back on the host: dict(obs, act, rew, old_mean, old_log_std)"""
        R, A, O = self.step_rows[step], self.dims.act_dim, self.dims.obs_dim
        ls_rows = self.step_ls_rows.get(step, R)
        out = dict(obs=np.empty((R, O), np.float32), act=np.empty((R, A), np.float32), rew=np.empty(R, np.float32),
                   old_mean=np.empty((R, A), np.float32), old_log_std=np.empty((ls_rows, A), np.float32))
        self._call('promp_download_step', int(step), _ptr(out['obs'], C.c_float), _ptr(out['act'], C.c_float),
                   _ptr(out['rew'], C.c_float), _ptr(out['old_mean'], C.c_float), _ptr(out['old_log_std'], C.c_float))
        return out

    # ---- algorithm ----
    def inner_adapt(self, step, inner_kind=INNER_RATIO):
        self._call('promp_inner_adapt', int(step), int(inner_kind))

    def meta_grad(self, clip_eps, inner_kl_coeff, inner_kind=INNER_RATIO, outer_kind=OUTER_CLIP):
        eta = _f32(inner_kl_coeff)
        assert eta.shape == (self.K,)
        grad, stats = np.empty(self.n_params, np.float32), np.empty(self.K + 2, np.float32)
        self._call('promp_meta_grad', float(clip_eps), _ptr(eta, C.c_float), int(inner_kind), int(outer_kind),
                   _ptr(grad, C.c_float), _ptr(stats, C.c_float))
        return grad, dict(loss=float(stats[0]), inner_kl=stats[1:1 + self.K].copy(), outer_kl=float(stats[1 + self.K]))

    def meta_eval(self, clip_eps, inner_kl_coeff, inner_kind=INNER_RATIO, outer_kind=OUTER_CLIP):
        """forward-only evaluation of the meta-objective (the compute_stats pass): loss, inner_kl [K], outer_kl"""
        r = self.optimize(0, 0.0, clip_eps, inner_kl_coeff, inner_kind, outer_kind)
        return dict(loss=r['loss_after'], inner_kl=r['inner_kl'], outer_kl=r['outer_kl'])

    def set_schedule(self, stage_overlap=-1, fuse_min_tasks=-1):
        """launch scheduling knobs (results do not depend on them); -1 keeps a value"""
        self._call('promp_set_schedule', int(stage_overlap), int(fuse_min_tasks))

    def set_primal_cache(self, on=True):
        """the second-order pass reads the inner gradient pass's activations back instead of recomputing them
        (True / False; None = the default: on)"""
        self._call('promp_set_primal_cache', -1 if on is None else int(bool(on)))

    def set_reuse_adapt(self, on=True):
        """the first epoch of an optimisation takes the inner pass promp_inner_adapt just ran instead of repeating it
        (bit-identical; default on)"""
        self._call('promp_set_reuse_adapt', int(bool(on)))

    def adapt_passes_skipped(self):
        return int(self.lib.cdll.promp_adapt_passes_skipped(self._h))

    def state_version(self):
        """moves whenever parameters, step sizes, min_std / learn_std or a step's data are replaced (promp_state_version)"""
        return int(self.lib.cdll.promp_state_version(self._h))

    def constraint_hvp_cached_passes(self):
        return int(self.lib.cdll.promp_constraint_hvp_cached_passes(self._h))

    def set_min_std(self, min_std):
        self._call('promp_set_min_std', float(min_std))

    def set_learn_std(self, on=True):
        self._call('promp_set_learn_std', int(bool(on)))

    def adam_step(self, lr):
        self._call('promp_adam_step', float(lr))

    def optimize(self, num_epochs, lr, clip_eps, inner_kl_coeff, inner_kind=INNER_RATIO, outer_kind=OUTER_CLIP):
        eta = _f32(inner_kl_coeff)
        assert eta.shape == (self.K,)
        lb, stats = C.c_float(0), np.empty(self.K + 2, np.float32)
        self._call('promp_optimize', int(num_epochs), float(lr), float(clip_eps), _ptr(eta, C.c_float), int(inner_kind),
                   int(outer_kind), C.byref(lb), _ptr(stats, C.c_float))
        return dict(loss_before=float(lb.value), loss_after=float(stats[0]), inner_kl=stats[1:1 + self.K].copy(),
                    outer_kl=float(stats[1 + self.K]))

    def optimize_begin(self, num_epochs, lr, clip_eps, inner_kl_coeff, inner_kind=INNER_RATIO, outer_kind=OUTER_CLIP):
        """promp_optimize_begin: enqueue the whole optimisation and return at once; optimize_end() collects the statistics
        (the host is free to enqueue the next batch's process_samples / inner_adapt in between)"""
        eta = _f32(inner_kl_coeff)
        assert eta.shape == (self.K,)
        self._call('promp_optimize_begin', int(num_epochs), float(lr), float(clip_eps), _ptr(eta, C.c_float), int(inner_kind),
                   int(outer_kind))

    def optimize_end(self):
        lb, stats = C.c_float(0), np.empty(self.K + 2, np.float32)
        self._call('promp_optimize_end', C.byref(lb), _ptr(stats, C.c_float))
        return dict(loss_before=float(lb.value), loss_after=float(stats[0]), inner_kl=stats[1:1 + self.K].copy(),
                    outer_kl=float(stats[1 + self.K]))

    def eval_loss_grad(self, step, kind, clip_eps=0.0, clip_log_std=False):
        g = np.empty((self.n_tasks, self.n_params), np.float32)
        l, k = np.empty(self.n_tasks, np.float32), np.empty(self.n_tasks, np.float32)
        self._call('promp_eval_loss_grad', int(step), int(kind), float(clip_eps), int(bool(clip_log_std)),
                   _ptr(g, C.c_float), _ptr(l, C.c_float), _ptr(k, C.c_float))
        return g, l, k

    def constraint_hvp(self, v, inner_kind=INNER_LOGLIK, refresh_chain=True):
        """exact Hessian-vector product of the mean outer KL through the adaptation (promp_constraint_hvp)"""
        v = _f32(v)
        assert v.shape == (self.n_params,)
        out = np.empty_like(v)
        self._call('promp_constraint_hvp', int(inner_kind), _ptr(v, C.c_float), int(bool(refresh_chain)), _ptr(out, C.c_float))
        return out

    def cg_solve(self, b, cg_iters=10, reg_coeff=0.0, eps=1e-5, hvp_mode=0, residual_tol=1e-10, inner_kind=INNER_LOGLIK):
        """ConjugateGradientOptimizer's solve on the device (promp_cg_solve): -> (x, x . (H + reg I) x).
        hvp_mode 0 symmetric finite differences of the constraint gradient, 1 one-sided, 2 the exact product"""
        b = _f32(b)
        assert b.shape == (self.n_params,)
        x = np.empty_like(b)
        xhx = C.c_double(0.0)
        self._call('promp_cg_solve', int(inner_kind), _ptr(b, C.c_float), int(cg_iters), float(reg_coeff), float(eps), int(hvp_mode),
                   float(residual_tol), _ptr(x, C.c_float), C.byref(xhx))
        return x, float(xhx.value)

    def eval_hvp(self, step, v, inner_kind=INNER_RATIO, clip_log_std=False, kl_weight=0.0):
        v = _f32(v)
        assert v.shape == (self.n_tasks, self.n_params)
        out = np.empty_like(v)
        self._call('promp_eval_hvp', int(step), int(inner_kind), int(bool(clip_log_std)), float(kl_weight),
                   _ptr(v, C.c_float), _ptr(out, C.c_float))
        return out

    # ---- multi-GPU / measurement ----
    def comm_init(self, rank, nranks, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._call('promp_comm_init', int(rank), int(nranks), C.cast(buf, _P), 128)

    def comm_info(self):
        """what the attached communicator reports (ncclCommCount / ncclCommUserRank; 1 / 0 without one) and this context's GPU"""
        n, r, fo = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        buf = C.create_string_buffer(64)
        self._call('promp_comm_info', C.byref(n), C.byref(r), C.byref(fo), buf, 64)
        return dict(nranks=int(n.value), rank=int(r.value), fixed_order=bool(fo.value), pci_bus_id=buf.value.decode())

    def comm_move_from(self, other):
        """take over `other`'s communicator (a context re-created with more capacity keeps it: no new rendezvous)"""
        self.lib.check(self.lib.cdll.promp_comm_move(self._h, other._h))

    def comm_fixed_order(self, on=True):
        """exchange = ncclAllGather + sum in rank order instead of ncclAllReduce: bitwise-identical replicas by construction"""
        self._call('promp_comm_fixed_order', int(bool(on)))

    def comm_split_path(self, on=True):
        self._call('promp_comm_split_path', int(bool(on)))

    def reduced_get(self):
        out = np.empty(self.n_params + self.K + 2, np.float32)
        self._call('promp_reduced_get', _ptr(out, C.c_float))
        return out

    def reduced_set(self, values):
        v = _f32(values)
        assert v.shape == (self.n_params + self.K + 2,)
        self._call('promp_reduced_set', _ptr(v, C.c_float))

    def allreduce_f64(self, values, op='sum'):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._call('promp_allreduce_f64', _ptr(a, C.c_double), int(a.size), 1 if op == 'max' else 0)
        return a

    def sync(self):
        self._call('promp_sync')
        self._upload_refs.clear()          # every enqueued copy has completed

    def prof_enable(self, on=True):
        self._call('promp_prof_enable', int(bool(on)))

    def prof_read(self, kernel_id):
        ms, n, rows = C.c_double(0), C.c_int64(0), C.c_int64(0)
        self._call('promp_prof_read', int(kernel_id), C.byref(ms), C.byref(n), C.byref(rows))
        return dict(total_ms=ms.value, launches=n.value, rows=rows.value)

    def split_events(self):
        """FP16 split of the fused pass kernels: segments walked a second time since the last call because a cotangent left the format
        at its wave's scale, for k_pass and k_chain_hvp (promp_split_events; the call clears the counters)"""
        out = (C.c_int64 * 2)()
        self._call('promp_split_events', out)
        return dict(pass_segments=out[0], hvp_segments=out[1])

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mhz = C.c_int32(0), C.c_int32(0)
        self._call('promp_device_info', name, 256, C.byref(cus), C.byref(mhz))
        return dict(name=name.value.decode(), n_cus=cus.value, clock_mhz=mhz.value)


def comm_unique_id(lib=None):
    lib = lib or get_library()
    buf = C.create_string_buffer(128)
    lib.check(lib.cdll.promp_comm_unique_id(C.cast(buf, _P), 128))
    return bytes(buf.raw)


def flatten_paths(paths_meta_batch, lib=None):
    """OrderedDict{task -> [path dicts]} (MetaSampler.obtain_samples' return value) -> flat arrays + CSR offsets.
    One concatenation per field over all paths of all tasks, straight into recycled (page-locked, when `lib` is given) host
    buffers: no per-path conversions (at 800 paths they cost more than the copy), no page faults of a fresh allocation.
    Observations / actions / agent_infos as float32 [rows, dim], rewards float64 if the environment's are, else float32."""
    intact = getattr(paths_meta_batch, 'flat_if_intact', None)
    if intact is not None:
        fl = intact()            # samplers.meta_sampler.HostPaths: the sampler built the flat arrays while it collected
        if fl is not None:
            return fl
    plists = list(paths_meta_batch.values())
    flat = [p for plist in plists for p in plist]
    lens = [len(p['rewards']) for p in flat]
    pro = np.zeros(len(flat) + 1, np.int32)
    np.cumsum(lens, out=pro[1:])
    tpo = np.zeros(len(plists) + 1, np.int32)
    np.cumsum([len(plist) for plist in plists], out=tpo[1:])
    rows = int(pro[-1])

    def cat2d(get):
        parts = [get(p) for p in flat]
        first = parts[0]
        if isinstance(first, np.ndarray) and first.ndim == 2:
            out = host_pool.get((rows, first.shape[1]), np.float32, lib)
            try:
                return np.concatenate(parts, out=out, casting='unsafe')
            except (ValueError, TypeError):
                pass                                             # ragged trailing shapes / non-array entries: the general way below
        a = np.concatenate([np.asarray(x).reshape(n, -1) for x, n in zip(parts, lens)])
        return np.ascontiguousarray(a, dtype=np.float32)
    rparts = [p['rewards'] for p in flat]
    if not (isinstance(rparts[0], np.ndarray) and rparts[0].ndim == 1):
        rparts = [np.asarray(r).reshape(-1) for r in rparts]
    is64 = any(r.dtype == np.float64 for r in rparts)
    rew = np.concatenate(rparts, out=host_pool.get((rows,), np.float64 if is64 else np.float32, lib), casting='unsafe')
    out = dict(task_path_offsets=tpo, path_row_offsets=pro, obs=cat2d(lambda p: p['observations']), rew=rew,
               act=None, old_mean=None, old_log_std=None)
    if all('actions' in p and p.get('agent_infos') and 'mean' in p['agent_infos'] for p in flat):
        out.update(act=cat2d(lambda p: p['actions']), old_mean=cat2d(lambda p: p['agent_infos']['mean']),
                   old_log_std=cat2d(lambda p: p['agent_infos']['log_std']))
    return out
