"""Key/value logger and snapshot writer with the reference's call surface and file formats
(meta_policy_search/utils/logger.py: configure / log / logkv / logkvs / dumpkvs / getkvs / save_itr_params / get_dir, output
formats 'stdout', 'log', 'json', 'csv', snapshot modes 'all', 'last', 'gap', 'last_gap', 'none').

Files, so that the reference's plotting / reload scripts work on runs of this package:
  progress.csv   one row per dumpkvs(), header = the keys in first-seen order; a key that first appears later extends the
                 header and the earlier rows get empty cells (the reference rewrites the file the same way)
  progress.json  one JSON object per dumpkvs()
  log.txt        the human-readable tables and log() lines
  params.pkl / itr_<n>.pkl   snapshots, written with joblib (compress=3) like the reference's, readable by joblib.load

Design: a `Logger` owns a list of sinks; the module-level functions forward to the current instance (`configure` replaces it).
"""
import csv
import json
import os
import sys
from collections import OrderedDict

SNAPSHOT_MODES = ('all', 'last', 'gap', 'last_gap', 'none')


def _cell(value):
    if hasattr(value, 'item') and getattr(value, 'size', 2) == 1:
        value = value.item()
    return value


class _TableSink(object):
    """aligned key | value tables plus free text, to a stream or a file"""

    def __init__(self, target):
        self._own = isinstance(target, str)
        self._stream = open(target, 'wt') if self._own else target

    def write_row(self, kvs):
        if not kvs:
            return
        shown = [(str(k), ('%-8.6g' % v).rstrip() if isinstance(v, float) else str(v)) for k, v in kvs.items()]
        kw, vw = max(len(k) for k, _ in shown), max(len(v) for _, v in shown)
        rule = '-' * (kw + vw + 7)
        lines = [rule] + ['| %-*s | %-*s |' % (kw, k, vw, v) for k, v in shown] + [rule]
        self._stream.write('\n'.join(lines) + '\n')
        self._stream.flush()

    def write_text(self, text):
        self._stream.write(text + '\n')
        self._stream.flush()

    def close(self):
        if self._own:
            self._stream.close()


class _JsonSink(object):
    def __init__(self, path):
        self._f = open(path, 'wt')

    def write_row(self, kvs):
        self._f.write(json.dumps({k: (float(v) if hasattr(v, 'dtype') else v) for k, v in kvs.items()}) + '\n')
        self._f.flush()

    def write_text(self, text):
        pass

    def close(self):
        self._f.close()


class _CsvSink(object):
    def __init__(self, path):
        self._path, self._keys, self._rows = path, [], 0
        open(path, 'wt').close()

    def write_row(self, kvs):
        fresh = [k for k in kvs if k not in self._keys]
        if fresh:
            old = []
            if self._rows:
                with open(self._path, 'rt', newline='') as f:
                    old = list(csv.reader(f))[1:]
            self._keys += fresh
            with open(self._path, 'wt', newline='') as f:
                w = csv.writer(f)
                w.writerow(self._keys)
                for row in old:
                    w.writerow(row + [''] * (len(self._keys) - len(row)))
        with open(self._path, 'at', newline='') as f:
            csv.writer(f).writerow([kvs.get(k, '') for k in self._keys])
        self._rows += 1

    def write_text(self, text):
        pass

    def close(self):
        pass


def _make_sink(fmt, directory, quiet):
    if fmt == 'stdout':
        return None if quiet else _TableSink(sys.stdout)
    if fmt == 'log':
        return _TableSink(os.path.join(directory, 'log.txt'))
    if fmt == 'json':
        return _JsonSink(os.path.join(directory, 'progress.json'))
    if fmt == 'csv':
        return _CsvSink(os.path.join(directory, 'progress.csv'))
    raise ValueError('Unknown format specified: %s' % (fmt,))


class Logger(object):
    def __init__(self, directory=None, format_strs=None, snapshot_mode='last', snapshot_gap=1, quiet=False):
        if snapshot_mode not in SNAPSHOT_MODES:
            raise NotImplementedError('unknown snapshot mode %r (one of %s)' % (snapshot_mode, ', '.join(SNAPSHOT_MODES)))
        self.dir, self.snapshot_mode, self.snapshot_gap = directory, snapshot_mode, snapshot_gap
        self.kvs = OrderedDict()
        if format_strs is None:
            format_strs = ['stdout', 'log', 'csv'] if directory else ['stdout']
        if directory:
            os.makedirs(directory, exist_ok=True)
        else:
            format_strs = [f for f in format_strs if f == 'stdout']
        self.sinks = [s for s in (_make_sink(f, directory, quiet) for f in format_strs) if s is not None]

    def log(self, *args):
        text = ' '.join(str(a) for a in args)
        for sink in self.sinks:
            sink.write_text(text)

    def dump(self):
        row = OrderedDict((k, _cell(v)) for k, v in self.kvs.items())
        for sink in self.sinks:
            sink.write_row(row)
        self.kvs.clear()
        return row

    def snapshot_file(self, itr):
        """file name the snapshot of iteration `itr` goes to, or None (utils/logger.py:376-396)"""
        mode, on_gap = self.snapshot_mode, itr % self.snapshot_gap == 0
        if not self.dir or mode == 'none':
            return None
        if mode == 'all' or (mode == 'gap' and on_gap):
            return 'itr_%d.pkl' % itr
        if mode == 'last' or (mode == 'last_gap' and on_gap):
            return 'params.pkl'
        return None

    def save_itr_params(self, itr, params):
        name = self.snapshot_file(itr)
        if name is not None:
            import joblib
            joblib.dump(params, os.path.join(self.dir, name), compress=3)

    def close(self):
        for sink in self.sinks:
            sink.close()


_current = Logger()


def configure(dir=None, format_strs=None, snapshot_mode='last', snapshot_gap=1, quiet=False):
    global _current
    _current.close()
    _current = Logger(dir, format_strs, snapshot_mode, snapshot_gap, quiet)


def get_dir():
    return _current.dir


def log(*args):
    _current.log(*args)


def logkv(key, val):
    _current.kvs[key] = val


def logkvs(d):
    for k, v in d.items():
        logkv(k, v)


def getkvs():
    return _current.kvs


def dumpkvs():
    return _current.dump()


def save_itr_params(itr, params):
    _current.save_itr_params(itr, params)


def load_params(path):
    """a snapshot written by save_itr_params (joblib)"""
    import joblib
    return joblib.load(path)
