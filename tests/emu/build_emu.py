"""Build tests/emu/libpromp_emu.so: the kernel sources of promp_amd/csrc compiled by g++ against the
SIMT interpreter in hip_emu.h.  TEST INFRASTRUCTURE ONLY -- never loaded by the promp_amd package."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, 'promp_amd', 'csrc', 'promp_hip.hip')
OUT = os.path.join(HERE, 'libpromp_emu.so')


def build(force=False):
    deps = [SRC, os.path.join(HERE, 'hip_emu.h')] + [os.path.join(ROOT, 'promp_amd', 'csrc', f)
                                                      for f in os.listdir(os.path.join(ROOT, 'promp_amd', 'csrc'))]
    deps.append(os.path.join(ROOT, 'include', 'promp_hip.h'))
    fresh = lambda: os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)
    if not force and fresh():
        return OUT
    # the workers of a parallel test run find the library stale together: one builds (80 s of g++), the others wait for it
    import fcntl
    with open(OUT + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        return _compile()


def _compile():
    tmp = '%s.%d.tmp' % (OUT, os.getpid())
    cmd = ['g++', '-std=c++20', '-O1', '-g', '-DPROMP_EMU', '-fPIC', '-shared', '-x', 'c++', SRC,
           '-I', HERE, '-I', os.path.join(ROOT, 'promp_amd', 'csrc'), '-o', tmp, '-lpthread',
           '-Wall', '-Wno-unknown-pragmas', '-Wno-unused-variable', '-Wno-unused-but-set-variable']
    subprocess.check_call(cmd)
    os.replace(tmp, OUT)                        # atomic: a reader sees the old or the new library, never a partial one
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
