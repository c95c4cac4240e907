"""Kernel-source verification without a GPU: the HIP kernels + host sequencing of promp_amd/csrc compiled
against the SIMT interpreter (tests/emu) and compared with the oracle at tiny shapes.  These are NOT the
parity tests proper (those are -m gpu in test_gpu_parity.py and run the hipcc-built library on an MI355X);
they exist so that indexing / MFMA fragment-layout / sequencing bugs are caught in the build container."""
import pytest

from tests import devlib, helpers, parity_checks as pc


@pytest.fixture(scope='module')
def lib():
    return devlib.emu_library()


@pytest.mark.parametrize('name', ['default', 'gae095', 'positive', 'ragged', 'clipped_obs', 'zero_base', 'time_base',
                                  'undiscounted', 'float64_obs'])
def test_sample_processing_vs_reference_outputs(lib, name):
    pc.check_sample_processing_golden(lib, name)


def test_sample_processing_long_ragged_paths(lib):
    # T > 64 exercises the chunk carry of the wave scans; O=20 -> 3 FP64-MFMA feature blocks
    pc.check_sample_processing_oracle(lib, 5, M=2, P=3, T=150, O=20, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_loss_grad_h64(lib):
    pc.check_loss_grad(lib, 7, M=2, P=2, T=37, O=20, A=6, hidden=(64, 64))


def test_loss_grad_h32_compact_log_std(lib):
    pc.check_loss_grad(lib, 8, M=2, P=1, T=70, O=3, A=2, hidden=(32, 32), compact_log_std=True)


def test_loss_grad_clipped_log_std(lib):
    pc.check_loss_grad(lib, 9, M=2, P=2, T=20, O=4, A=3, hidden=(32, 32), low_log_std=True)


def test_hvp_h64(lib):
    pc.check_hvp(lib, 10, M=2, P=2, T=37, O=20, A=6, hidden=(64, 64))


def test_hvp_h32_odd_obs(lib):
    pc.check_hvp(lib, 11, M=1, P=2, T=50, O=5, A=3, hidden=(32, 32), ragged=True)


def test_meta_k1_h64(lib):
    pc.check_meta(lib, 12, M=2, P=2, T=40, O=20, A=6, hidden=(64, 64), K=1, ragged=True, epochs=1)


def test_meta_k2_h32(lib):
    pc.check_meta(lib, 13, M=2, P=2, T=33, O=5, A=3, hidden=(32, 32), K=2, epochs=2, compact_log_std=True)


def test_loss_grad_h32_wide_obs(lib):
    # O=31, A=8, H=32: the compact (non hidden_1) part of the gradient is larger than the hidden_1 kernel
    pc.check_loss_grad(lib, 14, M=1, P=1, T=40, O=31, A=8, hidden=(32, 32))


def test_hvp_h32_wide_obs(lib):
    pc.check_hvp(lib, 15, M=1, P=1, T=40, O=31, A=8, hidden=(32, 32))
