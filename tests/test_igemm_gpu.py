"""GPU tests of the MFMA building blocks (csrc/selftest.hip, csrc/igemm.hip)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mfma_fragment_layout():
    from bsed_amd import _lib as L
    rng = np.random.default_rng(0)
    K = 64
    A = rng.standard_normal((32, K)).astype(np.float32)
    B = rng.standard_normal((K, 32)).astype(np.float32)  # asymmetric on purpose
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    c = torch.zeros((32, 32), device="cuda")
    L.call("bsed_selftest_mfma", L.ptr(a), L.ptr(b), L.ptr(c), L.c_int(K), L.stream())
    ref = A.astype(np.float64) @ B.astype(np.float64)
    np.testing.assert_allclose(c.cpu().numpy(), ref, atol=1e-4)
