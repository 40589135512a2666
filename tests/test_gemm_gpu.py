"""Kernel unit tests on the GPU: the fp32-MFMA GEMM + fused epilogues against plain PyTorch fp32
math computed on the CPU (tolerances are fp32 summation-order noise)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup():
    import torch
    from vap_realtime_amd import engine
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch, engine.load_library()


def _ref(epi, A, W, bias, gamma, beta, resid):
    import torch
    import torch.nn.functional as F
    acc = A.double() @ W.double().T
    if epi == 0:
        return (acc + (bias.double() if bias is not None else 0)).float(), None
    if epi == 1:
        return F.gelu(acc).float(), None
    if epi == 2:
        return (resid.double() + acc).float(), None
    if epi == 3:
        x = resid.double() + acc
        return x.float(), F.layer_norm(x, (256,), gamma.double(), beta.double(), 1e-5).float()
    if epi == 4:
        v = acc + bias.double()
        mean = v.mean(1, keepdim=True)
        var = v.var(1, keepdim=True)  # unbiased
        return F.relu((v - mean) * torch.rsqrt(var + 1e-5) * gamma.double() + beta.double()).float(), None
    if epi == 5:
        v = acc + bias.double()
        return F.gelu(F.layer_norm(v, (256,), gamma.double(), beta.double(), 1e-5)).float(), None
    raise ValueError


@pytest.mark.parametrize("tile", [32, 64, 128, 0])
@pytest.mark.parametrize("epi,N,K,M", [
    (0, 768, 256, 333), (0, 512, 256, 130), (1, 768, 256, 257), (2, 256, 768, 100), (3, 256, 256, 515),
    (3, 256, 768, 64), (4, 256, 2048, 200), (4, 256, 1024, 31), (5, 256, 1280, 70), (0, 256, 32, 1),
])
def test_gemm_epilogues(epi, N, K, M, tile):
    torch, lib = _setup()
    g = torch.Generator().manual_seed(1234 + epi * 7 + M)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.3
    gamma = 1 + 0.2 * torch.randn(N, generator=g)
    beta = 0.2 * torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    want, want2 = _ref(epi, A, W, bias if epi in (0, 4, 5) else None, gamma, beta, resid)
    d = lambda t: t.cuda().contiguous()
    dA, dW, db, dg, dbe, dr = map(d, (A, W, bias, gamma, beta, resid))
    dC = torch.full((M, N), float("nan"), device="cuda")
    dC2 = torch.full((M, N), float("nan"), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.vapx_gemm(None, M, N, K, p(dA), p(dW), p(dC), epi, p(db) if epi in (0, 4, 5) else None,
                       p(dg), p(dbe), p(dr), p(dC2), tile)
    assert rc == 0, lib.vapx_last_error(None)
    torch.cuda.synchronize()
    tol = 2e-5 * max(1.0, float(want.abs().max()))
    np.testing.assert_allclose(dC.cpu().numpy(), want.numpy(), rtol=1e-5, atol=tol)
    if want2 is not None:
        np.testing.assert_allclose(dC2.cpu().numpy(), want2.numpy(), rtol=1e-5, atol=3e-5)


def test_gemm_transpose_detecting():
    """A = I with an asymmetric W: catches a swapped C/D fragment mapping."""
    torch, lib = _setup()
    K = N = 256
    A = torch.eye(K)
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1000.0
    dA, dW = A.cuda(), W.cuda()
    dC = torch.zeros(K, N, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.vapx_gemm(None, K, N, K, p(dA), p(dW), p(dC), 0, None, None, None, None, None, 0) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dC.cpu().numpy(), W.T.numpy())
