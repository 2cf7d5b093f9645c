"""The fp32-grade f16-split kernels (csrc/gemm3.h: three v_mfma_f32_32x32x16_f16 per fp32 product tile) against float64 products, next to what a plain fp32
computation of the same thing loses: the exact-parity mode's GEMM (`k_gemm3`) and attention (`k_attn3`, strided and ragged-causal) through their test hooks.
Tolerances are stated per test as multiples of the fp32 reference's own error against float64 -- "fp32 grade" means the same error, not a looser one."""
import numpy as np
import pytest
import torch

from etude_amd import _lib

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _gemm3(x, w, b, bound, gelu=False, ln=None):
    lib = _lib.lib()
    dev = _dev()
    xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    M, K = x.shape
    N = w.shape[0]
    yd = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    st = torch.cuda.current_stream(dev).cuda_stream
    g, be = (np.ascontiguousarray(v, np.float32) for v in ln) if ln is not None else (None, None)
    _lib.check(lib.etd_debug_gemm3(xd.data_ptr(), M, K, w.ctypes.data, b.ctypes.data, N, float(bound), int(gelu), yd.data_ptr(),
                                   g.ctypes.data if ln is not None else None, be.ctypes.data if ln is not None else None, st), "etd_debug_gemm3")
    return yd.cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 131, 256), (515, 1536, 512), (2050, 512, 2048), (54, 1536, 512), (216, 512, 2048), (2, 154, 512)])
def test_gemm3_has_the_error_of_an_fp32_product(M, N, K):
    rng = np.random.default_rng(M + N + K)
    x = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-3, 1, (M, 1)))).astype(np.float32)        # rows of very different magnitude
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    y = _gemm3(x, w, b, np.abs(x).max())
    assert np.isfinite(y).all()
    # the yardstick: a k-ordered fp32 multiply-add chain (what v_mfma_f32_32x32x2_f32 and the library's former fp32 kernels compute), on a 96 x 96 corner
    r, c = min(M, 96), min(N, 96)
    acc = np.zeros((r, c), np.float32)
    for kk in range(K):
        acc = acc + x[:r, kk:kk + 1] * w[None, :c, kk]
    chain = acc + b[:c]
    scale = (np.abs(x[:r]).astype(np.float64) @ np.abs(w[:c]).astype(np.float64).T).mean()          # sum |x w|: what rounding errors are relative to
    e3, e32 = np.abs(y[:r, :c] - ref[:r, :c]).max() / scale, np.abs(chain - ref[:r, :c]).max() / scale
    assert e3 <= max(1.5 * e32, 2e-7), (e3, e32)
    # and everywhere: within 1e-6 of sum |x w| + |b| (K = 2048 chains of fp32 additions lose ~5e-7; the result itself is rounded to fp32)
    full = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    assert (np.abs(y - ref) / full).max() <= 1e-6


def test_gemm3_gelu_epilogue_and_a_loose_plane_bound():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((640, 512)).astype(np.float32)
    w = (rng.standard_normal((2048, 512)) * 0.04).astype(np.float32)
    b = rng.standard_normal(2048).astype(np.float32) * 0.1
    u = x.astype(np.float64) @ w.astype(np.float64).T + b
    ref = 0.5 * u * (1.0 + torch.erf(torch.from_numpy(u) / np.sqrt(2.0)).numpy())
    for bound in (np.abs(x).max(), 64.0 * np.abs(x).max()):      # a provable bound is looser than the data's own maximum: nothing may depend on that
        y = _gemm3(x, w, b, bound, gelu=True)
        assert np.abs(y - ref).max() <= 4e-6, (bound, np.abs(y - ref).max())


def test_small_m_kernel_with_fused_layernorm():
    """2 .. 512 rows take k_gemm3_s; its fused LayerNorm against float64 LayerNorm + product"""
    rng = np.random.default_rng(21)
    M, N, K = 100, 2048, 512
    x = (rng.standard_normal((M, K)) * 3.0 + 0.5).astype(np.float32)
    g = rng.uniform(0.5, 1.5, K).astype(np.float32); be = (rng.standard_normal(K) * 0.1).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    x64 = x.astype(np.float64)
    xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + be
    ref = xn @ w.astype(np.float64).T + b
    bound = np.sqrt(K - 1) * np.abs(g).max() + np.abs(be).max()           # the provable bound the library uses (g3_bound_ln), ~8 x the data's own maximum
    y = _gemm3(x, w, b, bound, ln=(g, be))
    xn32 = torch.nn.functional.layer_norm(torch.from_numpy(x), (K,), torch.from_numpy(g), torch.from_numpy(be), 1e-5)
    y32 = (xn32 @ torch.from_numpy(w).T + torch.from_numpy(b)).numpy()
    e3, e32 = np.abs(y - ref).max(), np.abs(y32 - ref).max()
    assert e3 <= max(2.0 * e32, 1e-6), (e3, e32)


def _attn_ref(q, k, v, causal_lens=None):
    """float64 softmax(q k^T / 8) v per (sequence, head); q [n][Sq][H], k / v [n][Sk][H]"""
    n, Sq, H = q.shape
    nh = H // 64
    out = np.zeros_like(q, dtype=np.float64)
    for s in range(n):
        for h in range(nh):
            qq, kk, vv = (a[s, :, h * 64:(h + 1) * 64].astype(np.float64) for a in (q, k, v))
            sc = qq @ kk.T / 8.0
            if causal_lens is not None:
                L = causal_lens[s]
                mask = np.arange(kk.shape[0])[None, :] > np.arange(Sq)[:, None]
                mask |= np.arange(kk.shape[0])[None, :] >= L
                sc = np.where(mask, -np.inf, sc)
            sc -= sc.max(-1, keepdims=True)
            p = np.exp(sc)
            out[s, :, h * 64:(h + 1) * 64] = (p / p.sum(-1, keepdims=True)) @ vv
    return out


@pytest.mark.parametrize("n,nh,Sq,Sk", [(3, 4, 256, 256), (5, 4, 88, 256), (4, 4, 88, 88), (2, 4, 512, 512), (3, 4, 32, 32), (2, 8, 200, 200)])
def test_attn3_strided_against_float64(n, nh, Sq, Sk):
    rng = np.random.default_rng(Sq + Sk + n)
    H = nh * 64
    q = (rng.standard_normal((n, Sq, H)) * 2.0).astype(np.float32)
    k = (rng.standard_normal((n, Sk, H)) * 2.0).astype(np.float32)        # scores with sigma 4: a softmax with real contrast
    v = rng.standard_normal((n, Sk, H)).astype(np.float32)
    ref = _attn_ref(q, k, v)
    dev = _dev()
    qd, kd, vd = (torch.from_numpy(a).to(dev) for a in (q, k, v))
    od = torch.full((n, Sq, H), float("nan"), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().etd_debug_attn3(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), od.data_ptr(), n, nh, Sq, Sk, float(np.abs(q).max()), float(np.abs(k).max()),
                                          float(np.abs(v).max()), 0, None, st), "etd_debug_attn3")
    o = od.cpu().numpy()
    assert np.isfinite(o).all()
    o32 = torch.nn.functional.scaled_dot_product_attention(*(torch.from_numpy(a).view(n, -1, nh, 64).transpose(1, 2) for a in (q, k, v))).transpose(1, 2).reshape(n, Sq, H).numpy()
    e3, e32 = np.abs(o - ref).max(), np.abs(o32 - ref).max()
    assert e3 <= max(3.0 * e32, 2e-6), (e3, e32)


def test_attn3_ragged_causal_reads_kv_cache_rows():
    rng = np.random.default_rng(9)
    n, nh, S = 5, 8, 600
    H = nh * 64
    lens = np.asarray([513, 1, 130, 600, 64], np.int32)
    M = int(lens.sum())
    qrows = (rng.standard_normal((M, H)) * 2.0).astype(np.float32)
    kc = (rng.standard_normal((n, nh, S, 64)) * 2.0).astype(np.float32)      # [slot][head][position][64]: positions past a prompt's length hold garbage the kernel must not use
    vc = rng.standard_normal((n, nh, S, 64)).astype(np.float32)
    dev = _dev()
    qd, kd, vd = (torch.from_numpy(a).to(dev) for a in (qrows, kc, vc))
    od = torch.full((M, H), float("nan"), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().etd_debug_attn3(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), od.data_ptr(), n, nh, S, S, float(np.abs(qrows).max()), float(np.abs(kc).max()),
                                          float(np.abs(vc).max()), 1, lens.ctypes.data, st), "etd_debug_attn3")
    o = od.cpu().numpy()
    assert np.isfinite(o).all()
    row = 0
    for s in range(n):
        L = int(lens[s])
        q = qrows[row:row + L][None]
        k = kc[s].transpose(1, 0, 2).reshape(1, S, H)[:, :L]
        v = vc[s].transpose(1, 0, 2).reshape(1, S, H)[:, :L]
        ref = _attn_ref(q, k, v, causal_lens=[L])[0]
        o32 = torch.nn.functional.scaled_dot_product_attention(*(torch.from_numpy(np.ascontiguousarray(a)).view(1, L, nh, 64).transpose(1, 2) for a in (q, k, v)),
                                                               is_causal=True).transpose(1, 2).reshape(L, H).numpy()     # the yardstick: torch-CPU fp32
        e3, e32 = np.abs(o[row:row + L] - ref).max(), np.abs(o32 - ref).max()
        assert e3 <= max(3.0 * e32, 2e-6), (s, e3, e32)
        row += L
