"""GPU parity tests of the individual C-ABI ops against plain float64 references."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

TOL = 2e-5  # f32 accumulation noise on contractions up to K ~ 8k; parity bar is 1e-4


def _dev(a):
    return torch.tensor(a, dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("tA,tB,M,N,K,flags", [
    (0, 1, 80, 1024, 128, 0),          # table0: emb . Wx0^T  (skinny, NT)
    (0, 1, 64, 512, 512, 2),           # heads: tanh(comb . Wlh^T + b) (skinny)
    (0, 0, 64, 512, 128, 0),           # heads bwd: dmu_raw . Wmu (skinny, NN)
    (0, 1, 5120, 1024, 256, 0),        # decoder layer-1 projection (tile 128)
    (0, 1, 5120, 80, 256, 0),          # fc_out (N = 80)
    (0, 0, 5120, 256, 1024, 0),        # dh = dG . Wx (NN)
    (1, 0, 1024, 256, 5120, 1 | 4),    # dWx += dG^T . h (TN, split-K atomics)
    (1, 0, 80, 256, 5120, 1 | 4),      # dWout
    (1, 0, 128, 512, 64, 1),           # dWmu (K = B)
    (0, 1, 80, 1024, 129, 0),          # odd K, unaligned ld (scalar-load path)
    (0, 1, 37, 53, 19, 0),             # ragged everything
    (1, 1, 70, 90, 33, 0),             # TT
    (0, 1, 4, 256, 128, 0),            # tiny batch skinny
])
def test_gemm(tA, tB, M, N, K, flags):
    from arcvae_hip import _lib
    rs = np.random.RandomState(M * 7 + N * 3 + K)
    A = rs.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    Bm = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    C0 = rs.standard_normal((M, N)).astype(np.float32)
    use_bias = not (flags & 4)
    dA, dB, dC, db = _dev(A), _dev(Bm), _dev(C0), _dev(bias)
    _lib.gemm(bool(tA), bool(tB), M, N, K, dA, A.shape[1], dB, Bm.shape[1], dC, N,
              db if use_bias else None, flags)
    torch.cuda.synchronize()
    ref = (A.T if tA else A).astype(np.float64) @ (Bm.T if tB else Bm).astype(np.float64)
    if use_bias:
        ref = ref + bias
    if flags & 1:
        ref = ref + C0
    if flags & 2:
        ref = np.tanh(ref)
    assert rel_err(dC.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("tA,tB,M,N,K,flags", [(0, 1, 640, 384, 256, 0), (0, 0, 640, 384, 256, 2), (1, 0, 512, 256, 1000, 1 | 4),
                                                (1, 1, 300, 130, 72, 1), (0, 1, 16384, 2048, 512, 0), (0, 0, 16384, 512, 2048, 0),
                                                (1, 0, 2048, 512, 16384, 4), (0, 1, 1000, 80, 512, 0), (0, 0, 1000, 512, 80, 0),
                                                (0, 1, 260, 100, 129, 0)])
def test_bf16_mode_gemm_rounds_operands_only(tA, tB, M, N, K, flags):
    """ARCVAE_GEMM_BF16 (throughput mode, csrc/gemm.hip gemm_bf16_tile_kernel / the one-product form of the split TN
    kernel): the result must equal the fp64 product of the bf16-ROUNDED operands up to f32 accumulation error -- i.e.
    rounding the operands is the only approximation -- and sit within 2^-8 * sum|a||b| of the unrounded product.  The last
    case (K = 129: rows not 16-byte aligned) has to fall back to the f32 kernels, which the tighter bound shows."""
    from arcvae_hip import _lib
    rs = np.random.RandomState(M + 3 * N + 7 * K)
    A = rs.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    Bm = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    C0 = rs.standard_normal((M, N)).astype(np.float32)
    use_bias = not (flags & 4)
    dA, dB, dC, db = _dev(A), _dev(Bm), _dev(C0), _dev(bias)
    _lib.gemm(bool(tA), bool(tB), M, N, K, dA, A.shape[1], dB, Bm.shape[1], dC, N, db if use_bias else None,
              flags | _lib.GEMM_BF16 | _lib.GEMM_NO_SKINNY)
    torch.cuda.synchronize()
    got = dC.cpu().numpy().astype(np.float64)
    rnd = lambda x: torch.from_numpy(x).to(torch.bfloat16).to(torch.float64).numpy()
    opA, opB = (lambda X: X.T if tA else X), (lambda X: X.T if tB else X)
    def finish(ref):
        if use_bias:
            ref = ref + bias
        if flags & 1:
            ref = ref + C0
        if flags & 2:
            ref = np.tanh(ref)
        return ref
    mag = np.abs(opA(A).astype(np.float64)) @ np.abs(opB(Bm).astype(np.float64)) + 1.0
    exact = finish(opA(A).astype(np.float64) @ opB(Bm).astype(np.float64))
    rounded = finish(opA(rnd(A)) @ opB(rnd(Bm)))
    if K % 4:
        assert (np.abs(got - exact) / mag).max() < 1e-6            # f32 fallback
        return
    assert (np.abs(got - rounded) / mag).max() < 2e-6, (np.abs(got - rounded) / mag).max()
    assert (np.abs(got - exact) / mag).max() < 2.0 ** -8
    assert (np.abs(got - exact) / mag).max() > 1e-5               # the bf16 path did run


@pytest.mark.parametrize("tA,tB,M,N,K,flags", [(0, 1, 5120, 1024, 256, 0), (0, 0, 5120, 256, 1024, 0), (0, 1, 640, 80, 256, 0),
                                                (1, 0, 512, 256, 1000, 1), (1, 1, 300, 130, 72, 2), (0, 0, 1000, 512, 80, 0),
                                                (0, 1, 16384, 2048, 256, 0), (0, 0, 16500, 2048, 128, 1)])
def test_three_piece_gemm_has_f32_accuracy(tA, tB, M, N, K, flags):
    """ARCVAE_GEMM_SPLIT3 (gemm_bf16_tile_kernel with three bf16 pieces per operand, six products): any layout, the accuracy
    class of the exact-f32 MFMA kernel on the same data: worst element error relative to sum|a||b| within 2x the f32 kernel's
    (plus a floor), and the same standing against the 1e-4 / 1e-6 element-wise parity criterion."""
    from arcvae_hip import _lib
    from helpers import elem_err
    rs = np.random.RandomState(2 * M + N + K)
    A = (rs.standard_normal((K, M) if tA else (M, K)) * np.exp(rs.uniform(-2, 2, size=(1, M) if tA else (M, 1)))).astype(np.float32)
    Bm = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    C0 = rs.standard_normal((M, N)).astype(np.float32)
    opA, opB = (A.T if tA else A).astype(np.float64), (Bm.T if tB else Bm).astype(np.float64)
    ref = opA @ opB + bias
    if flags & 1:
        ref = ref + C0
    if flags & 2:
        ref = np.tanh(ref)
    mag = np.abs(opA) @ np.abs(opB) + 1.0
    dA, dB, db = _dev(A), _dev(Bm), _dev(bias)
    err, frac = {}, {}
    for name, extra in (("split3", _lib.GEMM_SPLIT3), ("f32", 0)):
        dC = _dev(C0)
        _lib.gemm(bool(tA), bool(tB), M, N, K, dA, A.shape[1], dB, Bm.shape[1], dC, N, db, flags | extra | _lib.GEMM_NO_SKINNY)
        torch.cuda.synchronize()
        got = dC.cpu().numpy().astype(np.float64)
        err[name] = (np.abs(got - ref) / mag).max()
        frac[name] = elem_err(got, ref, 1e-4, 1e-6)[0]           # worst element as a fraction of the parity bound
    assert err["split3"] <= max(2.0 * err["f32"], 3e-7), err
    assert frac["split3"] <= max(1.5 * frac["f32"], 0.3), frac   # (K = 1024 with this dynamic range: both sit near 0.2)


def test_gemm_strided_c():
    """C with ldc > N and B with ldb > K (the decoder's Wx0[:, :E] sub-block)."""
    from arcvae_hip import _lib
    rs = np.random.RandomState(3)
    M, N, K, ld = 256, 128, 80, 129
    A = rs.standard_normal((K, M)).astype(np.float32)   # stored [K,M] (transA)
    Bm = rs.standard_normal((K, N)).astype(np.float32)
    C0 = rs.standard_normal((M, ld)).astype(np.float32)
    dA, dB, dC = _dev(A), _dev(Bm), _dev(C0)
    _lib.gemm(True, False, M, N, K, dA, M, dB, N, dC, ld, None, 1)
    torch.cuda.synchronize()
    ref = C0.astype(np.float64).copy()
    ref[:, :N] += A.T.astype(np.float64) @ Bm.astype(np.float64)
    assert rel_err(dC.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("M,N,K,ldc_pad", [(1024, 256, 3072, 0), (80, 1024, 2000, 0), (256, 64, 100, 3), (36, 72, 531, 0),
                                           (512, 192, 16, 0), (128, 40, 4097, 5)])
def test_split_bf16_tn_gemm_has_f32_accuracy(M, N, K, ldc_pad, monkeypatch):
    """C += A^T . B on the split-bf16 kernel (three bf16 pieces per operand, six products, f32 accumulate:
    csrc/gemm.hip gemm_split_tn_group_kernel) against fp64, next to the exact-f32 MFMA kernel on the same data: the
    split path must sit in the same accuracy class (its worst element error within 2x the f32 kernel's, plus a
    floor), and pass the 1e-4 / 1e-6 element-wise parity criterion by orders of magnitude."""
    from arcvae_hip import _lib
    from helpers import elem_err
    rs = np.random.RandomState(M + N + K)
    A = (rs.standard_normal((K, M)) * np.exp(rs.uniform(-3, 3, size=(K, 1)))).astype(np.float32)   # wide dynamic range
    Bm = rs.standard_normal((K, N)).astype(np.float32)
    ld = N + ldc_pad
    C0 = rs.standard_normal((M, ld)).astype(np.float32)
    ref = C0.astype(np.float64).copy()
    ref[:, :N] += A.T.astype(np.float64) @ Bm.astype(np.float64)
    mag = np.abs(A.T.astype(np.float64)) @ np.abs(Bm.astype(np.float64))       # sum |a||b| per element
    dA, dB = _dev(A), _dev(Bm)
    out = {}
    for name, flags in (("split", _lib.GEMM_ACCUMULATE | _lib.GEMM_SPLITK), ("f32", _lib.GEMM_ACCUMULATE)):
        dC = _dev(C0)
        _lib.gemm(True, False, M, N, K, dA, M, dB, N, dC, ld, None, flags)
        torch.cuda.synchronize()
        got = dC.cpu().numpy().astype(np.float64)
        if ldc_pad:
            assert np.array_equal(got[:, N:], C0[:, N:].astype(np.float64))   # padding columns untouched
        out[name] = np.abs(got[:, :N] - ref[:, :N]) / (mag + 1e-30)
    assert out["split"].max() <= max(2.0 * out["f32"].max(), 2e-7), (out["split"].max(), out["f32"].max())
    dC = _dev(C0)
    _lib.gemm(True, False, M, N, K, dA, M, dB, N, dC, ld, None, _lib.GEMM_ACCUMULATE | _lib.GEMM_SPLITK)
    torch.cuda.synchronize()
    assert elem_err(dC.cpu().numpy()[:, :N], ref[:, :N], 1e-4, 1e-6)[0] < 0.2


def test_adam_matches_mlx_formula():
    from arcvae_hip import _lib
    from arcvae_hip._lib import call, ptr, stream_ptr
    import ctypes as C
    rs = np.random.RandomState(0)
    n = 1000 + 3
    p = rs.standard_normal(n).astype(np.float32)
    g = (rs.standard_normal(n) * 1e-2).astype(np.float32)
    g[::7] = 0.0  # dead parameters must stay bit-identical (Q7)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    dp, dg, dm, dv = _dev(p), _dev(g), _dev(m), _dev(v)
    for _ in range(3):
        call("arcvae_adam_update", ptr(dp), ptr(dg), ptr(dm), ptr(dv), C.c_long(n), 2e-4, 0.9, 0.999, 1e-8,
             None, None, stream_ptr())
    torch.cuda.synchronize()
    pp, mm, vv = p.copy(), m.copy(), v.copy()
    f = np.float32
    for _ in range(3):
        mm = f(0.9) * mm + f(1 - 0.9) * g
        vv = f(0.999) * vv + f(1 - 0.999) * np.square(g)
        pp = pp - f(2e-4) * mm / (np.sqrt(vv) + f(1e-8))
    out = dp.cpu().numpy()
    assert np.array_equal(out[::7], p[::7])
    assert rel_err(out, pp) < 1e-6
    assert rel_err(dm.cpu().numpy(), mm) < 1e-6
    # step-1 identity from SURVEY section 4: update = lr*0.1g / (sqrt(0.001) |g| + 1e-8)


def test_adam_skips_the_update_when_a_guard_word_is_set():
    """An expired gate / a persistent sweep that gave up leaves a non-zero device word: the Adam kernel must then leave
    parameters and state untouched (gradients formed after a lost stream order destroyed the weights)."""
    from arcvae_hip._lib import call, ptr, stream_ptr
    import ctypes as C
    rs = np.random.RandomState(2)
    n = 4096 + 5
    p, g = rs.standard_normal(n).astype(np.float32), rs.standard_normal(n).astype(np.float32)
    dp, dg, dm, dv = _dev(p), _dev(g), _dev(np.zeros(n, np.float32)), _dev(np.zeros(n, np.float32))
    words = torch.zeros(64, dtype=torch.int32, device="cuda")
    wa, wb = C.c_void_p(words.data_ptr()), C.c_void_p(words.data_ptr() + 128)
    for tripped in ((1, 0), (0, 3)):
        words[0], words[32] = tripped
        call("arcvae_adam_update", ptr(dp), ptr(dg), ptr(dm), ptr(dv), C.c_long(n), 2e-4, 0.9, 0.999, 1e-8, wa, wb,
             stream_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(dp.cpu().numpy(), p) and float(dm.abs().max()) == 0.0 and float(dv.abs().max()) == 0.0
    words.zero_()
    call("arcvae_adam_update", ptr(dp), ptr(dg), ptr(dm), ptr(dv), C.c_long(n), 2e-4, 0.9, 0.999, 1e-8, wa, wb,
         stream_ptr())
    torch.cuda.synchronize()
    assert not np.array_equal(dp.cpu().numpy(), p) and float(dm.abs().max()) > 0.0


def test_segsum_and_colsum():
    from arcvae_hip._lib import call, ptr, stream_ptr
    rs = np.random.RandomState(1)
    R, S, Cc = 1000, 80, 200
    X = rs.standard_normal((R, Cc)).astype(np.float32)
    seg = rs.randint(0, S, size=R).astype(np.int32)
    dX = _dev(X)
    dseg = torch.tensor(seg, dtype=torch.int32, device="cuda")
    out = torch.zeros(S, Cc, device="cuda")
    call("arcvae_segsum_rows_accum", ptr(dX), ptr(dseg), R, S, Cc, ptr(out), stream_ptr())
    cs = torch.zeros(Cc, device="cuda")
    call("arcvae_colsum_accum", ptr(dX), R, Cc, Cc, ptr(cs), 1.0, stream_ptr())
    torch.cuda.synchronize()
    ref = np.zeros((S, Cc))
    np.add.at(ref, seg, X.astype(np.float64))
    assert rel_err(out.cpu().numpy(), ref) < TOL
    assert rel_err(cs.cpu().numpy(), X.astype(np.float64).sum(0)) < TOL


@pytest.mark.parametrize("V,E,C,G", [(80, 128, 0, 1024), (80, 128, 1, 1024), (95, 20, 2, 256), (33, 130, 3, 512), (127, 18, 0, 256)])
def test_table_finalize(V, E, C, G):
    """arcvae_table_finalize: dEmb += dT . Wx0[:, :E];  dWx0[:, :E] += dT^T . emb (row stride E + C);  db0 += colsum(dT)
    -- the backward of nn.Embedding + the x . Wx^T term of the layer-0 nn.LSTM (models/encoder.py:93,98,
    models/decoder.py:154-166) through the [V,4H] token table.  '+=' semantics, columns E..E+C-1 untouched."""
    from arcvae_hip import _lib
    rs = np.random.RandomState(V + E)
    dT = rs.standard_normal((V, G)); Wx0 = rs.standard_normal((G, E + C)); emb = rs.standard_normal((V, E))
    dEmb0 = rs.standard_normal((V, E)); dWx0_0 = rs.standard_normal((G, E + C)); db0_0 = rs.standard_normal(G)
    d = {k: _dev(v) for k, v in dict(dT=dT, Wx0=Wx0, emb=emb, dEmb=dEmb0, dWx0=dWx0_0, db0=db0_0).items()}
    _lib.call("arcvae_table_finalize", _lib.ptr(d["dT"]), _lib.ptr(d["Wx0"]), E + C, _lib.ptr(d["emb"]),
              _lib.ptr(d["dEmb"]), _lib.ptr(d["dWx0"]), _lib.ptr(d["db0"]), V, E, G, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rel_err(d["dEmb"].cpu().numpy(), dEmb0 + dT @ Wx0[:, :E]) < TOL
    want = dWx0_0.copy()
    want[:, :E] += dT.T @ emb
    got = d["dWx0"].cpu().numpy()
    assert rel_err(got, want) < TOL
    if C:
        assert np.array_equal(got[:, E:], dWx0_0[:, E:].astype(np.float32))
    assert rel_err(d["db0"].cpu().numpy(), db0_0 + dT.sum(0)) < TOL
    with pytest.raises(_lib.ArcvaeHipError):
        _lib.call("arcvae_table_finalize", _lib.ptr(d["dT"]), _lib.ptr(d["Wx0"]), E - 1, _lib.ptr(d["emb"]),
                  _lib.ptr(d["dEmb"]), _lib.ptr(d["dWx0"]), _lib.ptr(d["db0"]), V, E, G, _lib.stream_ptr())


def test_copy_buffers_one_launch():
    """arcvae_copy_buffers: several small device-to-device copies in one launch (a step's inputs), every size and
    alignment: 16-byte vector path, byte tails, unaligned views, an empty buffer."""
    import ctypes as C
    from arcvae_hip import _lib
    rs = np.random.RandomState(5)
    sizes = [64 * 128 * 4, 256, 64 * 128 * 4 + 4, 129, 0, 7, 4096 + 3]
    srcs = [torch.tensor(rs.randint(0, 256, size=n + 32).astype(np.uint8), device="cuda") for n in sizes]
    dsts = [torch.full((n + 32,), 255, dtype=torch.uint8, device="cuda") for n in sizes]
    offs = [0, 0, 4, 1, 0, 3, 16]                                  # some views start off the 16-byte grid
    sv = [s[o:o + n] for s, o, n in zip(srcs, offs, sizes)]
    dv = [d[o:o + n] for d, o, n in zip(dsts, offs, sizes)]
    keep = [(a, b) for a, b in zip(sv, dv) if a.numel() > 0]       # a null pointer (empty view) is an argument error
    src, _k1 = _lib.ptr_array([a for a, _ in keep])
    dst, _k2 = _lib.ptr_array([b for _, b in keep])
    nb = (C.c_long * len(keep))(*[a.numel() for a, _ in keep])
    _lib.call("arcvae_copy_buffers", src, dst, nb, len(keep), _lib.stream_ptr())
    torch.cuda.synchronize()
    for s, d, o, n in zip(srcs, dsts, offs, sizes):
        got = d.cpu().numpy()
        assert np.array_equal(got[o:o + n], s.cpu().numpy()[o:o + n])
        assert (got[:o] == 255).all() and (got[o + n:] == 255).all()      # nothing outside the range was touched


def test_plane_gemm_with_pointers_whose_low_half_has_the_top_bit_set():
    """The round-3 memory fault, pinned (DESIGN_HISTORY round-3 log): wgrad_planes_kernel moves its problem's pointers
    into scalar registers half by half with v_readfirstlane, whose builtin returns a SIGNED int -- assembled without an unsigned
    cast, a low half >= 0x80000000 sign-extended over the high half (fault address 0xffffd7560000; the earlier form that
    broadcast only the low half faulted at 0x35a54000 ...).  Whether a run hit it depended on where the allocator had put the
    buffers, so the plane tests passed either way.  Here every pointer the kernel broadcasts -- both operand-plane buffers, the
    output and the bias-sum rider -- is carved out of a large allocation at addresses whose low 32 bits have the top bit set, and
    the result is checked against a float64 product."""
    import ctypes as C
    from arcvae_hip import _lib
    lib = _lib.load()
    B, H, L, T, V, E = 32, 64, 2, 4, 20, 16
    G = 4 * H
    big = torch.empty(6 * 2 ** 30, dtype=torch.uint8, device="cuda")
    base = big.data_ptr()
    off = (0x80000000 - (base & 0xffffffff)) % 2 ** 32          # first byte whose low half is exactly 0x80000000
    assert off + 2 ** 30 <= big.numel()

    def carve(nfloats):
        nonlocal off
        t = big[off:off + 4 * nfloats].view(torch.float32)
        assert (t.data_ptr() & 0xffffffff) >= 0x80000000 and t.data_ptr() % 256 == 0
        off += (4 * nfloats + 255) // 256 * 256
        return t

    rs = np.random.RandomState(11)
    hseq = _dev(rs.standard_normal((L, T, B, H)))
    dG = _dev(rs.standard_normal((L, T, B, G)))
    hpl, gpl = carve(L * T * B * H * 3 // 2), carve(L * T * B * G * 3 // 2)
    dWh = [carve(G * H).zero_() for _ in range(L)]
    dWx = [torch.zeros(G, E, device="cuda"), carve(G * H).zero_()]
    dbias = [torch.zeros(G, device="cuda"), carve(G).zero_()]
    x_tb = torch.zeros(T, B, dtype=torch.int32, device="cuda")
    emb, wx0 = _dev(rs.standard_normal((V, E))), _dev(rs.standard_normal((G, E)))
    dtab, onehot, demb = torch.zeros(V, G, device="cuda"), torch.zeros(T * B, V, device="cuda"), torch.zeros(V, E, device="cuda")
    pwx, _a = _lib.ptr_array(dWx); pwh, _b = _lib.ptr_array(dWh); pbs, _c = _lib.ptr_array(dbias)
    caps = (C.c_long * 4)(hpl.numel(), gpl.numel(), 0, 0)
    # parts: per-layer GEMMs | exact-f32 table path (unused) | planes | "the rings are scratch: split dG / h here first"
    rc = lib.arcvae_enc_lstm_wgrad(_lib.ptr(x_tb), _lib.ptr(emb), _lib.ptr(wx0), _lib.ptr(hseq), _lib.ptr(dG), _lib.ptr(dtab),
                                   _lib.ptr(onehot), _lib.ptr(demb), pwx, pwh, pbs, B, T, V, E, H, L, 0, T, 0, 0,
                                   1 | 16 | 2048 | 4096, _lib.ptr(hpl), _lib.ptr(gpl), caps, _lib.stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    g64, h64 = dG.double().cpu().numpy(), hseq.double().cpu().numpy()
    for l in range(L):                                            # dWh_l = sum_t dG_l[t]^T h_l[t-1]
        assert rel_err(dWh[l].cpu().numpy().reshape(G, H), sum(g64[l, t].T @ h64[l, t - 1] for t in range(1, T))) < TOL
    assert rel_err(dWx[1].cpu().numpy().reshape(G, H), sum(g64[1, t].T @ h64[0, t] for t in range(T))) < TOL
    assert rel_err(dbias[1].cpu().numpy(), g64[1].sum((0, 1))) < TOL   # the bias rider's pointer too
    del big
