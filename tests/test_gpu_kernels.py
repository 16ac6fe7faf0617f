"""Per-kernel parity on a real MI355X: every HIP kernel, called through the C ABI,
against a plain PyTorch fp32 reference of the same op computed on the CPU from the
SAME (already rounded) operands, so the tolerances only have to cover fp32
accumulation order and the final rounding -- a wrong fragment layout, swizzle or
index fails by orders of magnitude."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = ["fp16", "bf16"]


def _td(dtype):
    return torch.float16 if dtype == "fp16" else torch.bfloat16


def _eps(dtype):
    return 1e-3 if dtype == "fp16" else 8e-3


@pytest.fixture(scope="module")
def K():
    from afx import kernels
    return kernels


def _close(got, want, rtol, atol):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (max |ref| {want.abs().max().item():.3e})"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K_", [(300, 256, 192), (77, 144, 192), (128, 64, 64), (1000, 512, 1536), (257, 432, 192)])
def test_gemm_epilogues(K, dtype, M, N, K_):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = (torch.randn(M, K_, generator=g)).to(_td(dtype))
    W = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).to(_td(dtype))
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = A.float() @ W.float().t()
    of, oh = K.gemm(dtype, A.cuda(), W.cuda(), out_f=True, out_h=True)
    _close(of, ref, 1e-4, 1e-4)
    _close(oh, ref, _eps(dtype) * 4, 1e-3)
    of, _ = K.gemm(dtype, A.cuda(), W.cuda(), bias=bias.cuda(), act="gelu", alpha=0.5, resid=resid.cuda())
    _close(of, resid + 0.5 * F.gelu(ref + bias), 1e-4, 1e-4)
    of, _ = K.gemm(dtype, A.cuda(), W.cuda(), bias=bias.cuda(), act="swish")
    _close(of, F.silu(ref + bias), 1e-4, 1e-4)


@pytest.mark.parametrize("M,N,K_", [(300, 256, 192), (77, 144, 192), (1000, 512, 1536), (3184, 1024, 1024), (12736, 3072, 1024), (515, 64, 128)])
def test_split_precision_gemm_is_fp32_accurate(K, M, N, K_):
    """dtype "fp16x3": x.w ~ xh.wh + xl.wh + xh.wl on the fp16 matrix pipe (fp16 hi / lo pairs of both operands, power-of-two
    row scales on the weights, x 16 on the activations) -- ONE launch whose K walks the three segments.  Against an fp64
    reference of the FP32 operands: the error must be at fp32 level (a dropped segment, a wrong plane or a wrong scale is
    off by 1e-3 .. 1), on every tile instance the dispatcher can pick for these shapes, with operand magnitudes from 1e-3
    to 1e+2 in one matrix (the lo parts of small entries fall into fp16's subnormals: bounded absolute error)."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(M + 3 * N + K_)
    A = torch.randn(M, K_, generator=g)
    A[:, : K_ // 4] *= 30.0      # outlier channels (transformer activations have them)
    A[:, K_ // 2:] *= 1e-2       # and small ones
    W = torch.randn(N, K_, generator=g) / math.sqrt(K_)
    W[: N // 3] *= 1e-2          # rows of very different scale: each gets its own power of two
    W[N // 3: N // 2] *= 50.0
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref64 = A.double() @ W.double().t()
    scale = (A.double().abs() @ W.double().abs().t())  # the magnitude the rounding errors are relative to
    for tile in (-1, 0, 3, 5):
        try:
            check(lib().afx_debug_set(b"gemm_tile", tile))
            of, oh = K.gemm("fp16x3", A.cuda(), W.cuda(), out_f=True, out_h=True)
        finally:
            check(lib().afx_debug_set(b"gemm_tile", -1))
        assert oh.dtype == torch.float32 and torch.equal(of, oh)
        rel = ((of.cpu().double() - ref64).abs() / scale).max().item()
        assert rel < 3e-6, f"tile {tile}: max error {rel:.2e} of sum |a||w|"  # fp32 GEMM: ~1e-7; fp16 operands: ~5e-4
    of, _ = K.gemm("fp16x3", A.cuda(), W.cuda(), bias=bias.cuda(), act="gelu", alpha=0.5, resid=resid.cuda())
    want = resid.double() + 0.5 * F.gelu(ref64 + bias.double())
    assert ((of.cpu().double() - want).abs() / (1.0 + scale)).max().item() < 3e-6
    # the true-fp32 matrix instruction on the same operands is no closer to the fp64 result than a few times this
    f32, _ = K.gemm("fp32", A.cuda(), W.cuda())
    print(f"M {M} N {N} K {K_}: split precision {rel:.2e}, fp32 MFMA {((f32.cpu().double() - ref64).abs() / scale).max().item():.2e} (of sum |a||w|)")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [0, 1, 3])
def test_gemm_tile_variants_agree_with_reference(K, dtype, tile):
    """Every tile instance (128x128 / 4 waves, 256x256 / 8 waves 2-stage, and the 8-phase
    256x256 kernel = tile 3) and every workgroup->tile mapping must give the same
    numbers; M, N tails included."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(11)
    M, N, K_ = 1000, 768, 1536
    A = torch.randn(M, K_, generator=g).to(_td(dtype))
    W = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).to(_td(dtype))
    bias = torch.randn(N, generator=g)
    ref = F.gelu(A.float() @ W.float().t() + bias)
    try:
        check(lib().afx_debug_set(b"gemm_tile", tile))
        for mode in (0, 1, 2):
            check(lib().afx_debug_set(b"gemm_map", mode))
            of, _ = K.gemm(dtype, A.cuda(), W.cuda(), bias=bias.cuda(), act="gelu")
            _close(of, ref, 1e-4, 1e-4)
    finally:
        check(lib().afx_debug_set(b"gemm_tile", -1))
        check(lib().afx_debug_set(b"gemm_map", -1))
        check(lib().afx_debug_set(b"gemm_deep", -1))


@pytest.mark.parametrize("M,N,K_", [(1000, 768, 64), (515, 512, 192), (300, 256, 128), (2048, 1024, 4096), (12736, 1024, 1024)])
def test_gemm_8phase_kernel_is_bit_identical_to_the_2stage_kernel(K, M, N, K_):
    """The 8-phase 256x256 kernel keeps DMA in flight across barriers (counted vmcnt, two wave
    rows one barrier apart): a misplaced wait shows up as rare stale tiles, so beyond the
    reference check every result is compared BIT FOR BIT with the simple 2-stage kernel (same
    k order per output element) over repeated launches, at 1, 2, 3 and many K-tiles."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(K_ + M)
    A = torch.randn(M, K_, generator=g).half().cuda()
    W = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    try:
        check(lib().afx_debug_set(b"gemm_tile", 0))
        want, _ = K.gemm("fp16", A, W, bias=bias, act="gelu")
        _close(want, F.gelu(A.float().cpu() @ W.float().cpu().t() + bias.cpu()), 1e-4, 2e-4)
        check(lib().afx_debug_set(b"gemm_tile", 3))  # same k order per element as the 2-stage kernel
        for _ in range(12):
            got, _ = K.gemm("fp16", A, W, bias=bias, act="gelu")
            assert torch.equal(got, want)
    finally:
        check(lib().afx_debug_set(b"gemm_tile", -1))


@pytest.mark.parametrize("dtype", ["fp16", "fp16x3"])
@pytest.mark.parametrize("K_", [1024, 4096])
@pytest.mark.parametrize("M", [199, 1592, 3184])
def test_gemm_deep_tile_is_bit_identical_to_the_2stage_tile(K, dtype, M, K_):
    """gemm_deep_kernel (the 128x64 tile with three K-tile buffers: TWO tiles of LDS-DMA in flight across barriers, a counted
    vmcnt, both k-steps' fragments requested before the first MFMA) serves the teacher's N = 1024 products -- the hazard
    class of the 8-phase kernel above, so the same test: the dispatcher's choice at these shapes (the deep tile) against the
    two-buffer 128x64 tile, bit for bit, over repeated launches, fp32 and operand-type outputs, with residual; fp16 and the
    split-precision walk.  The per-engine switch of the same choice ("gemm_small_deep") is exercised in test_gpu_models."""
    from afx._lib import check, lib
    N = 1024
    g = torch.Generator().manual_seed(K_ + M)
    A = torch.randn(M, K_, generator=g)
    W = torch.randn(N, K_, generator=g) / math.sqrt(K_)
    if dtype == "fp16":
        A, W = A.half(), W.half()
    A, W = A.cuda(), W.cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda()
    ref = A.float().cpu().double() @ W.float().cpu().double().t() + bias.cpu().double() + R.cpu().double()
    try:
        check(lib().afx_debug_set(b"gemm_tile", 5))  # the two-buffer 128x64 tile
        want_f, want_h = K.gemm(dtype, A, W, bias=bias, resid=R, out_f=True, out_h=True)
        assert ((want_f.cpu().double() - ref).abs() / (1.0 + ref.abs())).max().item() < (2e-4 if dtype == "fp16" else 2e-5)
        for forced in (8, -1):  # the deep tile forced, then the dispatcher's own choice at this shape (the deep tile)
            check(lib().afx_debug_set(b"gemm_tile", forced))
            for _ in range(12):
                got_f, got_h = K.gemm(dtype, A, W, bias=bias, resid=R, out_f=True, out_h=True)
                assert torch.equal(got_f, want_f) and torch.equal(got_h, want_h), forced
    finally:
        check(lib().afx_debug_set(b"gemm_tile", -1))


@pytest.mark.parametrize("M,N,K_", [(12736, 1024, 1024), (3184, 3072, 1024), (3184, 4096, 1024), (1000, 768, 64), (447, 512, 192)])
def test_gemm_8phase_short_tiles_are_bit_identical(K, M, N, K_):
    """Tile heights 160 / 192 / 224 / 256 of the 8-phase kernel (the launcher fits the height to one round of the
    CUs): same LDS layout, fewer live fragments per wave row -- every height, forced and fitted, must reproduce the
    2-stage kernel bit for bit (fp32 and fp16 outputs, residual, ragged last tile), over repeated launches."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(K_ + M + N)
    A = torch.randn(M, K_, generator=g).half().cuda()
    W = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda()
    try:
        check(lib().afx_debug_set(b"gemm_tile", 0))
        want_f, want_h = K.gemm("fp16", A, W, bias=bias, resid=R, out_f=True, out_h=True)
        _close(want_f, A.float().cpu() @ W.float().cpu().t() + bias.cpu() + R.cpu(), 1e-4, 2e-4)
        check(lib().afx_debug_set(b"gemm_tile", 3))
        for ph in (0, 2):  # the shipped form (two-phase K-tile over a three-buffer A ring), then the two-buffer form (A/B knob)
            check(lib().afx_debug_set(b"gemm_ph4", ph))
            for fit in (5, 6, 7, 8, 1):
                check(lib().afx_debug_set(b"gemm_fit", fit))
                for _ in range(4):
                    got_f, got_h = K.gemm("fp16", A, W, bias=bias, resid=R, out_f=True, out_h=True)
                    assert torch.equal(got_f, want_f) and torch.equal(got_h, want_h), (ph, fit)
    finally:
        check(lib().afx_debug_set(b"gemm_tile", -1))
        check(lib().afx_debug_set(b"gemm_fit", 1))
        check(lib().afx_debug_set(b"gemm_ph4", 0))


@pytest.mark.parametrize("M,N,K_,resid", [(12736, 3072, 1024, False), (12736, 4096, 1024, False), (6500, 2048, 512, True)])
def test_gemm_round_split_matches_the_single_kernel(K, M, N, K_, resid):
    """Dispatch of a multi-round plain GEMM: either a fitted tile height on the whole problem (default where the cost
    model prefers it) or whole rounds on the 8-wave kernel + the remaining rows on the 128x128 kernel (gemm_fit 0 forces
    this form).  Same k order per element everywhere, so every form must equal the single-kernel result bit for bit."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(N + K_)
    A = torch.randn(M, K_, generator=g).half().cuda()
    W = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda() if resid else None
    try:
        check(lib().afx_debug_set(b"gemm_tile", 0))
        want_f, want_h = K.gemm("fp16", A, W, bias=bias, act="gelu", resid=R, out_f=True, out_h=True)
        check(lib().afx_debug_set(b"gemm_tile", -1))
        for fit in (1, 0):  # default dispatch, then the round split
            check(lib().afx_debug_set(b"gemm_fit", fit))
            got_f, got_h = K.gemm("fp16", A, W, bias=bias, act="gelu", resid=R, out_f=True, out_h=True)
            assert torch.equal(got_f, want_f) and torch.equal(got_h, want_h), fit
        check(lib().afx_debug_set(b"gemm_split", 0))
        got_f, _ = K.gemm("fp16", A, W, bias=bias, act="gelu", resid=R, out_f=True, out_h=True)
        assert torch.equal(got_f, want_f)
    finally:
        check(lib().afx_debug_set(b"gemm_tile", -1))
        check(lib().afx_debug_set(b"gemm_split", 1))
        check(lib().afx_debug_set(b"gemm_fit", 1))


@pytest.mark.parametrize("dtype", DT)
def test_gemm_asymmetric_identity_catches_transposes(K, dtype):
    # A = I, asymmetric W: C must equal W^T exactly (cdna guide: A=I check with asymmetric B)
    n = 128
    A = torch.eye(n).to(_td(dtype))
    W = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(_td(dtype))
    of, _ = K.gemm(dtype, A.cuda(), W.cuda())
    assert torch.equal(of.cpu(), W.float().t())


def test_epilogue_gelu_polynomial_is_erf_gelu_to_3e6(K):
    """The half-precision epilogues evaluate erf-GELU without transcendentals (afx_common.h gelu_poly2).  A = I passes
    a dense grid of fp16 values (|x| <= 12, clamp region included) through the GEMM untouched: the fp32 output must
    be the exact erf-GELU of those values to 3e-6 absolute, and exactly x / exactly 0 beyond the clamp at +-5."""
    n = 256
    A = torch.eye(n).half()
    W = (torch.linspace(-12, 12, n * n).reshape(n, n)).half()
    of, _ = K.gemm("fp16", A.cuda(), W.cuda(), act="gelu")
    x = W.float().t().double()
    ref = 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))
    got = of.cpu().double()
    assert (got - ref).abs().max().item() <= 3e-6
    assert torch.equal(got[x >= 5], x[x >= 5]) and bool((got[x <= -5] == 0).all())


@pytest.mark.parametrize("dtype", DT)
def test_weight_packing_layouts(K, dtype):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(10, 144, generator=g)
    p = K.pack_linear(dtype, w.cuda(), 192).cpu()
    assert torch.equal(p[:, :144], w.to(_td(dtype))) and bool((p[:, 144:] == 0).all())
    wc = torch.randn(6, 5, 3, generator=g)
    pc = K.pack_conv(dtype, wc.cuda()).cpu()
    assert torch.equal(pc, wc.permute(0, 2, 1).reshape(6, 15).to(_td(dtype)))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k,s,Tin", [(3, 2, 85), (2, 2, 41)])
def test_conv_layer_as_gemm(K, dtype, k, s, Tin):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(3, Tin, 512, generator=g).to(_td(dtype))
    w = (torch.randn(512, 512, k, generator=g) / math.sqrt(512 * k))
    bias = torch.randn(512, generator=g) * 0.1
    wp = K.pack_conv(dtype, w.cuda())
    got = K.conv_gemm(dtype, x.cuda(), wp, k, s, bias.cuda())
    ref = F.conv1d(x.float().transpose(1, 2), w.to(_td(dtype)).float(), bias, stride=s).transpose(1, 2)
    assert got.shape == ref.shape
    _close(got, ref, 1e-4, 2e-4)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k,s,Tin", [(3, 2, 301), (2, 2, 40)])
@pytest.mark.parametrize("deep", [0, 2])  # 2-stage kernel, 8-phase kernel
def test_conv_layer_with_fused_layernorm_gelu(K, dtype, k, s, Tin, deep):
    """Row-complete 128x512 tile: conv + bias + LayerNorm(512) + GELU in one kernel, M tail
    (B*Tout not a multiple of 128) included; fp32 and operand-type outputs."""
    g = torch.Generator().manual_seed(100 + k)
    x = torch.randn(3, Tin, 512, generator=g).to(_td(dtype))
    w = torch.randn(512, 512, k, generator=g) / math.sqrt(512 * k)
    bias = torch.randn(512, generator=g) * 0.5
    ga = 1 + 0.1 * torch.randn(512, generator=g)
    be = 0.1 * torch.randn(512, generator=g)
    wp = K.pack_conv(dtype, w.cuda())
    from afx._lib import check, lib
    check(lib().afx_debug_set(b"gemm_deep", deep))
    try:
        of, oh = K.conv_ln_act(dtype, x.cuda(), wp, k, s, bias.cuda(), ga.cuda(), be.cuda(), out_f=True, out_h=True)
    finally:
        check(lib().afx_debug_set(b"gemm_deep", -1))
    ref = F.conv1d(x.float().transpose(1, 2), w.to(_td(dtype)).float(), bias, stride=s).transpose(1, 2)
    ref = F.gelu(F.layer_norm(ref, (512,), ga, be, 1e-5))
    assert of.shape == ref.shape
    _close(of, ref, 2e-4, 2e-4)
    _close(oh, ref, _eps(dtype) * 4, 1e-3)


def test_conv_layernorm_8phase_tile_is_bit_identical_to_the_2stage_tile(K):
    """Race screen for the 8-phase 128x512 row-complete tile (see the 256x256 one above)."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 3199, 512, generator=g).half().cuda()
    bias = (torch.randn(512, generator=g) * 0.5).cuda()
    ga = (1 + 0.1 * torch.randn(512, generator=g)).cuda()
    be = (0.1 * torch.randn(512, generator=g)).cuda()
    for k in (3, 2):
        wp = K.pack_conv("fp16", (torch.randn(512, 512, k, generator=g) / math.sqrt(512 * k)).cuda())
        try:
            check(lib().afx_debug_set(b"gemm_deep", 0))  # the 2-stage tile
            want, _ = K.conv_ln_act("fp16", x, wp, k, 2, bias, ga, be, out_f=True, out_h=False)
            check(lib().afx_debug_set(b"gemm_deep", 2))
            for fit in (14, 13, 12, 1):  # tile heights 128 / 96 / 64 rows, then the fitted choice: same rows, bit for bit
                check(lib().afx_debug_set(b"gemm_fit", fit))
                for _ in range(4):
                    got, _ = K.conv_ln_act("fp16", x, wp, k, 2, bias, ga, be, out_f=True, out_h=False)
                    assert torch.equal(got, want), fit
        finally:
            check(lib().afx_debug_set(b"gemm_deep", -1))
            check(lib().afx_debug_set(b"gemm_fit", 1))


def test_conv_layer_remainder_split_is_bit_identical(K):
    """A conv layer of more than one round of 128-row tiles whose last round is at most half full runs as whole rounds +
    a second launch of 64-row tiles over the remaining rows (rows [m_lo, M) of the same kernel): 300 tiles here = one round
    + 44 tiles.  Same rows as the single launch and as the 2-stage tile, bit for bit."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(91)
    x = torch.randn(24, 3199, 512, generator=g).half().cuda()
    bias = (torch.randn(512, generator=g) * 0.5).cuda()
    ga = (1 + 0.1 * torch.randn(512, generator=g)).cuda()
    be = (0.1 * torch.randn(512, generator=g)).cuda()
    wp = K.pack_conv("fp16", (torch.randn(512, 512, 3, generator=g) / math.sqrt(1536)).cuda())
    try:
        check(lib().afx_debug_set(b"gemm_deep", 0))  # the 2-stage tile
        want, _ = K.conv_ln_act("fp16", x, wp, 3, 2, bias, ga, be, out_f=True, out_h=False)
        check(lib().afx_debug_set(b"gemm_deep", 2))
        for split in (1, 0):
            check(lib().afx_debug_set(b"gemm_conv_split", split))
            for _ in range(3):
                got, _ = K.conv_ln_act("fp16", x, wp, 3, 2, bias, ga, be, out_f=True, out_h=False)
                assert torch.equal(got, want), split
    finally:
        check(lib().afx_debug_set(b"gemm_deep", -1))
        check(lib().afx_debug_set(b"gemm_conv_split", 1))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("pre", [False, True])
def test_conv0_layernorm_gelu(K, dtype, pre):
    from oracle import pre as opre
    g = torch.Generator().manual_seed(5)
    wave = torch.randn(3, 4000 + 7, generator=g) * 0.1
    w = torch.randn(512, 1, 10, generator=g) * math.sqrt(0.2)
    b = torch.randn(512, generator=g) * 0.02
    ga = 1 + 0.1 * torch.randn(512, generator=g)
    be = 0.1 * torch.randn(512, generator=g)
    got = K.conv0(dtype, wave.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda(), pre_emph=pre)
    xin = opre.pre_emphasis(wave) if pre else wave
    ref = F.conv1d(xin.unsqueeze(1), w, b, stride=5).transpose(1, 2)
    ref = F.gelu(F.layer_norm(ref, (512,), ga, be, 1e-5))
    assert got.shape == ref.shape
    _close(got, ref, _eps(dtype), 2e-3 if dtype == "fp16" else 1e-2)


@pytest.mark.parametrize("amp,wscale", [(0.1, 1.0), (1e-4, 1.0), (30.0, 1.0), (0.05, 0.01), (3e-7, 20.0), (3e7, 1.0)])
def test_conv0_split_precision_form_is_fp32_accurate_at_any_level(K, amp, wscale):
    """The half-precision engines run conv layer 0 as ONE fp16 MFMA per tile (hi/lo split of samples and weights, per-frame
    and per-layer power-of-two scaling folded into the LayerNorm).  Against the true-fp32 matrix-core form of the same
    layer (conv0_mfma = 2) the fp16 outputs may differ only where a value sits on a rounding boundary -- whatever the
    recording level and the weight scale, silence between loud frames included."""
    from afx._lib import check, lib
    g = torch.Generator().manual_seed(int(amp * 1e7) % 1000 + 3)
    wave = torch.randn(2, 8000 + 3, generator=g) * amp
    wave[0, 1000:1400] *= 1e-3      # a quiet stretch inside a loud clip: the scale is per frame
    wave[1, 3000:3050] = 0.0        # digital silence: the bias decides
    w = torch.randn(512, 1, 10, generator=g) * math.sqrt(0.2) * wscale
    b = torch.randn(512, generator=g) * 0.02
    ga = 1 + 0.1 * torch.randn(512, generator=g)
    be = 0.1 * torch.randn(512, generator=g)
    args = (wave.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda())
    try:
        check(lib().afx_debug_set(b"conv0_mfma", 2))
        want = K.conv0("fp16", *args).float()
        check(lib().afx_debug_set(b"conv0_mfma", 1))
        got = K.conv0("fp16", *args).float()
    finally:
        check(lib().afx_debug_set(b"conv0_mfma", 1))
    ref = F.gelu(F.layer_norm(F.conv1d(wave.double().unsqueeze(1), w.double(), b.double(), stride=5).transpose(1, 2), (512,),
                              ga.double(), be.double(), 1e-5)).float()
    ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
    assert ((got.cpu() - ref).abs() <= 1.01 * ulp + 4e-6).all(), float(((got.cpu() - ref).abs() / ulp).max())
    differ = (got != want).float().mean().item()
    assert differ < 2e-3, differ  # one-ulp disagreements on rounding boundaries only
    assert ((got - want).abs().cpu() <= 2.02 * ulp + 4e-6).all()  # each within one ulp of the exact value


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,act", [(512, "gelu"), (1024, None), (144, None)])
def test_row_layernorm(K, dtype, C, act):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(37, C, generator=g) * 3 + 0.5
    ga = 1 + 0.1 * torch.randn(C, generator=g)
    be = 0.1 * torch.randn(C, generator=g)
    of, oh = K.rownorm(dtype, x.cuda(), ga.cuda(), be.cuda(), act=act, out_h=True)
    ref = F.layer_norm(x, (C,), ga, be, 1e-5)
    if act == "gelu":
        ref = F.gelu(ref)
    _close(of, ref, 1e-5, 1e-5)
    _close(oh, ref, _eps(dtype) * 4, 1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("T", [199, 201, 49, 12, 224])
def test_transformer_attention(K, dtype, T):
    B, H = 2, 16
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g)).to(_td(dtype))
    got = K.mhsa(dtype, qkv.cuda(), B, T, H).float().cpu().view(B, T, H, 64)
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3)
    _close(got, ref, 2e-2 if dtype == "bf16" else 4e-3, 2e-2 if dtype == "bf16" else 3e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("T", [225, 300, 499, 1000])
def test_transformer_attention_any_length(K, dtype, T):
    """Clips longer than 4.5 s (the reference's test_duration_sec is a free config value): keys stream through
    LDS in blocks of 128 with running row statistics (mhsa_long_kernel)."""
    B, H = 2, 3
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g)).to(_td(dtype))
    got = K.mhsa(dtype, qkv.cuda(), B, T, H).float().cpu().view(B, T, H, 64)
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3)
    _close(got, ref, 2e-2 if dtype == "bf16" else 4e-3, 2e-2 if dtype == "bf16" else 3e-3)


@pytest.mark.parametrize("T", [1, 17, 128, 129, 199, 224])
def test_transformer_attention_blocked_kernel_agrees_with_the_one_pass_kernel(K, T):
    """The any-length kernel forced onto short clips: same probabilities up to the rounding of P (relative to the
    running instead of the final row maximum), block edges at 128 included; and launch-to-launch identical."""
    from afx._lib import check, lib
    B, H = 3, 4
    g = torch.Generator().manual_seed(1000 + T)
    qkv = torch.randn(B * T, 3 * H * 64, generator=g).half().cuda()
    want = K.mhsa("fp16", qkv, B, T, H).float()
    try:
        check(lib().afx_debug_set(b"mhsa_force_long", 1))
        got = K.mhsa("fp16", qkv, B, T, H).float()
        again = K.mhsa("fp16", qkv, B, T, H).float()
    finally:
        check(lib().afx_debug_set(b"mhsa_force_long", 0))
    _close(got, want, 2e-3, 2e-3)
    assert torch.equal(got, again)


@pytest.mark.parametrize("N,H,dh", [(200, 4, 36), (50, 4, 36), (13, 4, 32)])
def test_conformer_relative_attention(K, N, H, dh):
    B = 2
    g = torch.Generator().manual_seed(N)
    q = torch.randn(B * N, H * dh, generator=g)
    kv = torch.randn(B * N, 2 * H * dh, generator=g)
    rel = torch.randn(1025, dh, generator=g)
    got = K.conf_attn("fp16", q.cuda(), kv.cuda(), rel.cuda(), B, N, H, dh).float().cpu()
    qq = q.view(B, N, H, dh).transpose(1, 2)
    kk = kv[:, : H * dh].reshape(B, N, H, dh).transpose(1, 2)
    vv = kv[:, H * dh:].reshape(B, N, H, dh).transpose(1, 2)
    seq = torch.arange(N)
    dist = (seq[:, None] - seq[None, :]).clamp(-512, 512) + 512
    dots = (torch.einsum("bhid,bhjd->bhij", qq, kk) + torch.einsum("bhnd,nrd->bhnr", qq, rel[dist])) * dh ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", torch.softmax(dots, -1), vv).transpose(1, 2).reshape(B * N, H * dh)
    _close(got, ref, 2e-3, 2e-3)


@pytest.mark.parametrize("N,H,dh", [(300, 4, 36), (700, 2, 36), (257, 2, 64), (1200, 1, 36)])
def test_conformer_relative_attention_any_length(K, N, H, dh):
    """Beyond 256 tokens the fp32 kernel takes 256-query chunks and 64-key LDS blocks; beyond 513 the Shaw
    distance clamp (max_pos_emb = 512) is active."""
    B = 2
    g = torch.Generator().manual_seed(N)
    q = torch.randn(B * N, H * dh, generator=g)
    kv = torch.randn(B * N, 2 * H * dh, generator=g)
    rel = torch.randn(1025, dh, generator=g)
    got = K.conf_attn("fp16", q.cuda(), kv.cuda(), rel.cuda(), B, N, H, dh).float().cpu()
    qq = q.view(B, N, H, dh).transpose(1, 2)
    kk = kv[:, : H * dh].reshape(B, N, H, dh).transpose(1, 2)
    vv = kv[:, H * dh:].reshape(B, N, H, dh).transpose(1, 2)
    seq = torch.arange(N)
    dist = (seq[:, None] - seq[None, :]).clamp(-512, 512) + 512
    dots = torch.einsum("bhid,bhjd->bhij", qq, kk)
    for b in range(B):  # (N, N, dh) gather per utterance: bounded memory
        dots[b] += torch.einsum("hnd,nrd->hnr", qq[b], rel[dist])
    ref = torch.einsum("bhij,bhjd->bhid", torch.softmax(dots * dh ** -0.5, -1), vv).transpose(1, 2).reshape(B * N, H * dh)
    _close(got, ref, 2e-3, 2e-3)


@pytest.mark.parametrize("N,block", [(200, 64), (200, 12), (50, 48), (13, 4)])
def test_conformer_relative_attention_blocked_form_is_bit_identical(K, N, block):
    """Keys are walked in the same order whether they sit in LDS all at once or a block at a time."""
    from afx._lib import check, lib
    B, H, dh = 2, 4, 36
    g = torch.Generator().manual_seed(N + block)
    q = torch.randn(B * N, H * dh, generator=g).cuda()
    kv = torch.randn(B * N, 2 * H * dh, generator=g).cuda()
    rel = torch.randn(1025, dh, generator=g).cuda()
    want = K.conf_attn("fp16", q, kv, rel, B, N, H, dh)
    try:
        check(lib().afx_debug_set(b"conf_attn_block", block))
        got = K.conf_attn("fp16", q, kv, rel, B, N, H, dh)
    finally:
        check(lib().afx_debug_set(b"conf_attn_block", 0))
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("N", [200, 50, 209, 13, 210, 400, 700, 1200])  # > 209: keys in blocks of 128; > 513: distance clamp
def test_conformer_relative_attention_on_matrix_cores(K, dtype, N):
    """The MFMA form (S1 = K Q^T, R = E_win Q^T skewed through LDS, P V): against the fp32 definition
    computed from the ROUNDED q, k, v, E (the kernel rounds them to the operand type)."""
    B, H, dh = 2, 4, 36
    g = torch.Generator().manual_seed(N)
    rnd = lambda t: t.to(_td(dtype)).float()
    q = torch.randn(B * N, H * dh, generator=g)
    kv = torch.randn(B * N, 2 * H * dh, generator=g)
    rel = torch.randn(1025, dh, generator=g)
    got = K.conf_attn_mfma(dtype, q.cuda(), kv.cuda(), rel.cuda(), B, N, H, dh).float().cpu()
    qq = rnd(q * dh ** -0.5).view(B, N, H, dh).transpose(1, 2)
    kk = rnd(kv[:, : H * dh]).reshape(B, N, H, dh).transpose(1, 2)
    vv = rnd(kv[:, H * dh:]).reshape(B, N, H, dh).transpose(1, 2)
    seq = torch.arange(N)
    dist = (seq[:, None] - seq[None, :]).clamp(-512, 512) + 512
    dots = torch.einsum("bhid,bhjd->bhij", qq, kk)
    for bi in range(B):  # (N, N, dh) gather per utterance: bounded memory
        dots[bi] += torch.einsum("hnd,nrd->hnr", qq[bi], rnd(rel)[dist])
    ref = torch.einsum("bhij,bhjd->bhid", torch.softmax(dots, -1), vv).transpose(1, 2).reshape(B * N, H * dh)
    _close(got, ref, _eps(dtype) * 6, _eps(dtype) * 6)


@pytest.mark.parametrize("N", [200, 129, 128, 50, 13])
def test_conformer_relative_attention_blocked_matrix_core_kernel_agrees_with_the_one_pass_kernel(K, N):
    from afx._lib import check, lib
    B, H, dh = 2, 4, 36
    g = torch.Generator().manual_seed(7 * N)
    q = torch.randn(B * N, H * dh, generator=g).cuda()
    kv = torch.randn(B * N, 2 * H * dh, generator=g).cuda()
    rel = torch.randn(1025, dh, generator=g).cuda()
    want = K.conf_attn_mfma("fp16", q, kv, rel, B, N, H, dh).float()
    try:
        check(lib().afx_debug_set(b"conf_attn_force_long", 1))
        got = K.conf_attn_mfma("fp16", q, kv, rel, B, N, H, dh).float()
        again = K.conf_attn_mfma("fp16", q, kv, rel, B, N, H, dh).float()
    finally:
        check(lib().afx_debug_set(b"conf_attn_force_long", 0))
    _close(got, want, 3e-3, 3e-3)
    assert torch.equal(got, again)


@pytest.mark.parametrize("k", [31, 16])
@pytest.mark.parametrize("N", [200, 1024, 1025, 2500])  # one chunk; the chunk size; chunks with halos
def test_conformer_glu_depthwise_bn_swish(K, k, N):
    B, C = 2, 288
    g = torch.Generator().manual_seed(k)
    x = torch.randn(B * N, 2 * C, generator=g)
    w = torch.randn(C, 1, k, generator=g) / math.sqrt(k)
    b = torch.randn(C, generator=g) * 0.1
    sc = 1 + 0.1 * torch.randn(C, generator=g)
    sh = 0.1 * torch.randn(C, generator=g)
    got = K.conf_dwconv("fp16", x.cuda(), w.cuda(), b.cuda(), sc.cuda(), sh.cuda(), B, N, C, k).float().cpu()
    h = x.view(B, N, 2 * C).transpose(1, 2)
    u = h[:, :C] * torch.sigmoid(h[:, C:])
    pad = (k // 2, k // 2 - (k + 1) % 2)
    y = F.conv1d(F.pad(u, pad), w, b, groups=C)
    y = y * sc[None, :, None] + sh[None, :, None]
    ref = (y * torch.sigmoid(y)).transpose(1, 2).reshape(B * N, C)
    _close(got, ref, 2e-3, 2e-3)
