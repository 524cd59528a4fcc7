"""GPU parity tests of the building-block kernels, called through the C ABI (ctypes) and compared
with plain PyTorch fp32 on the SAME (already rounded) operands. Tolerances: fp32 path 2e-5 relative
to the output scale (accumulation order only); 16-bit paths additionally carry ONE output rounding
(2^-8 bf16, 2^-11 f16) where the kernel stores 16-bit."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import probe_arms

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
OUT_TOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3, torch.float16: 8e-4}


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C', [(7, 128), (1000, 768), (513, 1024)])
def test_layernorm(gpu, dtype, rows, C):
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(rows + C)
    x = (torch.randn((rows, C), generator=g) * 3 + 0.5).to(gpu)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(gpu), (0.1 * torch.randn(C, generator=g)).to(gpu)
    out = ops.layernorm(x, gamma, beta, eps=1e-6, dtype=dtype)
    ref = F.layer_norm(x, (C,), gamma, beta, eps=1e-6)
    assert relerr(out, ref) < OUT_TOL[dtype]


@pytest.mark.parametrize('rows,C', [(7, 128), (1000, 768), (513, 1024), (33, 1032), (5, 2048), (9, 16)])
def test_layernorm_split_fp16_rows(gpu, rows, C):
    """LayerNorm into the default engine's split-fp16 rows (hi + lo fp16 per element, 32-byte groups [hi x8][lo x8]): 22 significand bits,
    so fp32-class against torch's fp64 LayerNorm."""
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(rows * 3 + C)
    x = (torch.randn((rows, C), generator=g) * 3 + 0.5).to(gpu)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(gpu), (0.1 * torch.randn(C, generator=g)).to(gpu)
    ref = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), eps=1e-6)
    out = ops.layernorm_x3(x, gamma, beta, eps=1e-6)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(300, 192, 256), (128, 128, 128), (1000, 3072, 1024), (77, 96, 768)])
def test_linear_epilogues(gpu, dtype, M, N, K):
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N)
    a = torch.randn((M, K), generator=g).to(gpu).to(dtype)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu).to(dtype)
    b = torch.randn(N, generator=g).to(gpu)
    res = torch.randn((M, N), generator=g).to(gpu)
    ref = a.float() @ w.float().T + b
    out = ops.linear(a, w, b, 'store')
    assert relerr(out, ref) < OUT_TOL[dtype], 'store epilogue'
    out = ops.linear(a, w, b, 'f32', residual=res)
    assert out.dtype == torch.float32 and relerr(out, ref + res) < 2e-5, 'fp32 + residual epilogue'
    out = ops.linear(a, w, None, 'f32')
    assert relerr(out, a.float() @ w.float().T) < 2e-5, 'no-bias'
    out = ops.linear(a, w, b, 'gelu')
    assert relerr(out, F.gelu(ref)) < OUT_TOL[dtype], 'gelu epilogue'


@pytest.mark.parametrize('sw', probe_arms(['1', '0'], ['0']))
@pytest.mark.parametrize('cfg', ['0', '0w8', '1', '2', '3', '7', '8', '9', '11'])
@pytest.mark.parametrize('M,N,K', [(300, 192, 256), (1000, 3072, 1024), (77, 96, 768), (2100, 1032, 32), (515, 328, 64)])
def test_linear_split_fp16(gpu, M, N, K, cfg, sw, monkeypatch):
    """The parity-grade precision mode (fp16x3: operands split into fp16 hi + lo, three MFMAs per product) at kernel level:
    every tile configuration ('7' = the two-blocks-per-CU 256 x 128 shape with a K step's weight fragments in registers, '8' = the
    64 x 64 tile of the small-batch forwards), the
    software-pipelined K loop and the plain two-stage loop, the LDS-staged wide epilogue and the direct stores; K = 32 / 64 are one- and two-step K loops (pipeline prologue / drain only). Reference =
    fp32 matmul of the SAME fp32 operands: the split keeps 22 significand bits per operand, so the result is fp32-class."""
    from dust3r_amd import ops
    from conftest import need_probes
    if sw == '1':
        need_probes('the software-pipelined K loop')
    monkeypatch.setenv('D3R_GEMM_CFG', cfg[:-2] if cfg.endswith(('w8', 'w4')) else cfg)
    monkeypatch.setenv('D3R_GEMM_T128W8', '1000000' if cfg.endswith('w8') else '0')     # '0w8': the 128 x 128 tile by eight waves (small-batch forwards)
    monkeypatch.setenv('D3R_GEMM_X3SW', sw)
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N)
    a = torch.randn((M, K), generator=g).to(gpu)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    res = torch.randn((M, N), generator=g).to(gpu)
    ref = (a.double() @ w.double().T + b.double()).float()
    got = {}
    for wide in ('0', '1'):
        monkeypatch.setenv('D3R_GEMM_NOWIDE', wide)
        o_store, o_res, o_gelu = ops.linear_x3(a, w, b, 'store'), ops.linear_x3(a, w, b, 'f32', residual=res), ops.linear_x3(a, w, b, 'gelu')
        assert relerr(o_store, ref) < 3e-6, 'store epilogue'
        assert relerr(o_res, ref + res) < 3e-6, 'fp32 + residual epilogue'
        assert relerr(o_gelu, F.gelu(ref)) < 3e-6, 'gelu epilogue'
        got[wide] = (o_store, o_res, o_gelu)
    for x, y in zip(got['0'], got['1']):       # two routes for the same values
        assert torch.equal(x, y)


@pytest.mark.parametrize('grid', probe_arms([None, '8'], [None]))
@pytest.mark.parametrize('M,N,K', [(2048, 256, 768), (4096, 384, 1024), (2048, 128, 1536), (6144, 256, 576), (16384, 1536, 768)])
def test_persistent_gemm_is_bit_identical(gpu, M, N, K, grid, monkeypatch):
    """gemm_p4.hip (one block per CU walking its tiles, the epilogue of tile t drained under the K loop of tile t + 1) against the one-tile-per-block
    kernels on the same operands: typed store, GELU, and the typed residual stream with and without a residual and with its LayerNorm partial sums --
    BIT-identical (same MFMA order per element, same epilogue expressions, same summation tree), and within fp32-class error of the fp64 product.
    K = 576 / 768: 18 / 24 K steps (the shortest loop the kernel takes); 1536: steps behind the draining ones; grid = 8: forty tiles per block (the
    overlapped path; probe builds), default: one or two (first tile without a drain, last tile drained with nothing to hide under) -- and three per block on
    the 16384 x 1536 problem (768 tiles on 256 resident blocks: first / overlapped / last tile, the steady state of the 32-pair step, in every build)."""
    from dust3r_amd import ops
    from conftest import need_probes
    if grid:
        need_probes('D3R_P4_GRID')
        if M > 8192:
            pytest.skip('covered by the default grid')
        monkeypatch.setenv('D3R_P4_GRID', grid)
    g = torch.Generator(device='cpu').manual_seed(M + 3 * N + K)
    a = torch.randn((M, K), generator=g).to(gpu)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    res = torch.randn((M, N), generator=g).to(gpu)
    ref = (a.double() @ w.double().T + b.double()).float()
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('D3R_GEMM_PERSIST', mode)
        o_store, o_gelu = ops.linear_x3(a, w, b, 'store'), ops.linear_x3(a, w, None, 'gelu')
        r1, p1, raw1 = ops.linear_x3res(a, w, b, res, with_sums=True)
        r2, _, raw2 = ops.linear_x3res(a, w, b, None, with_sums=False)
        out[mode] = (o_store, o_gelu, raw1, p1, raw2)
        assert relerr(o_store, ref) < 3e-6 and relerr(o_gelu, F.gelu(ref - b)) < 3e-6
        assert relerr(r1, ref + ops.unpack_x3(ops.pack_x3(res))) < 3e-6 and relerr(r2, ref) < 3e-6
        stored = r1.double().view(M, N // 32, 32)
        assert float((p1[..., 0].double() - stored.sum(-1)).abs().max()) < 1e-3 and float((p1[..., 1].double() / (stored * stored).sum(-1) - 1).abs().max()) < 1e-5
    for x, y in zip(out['0'], out['1']):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.float16 else x, y.view(torch.int16) if y.dtype == torch.float16 else y)


def test_persistent_gemm_dispatch_rule():
    """Host only: which launches the heuristic hands to the persistent kernel (d3r_gemm_tile_config = 10) -- the encoder's fc1 / typed-residual projection
    and the decoder's fc1 at 32 pairs per step; not fc2 at K = 4096 (the K loop dominates: measured equal), not the decoder's N = 768 GEMMs (576 tiles of
    256 x 128 = 2.25 rounds of 256 CUs), not small batches (fewer tiles than CUs)."""
    from dust3r_amd._lib import lib, DTYPE_F16X3
    cfg = lambda M, N, K, epi, res=0: lib.d3r_gemm_tile_config(DTYPE_F16X3, M, N, K, epi, res)     # noqa: E731
    assert cfg(49152, 4096, 1024, 2) == 10 and cfg(24576, 3072, 768, 2) == 10 and cfg(49152, 3072, 1024, 0) == 10
    assert cfg(1536, 4096, 1024, 2) != 10 and cfg(49152, 4096, 1024, 1, 1) != 10      # one pair; fp32 residual epilogue (not the typed stream)
    assert cfg(24576, 768, 768, 0) != 10


@pytest.mark.parametrize('cfg', probe_arms(['0', '1', '2', '3', '1w4'], ['0', '1', '2', '3']))
@pytest.mark.parametrize('M,N,K', [(300, 192, 256), (1000, 3072, 1024), (77, 128, 768), (2100, 1024, 64), (515, 320, 128)])
def test_linear_fp16_fp8(gpu, M, N, K, cfg, monkeypatch):
    """The fp16 + fp8 operand mode at kernel level (hi.hi on the f16 MFMA, both cross terms on one K-concatenated e4m3 MFMA with an E8M0
    scale of 2^-17): every tile configuration, wide and direct epilogues. The comparator is the SAME arithmetic in fp64 on the host-packed
    operands (oracle/f8_ref.py, an independent restatement of the encodings): the kernel must reproduce it to fp32 accumulation noise, i.e. the row layouts, the k-slot pairing of the
    two operands and the scale are all pinned; the distance to the exact product is checked too (an fp16-only product is ~30x worse).
    Activation-row outputs ('store' / 'gelu') are compared after decoding (hi + lo8 2^-11: one output rounding of 2^-15)."""
    from dust3r_amd import ops
    from oracle.f8_ref import f16f8_matmul
    # '1w4': the two-blocks-per-CU shape of the 256-wide configuration (256 x 128 tile by four waves, 64-byte K steps, three LDS slots)
    from conftest import need_probes
    if cfg.endswith('w4'):
        need_probes('the four-wave 256 x 128 fp16 + fp8 tile')
    monkeypatch.setenv('D3R_GEMM_CFG', cfg[:-2] if cfg.endswith(('w8', 'w4')) else cfg)
    monkeypatch.setenv('D3R_GEMM_F8W4', '1' if cfg.endswith('w4') else '0')
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N)
    a = torch.randn((M, K), generator=g).to(gpu)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    res = torch.randn((M, N), generator=g).to(gpu)
    emu = (f16f8_matmul(a, w) + b.double()).float()
    exact = (a.double() @ w.double().T + b.double()).float()
    assert relerr(emu, exact) < 4e-5          # the scheme itself: ~15-16 bits per operand
    got = {}
    for wide in ('0', '1'):
        monkeypatch.setenv('D3R_GEMM_NOWIDE', wide)
        o_res = ops.linear_f8(a, w, b, 'f32', residual=res)
        assert relerr(o_res, emu + res) < 3e-6, 'fp32 + residual epilogue'
        assert relerr(ops.linear_f8(a, w, None, 'f32'), emu - b) < 3e-6, 'no bias'
        if N % 64 == 0:
            o_store, o_gelu = ops.linear_f8(a, w, b, 'store'), ops.linear_f8(a, w, b, 'gelu')
            assert relerr(o_store, emu) < 4e-5, 'activation-row store'
            assert relerr(o_gelu, F.gelu(emu)) < 4e-5, 'gelu epilogue'
            got[wide] = (o_res, o_store, o_gelu)
        else:
            got[wide] = (o_res,)
    for x, y in zip(got['0'], got['1']):
        assert torch.equal(x, y)


@pytest.mark.parametrize('cfg', ['0', '1'])
@pytest.mark.parametrize('M,N,K', [(300, 192, 256), (1024, 768, 768), (515, 1024, 1024), (2048, 256, 4096), (100, 64, 128)])
def test_linear_2p5_unit(gpu, M, N, K, cfg, monkeypatch):
    """The 2.5-unit operand mode (D3R_DTYPE_F16X2F8) at kernel level: hi.hi and hi.w_lo on the f16 MFMA, a_lo.w_hi on the e4m3 MFMA whose 128 k of
    b8 the DMA gathers from two super-groups. Comparator: the SAME arithmetic in fp64 (oracle/f8_ref.py f16x2f8_matmul) -- the kernel must reproduce it
    to fp32 accumulation noise, which pins the five-chunk weight rows, the chunk schedule of the K loop, the gather and the scale --, and the distance of
    the scheme to the exact product (weights exact to 22 bits, activations to ~15: 3x closer than fp16 + fp8)."""
    from dust3r_amd import ops
    from oracle.f8_ref import f16f8_matmul, f16x2f8_matmul
    monkeypatch.setenv('D3R_GEMM_CFG', cfg)
    g = torch.Generator(device='cpu').manual_seed(M * 11 + N)
    a = torch.randn((M, K), generator=g).to(gpu)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    res = torch.randn((M, N), generator=g).to(gpu)
    emu = (f16x2f8_matmul(a, w) + b.double()).float()
    exact = (a.double() @ w.double().T + b.double()).float()
    assert relerr(emu, exact) < 2e-5 and relerr(emu, exact) < relerr((f16f8_matmul(a, w) + b.double()).float(), exact)
    got = {}
    for wide in ('0', '1'):
        monkeypatch.setenv('D3R_GEMM_NOWIDE', wide)
        o_res = ops.linear_x2f8(a, w, b, 'f32', residual=res)
        assert relerr(o_res, emu + res) < 3e-6, 'fp32 + residual epilogue'
        assert relerr(ops.linear_x2f8(a, w, None, 'f32'), emu - b) < 3e-6, 'no bias'
        if N % 64 == 0:
            o_store, o_gelu = ops.linear_x2f8(a, w, b, 'store'), ops.linear_x2f8(a, w, b, 'gelu')
            assert relerr(o_store, emu) < 4e-5, 'activation-row store'
            assert relerr(o_gelu, F.gelu(emu)) < 4e-5, 'gelu epilogue'
            got[wide] = (o_res, o_store, o_gelu)
        else:
            got[wide] = (o_res,)
    for x, y in zip(got['0'], got['1']):
        assert torch.equal(x, y)


def test_linear_fp16_fp8_operand_beyond_4_gib(gpu):
    """The fp16 + fp8 K loop addresses its DMA sources as a wave-uniform 64-bit tile base (SGPRs) plus ONE 32-bit offset per lane: the
    offsets are relative to the tile's first row, so an activation operand larger than 4 GiB (here 280 000 x 4096 x 4 B = 4.6 GB: the
    MLP hidden of ~180 pairs at 512x384 in one engine call) is fine. First and last rows against the fp64 emulation."""
    from dust3r_amd import ops
    from oracle.f8_ref import f16f8_matmul
    M, N, K = 280000, 128, 4096
    g = torch.Generator(device=gpu).manual_seed(5)
    a = torch.randn((M, K), generator=g, device=gpu)
    w = torch.randn((N, K), generator=g, device=gpu) / math.sqrt(K)
    out = ops.linear_f8(a, w, None, 'f32')
    for sl in (slice(0, 512), slice(M - 777, M)):
        assert relerr(out[sl], f16f8_matmul(a[sl], w).float()) < 3e-6


@pytest.mark.parametrize('rows,C', [(7, 128), (1000, 768), (513, 1024)])
def test_layernorm_fp16_fp8_rows(gpu, rows, C):
    """LayerNorm into fp16 + fp8 activation rows: hi is the fp16 rounding of the fp32 LayerNorm, a8 = e4m3(hi) exactly, and
    b8 = e4m3((x - hi) 2^11) up to a half-step of its own rounding (the kernel's fp32 LayerNorm may differ from torch's by an ulp)."""
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(rows + C)
    x = (torch.randn((rows, C), generator=g) * 3 + 0.5).to(gpu)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(gpu), (0.1 * torch.randn(C, generator=g)).to(gpu)
    out = ops.layernorm_f8(x, gamma, beta, eps=1e-6)
    ref = F.layer_norm(x, (C,), gamma, beta, eps=1e-6)
    hi, a8, b8 = ops.unpack_f8(out, parts=True)
    assert relerr(hi, ref) < 8e-4
    assert torch.equal(a8, ops._e4m3(hi))
    assert relerr(ops.unpack_f8(out), ref) < 4e-5


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,W,Cin,Cout,k,stride,pad', [(2, 12, 16, 128, 256, 3, 1, 1), (1, 24, 32, 256, 128, 3, 2, 1), (3, 6, 8, 256, 96, 1, 1, 0),
                                                       (1, 7, 5, 64, 128, 3, 1, 1), (2, 21, 32, 128, 128, 3, 2, 1)])
def test_conv2d_nhwc(gpu, dtype, B, H, W, Cin, Cout, k, stride, pad):
    from dust3r_amd import ops
    if dtype == torch.float32 and Cin % 32:
        pytest.skip('Cin must be a multiple of the K tile')
    g = torch.Generator(device='cpu').manual_seed(B * 100 + Cin + Cout)
    x = torch.randn((B, H, W, Cin), generator=g).to(gpu).to(dtype)
    w = (torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(Cin * k * k)).to(gpu).to(dtype)
    b = torch.randn(Cout, generator=g).to(gpu)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=stride, padding=pad).permute(0, 2, 3, 1)
    out = ops.conv2d_nhwc(x, w, b, stride=stride, pad=pad)
    assert out.shape == ref.shape and relerr(out, ref) < OUT_TOL[dtype]
    r1 = torch.randn(ref.shape, generator=g).to(gpu).to(dtype)
    r2 = torch.randn(ref.shape, generator=g).to(gpu).to(dtype)
    out, out_relu = ops.conv2d_nhwc(x, w, b, stride=stride, pad=pad, res1=r1, res2=r2, relu_copy=True)
    full = ref + r1.float() + r2.float()
    assert relerr(out, full) < OUT_TOL[dtype] and relerr(out_relu, full.clamp_min(0)) < OUT_TOL[dtype]
    out = ops.conv2d_nhwc(x, w, None, stride=stride, pad=pad, relu=True)
    assert relerr(out, (ref - b).clamp_min(0)) < OUT_TOL[dtype]


@pytest.mark.parametrize('cfg', probe_arms(['0', '1', '2', '3', '4', '5', '6'], ['0', '1', '2', '3']))     # 4-6: probe-only tile shapes (W4, four-stage, ping-pong)
def test_gemm_tile_configurations(gpu, cfg, monkeypatch):
    """Every tile configuration of the GEMM template (128x128, 256x256, 256x128) on shapes with ragged M / N edges,
    K long enough to cycle both LDS stages many times, linear and implicit-GEMM convolution operands."""
    from dust3r_amd import ops
    monkeypatch.setenv('D3R_GEMM_CFG', cfg)
    g = torch.Generator(device='cpu').manual_seed(17)
    for dtype in (torch.bfloat16, torch.float32):
        # the last shape has more tiles than resident block slots in every configuration: the persistent blocks walk
        # several tiles and prefetch the next tile's first K step across the epilogue
        for (M, N, K) in [(1000, 768, 1024), (515, 320, 256), (2048, 1024, 4096 if dtype == torch.bfloat16 else 512), (5137, 4096, 256)]:
            a = torch.randn((M, K), generator=g).to(gpu).to(dtype)
            w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu).to(dtype)
            b = torch.randn(N, generator=g).to(gpu)
            ref = a.float() @ w.float().T + b
            assert relerr(ops.linear(a, w, b, 'f32'), ref) < 2e-5, (cfg, dtype, M, N, K)
            assert relerr(ops.linear(a, w, b, 'gelu'), F.gelu(ref)) < OUT_TOL[dtype]
        x = torch.randn((2, 21, 32, 128), generator=g).to(gpu).to(dtype)
        w = (torch.randn((256, 128, 3, 3), generator=g) / math.sqrt(128 * 9)).to(gpu).to(dtype)
        b = torch.randn(256, generator=g).to(gpu)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=1, padding=1).permute(0, 2, 3, 1)
        assert relerr(ops.conv2d_nhwc(x, w, b, stride=1, pad=1), ref) < OUT_TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_wide_and_direct_epilogues_agree(gpu, dtype, monkeypatch):
    """The LDS-staged wide-row epilogue and the direct fragment stores are two routes for the same values: bit-identical."""
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(23)
    M, N, K = 1111, 840, 512
    a = torch.randn((M, K), generator=g).to(gpu).to(dtype)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(gpu).to(dtype)
    b = torch.randn(N, generator=g).to(gpu)
    res = torch.randn((M, N), generator=g).to(gpu)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('D3R_GEMM_NOWIDE', mode)
        outs[mode] = (ops.linear(a, w, b, 'store'), ops.linear(a, w, b, 'gelu'), ops.linear(a, w, b, 'f32', residual=res))
    for x, y in zip(outs['0'], outs['1']):
        assert torch.equal(x, y)


def _attention_ref(q, k, v, scale):
    a = (q.float() @ k.float().transpose(-1, -2)) * scale
    return (a.softmax(-1) @ v.float()).transpose(1, 2).flatten(2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 3, 196, 196), (1, 2, 768, 768), (2, 1, 6, 6), (1, 4, 130, 70), (1, 1, 768, 196)])
def test_attention(gpu, dtype, B, H, Nq, Nk):
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(Nq * 3 + Nk)
    q = (torch.randn((B, H, Nq, 64), generator=g) * 1.5).to(gpu).to(dtype)
    k = (torch.randn((B, H, Nk, 64), generator=g) * 1.5).to(gpu).to(dtype)
    v = torch.randn((B, H, Nk, 64), generator=g).to(gpu).to(dtype)
    ldv = (Nk + 63) // 64 * 64
    vt = torch.zeros((B, H, 64, ldv), dtype=dtype, device=gpu)
    vt[..., :Nk] = v.transpose(-1, -2)
    out = ops.attention(q, k, vt, Nk=Nk, scale=0.125)
    ref = _attention_ref(q, k, v, 0.125)
    # P is rounded to the 16-bit type before P V: tolerance is that rounding (not accumulated: P sums to 1)
    tol = {torch.float32: 3e-5, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}[dtype]
    assert relerr(out, ref) < tol


@pytest.mark.parametrize('v1', probe_arms(['0', '1', 'dma', 'reg', 'pk', 'lz'], ['0', '1', 'dma', 'reg']))
@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 3, 196, 196), (1, 2, 768, 768), (2, 1, 6, 6), (1, 4, 130, 70), (1, 1, 768, 196), (1, 2, 40, 129), (1, 1, 300, 64), (1, 1, 64, 128)])
def test_attention_split_fp16(gpu, v1, B, H, Nq, Nk, monkeypatch):
    """The split-fp16 attention of the default engine at kernel level, every kernel (D3R_ATTN_V1=1: the round-2 kernel; the
    software-pipelined one with its K / V^T tiles staged through registers, D3R_ATTN_DMA=0, or by global_load_lds DMA into swizzled
    256-byte rows, D3R_ATTN_DMA=1; '0' = whatever the default is): against the fp64 softmax(Q K^T / 8) V of the SAME fp32 operands. Operands
    keep 22 significand bits and the probabilities are split too, so the result is fp32-class (3e-5 like the exact-fp32 kernel); 1 to 12
    key tiles, ragged last tiles, query blocks with idle lanes. A subprocess-free switch: the choice is read on every launch."""
    from dust3r_amd import ops
    from conftest import need_probes
    if v1 in ('pk', 'lz'):
        need_probes('the packed-softmax / lazy-maximum attention instances')
    monkeypatch.setenv('D3R_ATTN_V1', '1' if v1 == '1' else '0')
    if v1 in ('dma', 'reg'):
        monkeypatch.setenv('D3R_ATTN_DMA', '1' if v1 == 'dma' else '0')
    if v1 == 'pk':            # the softmax / split slices on packed fp32 VALU (rounds 3-4; round 5's default is the scalar form, DMA staging)
        monkeypatch.setenv('D3R_ATTN_DMA', '1')
        monkeypatch.setenv('D3R_ATTN_SC', '0')
    monkeypatch.setenv('D3R_ATTN_LAZY', '1' if v1 == 'lz' else '0')     # lazy running maximum (round 5): the exponents' reference moves only by more than six octaves
    g = torch.Generator(device='cpu').manual_seed(Nq * 5 + Nk)
    q = (torch.randn((B, H, Nq, 64), generator=g) * 1.5).to(gpu)
    k = (torch.randn((B, H, Nk, 64), generator=g) * 1.5).to(gpu)
    v = torch.randn((B, H, Nk, 64), generator=g).to(gpu)
    out = ops.attention_x3(q, k, v, scale=0.125)
    a = (q.double() @ k.double().transpose(-1, -2)) * 0.125
    ref = (a.softmax(-1) @ v.double()).transpose(1, 2).flatten(2)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print(f'attention x3 v1={v1} B{B} H{H} Nq{Nq} Nk{Nk}: rel err {err:.2e}')
    assert err < 3e-5


@pytest.mark.parametrize('dma', probe_arms(['0', '1', 'lazy'], ['0', '1']))
def test_attention_split_fp16_sharp_rows_and_late_maximum(gpu, dma, monkeypatch):
    """Running-maximum rescale of the pipelined kernel: one key dominating by a huge margin in a LATE tile (alpha = 0 there), and rows
    whose maximum moves in every tile."""
    from dust3r_amd import ops
    from conftest import need_probes
    if dma == 'lazy':
        need_probes('the lazy-maximum attention instance')
    monkeypatch.setenv('D3R_ATTN_DMA', '1' if dma == 'lazy' else dma)
    monkeypatch.setenv('D3R_ATTN_LAZY', '1' if dma == 'lazy' else '0')     # head 1's maximum creeps up by 0.16 octaves per key: the lazy reference moves every ~38 keys only
    B, H, N = 1, 2, 320
    q = torch.zeros((B, H, N, 64), device=gpu)
    k = torch.zeros((B, H, N, 64), device=gpu)
    q[..., 0] = 30.0
    k[0, 0, 200, 0] = 30.0                                   # head 0: score 112.5 for key 200, 0 elsewhere
    k[0, 1, :, 0] = torch.linspace(0, 20, N, device=gpu)     # head 1: the maximum grows with the key index
    v = torch.randn((B, H, N, 64), device=gpu)
    out = ops.attention_x3(q, k, v, scale=0.125)
    a = (q.double() @ k.double().transpose(-1, -2)) * 0.125
    ref = (a.softmax(-1) @ v.double()).transpose(1, 2).flatten(2)
    assert float((out.double() - ref).abs().max()) < 1e-5


def test_attention_split_fp16_dma_staging_is_bit_identical(gpu, monkeypatch):
    """The DMA-staged kernel computes the same MFMAs on the same operands in the same order as the register-staged one: bit-identical
    outputs (encoder and decoder shapes of the BASELINE model, a ragged key count, a cross-attention shape with Nq != Nk)."""
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(77)
    for B, H, Nq, Nk in ((2, 16, 768, 768), (1, 12, 768, 768), (1, 3, 200, 333), (1, 2, 768, 196)):
        q = (torch.randn((B, H, Nq, 64), generator=g) * 1.5).to(gpu)
        k = (torch.randn((B, H, Nk, 64), generator=g) * 1.5).to(gpu)
        v = torch.randn((B, H, Nk, 64), generator=g).to(gpu)
        outs = []
        from conftest import probes_built
        # register staging (packed softmax slices), [probe builds: DMA staging + packed slices,] DMA staging + scalar-VALU softmax slices (round 5, the default)
        for dma, sc in ((('0', '0'), ('1', '0'), ('1', '1')) if probes_built() else (('0', '0'), ('1', '1'), ('1', '1'))):
            monkeypatch.setenv('D3R_ATTN_DMA', dma)
            monkeypatch.setenv('D3R_ATTN_SC', sc)
            outs.append(ops.attention_x3(q, k, v, scale=0.125).clone())
        assert torch.equal(outs[0], outs[1]), (B, H, Nq, Nk, float((outs[0] - outs[1]).abs().max()))
        assert torch.equal(outs[1], outs[2]), ('scalar VALU', B, H, Nq, Nk, float((outs[1] - outs[2]).abs().max()))


def test_attention_softmax_is_stable(gpu):
    """one key dominating by a huge margin at a late tile: the running-max rescale branch must be exact."""
    from dust3r_amd import ops
    B, H, N = 1, 1, 256
    q = torch.zeros((B, H, N, 64), device=gpu)
    k = torch.zeros((B, H, N, 64), device=gpu)
    q[..., 0] = 30.0
    k[0, 0, 200, 0] = 30.0           # score 900 * 0.125 for key 200, 0 elsewhere
    v = torch.randn((B, H, N, 64), device=gpu)
    out = ops.attention(q, k, v.transpose(-1, -2).contiguous(), scale=0.125)
    assert relerr(out[0, :, :], v[0, 0, 200].expand(N, 64)) < 1e-5


@pytest.mark.parametrize('dtype', DTYPES)
def test_rope2d_matches_reference_op(gpu, dtype):
    """d3r_rope2d vs the restated curope arithmetic (oracle) and vs croco's pure-torch RoPE2D module."""
    from dust3r_amd import ops
    from oracle.croco_ref.models.pos_embed import RoPE2D, rope_2d_inplace_ref
    B, N, H, D = 2, 35, 3, 64
    g = torch.Generator(device='cpu').manual_seed(5)
    tok = torch.randn((B, N, H, D), generator=g)
    pos = torch.stack((torch.randint(0, 24, (B, N), generator=g), torch.randint(0, 32, (B, N), generator=g)), dim=-1)
    ref = rope_2d_inplace_ref(tok.clone(), pos, 100.0, 1.0)
    ref2 = RoPE2D(freq=100.0)(tok.transpose(1, 2), pos).transpose(1, 2)
    assert relerr(ref, ref2) < 1e-5
    out = ops.rope_2d(tok.to(gpu).to(dtype).contiguous(), pos.to(gpu).contiguous(), 100.0, 1.0)
    ref16 = rope_2d_inplace_ref(tok.to(dtype).float(), pos, 100.0, 1.0)
    assert relerr(out.cpu(), ref16) < {torch.float32: 1e-5, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    # rotation => norm preserving per (y | x) half
    if dtype == torch.float32:
        assert torch.allclose(out.cpu().norm(dim=-1), tok.norm(dim=-1), rtol=1e-5)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,W,C,crop', [(2, 12, 16, 256, None), (1, 11, 16, 256, (21, 32)), (1, 5, 3, 128, None)])
def test_upsample2x(gpu, dtype, B, H, W, C, crop):
    from dust3r_amd import ops
    g = torch.Generator(device='cpu').manual_seed(H * W)
    x = torch.randn((B, H, W, C), generator=g).to(gpu).to(dtype)
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    if crop:
        ref = ref[:, :crop[0], :crop[1]]
    out = ops.upsample2x_nhwc(x, crop)
    assert out.shape == ref.shape and relerr(out, ref) < OUT_TOL[dtype]


@pytest.mark.parametrize('v1', ['0', '1'])
@pytest.mark.parametrize('B,H,W,C,crop', [(2, 12, 16, 256, None), (1, 24, 32, 128, None), (2, 7, 5, 64, None), (1, 12, 16, 256, (24, 31)), (1, 9, 11, 8, (17, 22)),
                                          (1, 1, 6, 16, None), (3, 6, 1, 16, None), (5, 13, 4, 8, None)])
def test_upsample2x_split_fp16(gpu, v1, B, H, W, C, crop, monkeypatch):
    """x2 bilinear (align_corners=True, crop like dpt_head.py:57) on split-fp16 maps, both kernels (D3R_UPSAMPLE_V1=1: one output pixel per
    thread; default: a 2 x 2 output block per thread from its 3 x 3 input neighbourhood): fp32-class against torch, odd / cropped output
    sizes (blocks with a missing second row or column), single-row and single-column inputs."""
    from dust3r_amd import ops
    monkeypatch.setenv('D3R_UPSAMPLE_V1', v1)
    g = torch.Generator(device='cpu').manual_seed(H * W + C)
    x = torch.randn((B, H, W, C), generator=g).to(gpu)
    ref = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    if crop:
        ref = ref[:, :crop[0], :crop[1]]
    out = ops.upsample2x_x3(x, crop)
    assert out.shape == ref.shape and relerr(out, ref) < 2e-6


def test_find_reciprocal_matches_matches_kdtree(gpu):
    """dust3r_amd.utils.geometry.find_reciprocal_matches (exhaustive GPU scan) vs the reference's SciPy KD-tree formulation."""
    import numpy as np
    from scipy.spatial import KDTree
    from dust3r_amd.utils.geometry import find_reciprocal_matches
    rng = np.random.RandomState(0)
    P1 = rng.randn(5000, 3).astype(np.float32)
    P2 = (P1[rng.permutation(5000)[:4000]] + 0.01 * rng.randn(4000, 3)).astype(np.float32)
    rec, nn2, cnt = find_reciprocal_matches(P1, P2)
    _, nn1_ref = KDTree(P2).query(P1)
    _, nn2_ref = KDTree(P1).query(P2)
    rec_ref = nn1_ref[nn2_ref] == np.arange(len(nn2_ref))
    assert rec.dtype == bool and rec.shape == (4000,) and cnt == int(rec.sum())
    assert (nn2 == nn2_ref).mean() > 0.9999 and (rec == rec_ref).mean() > 0.9999 and abs(cnt - int(rec_ref.sum())) <= 2
