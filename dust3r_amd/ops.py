"""Thin torch-tensor wrappers over the building-block entry points of libdust3r_hip.so
(d3r_rope2d / d3r_layernorm / d3r_linear / d3r_conv2d_nhwc / d3r_attention / d3r_upsample2x_nhwc).

`rope_2d` is the drop-in for the reference's only native op, croco's `curope.rope_2d(tokens,
positions, base, F0)` (in place; see include/dust3r_hip.h). The other wrappers exist so the parity
tests can pin each kernel against PyTorch; the model engine calls the kernels from C++ directly.
"""
import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr


def _dt(t):
    return {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16, torch.float32: _lib.DTYPE_F32}[t.dtype]


def rope_2d(tokens, positions, base, F0=1.0):
    """In-place 2-D RoPE on tokens (B, N, H, D) contiguous; positions (B, N, 2) int64 (y, x)."""
    _lib.require_device()
    assert tokens.is_cuda and tokens.is_contiguous() and tokens.ndim == 4, 'tokens must be a contiguous CUDA (B,N,H,D) tensor'
    B, N, H, D = tokens.shape
    assert D % 4 == 0, 'D must be a multiple of 4'
    assert positions.shape == (B, N, 2) and positions.dtype == torch.int64 and positions.is_contiguous()
    check(lib.d3r_rope2d(ptr(tokens), ptr(positions), B, N, H, D, float(base), float(F0), _dt(tokens), current_stream()), 'rope2d')
    return tokens


def layernorm(x, gamma, beta, eps=1e-6, dtype=torch.bfloat16):
    _lib.require_device()
    rows, Cc = x.shape
    out = torch.empty((rows, Cc), dtype=dtype, device=x.device)
    check(lib.d3r_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(out), rows, Cc, eps, _dt(out), current_stream()), 'layernorm')
    return out


def layernorm_x3(x, gamma, beta, eps=1e-6):
    """LayerNorm into split-fp16 rows (the default engine's operand layout), returned unpacked as fp32 = hi + lo."""
    _lib.require_device()
    rows, Cc = x.shape
    out = torch.empty((rows, 2 * Cc), dtype=torch.float16, device=x.device)
    check(lib.d3r_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(out), rows, Cc, eps, _lib.DTYPE_F16X3, current_stream()), 'layernorm(x3)')
    return unpack_x3(out)


def pad_rows(w, mult=256):
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return w.contiguous()
    return torch.cat((w, w.new_zeros((n_pad - n,) + tuple(w.shape[1:]))), dim=0).contiguous()


def linear(act, weight, bias=None, epilogue='store', residual=None):
    """act (M,K), weight (N,K) in the same 16/32-bit dtype -> (M,N). epilogue: 'store' | 'f32' | 'gelu'."""
    _lib.require_device()
    M, K = act.shape
    N = weight.shape[0]
    wp = pad_rows(weight)
    bp = None if bias is None else pad_rows(bias.float())
    epi = {'store': 0, 'f32': 1, 'gelu': 2}[epilogue]
    out = torch.empty((M, N), dtype=torch.float32 if epi == 1 else act.dtype, device=act.device)
    check(lib.d3r_linear(ptr(act), ptr(wp), ptr(bp), ptr(out), ptr(residual), M, N, K, epi, _dt(act), current_stream()), 'linear')
    return out


def pack_x3(x):
    """fp32 (..., K) with K % 8 == 0 -> the engine's split-fp16 row layout, an fp16 tensor (..., 2K): per 8 logical elements
    a 32-byte group [hi x8][lo x8], hi = fp16(x), lo = fp16(x - hi) (csrc/common.hpp, Traits<D3R_F16X3>)."""
    x = x.float()
    hi = x.half()
    lo = (x - hi.float()).half()
    sh = x.shape
    g = torch.stack((hi.reshape(*sh[:-1], sh[-1] // 8, 8), lo.reshape(*sh[:-1], sh[-1] // 8, 8)), dim=-2)
    return g.reshape(*sh[:-1], sh[-1] * 2).contiguous()


def unpack_x3(t):
    """Inverse of pack_x3: fp16 (..., 2K) -> fp32 (..., K) = hi + lo."""
    sh = t.shape
    g = t.reshape(*sh[:-1], sh[-1] // 16, 2, 8).float()
    return (g[..., 0, :] + g[..., 1, :]).reshape(*sh[:-1], sh[-1] // 2)


def linear_x3res(act, weight, bias=None, residual=None, with_sums=True):
    """nn.Linear as the producer of a folded LayerNorm (d3r_linear_x3res, split-fp16): returns (rows fp32 = act . weight^T + bias + residual, rounded to
    split-fp16 like the engine's residual stream, and -- with_sums -- the (sum, sum of squares) pairs [M][N / 32][2] of the stored rows)."""
    _lib.require_device()
    M, K = act.shape
    N = weight.shape[0]
    ap, wp = pack_x3(act), pad_rows(pack_x3(weight))
    bp = None if bias is None else pad_rows(bias.float())
    rp = None if residual is None else pack_x3(residual)
    out = torch.empty((M, 2 * N), dtype=torch.float16, device=act.device)
    part = torch.zeros((M, N // 32, 2), dtype=torch.float32, device=act.device) if with_sums else None
    check(lib.d3r_linear_x3res(ptr(ap), ptr(wp), ptr(bp), ptr(out), ptr(rp), ptr(part), M, N, K, current_stream()), 'linear_x3res')
    return unpack_x3(out), part, out


def linear_x3(act, weight, bias=None, epilogue='store', residual=None):
    """`linear` in the split-fp16 precision mode: act (M,K), weight (N,K) fp32 (N, K multiples of 8) are packed to the
    x3 layout, the result comes back as fp32 (unpacked for the 'store' / 'gelu' epilogues)."""
    _lib.require_device()
    M, K = act.shape
    N = weight.shape[0]
    ap, wp = pack_x3(act), pad_rows(pack_x3(weight))
    bp = None if bias is None else pad_rows(bias.float())
    epi = {'store': 0, 'f32': 1, 'gelu': 2}[epilogue]
    out = torch.empty((M, N), dtype=torch.float32, device=act.device) if epi == 1 else torch.empty((M, 2 * N), dtype=torch.float16, device=act.device)
    check(lib.d3r_linear(ptr(ap), ptr(wp), ptr(bp), ptr(out), ptr(residual), M, N, K, epi, _lib.DTYPE_F16X3, current_stream()), 'linear(x3)')
    return out if epi == 1 else unpack_x3(out)


def upsample2x_x3(x, out_hw=None):
    """`upsample2x_nhwc` on split-fp16 maps: x (B,H,W,C) fp32, C % 8 == 0, packed along the channel axis; fp32 result."""
    _lib.require_device()
    B, H, W, Cc = x.shape
    Ho, Wo = out_hw if out_hw is not None else (2 * H, 2 * W)
    xp = pack_x3(x)
    out = torch.empty((B, Ho, Wo, 2 * Cc), dtype=torch.float16, device=x.device)
    check(lib.d3r_upsample2x_nhwc(ptr(xp), ptr(out), B, H, W, Cc, Ho, Wo, _lib.DTYPE_F16X3, current_stream()), 'upsample2x(x3)')
    return unpack_x3(out)


def _e4m3(x):
    """Round to OCP e4m3 (the encoding v_cvt_pk_fp8_f32 produces on gfx950: nearest even, subnormals kept), saturating at +-448
    as the kernels clamp before converting. Returns the bytes (uint8)."""
    return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def pack_f8(x, weight=False):
    """fp32 (..., K), K % 64 == 0 -> the fp16 + fp8 row layout as uint8 (..., 4K): per 64 logical elements a 256-byte super-group
    [hi fp16 x64 | a8 e4m3 x64 | b8 e4m3 x64]; activations a8 = e4m3(hi), b8 = e4m3(lo 2^11); weights a8 = e4m3(lo 2^17),
    b8 = e4m3(hi 2^6)  (hi = fp16(x), lo = x - hi; csrc/common.hpp, Traits<D3R_F16F8>)."""
    x = x.float().clamp(-65504.0, 65504.0)
    sh = x.shape
    assert sh[-1] % 64 == 0
    hi = x.half()
    lo = x - hi.float()
    h8 = _e4m3(hi.float() * (64.0 if weight else 1.0))
    l8 = _e4m3(lo * (131072.0 if weight else 2048.0))
    a8, b8 = (l8, h8) if weight else (h8, l8)
    g = lambda t: t.reshape(*sh[:-1], sh[-1] // 64, -1)
    out = torch.cat((g(hi.view(torch.uint8)), g(a8), g(b8)), dim=-1)
    return out.reshape(*sh[:-1], sh[-1] * 4).contiguous()


def unpack_f8(t, weight=False, parts=False):
    """Inverse of pack_f8 (uint8 (..., 4K) -> fp32 (..., K)): the value an element stands for, hi + lo8 2^-11 (activations) or
    hi + lo8 2^-17 (weights). parts=True returns (hi fp32, a8 bytes, b8 bytes)."""
    sh = t.shape
    g = t.reshape(*sh[:-1], sh[-1] // 256, 256)
    hi = g[..., :128].contiguous().view(torch.float16).float().reshape(*sh[:-1], sh[-1] // 4)
    a8 = g[..., 128:192].reshape(*sh[:-1], sh[-1] // 4)
    b8 = g[..., 192:].reshape(*sh[:-1], sh[-1] // 4)
    if parts:
        return hi, a8, b8
    lo8 = (a8 if weight else b8).contiguous().view(torch.float8_e4m3fn).float()
    return hi + lo8 / (131072.0 if weight else 2048.0)


def linear_f8(act, weight, bias=None, epilogue='store', residual=None):
    """`linear` in the fp16 + fp8 mode: act (M,K), weight (N,K) fp32 (K % 64 == 0; N % 64 == 0 for 'store' / 'gelu') are packed to the
    activation / weight row layouts; 'f32' returns the fp32 result, the others the decoded activation rows."""
    _lib.require_device()
    M, K = act.shape
    N = weight.shape[0]
    ap, wp = pack_f8(act), pad_rows(pack_f8(weight, weight=True))
    bp = None if bias is None else pad_rows(bias.float())
    epi = {'store': 0, 'f32': 1, 'gelu': 2}[epilogue]
    out = torch.empty((M, N), dtype=torch.float32, device=act.device) if epi == 1 else torch.empty((M, 4 * N), dtype=torch.uint8, device=act.device)
    check(lib.d3r_linear(ptr(ap), ptr(wp), ptr(bp), ptr(out), ptr(residual), M, N, K, epi, _lib.DTYPE_F16F8, current_stream()), 'linear(f16f8)')
    return out if epi == 1 else unpack_f8(out)


def pack_w5(w):
    """fp32 weights (N, K), K % 128 == 0 -> the 2.5-unit weight rows as uint8 (N, 5K): per 128 logical k five 128-byte chunks
    [w_hi k 0..63 fp16 | w_lo k 0..63 fp16 | w_hi k 64..127 | w_lo k 64..127 | e4m3(w_hi 2^6) k 0..127]  (csrc/common.hpp, Traits<D3R_F16X2F8>)."""
    w = w.float().clamp(-65504.0, 65504.0)
    N, K = w.shape
    assert K % 128 == 0
    hi = w.half()
    lo = (w - hi.float()).half()
    h8 = _e4m3(hi.float() * 64.0)
    hb, lb = hi.view(torch.uint8).reshape(N, K // 128, 2, 128), lo.view(torch.uint8).reshape(N, K // 128, 2, 128)
    out = torch.cat((hb[:, :, 0], lb[:, :, 0], hb[:, :, 1], lb[:, :, 1], h8.reshape(N, K // 128, 128)), dim=-1)
    return out.reshape(N, 5 * K).contiguous()


def linear_x2f8(act, weight, bias=None, epilogue='store', residual=None):
    """`linear` in the 2.5-unit mode (D3R_DTYPE_F16X2F8): fp16 + fp8 activation rows x five-chunk weight rows; K % 128 == 0."""
    _lib.require_device()
    M, K = act.shape
    N = weight.shape[0]
    ap, wp = pack_f8(act), pad_rows(pack_w5(weight))
    bp = None if bias is None else pad_rows(bias.float())
    epi = {'store': 0, 'f32': 1, 'gelu': 2}[epilogue]
    out = torch.empty((M, N), dtype=torch.float32, device=act.device) if epi == 1 else torch.empty((M, 4 * N), dtype=torch.uint8, device=act.device)
    check(lib.d3r_linear(ptr(ap), ptr(wp), ptr(bp), ptr(out), ptr(residual), M, N, K, epi, _lib.DTYPE_F16X2F8, current_stream()), 'linear(f16x2f8)')
    return out if epi == 1 else unpack_f8(out)


def layernorm_f8(x, gamma, beta, eps=1e-6):
    """LayerNorm into fp16 + fp8 activation rows (uint8 (rows, 4C))."""
    _lib.require_device()
    rows, Cc = x.shape
    out = torch.empty((rows, 4 * Cc), dtype=torch.uint8, device=x.device)
    check(lib.d3r_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(out), rows, Cc, eps, _lib.DTYPE_F16F8, current_stream()), 'layernorm(f16f8)')
    return out


def pack_conv_weight(w, dtype=None):
    """torch (Cout, Cin, kh, kw) -> (round_up(Cout,256), kh*kw*Cin) in the K order d3r_conv2d_nhwc expects (include/dust3r_hip.h):
    channel slices of one K step (128 bytes) outermost, then the taps, then the channels of the slice -- or (ky, kx, cin) when the
    library runs with D3R_CONV_KORDER=0."""
    Cout, Cin, kh, kw = w.shape
    if lib.d3r_conv_k_slice_major():
        S = 128 // torch.empty((), dtype=dtype or w.dtype).element_size()
        assert Cin % S == 0, f'{Cin=} must be a multiple of {S}'
        return pad_rows(w.reshape(Cout, Cin // S, S, kh * kw).permute(0, 1, 3, 2).reshape(Cout, -1))
    return pad_rows(w.permute(0, 2, 3, 1).reshape(Cout, -1))


_zero_pages = {}


def _zero_page(device):
    if device not in _zero_pages:
        _zero_pages[device] = torch.zeros(4096, dtype=torch.uint8, device=device)
    return _zero_pages[device]


def conv2d_nhwc(x, weight, bias=None, stride=1, pad=0, relu=False, res1=None, res2=None, relu_copy=False):
    """x (B,H,W,Cin) NHWC; weight torch layout (Cout,Cin,k,k); returns (B,Ho,Wo,Cout) [, relu copy]."""
    _lib.require_device()
    B, H, W, Cin = x.shape
    Cout, _, k, _ = weight.shape
    wp = pack_conv_weight(weight.to(x.dtype))
    bp = None if bias is None else pad_rows(bias.float())
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    out2 = torch.empty_like(out) if relu_copy else None
    check(lib.d3r_conv2d_nhwc(ptr(x), ptr(wp), ptr(bp), ptr(out), ptr(res1), ptr(res2), ptr(out2), B, H, W, Cin, Cout, k, stride,
                              pad, int(relu), ptr(_zero_page(x.device)), _dt(x), current_stream()), 'conv2d_nhwc')
    return (out, out2) if relu_copy else out


def attention(q, k, vt, Nk=None, scale=0.125):
    """q (B,H,Nq,64), k (B,H,Nk,64), vt (B,H,64,ldv) with ldv % 64 == 0 and zero padding -> (B,Nq,H*64)."""
    _lib.require_device()
    B, H, Nq, D = q.shape
    assert D == 64
    Nk = k.shape[2] if Nk is None else Nk
    ldv = vt.shape[3]
    out = torch.empty((B, Nq, H * 64), dtype=q.dtype, device=q.device)
    check(lib.d3r_attention(ptr(q), ptr(k), ptr(vt), ptr(out), B, H, Nq, Nk, ldv, scale, _dt(q), current_stream()), 'attention')
    return out


def attention_x3(q, k, v, scale=0.125):
    """`attention` in the split-fp16 precision mode: q (B,H,Nq,64), k, v (B,H,Nk,64) fp32 are packed to the x3 row layout (q, k per
    token; v transposed to (B,H,64,ldv) with the keys zero padded to a multiple of 64), the (B,Nq,H*64) result comes back as fp32."""
    _lib.require_device()
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    assert D == 64
    ldv = (Nk + 63) // 64 * 64
    vt = torch.zeros((B, H, 64, ldv), dtype=torch.float32, device=q.device)
    vt[..., :Nk] = v.float().transpose(-1, -2)
    qp, kp, vp = pack_x3(q), pack_x3(k), pack_x3(vt)
    out = torch.empty((B, Nq, H * 64 * 2), dtype=torch.float16, device=q.device)
    check(lib.d3r_attention(ptr(qp), ptr(kp), ptr(vp), ptr(out), B, H, Nq, Nk, ldv, scale, _lib.DTYPE_F16X3, current_stream()), 'attention(x3)')
    return unpack_x3(out)


def upsample2x_nhwc(x, out_hw=None):
    _lib.require_device()
    B, H, W, Cc = x.shape
    Ho, Wo = out_hw if out_hw is not None else (2 * H, 2 * W)
    out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    check(lib.d3r_upsample2x_nhwc(ptr(x), ptr(out), B, H, W, Cc, Ho, Wo, _dt(x), current_stream()), 'upsample2x')
    return out
