"""oracle/f8_ref.py -- TEST INFRASTRUCTURE ONLY.

What the engine's fp16 + fp8 contraction (D3R_DTYPE_F16F8, csrc/common.hpp Traits<D3R_F16F8>, csrc/gemm.hip) computes, restated
independently of the product's host packers (dust3r_amd/ops.py) and evaluated in fp64:

    x . w  ~=  hi_x . hi_w  +  ( e4m3(hi_x) . e4m3(lo_w 2^17)  +  e4m3(lo_x 2^11) . e4m3(hi_w 2^6) ) 2^-17
    hi = fp16(v) (inputs clamped to the fp16 range), lo = v - hi, e4m3 = OCP e4m3fn, round to nearest even, saturating at +-448

There is no reference code for this scheme (the reference computes in fp32, dust3r/inference.py:44): this file pins the ARITHMETIC
of the engine's mode -- the GPU tests hold the kernel to it at fp32-accumulation noise -- and tools/precision_fp8cross.py measures
what the scheme costs against the reference's fp32 forward. The e4m3 encoding itself is pinned against the hardware's
v_cvt_pk_fp8_f32 by tools/f8_probe.hip (profiles/r02_f8/f8_probe.log).
"""
import torch

LO_SHIFT = 2.0 ** 11        # activations: lo parts are scaled into e4m3's range before the conversion
W_SHIFT = 2.0 ** 6          # weights: |w| of O(0.01) sits in e4m3's normal range after this shift


def e4m3(x):
    """fp64/fp32 tensor -> the value of its OCP e4m3 rounding (saturating)."""
    return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).double()


def split(v):
    v = v.float().clamp(-65504.0, 65504.0)
    hi = v.half()
    return hi.double(), (v - hi.float()).double()


def f16f8_matmul(act, weight):
    """act (M, K), weight (N, K) fp32 -> (M, N) fp64: the engine's fp16 + fp8 evaluation of act @ weight.T."""
    xh, xl = split(act)
    wh, wl = split(weight)
    cross = e4m3(xh) @ e4m3(wl * LO_SHIFT * W_SHIFT).T + e4m3(xl * LO_SHIFT) @ e4m3(wh * W_SHIFT).T
    return xh @ wh.T + cross / (LO_SHIFT * W_SHIFT)


def f16x2f8_matmul(act, weight):
    """The 2.5-unit evaluation (D3R_DTYPE_F16X2F8, csrc/common.hpp Traits<D3R_F16X2F8>): the weights keep both fp16 halves on the f16 MFMA, only the
    activations' lo halves go through e4m3:
        x . w  ~=  hi_x . hi_w  +  hi_x . fp16(lo_w)  +  e4m3(lo_x 2^11) . e4m3(hi_w 2^6) 2^-17"""
    xh, xl = split(act)
    wh, wl = split(weight)
    wl16 = wl.float().half().double()
    return xh @ wh.T + xh @ wl16.T + (e4m3(xl * LO_SHIFT) @ e4m3(wh * W_SHIFT).T) / (LO_SHIFT * W_SHIFT)
