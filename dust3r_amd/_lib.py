"""ctypes binding of libdust3r_hip.so (C ABI declared in include/dust3r_hip.h).

The product path has NO CPU fallback: importing this module without the built library raises,
and every compute entry point checks for a gfx950 device (`require_device`) before touching it.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so that the engine shares its HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libdust3r_hip.so')

DTYPE_BF16, DTYPE_F16, DTYPE_F32, DTYPE_F16X3, DTYPE_F16F8, DTYPE_F16X2F8 = 0, 1, 2, 3, 4, 5
DTYPES = {'bf16': DTYPE_BF16, 'bfloat16': DTYPE_BF16, 'f16': DTYPE_F16, 'fp16': DTYPE_F16, 'float16': DTYPE_F16,
          'f32': DTYPE_F32, 'fp32': DTYPE_F32, 'float32': DTYPE_F32, 'fp16x3': DTYPE_F16X3, 'f16x3': DTYPE_F16X3,
          'fp16f8': DTYPE_F16F8, 'f16f8': DTYPE_F16F8, 'fp16x2f8': DTYPE_F16X2F8}
TORCH_DTYPE = {DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16, DTYPE_F32: torch.float32}

ERRORS = {0: 'OK', -1: 'invalid argument', -2: 'allocation failed', -3: 'kernel launch failed', -4: 'unknown state-dict key',
          -5: 'shape mismatch', -6: 'bad state (weights missing / no gfx950 device)'}


class D3RError(RuntimeError):
    pass


class ModelConfig(C.Structure):
    _fields_ = [('enc_embed_dim', C.c_int), ('enc_depth', C.c_int), ('enc_num_heads', C.c_int),
                ('dec_embed_dim', C.c_int), ('dec_depth', C.c_int), ('dec_num_heads', C.c_int),
                ('patch_size', C.c_int), ('head_type', C.c_int), ('dtype', C.c_int), ('rope_freq', C.c_float),
                ('dpt_skip_relu_inplace', C.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(or `python dust3r_amd/build.py`). dust3r_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    vp, i, f, fp, ip = C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p
    sig = {
        'd3r_version': (C.c_char_p, []),
        'd3r_device_check': (i, []),
        'd3r_rope2d': (i, [vp, vp, i, i, i, i, f, f, i, vp]),
        'd3r_layernorm': (i, [fp, fp, fp, vp, i, i, f, i, vp]),
        'd3r_linear': (i, [vp, vp, fp, vp, fp, i, i, i, i, i, vp]),
        'd3r_linear_x3res': (i, [vp, vp, fp, vp, vp, fp, i, i, i, vp]),
        'd3r_conv2d_nhwc': (i, [vp, vp, fp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, i, vp]),
        'd3r_conv_k_slice_major': (i, []),
        'd3r_attention': (i, [vp, vp, vp, vp, i, i, i, i, i, f, i, vp]),
        'd3r_upsample2x_nhwc': (i, [vp, vp, i, i, i, i, i, i, i, vp]),
        'd3r_gemm_set_trace': (i, [vp, C.c_size_t]),
        'd3r_gemm_tile_config': (i, [i, i, i, i, i, i]),
        'd3r_build_has_probes': (i, []),
        'd3r_model_create': (i, [C.POINTER(vp), C.POINTER(ModelConfig)]),
        'd3r_model_destroy': (i, [vp]),
        'd3r_model_load_tensor': (i, [vp, C.c_char_p, fp, i, C.POINTER(C.c_int64)]),
        'd3r_model_load_tensor_device': (i, [vp, C.c_char_p, fp, i, C.POINTER(C.c_int64)]),
        'd3r_model_missing': (i, [vp]),
        'd3r_model_forward': (i, [vp, fp, fp, i, i, i, fp, fp, fp, fp, vp]),
        'd3r_model_forward_mixed': (i, [vp, fp, i, i, fp, i, i, i, fp, fp, fp, fp, vp]),
        'd3r_model_forward_packed': (i, [vp, fp, fp, i, i, i, fp, vp]),
        'd3r_model_device_bytes': (C.c_size_t, [vp]),
        'd3r_model_graph_replays': (C.c_long, [vp]),
        'd3r_model_feature_bytes': (C.c_size_t, [vp, i, i]),
        'd3r_model_encode': (i, [vp, fp, i, i, i, vp, vp]),
        'd3r_model_decode': (i, [vp, vp, i, i, i, fp, fp, fp, fp, vp]),
        'd3r_model_decode_packed': (i, [vp, vp, i, i, i, fp, vp]),
        'd3r_model_debug_read': (i, [vp, i, fp, C.c_size_t, vp]),
        'd3r_model_set_option': (i, [vp, i, i]),
        'd3r_model_set_postprocess': (i, [vp, i, i, f, f]),
        'd3r_model_profile_read': (i, [vp, i, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        'd3r_model_profile_launch': (i, [vp, i] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double)] * 2),
        'd3r_aligner_create': (i, [C.POINTER(vp), i, i, ip, ip, ip, ip, i, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, f, f, f,
                                   i, i, i, i, i, vp]),
        'd3r_aligner_destroy': (i, [vp]),
        'd3r_aligner_set_option': (i, [vp, i, i]),
        'd3r_aligner_run': (i, [vp, i, i, i, f, f, i, fp, vp]),
        'd3r_aligner_loss_grad': (i, [vp, fp, fp, fp, fp, fp, fp, fp, vp]),
        'd3r_aligner_set_image_range': (i, [vp, i, i]),
        'd3r_aligner_step_begin': (i, [vp, i, i, i, f, f, i, vp]),
        'd3r_aligner_step_end': (i, [vp, i, i, i, f, f, i, vp]),
        'd3r_aligner_reduced_sums': (i, [vp, C.POINTER(vp), C.POINTER(C.c_longlong)]),
        'd3r_aligner_read_losses': (i, [vp, i, fp, vp]),
        'd3r_nearest_neighbors': (i, [fp, i, fp, i, ip, vp]),
        'd3r_clean_pointcloud': (i, [i, fp, fp, fp, fp, fp, ip, ip, i, f, f, vp]),
        'd3r_row_means': (i, [fp, i, i, i, fp, vp]),
        'd3r_similarity_moments_workspace': (C.c_size_t, [i, i]),
        'd3r_similarity_moments': (i, [i, vp, vp, vp, ip, i, vp, vp, vp]),
        'd3r_weiszfeld_focals': (i, [i, vp, ip, ip, i, fp, vp]),
        'd3r_anchor_depth': (i, [i, vp, fp, ip, i, i, fp, vp]),
        'd3r_pnp_job_bytes': (i, []),
        'd3r_pnp_max_hypotheses': (i, []),
        'd3r_pnp_sum_count': (i, []),
        'd3r_pnp_workspace': (C.c_size_t, [i]),
        'd3r_pnp_score': (i, [i, vp, fp, i, f, ip, vp]),
        'd3r_pnp_sums': (i, [i, vp, fp, f, vp, vp, vp]),
        'd3r_selftest_aligner_math_host': (i, [i, i, ip, ip, i, i, fp, fp, fp, fp, fp, fp, fp, fp, f, f, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTED = _load()


def check(rc, what=''):
    if rc != 0:
        msg = ERRORS.get(rc, f'hipError {rc - 1000}' if rc >= 1000 else f'error {rc}')
        raise D3RError(f'libdust3r_hip: {what}: {msg}')


_device_ok = None


def require_device():
    """Raise unless a gfx950 GPU is usable. Called by every compute entry point of the package."""
    global _device_ok
    if _device_ok is None:
        _device_ok = torch.cuda.is_available() and lib.d3r_device_check() == 0
    if not _device_ok:
        raise D3RError('dust3r_amd needs an AMD gfx950 (MI355X) device; there is no CPU fallback in the product path')


def ptr(t):
    """Device (or host) address of a contiguous tensor, or NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return C.c_void_p(t.data_ptr())


def current_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
