"""GPU parity of the model engine (C ABI d3r_model_forward through dust3r_amd.model) against the CPU
fp32 oracle and the golden vectors generated from the reference.

Tolerances. The error measure is SURVEY.md 8(d)'s: per pixel, ||pts_hip - pts_ref||_2 / max(||pts_ref||_2, eps) with ONE eps for
every mode and test: eps = 1e-8, i.e. no floor -- the strict per-pixel ratio.
  fp32 mode : max  <= 1e-3  -- the north-star bar; the engine's fp32-MFMA path is held to it strictly
  fp16x3    : max  <= 1e-3  -- THE DEFAULT ENGINE (3 f16 MFMAs per product, 22-bit operands): held to the same bar, same assertions
  fp16f8 / fp16 / bf16 : opt-in fast modes, NOT claimed to meet the bar: bounded at their measured rounding floor (99th percentile
              and mean; the per-pixel max is heavy-tailed at pixels whose pointmap norm is near zero) and reported in DESIGN.md
"""
import os

import numpy as np

import pytest
from conftest import probe_arms
import torch

from dust3r_amd.synthetic import MODEL_CONFIGS, synthetic_views

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pix_rel(a, b):
    e = (a.float().cpu() - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-8)
    return float(e.max()), float(e.mean())


def pix_rel_p99(a, b):
    e = ((a.float().cpu() - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-8)).flatten()
    return float(e.kthvalue(max(1, int(0.99 * e.numel())))[0])


def engine_from_oracle(oracle, config, precision, gpu):
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    cfg = MODEL_CONFIGS[config] if isinstance(config, str) else config
    m = AsymmetricCroCo3DStereo(precision=precision, landscape_only=False, **cfg)
    m.load_state_dict(oracle.state_dict(), strict=True)
    return m.to(gpu)


# precision -> (bound on the statistic, bound on the mean, which statistic: 'max' over ALL pixels or the 99th percentile)
#   fp32 / fp16x3: the north-star bar, max over ALL pixels <= 1e-3 (fp16x3 is the default engine)
#   fp16f8       : opt-in (the transformer blocks' cross terms on the e4m3 MFMA, ~2^-16 per operand): 99th percentile <= 1e-3, mean <= 2e-4
#                  (measured p99 1-2e-4, mean 3-4e-5, per-pixel max up to 1.8e-3 on these tiny networks -- which is why it is NOT the default)
#   fp16 / bf16  : single-pass 16-bit operands cannot meet 1e-3 (unit roundoff 4.9e-4 / 3.9e-3 per operand, 36 blocks deep,
#                  expm1 at the end); they are bounded at the rounding floor of the network instead: mean and 99th percentile
TOLS = {'fp32': (1e-3, 2e-4, 'max'), 'fp16x3': (1e-3, 2e-4, 'max'), 'fp16x2f8': (1e-3, 2e-4, 'max'), 'fp16f8': (1e-3, 2e-4, 'p99'), 'fp16': (5e-2, 8e-3, 'p99'), 'bf16': (3e-1, 5e-2, 'p99')}


def compare(engine, oracle, v1, v2, max_tol, mean_tol, stat='max', tag=''):
    with torch.no_grad():
        r1, r2 = oracle(v1, v2)
    e1, e2 = engine({k: v for k, v in v1.items()}, {k: v for k, v in v2.items()})
    torch.cuda.synchronize()
    assert e1['pts3d'].dtype == torch.float32 and e1['pts3d'].shape == r1['pts3d'].shape and e2['conf'].shape == r2['conf'].shape
    for name, a, b in (('pts1', e1['pts3d'], r1['pts3d']), ('pts2', e2['pts3d_in_other_view'], r2['pts3d_in_other_view'])):
        mx, mean = pix_rel(a, b)
        p99 = pix_rel_p99(a, b)
        print(f'[{tag}] {name}: rel err max {mx:.3e} p99 {p99:.3e} mean {mean:.3e}')
        assert (mx if stat == 'max' else p99) < max_tol and mean < mean_tol, (name, mx, p99, mean)
    for name, a, b in (('conf1', e1['conf'], r1['conf']), ('conf2', e2['conf'], r2['conf'])):
        e = ((a.cpu() - b).abs() / b.abs()).flatten()
        err = float(e.max()) if stat == 'max' else float(e.kthvalue(max(1, int(0.99 * e.numel())))[0])
        print(f'[{tag}] {name}: rel err {stat} {err:.3e}')
        assert err < max_tol * (1 if any(t in ('fp32', 'fp16x3') for t in tag.split()) else 3), (name, err)      # the parity-grade modes: the stated bar, also on conf


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'fp16x2f8', 'fp16f8', 'fp16', 'bf16'])
@pytest.mark.parametrize('config,B,H,W', [('tiny_dpt', 2, 32, 48), ('tiny_dpt', 1, 64, 64), ('tiny_dpt', 1, 48, 80), ('tiny_linear', 3, 32, 32),
                                          ('tiny_linear', 1, 224, 224)])
def test_forward_matches_oracle(gpu, precision, config, B, H, W):
    from oracle.dust3r_ref import build_ref_model
    oracle = build_ref_model(config)
    eng = engine_from_oracle(oracle, config, precision, gpu)
    v1, v2 = synthetic_views(B, H, W, seed=H + W)
    compare(eng, oracle, v1, v2, *TOLS[precision], tag=f'{config} {precision} {B}x{H}x{W}')


@pytest.mark.parametrize('cfg', probe_arms(['0', '1', '2', '3', '4', '5', '6', '7', '8', '9'], ['0', '1', '2', '3', '7', '8', '9']))     # 4-6: probe-only tile shapes
@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'fp16x2f8', 'fp16f8'])
def test_forward_with_pinned_gemm_tile(gpu, precision, cfg, monkeypatch):
    """The whole network with the GEMM tile configuration pinned (D3R_GEMM_CFG): the 256-wide tiles' q/k RoPE scatter,
    V^T role swap and implicit-GEMM paths must give the same pointmaps as the 128x128 tiles (all <= 1e-3 vs the oracle)."""
    from oracle.dust3r_ref import build_ref_model
    monkeypatch.setenv('D3R_GEMM_CFG', cfg)
    oracle = build_ref_model('tiny_dpt')
    eng = engine_from_oracle(oracle, 'tiny_dpt', precision, gpu)
    v1, v2 = synthetic_views(3, 64, 96, seed=5)
    compare(eng, oracle, v1, v2, *TOLS[precision], tag=f'tiny_dpt {precision} cfg{cfg}')


@pytest.mark.parametrize('cfg', probe_arms(['0', '1', '2', '3', '4', '5', '6'], ['0', '1', '2', '3']))
def test_forward_16bit_with_pinned_gemm_tile(gpu, cfg, monkeypatch):
    """fp16 engine, 128x128 images = 64 tokens: the wide epilogues incl. the LDS-transposed V^T scatter on every tile shape."""
    from oracle.dust3r_ref import build_ref_model
    monkeypatch.setenv('D3R_GEMM_CFG', cfg)
    oracle = build_ref_model('tiny_dpt')
    eng = engine_from_oracle(oracle, 'tiny_dpt', 'fp16', gpu)
    v1, v2 = synthetic_views(2, 128, 128, seed=6)
    compare(eng, oracle, v1, v2, *TOLS['fp16'], tag=f'tiny_dpt fp16 128x128 cfg{cfg}')


@pytest.mark.parametrize('cfg', ['0', '0w8', '1', '2', '3', '7', '8', '9', '11'])
def test_forward_split_fp16_kernel_variants(gpu, cfg, monkeypatch):
    """fp16x3 (the default, parity-grade mode) at 128x128 = 64 tokens, where the attention projections take the LDS-staged
    x3 epilogue (q / k RoPE scatter, operand-swapped V^T): every (software-pipelined | plain K loop) x (wide | direct epilogue)
    combination within 1e-3 of the oracle, and all of them within 1e-4 of each other (same MFMA order; RoPE / GELU may
    contract differently between the two epilogue routes)."""
    from oracle.dust3r_ref import build_ref_model
    monkeypatch.setenv('D3R_GEMM_CFG', cfg[:-2] if cfg.endswith(('w8', 'w4')) else cfg)
    monkeypatch.setenv('D3R_GEMM_T128W8', '1000000' if cfg.endswith('w8') else '0')     # '0w8': 128 x 128 by eight waves
    oracle = build_ref_model('tiny_dpt')
    eng = engine_from_oracle(oracle, 'tiny_dpt', 'fp16x3', gpu)
    v1, v2 = synthetic_views(2, 128, 128, seed=6)
    outs = []
    from conftest import probes_built
    for sw in (('1', '0') if probes_built() else ('0',)):       # the software-pipelined K loop: probe builds only
        for nowide in ('0', '1'):
            monkeypatch.setenv('D3R_GEMM_X3SW', sw)
            monkeypatch.setenv('D3R_GEMM_NOWIDE', nowide)
            compare(eng, oracle, v1, v2, *TOLS['fp16x3'], tag=f'tiny_dpt fp16x3 128x128 cfg{cfg} sw{sw} nowide{nowide}')
            e1, e2 = eng(v1, v2)
            outs.append(torch.cat((e1['pts3d'], e2['pts3d_in_other_view'])))
    for o in outs[1:]:
        assert pix_rel(o, outs[0].cpu())[0] < 1e-4
    if len(outs) == 4:
        assert torch.equal(outs[0], outs[2])        # pipelined vs plain K loop, same epilogue route: bit-identical (probe builds)


@pytest.mark.parametrize('name', ['forward_tiny_dpt.pt', 'forward_tiny_linear.pt'])
def test_forward_matches_reference_golden(gpu, name):
    """fp32 engine against vectors produced by the unmodified reference files (oracle/make_golden.py)."""
    from oracle.dust3r_ref import build_ref_model
    g = torch.load(os.path.join(GOLD, name), weights_only=False)
    eng = engine_from_oracle(build_ref_model(g['config'], seed=g['weight_seed']), g['config'], 'fp32', gpu)
    v1, v2 = synthetic_views(g['B'], g['H'], g['W'], seed=g['view_seed'])
    e1, e2 = eng(v1, v2)
    for a, b in ((e1['pts3d'], g['pts3d']), (e2['pts3d_in_other_view'], g['pts3d_in_other_view'])):
        mx, mean = pix_rel(a, b)
        assert mx < 1e-3, (mx, mean)
    assert float(((e1['conf'].cpu() - g['conf']).abs() / g['conf']).max()) < 1e-3


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'bf16'])
def test_forward_postprocess_modes_match_reference_golden(gpu, precision):
    """depth_mode 'linear' / 'square', conf_mode 'sigmoid' and finite conf bounds (heads/postprocess.py:23-58; constructor keywords of
    model.py:58-62) on the DPT head (fused EPI_HEAD4 tail in split-fp16, head_final_kernel otherwise) and the linear head, against
    vectors produced by the unmodified reference model in those modes (oracle/make_golden.py forward_modes_golden)."""
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import OUT_GAIN, synthetic_state_dict
    g = torch.load(os.path.join(GOLD, 'forward_post_modes.pt'), weights_only=False)
    v1, v2 = synthetic_views(g['B'], g['H'], g['W'], seed=g['view_seed'])
    tol, mean_tol, stat = TOLS[precision]
    for c in g['cases']:
        m = AsymmetricCroCo3DStereo(precision=precision, landscape_only=False, depth_mode=c['depth_mode'], conf_mode=c['conf_mode'], **MODEL_CONFIGS[c['config']])
        m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, g['weight_seed'], OUT_GAIN[c['config']]))
        e1, e2 = m.to(gpu)(v1, v2)
        tag = (c['config'], c['depth_mode'][0], c['conf_mode'])
        for a, b in ((e1['pts3d'], c['pts3d']), (e2['pts3d_in_other_view'], c['pts3d_in_other_view'])):
            mx, mean = pix_rel(a, b)
            assert (mx if stat == 'max' else pix_rel_p99(a, b)) < tol and mean < mean_tol, (tag, mx, mean)
        for a, b in ((e1['conf'], c['conf']), (e2['conf'], c['conf2'])):
            e = ((a.cpu() - b).abs() / b.abs().clamp_min(1e-3)).flatten()
            assert float(e.max() if stat == 'max' else e.kthvalue(int(0.99 * e.numel()))[0]) < 3 * tol, (tag, float(e.max()))
            lo, hi = c['conf_mode'][1], c['conf_mode'][2]
            assert float(a.min()) >= lo - 1e-6 and float(a.max()) <= hi + 1e-6


def test_forward_batch_position_independence(gpu):
    """A pair's result must not depend on where it sits in the batch (bit-exact): this is what makes pair
    sharding across ranks and the symmetrised-batch shortcut output-identical."""
    from oracle.dust3r_ref import build_ref_model
    eng = engine_from_oracle(build_ref_model('tiny_dpt'), 'tiny_dpt', 'bf16', gpu)
    v1, v2 = synthetic_views(4, 32, 48, seed=11)
    full1, full2 = eng(v1, v2)
    for b in (0, 3):
        s1 = dict(img=v1['img'][b:b + 1], true_shape=v1['true_shape'][b:b + 1], idx=[0], instance=['0'])
        s2 = dict(img=v2['img'][b:b + 1], true_shape=v2['true_shape'][b:b + 1], idx=[1], instance=['1'])
        o1, o2 = eng(s1, s2)
        assert torch.equal(o1['pts3d'][0], full1['pts3d'][b]) and torch.equal(o2['conf'][0], full2['conf'][b])


def test_wide_epilogue_network_matches_direct_stores(gpu, monkeypatch):
    """Whole network with the wide (LDS-staged) epilogues vs the direct fragment stores, including the q/k RoPE scatter and
    the V^T scatter of the attention projections (tiny_dpt at 128x192 has 96 tokens: the ragged V^T falls back to direct
    stores; 256x256 has 256 = 4 x 64 tokens: wide V^T). fp32 mode (only the fp32-residual epilogue changes route, same
    arithmetic): bit-identical. fp16 mode (RoPE / GELU are re-associated between the two routes, so single-ulp differences
    are legitimate and get amplified by the depth): both within the mode's tolerance of the oracle and close to each other."""
    from oracle.dust3r_ref import build_ref_model
    oracle = build_ref_model('tiny_dpt')
    for (H, W) in ((128, 192), (256, 256)):
        v1, v2 = synthetic_views(2, H, W, seed=31)
        with torch.no_grad():
            r1, _ = oracle(v1, v2)
        for precision in ('fp32', 'fp16'):
            outs = []
            for mode in ('0', '1'):
                monkeypatch.setenv('D3R_GEMM_NOWIDE', mode)
                eng = engine_from_oracle(oracle, 'tiny_dpt', precision, gpu)
                e1, e2 = eng(v1, v2)
                torch.cuda.synchronize()
                outs.append((e1['pts3d'].clone(), e2['pts3d_in_other_view'].clone(), e1['conf'].clone()))
            if precision == 'fp32':
                for x, y in zip(*outs):
                    assert torch.equal(x, y)
            else:
                for o in outs:
                    assert pix_rel_p99(o[0], r1['pts3d']) < TOLS['fp16'][0] and pix_rel(o[0], r1['pts3d'])[1] < TOLS['fp16'][1]
                assert pix_rel(outs[0][0], outs[1][0].cpu())[1] < 5e-3


def test_two_stream_decoder_is_bit_identical(gpu):
    """The second-stream schedule of decoder side 2 / head 2 only reorders independent launches: outputs are bit-identical
    to the single-stream schedule, repeatedly (a missing cross-stream dependency would show as run-to-run differences)."""
    from oracle.dust3r_ref import build_ref_model
    eng = engine_from_oracle(build_ref_model('tiny_dpt'), 'tiny_dpt', 'bf16', gpu)
    v1, v2 = synthetic_views(4, 64, 96, seed=21)
    eng.set_two_streams(False)
    a1, a2 = eng(v1, v2)
    torch.cuda.synchronize()
    eng.set_two_streams(True)
    for _ in range(5):
        b1, b2 = eng(v1, v2)
        torch.cuda.synchronize()
        assert torch.equal(a1['pts3d'], b1['pts3d']) and torch.equal(a1['conf'], b1['conf'])
        assert torch.equal(a2['pts3d_in_other_view'], b2['pts3d_in_other_view']) and torch.equal(a2['conf'], b2['conf'])


def test_inference_api_end_to_end(gpu):
    """make_pairs -> inference(): same structure / edge order as the reference golden, values from the fp32 engine."""
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.synthetic import synthetic_image_list
    from oracle.dust3r_ref import build_ref_model
    g = torch.load(os.path.join(GOLD, 'inference_tiny_dpt.pt'), weights_only=False)
    eng = engine_from_oracle(build_ref_model(g['config']), g['config'], 'fp32', gpu)
    imgs = synthetic_image_list(g['n_views'], g['H'], g['W'], seed=g['view_seed'])
    out = inference(make_pairs(imgs, 'complete', None, True), eng, gpu, batch_size=2, verbose=False)
    assert out['view1']['idx'] == g['idx1'] and out['view2']['idx'] == g['idx2'] and out['loss'] is None
    assert out['pred1']['pts3d'].device.type == 'cpu' and out['view1']['img'].device.type == 'cpu'
    mx, _ = pix_rel(out['pred1']['pts3d'], g['pts3d'])
    mx2, _ = pix_rel(out['pred2']['pts3d_in_other_view'], g['pts3d_in_other_view'])
    assert mx < 1e-3 and mx2 < 1e-3


def test_inference_batching_is_output_identical(gpu):
    """`inference()` coalesces to `model.engine_batch` pairs per engine call whatever batch_size the caller names (the reference demo
    passes 1), and groups pairs of mixed image sizes by shape instead of running them one by one (dust3r/inference.py:60-68): both
    must be bit-identical to the reference's schedule, on the default-precision engine."""
    from dust3r_amd.inference import inference
    from oracle.dust3r_ref import build_ref_model
    eng = engine_from_oracle(build_ref_model('tiny_dpt'), 'tiny_dpt', None, gpu)
    g = torch.Generator().manual_seed(4)
    shapes = [(32, 48), (48, 32), (32, 48), (32, 32), (48, 32), (32, 48), (32, 48)]
    imgs = [dict(img=torch.rand((1, 3, h, w), generator=g) * 2 - 1, true_shape=torch.tensor([[h, w]], dtype=torch.int32), idx=k, instance=str(k))
            for k, (h, w) in enumerate(shapes)]
    mixed = [(imgs[i], imgs[j]) for i in range(len(imgs)) for j in range(len(imgs)) if i != j]
    same = [(imgs[i], imgs[j]) for i in (0, 2, 5, 6) for j in (0, 2, 5, 6) if i != j]
    for pairs in (mixed, same):
        outs = []
        for eb in (1, 5):
            eng.engine_batch = eb
            outs.append(inference(pairs, eng, gpu, batch_size=1, verbose=False, encode_once=False))
        eng.engine_batch = 32
        outs.append(inference(pairs, eng, gpu, batch_size=8, verbose=False, encode_once=False, engine_batch=2))   # the caller pins a SMALLER call size
        a = outs[0]
        for b in outs[1:]:
            for view, keys in (('pred1', ('pts3d', 'conf')), ('pred2', ('pts3d_in_other_view', 'conf'))):
                for k in keys:
                    xa, xb = a[view][k], b[view][k]
                    if isinstance(xa, list):
                        assert len(xa) == len(xb) == len(pairs) and all(torch.equal(p, q) for p, q in zip(xa, xb)), (view, k)
                    else:
                        assert torch.equal(xa, xb), (view, k)
            assert a['view1']['idx'] == b['view1']['idx'] and a['view2']['instance'] == b['view2']['instance']


def test_full_size_fp32_pair_matches_oracle(gpu):
    """BASELINE config: DUSt3R_ViTLarge_BaseDecoder_512_dpt, one 512x384 pair, fp32 engine vs CPU oracle <= 1e-3."""
    from oracle.dust3r_ref import build_ref_model
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    from oracle.dust3r_ref import build_ref_model_fast
    oracle = build_ref_model_fast(cfg)          # same architecture; in-place seeded init (the engine loads ITS state dict)
    eng = engine_from_oracle(oracle, cfg, 'fp32', gpu)
    v1, v2 = synthetic_views(1, 384, 512, seed=0)
    with torch.no_grad():
        r1, r2 = oracle(v1, v2)
    for prec in ('fp32', 'fp16x3', 'fp16x2f8', 'fp16f8'):   # fp32, fp16x3 (the default) and the 2.5-unit mode are held to the north-star bar: per-pixel max <= 1e-3
        eng.set_precision(prec)
        e1, e2 = eng(v1, v2)
        for name, a, b in (('pts1', e1['pts3d'], r1['pts3d']), ('pts2', e2['pts3d_in_other_view'], r2['pts3d_in_other_view'])):
            mx, mean = pix_rel(a, b)
            p99 = pix_rel_p99(a, b)
            print(f'[512_dpt {prec}] {name} rel err max {mx:.3e} p99 {p99:.3e} mean {mean:.3e}')
            if prec == 'fp16f8':                # opt-in fast mode: characterised, not claimed (measured max 3.3e-4 on this seed)
                assert p99 < 1e-3 and mean < 2e-4, (prec, name, mx, p99, mean)
            else:
                assert mx < 1e-3 and mean < 2e-4, (prec, name, mx, mean)
        cerr = float(((e1['conf'].cpu() - r1['conf']).abs() / r1['conf']).max())
        print(f'[512_dpt {prec}] conf1 rel err max {cerr:.3e}')
        assert cerr < 1e-3
    # 16-bit modes on the full network: report (and bound) the error against the same fp32 oracle outputs
    for prec, bound in (('fp16', 2e-2), ('bf16', 1e-1)):
        eng.set_precision(prec)
        e1, _ = eng(v1, v2)
        mx, mean = pix_rel(e1['pts3d'], r1['pts3d'])
        print(f'[512_dpt {prec}] pts1 rel err max {mx:.3e} mean {mean:.3e}')
        assert mean < bound


@pytest.mark.parametrize('big', [40.0, 150.0])
def test_full_size_default_mode_under_sharp_attention_and_outlier_channels(gpu, big):
    """Robustness of the default engine (fp16x3) on weights with two traits of TRAINED ViTs that a seeded random network lacks: sharp
    attention (q / k projections x2: logit std ~2 instead of ~0.5) and outlier channels in the MLP inputs (LayerNorm gains of norm2 /
    norm3: every 97th channel x8, one channel x40 or x150 -- activations in the hundreds). (Outliers in the ATTENTION inputs turn the
    softmax into an argmax and the network into a discontinuous function that no arithmetic reproduces, fp32 engine vs fp32 oracle
    included.) These weights also push pointmaps through the origin (min |pts| 0.02), so the per-pixel RELATIVE error is
    ill-conditioned in every mode: measured max 6e-4 / 2.6e-3 for the exact-fp32 ENGINE against the fp32 oracle (accumulation order).
    Held against the CPU oracle, per view:
      fp16x3 (default): 99th percentile <= 1e-3, mean <= 3e-4, and a per-pixel max that is either inside the bar or within 2x of what
                        the exact-fp32 engine itself shows on the same weights (i.e. the conditioning of the test, not the arithmetic);
      fp16f8 (opt-in) : printed. Its un-scaled e4m3 copies saturate at 448: p99 1.3e-3 (x40) / 2.2-4.0e-3 (x150) in round 2 --
                        the reason the mode is not the default. Only its mean is bounded (a broken kernel, not a rounding floor)."""
    from oracle.dust3r_ref import build_ref_model_fast
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    oracle = build_ref_model_fast(cfg)
    with torch.no_grad():
        for name, p in oracle.named_parameters():
            if name.endswith('attn.qkv.weight'):
                p[:2 * p.shape[1]] *= 2.0
            elif name.endswith('cross_attn.projq.weight') or name.endswith('cross_attn.projk.weight'):
                p *= 2.0
            elif 'blocks' in name and (name.endswith('.norm2.weight') and 'enc_blocks' in name or name.endswith('.norm3.weight')):
                p[5::97] *= 8.0
                p[3] *= big
    eng = engine_from_oracle(oracle, cfg, None, gpu)
    assert eng.precision == 'fp16x3'
    v1, v2 = synthetic_views(1, 384, 512, seed=3)
    with torch.no_grad():
        r1, r2 = oracle(v1, v2)
    worst, stats = {}, {}
    for prec in ('fp32', 'fp16x3', 'fp16x2f8', 'fp16f8'):
        eng.set_precision(prec)
        e1, e2 = eng(v1, v2)
        for name, a, b in (('pts1', e1['pts3d'], r1['pts3d']), ('pts2', e2['pts3d_in_other_view'], r2['pts3d_in_other_view'])):
            mx, mean = pix_rel(a, b)
            p99 = pix_rel_p99(a, b)
            print(f'[512_dpt {prec}, sharp attention + outlier channels x{big:g}] {name} rel err max {mx:.3e} p99 {p99:.3e} mean {mean:.3e}   |pts| min {float(b.norm(dim=-1).min()):.3e}')
            worst[prec] = max(worst.get(prec, 0.0), mx)
            stats[(prec, name)] = (mx, p99, mean)
    for name in ('pts1', 'pts2'):
        mx, p99, mean = stats[('fp16x3', name)]
        assert p99 < 1e-3 and mean < 3e-4, ('fp16x3', name, mx, p99, mean)
        assert stats[('fp16f8', name)][2] < 1.5e-3
    assert worst['fp16x3'] < max(1e-3, 2 * worst['fp32']), worst


@pytest.mark.parametrize('seed', [1, 3])
def test_default_mode_error_distribution_over_seeds(gpu, seed):
    """How the error of the default mode (fp16x3) is distributed, on the two weight seeds of tools/margin_survey.py (6 seeds x 4 pairs,
    profiles/r02_f8/margin_survey.log) with the largest per-pixel maxima. Comparator: the exact-fp32 ENGINE (within 5e-5 of the CPU
    oracle in test_full_size_fp32_pair_matches_oracle; a CPU oracle run per seed would take minutes), BASELINE model, 2 pairs 512x384.
    A seeded random network sends some pointmaps through the origin (|pts| down to 0.03 % of the mean norm), where |delta| / |pts| is
    ill-conditioned for ANY arithmetic. Held for the default: per-pixel max <= 1e-3 (the bar, strict ratio), 99.99th percentile <= 3e-4,
    max |delta| / mean |pts| <= 3e-4. The opt-in fp16f8 mode is characterised next to it (p99.99 and scale-relative max <= 1e-3; its
    per-pixel max passes the bar on these seeds: 1.9e-3 / 2.7e-3 in round 2 -- printed, not asserted)."""
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import OUT_GAIN, synthetic_state_dict
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    m = AsymmetricCroCo3DStereo(precision='fp32', landscape_only=False, **MODEL_CONFIGS[cfg])
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, seed, OUT_GAIN[cfg], device=gpu))
    m.to(gpu)
    v1, v2 = synthetic_views(2, 384, 512, seed=100 + seed, device=gpu)
    r1, r2 = m(v1, v2)
    ref = torch.cat((r1['pts3d'], r2['pts3d_in_other_view'])).clone()
    nrm = ref.norm(dim=-1).clamp_min(1e-8)
    for prec, bar in (('fp16x3', 3e-4), ('fp16f8', 1e-3)):
        m.set_precision(prec)
        e1, e2 = m(v1, v2)
        dn = (torch.cat((e1['pts3d'], e2['pts3d_in_other_view'])) - ref).norm(dim=-1)
        relm = (dn / nrm).flatten()
        rel = relm.sort().values
        p9999, scaled = float(rel[int(0.9999 * rel.numel())]), float(dn.max() / nrm.mean())
        worst_norm = float(nrm.flatten()[relm.argmax()] / nrm.mean())
        print(f'[512_dpt seed {seed} {prec} vs fp32 engine] per-pixel max {float(rel[-1]):.3e} (at a pixel with |pts| = {worst_norm:.2e} of the mean norm) '
              f'p99.99 {p9999:.3e} mean {float(rel.mean()):.3e}; max |delta| / mean |pts| {scaled:.3e}')
        assert p9999 < bar and scaled < bar and float(rel.mean()) < bar / 5, (prec, p9999, scaled)
        if prec == 'fp16x3':
            assert float(rel[-1]) < 1e-3, (prec, float(rel[-1]), worst_norm)


def test_config1_pairviewer_pipeline(gpu):
    """BASELINE configs[0] plumbing on the engine: a 224x224 linear-head model, 2 images -> 1 symmetrised pair ->
    inference() -> GlobalAlignerMode.PairViewer -> getters, with demo.py's call sequence and the reference's shapes."""
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.synthetic import synthetic_image_list
    from oracle.dust3r_ref import build_ref_model
    oracle = build_ref_model('tiny_linear')
    eng = engine_from_oracle(oracle, 'tiny_linear', 'fp16x3', gpu)
    imgs = synthetic_image_list(2, 224, 224, seed=9)
    pairs = make_pairs(imgs, scene_graph='complete', prefilter=None, symmetrize=True)
    assert len(pairs) == 2
    out = inference(pairs, eng, gpu, batch_size=1, verbose=False)
    assert out['pred1']['pts3d'].shape == (2, 224, 224, 3) and out['pred2']['conf'].shape == (2, 224, 224)
    # same numbers as the oracle's inference on the same pairs (<= 1e-3)
    with torch.no_grad():
        for e, (a, b) in enumerate(pairs):
            r1, r2 = oracle(a, b)
            mx, _ = pix_rel(out['pred1']['pts3d'][e:e + 1], r1['pts3d'])
            assert mx < 1e-3
    scene = global_aligner(out, gpu, mode=GlobalAlignerMode.PairViewer, verbose=False)
    poses, focals = scene.get_im_poses(), scene.get_focals()
    assert poses.shape == (2, 4, 4) and focals.shape[0] == 2 and torch.isfinite(poses).all() and torch.isfinite(focals).all()
    pts = scene.get_pts3d()
    assert len(pts) == 2 and pts[0].shape == (224, 224, 3) and len(scene.get_masks()) == 2


def test_encode_once_inference_is_bit_identical(gpu):
    """inference() with each distinct image encoded once (d3r_model_encode / d3r_model_decode) returns exactly what the
    pair-by-pair path returns: same structure, same view order, bit-identical predictions (complete symmetrised graph)."""
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.synthetic import synthetic_image_list
    from oracle.dust3r_ref import build_ref_model
    eng = engine_from_oracle(build_ref_model('tiny_dpt'), 'tiny_dpt', 'bf16', gpu)
    imgs = synthetic_image_list(4, 64, 96, seed=13)
    pairs = make_pairs(imgs, 'complete', None, True)
    a = inference(pairs, eng, gpu, batch_size=3, verbose=False, encode_once=False)
    b = inference(pairs, eng, gpu, batch_size=3, verbose=False, encode_once=True)
    assert a['view1']['idx'] == b['view1']['idx'] and a['view2']['idx'] == b['view2']['idx'] and b['loss'] is None
    assert torch.equal(a['view1']['img'], b['view1']['img']) and b['pred1']['pts3d'].device.type == 'cpu'
    for k in ('pts3d', 'conf'):
        assert torch.equal(a['pred1'][k], b['pred1'][k])
    assert torch.equal(a['pred2']['pts3d_in_other_view'], b['pred2']['pts3d_in_other_view']) and torch.equal(a['pred2']['conf'], b['pred2']['conf'])


def test_packed_forward_equals_forward(gpu):
    """d3r_model_forward_packed writes the same numbers as forward, interleaved per pixel (the multi-GPU gather payload)."""
    from dust3r_amd.parallel import unpack_predictions
    from oracle.dust3r_ref import build_ref_model
    for cfg in ('tiny_dpt', 'tiny_linear'):
        eng = engine_from_oracle(build_ref_model(cfg), cfg, 'bf16', gpu)
        v1, v2 = synthetic_views(3, 64, 96, seed=17)
        r1, r2 = eng(v1, v2)
        p1, p2 = unpack_predictions(eng.forward_packed(v1, v2))
        assert torch.equal(p1['pts3d'], r1['pts3d']) and torch.equal(p1['conf'], r1['conf'])
        assert torch.equal(p2['pts3d_in_other_view'], r2['pts3d_in_other_view']) and torch.equal(p2['conf'], r2['conf'])


def test_sharded_inference_on_the_engine_with_rccl(gpu):
    """The CUDA branch of dust3r_amd.parallel.inference_sharded (engine heads write the packed all-gather payload in place, then ONE
    all_gather_into_tensor over RCCL): world size 1 on this box's GPU, bit-identical to the single-process inference(). The multi-rank
    behaviour of the same code is covered by the gloo tests (tests/test_parallel_cpu.py, world 2 and 4)."""
    import socket
    import torch.distributed as dist
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.parallel import inference_sharded
    from dust3r_amd.synthetic import synthetic_image_list
    from oracle.dust3r_ref import build_ref_model
    eng = engine_from_oracle(build_ref_model('tiny_dpt'), 'tiny_dpt', 'fp16x3', gpu)
    pairs = make_pairs(synthetic_image_list(4, 32, 48, seed=8), 'complete', None, symmetrize=False)
    ref = inference(pairs, eng, gpu, batch_size=4, verbose=False, encode_once=False)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=gpu)
    try:
        out = inference_sharded(pairs, eng, gpu, batch_size=4)
    finally:
        dist.destroy_process_group()
    assert out['view1']['idx'] == ref['view1']['idx'] and out['view2']['idx'] == ref['view2']['idx']
    assert torch.equal(out['pred1']['pts3d'], ref['pred1']['pts3d']) and torch.equal(out['pred1']['conf'], ref['pred1']['conf'])
    assert torch.equal(out['pred2']['pts3d_in_other_view'], ref['pred2']['pts3d_in_other_view']) and torch.equal(out['pred2']['conf'], ref['pred2']['conf'])


@pytest.mark.parametrize('B', [1, 2])
def test_fused_head_tail_equals_the_two_kernel_route(gpu, B, monkeypatch):
    """Split-fp16, BASELINE model: the DPT head's Conv2d(128, 4, 1) + postprocess run in the epilogue of the last 3x3 convolution (EPI_HEAD4,
    on the fp32 accumulators; 512 x 128 tiles from two images per head on, the four-wave 256 x 128 tile below) instead of storing the
    128-channel full-resolution map and reading it back (D3R_HEAD_FUSE=0). Same values up to the rounding of the map to 22 bits."""
    from bench import build_model       # the bench's synthetic full-size weights, generated in HBM (tests/conftest.py puts the repo root on sys.path)
    eng = build_model('fp16x3', gpu)
    v1, v2 = synthetic_views(B, 384, 512, seed=3, device=gpu)
    out = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('D3R_HEAD_FUSE', fuse)
        e1, e2 = eng(v1, v2)
        out[fuse] = [t.clone() for t in (e1['pts3d'], e2['pts3d_in_other_view'], e1['conf'], e2['conf'])]
        pk = eng.forward_packed(v1, v2)     # the all-gather payload (B,H,W,8) = [pts1 conf1 pts2 conf2] written by the same epilogue with strides (8, 8)
        assert torch.equal(pk[..., 0:3], e1['pts3d']) and torch.equal(pk[..., 3], e1['conf'])
        assert torch.equal(pk[..., 4:7], e2['pts3d_in_other_view']) and torch.equal(pk[..., 7], e2['conf'])
    for a, b in zip(out['1'][:2], out['0'][:2]):
        mx, mean = pix_rel(a, b.cpu())
        print(f'[fused head B={B}] pointmap rel diff max {mx:.3e} mean {mean:.3e}')
        assert mx < 2e-5 and mean < 1e-6
    for a, b in zip(out['1'][2:], out['0'][2:]):
        assert float(((a - b).abs() / b).max()) < 1e-5


@pytest.mark.parametrize('precision', ['fp16x3', 'bf16'])
def test_small_batch_graph_replay_is_bit_identical(gpu, precision):
    """One or two pairs per call (dust3r/demo.py:156 passes batch_size=1, visloc.py:88 one pair per query): from the third call with the
    same shapes on, the engine replays a captured hipGraph through its own staging buffers. Results must be bit-identical to the eager
    schedule, for the plain, the mixed-size and the packed entry points, with fresh input / output tensors on every call, and a call with
    other shapes in between must not disturb a captured graph."""
    from oracle.dust3r_ref import build_ref_model
    eng = engine_from_oracle(build_ref_model('tiny_dpt'), 'tiny_dpt', precision, gpu)
    g = torch.Generator().manual_seed(12)

    def views(B, hw1, hw2):
        return (dict(img=torch.rand((B, 3) + hw1, generator=g) * 2 - 1, true_shape=torch.tensor([hw1] * B, dtype=torch.int32), idx=list(range(B)), instance=['a'] * B),
                dict(img=torch.rand((B, 3) + hw2, generator=g) * 2 - 1, true_shape=torch.tensor([hw2] * B, dtype=torch.int32), idx=list(range(B)), instance=['b'] * B))
    cases = [views(1, (64, 96), (64, 96)), views(2, (32, 48), (32, 48)), views(1, (32, 48), (48, 32))]
    eng.set_graph_max_pairs(0)
    want = []
    for v1, v2 in cases:
        r1, r2 = eng(v1, v2)
        want.append([t.clone() for t in (r1['pts3d'], r1['conf'], r2['pts3d_in_other_view'], r2['conf'])])
    want_packed = eng.forward_packed(*cases[0]).clone()
    eng.set_graph_max_pairs(4)
    before = eng.graph_replays()
    keep = []
    for rep in range(4):                      # eager, capture + replay, replay, replay -- interleaved over the three shapes
        for (v1, v2), w in zip(cases, want):
            r1, r2 = eng(v1, v2)
            keep.append((r1, r2))             # outputs stay alive: every call sees new output addresses
            got = (r1['pts3d'], r1['conf'], r2['pts3d_in_other_view'], r2['conf'])
            assert all(torch.equal(a, b) for a, b in zip(got, w)), rep
        assert torch.equal(eng.forward_packed(*cases[0]), want_packed), rep
    torch.cuda.synchronize()
    assert eng.graph_replays() - before >= 8, eng.graph_replays() - before
    # a bigger batch than the limit stays eager and still agrees row by row
    v1, v2 = views(5, (32, 48), (32, 48))
    full1, _ = eng(v1, v2)
    one1, _ = eng(dict(img=v1['img'][3:4], true_shape=v1['true_shape'][3:4], idx=[0], instance=['a']),
                  dict(img=v2['img'][3:4], true_shape=v2['true_shape'][3:4], idx=[0], instance=['b']))
    assert torch.equal(full1['pts3d'][3], one1['pts3d'][0])


def test_two_ranks_on_one_gpu_sharded_inference(gpu, tmp_path):
    """dust3r_amd.parallel.inference_sharded with TWO ranks, both on this box's one GPU, each with its own engine (gloo moves the CUDA
    payload; RCCL refuses two ranks on one device and is exercised at world size 1 above): the one-size list takes the encode-once route
    (each rank encodes the images of its shard once, the heads write the packed payload in place), the mixed-size list the flat padded
    payload. Every rank's result must be bit-identical to single-process inference() on the same engine."""
    import socket
    import torch.multiprocessing as mp
    sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    if sys_path_root not in sys.path:
        sys.path.insert(0, sys_path_root)
    from tests._dist_worker import build_engine, scene_pairs, worker
    from dust3r_amd.inference import inference
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    eng = build_engine(gpu)                                  # meanwhile: the single-process reference on a third engine
    ref = {kind: inference(scene_pairs(kind), eng, gpu, batch_size=4, verbose=False) for kind in ('same', 'mixed')}
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f'rank{r}.pt'), weights_only=False)
        for kind in ('same', 'mixed'):
            a, b = ref[kind], got[kind]
            assert a['view1']['idx'] == b['view1']['idx'] and a['view2']['idx'] == b['view2']['idx']
            for view, keys in (('pred1', ('pts3d', 'conf')), ('pred2', ('pts3d_in_other_view', 'conf'))):
                for k in keys:
                    xa, xb = a[view][k], b[view][k]
                    if isinstance(xa, list):
                        assert len(xa) == len(xb) and all(torch.equal(p, q) for p, q in zip(xa, xb)), (kind, view, k)
                    else:
                        assert torch.equal(xa, xb), (kind, view, k)


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'fp16x2f8', 'fp16f8'])
@pytest.mark.parametrize('hw1,hw2', [((32, 48), (48, 32)), ((64, 64), (32, 48)), ((48, 80), (64, 128))])
def test_pairs_of_two_image_sizes(gpu, precision, hw1, hw2):
    """The else-branch of dust3r/model.py:148-150 (a landscape image paired with a portrait one): the two views are encoded separately
    and cross attention runs with Nq != Nk (d3r_model_forward_mixed). Against the oracle, same 1e-3 bar; and through inference()'s
    mixed-shape path (batch size 1, lists instead of concatenated tensors, inference.py:60-72)."""
    from oracle.dust3r_ref import build_ref_model
    from dust3r_amd.inference import inference
    oracle = build_ref_model('tiny_dpt')
    eng = engine_from_oracle(oracle, 'tiny_dpt', precision, gpu)
    g = torch.Generator().manual_seed(hw1[0] * 7 + hw2[1])
    B = 2
    v1 = dict(img=torch.rand((B, 3) + hw1, generator=g) * 2 - 1, true_shape=torch.tensor([hw1] * B, dtype=torch.int32), idx=[0, 2], instance=['0', '2'])
    v2 = dict(img=torch.rand((B, 3) + hw2, generator=g) * 2 - 1, true_shape=torch.tensor([hw2] * B, dtype=torch.int32), idx=[1, 3], instance=['1', '3'])
    compare(eng, oracle, v1, v2, *TOLS[precision], tag=f'tiny_dpt {precision} {hw1} x {hw2}')
    e1, e2 = eng(v1, v2)
    assert e1['pts3d'].shape == (B,) + hw1 + (3,) and e2['pts3d_in_other_view'].shape == (B,) + hw2 + (3,) and e2['conf'].shape == (B,) + hw2
    # inference(): one same-size pair + one mixed pair -> the multiple-shapes path
    mk = lambda img, hw, i: dict(img=img[None], true_shape=np.int32([hw]), idx=i, instance=str(i))  # noqa: E731
    pairs = [(mk(v1['img'][0], hw1, 0), mk(v2['img'][0], hw2, 1)), (mk(v1['img'][1], hw1, 2), mk(v1['img'][0], hw1, 0))]
    out = inference(pairs, eng, gpu, batch_size=8, verbose=False)
    assert isinstance(out['pred1']['pts3d'], list) and len(out['pred1']['pts3d']) == 2
    assert out['pred2']['pts3d_in_other_view'][0].shape[-3:-1] == hw2 and out['pred2']['pts3d_in_other_view'][1].shape[-3:-1] == hw1
    assert torch.equal(out['pred1']['pts3d'][0].reshape(hw1 + (3,)), e1['pts3d'][0].cpu())


# ---- LayerNorm folded into the GEMMs around it (round 5, D3R_LN_FOLD=1 at engine creation) --------------------------------------------------
def _randomize_norms(model, seed, gain=0.3, shift=0.2):
    """every LayerNorm weight 1 + gain N(0, 1), every LayerNorm bias shift N(0, 1): the fold moves gamma into the following matrix and beta into
    its bias, so the test weights must not be the identity affine"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if 'norm' in name and prm.ndim == 1:
                if name.endswith('weight'):
                    prm.copy_(1 + gain * torch.randn(prm.shape, generator=g))
                else:
                    prm.copy_(shift * torch.randn(prm.shape, generator=g))
    return model


def _fold_engines(oracle, config, gpu, monkeypatch):
    monkeypatch.setenv('D3R_LN_FOLD', '0')
    plain = engine_from_oracle(oracle, config, 'fp16x3', gpu)
    monkeypatch.setenv('D3R_LN_FOLD', '1')
    folded = engine_from_oracle(oracle, config, 'fp16x3', gpu)
    return plain, folded


@pytest.mark.parametrize('config,B,H,W', [('tiny_dpt', 2, 32, 48), ('tiny_dpt', 1, 64, 64), ('tiny_dpt', 3, 128, 128), ('tiny_dpt', 2, 64, 96), ('tiny_linear', 3, 32, 32),
                                          ('tiny_linear', 1, 224, 224), ('tiny_dpt', 1, 128, 256)])
def test_layernorm_fold_matches_oracle_and_the_unfolded_engine(gpu, config, B, H, W, monkeypatch):
    """norm1 / norm2 (encoder) and norm1 / norm2 / norm3 / norm_y (decoder) folded into the GEMMs around them: LN(x) W^T + b = rstd (x (W diag(gamma))^T -
    mean s) + b'. The producer's fp32-residual epilogue stores the raw typed rows and per-row partial sums, the consumer's epilogue applies rstd / mean
    (csrc/kernels.hpp GemmParams::ln_*). Random gamma / beta. Held to the default engine's bar against the CPU oracle, within 5e-5 of the engine
    that launches the LayerNorm kernels, and -- token counts 6 .. 512, i.e. both the LDS-staged and the direct epilogue routes, q / k / V^T regions --
    a batch bit-equal to its one-pair calls."""
    from oracle.dust3r_ref import build_ref_model
    oracle = _randomize_norms(build_ref_model(config), seed=H * 3 + W)
    plain, folded = _fold_engines(oracle, config, gpu, monkeypatch)
    v1, v2 = synthetic_views(B, H, W, seed=H + W)
    compare(folded, oracle, v1, v2, *TOLS['fp16x3'], tag=f'{config} LN-fold {B}x{H}x{W}')
    a1, a2 = plain(v1, v2)
    b1, b2 = folded(v1, v2)
    for x, y in ((a1['pts3d'], b1['pts3d']), (a2['pts3d_in_other_view'], b2['pts3d_in_other_view'])):
        mx, mean = pix_rel(y, x.cpu())
        print(f'[{config} {B}x{H}x{W}] folded vs LayerNorm kernels: max {mx:.3e} mean {mean:.3e}')
        assert mx < 3e-4 and mean < 1e-5          # two 22-bit evaluations of the same network (the folded engine also keeps the residual stream in split-fp16 rows)
    assert float(((b1['conf'] - a1['conf']).abs() / a1['conf']).max()) < 2e-4
    if B > 1:
        for k in range(B):
            s1 = dict(img=v1['img'][k:k + 1], true_shape=v1['true_shape'][k:k + 1], idx=[0], instance=['0'])
            s2 = dict(img=v2['img'][k:k + 1], true_shape=v2['true_shape'][k:k + 1], idx=[1], instance=['1'])
            o1, o2 = folded(s1, s2)
            assert torch.equal(o1['pts3d'][0], b1['pts3d'][k]) and torch.equal(o2['pts3d_in_other_view'][0], b2['pts3d_in_other_view'][k]) and torch.equal(o2['conf'][0], b2['conf'][k])


def test_layernorm_fold_is_tile_independent_and_handles_two_image_sizes(gpu, monkeypatch):
    """The row statistics are summed in one fixed tree per 32-column group whatever tile produced them: with the GEMM tile pinned (128 x 128, 256 x 256,
    the two-blocks-per-CU 256 x 128, 64 x 64, 384 x 192; infeasible pins fall back per launch) the folded engine's outputs are BIT-identical. Pairs of two
    image sizes (the other view's statistics feed norm_y with Nq != Nk) against the oracle."""
    from oracle.dust3r_ref import build_ref_model
    oracle = _randomize_norms(build_ref_model('tiny_dpt'), seed=11)
    monkeypatch.setenv('D3R_LN_FOLD', '1')
    eng = engine_from_oracle(oracle, 'tiny_dpt', 'fp16x3', gpu)
    v1, v2 = synthetic_views(3, 128, 128, seed=9)
    outs = []
    for cfg in ('', '0', '1', '7', '8', '9'):
        if cfg:
            monkeypatch.setenv('D3R_GEMM_CFG', cfg)
        e1, e2 = eng(v1, v2)
        outs.append((e1['pts3d'].clone(), e2['pts3d_in_other_view'].clone(), e2['conf'].clone()))
    monkeypatch.delenv('D3R_GEMM_CFG')
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    g = torch.Generator().manual_seed(3)
    for hw1, hw2 in (((32, 48), (48, 32)), ((64, 64), (32, 48)), ((64, 128), (128, 128))):
        w1 = dict(img=torch.rand((2, 3) + hw1, generator=g) * 2 - 1, true_shape=torch.tensor([hw1] * 2, dtype=torch.int32), idx=[0, 2], instance=['0', '2'])
        w2 = dict(img=torch.rand((2, 3) + hw2, generator=g) * 2 - 1, true_shape=torch.tensor([hw2] * 2, dtype=torch.int32), idx=[1, 3], instance=['1', '3'])
        compare(eng, oracle, w1, w2, *TOLS['fp16x3'], tag=f'tiny_dpt LN-fold {hw1} + {hw2}')


def test_layernorm_fold_full_size_against_oracle(gpu, monkeypatch):
    """BASELINE model with random LayerNorm affines, one 512x384 pair: the folded engine against the CPU oracle at the north-star bar (per-pixel max
    <= 1e-3, mean <= 5e-5) and against the engine with LayerNorm kernels; four pairs per call bit-equal to one-pair calls; encode / decode entry points."""
    from oracle import tune_threads
    from oracle.dust3r_ref import build_ref_model_fast
    tune_threads()
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    oracle = _randomize_norms(build_ref_model_fast(cfg, seed=4), seed=21)
    v1, v2 = synthetic_views(1, 384, 512, seed=14)
    with torch.no_grad():
        r1, r2 = oracle(v1, v2)
    plain, folded = _fold_engines(oracle, cfg, gpu, monkeypatch)
    res, worst = {}, {}
    ref = torch.cat((r1['pts3d'], r2['pts3d_in_other_view']))
    for name, eng in (('LayerNorm kernels', plain), ('folded', folded)):
        e1, e2 = eng(v1, v2)
        res[name] = torch.cat((e1['pts3d'], e2['pts3d_in_other_view'])).cpu()
        mx, mean = pix_rel(res[name], ref)
        p99 = pix_rel_p99(res[name], ref)
        worst[name] = mx
        print(f'[512_dpt random LN affines, {name} vs CPU oracle] max {mx:.3e} p99 {p99:.3e} mean {mean:.3e}')
        assert p99 < 2e-4 and mean < 5e-5
        assert float(((e1['conf'].cpu() - r1['conf']).abs() / r1['conf']).max()) < 1e-3
    # these weights send pointmaps through the origin (per-pixel max of the LayerNorm-kernel engine itself: 2e-3 at one pixel): the folded engine is
    # held to the bar, or to the unfolded engine's own worst pixel where that is above it
    assert worst['folded'] < max(1e-3, 1.5 * worst['LayerNorm kernels'])
    mx, mean = pix_rel(res['folded'], res['LayerNorm kernels'])
    print(f'[512_dpt folded vs LayerNorm kernels] max {mx:.3e} mean {mean:.3e}')
    assert mean < 1e-5
    plain._destroy_engine()
    w1, w2 = synthetic_views(4, 384, 512, seed=15, device=gpu)
    b1, b2 = folded(w1, w2)
    for k in (0, 3):
        s1 = dict(img=w1['img'][k:k + 1], true_shape=w1['true_shape'][k:k + 1], idx=[0], instance=['0'])
        s2 = dict(img=w2['img'][k:k + 1], true_shape=w2['true_shape'][k:k + 1], idx=[1], instance=['1'])
        o1, o2 = folded(s1, s2)
        assert torch.equal(o1['pts3d'][0], b1['pts3d'][k]) and torch.equal(o2['pts3d_in_other_view'][0], b2['pts3d_in_other_view'][k])
    feat = folded.encode_images(torch.cat((w1['img'], w2['img'])))
    d1, d2 = folded.decode_pairs(feat, 384, 512)
    assert torch.equal(d1['pts3d'], b1['pts3d']) and torch.equal(d2['conf'], b2['conf'])
    folded._destroy_engine()


def test_encoder_views_on_two_streams_is_bit_identical(gpu, monkeypatch):
    """D3R_ENC_SPLIT=n at engine creation (round 5 probe, default off: measured 10.05 vs 10.10 ms for one 512x384 pair, slower from four pairs on): calls of at most n
    images run the encoder of view 1 and of view 2 as two concurrent chains on the engine's two streams, each in its own rows of every scratch buffer. Same kernels on the
    same rows: bit-identical to the one-chain schedule, for pairs of one and of two image sizes, folded LayerNorm on and off."""
    from oracle.dust3r_ref import build_ref_model
    from conftest import need_probes
    need_probes('D3R_ENC_SPLIT (a schedule that never became the default)')
    oracle = build_ref_model('tiny_dpt')
    g = torch.Generator().manual_seed(5)
    cases = []
    for hw1, hw2 in (((64, 96), (64, 96)), ((32, 48), (48, 32)), ((128, 128), (128, 128))):
        cases.append((dict(img=torch.rand((2, 3) + hw1, generator=g) * 2 - 1, true_shape=torch.tensor([hw1] * 2, dtype=torch.int32), idx=[0, 2], instance=['0', '2']),
                      dict(img=torch.rand((2, 3) + hw2, generator=g) * 2 - 1, true_shape=torch.tensor([hw2] * 2, dtype=torch.int32), idx=[1, 3], instance=['1', '3'])))
    for fold in ('1', '0'):
        monkeypatch.setenv('D3R_LN_FOLD', fold)
        outs = {}
        for split in ('0', '64'):
            monkeypatch.setenv('D3R_ENC_SPLIT', split)
            eng = engine_from_oracle(oracle, 'tiny_dpt', 'fp16x3', gpu)
            outs[split] = []
            for v1, v2 in cases:
                e1, e2 = eng(v1, v2)
                outs[split].append((e1['pts3d'].clone(), e1['conf'].clone(), e2['pts3d_in_other_view'].clone(), e2['conf'].clone()))
            eng._destroy_engine()
        for a, b in zip(outs['0'], outs['64']):
            assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_layernorm_statistics_in_the_consumer_prologue_are_bit_identical(gpu, monkeypatch):
    """D3R_LN_INLINE_ROWS=n at engine creation (default 3072: calls of one or two pairs): folded LayerNorms of at most n rows get rstd / -mean rstd from the consumer GEMM's prologue
    instead of an ln_finalize launch. The prologue repeats that kernel's arithmetic (kernels.hpp ln_row_stats: its 32-lane fp64 butterfly as a binary tree, 1 / 2 / 4 adjacent lanes
    per row by tile), so outputs are bit-identical -- which is what keeps a batch (launch route) bit-equal to its one-pair calls (prologue route)."""
    from oracle.dust3r_ref import build_ref_model
    oracle = _randomize_norms(build_ref_model('tiny_dpt'), seed=5)
    monkeypatch.setenv('D3R_LN_FOLD', '1')
    outs = {}
    g = torch.Generator().manual_seed(8)
    cases = [synthetic_views(3, 128, 128, seed=2), synthetic_views(2, 32, 48, seed=3),
             (dict(img=torch.rand((2, 3, 64, 128), generator=g) * 2 - 1, true_shape=torch.tensor([(64, 128)] * 2, dtype=torch.int32), idx=[0, 2], instance=['0', '2']),
              dict(img=torch.rand((2, 3, 128, 64), generator=g) * 2 - 1, true_shape=torch.tensor([(128, 64)] * 2, dtype=torch.int32), idx=[1, 3], instance=['1', '3']))]
    for rows in ('0', '1000000'):
        monkeypatch.setenv('D3R_LN_INLINE_ROWS', rows)
        eng = engine_from_oracle(oracle, 'tiny_dpt', 'fp16x3', gpu)
        outs[rows] = []
        for v1, v2 in cases:
            e1, e2 = eng(v1, v2)
            outs[rows].append((e1['pts3d'].clone(), e2['pts3d_in_other_view'].clone(), e2['conf'].clone()))
        eng._destroy_engine()
    for a, b in zip(outs['0'], outs['1000000']):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_cross_attention_kv_projected_ahead_is_bit_identical(gpu, monkeypatch):
    """D3R_DEC_KV_AHEAD=rows at engine creation: the decoder blocks' cross-attention K | V projection (its input is the OTHER side's previous-layer output, not the
    side's own chain) runs on a third / fourth stream beside the self attention, into its own K / V^T buffers. Same kernels, same rows: bit-identical to the in-line
    schedule -- equal views, a ragged token count (96 tokens: V^T rows padded to 128), views of two sizes (Nq != Nk), and repeated calls on one engine (buffer reuse
    across the layer boundaries and across calls)."""
    from oracle.dust3r_ref import build_ref_model
    from conftest import need_probes
    need_probes('D3R_DEC_KV_AHEAD (a schedule that never became the default)')
    oracle = _randomize_norms(build_ref_model('tiny_dpt'), seed=6)
    g = torch.Generator().manual_seed(9)
    cases = [synthetic_views(3, 128, 128, seed=2), synthetic_views(2, 128, 192, seed=3), synthetic_views(1, 64, 64, seed=4),
             (dict(img=torch.rand((2, 3, 64, 128), generator=g) * 2 - 1, true_shape=torch.tensor([(64, 128)] * 2, dtype=torch.int32), idx=[0, 2], instance=['0', '2']),
              dict(img=torch.rand((2, 3, 128, 64), generator=g) * 2 - 1, true_shape=torch.tensor([(128, 64)] * 2, dtype=torch.int32), idx=[1, 3], instance=['1', '3']))]
    outs = {}
    for rows in ('0', '1000000'):
        monkeypatch.setenv('D3R_DEC_KV_AHEAD', rows)
        eng = engine_from_oracle(oracle, 'tiny_dpt', 'fp16x3', gpu)
        outs[rows] = []
        for rep in range(2):
            for v1, v2 in cases:
                e1, e2 = eng(v1, v2)
                outs[rows].append((e1['pts3d'].clone(), e1['conf'].clone(), e2['pts3d_in_other_view'].clone(), e2['conf'].clone()))
        torch.cuda.synchronize()
        eng._destroy_engine()
    for a, b in zip(outs['0'], outs['1000000']):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    n = len(cases)
    for i in range(n):       # and the second round of calls reproduces the first
        assert all(torch.equal(x, y) for x, y in zip(outs['1000000'][i], outs['1000000'][n + i]))
