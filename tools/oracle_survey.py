"""GPU experiment (not product, not a test): the error of the engine's precision modes against the CPU ORACLE (the fp32 restatement of the
reference's forward, oracle/dust3r_ref.py) on the BASELINE model at 512x384, over several weight seeds AND on weights with two traits of
trained ViTs (sharp attention, outlier channels x40 / x150: tests/test_forward_gpu.py). Error measure: SURVEY.md 8(d),
||pts_hip - pts_ref||_2 / max(||pts_ref||_2, eps) per pixel with eps = 1e-8 (no floor), both views of one pair; reported: max, 99.99th
and 99th percentile, mean, and the smallest |pts| relative to the mean (how close the random network's pointmap comes to the origin,
where the ratio is ill-conditioned for any arithmetic).
Usage (GPU box): python tools/oracle_survey.py [n_seeds] [ln]      (about 5 s per weight set: one CPU oracle forward + three engine modes)
`ln` (round 5): every LayerNorm weight 1 + 0.3 N(0, 1) and bias 0.2 N(0, 1) instead of the identity affine of build_ref_model_fast, so that the fold of the blocks' LayerNorms
into the GEMMs around them (D3R_LN_FOLD=0 in the environment = the LayerNorm kernels) is measured on non-trivial gamma / beta."""
import sys
import time

import torch

sys.path.insert(0, '.')
from dust3r_amd.model import AsymmetricCroCo3DStereo  # noqa: E402
from dust3r_amd.synthetic import MODEL_CONFIGS, synthetic_views  # noqa: E402
from oracle import tune_threads  # noqa: E402
from oracle.dust3r_ref import build_ref_model_fast  # noqa: E402

MODEL = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
RANDOM_LN = 'ln' in sys.argv


def random_ln(model, seed):
    if not RANDOM_LN:
        return model
    g = torch.Generator().manual_seed(1000 + seed)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if 'norm' in name and prm.ndim == 1:
                prm.copy_(1 + 0.3 * torch.randn(prm.shape, generator=g) if name.endswith('weight') else 0.2 * torch.randn(prm.shape, generator=g))
    return model
dev = torch.device('cuda:0')
tune_threads()
MODES = ('fp32', 'fp16x3')


def outlier_weights(oracle, big):
    with torch.no_grad():
        for name, p in oracle.named_parameters():
            if name.endswith('attn.qkv.weight'):
                p[:2 * p.shape[1]] *= 2.0
            elif name.endswith('cross_attn.projq.weight') or name.endswith('cross_attn.projk.weight'):
                p *= 2.0
            elif 'blocks' in name and (name.endswith('.norm2.weight') and 'enc_blocks' in name or name.endswith('.norm3.weight')):
                p[5::97] *= 8.0
                p[3] *= big


def survey(tag, oracle, view_seed, worst):
    v1, v2 = synthetic_views(1, 384, 512, seed=view_seed)
    t = time.time()
    with torch.no_grad():
        r1, r2 = oracle(v1, v2)
    t_or = time.time() - t
    ref = torch.cat((r1['pts3d'], r2['pts3d_in_other_view']))
    nrm = ref.norm(dim=-1).clamp_min(1e-8)
    eng = AsymmetricCroCo3DStereo(precision='fp32', landscape_only=False, **MODEL_CONFIGS[MODEL])
    eng.load_state_dict(oracle.state_dict(), strict=True)
    eng.to(dev)
    for prec in MODES:
        eng.set_precision(prec)
        e1, e2 = eng(v1, v2)
        got = torch.cat((e1['pts3d'], e2['pts3d_in_other_view'])).cpu()
        rel = ((got - ref).norm(dim=-1) / nrm).flatten()
        srt = rel.sort().values
        q = lambda f: float(srt[min(int(f * srt.numel()), srt.numel() - 1)])   # noqa: E731
        print(f'{tag:26s} {prec:7s} max {float(rel.max()):9.3e}  p99.99 {q(0.9999):9.3e}  p99 {q(0.99):9.3e}  mean {float(rel.mean()):9.3e}   '
              f'min |pts| / mean |pts| {float(nrm.min() / nrm.mean()):.2e}   (oracle {t_or:.1f} s)', flush=True)
        for k, v in (('max', float(rel.max())), ('p99.99', q(0.9999)), ('p99', q(0.99)), ('mean', float(rel.mean()))):
            worst[(prec, k)] = max(worst.get((prec, k), 0.0), v)
    del eng
    torch.cuda.empty_cache()


import os  # noqa: E402
print(f'{MODEL}, one 512x384 pair per weight set, engine vs the CPU oracle; eps = 1e-8; random LayerNorm affines: {RANDOM_LN}; D3R_LN_FOLD={os.environ.get("D3R_LN_FOLD", "(default: on)")}')
worst_plain, worst_out = {}, {}
for seed in range(n_seeds):
    survey(f'random weights, seed {seed}', random_ln(build_ref_model_fast(MODEL, seed=seed), seed), 100 + seed, worst_plain)
for big in (40.0, 150.0):
    oracle = random_ln(build_ref_model_fast(MODEL, seed=0), 0)
    outlier_weights(oracle, big)
    survey(f'sharp attn + outliers x{big:g}', oracle, 3, worst_out)
for name, w in (('random weights', worst_plain), ('sharp attention + outlier channels', worst_out)):
    print(f'WORST over {name}: ' + '; '.join(f"{p}: " + ' '.join(f'{k} {w[(p, k)]:.2e}' for k in ('max', 'p99.99', 'p99', 'mean')) for p in MODES))
