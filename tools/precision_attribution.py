"""Build-container experiment (not product, not a test): WHICH contractions of the forward need the split-fp16 (hi, lo) operands to stay
inside the 1e-3 pointmap bar?  Per group of layers, round the GEMM / convolution / attention operands of that group ONLY to fp16
(fp32 accumulate, everything else exact fp32) and measure the per-pixel relative pointmap error of the full 512x384 forward against the
all-fp32 run, on the oracle (the unmodified reference model files + oracle/shims, seeded random weights, CPU).
Variants per group: both operands fp16 ('both'), weights only ('w'), activations only ('a').
Usage: python tools/precision_attribution.py [H W]            (a few minutes on 8 cores)
"""
import sys
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, '.')
from oracle.ref_import import import_reference  # noqa
import_reference()
from dust3r.model import AsymmetricCroCo3DStereo  # noqa
import models.blocks as blocks  # noqa

ACTIVE = {}          # group name -> 'both' | 'w' | 'a'


def rounding(mod):
    return ACTIVE.get(getattr(mod, '_pgroup', None))


def qa(mod, x):
    return x.half().float() if rounding(mod) in ('both', 'a') else x


def qw(mod, w):
    return w.half().float() if rounding(mod) in ('both', 'w') else w


nn.Linear.forward = lambda self, x: F.linear(qa(self, x), qw(self, self.weight), self.bias)
nn.Conv2d.forward = lambda self, x: F.conv2d(qa(self, x), qw(self, self.weight), self.bias, self.stride, self.padding)
nn.ConvTranspose2d.forward = lambda self, x: F.conv_transpose2d(qa(self, x), qw(self, self.weight), self.bias, self.stride, self.padding)


def attn_core(mod, qq, k, v, scale):
    r = ACTIVE.get(getattr(mod, '_agroup', None))
    h = (lambda t: t.half().float()) if r in ('both', 'qk', 'pv') else (lambda t: t)
    hq = h if r in ('both', 'qk') else (lambda t: t)
    hp = h if r in ('both', 'pv') else (lambda t: t)
    a = ((hq(qq) @ hq(k).transpose(-2, -1)) * scale).softmax(dim=-1)
    return hp(a) @ hp(v)


def attn_fwd(self, x, xpos):
    B, N, C = x.shape
    qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).transpose(1, 3)
    qq, k, v = [qkv[:, :, i] for i in range(3)]
    qq, k = self.rope(qq, xpos), self.rope(k, xpos)
    return self.proj(attn_core(self, qq, k, v, self.scale).transpose(1, 2).reshape(B, N, C))


def xattn_fwd(self, query, key, value, qpos, kpos):
    B, Nq, C = query.shape
    Hh = self.num_heads
    qq = self.projq(query).reshape(B, Nq, Hh, C // Hh).permute(0, 2, 1, 3)
    k = self.projk(key).reshape(B, -1, Hh, C // Hh).permute(0, 2, 1, 3)
    v = self.projv(value).reshape(B, -1, Hh, C // Hh).permute(0, 2, 1, 3)
    qq, k = self.rope(qq, qpos), self.rope(k, kpos)
    return self.proj(attn_core(self, qq, k, v, self.scale).transpose(1, 2).reshape(B, Nq, C))


blocks.Attention.forward = attn_fwd
blocks.CrossAttention.forward = xattn_fwd

inf = float('inf')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 512)
torch.manual_seed(0)
m = AsymmetricCroCo3DStereo(pos_embed='RoPE100', img_size=(512, 512), head_type='dpt', output_mode='pts3d',
                            depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), enc_embed_dim=1024, enc_depth=24,
                            enc_num_heads=16, dec_embed_dim=768, dec_depth=12, dec_num_heads=12,
                            landscape_only=False).eval()
for hd in (m.downstream_head1, m.downstream_head2):      # bring |xyz| to O(1) as a trained head would (as oracle/dust3r_ref.py does)
    hd.dpt.head[4].weight.data *= 40


def group_of(name):
    if name.startswith(('patch_embed', 'enc_blocks')):
        return 'enc'
    if name.startswith(('decoder_embed', 'dec_blocks')):
        return 'dec'
    if 'downstream_head' in name:
        if '.act_postprocess' in name or '.scratch.layer' in name:
            return 'head.reassemble'
        if '.scratch.refinenet' in name:
            return 'head.refine'
        if '.dpt.head.' in name:
            idx = int(name.split('.dpt.head.')[1].split('.')[0])
            return 'head.conv1' if idx == 0 else ('head.conv2' if idx == 2 else 'head.out')
        return 'head.other'
    return None


for name, mod in m.named_modules():
    if isinstance(mod, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
        mod._pgroup = group_of(name)
    if isinstance(mod, (blocks.Attention, blocks.CrossAttention)):
        mod._agroup = ('enc' if name.startswith('enc') else 'dec') + '.attn'

QK_GAIN = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0     # > 1: sharpen every attention (logits x gain^2), as trained weights would
if QK_GAIN != 1.0:
    for name, mod in m.named_modules():
        if isinstance(mod, blocks.Attention):
            C = mod.qkv.weight.shape[1]
            mod.qkv.weight.data[:2 * C] *= QK_GAIN
        if isinstance(mod, blocks.CrossAttention):
            mod.projq.weight.data *= QK_GAIN
            mod.projk.weight.data *= QK_GAIN
torch.manual_seed(1)
v1 = dict(img=torch.rand(1, 3, H, W) * 2 - 1, true_shape=torch.tensor([[H, W]]), idx=[0], instance=['0'])
v2 = dict(img=torch.rand(1, 3, H, W) * 2 - 1, true_shape=torch.tensor([[H, W]]), idx=[1], instance=['1'])


def run(active):
    ACTIVE.clear()
    ACTIVE.update(active)
    with torch.no_grad():
        r1, r2 = m(v1, v2)
    return torch.cat((r1['pts3d'], r2['pts3d_in_other_view']))


ref = run({})
print(f'|pts| mean {ref.norm(dim=-1).mean():.3f} min {ref.norm(dim=-1).min():.3e}')
HEAD = ['head.reassemble', 'head.refine', 'head.conv1', 'head.conv2', 'head.out', 'head.other']
cases = [
    ('everything fp16 (the fp16 fast mode)', {g: 'both' for g in ['enc', 'dec', 'enc.attn', 'dec.attn'] + HEAD}),
    ('encoder linears', {'enc': 'both'}), ('encoder linears, weights only', {'enc': 'w'}), ('encoder linears, activations only', {'enc': 'a'}),
    ('encoder attention QK^T', {'enc.attn': 'qk'}), ('encoder attention PV', {'enc.attn': 'pv'}),
    ('decoder linears', {'dec': 'both'}), ('decoder linears, weights only', {'dec': 'w'}),
    ('decoder attention QK^T', {'dec.attn': 'qk'}), ('decoder attention PV', {'dec.attn': 'pv'}),
    ('DPT head, all convolutions', {g: 'both' for g in HEAD}), ('DPT head, weights only', {g: 'w' for g in HEAD}),
    ('DPT reassemble + layer_rn', {'head.reassemble': 'both'}), ('DPT refinenets', {'head.refine': 'both'}),
    ('DPT head conv1 3x3 256->128', {'head.conv1': 'both'}), ('DPT head conv2 3x3 128->128', {'head.conv2': 'both'}),
    ('DPT head 1x1 128->4', {'head.out': 'both'}),
    ('DPT head except the last 1x1', {g: 'both' for g in HEAD if g != 'head.out'}),
    ('DPT refinenets + conv1 + conv2', {'head.refine': 'both', 'head.conv1': 'both', 'head.conv2': 'both'}),
]
if len(sys.argv) > 4:
    cases = [c for c in cases if 'attention' in c[0] or 'activations only' in c[0] or 'everything' in c[0]]
print(f'{"fp16 operands in":44s} {"max":>9s} {"p99":>9s} {"mean":>9s}')
for label, active in cases:
    e = ((run(active) - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-12)).flatten()
    print(f'{label:44s} {e.max():9.2e} {e.quantile(0.99):9.2e} {e.mean():9.2e}', flush=True)
