"""Build-container experiment (not product, not a test): how much pointmap error does
16-bit GEMM-input rounding (bf16 vs fp16, fp32 accumulate) cause through the
ViT-L encoder / ViT-B decoder / DPT head with seeded random weights?
Emulates the HIP engine's rounding points on CPU: inputs and weights of every
Linear/Conv and q,k,v,P of every attention are rounded to the 16-bit type; the
residual stream, LayerNorm, softmax and accumulation stay fp32.
Usage: python tools/precision_probe.py [H W]
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

QDT = None


def q(x):
    return x if QDT is None else x.to(QDT).float()


def lin_fwd(self, x):
    return F.linear(q(x), q(self.weight), self.bias)


def conv_fwd(self, x):
    return F.conv2d(q(x), q(self.weight), self.bias, self.stride, self.padding)


def convt_fwd(self, x):
    return F.conv_transpose2d(q(x), q(self.weight), self.bias, self.stride, self.padding)


def attn_core(qq, k, v, scale):
    a = (q(qq) @ q(k).transpose(-2, -1)) * scale
    a = a.softmax(dim=-1)
    return q(a) @ q(v)


def attn_fwd(self, x, xpos):
    B, N, C = x.shape
    qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).transpose(1, 3)
    qq, k, v = [qkv[:, :, i] for i in range(3)]
    qq, k = self.rope(qq, xpos), self.rope(k, xpos)
    x = attn_core(qq, k, v, self.scale).transpose(1, 2).reshape(B, N, C)
    return self.proj(x)


def xattn_fwd(self, query, key, value, qpos, kpos):
    B, Nq, C = query.shape
    H = self.num_heads
    qq = self.projq(query).reshape(B, Nq, H, C // H).permute(0, 2, 1, 3)
    k = self.projk(key).reshape(B, -1, H, C // H).permute(0, 2, 1, 3)
    v = self.projv(value).reshape(B, -1, H, C // H).permute(0, 2, 1, 3)
    qq, k = self.rope(qq, qpos), self.rope(k, kpos)
    x = attn_core(qq, k, v, self.scale).transpose(1, 2).reshape(B, Nq, C)
    return self.proj(x)


nn.Linear.forward = lin_fwd
nn.Conv2d.forward = conv_fwd
nn.ConvTranspose2d.forward = convt_fwd
blocks.Attention.forward = attn_fwd
blocks.CrossAttention.forward = xattn_fwd

inf = float('inf')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 512)
torch.manual_seed(0)
m = AsymmetricCroCo3DStereo(pos_embed='RoPE100', img_size=(512, 512), head_type='dpt', output_mode='pts3d',
                            depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), enc_embed_dim=1024, enc_depth=24,
                            enc_num_heads=16, dec_embed_dim=768, dec_depth=12, dec_num_heads=12,
                            landscape_only=False).eval()
for hd in (m.downstream_head1, m.downstream_head2):      # bring |xyz| to O(1) as a trained head would
    hd.dpt.head[4].weight.data *= 40
torch.manual_seed(1)
v1 = dict(img=torch.rand(1, 3, H, W) * 2 - 1, true_shape=torch.tensor([[H, W]]), idx=[0], instance=['0'])
v2 = dict(img=torch.rand(1, 3, H, W) * 2 - 1, true_shape=torch.tensor([[H, W]]), idx=[1], instance=['1'])
res = {}
for name, dt in (('fp32', None), ('bf16', torch.bfloat16), ('fp16', torch.float16)):
    QDT = dt
    with torch.no_grad():
        r1, r2 = m(v1, v2)
    res[name] = (r1['pts3d'], r1['conf'], r2['pts3d_in_other_view'], r2['conf'])
ref = res['fp32']
print('|pts| mean', ref[0].norm(dim=-1).mean().item(), 'max', ref[0].norm(dim=-1).max().item())
for name in ('bf16', 'fp16'):
    for k, lab in ((0, 'pts1'), (2, 'pts2')):
        e = (res[name][k] - ref[k]).norm(dim=-1) / ref[k].norm(dim=-1).clamp(min=1e-8)
        print(f'{name} {lab}: rel err mean {e.mean():.2e} p99 {e.flatten().quantile(0.99):.2e} max {e.max():.2e}')
    e = ((res[name][1] - ref[1]).abs() / ref[1].abs())
    print(f'{name} conf1: rel err mean {e.mean():.2e} max {e.max():.2e}')
