"""ORACLE (test infrastructure only) -- restatement of naver/croco
`models/dpt_block.py::DPTOutputAdapter` (itself derived from MultiMAE / MiDaS),
the base class of the reference's `DPTOutputAdapter_fix`
(`dust3r/heads/dpt_head.py:20-65`).

PARITY UNPINNED (croco submodule absent). Follows SURVEY.md Appendix A.5 and
the attributes the reference's subclass reads: `dim_tokens_enc, image_size,
stride_level, P_H, P_W, hooks, adapt_tokens, act_postprocess, scratch.layer_rn,
scratch.refinenet{1..4}, head, act_{1..4}_postprocess`.

One value-affecting recollection is kept as a switch: `RELU_INPLACE`. The public
code builds the fusion blocks with `nn.ReLU(False)` (not in place), so the
residual unit's skip adds the UN-activated input. The HIP engine follows the
same switch (`D3R_DPT_SKIP_RELU` in dust3r_amd/csrc/engine.hip).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

RELU_INPLACE = False


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class ResidualConvUnit_custom(nn.Module):
    def __init__(self, features, activation, bn=False):
        super().__init__()
        assert not bn
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.activation = activation

    def forward(self, x):
        out = self.activation(x)
        out = self.conv1(out)
        out = self.activation(out)
        out = self.conv2(out)
        return out + x


class FeatureFusionBlock_custom(nn.Module):
    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True,
                 width_ratio=1):
        super().__init__()
        assert width_ratio == 1 and not expand and not deconv
        self.align_corners = align_corners
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.resConfUnit1 = ResidualConvUnit_custom(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit_custom(features, activation, bn)

    def forward(self, *xs):
        output = xs[0]
        if len(xs) == 2:
            output = output + self.resConfUnit1(xs[1])
        output = self.resConfUnit2(output)
        output = F.interpolate(output, scale_factor=2, mode='bilinear', align_corners=self.align_corners)
        return self.out_conv(output)


def make_fusion_block(features, use_bn, width_ratio=1):
    return FeatureFusionBlock_custom(features, nn.ReLU(RELU_INPLACE), deconv=False, bn=use_bn, expand=False,
                                     align_corners=True, width_ratio=width_ratio)


def make_scratch(in_shape, out_shape):
    scratch = nn.Module()
    scratch.layer1_rn = nn.Conv2d(in_shape[0], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    scratch.layer2_rn = nn.Conv2d(in_shape[1], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    scratch.layer3_rn = nn.Conv2d(in_shape[2], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    scratch.layer4_rn = nn.Conv2d(in_shape[3], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    scratch.layer_rn = nn.ModuleList([scratch.layer1_rn, scratch.layer2_rn, scratch.layer3_rn, scratch.layer4_rn])
    return scratch


class Interpolate(nn.Module):
    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners

    def forward(self, x):
        return F.interpolate(x, scale_factor=self.scale_factor, mode=self.mode, align_corners=self.align_corners)


class DPTOutputAdapter(nn.Module):
    def __init__(self, num_channels=1, stride_level=1, patch_size=16, main_tasks=('rgb',), hooks=(2, 5, 8, 11),
                 layer_dims=(96, 192, 384, 768), feature_dim=256, last_dim=32, use_bn=False, dim_tokens_enc=None,
                 head_type='regression', output_width_ratio=1, **kwargs):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size = _pair(patch_size)
        self.main_tasks = main_tasks
        self.hooks = list(hooks)
        self.layer_dims = list(layer_dims)
        self.feature_dim = feature_dim
        self.dim_tokens_enc = dim_tokens_enc * len(self.main_tasks) if dim_tokens_enc is not None else None
        self.head_type = head_type
        self.image_size = None

        self.P_H = max(1, self.patch_size[0] // stride_level)
        self.P_W = max(1, self.patch_size[1] // stride_level)

        self.scratch = make_scratch(self.layer_dims, feature_dim)
        self.scratch.refinenet1 = make_fusion_block(feature_dim, use_bn, output_width_ratio)
        self.scratch.refinenet2 = make_fusion_block(feature_dim, use_bn, output_width_ratio)
        self.scratch.refinenet3 = make_fusion_block(feature_dim, use_bn, output_width_ratio)
        self.scratch.refinenet4 = make_fusion_block(feature_dim, use_bn, output_width_ratio)

        assert head_type == 'regression'
        self.head = nn.Sequential(
            nn.Conv2d(feature_dim, feature_dim // 2, kernel_size=3, stride=1, padding=1),
            Interpolate(scale_factor=2, mode='bilinear', align_corners=True),
            nn.Conv2d(feature_dim // 2, last_dim, kernel_size=3, stride=1, padding=1),
            nn.ReLU(True),
            nn.Conv2d(last_dim, self.num_channels, kernel_size=1, stride=1, padding=0))

        if self.dim_tokens_enc is not None:
            self.init(dim_tokens_enc=dim_tokens_enc)

    def init(self, dim_tokens_enc=768):
        if isinstance(dim_tokens_enc, int):
            dim_tokens_enc = 4 * [dim_tokens_enc]
        self.dim_tokens_enc = [dt * len(self.main_tasks) for dt in dim_tokens_enc]
        ld, de = self.layer_dims, self.dim_tokens_enc
        self.act_1_postprocess = nn.Sequential(
            nn.Conv2d(de[0], ld[0], kernel_size=1, stride=1, padding=0),
            nn.ConvTranspose2d(ld[0], ld[0], kernel_size=4, stride=4, padding=0, bias=True))
        self.act_2_postprocess = nn.Sequential(
            nn.Conv2d(de[1], ld[1], kernel_size=1, stride=1, padding=0),
            nn.ConvTranspose2d(ld[1], ld[1], kernel_size=2, stride=2, padding=0, bias=True))
        self.act_3_postprocess = nn.Sequential(
            nn.Conv2d(de[2], ld[2], kernel_size=1, stride=1, padding=0))
        self.act_4_postprocess = nn.Sequential(
            nn.Conv2d(de[3], ld[3], kernel_size=1, stride=1, padding=0),
            nn.Conv2d(ld[3], ld[3], kernel_size=3, stride=2, padding=1))
        self.act_postprocess = nn.ModuleList([
            self.act_1_postprocess, self.act_2_postprocess, self.act_3_postprocess, self.act_4_postprocess])

    def adapt_tokens(self, encoder_tokens):
        return encoder_tokens
