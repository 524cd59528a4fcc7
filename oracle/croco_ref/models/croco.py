"""ORACLE (test infrastructure only) -- restatement of naver/croco
`models/croco.py::CroCoNet`, the base class of the reference's
`AsymmetricCroCo3DStereo` (`dust3r/model.py:46-73`).

PARITY UNPINNED (croco submodule absent). Follows SURVEY.md Appendix A.1 and the
attributes the reference reads: `enc_blocks, enc_norm, enc_pos_embed,
decoder_embed, dec_blocks, dec_norm, mask_token, dec_depth, enc_embed_dim,
dec_embed_dim, patch_embed` (`dust3r/model.py:68-73,104-105,128-191`,
`dust3r/heads/dpt_head.py:100-106`).
"""
from functools import partial

import torch
import torch.nn as nn

from models.blocks import Block, DecoderBlock, PatchEmbed
from models.pos_embed import RoPE2D


class CroCoNet(nn.Module):
    def __init__(self,
                 img_size=224, patch_size=16, mask_ratio=0.9,
                 enc_embed_dim=768, enc_depth=12, enc_num_heads=12,
                 dec_embed_dim=512, dec_depth=8, dec_num_heads=16,
                 mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 norm_im2_in_dec=True, pos_embed='cosine'):
        super().__init__()
        self._set_patch_embed(img_size, patch_size, enc_embed_dim)
        self._set_mask_generator(self.patch_embed.num_patches, mask_ratio)

        self.pos_embed = pos_embed
        if pos_embed.startswith('RoPE'):
            self.enc_pos_embed = None
            self.dec_pos_embed = None
            self.rope = RoPE2D(freq=float(pos_embed[len('RoPE'):]))
        else:
            raise NotImplementedError('only RoPE positional embedding is on the DUSt3R path')

        self.enc_depth = enc_depth
        self.enc_embed_dim = enc_embed_dim
        self.enc_blocks = nn.ModuleList([
            Block(enc_embed_dim, enc_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer, rope=self.rope)
            for _ in range(enc_depth)])
        self.enc_norm = norm_layer(enc_embed_dim)

        self._set_mask_token(dec_embed_dim)
        self._set_decoder(enc_embed_dim, dec_embed_dim, dec_num_heads, dec_depth, mlp_ratio, norm_layer,
                          norm_im2_in_dec)
        self._set_prediction_head(dec_embed_dim, patch_size)
        self.initialize_weights()

    def _set_patch_embed(self, img_size=224, patch_size=16, enc_embed_dim=768):
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, enc_embed_dim)

    def _set_mask_generator(self, num_patches, mask_ratio):
        self.mask_generator = None          # masking is a pre-training feature, never used by DUSt3R

    def _set_mask_token(self, dec_embed_dim):
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dec_embed_dim))

    def _set_decoder(self, enc_embed_dim, dec_embed_dim, dec_num_heads, dec_depth, mlp_ratio, norm_layer,
                     norm_im2_in_dec):
        self.dec_depth = dec_depth
        self.dec_embed_dim = dec_embed_dim
        self.decoder_embed = nn.Linear(enc_embed_dim, dec_embed_dim, bias=True)
        self.dec_blocks = nn.ModuleList([
            DecoderBlock(dec_embed_dim, dec_num_heads, mlp_ratio=mlp_ratio, qkv_bias=True, norm_layer=norm_layer,
                         norm_mem=norm_im2_in_dec, rope=self.rope)
            for _ in range(dec_depth)])
        self.dec_norm = norm_layer(dec_embed_dim)

    def _set_prediction_head(self, dec_embed_dim, patch_size):
        self.prediction_head = nn.Linear(dec_embed_dim, patch_size ** 2 * 3, bias=True)

    def initialize_weights(self):
        self.patch_embed._init_weights()
        if self.mask_token is not None:
            torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
