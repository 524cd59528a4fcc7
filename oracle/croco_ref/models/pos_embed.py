"""ORACLE (test infrastructure only) -- restatement of naver/croco
`models/pos_embed.py::RoPE2D` (the pure-torch fallback that the reference uses
when the `curope` CUDA extension is not compiled) and of the curope kernel's
arithmetic (`models/curope/kernels.cu`).

PARITY UNPINNED (croco submodule absent; restated from the published code).

Convention: head dim D is split in two halves; the first half is rotated by the
token's y coordinate, the second by x. Inside a half of size D/2 the rotation
pairs element i with element i + D/4, angle = pos * base^(-i/(D/4)).
"""
import torch


class RoPE2D(torch.nn.Module):
    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0
        self.cache = {}

    def get_cos_sin(self, D, seq_len, device, dtype):
        key = (D, seq_len, device, dtype)
        if key not in self.cache:
            inv_freq = 1.0 / (self.base ** (torch.arange(0, D, 2).float().to(device) / D))
            t = torch.arange(seq_len, device=device, dtype=inv_freq.dtype)
            freqs = torch.einsum("i,j->ij", t, inv_freq).to(dtype)
            freqs = torch.cat((freqs, freqs), dim=-1)
            self.cache[key] = (freqs.cos(), freqs.sin())
        return self.cache[key]

    @staticmethod
    def rotate_half(x):
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
        return torch.cat((-x2, x1), dim=-1)

    def apply_rope1d(self, tokens, pos1d, cos, sin):
        assert pos1d.ndim == 2
        cos = torch.nn.functional.embedding(pos1d, cos)[:, None, :, :]
        sin = torch.nn.functional.embedding(pos1d, sin)[:, None, :, :]
        return (tokens * cos) + (self.rotate_half(tokens) * sin)

    def forward(self, tokens, positions):
        """tokens (B, H, N, D); positions (B, N, 2) int64 in (y, x) order."""
        assert tokens.size(3) % 2 == 0
        D = tokens.size(3) // 2
        assert positions.ndim == 3 and positions.shape[-1] == 2
        cos, sin = self.get_cos_sin(D, int(positions.max()) + 1, tokens.device, tokens.dtype)
        y, x = tokens.chunk(2, dim=-1)
        y = self.apply_rope1d(y, positions[:, :, 0], cos, sin)
        x = self.apply_rope1d(x, positions[:, :, 1], cos, sin)
        return torch.cat((y, x), dim=-1)


def rope_2d_inplace_ref(tokens, positions, base, F0):
    """Loop-free restatement of curope's `rope_2d(tokens[B,N,H,D], positions[B,N,2],
    base, F0)` (in place): the native op the HIP kernel `d3r_rope2d` replaces.
    """
    B, N, H, D = tokens.shape
    Q = D // 4
    inv_freq = F0 / (base ** (torch.arange(Q, dtype=torch.float32, device=tokens.device) / Q))
    for half in range(2):
        ang = positions[:, :, half].to(torch.float32)[:, :, None, None] * inv_freq          # (B,N,1,Q)
        c, s = ang.cos(), ang.sin()
        u = tokens[..., half * 2 * Q: half * 2 * Q + Q].float().clone()
        v = tokens[..., half * 2 * Q + Q: half * 2 * Q + 2 * Q].float().clone()
        tokens[..., half * 2 * Q: half * 2 * Q + Q] = (u * c - v * s).to(tokens.dtype)
        tokens[..., half * 2 * Q + Q: half * 2 * Q + 2 * Q] = (v * c + u * s).to(tokens.dtype)
    return tokens
