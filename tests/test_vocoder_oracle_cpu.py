"""The CPU oracle of the HiFi-GAN generator (oracle/hifigan_oracle.py) against an independent module-style restatement
(torch.nn modules, the published class layout) and the committed fixture.  The reference holds no fixture for the
vocoder (parity unpinned)."""
import os

import numpy as np
import torch

from oracle import hifigan_oracle as ho

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


class _ResBlock(torch.nn.Module):
    def __init__(self, ch, k, dils):
        super().__init__()
        self.convs1 = torch.nn.ModuleList([torch.nn.Conv1d(ch, ch, k, 1, dilation=d, padding=(k * d - d) // 2) for d in dils])
        self.convs2 = torch.nn.ModuleList([torch.nn.Conv1d(ch, ch, k, 1, dilation=1, padding=(k - 1) // 2) for _ in dils])

    def forward(self, x):
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = c2(torch.nn.functional.leaky_relu(c1(torch.nn.functional.leaky_relu(x, 0.1)), 0.1))
            x = xt + x
        return x


class _Generator(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.nk = len(cfg.resblock_kernels)
        self.conv_pre = torch.nn.Conv1d(cfg.n_mels, cfg.initial_channel, 7, 1, padding=3)
        self.ups = torch.nn.ModuleList()
        self.resblocks = torch.nn.ModuleList()
        ch = cfg.initial_channel
        for u, k in zip(cfg.upsample_rates, cfg.upsample_kernels):
            self.ups.append(torch.nn.ConvTranspose1d(ch, ch // 2, k, u, padding=(k - u) // 2))
            ch //= 2
            for rk, d in zip(cfg.resblock_kernels, cfg.resblock_dilations):
                self.resblocks.append(_ResBlock(ch, rk, d))
        self.conv_post = torch.nn.Conv1d(ch, 1, 7, 1, padding=3)

    def forward(self, x):
        x = self.conv_pre(x)
        for i, up in enumerate(self.ups):
            x = up(torch.nn.functional.leaky_relu(x, 0.1))
            x = sum(self.resblocks[i * self.nk + j](x) for j in range(self.nk)) / self.nk
        return torch.tanh(self.conv_post(torch.nn.functional.leaky_relu(x))).squeeze(1)


def test_oracle_matches_module_style_generator():
    cfg = ho.HifiganConfig(initial_channel=64, upsample_rates=(4, 2), upsample_kernels=(8, 4))
    sd = {k: v.double() for k, v in ho.make_state_dict(cfg, seed=1).items()}
    gen = _Generator(cfg).double()
    gen.load_state_dict(sd)
    mel = torch.randn(2, 80, 21, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    with torch.no_grad():
        want = gen(mel)
    got = ho.generator_forward(mel, sd, cfg)
    assert got.shape == (2, 21 * 8) and float((got - want).abs().max()) < 1e-12


def test_vocoder_fixture():
    g = np.load(os.path.join(GOLDEN, "vocoder_tiny.npz"))
    cfg = ho.HifiganConfig(initial_channel=int(g["initial_channel"]), upsample_rates=tuple(g["rates"].tolist()),
                           upsample_kernels=tuple(g["kernels"].tolist()))
    sd = {k: v.double() for k, v in ho.make_state_dict(cfg, seed=int(g["seed"])).items()}
    wav = ho.generator_forward(torch.from_numpy(g["mel"]).double(), sd, cfg)
    assert np.allclose(wav.numpy(), g["wav"], atol=1e-6)
