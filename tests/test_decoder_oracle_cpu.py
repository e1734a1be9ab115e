"""The CPU oracle of the CIF predictor + Paraformer decoder (oracle/paraformer_decoder_oracle.py): properties the
published algorithms guarantee, an independent module-style restatement of one decoder block, and the committed
fixture.  The reference holds no fixture for this step (parity unpinned)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import paraformer_decoder_oracle as po

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_cif_conserves_the_weighted_frames_and_counts_tokens():
    g = torch.Generator().manual_seed(0)
    b, t, d = 3, 57, 16
    hidden = torch.randn(b, t, d, generator=g, dtype=torch.float64)
    alphas = torch.rand(b, t, generator=g, dtype=torch.float64) * 0.9
    emb, fires, counts = po.cif(hidden, alphas, 1.0)
    total = alphas.sum(1)
    assert counts.tolist() == torch.floor(total).to(torch.int32).tolist()
    # every fired token carries weight exactly 1; what is left over is the unfired remainder
    weighted = (alphas[:, :, None] * hidden).sum(1)
    remainder = total - torch.floor(total)
    # re-run the scan to get the pending frame: weighted sum = sum of tokens + pending partial frame
    for k in range(b):
        integrate, frame, acc = 0.0, torch.zeros(d, dtype=torch.float64), torch.zeros(d, dtype=torch.float64)
        for i in range(t):
            a = float(alphas[k, i])
            comp = 1.0 - integrate
            integrate += a
            if integrate >= 1.0:
                acc += frame + comp * hidden[k, i]
                frame = (a - comp) * hidden[k, i]
                integrate -= 1.0
            else:
                frame = frame + a * hidden[k, i]
        assert torch.allclose(emb[k, : counts[k]].sum(0), acc, atol=1e-12)
        assert torch.allclose(acc + frame, weighted[k], atol=1e-12)
        assert abs(integrate - float(remainder[k])) < 1e-12


def test_tail_process_puts_the_tail_weight_behind_the_last_frame():
    cfg = po.DecoderConfig()
    hidden = torch.ones(2, 5, 4)
    alphas = torch.tensor([[0.3, 0.3, 0.3, 0.3, 0.3], [0.5, 0.5, 0.0, 0.0, 0.0]])
    lens = torch.tensor([5, 2])
    h, a, tn = po.tail_process(hidden, alphas, lens, cfg)
    assert h.shape == (2, 6, 4) and torch.all(h[:, 5] == 0)
    assert torch.allclose(a[0], torch.tensor([0.3, 0.3, 0.3, 0.3, 0.3, 0.45]))
    assert torch.allclose(a[1], torch.tensor([0.5, 0.5, 0.45, 0.0, 0.0, 0.0]))
    assert tn.tolist() == [1.0, 1.0]


class _Block(torch.nn.Module):
    """One DecoderLayerSANM built from torch modules, written independently of the oracle's functional code."""

    def __init__(self, d, ff, k, heads):
        super().__init__()
        self.norm1, self.norm2, self.norm3 = (torch.nn.LayerNorm(d, eps=1e-12) for _ in range(3))
        self.w_1 = torch.nn.Linear(d, ff)
        self.ffn_norm = torch.nn.LayerNorm(ff, eps=1e-12)
        self.w_2 = torch.nn.Linear(ff, d, bias=False)
        self.fsmn = torch.nn.Conv1d(d, d, k, groups=d, bias=False)
        self.pad = torch.nn.ConstantPad1d(((k - 1) // 2, k - 1 - (k - 1) // 2), 0.0)
        self.linear_q = torch.nn.Linear(d, d)
        self.linear_k_v = torch.nn.Linear(d, 2 * d)
        self.linear_out = torch.nn.Linear(d, d)
        self.heads = heads

    def forward(self, tgt, tgt_mask, memory, memory_mask):
        residual = tgt
        x = self.w_2(self.ffn_norm(torch.relu(self.w_1(self.norm1(tgt)))))
        x = self.norm2(x) * tgt_mask
        y = self.fsmn(self.pad(x.transpose(1, 2))).transpose(1, 2) + x
        x = residual + y * tgt_mask
        residual = x
        q = self.linear_q(self.norm3(x))
        k, v = self.linear_k_v(memory).chunk(2, dim=-1)
        b, n, d = q.shape
        dk = d // self.heads
        split = lambda z: z.reshape(b, -1, self.heads, dk).transpose(1, 2)
        att = F.scaled_dot_product_attention(split(q), split(k), split(v), attn_mask=memory_mask[:, None, None, :] > 0)
        return residual + self.linear_out(att.transpose(1, 2).reshape(b, n, d))


def test_decoder_block_matches_a_module_style_restatement():
    cfg = po.DecoderConfig(num_blocks=1, linear_units=1024)
    sd = {k: v.double() for k, v in po.make_state_dict(cfg, seed=3).items()}
    blk = _Block(512, 1024, 11, 4).double()
    p = "decoder.decoders.0"
    blk.load_state_dict({
        "norm1.weight": sd[f"{p}.norm1.weight"], "norm1.bias": sd[f"{p}.norm1.bias"],
        "norm2.weight": sd[f"{p}.norm2.weight"], "norm2.bias": sd[f"{p}.norm2.bias"],
        "norm3.weight": sd[f"{p}.norm3.weight"], "norm3.bias": sd[f"{p}.norm3.bias"],
        "w_1.weight": sd[f"{p}.feed_forward.w_1.weight"], "w_1.bias": sd[f"{p}.feed_forward.w_1.bias"],
        "ffn_norm.weight": sd[f"{p}.feed_forward.norm.weight"], "ffn_norm.bias": sd[f"{p}.feed_forward.norm.bias"],
        "w_2.weight": sd[f"{p}.feed_forward.w_2.weight"], "fsmn.weight": sd[f"{p}.self_attn.fsmn_block.weight"],
        "linear_q.weight": sd[f"{p}.src_attn.linear_q.weight"], "linear_q.bias": sd[f"{p}.src_attn.linear_q.bias"],
        "linear_k_v.weight": sd[f"{p}.src_attn.linear_k_v.weight"], "linear_k_v.bias": sd[f"{p}.src_attn.linear_k_v.bias"],
        "linear_out.weight": sd[f"{p}.src_attn.linear_out.weight"], "linear_out.bias": sd[f"{p}.src_attn.linear_out.bias"],
    })
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 9, 512, generator=g, dtype=torch.float64)
    mem = torch.randn(2, 23, 512, generator=g, dtype=torch.float64)
    tok_mask = (torch.arange(9)[None, :] < torch.tensor([9, 6])[:, None]).double()
    mem_mask = (torch.arange(23)[None, :] < torch.tensor([23, 17])[:, None]).double()
    with torch.no_grad():
        want = blk(x, tok_mask[:, :, None], mem, mem_mask)
        got = po.decoder_layer(x, tok_mask[:, :, None], mem, mem_mask, sd, p, cfg)
    valid = tok_mask[:, :, None] > 0
    assert float(((got - want) * valid).abs().max()) < 1e-10


def test_decoder_fixture():
    g = np.load(os.path.join(GOLDEN, "decoder_tiny.npz"))
    cfg = po.DecoderConfig(num_blocks=int(g["num_blocks"]), vocab_size=int(g["vocab_size"]))
    sd = {k: v.double() for k, v in po.make_state_dict(cfg, seed=int(g["seed"])).items()}
    mem = torch.from_numpy(g["memory"]).double()
    lens = torch.from_numpy(g["lens"])
    ids, logits, tn, alphas = po.recognize(mem, lens, sd, cfg)
    assert tn.tolist() == g["token_num"].tolist()
    assert np.allclose(alphas.numpy(), g["alphas"], atol=1e-6)
    for k in range(mem.shape[0]):
        n = int(tn[k])
        assert np.allclose(logits[k, :n].numpy(), g["logits"][k, :n], atol=1e-5)
        assert ids[k] == [t for t in g["ids"][k, :n].tolist() if t not in (0, 1, 2)]


def test_fire_frames_are_where_the_integrator_crosses_the_threshold():
    """cif_fire_frames restates the recurrence of cif on its own: its frames are the positions where cif's running sum
    reaches the threshold, and a hand-made sequence fires where it must."""
    cfg = po.DecoderConfig(num_blocks=1, vocab_size=50)
    sd = {k: v.double() for k, v in po.make_state_dict(cfg, seed=5).items()}
    g = torch.Generator().manual_seed(6)
    hidden = torch.randn(3, 40, 512, generator=g, dtype=torch.float64)
    lens = torch.tensor([40, 17, 0], dtype=torch.int32)
    _, token_num, alphas, counts = po.predictor_forward(hidden, lens, sd, cfg)
    ff = po.cif_fire_frames(alphas, cfg.threshold, 41, dtype=torch.float64)
    mask = (torch.arange(40)[None, :] < lens[:, None]).double()
    h, a, _ = po.tail_process(hidden * mask[:, :, None], po.predictor_alphas(hidden, lens, sd, cfg), lens, cfg)
    _, fires, _ = po.cif(h, a, cfg.threshold)
    for k in range(3):
        want = torch.nonzero(fires[k] >= cfg.threshold).reshape(-1).tolist()
        got = [int(v) for v in ff[k] if int(v) >= 0]
        assert got == want and len(got) == int(counts[k])
        assert all(int(v) == -1 for v in ff[k, len(got):])
    hand = torch.tensor([[0.5, 0.6, 0.0, 0.9, 0.2, 1.0, 0.45]])  # sums 0.5 1.1| 0.1 1.0| 0.2 1.2| 0.65
    assert po.cif_fire_frames(hand, 1.0, 4).tolist() == [[1, 3, 5, -1]]
    assert po.cif_fire_frames(hand, 1.0, 2).tolist() == [[1, 3]]
