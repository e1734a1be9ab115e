"""asrfe_encoder_forward_static (the entry point that allocates nothing: its scratch is a device array inside the
library) against asrfe_encoder_forward (caller workspace), through the C ABI: identical bits on two calls in a row, the
second smaller than the first, so that what the first call left in the shared array cannot leak into the second."""
import ctypes as C

import pytest
import torch

from oracle import sanm_oracle as so

pytestmark = pytest.mark.gpu  # (the file name sorts last on purpose: a failure here must not hide the other GPU tests)


@pytest.fixture(scope="module")
def enc():
    import toolbox_for_asr_and_tts_amd as pkg

    pkg.load()
    cfg = so.SanmConfig(num_blocks=3)
    e = pkg.SANMEncoder(num_blocks=3)
    e.load_state_dict(so.make_state_dict(cfg, seed=41, dtype=torch.float32))
    return e


def _both(enc, feats, lens):
    ctx = enc.ctx
    b, tp, _ = feats.shape
    x = feats.cuda().contiguous()
    ln = torch.tensor(lens, dtype=torch.int32, device="cuda")
    nbytes = enc.workspace_bytes(b, tp)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda")
    ptr = (ws.data_ptr() + 255) // 256 * 256
    want = torch.full((b, tp, 512), float("nan"), dtype=torch.float32, device="cuda")
    got = torch.full((b, tp, 512), float("nan"), dtype=torch.float32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ctx.check(ctx.lib.asrfe_encoder_forward(ctx.handle, x.data_ptr(), ln.data_ptr(), b, tp, want.data_ptr(), 0, ptr, nbytes,
                                            stream))
    ctx.check(ctx.lib.asrfe_encoder_forward_static(ctx.handle, x.data_ptr(), ln.data_ptr(), b, tp, got.data_ptr(), 0, stream))
    torch.cuda.synchronize()
    return got, want


def test_static_scratch_forward_equals_workspace_forward(enc):
    got, want = _both(enc, so.synth_feats(3, 150, 560, seed=42), [150, 97, 128])
    assert bool(torch.isfinite(want).all()) and torch.equal(got, want)
    # a smaller call right behind it: the array still holds the first call's buffers
    got2, want2 = _both(enc, so.synth_feats(2, 40, 560, seed=43), [40, 27])
    assert bool(torch.isfinite(want2).all()) and torch.equal(got2, want2)
    # and against the oracle, so that "equal" is not "equally wrong"
    ref, _ = so.encoder_forward(so.synth_feats(2, 40, 560, seed=43), torch.tensor([40, 27], dtype=torch.int32),
                                so.make_state_dict(so.SanmConfig(num_blocks=3), seed=41, dtype=torch.float32),
                                so.SanmConfig(num_blocks=3))
    assert float((got2.cpu() - ref).abs().max()) <= 1e-2


def test_static_scratch_on_the_throughput_plan(enc):
    """Above 2048 rows a call takes the throughput plan, whose residual stream lives in the array in whole 128-row tiles
    (register order): 9 x 256 = 2304 rows right behind a call with a larger footprint, then 2057 rows (9 x 230 with ragged
    lengths: not a multiple of 128, so the last tile's padding rows hold what the call before left there)."""
    big, _ = _both(enc, so.synth_feats(10, 300, 560, seed=44), [300] * 10)        # 3000 rows: the larger footprint
    assert bool(torch.isfinite(big).all())
    got, want = _both(enc, so.synth_feats(9, 256, 560, seed=45), [256] * 9)
    assert bool(torch.isfinite(want).all()) and torch.equal(got, want)
    lens = [230, 229, 128, 1, 230, 64, 0, 200, 230]
    feats = so.synth_feats(9, 230, 560, seed=46)
    assert 9 * 230 > 2048 and (9 * 230) % 128 != 0
    got2, want2 = _both(enc, feats, lens)
    assert bool(torch.isfinite(want2).all()) and torch.equal(got2, want2)
    ref, _ = so.encoder_forward(feats, torch.tensor(lens, dtype=torch.int32),
                                so.make_state_dict(so.SanmConfig(num_blocks=3), seed=41, dtype=torch.float32),
                                so.SanmConfig(num_blocks=3))
    assert float((got2.cpu() - ref).abs().max()) <= 1e-2


def test_static_scratch_refuses_what_does_not_fit(enc):
    import toolbox_for_asr_and_tts_amd as pkg
    ctx = enc.ctx
    b, tp = 8, 500      # 4000 rows: more than the 32 MiB array holds
    assert enc.workspace_bytes(b, tp) > 32 << 20
    x = torch.zeros(b, tp, 560, device="cuda")
    ln = torch.full((b,), tp, dtype=torch.int32, device="cuda")
    out = torch.zeros(b, tp, 512, device="cuda")
    rc = ctx.lib.asrfe_encoder_forward_static(ctx.handle, x.data_ptr(), ln.data_ptr(), b, tp, out.data_ptr(), 0,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -6
    with pytest.raises(pkg.AsrfeError, match="static array holds 33554432"):
        ctx.check(rc)
    assert float(out.abs().max()) == 0.0  # nothing was launched
