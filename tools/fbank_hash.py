"""Hashes of the fbank and fused-frontend outputs on a fixed ragged batch: equal hashes = bit-identical kernels (A/B of experiment builds)."""
import hashlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import toolbox_for_asr_and_tts_amd as pkg
from toolbox_for_asr_and_tts_amd import _lib
if len(sys.argv) > 1:  # path of another build of libasrfe.so
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from toolbox_for_asr_and_tts_amd.synth import bench_pcm
fe = pkg.WavFrontend()
pcm = torch.from_numpy(bench_pcm(16, 480000, seed=5)).cuda()
out, nfr = fe.forward_fbank(pcm, [480000 - 137 * i for i in range(16)])
f, l = fe.forward(pcm, [480000 - 137 * i for i in range(16)])
torch.cuda.synchronize()
print("fbank", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], "frontend", hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest()[:16])
