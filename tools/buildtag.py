"""Tag of the kernel sources a measurement was taken with: sha256 over the files of csrc/ that the frontend and encoder
kernels are built from and include/asrfe.h (csrc_tag), and over ALL of csrc/ (side_tag: the decoder head and the vocoder
also run the general GEMM and the attention kernel, so their whole-program counters belong to the whole directory).  Counter profiles under profiles/ record it; bench.py only quotes a profile whose tag equals that of
the tree it runs from, so numbers of an older build cannot be mistaken for this one's."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_tag() -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "toolbox-for-asr-and-tts_amd", "csrc", "*")))
    files.append(os.path.join(ROOT, "include", "asrfe.h"))
    for f in files:
        if os.path.basename(f).startswith(("decoder", "vocoder", "vad")):
            continue
        if os.path.isfile(f) and f.endswith((".hip", ".cpp", ".h", "Makefile")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def side_tag() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "toolbox-for-asr-and-tts_amd", "csrc", "*"))):
        if os.path.isfile(f) and f.endswith((".hip", ".cpp", ".h", "Makefile")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_tag(), side_tag())
