"""Regenerate vorbis_amd/data/setup_*.bin: run the reference's libvorbisenc +
vorbis_analysis_init (oracle/_ref) and serialise its lookups with the reference-side packer
(integration/vamd_pack_setup.c).  Needs /root/reference (build container only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vorbis_amd", "data")
SETUPS = {"44k_stereo_q4": (2, 44100, 0.4), "44k_stereo_q9": (2, 44100, 0.9), "44k_stereo_q1": (2, 44100, 0.1),
          "44k_mono_q5": (1, 44100, 0.5), "44k_51_q3": (6, 44100, 0.3)}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, (ch, rate, q) in SETUPS.items():
        e = ref.RefEncoder(ch, rate, q)
        blob = e.pack_setup()
        blob.tofile(os.path.join(OUT, "setup_%s.bin" % name))
        print(name, blob.size, "bytes")
