"""GPU suite: the random-signal soak (tests/soak_lib.py) -- more than ten thousand blocks over twelve configurations
through the batch path, every packet and ampmax against the reference's real vorbis_analysis()."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_soak_ten_thousand_random_blocks():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libvorbis_ref.so is not here (it is built where /root/reference exists and travels prebuilt)")
    from tests import soak_lib
    nb = int(os.environ.get("VAMD_SOAK_BLOCKS", "650"))   # 12 configurations x (650 long + 216 short) = 10 392 blocks
    total, bad = soak_lib.run(nb, managed=True, log=lambda *a: None)
    assert total >= (10000 if nb >= 650 else 1)
    assert bad == 0, "%d of %d soak blocks differ from the reference" % (bad, total)
