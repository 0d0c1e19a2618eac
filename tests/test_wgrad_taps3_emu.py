"""The index logic of wgrad_taps3_kernel (csrc/btx_wgrad_taps.h) emulated thread by thread on the CPU — staging pieces, the ring of
input pixels, transpose-read lane addressing, tap validity, MFMA layouts, slab indices — against the definition of the weight
gradient on exact small integers (tools/wgrad_taps3_emu.py).  The GPU parity tests are tests/test_gpu_backward.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_all_taps_wgrad_index_logic_is_exact_on_the_cpu():
    import wgrad_taps3_emu as emu
    # ragged pixel count and two chunks; seven steps in two chunks (the ring wraps); the widest supported row
    for case in [(3, 7, 9, 64, 64, 128), (2, 14, 14, 64, 64, 256), (1, 3, 63, 64, 64, 192)]:
        assert emu.emulate(*case) == 0.0, case
