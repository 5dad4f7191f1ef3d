"""The LDS pitch of the MFMA operand tiles (YK_LDPAD in csrc/yk_conv.hip) against the documented ds_read_b128 lane-group model."""
import importlib.util
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location('lds_sim', ROOT / 'tools' / 'lds_sim.py')
lds_sim = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lds_sim)


def _ldpad():
    src = (ROOT / 'k210_yolo_framework_amd' / 'csrc' / 'yk_conv.hip').read_text()
    return int(re.search(r'#define YK_LDPAD (\d+)', src).group(1))


def test_igemm_tile_pitch_is_conflict_free_and_the_old_one_was_not():
    pad = _ldpad()
    for bk in (32, 64):
        assert lds_sim.worst_over_ksteps(bk + pad, bk) == 4
        assert lds_sim.worst_over_ksteps(bk + 8, bk) == 8          # the pitch used before: 2-way conflicts


def test_fused_block_pitches_are_conflict_free_for_every_channel_count_in_the_four_networks():
    pad = _ldpad()
    for cin in (16, 24, 32, 48, 64, 96, 128, 144, 192, 256, 384, 512, 576, 768, 960, 1024):
        kp = (cin + 31) // 32 * 32
        assert lds_sim.worst_over_ksteps(kp + pad, kp) == 4, cin


def test_unpadded_tiles_need_and_have_an_xor_swizzle():
    assert lds_sim.worst_over_ksteps(64, 64) == 16 and lds_sim.worst_over_ksteps(32, 32) == 8
    assert lds_sim.worst_over_ksteps(64, 64, lambda r, c: (c ^ r) % 8) == 4
    assert lds_sim.worst_over_ksteps(32, 32, lambda r, c: (c ^ (r >> 1)) % 4) == 4
