"""Host logic: how the training plan schedules a weight-gradient launch (train_engine.scheduled_wgrad) — the per-kernel tuner's
choice is measured alone on an idle chip, the step runs the weight gradients beside the frame chains."""
from streamyolo_amd.train_engine import NINE_TAP_WGRAD_TILES, scheduled_wgrad


def test_general_split_k_cap():
    assert scheduled_wgrad((17, 1024), 256, env={}) == (17, 512)
    assert scheduled_wgrad((18, 256), 256, env={}) == (18, 256)
    assert scheduled_wgrad((20, 1024), 96, env={"STREAMYOLO_WGRAD_BLOCKS_CAP": "0"}) == (20, 1024)
    assert scheduled_wgrad((0, 0), 256, env={}) == (0, 0)                  # tuner off: the library's heuristic, untouched


def test_nine_tap_kernel_is_confined_and_widened():
    for t in NINE_TAP_WGRAD_TILES:
        assert scheduled_wgrad((t, 256), 256, env={}) == (60, 96)           # 64-input-channel workgroups, 96 of them
        assert scheduled_wgrad((t, 64), 128, env={}) == (60, 64)            # (fewer than the cap: kept)
        assert scheduled_wgrad((t, 256), 96, env={}) == (t, 128)            # Cin % 64 != 0: the tuner's tile on 128 workgroups
        assert scheduled_wgrad((t, 512), 32, env={}) == (t, 128)


def test_switches():
    env = {"STREAMYOLO_WGRAD9_WIDE": "0"}
    assert scheduled_wgrad((59, 256), 256, env=env) == (59, 128)
    env = {"STREAMYOLO_WGRAD9_WIDE_BLOCKS": "112"}
    assert scheduled_wgrad((52, 256), 512, env=env) == (60, 112)
    env = {"STREAMYOLO_WGRAD9_WIDE": "0", "STREAMYOLO_WGRAD9_BLOCKS": "0"}
    assert scheduled_wgrad((52, 256), 512, env=env) == (52, 256)
