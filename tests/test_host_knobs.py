"""Knob defaults of the host side (climate2weather_amd/_lib.py::HOST_KNOB_DEFAULTS): applied before the library reads its environment, never over an
explicit setting, and re-applied by ops.knobs_reload() after a test deleted a variable.  CPU tier: no library call."""
import os

from climate2weather_amd import _lib


def test_host_knob_defaults_fill_in_and_never_override(monkeypatch):
    assert _lib.HOST_KNOB_DEFAULTS.get("C2W_CONV_S2_PATCH") == "0"  # the stride-2 forward kernel is opt-in (profiles/r06_experiments.md section 10d)
    monkeypatch.delenv("C2W_CONV_S2_PATCH", raising=False)
    _lib.apply_host_knob_defaults()
    assert os.environ["C2W_CONV_S2_PATCH"] == "0"
    monkeypatch.setenv("C2W_CONV_S2_PATCH", "2")
    _lib.apply_host_knob_defaults()
    assert os.environ["C2W_CONV_S2_PATCH"] == "2"


def test_the_library_reads_the_knob_the_host_sets():
    """csrc/conv_igemm.hip reads C2W_CONV_S2_PATCH with getenv (default 1 when unset): the host's default only works if the names agree."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(here, "climate2weather_amd", "csrc", "conv_igemm.hip")).read()
    for name in _lib.HOST_KNOB_DEFAULTS:
        assert f'getenv("{name}")' in src, name
