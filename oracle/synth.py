"""Synthetic inputs of SURVEY.md App. C — re-exported for the tests; the generators live in imcvt_amd/synth.py
(they are workload generation shared with bench.py, not part of the checker)."""
from imcvt_amd.synth import *            # noqa: F401,F403
from imcvt_amd.synth import _xorshift_stream, _xs, seed0  # noqa: F401
