"""Host logic of the BPTT chunk schedule (engine.EncoderBackwardPlan.chunk_schedule): no GPU needed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))

from arcvae_hip.engine import EncoderBackwardPlan  # noqa: E402


@pytest.mark.parametrize("T,L", [(128, 2), (12, 2), (9, 2), (5, 1), (2, 2), (1, 1), (40, 4), (3, 3)])
@pytest.mark.parametrize("fractions", [(0.3, 0.6, 0.85, 1.0), (0.63, 1.0), (1.0,), (0.1, 0.2, 0.3, 0.5, 0.7, 0.9, 1.0)])
def test_chunks_tile_the_sweep_and_the_time_axis(T, L, fractions):
    S = T + 2 * (L - 1)
    chunks = EncoderBackwardPlan.chunk_schedule(T, L, fractions)
    assert chunks[0][0] == 0 and chunks[-1][1] == S
    assert chunks[0][4] and chunks[-1][5] and sum(c[5] for c in chunks) == 1
    firsts = [c[4] for c in chunks]            # `first` (zero the token-table workspace) may repeat, but only while
    nf = sum(firsts)                           # nothing has been accumulated: a prefix of chunks with empty time ranges
    assert firsts[:nf] == [True] * nf and all(c[2] == c[3] == T for c in chunks[:nf - 1])
    t_next = T
    for (s0, s1, t_lo, t_hi, first, last), nxt in zip(chunks, chunks[1:] + [None]):
        assert s0 < s1 and 0 <= t_lo <= t_hi <= T and t_hi == t_next
        if nxt is not None:
            assert nxt[0] == s1
        # layer l handles time t at tick (T-1-t) + 2(L-1-l): the bottom layer is the last to reach t_lo
        if not last and t_lo < T:
            assert (T - 1 - t_lo) + 2 * (L - 1) <= s1 - 1
        t_next = t_lo
    assert t_next == 0


def test_default_schedules():
    assert [c[:2] for c in EncoderBackwardPlan.chunk_schedule(128, 2, (0.3, 0.6, 0.85, 1.0))] == [(0, 39), (39, 78), (78, 110), (110, 130)]
    assert [c[:4] for c in EncoderBackwardPlan.chunk_schedule(128, 2, (0.63, 1.0))] == [(0, 82, 48, 128), (82, 130, 0, 48)]
