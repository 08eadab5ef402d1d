"""The hand-derived known answers of known_answers.py on the oracle (so that the checker itself is pinned by them) and on the
host build of the kernels' per-file pipeline code."""
import numpy as np

import known_answers as ka
from helpers import HostSim, Oracle, shipped_config


def test_oracle_gives_the_hand_derived_answers():
    assert ka.check_known_answers(Oracle, shipped_config) == 6


def test_kernel_code_on_the_host_gives_the_hand_derived_captions():
    """Captions, facet choices and audit lists of KA1-KA5 from the kernels' per-file code, compiled for the host."""
    sim = HostSim(shipped_config())
    for run in (sim.run_texts, sim.run_texts_fast, sim.run_texts_lean):
        r = run([ka.KA1_LINE])
        assert r.caption(0).decode() == ka.KA1_CAPTION
        assert [int(x) for x in r.facet_choice[0]] == ka.KA1_FACET_CHOICE
    sim = HostSim(shipped_config(vocabulary=None))
    for run in (sim.run_texts, sim.run_texts_fast, sim.run_texts_lean):
        assert run([ka.KA2_LINE]).caption(0).decode() == ka.KA2_CAPTION
    from helpers import pack_lines
    arena, offsets = pack_lines([ka.KA2_LINE])
    lists, ver = sim.run_audit(arena, offsets)
    row = ka.KA2_ROW[1:-1].split('","')
    assert [x.decode() for x in lists[0][:3]] == [row[1], row[2], row[4]]
    arena, toffs = ka.pack_tags(ka.KA4_TAGS)
    begin, src = np.array([0, 3], dtype=np.uint64), np.array(ka.KA4_SOURCES, dtype=np.uint8)
    for run in (sim.run_scored, sim.run_scored_fast):
        assert run(arena, toffs, begin, np.array(ka.KA4A_SCORES), src).caption(0).decode() == ka.KA4A_CAPTION
        assert run(arena, toffs, begin, np.array(ka.KA4B_SCORES), src).caption(0).decode() == ka.KA4B_CAPTION
    arena, toffs = ka.pack_tags(ka.KA5_TAGS)
    r = sim.run_merged(arena, toffs, np.array([0, 4], dtype=np.uint64), np.array([2], dtype=np.uint64), np.array(ka.KA5_SCORES), 15)
    assert r.caption(0).decode() == ka.KA5_CAPTION
    r = sim.run_scored_fast(arena, toffs, np.array([0, 4], dtype=np.uint64), np.array(ka.KA5_SCORES), None, secondary_begin=np.array([2], dtype=np.uint64), secondary_cap=15)
    assert r.caption(0).decode() == ka.KA5_CAPTION
    sim = HostSim(shipped_config(strict_known=False, strict_known_explicit=True))
    for run in (sim.run_texts, sim.run_texts_fast, sim.run_texts_lean):
        assert run([ka.KA3_LINE]).caption(0).decode() == ka.KA3_CAPTION
