"""Whole-horizon small-policy rollout bodies compiled for the host, against the reference's golden vectors."""
import pytest

import hostsim_util
import small_rollout_checks as src


@pytest.mark.parametrize("name", src.SMALL_CASES)
def test_small_rollout_bodies_match_reference(name):
    h = hostsim_util.load()
    P = src.P

    def fwd(desc, rewards, final, sh, hh, lh):
        h.hostsim_small_rollout_fwd(desc, P(rewards), P(final), P(sh), P(hh), P(lh))

    def bwd(desc, sh, hh, lh, gr, dzh, dzo):
        h.hostsim_small_rollout_bwd(desc, P(sh), P(hh), P(lh), gr, P(dzh), P(dzo))

    src.run_case(name, fwd, bwd, "cpu")
