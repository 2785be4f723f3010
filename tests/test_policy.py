"""CPU: the screen-form policy (sparsifiedkmeans_amd/csrc/policy.h) walked through tables of counters.

The policy decides how much work the next fused call does -- all-exact kernels, plain screen, unconditional or hinted
two-phase form, early or late split, point or step lists, incremental or full sums -- from the counters of the previous
screen call; it never decides a result.  Compiled here with g++ behind a small C harness (tests/native)."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
N, TILES, NR = 1_000_000.0, 3, 13          # a shard of 1e6 points, K = 100 (three tiles after the remainder is carried), s = 51


@pytest.fixture(scope="module")
def pol(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("policy") / "libpolicy.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", os.path.join(HERE, "native", "policy_harness.cpp"), "-o", so])
    L = C.CDLL(so)
    L.pol_new.restype = C.c_void_p
    for f in (L.pol_free, L.pol_reset):
        f.argtypes = [C.c_void_p]
    L.pol_observe.argtypes = [C.c_void_p] + [C.c_double] * 7 + [C.c_int, C.c_int]
    L.pol_next.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.pol_take_hinted_split.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.pol_launched.argtypes = [C.c_void_p] + [C.c_int] * 8
    L.pol_observe_full_opened.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int]
    L.pol_pt_next.argtypes = [C.c_void_p]
    L.pol_blocks_next.argtypes = [C.c_void_p]
    L.pol_few_movers.argtypes = [C.c_void_p, C.c_double]
    L.pol_few_movers_pair.argtypes = [C.c_void_p, C.c_double]
    L.pol_form_on_device.argtypes = [C.c_void_p]
    L.pol_events_direct.argtypes = [C.c_void_p]
    L.pol_sums_by_events.argtypes = [C.c_void_p]
    L.pol_sums_by_full_pass.argtypes = [C.c_void_p]
    L.pol_refresh_due.argtypes = [C.c_void_p, C.c_double]
    L.pol_event_cap.argtypes = [C.c_uint64]
    L.pol_event_cap.restype = C.c_uint64
    L.pol_event_cap_pair.argtypes = [C.c_uint64]
    L.pol_event_cap_pair.restype = C.c_uint64
    return L


class Walk:
    """one shard's policy: call() = 'a fused call is issued' (returns its form), seen() = 'its counters arrived'"""

    def __init__(self, L, no_prune=0, no_hint=0, no_late=0):
        self.L, self.p = L, C.c_void_p(L.pol_new())
        self.sw = (no_prune, no_hint, no_late)
        self.bounds = False                  # the library holds bounds from a previous screen call

    def call(self, sums=None, both=False):
        """one fused call: the policy's choice, then -- in the library's order -- how the call gets its sums ("events" /
        "full" / None: not said), then what was launched"""
        out = (C.c_int * 3)()
        self.L.pol_next(self.p, self.sw[0], self.sw[1], 1, out)
        exact, prune_a, want_hint = out[0], out[1], out[2]
        if exact:
            self.bounds = False
            return "exact"
        hinted = bool(want_hint and self.bounds and prune_a == 0)
        late = bool(self.L.pol_take_hinted_split(self.p, NR, self.sw[2])) if hinted else False
        rounds_all = (self.L.pol_quad_split_late(NR) if late else self.L.pol_quad_split(NR)) if hinted else (prune_a or NR)
        if sums == "events":
            self.L.pol_sums_by_events(self.p)
        elif sums == "full":
            self.L.pol_sums_by_full_pass(self.p)
        self.L.pol_launched(self.p, rounds_all, NR, int(hinted), int(late), int(self.bounds), int(self.bounds),
                            int(sums == "events"), int(both))
        self.bounds = True
        if hinted:
            return "hinted-late" if late else "hinted-early"
        return "two-phase" if prune_a else "plain"

    def seen(self, listed=0, ambig=0, early=0, skipped=0, kept=0, movers=0):
        self.L.pol_observe(self.p, listed, ambig, early, skipped, kept, movers, N, TILES, NR)


def test_compiled_splits(pol):
    # the step-major copy lists a point's entries by |x| descending: an eighth / a quarter of the rounds
    assert [pol.pol_quad_split(nr) for nr in (1, 2, 3, 8, 13, 16)] == [1, 2, 1, 1, 1, 2]
    assert [pol.pol_quad_split_late(nr) for nr in (5, 6, 10, 13, 16)] == [0, 2, 3, 3, 4]
    # the point-list kernels (entries possibly in storage order): a quarter / half of the rounds, as in round 4
    assert [pol.pol_quad_split_pts(nr) for nr in (1, 2, 3, 8, 13, 16)] == [1, 2, 2, 2, 3, 4]
    assert [pol.pol_quad_split_late_pts(nr) for nr in (9, 10, 13, 16)] == [0, 5, 7, 8]
    for nr in range(1, 17):                                   # a late split, where there is one, comes after the early one
        for f, g in ((pol.pol_quad_split, pol.pol_quad_split_late), (pol.pol_quad_split_pts, pol.pol_quad_split_late_pts)):
            assert g(nr) == 0 or f(nr) < g(nr) < nr


def test_first_call_is_plain_and_well_separated_data_goes_two_phase(pol):
    w = Walk(pol, no_hint=1)
    assert w.call() == "plain"
    w.seen(listed=10, ambig=0.001 * N)                       # < 0.2 % ambiguous: the unconditional two-phase form is safe
    assert w.call() == "two-phase"
    w.seen(listed=0.001 * N)                                 # it certifies well: stays
    assert w.call() == "two-phase"
    w.seen(listed=0.006 * N)                                 # > 0.5 % listed: plain again, and no retry for 16 calls
    forms = []
    for _ in range(17):
        forms.append(w.call())
        w.seen(ambig=0.0)                                    # (a plain call with nothing ambiguous asks for two-phase again ...)
    assert forms[:15] == ["plain"] * 15 and forms[-1] == "two-phase"   # ... but the pause holds for its 16 calls


def test_many_listed_points_send_the_next_calls_to_the_exact_kernels(pol):
    w = Walk(pol)
    assert w.call() == "plain"
    w.seen(listed=0.06 * N, ambig=0.5 * N)
    assert [w.call() for _ in range(8)] == ["exact"] * 8
    assert w.call() == "plain"                               # the bounds are gone with the exact calls: no hints yet


def test_hinted_form_late_split_first_then_early_and_back(pol):
    w = Walk(pol)
    assert w.call() == "plain"
    w.seen(ambig=0.3 * N)                                    # a cold run: lots of close runners-up, no two-phase
    forms = []
    for it in range(3):
        forms.append(w.call())
        w.seen(ambig=0.3 * N, early=0.4 * N / 16 * TILES)    # 40 % of the pairs finished early: it pays
    assert forms == ["hinted-late"] * 3                      # a run's first three hinted calls ask after half of the rounds
    assert w.call() == "hinted-early"
    w.seen(ambig=0.3 * N, early=0.10 * N / 16 * TILES)       # the early split finishes < 15 % early on a full screen ...
    assert [w.call(), ] == ["hinted-late"]                   # ... two calls back on the late split
    w.seen(ambig=0.3 * N, early=0.4 * N / 16 * TILES)
    assert w.call() == "hinted-late"
    w.seen(ambig=0.3 * N, early=0.4 * N / 16 * TILES)
    assert w.call() == "hinted-early"
    # with most steps settled by the carried bounds a poor early share does NOT go back to the late split
    w.seen(ambig=0.1 * N, early=0.10 * 0.1 * N / 16 * TILES, skipped=0.9 * N / 16)
    assert w.call() == "hinted-early"
    # SPKM_NO_LATE_SPLIT: always the early one
    w2 = Walk(pol, no_late=1)
    w2.call(); w2.seen(ambig=0.3 * N)
    assert w2.call() == "hinted-early"


def test_hints_that_do_not_pay_are_paused_with_doubling(pol):
    w = Walk(pol)
    w.call(); w.seen(ambig=0.3 * N)
    forms = []
    for _ in range(60):
        forms.append(w.call())
        w.seen(ambig=0.3 * N, early=0.0)                     # no step ever finishes early: the hints mislead
    at = [i for i, f in enumerate(forms) if f.startswith("hinted")]
    gaps = [b - a - 1 for a, b in zip(at, at[1:])]           # plain calls between two hinted ones
    # the three late-split calls of a run: pauses of 2, 4, 8 calls (the hinted call included); the first early-split call
    # that fails on a full screen is not a pause but the way back to the late split (next call); the failures so far still
    # count, so when that one fails too the pause is the longest one, 16 calls
    assert gaps[:6] == [1, 3, 7, 0, 15, 15], (gaps, forms[:40])
    assert forms[at[3]] == "hinted-early" and forms[at[4]] == "hinted-late"


def test_point_lists_enter_at_four_leave_below_two_and_a_half_and_lower_bars_for_short_lists(pol):
    w = Walk(pol)
    w.call(); w.seen(ambig=0.3 * N)
    w.call()                                                 # a call that ran the bounds test
    steps = N / 16

    def after(kept_share, skipped_share):
        w.seen(ambig=0.3 * N, early=0.5 * N, kept=kept_share * N, skipped=skipped_share * steps)
        r = pol.pol_pt_next(w.p)
        w.call()
        return r

    assert after(0.55, 0.0) == 0                             # < 60 % of the points passed
    assert after(0.85, 0.1) == 1                             # 15 % failing, scattered (90 % of the steps left): 6x -> lists
    assert after(0.55, 0.0) == 0                             # ... and off again when too many fail
    assert after(0.97, 0.925) == 0                           # steps left hold 7.5 % of the points vs 3 % failing: 2.5x < 4x
    assert after(0.97, 0.85) == 1                            # 15 % vs 3 %: 5x -> point lists
    assert after(0.97, 0.91) == 1                            # 3x: stays (left only below 2.5x)
    assert after(0.97, 0.94) == 0                            # 2x: back to steps
    # short lists (<= 2 % failing): the bars are 2.5x / 1.5x -- a listed point costs 1.4x a point of a listed step there
    assert after(0.97, 0.93) == 0                            # 3 % failing is not a short list: 2.3x < 4x
    assert after(0.99, 0.97) == 1                            # 1 % failing in 3 % of the steps: 3x -> point lists
    assert after(0.99, 0.982) == 1                           # 1.8x: stays (left only below 1.5x)
    assert after(0.99, 0.986) == 0                           # 1.4x: back to steps
    assert after(0.99, 0.98) == 0                            # 2x: not entered below 2.5x


def test_no_hinted_calls_on_overlapping_clusters(pol):
    """A plain call over all points that finds >= 90 % of them with a runner-up within 2.25x of the winner marks the data
    as crowded: no hinted call is issued (nothing would finish early) until a plain call over all points says otherwise."""
    w = Walk(pol)
    assert w.call() == "plain"
    w.seen(ambig=0.95 * N)                                   # overlapping clusters
    assert [w.call() for _ in range(1)] == ["plain"]         # (bounds exist now: a hinted call would be possible)
    w.seen(ambig=0.5 * N, skipped=0.0)                       # a call that ran the bounds test: its count says nothing
    assert w.call() == "plain"
    w2 = Walk(pol)
    assert w2.call() == "plain"
    w2.seen(ambig=0.4 * N)                                   # a third of the points between two centroids: hints pay
    assert w2.call().startswith("hinted")


def test_block_summaries_only_where_whole_blocks_settle(pol):
    """k_bounds_steps keeps its per-block summaries (settled blocks of 1024 points are not read) only when the previous
    bounds test passed >= 90 % of the points AND the failing points sit together (the steps left on the screen are at least
    an eighth full of them): with data in arbitrary order every block holds every cluster and one moving centroid keeps
    them all on the per-point path."""
    w = Walk(pol)
    w.call(); w.seen(ambig=0.3 * N)
    w.call()
    steps = N / 16

    def after(kept_share, skipped_share):
        w.seen(ambig=0.3 * N, early=0.5 * N, kept=kept_share * N, skipped=skipped_share * steps)
        r = (pol.pol_blocks_next(w.p), pol.pol_pt_next(w.p))
        w.call()
        return r

    assert after(0.5, 0.4)[0] == 0                           # half the points still fail
    assert after(0.97, 0.96)[0] == 1                         # settled, failing points sit together (4 % of the steps hold the 3 %): summaries
    assert after(0.97, 0.70) == (0, 1)                       # settled, but scattered (30 % of the steps left: point lists): no summaries
    assert after(0.85, 0.80)[0] == 0                         # not settled enough
    assert after(1.0, 1.0)[0] == 1                           # nothing fails at all
    assert after(0.999, 0.99)[0] == 0                        # 0.1 % failing, scattered over 1 % of the steps (16x): no
    assert after(0.999, 0.9985)[0] == 1                      # ... sitting together: yes
    pol.pol_reset(w.p)
    assert pol.pol_blocks_next(w.p) == 0


def test_incremental_sums_while_at_most_a_third_of_the_points_move(pol):
    w = Walk(pol)
    assert pol.pol_few_movers(w.p, N) == 1                   # no count yet (a run's second call): taken as few
    w.call(); w.seen()                                       # the first call cannot count movers (no previous assignment)
    assert pol.pol_few_movers(w.p, N) == 1
    w.call(); w.seen(movers=0.4 * N)
    assert pol.pol_few_movers(w.p, N) == 0
    w.call(); w.seen(movers=0.3 * N)
    assert pol.pol_few_movers(w.p, N) == 1
    pol.pol_reset(w.p)
    assert pol.pol_few_movers(w.p, N) == 1
    # pair events (one per mover, its record read once): the bar is half of the points, on the host and on the device
    w.call(); w.seen()
    w.call(); w.seen(movers=0.45 * N)
    assert pol.pol_few_movers(w.p, N) == 0 and pol.pol_few_movers_pair(w.p, N) == 1
    w.call(); w.seen(movers=0.55 * N)
    assert pol.pol_few_movers_pair(w.p, N) == 0
    for n in (0, 1, 2, 3, 100, 10**8):
        assert pol.pol_event_cap_pair(n) == n // 2


def test_a_call_without_a_mover_count_lets_the_device_choose_the_form(pol):
    """Table for the accumulation form of a lazy call (api_lloyd.hip `dual`, screen.hip k_pick_form): while no mover count has
    come back -- a run's second call, or any call whose predecessor's counters are still in flight -- both forms are
    queued and the device opens the events iff there are at most event_cap(n) of them (two per mover: a third of the
    points, the same bar few_movers applies to a known count); once a count is known the host decides alone."""
    w = Walk(pol)
    assert pol.pol_form_on_device(w.p) == 1                  # nothing known: the device decides
    w.call(); w.seen()                                       # first call: cannot count movers
    assert pol.pol_form_on_device(w.p) == 1                  # the run's second call is issued like this
    w.call()                                                 # ... its counters are not back yet:
    assert pol.pol_form_on_device(w.p) == 1                  # a third call issued now still lets the device decide
    w.seen(movers=0.9 * N)                                   # (from a random start nearly every point moves)
    assert pol.pol_form_on_device(w.p) == 0 and pol.pol_few_movers(w.p, N) == 0   # known and many: the full pass, host-side
    w.call(); w.seen(movers=0.1 * N)
    assert pol.pol_form_on_device(w.p) == 0 and pol.pol_few_movers(w.p, N) == 1   # known and few: the events, host-side
    pol.pol_reset(w.p)
    assert pol.pol_form_on_device(w.p) == 1                  # a new replicate starts over
    # the device's bar: events <= 2 * floor(n / 3)  <=>  movers <= n / 3 (every mover with a valid old cluster is 2 events)
    for n in (0, 1, 2, 3, 4, 100, 10**8, 125_000_000, 2**31):
        cap = pol.pol_event_cap(n)
        assert cap == 2 * (n // 3) and cap <= 2 * n
        movers_ok, movers_too_many = n // 3, n // 3 + 1
        assert 2 * movers_ok <= cap < 2 * movers_too_many


def test_few_movers_are_applied_without_a_sort(pol):
    """An incremental call applies its events one by one (k_events_direct: no plan, no placement, no slab kernel) when the
    previous call counted fewer than 2048 movers -- never on a guess: not before a count is back."""
    w = Walk(pol)
    assert pol.pol_events_direct(w.p) == 0                   # nothing known
    w.call(); w.seen()                                       # a first call counts no movers
    assert pol.pol_events_direct(w.p) == 0
    w.call(); w.seen(movers=5000)
    assert pol.pol_events_direct(w.p) == 0
    w.call(); w.seen(movers=2047)
    assert pol.pol_events_direct(w.p) == 1
    w.call(); w.seen(movers=0)
    assert pol.pol_events_direct(w.p) == 1
    w.call(); w.seen(movers=2048)
    assert pol.pol_events_direct(w.p) == 0
    pol.pol_reset(w.p)
    assert pol.pol_events_direct(w.p) == 0


def test_incremental_sums_are_refreshed_by_a_full_pass(pol):
    """Sums moved by events accumulate rounding relative to everything an entry ever held: once the movers counted since
    the last full pass add up to eight times the shard (or after 256 incremental calls) the next call runs the full pass again."""
    w = Walk(pol)
    w.call(sums="full"); w.seen()
    for _ in range(26):                                      # 0.3 N movers per incremental call: due after the 27th
        assert pol.pol_refresh_due(w.p, N) == 0
        w.call(sums="events"); w.seen(movers=0.3 * N)
    assert pol.pol_refresh_due(w.p, N) == 0                  # 7.8 N so far
    w.call(sums="events"); w.seen(movers=0.3 * N)
    assert pol.pol_refresh_due(w.p, N) == 1                  # 8.1 N > 8 N
    w.call(sums="full"); w.seen(movers=0.01 * N)
    assert pol.pol_refresh_due(w.p, N) == 0                  # the full pass starts the count over (its own movers do not count)
    for _ in range(255):
        w.call(sums="events"); w.seen(movers=1.0)
    assert pol.pol_refresh_due(w.p, N) == 0
    w.call(sums="events"); w.seen(movers=1.0)
    assert pol.pol_refresh_due(w.p, N) == 1                  # 256 incremental calls in a row


def test_movers_are_credited_to_the_call_that_was_observed(pol):
    """The events flag is latched with the launch it describes: a report that lags (no launched() for the newer call) is
    credited by what ITS call did, and a call that queued both forms and saw the device open the full pass starts the
    refresh count over."""
    w = Walk(pol)
    w.call(sums="full"); w.seen()
    w.call(sums="events"); w.seen(movers=0.5 * N)
    pol.pol_sums_by_full_pass(w.p)                           # a newer call (full pass) was issued while this report was pending:
    pol.pol_sums_by_events(w.p)                              # ... and another by events -- neither launched(): reports lag
    w.seen(movers=9.0 * N)                                   # the pending report is the EVENTS call's: its movers count
    assert pol.pol_refresh_due(w.p, N) == 1
    w2 = Walk(pol)
    w2.call(sums="full"); w2.seen()
    for _ in range(3):
        w2.call(sums="events"); w2.seen(movers=2.0 * N)
    w2.call(sums="events", both=True)                        # both forms queued; the device opened the full pass
    pol.pol_observe_full_opened(w2.p, 3.0 * N, N, TILES, NR)
    assert pol.pol_refresh_due(w2.p, N) == 0                 # 6 N + a fresh summation: the count starts over
    w2.call(sums="events"); w2.seen(movers=2.0 * N)
    assert pol.pol_refresh_due(w2.p, N) == 0
