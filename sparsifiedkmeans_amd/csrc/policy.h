// Which form the next fused call takes -- the host-side policy of the screen path, on its own so that it can be read,
// and tested (tests/test_policy.py drives it through tables on the CPU), apart from the launch code in api_lloyd.hip.
//
// Nothing here can change an output: every form computes the reference's assignment; the policy only chooses how much
// work the next call does.  It sees the counters of a screen call ONE CALL LATE (copied back asynchronously: no host
// sync on the hot path) and keeps its state per shard.
//
//   counters of a screen call (k_combine_screen, k_screen_quad, k_bounds_steps, k_call_tail):
//     listed   points the certificate could not settle (exact evaluation over all K)
//     ambig    points whose runner-up is within 2.25x of the winner
//     early    (16-point step, centroid tile) pairs the hinted form finished after its first rounds
//     skipped  16-point steps settled by the carried bounds (never screened)
//     kept     points that passed the carried-bounds test
//     movers   points that changed cluster (counted when the library holds the previous assignment)
#pragma once
#include <algorithm>
#include <cmath>

#ifdef __HIPCC__
#define SPKM_HD __host__ __device__
#else
#define SPKM_HD
#endif

// Two-phase forms of the 4-lanes-per-point screen (screen_quad.hip, TWO): the first A rounds for all centroids, the rest
// only for each tile's leader (or, hinted, for all again when a step's points do not clear their hints).  The split is a
// compile-time constant: with a run-time split every round sits behind its own branch and the finish's LDS reads are
// waited for one by one.  Two splits are compiled per round count, an EARLY one, A = quad_split(NR), and a LATE one,
// A = quad_split_late(NR) (0: none) for the first hinted calls of a run, when the hints are loose (the own centroid has
// just moved a long way) and the competition's partial sums clear them later.
// Round 5: the screen's step-major copy lists a point's entries by |x| DESCENDING (k_screen_reorder), so that the partial
// sums over the first 4 A entries -- lower bounds of the full sums whatever the order -- grow as fast as they can: a far
// centroid's term (x_j - c_j)^2 is x_j^2 + c_j^2 on average, and the 4 largest of 51 Gaussian |x| carry 38 % of sum x^2,
// the 12 largest 70 %.  Measured on the headline run (tools/exp_bounds_r5.py, share of the (16-point step, tile) pairs whose
// points all clear 1.5 x hint^2): iterations 2-4, 3 rounds in |x| order 0.53 / 0.57 / 0.59 against 7 rounds in storage
// order 0.57 / 0.57 / 0.59; iterations 5-9, ONE round in |x| order 0.61 / 0.64 / 0.79 / 0.86 / 0.90 against 3 rounds in
// storage order 0.62 / 0.65 / 0.81 / 0.87 / 0.91.  So for the ordered copy: early = an eighth of the rounds (s = 51: 1 of
// 13), late = a quarter (3 of 13).  pts = the point-list kernels, whose entries may come from the records in STORAGE
// order: they keep round 4's splits (a quarter / half of the rounds: 3 and 7 of 13).
SPKM_HD constexpr int quad_split_late(int nr, bool pts = false)
{
    return pts ? (nr >= 10 ? (nr + 1) / 2 : 0) : (nr >= 6 ? ((nr + 2) / 4 > 2 ? (nr + 2) / 4 : 2) : 0); // 0: none
}
SPKM_HD constexpr int quad_split(int nr, bool pts = false)
{
    return nr < 3 ? nr : (pts ? ((nr + 2) / 4 > 2 ? (nr + 2) / 4 : 2) : (nr / 8 > 1 ? nr / 8 : 1));
}

struct spkm_policy_counters {
    double listed = 0, ambig = 0, early = 0, skipped = 0, kept = 0, movers = 0;
    bool full_opened = false; // a call that queued both accumulation forms: the device opened the full pass (k_pick_form)
};

struct spkm_policy {
    // --- what the call whose counters are pending did ---
    int prune_pending_a = 0;        // rounds it evaluated for all centroids (0: all of them, the plain form)
    bool hint_pending = false;      // it used the hinted two-phase form ...
    bool hint_late_pending = false; // ... with the late split
    bool skip_pending = false;      // it ran the carried-bounds test
    bool mov_pending_valid = false; // it counted the movers
    bool ev_latched = false;        // it updated the sums by events (latched WITH the other pending facts: ev_pending below
                                    // describes the latest call issued, which is a later one when a report lags)
    bool dual_latched = false;      // ... and had queued the full pass as well (the device chose)
    // --- what the next call should do ---
    int exact_cooldown = 0;         // calls left on the all-exact kernels after a poorly certifying screen
    int prune_next_a = 0;           // unconditional two-phase form with this many rounds for all centroids (0: no)
    int prune_cooldown = 0;         // calls to wait before the unconditional form is tried again
    bool hint_late = true;          // the next hinted call uses the late split
    int hint_late_left = 3;         // hinted calls left on the late split
    int hint_cooldown = 0;          // calls to wait before the next hinted call
    int hint_fail_streak = 0;       // consecutive hinted calls that did not pay: the pause doubles (2, 4, 8, 16 calls)
    bool pt_next = false;           // the next bounds test lists POINTS, not 16-point steps
    bool blocks_next = false;       // the next bounds test keeps block summaries (k_bounds_steps: settled blocks are not read)
    bool crowded = false;           // the latest plain call over ALL points found >= 90 % of them with a runner-up within
                                    // 2.25x of the winner: clusters that overlap -- no partial sum clears a hint there
                                    // (hinted calls only cost: 32.3 against 31.0 ms at N = 1e8), so none is issued
    bool movers_known = false;      // last_movers is a count (not before a run's second screen call has been read back)
    unsigned long long last_movers = 0;
    // incremental sums are a running add / subtract: their rounding error is relative to everything a table entry has
    // ever held, so a full pass starts them afresh once the points that moved since the last one add up to the whole shard
    // eight times over (or after 256 incremental calls; a headline run moves its points once over in its first six
    // iterations -- a refresh there would cost a full pass, 8 ms at N = 1e8, for rounding noise of 1e-13)
    bool ev_pending = false;        // the latest call issued updated the sums by events
    int ev_calls = 0;               // incremental calls since the last full accumulation pass
    unsigned long long ev_cum_movers = 0; // movers counted over those calls

    // an incremental call whose predecessor counted fewer movers than this applies its events one by one (k_events_direct:
    // one launch, one f64 atomic per entry) instead of sorting them by cluster first (three launches)
    static constexpr unsigned long long direct_events_below = 2048;
    bool events_direct() const { return movers_known && last_movers < direct_events_below; }

    // a new start / new replicate (spkm_shard_reset_policy): nothing learned carries over
    void reset()
    {
        *this = spkm_policy();
    }

    // The counters of the pending call have arrived.  n points, tiles = centroid tiles of the screen, nr = rounds per
    // column (ceil(s / 4)).
    void observe(const spkm_policy_counters& c, double n, int tiles, int nr)
    {
        if (mov_pending_valid) { last_movers = (unsigned long long)c.movers; movers_known = true; }
        if (!hint_pending && prune_pending_a == 0 && !skip_pending) crowded = c.ambig >= 0.9 * n; // (a plain call that screened every point)
        if (ev_latched && dual_latched && c.full_opened) { ev_calls = 0; ev_cum_movers = 0; } // (the sums are fresh after all)
        else if (ev_latched && mov_pending_valid) ev_cum_movers += (unsigned long long)c.movers;
        // more than 5 % of the points on the exact list: the screen pays K-fold exact work for each; 8 calls all-exact
        if (c.listed > 0.05 * n) exact_cooldown = 8;
        // point-granular list for the next bounds test: worth its 16-B fetches only while few points are listed (in
        // cluster-contiguous order the failing points sit together and whole steps are as good).  Entered at 4x, left
        // below 2.5x: the two forms leave slightly different bounds behind, and a choice that flips every call pays for both.
        // At least 60 % of the points must have passed: a listed point's entries are gathered once per centroid tile
        // (512 B each at s = 51), which stops paying against whole steps somewhere below half.  (The bar was 90 %; on
        // overlapping clusters -- slack between the bounds small everywhere, eroded by a steady drift -- 10-40 % of the
        // points fail for many iterations in a row, scattered over all steps: whole steps meant the full screen, 27 ms
        // at N = 1e8 where the list takes 5-19; a run to convergence there went from 57 to 63 it/s.)
        // SHORT lists (at most 2 % of the points fail -- a settled run): the gathers meet in L2 and a listed point costs
        // 1.4x a point of a listed step, not 2x (N = 1e8, block order: 0.20 ms + 0.32 ns per listed point against
        // 0.20 ms + 0.23 ns per point of a listed step), so the bars are 2.5x / 1.5x there.  With 4x / 2.5x a settled run
        // left the point lists every sixth call -- the points that fail while their neighbours' bounds age grow from 0.3
        // to 0.9 % between two step calls (which refresh all 2 % that share a step with one) -- and paid 0.66 ms twice
        // where a point call takes 0.47.
        const bool short_list = (n - c.kept) <= 0.02 * n;
        const double bar = short_list ? (pt_next ? 1.5 : 2.5) : (pt_next ? 2.5 : 4.0);
        pt_next = skip_pending && c.kept >= 0.6 * n && (std::ceil(n / 16.0) - c.skipped) * 16.0 > bar * (n - c.kept);
        // block summaries pay when whole 1024-point blocks are settled: nearly every point passes and the points of a
        // cluster sit together (step lists: with data in arbitrary order every block holds every cluster, and one moving
        // centroid keeps them all on the per-point path -- the summaries would only cost their upkeep)
        // -- judged by how the failing points lie, not by the list form: the steps left on the screen are at least an eighth
        // full of them (scattered failures, a few per cent of the points, leave a quarter of all steps)
        blocks_next = skip_pending && c.kept >= 0.9 * n && (std::ceil(n / 16.0) - c.skipped) * 16.0 <= 8.0 * (n - c.kept);
        const int a_prune = quad_split(nr); // the early split (api_lloyd.hip takes the point-list kernels' value where it applies): a runner-up 2.25x away clears it
        const int t = std::max(1, tiles);
        if (hint_pending) {
            // hinted call: worth it only if a fair share of the (step, tile) pairs was finished early, and only while the
            // hints do not mislead (many listed points).  Steps skipped on the carried bounds never got as far as their hints.
            const double steps = std::max(0.0, n / 16.0 - c.skipped) * t;
            // early or late split: a run's first three hinted calls use the late one, then the early one; an early call
            // that finishes fewer than 15 % of its pairs early sends the next two back to the late split.  (The late
            // split's own early-finish share says little about when to leave it -- it moves from 0.37 to 0.44 over the
            // iterations in which the early split's goes from 0.1 to 0.4 -- so the way back is a fixed count.)
            const double share = steps > 0.0 ? c.early / steps : 1.0;
            const bool was_late = hint_late_pending;
            // (only while a good part of the data is on the screen: with most steps settled by the carried bounds the few
            //  that are left are the hard ones, and the late split just costs more rounds on them)
            const double all_pairs = n / 16.0 * t;
            if (!was_late && share < 0.15 && steps > 0.25 * all_pairs && hint_late_left == 0) hint_late_left = 2;
            hint_late = hint_late_left > 0;
            const bool fallback_to_late = !was_late && hint_late && quad_split_late(nr) > quad_split(nr);
            const bool poor = c.early < 0.05 * steps && steps > 0.01 * n / 16.0;
            if (c.listed > 0.005 * n || (poor && !fallback_to_late)) {
                hint_fail_streak = std::min(hint_fail_streak + 1, 4);
                hint_cooldown = 1 << hint_fail_streak; // early iterations mislead briefly, not for 16 calls
            } else if (poor) {
                // a poor early-split call whose way back to the late split is open: the next call tries that, without a
                // pause -- and without forgetting the failures so far (clusters that overlap never finish a step early
                // on either split: forgetting meant two wasted hinted calls in every seven)
            } else {
                hint_fail_streak = 0;
                // runner-up bounds of early-finished steps are partial sums, so `ambig` over-counts: still small means
                // the unconditional form (no hint loads, no second evaluation) is safe to try
                if (c.ambig <= 0.002 * n && prune_cooldown == 0 && a_prune < nr) prune_next_a = a_prune;
            }
        } else if (prune_pending_a == 0)
            prune_next_a = (c.ambig <= 0.002 * n && prune_cooldown == 0 && a_prune < nr) ? a_prune : 0;
        else if (c.listed > 0.005 * n) { prune_next_a = 0; prune_cooldown = 16; }
    }

    struct choice {
        bool exact;     // this call runs the all-exact kernels
        int prune_a;    // unconditional two-phase form: rounds for all centroids (0: not that form)
        bool want_hint; // the hinted form, if the carried bounds allow it
    };
    // One call is about to be issued: tick the pauses, say which form it takes.
    choice next(bool no_prune, bool no_hint, bool quad)
    {
        if (prune_cooldown > 0) prune_cooldown--;
        if (hint_cooldown > 0) hint_cooldown--;
        const bool cooling = exact_cooldown > 0;
        if (cooling) exact_cooldown--;
        choice ch;
        ch.exact = cooling;
        ch.prune_a = no_prune ? 0 : prune_next_a;
        ch.want_hint = ch.prune_a == 0 && !no_prune && !no_hint && hint_cooldown == 0 && quad && !crowded;
        return ch;
    }
    // The late-split bookkeeping of a hinted call that is actually issued (run_screen); returns whether it is a late one.
    bool take_hinted_split(int nr, bool no_late_split)
    {
        const bool late = hint_late && quad_split_late(nr) > quad_split(nr) && !no_late_split;
        if (hint_late_left > 0) hint_late_left--;
        if (hint_late_left == 0) hint_late = false;
        return late;
    }
    // A screen call has been queued and its counters' read-back started: remember what it was.
    void launched(int rounds_all, int rounds, bool hinted, bool hinted_late, bool skipping, bool movers_counted,
                  bool by_events = false, bool both_forms = false)
    {
        prune_pending_a = rounds_all < rounds ? rounds_all : 0;
        hint_pending = hinted;
        hint_late_pending = hinted && hinted_late;
        skip_pending = skipping;
        mov_pending_valid = movers_counted;
        ev_latched = by_events;
        dual_latched = by_events && both_forms;
    }
    // Incremental sums (events) instead of a full accumulation pass: while not too many points move -- at most a third
    // in the previous counted call (an event pair reads the point twice, through a gather: 0.2 ms per million movers at
    // s = 51 against 10.4 ms for a full pass over 1e8 points); no count yet (a run's second call): taken as few.
    // pair_events (api_lloyd.hip: K <= 128): one event per mover, its record read once -- 12.9 ms per 1e8 movers, sort included,
    // against 8.8 ms for the full pass with its own sort at N = 1e8: events pay up to two thirds of the points; taken up to half.
    bool few_movers(double n, bool pair_events = false) const
    {
        return !movers_known || (double)last_movers * (pair_events ? 2.0 : 3.0) <= n;
    }
    // ... and a call issued without a count (a run's second: the counters come back one call late; from a random start
    // nearly every point moves there) does not guess: it queues BOTH forms and the device opens one of them once the
    // events are counted (screen.hip, k_pick_form) -- the events while there are at most event_cap(n) of them, two per
    // mover, i.e. the same third of the points as above; the full sums-only pass otherwise.
    bool form_on_device() const { return !movers_known; }
    // A fused call has chosen how it gets its sums: by events (or both forms queued: counted as events) / by a full pass.
    void sums_by_events() { ev_pending = true; ev_calls++; }
    void sums_by_full_pass() { ev_pending = false; ev_calls = 0; ev_cum_movers = 0; }
    // the sums are due for a fresh summation (see ev_calls above)
    bool refresh_due(double n) const { return ev_calls >= 256 || (double)ev_cum_movers > 8.0 * n; }
    static unsigned long long event_cap(unsigned long long n, bool pair_events = false) { return pair_events ? n / 2ull : 2ull * (n / 3ull); }
};
