// Test harness (CPU, g++): a C face on sparsifiedkmeans_amd/csrc/policy.h so that tests/test_policy.py can walk the
// screen-form policy through tables of counters with ctypes.  Not part of the product.
#include "../../sparsifiedkmeans_amd/csrc/policy.h"

extern "C" {
void* pol_new() { return new spkm_policy(); }
void pol_free(void* p) { delete (spkm_policy*)p; }
void pol_reset(void* p) { ((spkm_policy*)p)->reset(); }
void pol_observe(void* p, double listed, double ambig, double early, double skipped, double kept, double movers, double n,
                 int tiles, int nr)
{
    spkm_policy_counters c;
    c.listed = listed; c.ambig = ambig; c.early = early; c.skipped = skipped; c.kept = kept; c.movers = movers;
    ((spkm_policy*)p)->observe(c, n, tiles, nr);
}
// out: exact, prune_a, want_hint
void pol_next(void* p, int no_prune, int no_hint, int quad, int* out)
{
    const spkm_policy::choice ch = ((spkm_policy*)p)->next(no_prune != 0, no_hint != 0, quad != 0);
    out[0] = ch.exact; out[1] = ch.prune_a; out[2] = ch.want_hint;
}
int pol_take_hinted_split(void* p, int nr, int no_late) { return ((spkm_policy*)p)->take_hinted_split(nr, no_late != 0); }
void pol_launched(void* p, int rounds_all, int rounds, int hinted, int late, int skipping, int movers_counted, int by_events,
                  int both_forms)
{
    ((spkm_policy*)p)->launched(rounds_all, rounds, hinted != 0, late != 0, skipping != 0, movers_counted != 0, by_events != 0,
                                both_forms != 0);
}
void pol_observe_full_opened(void* p, double movers, double n, int tiles, int nr)
{
    spkm_policy_counters c;
    c.movers = movers; c.full_opened = true;
    ((spkm_policy*)p)->observe(c, n, tiles, nr);
}
int pol_blocks_next(void* p) { return ((spkm_policy*)p)->blocks_next; }
int pol_pt_next(void* p) { return ((spkm_policy*)p)->pt_next; }
int pol_few_movers(void* p, double n) { return ((spkm_policy*)p)->few_movers(n); }
int pol_few_movers_pair(void* p, double n) { return ((spkm_policy*)p)->few_movers(n, true); }
void pol_sums_by_events(void* p) { ((spkm_policy*)p)->sums_by_events(); }
void pol_sums_by_full_pass(void* p) { ((spkm_policy*)p)->sums_by_full_pass(); }
int pol_refresh_due(void* p, double n) { return ((spkm_policy*)p)->refresh_due(n); }
int pol_form_on_device(void* p) { return ((spkm_policy*)p)->form_on_device(); }
int pol_events_direct(void* p) { return ((spkm_policy*)p)->events_direct(); }
unsigned long long pol_event_cap(unsigned long long n) { return spkm_policy::event_cap(n); }
unsigned long long pol_event_cap_pair(unsigned long long n) { return spkm_policy::event_cap(n, true); }
int pol_quad_split(int nr) { return quad_split(nr); }
int pol_quad_split_late(int nr) { return quad_split_late(nr); }
int pol_quad_split_pts(int nr) { return quad_split(nr, true); }
int pol_quad_split_late_pts(int nr) { return quad_split_late(nr, true); }
}
