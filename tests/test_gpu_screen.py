"""-m gpu: the certified f32 screen + exact confirmation path (csrc/screen.hip) must give exactly
what the all-exact kernels and the oracle give: assignments and min distances bit for bit."""
import numpy as np
import pytest
import torch

from util import parts, random_csc, set_switch

pytestmark = pytest.mark.gpu

# SPKM_SWEEP=n multiplies the number of seeds of the random-shape sweeps (bug hunting; default 1)
import os as _os
_SW = max(1, int(_os.environ.get("SPKM_SWEEP", "1")))


def _run(ctx, X, Cm, gamma):
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    K = Cm.shape[1]
    eng = LloydEngine(Shard.from_scipy(ctx, X), K, gamma)
    centers = torch.tensor(np.ascontiguousarray(Cm.T), device=f"cuda:{ctx.device}")
    eng.assign_accumulate_step(centers)
    torch.cuda.synchronize()
    path, listed = eng.last_path_info()
    return eng, path, listed


def _check(eng, oracle, X, Cm, gamma):
    p, n = X.shape
    K = Cm.shape[1]
    ra, rd = oracle.assign(p, n, *parts(X), Cm, gamma)
    a, d = eng.assign.cpu().numpy(), eng.mind.cpu().numpy()
    assert np.array_equal(a, ra)
    assert np.array_equal(d, rd)
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), ra)
    red = eng.reduce.cpu().numpy()
    pk = p * K
    assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt)
    assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(float))
    assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-12 * max(np.abs(S).max(), 1e-300)
    assert abs(red[-1] - np.sum(rd * rd)) <= 1e-12 * np.sum(rd * rd)
    st = eng.stats.cpu().numpy()
    assert st[1] == rd.max() and int(st[2]) == int(np.argmax(rd))
    assert np.array_equal(eng.nk.cpu().numpy(), nk)


@pytest.mark.parametrize("p,n,K,s", [(1024, 20000, 100, 51), (512, 9000, 37, 26), (256, 5001, 17, 13),
                                     (1024, 3000, 128, 51), (64, 2000, 33, 64)])
def test_screen_path_equals_oracle(gpu_ctx, oracle, p, n, K, s):
    X = random_csc(p, n, min(s, p), seed=p + K)           # fixed stride: eligible for the screen
    Cm = np.random.default_rng(K).standard_normal((p, K)) * 0.2
    eng, path, listed = _run(gpu_ctx, X, Cm, s / p)
    assert path == 1, "screen path expected"
    _check(eng, oracle, X, Cm, s / p)
    assert listed < 0.05 * n                                # random data: almost everything certifies


@pytest.mark.parametrize("p,n,K,s", [(256, 4000, 44, 13),    # last tile 12 centroids: 2 pairs per lane
                                     (256, 4000, 48, 16),    # last tile 16: 2 pairs per lane, rounds of exactly 4
                                     (128, 3000, 64, 1),     # one entry per point, full tiles only
                                     (128, 3000, 35, 4),     # one full round
                                     (128, 3100, 41, 5),     # one full round + 1 entry; last tile 9 centroids
                                     (512, 2500, 70, 63),    # 16 rounds, 3 entries in the last
                                     (512, 2500, 20, 62),    # single tile of 20 (4 pairs per lane, padded)
                                     (256, 2000, 40, 65)])   # 65 entries: the 16-lanes-per-point kernel
def test_screen_kernel_variants_equal_oracle(gpu_ctx, oracle, p, n, K, s):
    """Every shape class of the 4-lanes-per-point screen (rounds 1..16, 1..4 entries in the last round, narrow
    last tiles of 1 / 2 centroid pairs per lane) and the fallback for columns longer than 64 entries."""
    X = random_csc(p, n, s, seed=3 * p + K + s)
    Cm = np.random.default_rng(K + s).standard_normal((p, K)) * 0.3
    eng, path, listed = _run(gpu_ctx, X, Cm, s / p)
    assert path == 1, "screen path expected"
    _check(eng, oracle, X, Cm, s / p)


@pytest.mark.parametrize("p,n,K,s", [(1024, 6000, 100, 70), (512, 5000, 37, 66), (256, 3000, 64, 80)])
def test_sixteen_lane_screen_kernel_equals_oracle(gpu_ctx, oracle, p, n, K, s):
    """The first-generation screen kernel (16 lanes per point): what columns longer than 64 entries use."""
    X = random_csc(p, n, s, seed=p + K + 1)
    Cm = np.random.default_rng(K + 1).standard_normal((p, K)) * 0.2
    eng, path, listed = _run(gpu_ctx, X, Cm, s / p)
    assert path == 1
    _check(eng, oracle, X, Cm, s / p)


@pytest.mark.parametrize("seed", range(72 * _SW))
def test_random_shapes_equal_oracle(gpu_ctx, oracle, seed, monkeypatch):
    """Seeded sweep over (p, n, K, s): whatever path the library picks, the outputs are the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.choice([64, 100, 128, 200, 256, 500, 512, 784, 1000, 1024]))
    s = int(rng.integers(1, min(64, p) + 1))
    K = int(rng.integers(2, 141))
    n = int(rng.integers(500, 4001))
    X = random_csc(p, n, s, seed=seed)
    scale = float(rng.choice([1e-3, 0.3, 1.0, 50.0]))
    Cm = rng.standard_normal((p, K)) * scale
    if seed % 4 == 0:
        Cm[:, K // 2] = Cm[:, 0]                                # an exact tie -> exact list
    if seed % 3 == 2:
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_SCREEN")       # the all-exact kernels on the same shapes
    eng, path, listed = _run(gpu_ctx, X, Cm, s / p)
    _check(eng, oracle, X, Cm, s / p)


@pytest.mark.parametrize("seed", range(24 * _SW))
def test_random_wide_shapes_equal_oracle(gpu_ctx, oracle, seed):
    """Shapes around the eligibility limits of the screen: rows beyond what an LDS tile holds, columns longer
    than 64 entries, many hundreds of centroids.  Whatever path is taken, the outputs are the oracle's."""
    rng = np.random.default_rng(11000 + seed)
    p = int(rng.choice([1024, 1100, 1136, 1137, 1278, 1279, 1500, 2048, 3000]))
    s = int(rng.integers(1, 101))
    K = int(rng.choice([2, 17, 33, 100, 129, 300, 600, 1025]))
    n = int(rng.integers(200, 1500))
    X = random_csc(p, n, s, seed=seed + 3)
    Cm = rng.standard_normal((p, K)) * 0.3
    eng, path, listed = _run(gpu_ctx, X, Cm, s / p)
    _check(eng, oracle, X, Cm, s / p)


def test_more_tiles_than_workgroups_per_xcd_takes_the_exact_path(gpu_ctx, oracle):
    """K = 1100 needs 35 screen tiles; an XCD has 32 workgroups, so the call must fall back to the exact tiles
    (and still be right) instead of failing."""
    p, n, K, s = 64, 1500, 1100, 8
    X = random_csc(p, n, s, seed=12)
    Cm = np.random.default_rng(12).standard_normal((p, K)) * 0.3
    eng, path, listed = _run(gpu_ctx, X, Cm, s / p)
    assert path == 0
    _check(eng, oracle, X, Cm, s / p)


def test_screen_with_32bit_row_ids_on_device(gpu_ctx, oracle):
    """Device-resident shard handed over with int32 row ids (spkm_shard_create_dev, ir_bits = 32)."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    p, n, K, s = 512, 5000, 50, 26
    X = random_csc(p, n, s, seed=91)
    Cm = np.random.default_rng(4).standard_normal((p, K)) * 0.3
    dev = f"cuda:{gpu_ctx.device}"
    pad = 48
    jc = torch.tensor(X.indptr.astype(np.int64), device=dev)
    ir = torch.zeros(X.nnz + pad, dtype=torch.int32, device=dev)
    xv = torch.zeros(X.nnz + pad, dtype=torch.float64, device=dev)
    ir[:X.nnz] = torch.tensor(X.indices.astype(np.int32), device=dev)
    xv[:X.nnz] = torch.tensor(X.data, device=dev)
    eng = LloydEngine(Shard.from_device(gpu_ctx, p, jc, ir, xv, nnz=X.nnz), K, s / p)
    eng.assign_accumulate_step(torch.tensor(np.ascontiguousarray(Cm.T), device=dev))
    torch.cuda.synchronize()
    assert eng.last_path_info()[0] == 1
    _check(eng, oracle, X, Cm, s / p)


def test_screen_sends_ties_to_the_exact_list(gpu_ctx, oracle):
    """Duplicate centroids (exact ties), near-duplicates (sqrt-collapse band) and empty-ish points
    cannot be certified by any f32 screen: they must take the exact route and still match."""
    p, n, K = 512, 6000, 40
    X = random_csc(p, n, 26, seed=77)
    rng = np.random.default_rng(5)
    Cm = rng.standard_normal((p, K)) * 0.3
    Cm[:, 21] = Cm[:, 4]                                    # same value, different screen tiles or same
    Cm[:, 39] = Cm[:, 4]
    Cm[:, 9] = np.nextafter(Cm[:, 8], np.inf)               # one ulp apart
    eng, path, listed = _run(gpu_ctx, X, Cm, 26 / 512)
    assert path == 1
    _check(eng, oracle, X, Cm, 26 / 512)
    a = eng.assign.cpu().numpy()
    assert not np.any(np.isin(a, [21, 39]))                 # first index wins the exact ties
    assert listed >= np.count_nonzero(np.isin(a, [4, 8, 9]))


def test_screen_with_extreme_magnitudes(gpu_ctx, oracle):
    """Values far outside f32 range make the estimates overflow: everything goes to the list, the
    answer stays exact."""
    p, n, K = 128, 1500, 20
    X = random_csc(p, n, 9, seed=3)
    X.data *= 1e30
    Cm = np.random.default_rng(2).standard_normal((p, K)) * 1e30
    eng, path, listed = _run(gpu_ctx, X, Cm, 9 / 128)
    assert path == 1 and listed == n
    _check(eng, oracle, X, Cm, 9 / 128)


def test_exact_path_is_used_when_not_eligible(gpu_ctx, oracle):
    X = random_csc(512, 4000, 26, seed=9, ragged=True)       # ragged: not fixed-stride
    Cm = np.random.default_rng(1).standard_normal((512, 30))
    eng, path, listed = _run(gpu_ctx, X, Cm, 0.05)
    assert path == 0
    _check(eng, oracle, X, Cm, 0.05)
    X2 = random_csc(512, 4000, 26, seed=9)                   # K = 1: nothing to screen
    eng, path, _ = _run(gpu_ctx, X2, Cm[:, :1], 0.05)
    assert path == 0
    _check(eng, oracle, X2, Cm[:, :1], 0.05)


@pytest.mark.parametrize("K", [2, 3, 5, 8, 9, 10, 16])
def test_small_k_takes_the_screen_and_equals_oracle(gpu_ctx, oracle, K):
    """K <= 16: a single (narrow) screen tile + exact confirmation."""
    X = random_csc(512, 4000, 26, seed=40 + K)
    Cm = np.random.default_rng(K).standard_normal((512, K)) * 0.3
    eng, path, listed = _run(gpu_ctx, X, Cm, 0.05)
    assert path == 1
    _check(eng, oracle, X, Cm, 0.05)


def test_full_lloyd_with_screen_matches_oracle_loop(gpu_ctx, oracle):
    from sparsifiedkmeans_amd import synth

    data = synth.sparsified_gmm_host(p=256, n=6000, K=24, gamma=0.06, seed=11, fwht=oracle.fwht)
    Y, p2, gamma, K = data["Y"], data["p2"], data["gamma"], 24
    rng = np.random.default_rng(4)
    C0 = oracle.mix(data["X"][:, rng.choice(6000, K, replace=False)], data["d"], p2)
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    eng = LloydEngine(Shard.from_scipy(gpu_ctx, Y), K, gamma)
    centers = torch.tensor(np.ascontiguousarray(C0.T), device="cuda:0")
    Cref = C0.copy()
    for it in range(5):                                      # teacher-forced
        centers.copy_(torch.tensor(np.ascontiguousarray(Cref.T)))
        out = eng.iterate(centers).cpu().numpy()
        assert eng.last_path_info()[0] == 1
        ref = oracle.lloyd(p2, Y.shape[1], *parts(Y), Cref, gamma, maxiter=1, tol=0.0)
        assert np.array_equal(eng.assign.cpu().numpy(), ref["assign"])
        assert np.array_equal(eng.mind.cpu().numpy(), ref["mind"])
        got = centers.cpu().numpy().T
        assert np.abs(got - ref["centers"]).max() <= 1e-6 * np.abs(ref["centers"]).max()
        assert abs(np.sqrt(out[1]) - ref["obj"][0]) <= 1e-12 * ref["obj"][0]
        Cref = ref["centers"]


def test_poorly_certifying_screen_backs_off_to_exact_kernels(gpu_ctx, oracle):
    """When more than 5 % of the points need the exact list, later iterations use the all-exact kernels
    (decided one call late, from an asynchronous copy of the count); results stay exact throughout."""
    import time

    p, n, K = 128, 3000, 20
    X = random_csc(p, n, 9, seed=3)
    X.data *= 1e30                                            # nothing certifies (f32 overflow)
    Cm = np.random.default_rng(2).standard_normal((p, K)) * 1e30
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, 9 / 128)
    centers = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda:0")
    paths = []
    for it in range(4):
        eng.assign_accumulate_step(centers)
        torch.cuda.synchronize()
        time.sleep(0.01)
        paths.append(eng.last_path_info()[0])
        _check(eng, oracle, X, Cm, 9 / 128)
    assert paths[0] == 1 and 0 in paths[1:]


def test_two_phase_screen_switches_itself_on_and_off(gpu_ctx, oracle):
    """Separated clusters with one centre each: after a plain screen has seen that no point has a runner-up within
    2.25x of the winner, the next call evaluates only a quarter of the rounds for all centroids and finishes the tile
    leaders (spkm_last_screen_rounds) -- with the oracle's outputs.  Ambiguous data (random centres) keeps or
    brings back the plain screen."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    data = synth.sparsified_gmm_host(p=256, n=6000, K=40, gamma=0.2, seed=5, fwht=oracle.fwht)
    Y, p2, gam = data["Y"], data["p2"], data["gamma"]
    true_c = np.zeros((p2, 40))
    for k in range(40):                                 # the ML-corrected centre of each planted cluster
        Yk = Y[:, data["labels"] == k]
        S = np.asarray(Yk.sum(axis=1)).ravel()
        Cnt = np.asarray((Yk != 0).sum(axis=1)).ravel()
        true_c[:, k] = gam * S / (Cnt + 1e-16)          # kmeans_sparsified.m:448
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, Y), 40, gam)
    good = torch.tensor(np.ascontiguousarray(true_c.T), device="cuda")
    seen = []
    for it in range(4):
        eng.assign_accumulate_step(good)
        torch.cuda.synchronize()
        seen.append(eng.last_screen_rounds())
        ra, rd = oracle.assign(p2, 6000, *parts(Y), true_c, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)
    nr = seen[0][1]
    assert seen[0] == (nr, nr)                      # first call: plain screen
    assert any(a < r for a, r in seen[1:])          # then the two-phase screen
    bad = torch.tensor(np.random.default_rng(0).standard_normal((40, p2)) * 0.01, device="cuda")   # every centre ~ equally far
    for it in range(4):
        eng.assign_accumulate_step(bad)
        torch.cuda.synchronize()
        last = eng.last_screen_rounds()
        ra, rd = oracle.assign(p2, 6000, *parts(Y), bad.cpu().numpy().T, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)
    assert last[0] == last[1] or eng.last_path_info()[0] == 0    # back on the plain screen (or the exact kernels)


def test_hinted_two_phase_screen_uses_previous_distances(gpu_ctx, oracle, monkeypatch):
    """Mid-run Lloyd state: some clusters are split between two nearby centres (runner-up within 2.25x: the
    unconditional two-phase form stays off) and some have no centre of their own.  From the second call on the
    library's per-point distance estimates let 16-point steps finish early (form 2, counter of early-finished steps
    > 0); outputs equal the oracle's bit for bit on every call, also after the caller has scribbled over its output
    buffers (the hints live in the library's own state)."""
    import time
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_BOUNDS")       # same centres every call: the carried bounds would skip it all
    K, n = 40, 8000
    data = synth.sparsified_gmm_host(p=256, n=n, K=K, gamma=0.2, seed=11, fwht=oracle.fwht)
    Y, p2, gam = data["Y"], data["p2"], data["gamma"]
    cen = np.zeros((p2, K))
    for k in range(K):
        Yk = Y[:, data["labels"] == k]
        S = np.asarray(Yk.sum(axis=1)).ravel()
        Cnt = np.asarray((Yk != 0).sum(axis=1)).ravel()
        cen[:, k] = gam * S / (Cnt + 1e-16)
    rng = np.random.default_rng(3)
    spread = np.abs(cen).mean()
    for k in range(0, 10):                                  # clusters 30..39 lose their centre; 0..9 get two
        delta = 0.3 * spread * rng.standard_normal(p2)
        cen[:, 30 + k] = cen[:, k] + delta
        cen[:, k] = cen[:, k] - delta
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, Y), K, gam)
    centers = torch.tensor(np.ascontiguousarray(cen.T), device="cuda")
    ra, rd = oracle.assign(p2, n, *parts(Y), cen, gam)
    modes = []
    for it in range(6):
        if it == 4:
            eng.mind.mul_(0.01)                             # a caller that reuses its buffers: no effect on the hints
        eng.assign_accumulate_step(centers)
        torch.cuda.synchronize()
        time.sleep(0.01)
        modes.append(eng.last_screen_mode())
        assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)
    assert modes[0][0] == 0                                 # nothing to go by on the first call
    assert any(m[0] == 2 for m in modes[1:4]), modes
    assert all(m[3] > 0 for m in modes if m[0] == 2), modes  # some steps did finish early
    _check(eng, oracle, Y, cen, gam)


def test_carried_bounds_skip_steps_and_stay_exact(gpu_ctx, oracle):
    """A converging Lloyd run: from the second call on, the bounds carried inside the library (previous assignment,
    upper / lower distance bounds, centroid drift) let the screen skip whole 16-point steps; every call's outputs
    still equal the oracle's bit for bit -- also when the caller scribbles over its output buffers in between (the
    bounds live in the library's own copies), after a jump of the centres (nothing may be skipped wrongly), and
    with SPKM_NO_BOUNDS-like behaviour after reset_policy (bounds forgotten)."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    K, n = 48, 9600       # (the jump of call 3 leaves two clusters of 200 points to the exact list: 4 %, under the 5 % at which
                          #  the library would answer with eight calls on the all-exact kernels -- no bounds carried there)
    data = synth.sparsified_gmm_host(p=256, n=n, K=K, gamma=0.2, seed=13, fwht=oracle.fwht)
    Y, p2, gam = data["Y"], data["p2"], data["gamma"]
    cen = np.zeros((p2, K))
    for k in range(K):
        Yk = Y[:, data["labels"] == k]
        cen[:, k] = gam * np.asarray(Yk.sum(axis=1)).ravel() / (np.asarray((Yk != 0).sum(axis=1)).ravel() + 1e-16)
    rng = np.random.default_rng(5)
    sh = Shard.from_scipy(gpu_ctx, Y)
    eng = LloydEngine(sh, K, gam)
    skipped = []

    def call(cm):
        centers = torch.tensor(np.ascontiguousarray(cm.T), device="cuda")
        eng.assign_accumulate_step(centers)
        torch.cuda.synchronize()
        skipped.append(eng.last_screen_mode()[4])
        _check(eng, oracle, Y, cm, gam)

    call(cen)                                               # 0: nothing carried yet
    call(cen * (1 + 1e-6))                                  # 1: tiny drift
    eng.assign.fill_(7)                                     # the caller's buffers are not what the bounds live in
    eng.mind.fill_(0.0)
    call(cen * (1 + 2e-6))                                  # 2
    moved = cen.copy()
    moved[:, 3] = cen[:, 17]                                # 3: centre 3 jumps onto centre 17: ties and reassignments
    call(moved)
    call(moved)                                             # 4: no drift at all
    jitter = moved + 0.05 * np.abs(cen).mean() * rng.standard_normal(cen.shape)
    call(jitter)                                            # 5: every centre moves a little
    sh.reset_policy()
    call(jitter)                                            # 6: bounds forgotten
    call(jitter)                                            # 7: carried again
    eng.gamma = gam * 1.25                                  # 8: another scaling of the centres: the bounds do not apply
    centers = torch.tensor(np.ascontiguousarray(jitter.T), device="cuda")
    eng.assign_accumulate_step(centers)
    torch.cuda.synchronize()
    skipped.append(eng.last_screen_mode()[4])
    _check(eng, oracle, Y, jitter, gam * 1.25)
    assert skipped[7] > 0 and skipped[8] == 0, skipped
    assert skipped[0] == 0 and skipped[6] == 0, skipped
    assert skipped[1] > 0.5 * (n // 16) and skipped[2] > 0.5 * (n // 16), skipped
    assert skipped[3] < skipped[2], skipped
    assert skipped[4] > 0, skipped


def test_kept_counting_sort_is_reused_only_when_it_is_still_this_call_s(gpu_ctx, oracle):
    """Converged calls reuse the previous call's counting sort (no assignment changed: the histogram / plan / scatter
    kernels return at once).  The sort lives in buffers of the CONTEXT: calls on another shard, the non-fused
    entry points and a change of assignments in between must all lead to correct sums, counts and cluster sizes."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    K = 40
    dA = synth.sparsified_gmm_host(p=256, n=6000, K=K, gamma=0.2, seed=21, fwht=oracle.fwht)
    dB = synth.sparsified_gmm_host(p=256, n=4500, K=K, gamma=0.2, seed=22, fwht=oracle.fwht)

    def centres(d):
        Y, gam = d["Y"], d["gamma"]
        c = np.zeros((d["p2"], K))
        for k in range(K):
            Yk = Y[:, d["labels"] == k]
            c[:, k] = gam * np.asarray(Yk.sum(axis=1)).ravel() / (np.asarray((Yk != 0).sum(axis=1)).ravel() + 1e-16)
        return c

    cA, cB = centres(dA), centres(dB)
    eA = LloydEngine(Shard.from_scipy(gpu_ctx, dA["Y"]), K, dA["gamma"])
    eB = LloydEngine(Shard.from_scipy(gpu_ctx, dB["Y"]), K, dB["gamma"])

    def call(e, d, c):
        e.assign_accumulate_step(torch.tensor(np.ascontiguousarray(c.T), device="cuda"))
        torch.cuda.synchronize()
        _check(e, oracle, d["Y"], c, d["gamma"])

    call(eA, dA, cA)
    call(eA, dA, cA)                                        # reuse possible from here on
    call(eA, dA, cA)
    call(eB, dB, cB)                                        # another shard writes the context's sort buffers
    call(eA, dA, cA)
    call(eB, dB, cB)
    call(eB, dB, cB)
    t = torch.tensor(np.ascontiguousarray(cA.T), device="cuda")
    eA.assign_step(t)                                       # the non-fused entry points use the same buffers
    eA.accumulate_step()
    torch.cuda.synchronize()
    call(eA, dA, cA)
    call(eA, dA, cA)
    moved = cA.copy()
    moved[:, 5] = cA[:, 6] * 1.001                          # assignments change: the sort must be redone
    call(eA, dA, moved)
    call(eA, dA, moved)
    call(eA, dA, cA)


def test_drift_is_measured_on_the_support_not_in_the_full_norm(gpu_ctx, oracle, monkeypatch):
    """The carried bounds move by || (c' - c) / gamma || restricted to a point's support (s of p rows); for ANY s rows
    that is at most the root of the s largest squared entries of the difference (k_center_drift), about half of the
    full 2-norm for a spread-out move and far below it for a move in few coordinates.  Centres that all drift a little
    in every coordinate: with the support-aware drift most steps are skipped, with the full norm
    (SPKM_NO_SUPPORT_DRIFT=1) clearly fewer -- and every call's outputs equal the oracle's either way."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    K, n = 48, 9000
    data = synth.sparsified_gmm_host(p=512, n=n, K=K, gamma=0.05, seed=31, fwht=oracle.fwht)
    Y, p2, gam = data["Y"], data["p2"], data["gamma"]
    cen = np.zeros((p2, K))
    for k in range(K):
        Yk = Y[:, data["labels"] == k]
        cen[:, k] = gam * np.asarray(Yk.sum(axis=1)).ravel() / (np.asarray((Yk != 0).sum(axis=1)).ravel() + 1e-16)
    shard = Shard.from_scipy(gpu_ctx, Y)
    skipped = {}
    for full_norm in (False, True):
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_SUPPORT_DRIFT", full_norm)
        shard.reset_policy()
        eng = LloydEngine(shard, K, gam)
        rng = np.random.default_rng(7)
        got = []
        cur = cen.copy()
        for it in range(12):
            eng.assign_accumulate_step(torch.tensor(np.ascontiguousarray(cur.T), device="cuda"))
            torch.cuda.synchronize()
            got.append(eng.last_screen_mode()[4])
            _check(eng, oracle, Y, cur, gam)
            # every centre moves in every coordinate, a dense difference, by a step that grows 1.6-fold per call: at some
            # call the accumulated drift eats the gap between the bounds -- earlier in the full norm than on the support
            cur = cen + 0.004 * 1.6 ** it * np.abs(cen).mean() * rng.standard_normal(cen.shape)
        skipped[full_norm] = got
    assert skipped[False][0] == 0 and skipped[True][0] == 0                       # (first call of a run: nothing carried)
    assert max(skipped[False]) > 0.5 * (n // 16) and max(skipped[True]) > 0.5 * (n // 16), skipped
    assert sum(skipped[False]) >= sum(skipped[True]) + (n // 16) // 2, skipped    # the support-aware drift skips clearly more


@pytest.mark.parametrize("n", [40000, 4099, 33])
def test_point_granular_bounds_list_on_data_in_arbitrary_order(gpu_ctx, oracle, monkeypatch, n):
    """Points of a cluster scattered over the shard: a 16-point step is rarely settled as a whole, so once most points
    pass the bounds test the library lists POINTS (spkm_last_screen_mode info[7] == 2) and the screen runs on them
    alone.  Outputs stay the oracle's, call after call; the A/B switch SPKM_NO_POINT_LIST=1 gives the same."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, K, gopt = 256, (24 if n > 1000 else 3), 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=31, noise=0.3)
    X = X[:, np.random.default_rng(0).permutation(n)]                # arbitrary order
    rng = np.random.default_rng(2)
    d = np.sign(rng.standard_normal(p))
    Xm = oracle.mix(X, d, p)
    Y = synth.sparsify_dense(Xm, synth.small_p_of(gopt, p), rng)
    gam = synth.small_p_of(gopt, p) / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    C0 = oracle.mix(X[:, rng.choice(n, K, replace=False)], d, p)     # K mixture points: a run that takes a while
    seen = {}
    for nolist in (False, True):
        if nolist:
            set_switch(monkeypatch, gpu_ctx, "SPKM_NO_POINT_LIST")
        shard.reset_policy()
        eng = LloydEngine(shard, K, gam)
        c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
        modes = []
        for it in range(16):
            used = c.cpu().numpy().T.copy()
            eng.iterate(c)
            torch.cuda.synchronize()                                 # (lets the library's asynchronous counters land)
            modes.append(eng.last_screen_mode()[7])
            ra, rd = oracle.assign(p, n, *parts(Y), used, gam)
            assert np.array_equal(eng.assign.cpu().numpy(), ra), (nolist, it)
            assert np.array_equal(eng.mind.cpu().numpy(), rd), (nolist, it)
        seen[nolist] = modes
    if n >= 4099:
        assert 2 in seen[False], seen                                # the point list was used ...
    assert 2 not in seen[True], seen                                 # ... and not when switched off


def test_unchanged_clusters_are_not_streamed_again_and_outputs_stay_exact(gpu_ctx, oracle, monkeypatch):
    """The exact pass skips clusters whose centroid is bitwise unchanged and that no point left or entered
    (spkm_exact_pass_points tells how many points it streamed).  A free-running loop converges: from then on nothing is
    streamed; then one centroid is nudged: its cluster (and whatever its points do) is streamed again, the rest is not.
    Every call: assignments = oracle, the distances on demand = oracle, sums / counts / obj2 as the all-processing run."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 256, 30000, 12, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=3, noise=0.2)
    rng = np.random.default_rng(5)
    d = np.sign(rng.standard_normal(p))
    Y = synth.sparsify_dense(oracle.mix(X, d, p), synth.small_p_of(gopt, p), rng)
    gam = synth.small_p_of(gopt, p) / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    C0 = oracle.mix(X[:, rng.choice(n, K, replace=False)], d, p)
    eng = LloydEngine(shard, K, gam)
    c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
    streamed = []
    nudged = False
    for it in range(30):
        used = c.clone()
        eng.assign_accumulate_step(c, want_mind=False)
        pts = eng.exact_pass_points()[1]
        streamed.append(pts)
        ra, rd = oracle.assign(p, n, *parts(Y), used.cpu().numpy().T, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra), it
        # sums / counts / nk / obj2 of the call against the oracle's accumulation of the same assignment
        S, Cnt, nk = oracle.accumulate(p, n, K, *parts(Y), ra)
        pk = p * K
        red = eng.reduce.cpu().numpy()
        assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt), it
        assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(np.float64)), it
        assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-11 * max(np.abs(S).max(), 1.0), it
        assert abs(red[-1] - np.sum(rd * rd)) <= 1e-11 * np.sum(rd * rd), it
        st = eng.stats.cpu().numpy()
        assert st[1] == rd.max() and int(st[2]) == int(np.argmax(rd)), it
        assert np.array_equal(eng.distances(used).cpu().numpy(), rd), it          # on demand, all clusters
        eng.allreduce_step()
        eng.finalize_step(c)
        if it >= 6 and streamed[-1] == 0 and not nudged:
            c[3] += 1e-3                                                          # one centroid moves a little
            nudged = True
            nk3 = int(nk[3])
        elif nudged and len(streamed) >= 2 and streamed[-2] == 0 and pts > 0:
            assert pts < n // 2 and pts >= nk3 - 50, (pts, nk3)                   # its cluster again, hardly more
    assert nudged and streamed[0] == n and 0 in streamed, streamed
    # the A/B switch streams everything every time and gives the same outputs
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_CLUSTER_SKIP")
    shard.reset_policy()
    eng2 = LloydEngine(shard, K, gam)
    c2 = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
    for it in range(8):
        eng2.iterate(c2, want_mind=False)
        assert eng2.exact_pass_points()[1] == n


def test_hinted_screen_early_and_late_split_give_the_oracles_answers(gpu_ctx, oracle, monkeypatch):
    """Columns of 51 entries (13 rounds), listed by |x| descending in the screen's copy: a run's first hinted calls ask after
    3 rounds (spkm_last_screen_rounds reports (3, 13)), later ones after 1 -- the point-list kernels, whose entries may be
    in storage order, after 7 and 3; SPKM_NO_LATE_SPLIT=1 keeps to the early split.  Which split runs changes the work,
    never an output: assignments and distances are the oracle's in every call of both runs."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 512, 24000, 40, 0.1                               # s = 51
    X, centres, labels = synth.gmm_dense(p, n, K, seed=17, noise=0.25)
    rng = np.random.default_rng(8)
    d = np.sign(rng.standard_normal(p))
    s = synth.small_p_of(gopt, p)
    assert s == 51
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    C0 = oracle.mix(X[:, rng.choice(n, K, replace=False)], d, p)     # K mixture points: duplicates, uncovered clusters
    seen = {}
    for nolate in (False, True):
        if nolate:
            set_switch(monkeypatch, gpu_ctx, "SPKM_NO_LATE_SPLIT")
        shard.reset_policy()
        eng = LloydEngine(shard, K, gam)
        c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
        rounds = []
        for it in range(12):
            used = c.cpu().numpy().T.copy()
            eng.iterate(c)
            torch.cuda.synchronize()                                 # (lets the library's asynchronous counters land)
            md = eng.last_screen_mode()
            rounds.append(eng.last_screen_rounds() + (md[7] == 2, md[0] == 2))   # (.., the list named points, hinted form)
            assert eng.last_path_info()[0] == 1
            ra, rd = oracle.assign(p, n, *parts(Y), used, gam)
            assert np.array_equal(eng.assign.cpu().numpy(), ra), (nolate, it)
            assert np.array_equal(eng.mind.cpu().numpy(), rd), (nolate, it)
        seen[nolate] = rounds
    late = {(3, 13, False, True), (7, 13, True, True)}               # hinted calls over 16-point steps / point lists
    assert late & set(seen[False]), seen                             # the late split ran ...
    assert not (late & set(seen[True])), seen                        # ... and not when switched off (the UNCONDITIONAL two-phase
    #                                                                  form asks after a quarter of the rounds either way)
    assert {(1, 13, False, True), (3, 13, True, True)} & set(seen[True]), seen   # (the early one did)
    assert all(r[1] == 13 for r in seen[False] + seen[True]), seen


@pytest.mark.parametrize("n,K,shuffled", [(40000, 24, False), (30011, 40, True), (4099, 5, True)])
def test_lazy_statistics_incremental_sums_stay_the_members_sums(gpu_ctx, oracle, monkeypatch, n, K, shuffled):
    """spkm_shard_set_lazy_stats: once few points move, a fused call without distances leaves the exact pass out -- the
    per-cluster sums and counts are moved by the points that changed cluster (events), upper bounds come from the
    screen -- and returns NaN for obj2 / the largest distance.  Every call: assignment and cluster sizes the oracle's
    bit for bit, counts exact, sums the members' sums to rounding; at the end distances + statistics on demand equal
    the oracle's.  The run takes incremental calls (NaN objectives) and is the run the A/B switch
    SPKM_NO_INCREMENTAL=1 gives."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, gopt = 256, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=41, noise=0.3)
    if shuffled:
        X = X[:, np.random.default_rng(0).permutation(n)]
    rng = np.random.default_rng(3)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    C0 = oracle.mix(X[:, rng.choice(n, K, replace=False)], d, p)
    runs = {}
    for incremental in (True, False):
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_INCREMENTAL", not incremental)
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_SUMS_ONLY", not incremental)   # (the other form that leaves the objective out)
        shard.reset_policy()
        shard.set_lazy_stats(True)
        eng = LloydEngine(shard, K, gam)
        c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
        assigns, lazy_calls = [], 0
        for it in range(14):
            used = c.cpu().numpy().T.copy()
            out = eng.iterate(c, want_mind=False).cpu().numpy()
            ra, rd = oracle.assign(p, n, *parts(Y), used, gam)
            a = eng.assign.cpu().numpy()
            assert np.array_equal(a, ra), f"iteration {it}"
            S, Cnt, nk = oracle.accumulate(p, n, K, *parts(Y), ra)
            red = eng.reduce.cpu().numpy()
            pk = p * K
            assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt), f"iteration {it}"
            assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(float))
            assert np.array_equal(eng.nk.cpu().numpy(), nk)
            assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-10 * np.abs(S).max(), f"iteration {it}"
            want = oracle.finalize_centers(S, Cnt, nk, gam, used)
            assert np.abs(c.cpu().numpy().T - want).max() <= 1e-9 * np.abs(want).max()
            if np.isnan(out[1]):
                lazy_calls += 1
                assert np.isnan(red[-1]) and np.all(np.isnan(eng.stats.cpu().numpy()))
            else:
                assert abs(out[1] - np.sum(rd * rd)) <= 1e-11 * np.sum(rd * rd)
            assigns.append(a)
        eng.distances(torch.tensor(np.ascontiguousarray(used.T), device="cuda"))
        assert np.array_equal(eng.mind.cpu().numpy(), rd)
        st = eng.stats.cpu().numpy()
        assert abs(st[0] - np.sum(rd * rd)) <= 1e-11 * np.sum(rd * rd) and st[1] == rd.max() and int(st[2]) == int(np.argmax(rd))
        runs[incremental] = (assigns, lazy_calls)
    shard.set_lazy_stats(False)
    assert runs[True][1] >= 3, runs[True][1]                      # the run did take incremental calls ...
    assert runs[False][1] == 0                                    # ... and the switch keeps the full pass in every call
    # (both runs were held to the oracle call by call on their OWN centres; against each other they may part at a razor's
    #  edge -- their sums differ in the last bits -- so: the same run, but for a handful of points at most)
    for a1, a0 in zip(runs[True][0], runs[False][0]):
        assert np.count_nonzero(a1 != a0) <= max(2, n // 10000)


def test_settled_blocks_are_not_visited_in_lazy_runs(gpu_ctx, oracle, monkeypatch):
    """Block summaries of the carried bounds (k_bounds_steps): in a lazy run a block of 1024 points that passed the test
    as a whole is tested as one point from then on and, passing, is not read -- its part of the caller's assignment
    buffer included (the lazy contract, spkm.h).  A run into the settled regime with drifting centres (teacher-forced:
    small random moves, so that the summaries' lag is exercised, then a larger one that sends blocks back to the per-point
    test): every call's assignment is the oracle's and the bounds stay bounds.  Then the contract, as a witness that blocks
    really are skipped: a value scribbled into a settled block of the caller's buffer stays there (with
    SPKM_NO_BLOCK_SKIP=1 the library repairs it, as it does without lazy statistics); a different buffer is filled completely."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 256, 40000, 12, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=15, noise=0.15)
    rng = np.random.default_rng(3)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    jc, ir, x = parts(Y)
    base = oracle.mix(centres, d, p) * gam
    sc = np.abs(base).max()
    shard = Shard.from_scipy(gpu_ctx, Y)
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_PRUNE")                    # (plain screen + bounds: the forms are not under test)
    for noskip in (False, True):
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_BLOCK_SKIP", noskip)
        shard.reset_policy()
        shard.set_lazy_stats(True)
        eng = LloydEngine(shard, K, gam)
        eps = [0.0, 1e-4, 2e-4, 3e-4, 4e-4, 5e-4, 5e-3, 5.1e-3, 5.2e-3, 5.3e-3]
        kept_share = []
        for it, e in enumerate(eps):
            Cm = base + e * sc * np.random.default_rng(50 + it // 1).standard_normal((p, K))
            c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
            eng.assign_accumulate_step(c, want_mind=False)
            torch.cuda.synchronize()
            m = eng.last_screen_mode()
            assert eng.last_path_info()[0] == 1
            ra, rd = oracle.assign(p, n, jc, ir, x, Cm, gam)
            assert np.array_equal(eng.assign.cpu().numpy(), ra), (noskip, it)
            D = oracle.dist_csc(p, n, jc, ir, x, Cm / gam)
            ub, lb, la = shard.debug_bounds()
            own = D[ra, np.arange(n)]
            Do = D.copy(); Do[ra, np.arange(n)] = np.inf
            assert np.array_equal(la, ra) and not np.any(ub.astype(np.float64) < own) and not np.any(lb > Do.min(axis=0)), (noskip, it)
            kept_share.append(m[4] * 16 / n)
        assert max(kept_share[2:6]) > 0.9, kept_share                    # the quiet calls settle nearly every step
        # the contract: scribble into a block that is certainly settled (all its points far inside their cluster)
        Cm = base + 5.3e-3 * sc * np.random.default_rng(50 + 9).standard_normal((p, K))
        c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
        for _ in range(3):                                               # same centres again: nothing moves, every point passes; the
            eng.assign_accumulate_step(c, want_mind=False)               # policy sees that one call late and switches the summaries on,
            torch.cuda.synchronize()                                     # the call after that writes them, the next one uses them
        good = eng.assign.clone()
        eng.assign[5000:5010] = K + 5                                    # (the host breaks the contract on purpose)
        eng.assign_accumulate_step(c, want_mind=False)
        torch.cuda.synchronize()
        repaired = bool(torch.equal(eng.assign, good))
        assert repaired == noskip, (noskip, eng.assign[4995:5015].cpu().numpy())
        # a different buffer is filled completely whatever the switch says
        eng.assign = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        eng.assign_accumulate_step(c, want_mind=False)
        torch.cuda.synchronize()
        assert torch.equal(eng.assign, good)
    shard.set_lazy_stats(False)


@pytest.mark.parametrize("second_call,expect", [("far", 3), ("near", 2)])
def test_second_lazy_call_lets_the_device_choose_between_events_and_the_full_pass(gpu_ctx, oracle, monkeypatch, second_call, expect):
    """A run's second lazy call is issued before any mover count has come back: both accumulation forms are queued and
    k_pick_form opens one from the number of events.  "far": the second call's centres send most points elsewhere -> the
    full sums-only pass (form 3); "near": a small drift -> the events (form 2).  Either way assignment, cluster sizes and
    counts are the oracle's bit for bit and the sums the members' sums; SPKM_NO_DUAL=1 (round 3's behaviour: always the
    events) gives the same outputs.  The calls after it are held to the oracle too (bounds / cache left in order)."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 256, 30011, 20, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=9, noise=0.3)
    X = X[:, np.random.default_rng(4).permutation(n)]
    rng = np.random.default_rng(6)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    jc, ir, x = parts(Y)
    base = oracle.mix(centres, d, p) * gam
    sc = np.abs(base).max()
    far = base[:, np.roll(np.arange(K), 3)]                            # every centroid takes another one's place
    seq = [base,
           far if second_call == "far" else base + 1e-3 * sc * rng.standard_normal((p, K)),
           (far if second_call == "far" else base) + 2e-3 * sc * rng.standard_normal((p, K)),
           (far if second_call == "far" else base) + 3e-3 * sc * rng.standard_normal((p, K))]
    shard = Shard.from_scipy(gpu_ctx, Y)
    outs = {}
    for nodual in (False, True):
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_DUAL", nodual)
        shard.reset_policy()
        shard.set_lazy_stats(True)
        eng = LloydEngine(shard, K, gam)
        forms, reds = [], []
        for it, Cm in enumerate(seq):
            c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
            eng.assign_accumulate_step(c, want_mind=False)             # (no host sync in between: the count cannot be back for call 2)
            if it >= 1:
                torch.cuda.synchronize()
            forms.append(eng.last_screen_mode()[6])
            ra, rd = oracle.assign(p, n, jc, ir, x, Cm, gam)
            assert np.array_equal(eng.assign.cpu().numpy(), ra), (nodual, it)
            S, Cnt, nk = oracle.accumulate(p, n, K, jc, ir, x, ra)
            red = eng.reduce.cpu().numpy()
            pk = p * K
            assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt), (nodual, it)
            assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(float)) and np.array_equal(eng.nk.cpu().numpy(), nk)
            assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-10 * np.abs(S).max(), (nodual, it)
            reds.append(red)
        outs[nodual] = (forms, reds)
        assert forms[0] == 3, forms                                    # a run's first call: the full pass, sums only
        assert forms[1] == (2 if nodual else expect), (nodual, forms)
    shard.set_lazy_stats(False)


@pytest.mark.parametrize("shuffled", [False, True])
def test_carried_bounds_stay_bounds_through_every_kind_of_lazy_call(gpu_ctx, oracle, monkeypatch, shuffled):
    """The bounds a shard carries between screen calls must BE bounds after every call, whatever form the call took:
    ub[i] >= the distance to the point's centroid, lb[i] <= the distance to every other centroid (under the centres the
    call was given).  A lazy call (no distances) writes upper bounds from the screen's certificate only for the points
    it screened; a point that passed the carried-bounds test has to take its centroid's drift in k_bounds_steps -- in an
    incremental call AND in a sums-only full pass (round 3 eroded only in the former: a stale bound survived a sums-only
    call and could keep a point in a cluster the reference's argmin had left).  Teacher-forced centres: small drifts
    (events), then a jump that moves more than a third of the points (the call after it is a sums-only pass WITH valid
    bounds), then small drifts again.  Assignments are the oracle's in every call."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 256, 30000, 12, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=5, noise=0.2)
    if shuffled:
        X = X[:, np.random.default_rng(1).permutation(n)]
    rng = np.random.default_rng(2)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    # (plain screen + carried bounds only: the unconditional two-phase form, chosen after the quiet calls, would list every
    #  point of the jump call and send the next eight calls to the all-exact kernels -- correct, but not the sequence under test)
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_PRUNE")
    shard.reset_policy()
    shard.set_lazy_stats(True)
    eng = LloydEngine(shard, K, gam)
    jc, ir, x = parts(Y)
    base = oracle.mix(centres, d, p) * gam                            # near the optimum: most points keep their cluster
    scale = np.abs(base).max()
    jump = base.copy()
    jump[:, : K // 2] = base[:, np.roll(np.arange(K // 2), 1)]        # half of the centroids trade places: their members move
    seq = [("drift", 0.0), ("drift", 2e-3), ("drift", 4e-3), ("jump", 0.0), ("jump", 2e-3), ("jump", 4e-3), ("jump", 5e-3),
           ("drift", 5e-3), ("drift", 6e-3)]
    forms = []
    for it, (what, eps) in enumerate(seq):
        Cm = (jump if what == "jump" else base) + eps * scale * np.random.default_rng(100 + it).standard_normal((p, K))
        c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
        eng.assign_accumulate_step(c, want_mind=False)
        torch.cuda.synchronize()                                       # (lets the library's asynchronous counters land)
        forms.append(eng.last_screen_mode()[6])
        assert eng.last_path_info()[0] == 1
        D = oracle.dist_csc(p, n, jc, ir, x, Cm / gam)                 # K x n, the reference's distances
        ra, rd = oracle.assign(p, n, jc, ir, x, Cm, gam)
        a = eng.assign.cpu().numpy()
        assert np.array_equal(a, ra), (it, what)
        ub, lb, la = shard.debug_bounds()
        assert np.array_equal(la, ra), (it, what)
        own = D[ra, np.arange(n)]
        Do = D.copy(); Do[ra, np.arange(n)] = np.inf
        other = Do.min(axis=0)
        bad_ub = np.flatnonzero(ub.astype(np.float64) < own)
        bad_lb = np.flatnonzero(lb > other)
        assert bad_ub.size == 0, (it, what, forms, bad_ub[:5], ub[bad_ub[:5]], own[bad_ub[:5]])
        assert bad_lb.size == 0, (it, what, forms, bad_lb[:5], lb[bad_lb[:5]], other[bad_lb[:5]])
    shard.set_lazy_stats(False)
    assert 2 in forms and 3 in forms[1:], forms                        # events, and a sums-only pass on valid bounds, both ran
    # (data in arbitrary order may have been regrouped inside the library on the way: the bounds above were handed out in the
    #  caller's order through its map)
    assert not shard.order_info()[0] or shuffled, shard.order_info()   # cluster-contiguous data is left as it lies


@pytest.mark.parametrize("regroup", [True, False, "0"])
def test_data_in_arbitrary_order_is_regrouped_inside_the_library_and_nothing_the_caller_sees_moves(gpu_ctx, oracle, monkeypatch, regroup):
    """A lazy run on shuffled data: after its first call the library regroups ITS order of the points by cluster
    (spkm_shard_order_info; SPKM_NO_REGROUP=1: not; =0: as if unset).  Every iteration: assignments the oracle's bit for bit in the CALLER's
    order, centres the members' means; at the end the distances on demand and their statistics are the oracle's, a column
    read back is the caller's column, and a second run after reset_policy (a new start) is exact as well."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    import scipy.sparse as sp

    p, n, K, gopt = 256, 24000, 20, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=77, noise=0.1)   # (separated clusters: overlapping ones are not regrouped)
    X = X[:, np.random.default_rng(4).permutation(n)]
    rng = np.random.default_rng(6)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    Yones = Y.copy(); Yones.data[:] = 1.0
    gam = s / p
    if not regroup:
        set_switch(monkeypatch, gpu_ctx, "SPKM_NO_REGROUP")
    elif regroup == "0":   # a switch set to 0 is off, like one that is not set
        monkeypatch.setenv("SPKM_NO_REGROUP", "0")
        gpu_ctx.reload_switches()
        regroup = True
    shard = Shard.from_scipy(gpu_ctx, Y)
    shard.set_lazy_stats(True)
    jc, ir, x = parts(Y)
    for run in range(2):
        shard.reset_policy()
        eng = LloydEngine(shard, K, gam)
        C0 = oracle.mix(X[:, np.random.default_rng(10 + run).choice(n, K, replace=False)], d, p)
        c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
        for it in range(10):
            used = c.cpu().numpy().T.copy()
            eng.iterate(c, want_mind=False)
            torch.cuda.synchronize()
            ra, rd = oracle.assign(p, n, jc, ir, x, used, gam)
            assert np.array_equal(eng.assign.cpu().numpy(), ra), (run, it)
            ind = sp.csr_matrix((np.ones(n), (ra, np.arange(n))), shape=(K, n))
            S, Cnt = (Y @ ind.T).toarray(), (Yones @ ind.T).toarray()
            refc = np.where(np.bincount(ra, minlength=K)[None, :] > 0, gam * S / (Cnt + 1e-16), used)
            assert np.abs(c.cpu().numpy().T - refc).max() <= 1e-9 * np.abs(refc).max(), (run, it)
        assert shard.order_info()[0] == regroup, shard.order_info()
        eng.distances(torch.tensor(np.ascontiguousarray(used.T), device="cuda"))
        assert np.array_equal(eng.mind.cpu().numpy(), rd)
        st = eng.stats.cpu().numpy()
        assert abs(st[0] - np.sum(rd * rd)) <= 1e-9 * np.sum(rd * rd) and st[1] == rd.max() and int(st[2]) == int(np.argmax(rd))
    shard.release_csc()
    for col in (0, 1, n // 3, n - 1):
        rows, vals = shard.column(col)
        assert np.array_equal(rows, Y.indices[Y.indptr[col]:Y.indptr[col + 1]]) and np.array_equal(vals, Y.data[Y.indptr[col]:Y.indptr[col + 1]])
    shard.set_lazy_stats(False)


def test_a_new_buffer_at_an_old_address_is_written_in_full_on_a_regrouped_shard(gpu_ctx, oracle):
    """on a regrouped shard the fused call trusts the caller's assignment buffer (stores only moves)
    while the lazy contract's claim stands.  A host that re-declares the contract (spkm_shard_set_lazy_stats) or builds a
    new engine may hand over a NEW buffer at the OLD address (torch's caching allocator): simulated here by scribbling
    over the buffer before re-declaring.  The next call has to leave the oracle's assignment in EVERY place, not only
    where a point moved."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 256, 24000, 20, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=78, noise=0.1)
    X = X[:, np.random.default_rng(5).permutation(n)]
    rng = np.random.default_rng(7)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    shard.reset_policy()
    shard.set_lazy_stats(True)
    jc, ir, x = parts(Y)
    eng = LloydEngine(shard, K, gam)
    C0 = oracle.mix(X[:, np.random.default_rng(12).choice(n, K, replace=False)], d, p)
    c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
    for it in range(6):
        used = c.cpu().numpy().T.copy()
        eng.iterate(c, want_mind=False)
        torch.cuda.synchronize()
    assert shard.order_info()[0], "the shard was expected to be regrouped by now"
    ra, _ = oracle.assign(p, n, jc, ir, x, used, gam)
    assert np.array_equal(eng.assign.cpu().numpy(), ra)
    for how in ("set_lazy_stats", "new engine"):
        eng.assign.fill_(-7)                        # "a new buffer at the old address": nothing of the old contents survives
        torch.cuda.synchronize()
        if how == "set_lazy_stats":
            shard.set_lazy_stats(True)
        else:
            ptr = eng.assign.data_ptr()
            keep = eng.assign
            eng = LloydEngine(shard, K, gam)         # (re-declares the contract for its own buffer)
            eng.assign = keep                        # ... which here IS the old address
            assert eng.assign.data_ptr() == ptr
            shard.set_lazy_stats(True)
        used = c.cpu().numpy().T.copy()
        eng.iterate(c, want_mind=False)
        torch.cuda.synchronize()
        ra, _ = oracle.assign(p, n, jc, ir, x, used, gam)
        got = eng.assign.cpu().numpy()
        assert np.array_equal(got, ra), (how, int((got != ra).sum()))
    shard.set_lazy_stats(False)


@pytest.mark.parametrize("direct", [True, False])
def test_few_movers_are_applied_one_by_one_and_give_the_members_sums(gpu_ctx, oracle, monkeypatch, direct):
    """An incremental call whose predecessor counted fewer than 2048 movers applies its events without sorting them
    (k_events_direct; spkm_last_screen_mode info[6] = 4); SPKM_NO_DIRECT_EVENTS=1 keeps the sorted form (2).  Either way
    assignments and per-row counts are the oracle's, the sums its sums to 1e-10 (north-star bar 1e-6) -- through drifts
    that move a handful of points, none at all, and a jump that moves thousands (the jump call itself still runs direct,
    on the small count it knew: it must apply them all the same)."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, gopt = 256, 30000, 12, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=7, noise=1.0)   # (clusters that touch: a small drift moves some dozen points)
    rng = np.random.default_rng(3)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_PRUNE")
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_DIRECT_EVENTS", not direct)
    shard.reset_policy()
    shard.set_lazy_stats(True)
    eng = LloydEngine(shard, K, gam)
    jc, ir, x = parts(Y)
    base = oracle.mix(centres, d, p) * gam
    scale = np.abs(base).max()
    jump = base.copy()
    jump[:, :4] = base[:, [1, 2, 3, 0]]                               # four centroids trade places: a third of the points move
    seq = [("drift", 0.0), ("drift", 6e-3), ("drift", 9e-3), ("drift", 9e-3), ("drift", 12e-3), ("jump", 12e-3), ("jump", 15e-3),
           ("drift", 15e-3), ("drift", 18e-3)]
    forms, movers = [], []
    prev = None
    for it, (what, eps) in enumerate(seq):
        Cm = (jump if what == "jump" else base) + eps * scale * np.random.default_rng(200 + int(eps * 1e4)).standard_normal((p, K))
        c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
        eng.assign_accumulate_step(c, want_mind=False)
        torch.cuda.synchronize()                                       # (lets the counters land: the next call knows the movers)
        forms.append(eng.last_screen_mode()[6])
        ra, rd = oracle.assign(p, n, jc, ir, x, Cm, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra), (it, what)
        movers.append(-1 if prev is None else int(np.count_nonzero(ra != prev)))
        prev = ra
        S, Cnt, nk = oracle.accumulate(p, n, K, jc, ir, x, ra)
        red = eng.reduce.cpu().numpy()
        pk = p * K
        assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt), (it, what, forms)
        assert np.array_equal(eng.nk.cpu().numpy(), nk), (it, what)
        assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-10 * np.abs(S).max(), (it, what, forms)
    shard.set_lazy_stats(False)
    assert forms[0] == 3, forms                                        # the run's first call: full pass, sums only
    assert movers[3] == 0 and 0 < movers[2] < 2048 and movers[5] > 2048, movers
    if direct:
        assert forms[3] == 4 and forms[4] == 4, (forms, movers)        # few movers counted by the call before: no sort
        assert forms[5] == 4, (forms, movers)                          # ... the jump call too: the count it knew was still small
        assert forms[6] != 4, (forms, movers)                          # the jump's count is back: sorted events (or the full pass)
    else:
        assert 4 not in forms, forms


@pytest.mark.parametrize("pair,K", [(True, 12), (False, 12), (True, 100), (True, 128), (None, 130)])
def test_pair_events_read_a_mover_once_and_give_the_members_sums(gpu_ctx, oracle, monkeypatch, pair, K):
    """Sorted events in their two formats: PAIR events (K <= 128: one event per mover, two-level counting sort by (new, old),
    one slab per run of a pair added to the new cluster's rows and subtracted from the old one's -- a mover's record is
    read once) and two events per mover over 2 K keys (SPKM_NO_PAIR_EVENTS=1, and always for K > 128).  Direct application
    is switched off so that every incremental call sorts.  Drifts that move some dozen points, none, a jump that moves a
    third: assignments and per-row counts the oracle's, sums to 1e-10, whatever the format."""
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, gopt = 256, 30000, 0.1
    X, centres, labels = synth.gmm_dense(p, n, K, seed=9, noise=1.0)
    rng = np.random.default_rng(4)
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    s = synth.small_p_of(gopt, p)
    Y = synth.sparsify_dense(oracle.mix(X, d, p), s, rng)
    gam = s / p
    shard = Shard.from_scipy(gpu_ctx, Y)
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_PRUNE")
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_DIRECT_EVENTS")
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_PAIR_EVENTS", pair is False)
    set_switch(monkeypatch, gpu_ctx, "SPKM_FORCE_PAIR_EVENTS")         # (by itself the library takes pairs when it expects >= 256 movers per pair)
    shard.reset_policy()
    shard.set_lazy_stats(True)
    eng = LloydEngine(shard, K, gam)
    jc, ir, x = parts(Y)
    base = oracle.mix(centres, d, p) * gam
    scale = np.abs(base).max()
    jump = base.copy()
    q = max(4, K // 3)
    jump[:, :q] = base[:, np.roll(np.arange(q), 1)]                   # a third of the centroids trade places
    seq = [("drift", 0.0), ("drift", 6e-3), ("drift", 9e-3), ("drift", 9e-3), ("drift", 12e-3), ("jump", 12e-3), ("drift", 12e-3),
           ("drift", 15e-3), ("jump", 15e-3)]
    forms, movers = [], []
    prev = None
    for it, (what, eps) in enumerate(seq):
        Cm = (jump if what == "jump" else base) + eps * scale * np.random.default_rng(300 + int(eps * 1e4)).standard_normal((p, K))
        c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
        eng.assign_accumulate_step(c, want_mind=False)
        torch.cuda.synchronize()
        forms.append((eng.last_screen_mode()[6],) + eng.last_events_form())
        ra, rd = oracle.assign(p, n, jc, ir, x, Cm, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra), (it, what)
        movers.append(-1 if prev is None else int(np.count_nonzero(ra != prev)))
        prev = ra
        S, Cnt, nk = oracle.accumulate(p, n, K, jc, ir, x, ra)
        red = eng.reduce.cpu().numpy()
        pk = p * K
        assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt), (it, what, forms, movers)
        assert np.array_equal(eng.nk.cpu().numpy(), nk), (it, what)
        assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-10 * np.abs(S).max(), (it, what, forms, movers)
    shard.set_lazy_stats(False)
    want_pair = 1 if (pair is True) else 0
    inc = [f for f in forms if f[0] == 2]
    assert len(inc) >= 4, (forms, movers)                              # the drifts and the first jump are incremental calls
    assert all(f[1] == 1 and f[2] == want_pair for f in inc), (forms, movers)   # sorted, in the expected format
    assert forms[5][0] == 2 and movers[5] > 2048, (forms, movers)      # the jump call itself sorted thousands of events


def test_a_fresh_contexts_second_lazy_call_is_already_incremental(oracle, monkeypatch):
    """A context's first fused call allocates most of its buffers AFTER queueing its counting sort; those first-time
    allocations must not make the library forget the sort (and with it the previous assignment and cluster sizes): the
    second call of a run is then an incremental one, on buffers the first call sized for it.  Outputs are the oracle's."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context

    ctx = torch_context(0)                                          # its own context: no buffer exists yet
    # (SPKM_NO_DUAL: the second call takes the events whatever moves -- with the form chosen on the device, random centres
    #  on random data would open the full pass, which is also what a library that HAD forgotten the sort would run)
    set_switch(monkeypatch, ctx, "SPKM_NO_DUAL")
    p, n, K, s = 256, 30000, 24, 26
    X = random_csc(p, n, s, seed=123)
    gam = s / p
    shard = Shard.from_scipy(ctx, X)
    shard.set_lazy_stats(True)
    eng = LloydEngine(shard, K, gam)
    rng = np.random.default_rng(9)
    c = torch.tensor(np.ascontiguousarray((rng.standard_normal((p, K)) * 0.3).T), device="cuda")
    nan_obj = []
    for it in range(3):
        used = c.cpu().numpy().T.copy()
        out = eng.iterate(c, want_mind=False).cpu().numpy()
        ra, _ = oracle.assign(p, n, *parts(X), used, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra), f"iteration {it}"
        S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), ra)
        red = eng.reduce.cpu().numpy()
        assert np.array_equal(red[p * K:2 * p * K].reshape(K, p).T, Cnt)
        assert np.abs(red[:p * K].reshape(K, p).T - S).max() <= 1e-10 * np.abs(S).max()
        nan_obj.append((bool(np.isnan(out[1])), eng.last_screen_mode()[6]))
    # first call: the full pass, without distances (lazy); second: events.  (The third call's path depends on how many
    # points the second saw move: random centres on random data move most.)
    assert nan_obj[:2] == [(True, 3), (True, 2)], nan_obj
    shard.set_lazy_stats(False)


@pytest.mark.parametrize("adopted", [False, True])
def test_releasing_the_csc_arrays_changes_nothing_but_the_footprint(gpu_ctx, oracle, adopted):
    """spkm_shard_release_csc: the record layout becomes the only copy of the exact entries.  Fused calls, distances on
    demand and the exact list keep giving the oracle's outputs; an entry point that needs CSC (spkm_assign_dev: all-exact
    kernels, K = 1 stream) re-materialises library-owned arrays from the records and is the oracle's too; a second
    release lets them go again.  For an adopted shard the caller's tensors are simply no longer referenced."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, s = 256, 20000, 37, 26
    X = random_csc(p, n, s, seed=77)
    gam = s / p
    rng = np.random.default_rng(5)
    if adopted:
        pad = 48
        ir = torch.zeros(n * s + pad, dtype=torch.int16, device="cuda")
        xv = torch.zeros(n * s + pad, dtype=torch.float64, device="cuda")
        ir[: n * s] = torch.tensor(X.indices.astype(np.int16), device="cuda")
        xv[: n * s] = torch.tensor(X.data, device="cuda")
        jc = torch.arange(0, (n + 1) * s, s, dtype=torch.int64, device="cuda")
        shard = Shard.from_device(gpu_ctx, p, jc, ir, xv, nnz=n * s)
        del ir, xv
    else:
        shard = Shard.from_scipy(gpu_ctx, X)
    eng = LloydEngine(shard, K, gam)
    Cm = rng.standard_normal((p, K)) * 0.3
    Cm[:, 5] = Cm[:, 9]                                            # an exact tie: the exact list has work to do
    c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
    eng.assign_accumulate_step(c)
    _check(eng, oracle, X, Cm, gam)
    def col_ok(i):
        r, v = shard.column(i)
        return np.array_equal(r, X.indices[X.indptr[i]:X.indptr[i + 1]]) and np.array_equal(v, X.data[X.indptr[i]:X.indptr[i + 1]])

    assert col_ok(0) and col_ok(n - 1) and col_ok(777)             # spkm_shard_get_column_host from the CSC arrays
    assert shard.release_csc()
    assert shard.release_csc()                                     # idempotent
    assert col_ok(0) and col_ok(n - 1) and col_ok(4242)            # ... and from the records
    torch.cuda.empty_cache()
    for it in range(3):                                            # fused calls on the records alone
        Cm = Cm + 0.01 * rng.standard_normal((p, K))
        Cm[:, 5] = Cm[:, 9]
        c = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
        eng.assign_accumulate_step(c, want_mind=(it != 1))
        if it == 1:
            eng.distances(c)                                       # distances on demand, streamed from the records
        _check(eng, oracle, X, Cm, gam)
    # an entry point that reads CSC: the arrays come back from the records
    eng.assign_step(c)
    ra, rd = oracle.assign(p, n, *parts(X), Cm, gam)
    assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)
    eng.accumulate_step()
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), ra)
    red = eng.reduce.cpu().numpy()
    assert np.array_equal(red[p * K:2 * p * K].reshape(K, p).T, Cnt)
    e1 = LloydEngine(shard, 1, gam)                                # K = 1: the streaming kernel of the k-means++ rounds
    c1 = torch.tensor(np.ascontiguousarray(Cm[:, :1].T), device="cuda")
    e1.assign_step(c1)
    _, d1 = oracle.assign(p, n, *parts(X), Cm[:, :1], gam)
    assert np.array_equal(e1.mind.cpu().numpy(), d1)
    assert shard.release_csc()                                     # ... and go again
    eng.assign_accumulate_step(c)
    _check(eng, oracle, X, Cm, gam)
