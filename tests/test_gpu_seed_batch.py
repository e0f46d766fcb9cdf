"""Device-resident seed batches (svo_hip_seed_batch_*: what the drop-in DepthFilter keeps per keyframe) against the
stateless pass svo_hip_depth_filter_update_dev and the CPU oracle: same states bit for bit over several frames, the
events are exactly the seeds the reference's list walk acts on (converged -> callback + erase, NaN -> erase, on keyframes
every updated seed's px_cur), in list order, and erased seeds are never touched again (S/depth_filter.cpp:237-341)."""
import numpy as np
import pytest

from android_svo_amd import hip, seedsynth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return hip.Context(0)


def _case(ctx, n, sigma_scale, width=640, height=480):
    sc = seedsynth.make_seed_case(n_seeds=n, seed=21, width=width, height=height)
    kf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
    cf = hip.Pyramid(ctx, sc.cam.width, sc.cam.height, 5, 1)
    kf.upload(0, sc.ref_pyr)
    cf.upload(0, sc.cur_pyr)
    s2 = (sc.sigma2 * np.float32(sigma_scale)).astype(np.float32)
    return sc, kf, cf, s2


@pytest.mark.parametrize("n", [20000, 1500])               # 1500: the two-launch form of small passes (svo_hip_df_set_small_pass_limit)
def test_resident_batch_equals_the_stateless_pass_over_several_frames(ctx, n):
    sc, kf, cf, s2 = _case(ctx, n, 0.0045, 1280, 720)          # bench_c4's seed variance: about half converge in the first pass
    sb = hip.SeedBatch(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2)
    rs = hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2)
    alive = np.ones(n, bool)
    total_events = 0
    for frame in range(4):
        keyframe = frame == 2
        hip.depth_filter_update(ctx, kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, sb)
        ctx.sync()
        st = sb.status.download()
        ev, counts = rs.update(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w, report_updated=keyframe)
        gone = alive & ((st == hip.SEED_CONVERGED) | (st == hip.SEED_NAN))
        want = gone | (alive & (st >= hip.SEED_UPDATED)) if keyframe else gone
        idx = np.nonzero(want)[0]
        assert np.array_equal(ev["index"], idx)                       # ascending seed index = list order
        assert np.array_equal(ev["status"], st[idx])
        assert np.array_equal(ev["mu"], sb.mu.download()[idx]) and np.array_equal(ev["sigma2"], sb.sigma2.download()[idx])
        pc = sb.px_cur.download()[idx]
        assert np.array_equal(ev["px_cur"], pc)
        conv = ev["status"] == hip.SEED_CONVERGED
        assert np.array_equal(ev["xyz_world"][conv], sb.xyz.download()[idx][conv])
        # the pass's status histogram: erased seeds in slot 0, the others as the stateless pass reports them
        assert counts[0] == int((~alive).sum())
        for s_ in range(6):
            assert counts[s_ + 1] == int((alive & (st == s_)).sum()), (frame, s_)
        assert np.array_equal(rs.status()[alive], st[alive]) and (rs.status()[~alive] == hip.SEED_ERASED).all()
        alive &= ~gone
        total_events += len(ev)
        d = rs.download()
        assert np.array_equal(d["alive"].astype(bool), alive) and rs.n_alive() == int(alive.sum())
        for k, arr in (("a", sb.a), ("b", sb.b), ("mu", sb.mu), ("sigma2", sb.sigma2)):
            assert np.array_equal(d[k][alive], arr.download()[alive]), (frame, k)
    assert total_events > n // 40 and alive.sum() < n                 # the case does converge seeds
    rs.destroy()
    sb.free()


def test_erased_seeds_are_left_alone_and_first_pass_matches_the_oracle(ctx):
    from oracle import orc
    n = 3000
    sc, kf, cf, s2 = _case(ctx, n, 1.0)
    rs = hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2)
    erased = np.arange(5, n, 7)
    rs.erase(erased)
    rs.erase(erased[:10])                                             # erasing twice changes nothing
    assert rs.n_alive() == n - len(erased)
    ev, counts = rs.update(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
    a, b, mu, s2o = sc.a.copy(), sc.b.copy(), sc.mu.copy(), s2.copy()
    o = orc.update_seeds(sc.cam, sc.ref_pyr, sc.cur_pyr, sc.T_ref_w, sc.T_cur_w, sc.px, sc.f, sc.level, a, b, mu, sc.z_range, s2o)
    st = rs.status()
    keep = np.ones(n, bool)
    keep[erased] = False
    assert (st[erased] == hip.SEED_ERASED).all() and counts[0] == len(erased)
    assert np.array_equal(st[keep], o["status"][keep])
    d = rs.download()
    assert np.array_equal(d["mu"][erased], sc.mu[erased]) and np.array_equal(d["sigma2"][erased], s2[erased])   # untouched
    same = (d["mu"][keep] == mu[keep]).mean()
    assert same > 0.995                                                # the HIP pass vs the oracle: as the stateless pass
    assert not np.isin(ev["index"], erased).any()
    # an update without a collect in between is refused, so is a collect without an update
    rs.update_async(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
    with pytest.raises(hip.SvoHipError):
        rs.update_async(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
    rs.collect()
    with pytest.raises(hip.SvoHipError):
        rs.collect()
    rs.destroy()


def test_two_batches_enqueued_back_to_back_one_wait(ctx):
    n = 5000
    sc, kf, cf, s2 = _case(ctx, n, 0.0045, 1280, 720)
    half = n // 2
    whole = hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2)
    parts = [hip.ResidentSeeds(ctx, sc.px[s], sc.f[s], sc.level[s], sc.a[s], sc.b[s], sc.mu[s], sc.z_range[s], s2[s])
             for s in (slice(0, half), slice(half, n))]
    ev_w, _ = whole.update(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
    for p in parts:
        p.update_async(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
    evs = [p.collect()[0] for p in parts]
    idx = np.concatenate([evs[0]["index"], evs[1]["index"] + half])
    assert np.array_equal(idx, ev_w["index"]) and len(idx) > 500
    assert np.array_equal(np.concatenate([e["sigma2"] for e in evs]), ev_w["sigma2"])
    for o_ in [whole] + parts:
        o_.destroy()


@pytest.mark.parametrize("small_limit", [16384, 0])
@pytest.mark.parametrize("sizes", [(500, 1, 255, 256, 700), (300,) * 11, (3000, 4097)])
def test_grouped_pass_over_several_keyframes_equals_one_pass_per_keyframe(ctx, sizes, small_limit):
    """svo_hip_seed_batch_update_group_async: the seeds of every keyframe of a frame through ONE set of launches --
    states, events and status counts per batch bit-identical to one svo_hip_seed_batch_update_async per keyframe (sizes
    that end inside a block, a single seed, more batches than one launch set takes).  The passes per keyframe run as six
    launches each (the two-launch form of small passes switched off); the grouped ones in both forms."""
    mk = seedsynth.make_multi_keyframe_case(sizes, seed=31)
    K = len(sizes)
    kf = hip.Pyramid(ctx, mk.cam.width, mk.cam.height, 5, K)
    cf = hip.Pyramid(ctx, mk.cam.width, mk.cam.height, 5, 1)
    cf.upload(0, mk.cur_pyr)
    one, grp = [], []
    for k, sc in enumerate(mk.keyframes):
        kf.upload(k, sc.ref_pyr)
        s2 = (sc.sigma2 * np.float32(0.02)).astype(np.float32)
        one.append(hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2))
        grp.append(hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2))
    slots = list(range(K))[::-1] if K == 5 else list(range(K))        # (slot order need not be batch order)
    if K == 5:
        for k, sc in enumerate(mk.keyframes):
            kf.upload(slots[k], sc.ref_pyr)
    T_refs = np.stack([sc.T_ref_w for sc in mk.keyframes])
    n_events = 0
    for frame in range(3):
        keyframe = frame == 1
        ctx.set_small_pass_limit(0)
        for k, sc in enumerate(mk.keyframes):
            one[k].update_async(kf, slots[k], cf, 0, mk.cam, sc.T_ref_w, mk.T_cur_w, report_updated=keyframe)
        want = [r.collect() for r in one]
        ctx.set_small_pass_limit(small_limit)
        hip.ResidentSeeds.update_group_async(grp, kf, slots, cf, 0, mk.cam, T_refs, mk.T_cur_w, report_updated=keyframe)
        got = [r.collect() for r in grp]
        for k in range(K):
            assert np.array_equal(got[k][1], want[k][1]), (frame, k)
            assert got[k][0].tobytes() == want[k][0].tobytes(), (frame, k)
            assert np.array_equal(grp[k].status(), one[k].status())
            d, e = grp[k].download(), one[k].download()
            for key in d:
                assert d[key].tobytes() == e[key].tobytes(), (frame, k, key)
            assert grp[k].n_alive() == one[k].n_alive()
            n_events += len(got[k][0])
    assert n_events > sum(sizes) // 2                      # the keyframe frame reports every updated seed
    # a batch that is still pending refuses the grouped call as a whole: nothing is enqueued
    grp[0].update_async(kf, slots[0], cf, 0, mk.cam, T_refs[0], mk.T_cur_w)
    with pytest.raises(hip.SvoHipError):
        hip.ResidentSeeds.update_group_async(grp, kf, slots, cf, 0, mk.cam, T_refs, mk.T_cur_w)
    grp[0].collect()
    with pytest.raises(hip.SvoHipError):
        grp[1].collect()
    for r in one + grp:
        r.destroy()
    kf.destroy(); cf.destroy()
    ctx.set_small_pass_limit(8192)


def test_grouped_pass_random_shapes_both_forms_agree(ctx):
    """Eight random frames' worth of batches (1-12 keyframes, 1-2500 seeds each, some erased by hand, some frames keyframes):
    the two-launch form of small passes against the six-launch form, states and events bit for bit."""
    rng = np.random.default_rng(2024)
    for trial in range(8):
        K = int(rng.integers(1, 13))
        sizes = tuple(int(v) for v in rng.integers(1, 2500, K))
        mk = seedsynth.make_multi_keyframe_case(sizes, seed=100 + trial, width=320, height=240, border=24)
        kf = hip.Pyramid(ctx, 320, 240, 5, K)
        cf = hip.Pyramid(ctx, 320, 240, 5, 1)
        cf.upload(0, mk.cur_pyr)
        sets = ([], [])
        for k, sc in enumerate(mk.keyframes):
            kf.upload(k, sc.ref_pyr)
            s2 = (sc.sigma2 * np.float32(0.03)).astype(np.float32)
            for s_ in sets:
                s_.append(hip.ResidentSeeds(ctx, sc.px, sc.f, sc.level, sc.a, sc.b, sc.mu, sc.z_range, s2))
        for k in range(K):                                   # seeds removed behind the pass's back (removeKeyframe, reset)
            if sizes[k] > 10 and rng.random() < 0.5:
                gone = rng.choice(sizes[k], size=sizes[k] // 7, replace=False)
                for s_ in sets:
                    s_[k].erase(gone)
        T_refs = np.stack([sc.T_ref_w for sc in mk.keyframes])
        for frame in range(2):
            keyframe = bool(rng.random() < 0.4)
            out = []
            for s_, limit in zip(sets, (0, 16384)):
                ctx.set_small_pass_limit(limit)
                hip.ResidentSeeds.update_group_async(s_, kf, list(range(K)), cf, 0, mk.cam, T_refs, mk.T_cur_w, report_updated=keyframe)
                out.append([r.collect() for r in s_])
            for k in range(K):
                assert out[0][k][0].tobytes() == out[1][k][0].tobytes() and np.array_equal(out[0][k][1], out[1][k][1]), (trial, frame, k)
                d, e = sets[0][k].download(), sets[1][k].download()
                assert all(d[key].tobytes() == e[key].tobytes() for key in d), (trial, frame, k)
        for s_ in sets:
            for r in s_:
                r.destroy()
        kf.destroy(); cf.destroy()
    ctx.set_small_pass_limit(8192)


def test_seed_batches_at_keyframe_rate_do_not_allocate_after_warm_up(ctx):
    """The drop-in DepthFilter creates a keyframe's seed batch and drops the batches that age out or empty at keyframe rate, on
    the depth-filter thread (S/depth_filter.cpp:129-151,256-261).  hipFree synchronises the whole device -- the tracking thread's
    stream too -- so batches take their memory from a per-context pool: after the first keyframes have been seen, 20 more
    keyframes (a new batch of 100-500 seeds each, the oldest dropped, a pass in between) make NO allocator or free call, and a
    recycled block gives the same results as a fresh one."""
    rng = np.random.default_rng(5)
    sc, kf, cf, s2 = _case(ctx, 512, 0.02)

    def make(n):
        return hip.ResidentSeeds(ctx, sc.px[:n], sc.f[:n], sc.level[:n], sc.a[:n], sc.b[:n], sc.mu[:n], sc.z_range[:n], s2[:n])

    def one_pass(rs):
        ev, counts = rs.update(kf, 0, cf, 0, sc.cam, sc.T_ref_w, sc.T_cur_w)
        return ev, counts, rs.download()

    alive = []
    for _ in range(4):                                   # warm-up: max_n_kfs + 1 batches alive, both capacity classes (256..512) seen
        alive.append(make(int(rng.integers(300, 500))))
        one_pass(alive[-1])
        alive.append(make(int(rng.integers(100, 250))))
        one_pass(alive[-1])
    for rs in alive[:4]:
        rs.destroy()
    alive = alive[4:]
    first = make(400)
    ev0, counts0, d0 = one_pass(first)                   # reference outcome of a 400-seed batch (first pass)
    first.destroy()
    before = ctx.info()
    assert before["seed_blocks_free"] >= 1
    for k in range(20):
        n = int(rng.integers(300, 500)) if k % 2 == 0 else int(rng.integers(100, 250))
        alive.append(make(n))
        for rs in alive:
            one_pass(rs)
        alive.pop(0).destroy()                           # the oldest keyframe's seeds age out
    again = make(400)                                    # a recycled block: same results as the fresh one, bit for bit
    ev1, counts1, d1 = one_pass(again)
    again.destroy()
    after = ctx.info()
    assert after["allocator_calls"] == before["allocator_calls"], (before, after)
    assert after["free_calls"] == before["free_calls"], (before, after)
    assert list(counts0) == list(counts1) and np.array_equal(ev0["index"], ev1["index"])
    for key in ("a", "b", "mu", "sigma2", "alive"):
        assert np.array_equal(d0[key], d1[key], equal_nan=True), key
    for rs in alive:
        rs.destroy()
    info = ctx.info()
    assert info["seed_blocks_in_use"] == 0 and info["seed_blocks_free"] >= 4
    ctx.trim()
    info = ctx.info()
    assert info["seed_blocks_free"] == 0 and info["free_calls"] > after["free_calls"]
    kf.destroy(); cf.destroy()
