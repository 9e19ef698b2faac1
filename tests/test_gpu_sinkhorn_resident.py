"""The resident Sinkhorn kernel (-m gpu): all iterations in one launch in the exponential domain (one multiply-add per
element per half-iteration), workgroups of a problem exchanging column sums through tagged granules.  Checked against the oracle, against the streaming launch chain (the `stream` pin), for
bit-identical re-runs, across rounds (more problems than fit the chip at once), on ragged shapes, and under uneven load
(another stream saturating the memory system while the exchange runs)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scores(B, M, N, seed, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, M, N, generator=g) * scale


def _stream_mode(on):
    from e2e_multi_view_matching_amd import _lib
    _lib.context().set_sinkhorn_kernel("stream" if on else None)


def _rows_mode(mode):
    """None: the library's choice; "rows64" / "rows128": the 64-row kernel only / the 128-row kernel whenever the shape allows it
    (the context's pin, e2emv_set_sinkhorn_kernel)."""
    from e2e_multi_view_matching_amd import _lib
    _lib.context().set_sinkhorn_kernel(mode)


@pytest.mark.parametrize("B,M,N,iters", [(2, 128, 128, 100), (1, 100, 77, 20), (2, 33, 250, 5), (3, 1024, 1024, 100),
                                         (1, 1, 1, 3), (2, 5, 1000, 10), (2, 1000, 5, 10), (1, 513, 511, 30), (40, 256, 256, 50),
                                         (70, 300, 260, 7), (2, 2048, 2048, 30), (1, 1500, 2000, 10), (6, 2048, 2048, 12),
                                         (3, 1100, 1030, 25),
                                         # 513 ... 1024 columns = the 64-row workgroups of round 3: ragged rows, odd column slices
                                         # (no granule pairs), a partial last workgroup, many problems (several rounds)
                                         (3, 700, 900, 20), (2, 1000, 777, 15), (2, 640, 1000, 10), (20, 1024, 1000, 6), (2, 65, 1024, 12)])
def test_resident_vs_oracle_and_vs_the_streaming_chain(gpu, B, M, N, iters):
    import e2e_multi_view_matching_amd as E
    from oracle.sinkhorn import log_optimal_transport
    s = _scores(B, M, N, 7 * B + M + N)
    ref = log_optimal_transport(s, 1.0, iters)
    sg = s.to(gpu)
    try:
        _stream_mode(False)
        a = E.log_optimal_transport(sg, 1.0, iters)
        a2 = E.log_optimal_transport(sg, 1.0, iters)
        _stream_mode(True)
        b = E.log_optimal_transport(sg, 1.0, iters)
    finally:
        _stream_mode(False)
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, a2)                                        # fixed reduction orders: bit-identical re-run
    assert float((a.cpu() - ref).abs().max()) < 1e-4, float((a.cpu() - ref).abs().max())
    assert float((a - b).abs().max()) < 2e-5                         # two summation orders of the same algorithm
    ra, rb = a[:, :-1, :-1], ref[:, :-1, :-1]
    assert torch.equal(ra.argmax(2).cpu(), rb.argmax(2)) and torch.equal(ra.argmax(1).cpu(), rb.argmax(1))


def test_more_problems_than_resident_capacity_and_batch_independence(gpu):
    """16 problems of 1024 x 1024 fit the registers of the chip at once; 37 go through three rounds of the same
    workgroups (epochs keep counting) and every problem must come out exactly as when it runs alone."""
    import e2e_multi_view_matching_amd as E
    s = _scores(37, 1024, 1024, 5).to(gpu)
    for mode in ("rows64", "rows128"):  # (the same kernel for the batch and for the problem alone: three rounds / two)
        try:
            _rows_mode(mode)
            full = E.log_optimal_transport(s, 1.0, 25)
            for b in (0, 15, 16, 31, 32, 36):
                assert torch.equal(E.log_optimal_transport(s[b:b + 1].contiguous(), 1.0, 25)[0], full[b]), (mode, b)
        finally:
            _rows_mode(None)
    s = _scores(11, 2048, 2048, 6).to(gpu)  # 2048 columns: 4 resident (three rounds) / 8 (two rounds)
    for mode in ("rows64", "rows128"):
        try:
            _rows_mode(mode)
            full = E.log_optimal_transport(s, 1.0, 12)
            for b in (0, 3, 4, 7, 8, 10):
                assert torch.equal(E.log_optimal_transport(s[b:b + 1].contiguous(), 1.0, 12)[0], full[b]), (mode, b)
        finally:
            _rows_mode(None)


@pytest.mark.parametrize("B,M,N,iters,picked", [
    (32, 1024, 1024, 100, True),   # BASELINE configs[1]'s Sinkhorn: one round instead of two
    (64, 1024, 1024, 10, True),    # two rounds instead of four
    (65, 1024, 1024, 4, True),     # two rounds of 128-row workgroups + one problem on 64-row ones (a second launch)
    (80, 1024, 1024, 3, True),     # BASELINE configs[3]'s Sinkhorn: 64 + 16
    (20, 1024, 1008, 6, True),     # columns short of 1024 (masked chunk tails), 126-column slices
    (18, 1000, 1024, 8, True),     # rows short of 8 x 128: a partial last workgroup
    (24, 900, 1020, 12, True),     # both
    (30, 640, 1020, 9, True),      # 5 workgroups per problem, 204-column slices
    (52, 640, 1020, 4, True),      # 51 + 1
    (40, 640, 1024, 9, False),     # 205-column slices are odd (the exchange moves column pairs): the 64-row kernel
    (16, 1024, 1024, 5, False),    # fits one round of 64-row workgroups already: not taken
    # 1025 .. 2048 columns: sinkhorn_resident2k, 64 rows per workgroup (12 of a wave's 16 rows in registers by number)
    (16, 2048, 2048, 6, True),     # BASELINE configs[4]'s shape: two rounds of 8 instead of four of 4
    (16, 1984, 2040, 5, True),     # ragged rows and columns (31 workgroups, 66-column slices)
    (16, 1536, 1824, 4, True),     # 24 workgroups per problem, 76-column slices
    (16, 1500, 1800, 4, False),    # 75-column slices are odd: the 32-row kernel
    (4, 2048, 2048, 4, False),     # one round either way
    (12, 2048, 2048, 4, True),     # 8 + 4
])
def test_128_rows_per_workgroup(gpu, B, M, N, iters, picked):
    """sinkhorn_resident128: a workgroup of 4 waves holds 128 rows of K = exp(S - rowmax) - 24 rows per wave in registers it
    addresses by number (12 in v64 .. v255, 12 in a64 .. a255), 8 in LDS - so that twice as many problems are resident and a
    large batch needs half the rounds.  Against the oracle (a sample of the problems), against the 64-row kernel (another
    summation order of the same algorithm), bit-identical re-runs, and the library's choice reported by the statistics."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    from oracle.sinkhorn import log_optimal_transport
    s = _scores(B, M, N, 3 * B + M + N)
    sample = sorted({0, B // 2, B - 1})
    ref = log_optimal_transport(s[sample], 1.0, iters)
    sg = s.to(gpu)
    ctx = _lib.context(gpu)
    try:
        _rows_mode(None)
        ctx.stats(reset=True)
        a = E.log_optimal_transport(sg, 1.0, iters)
        a2 = E.log_optimal_transport(sg, 1.0, iters)
        st = ctx.stats()
        _rows_mode("rows64")
        b = E.log_optimal_transport(sg, 1.0, iters)
        assert ctx.stats()["sinkhorn_rows128_calls"] == st["sinkhorn_rows128_calls"]
    finally:
        _rows_mode(None)
    assert st["sinkhorn_rows128_calls"] == (2 if picked else 0), st
    assert st["sinkhorn_rescued"] == 0 and st["sinkhorn_timeouts"] == 0 and st["sinkhorn_bad"] == 0
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, a2)
    assert float((a[sample].cpu() - ref).abs().max()) < 1e-4, float((a[sample].cpu() - ref).abs().max())
    assert float((a - b).abs().max()) < 2e-5, float((a - b).abs().max())
    assert torch.equal(a[:, :-1, :-1].argmax(2), b[:, :-1, :-1].argmax(2))


def test_matches_from_the_resident_path_equal_the_streaming_chain(gpu):
    """The fused row / column arg-max of the final phase feeds the match block: same matches through both paths."""
    from e2e_multi_view_matching_amd import MultiViewMatcher
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
    torch.manual_seed(0)
    model = identity_like_state(MultiViewMatcher({"GNN_layers": ["self", "cross"], "sinkhorn_iterations": 50, "conf_mlp": True}).eval()).to(gpu)
    data = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in make_tuples(batch=3, tuple_size=3, n_kpts=700, seed=2).items()}
    model.config["multi_frame_matching"] = True
    try:
        _stream_mode(False)
        a = model(data)
        _stream_mode(True)
        b = model(data)
    finally:
        _stream_mode(False)
    for k in a:
        if k.startswith("matches"):
            assert torch.equal(a[k], b[k]), k
        elif k.startswith("scores_"):
            assert float((a[k] - b[k]).abs().max()) < 2e-5, k
    assert float((a["matches0_0_1"] >= 0).float().mean()) > 0.5


@pytest.mark.parametrize("B,N,iters", [(20, 512, 40), (32, 1024, 25), (40, 1024, 10), (9, 2048, 10)])
def test_exchange_under_uneven_load(gpu, B, N, iters):
    """Hand-offs must not depend on timing or placement: run the resident kernel while a second stream streams 2 GB
    copies (its workgroups occupy CUs and the memory queues unevenly), many times, and require the quiet result.
    (512 columns: the compiler-allocated kernel; 32 x 1024 / 9 x 2048: the kernels with K in registers addressed by number -
    their waves must find those registers untouched whatever else runs on the chip; 40 x 1024: two resident launches.)"""
    import e2e_multi_view_matching_amd as E
    s = _scores(B, N, N, 11).to(gpu)
    quiet = E.log_optimal_transport(s, 1.0, iters)
    side = torch.cuda.Stream(device=gpu)
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=gpu)
    other = torch.empty_like(big)
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4):
                other.copy_(big)
                big.add_(1.0)
        out = E.log_optimal_transport(s, 1.0, iters)
        assert torch.equal(out, quiet), rep
    torch.cuda.synchronize()


def test_dynamic_range_of_the_exponential_domain(gpu):
    """The resident kernel iterates a = exp(u + rowmax), b = exp(v) instead of log-sum-exps.  Measured against an fp64
    oracle it is MORE accurate than the fp32 log-domain forms up to |logZ| ~ 700 (scores spread over hundreds of nats);
    when a scaling finally leaves fp32's range the rescue pass behind the kernel re-solves that problem in the log domain
    inside the same call (counted in stats, nothing raised) - as the streaming chain (the `stream` pin) does."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    from oracle.sinkhorn import log_optimal_transport
    ctx = _lib.context(gpu)
    for scale, bar in ((10.0, 1e-4), (40.0, 1e-4)):
        s = _scores(2, 300, 280, int(scale), scale=scale)
        ref = log_optimal_transport(s.double(), 1.0, 100).float()
        out = E.log_optimal_transport(s.to(gpu), 1.0, 100).cpu()
        assert float((out - ref).abs().max()) < bar, scale
        assert torch.equal(out[:, :-1, :-1].argmax(2), ref[:, :-1, :-1].argmax(2))
        assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK
    s = _scores(2, 300, 280, 160, scale=160.0)  # |logZ| > 1200: far outside anything a descriptor network produces
    ctx.stats(reset=True)
    out = E.log_optimal_transport(s.to(gpu), 1.0, 100).cpu()
    ref = log_optimal_transport(s.double(), 1.0, 100).float()
    # the rescue pass behind the resident kernel re-solved what left fp32's range in the exponential domain: finite, right,
    # nothing raised - and counted
    assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK
    assert bool(torch.isfinite(out).all()) and float((out - ref).abs().max()) < 5e-3  # fp32 log domain at |logZ| ~ 1200
    st = ctx.stats()
    assert st["sinkhorn_bad"] == 0
    assert st["sinkhorn_timeouts"] == 0  # a range event, not a wait that gave up
    if st["sinkhorn_rescued"]:
        # ONE observed range event does not demote the context (round 4): the next call is rescued again ...
        again = E.log_optimal_transport(s.to(gpu), 1.0, 100).cpu()
        assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK and float((again - ref).abs().max()) < 5e-3
        st2 = ctx.stats()
        assert st2["sinkhorn_rescued"] > st["sinkhorn_rescued"]
        # ... and from the SECOND the log-domain chain serves this model: no new rescues, the same answers ...
        for _ in range(3):
            again = E.log_optimal_transport(s.to(gpu), 1.0, 100).cpu()
        assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK and float((again - ref).abs().max()) < 5e-3
        assert ctx.stats()["sinkhorn_rescued"] == st2["sinkhorn_rescued"]
        # ... where non-finite scores are still counted (the chain checks its final potentials)
        bad = s.clone()
        bad[0, 3, 3] = float("inf")
        E.log_optimal_transport(bad.to(gpu), 1.0, 100)
        assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.EHIP
        assert ctx.stats()["sinkhorn_bad"] == 1
        # after 16 calls on the chain the resident kernel gets another try: an ordinary problem then runs on it again
        # (an out-of-range model would be demoted again by its next event)
        easy = _scores(2, 300, 280, 3, scale=10.0)
        for _ in range(18):
            out_e = E.log_optimal_transport(easy.to(gpu), 1.0, 100).cpu()
        ref_e = log_optimal_transport(easy.double(), 1.0, 100).float()
        assert float((out_e - ref_e).abs().max()) < 1e-4
        third = E.log_optimal_transport(s.to(gpu), 1.0, 100).cpu()  # back on the resident kernel: rescued (and counted) again
        assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK and float((third - ref).abs().max()) < 5e-3
        assert ctx.stats()["sinkhorn_rescued"] > st2["sinkhorn_rescued"]
    ctx.stats(reset=True)                                                     # back to the resident kernel for the other tests
    # non-finite SCORES stay an error: loud, once
    bad = s.clone()
    bad[1, 5, 7] = float("nan")
    out = E.log_optimal_transport(bad.to(gpu), 1.0, 100)
    assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.EHIP and b"non-finite" in ctx.lib.e2emv_last_error(ctx.h)
    assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK                         # the report is consumed once
    assert bool(torch.isfinite(out[0]).all()) and not bool(torch.isfinite(out[1]).all())
    assert ctx.stats(reset=True)["sinkhorn_bad"] == 1
    try:
        _stream_mode(True)
        out = E.log_optimal_transport(s.to(gpu), 1.0, 100).cpu()
    finally:
        _stream_mode(False)
    ref = log_optimal_transport(s.double(), 1.0, 100).float()
    assert bool(torch.isfinite(out).all()) and float((out - ref).abs().max()) < 5e-3  # fp32 log domain at |logZ| ~ 1200


def test_the_launchers_plan_is_queryable_and_follows_the_pin(gpu):
    """e2emv_sinkhorn_plan (round 6): what launch_sinkhorn would run for a batch on this context - bench.py reports its Sinkhorn bound
    from it instead of re-deriving the heuristic - and the per-context kernel pin that replaces the per-call getenv."""
    from e2e_multi_view_matching_amd import _lib
    ctx = _lib.context(gpu)
    try:
        ctx.set_sinkhorn_kernel(None)
        assert ctx.sinkhorn_plan(32, 1024, 1024, 100) == [{"rows_per_workgroup": 128, "problems": 32, "resident_problems": 32, "rounds": 1}]
        p80 = ctx.sinkhorn_plan(80, 1024, 1024, 100)   # configs[3]: 64 on 128-row workgroups (2 rounds) + 16 on 64-row ones
        assert [(s["rows_per_workgroup"], s["problems"], s["rounds"]) for s in p80] == [(128, 64, 2), (64, 16, 1)]
        assert ctx.sinkhorn_plan(80, 2048, 2048, 100) == [{"rows_per_workgroup": 64, "problems": 80, "resident_problems": 8, "rounds": 10}]
        assert ctx.sinkhorn_plan(4, 256, 256, 0) == []                      # iters = 0: the log-domain chain
        ctx.set_sinkhorn_kernel("rows64")
        assert [s["rows_per_workgroup"] for s in ctx.sinkhorn_plan(32, 1024, 1024, 100)] == [64]
        ctx.set_sinkhorn_kernel("rows128")
        assert [(s["rows_per_workgroup"], s["rounds"]) for s in ctx.sinkhorn_plan(80, 1024, 1024, 100)] == [(128, 3)]
        ctx.set_sinkhorn_kernel("stream")
        assert ctx.sinkhorn_plan(32, 1024, 1024, 100) == []
    finally:
        ctx.set_sinkhorn_kernel(None)
