"""Host logic of the own GEMM family that needs no GPU: the weight-gradient kernel's workgroup -> (row split, tile) map
(vitxt_gqa_amd/csrc/gemm_bf16.hip tn_item, mirrored in vitxt_gqa_amd/gemm.py) takes every item exactly once."""
import itertools

from vitxt_gqa_amd import gemm as G


def test_wgrad_workgroup_map_is_a_bijection_onto_the_items():
    shapes = [1, 2, 3, 4, 9, 12, 27, 31, 32, 33, 36, 48, 64, 100, 144]
    for T, cus in itertools.product(shapes, (256, 304, 64)):
        for S in sorted({1, 2, 3, 5, 7, 8, 9, 28, max(1, cus // T), max(1, cus // T) + 3}):
            c = G.wgrad_grid(T, S, cus)
            seen = {}
            for wg in range(8 * c):
                it = G.wgrad_item(wg, c, T, S)
                if it is None:
                    continue
                assert 0 <= it[0] < S and 0 <= it[1] < T, (T, S, cus, wg, it)
                assert it not in seen, "item %s taken by workgroups %d and %d (T=%d S=%d c=%d)" % (it, seen[it], wg, T, S, c)
                seen[it] = wg
            assert len(seen) == T * S, "T=%d S=%d cus=%d c=%d: %d of %d items taken" % (T, S, cus, c, len(seen), T * S)


def test_wgrad_map_keeps_a_row_split_on_one_xcd_for_the_step_shapes():
    """The point of the map: the workgroups of one XCD (ids equal mod 8) read the same rows.  FFN weights (36 tiles, 7 splits on 256 CUs):
    XCDs 0..6 hold 32 tiles of ONE split each; QKV (27 tiles, 9 splits): every XCD hosts one whole split."""
    c = G.wgrad_grid(36, 7)
    assert c == 32
    for x in range(7):
        assert {G.wgrad_item(x + 8 * j, c, 36, 7)[0] for j in range(32)} == {x}
    c = G.wgrad_grid(27, 9)
    for x in range(8):
        hosted = [G.wgrad_item(x + 8 * j, c, 27, 9) for j in range(27)]
        assert {h[0] for h in hosted} == {x} and sorted(h[1] for h in hosted) == list(range(27))


def test_nt_workgroup_map_covers_every_tile_once_and_groups_the_weight_panels():
    """csrc/gemm_bf16.hip nt_item through its Python mirror: every (M-block, N-tile) exactly once for ragged shapes and every group width;
    with b = 4 on the step's FFN shape the 32 workgroups an XCD runs at a time see 4 weight panels and 8 activation panels."""
    from vitxt_gqa_amd import gemm as G
    for tiles_m in (1, 3, 7, 8, 9, 64, 2539):
        for tiles_n in (1, 3, 9, 12, 13):
            for b in sorted({1, 2, 3, 4, 5, tiles_n, G.nt_group(tiles_n)}):
                if b > tiles_n:
                    continue
                seen = {}
                for wg in range(G.nt_grid(tiles_m, tiles_n)):
                    it = G.nt_item(wg, tiles_m, tiles_n, b)
                    if it is not None:
                        assert 0 <= it[0] < tiles_m and 0 <= it[1] < tiles_n
                        assert it not in seen, (tiles_m, tiles_n, b, it, wg, seen[it])
                        seen[it] = wg
                assert len(seen) == tiles_m * tiles_n, (tiles_m, tiles_n, b, len(seen))
    assert G.nt_group(12) == 12      # shipped default: all N-tiles of an M-block side by side
    tiles_m, tiles_n, b = 2539, 12, 4
    for x in range(8):
        first = [G.nt_item(8 * j + x, tiles_m, tiles_n, b) for j in range(32)]      # the first 32 workgroups of XCD x
        assert len({t[1] for t in first}) == 4 and len({t[0] for t in first}) == 8
        assert len({G.nt_item(8 * j + x, tiles_m, tiles_n, b)[0] // ((tiles_m + 7) // 8) for j in range(0, 3000, 97)}) == 1      # one M range per XCD


def test_wgrad_supported_mirrors_the_span_limits_of_the_c_side():
    """`functional._wgrad` sends a shape to t2s_gemm_wgrad only when `wgrad_supported` says so, so the Python rule must hold
    the same limits as the C side (csrc/gemm_bf16.hip t2s_gemm_wgrad): <= 4096 splits and a row split (padded to 128 rows + one 64-row
    K-tile of read-ahead) spanning < 2 GB of either operand.  With the automatic split count (one round on 256 CUs: 7 for the FFN
    weights) ~2.4 M rows at ld = 3072 cross it: the call must be refused HERE (library fallback), not raise inside the C call."""
    assert G.wgrad_supported(649_984, 768, 3072)                   # the step's FFN shapes (B = 64)
    assert G.wgrad_supported(649_984, 3072, 768)
    assert G.wgrad_supported(649_984, 2304, 768) and G.wgrad_supported(649_984, 768, 768)
    assert not G.wgrad_supported(512, 768, 768)                    # short contraction: library
    assert not G.wgrad_supported(649_984, 768, 1000)               # not a multiple of 256
    auto = 7
    ok_rows = ((1 << 31) // (3072 * 2) - 64) // 128 * 128 * auto - 128
    assert G.wgrad_supported(ok_rows, 768, 3072, splits=auto)
    assert not G.wgrad_supported(3_000_000, 768, 3072, splits=auto)        # 428 672-row splits x 6 KB rows = 2.6 GB: refused
    assert G.wgrad_supported(3_000_000, 768, 3072, splits=16)              # ... unless the caller asks for more splits
    assert not G.wgrad_supported(3_000_000, 768, 3072, splits=5000)        # > 4096 splits
    assert not G.wgrad_supported(649_984, 768, 768, ld_dy=1 << 24)         # a wide row stride counts, not the row width
    assert G.nt_supported(649_984, 3072, 768) and not G.nt_supported(649_984, 3072, 768, lda=1 << 22)


def test_persistent_nt_walk_takes_every_tile_exactly_once():
    """Round 6: the NT kernel is persistent - one workgroup per CU (cus / 8 per XCD) walks the items of its XCD's range (slot, slot +
    per_xcd, ...) and stops at the first id without an item.  Every tile exactly once for ragged shapes, group widths and card sizes; a
    workgroup's walk stays inside its XCD's M-range; consecutive slots hold consecutive items (the L2 locality of the one-tile dispatch)."""
    for tiles_m, tiles_n in ((1, 3), (7, 12), (9, 9), (64, 12), (2539, 12), (2539, 3)):
        for b in sorted({1, 4, tiles_n}):
            for per_xcd in (1, 5, 32, 38):
                seen = {}
                for x in range(8):
                    for slot in range(per_xcd):
                        for t in G.nt_walk(x, slot, per_xcd, tiles_m, tiles_n, b):
                            assert t not in seen, (tiles_m, tiles_n, b, per_xcd, t, seen[t], (x, slot))
                            seen[t] = (x, slot)
                            assert t[0] // ((tiles_m + 7) // 8) == x
                assert len(seen) == tiles_m * tiles_n, (tiles_m, tiles_n, b, per_xcd, len(seen))
    first = [G.nt_walk(0, s, 32, 2539, 12, 12)[0] for s in range(32)]
    assert first == [G.nt_item(8 * s, 2539, 12, 12) for s in range(32)]
