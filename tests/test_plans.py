"""Host-side checks of the shipped kernel-configuration table (reconvat_amd/plans.py, tuned_plans.json): it parses, it is the
default mode, keys have the documented arity, tile codes are well-formed, and the batch-agnostic fall-back resolves."""
import os


def test_table_parses_and_is_default():
    from reconvat_amd import plans
    assert os.path.exists(plans.PLAN_FILE)
    assert plans.default_mode() == 'table' or os.environ.get('RV_AUTOTUNE') is not None
    conv, wgrad = plans.conv_entries(), plans.wgrad_entries()
    assert len(conv) >= 40 and len(wgrad) >= 20 and plans.digest()
    for key, algo in conv.items():
        assert len(key) == 10 and key[0] in (0, 1, 2, 3) and key[1] in (1, 8)
        fam, nt, mt, th = (algo >> 8) & 15, (algo >> 4) & 15, algo & 15, algo >> 12
        assert algo in (0, 1, 2) or (fam in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13) and 1 <= nt <= 4 and 1 <= mt <= 8 and 0 <= th < 256), hex(algo)
        if key[0] != 0:
            assert fam in (0, 1, 5), 'only the 3x3 mode has LDS tile families'
    gemm = plans.gemm_entries()
    assert len(gemm) >= 20 and all(len(k) == 8 and 1 <= v <= 32 for k, v in gemm.items())
    for key, (nw, wgs) in wgrad.items():
        assert len(key) == 6 and key[0] in (1, 4, 9) and nw in (0, 4, 8, 24) and (wgs == 0 or 32 <= wgs <= 4096)


def test_lookup_exact_and_borrowed():
    from reconvat_amd import plans
    key, algo = sorted(plans.conv_entries().items())[0]
    assert plans.lookup_conv(key) == (algo, True)
    # same layer geometry at another batch size: the entry of the nearest batch size that is not smaller, else the largest below
    conv = plans.conv_entries()
    both = [(k, v) for k, v in sorted(conv.items()) if k[1] == 1 and (k[0], 8) + k[2:] in conv]
    assert both, 'the table holds B = 1 and B = 8 entries of the same geometry (config 2 and the headline workload)'
    k1, a1 = both[0]
    a8 = conv[(k1[0], 8) + k1[2:]]
    assert plans.lookup_conv((k1[0], 4) + k1[2:]) == (a8, False)      # B = 4 -> the B = 8 tile
    assert plans.lookup_conv((k1[0], 16) + k1[2:]) == (a8, False)     # B = 16 -> the largest one below
    assert plans.lookup_conv((k1[0], 1) + k1[2:]) == (a1, True)
    assert plans.lookup_conv((0, 8, 33, 17, 16, 16, 16, 16, 0, 0)) is None
    wkey, plan = sorted(plans.wgrad_entries().items())[0]
    assert plans.lookup_wgrad(wkey) == plan
    w8 = plans.wgrad_entries().get((wkey[0], 8) + wkey[2:])
    if w8 is not None:
        assert plans.lookup_wgrad((wkey[0], 2) + wkey[2:]) == w8
    assert plans.lookup_wgrad((9, 8, 33, 17, 16, 16)) is None


def test_shipped_table_digest_is_pinned():
    """The digest bench.py prints and profiles/*_pmc_traffic.json records: a changed table must be a deliberate, reviewed change
    (regenerate with tools/tune_plans.py on an MI355X, re-run the -m gpu suite against it, then update this value)."""
    from reconvat_amd import plans
    assert plans.digest() == PINNED_DIGEST, (plans.digest(), 'tuned_plans.json changed: re-run the GPU suite and pin the new digest')
    meta = plans.meta()
    assert 'MI355X' in meta.get('device', '') and meta.get('git'), meta.get('device')


PINNED_DIGEST = '644fe159bb9eb70e'
