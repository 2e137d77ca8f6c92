"""nodedb/encoding_test.go on the oracle, and the order it pins on the product.

The reference's NodeDb index key is a byte string — node type id (big-endian u64), one sign-biased big-endian i64 per indexed resource holding the allocatable
rounded DOWN to the index resolution, node index (encoding.go:22-89) — compared with bytes.Compare.  The oracle keeps the same fields in a struct and compares them
field by field; the product packs the quotients and the node's rank into one integer.  Here:
  * TestRoundQuantityToResolution (:14-47, 4 cases), TestNodeIndexKeyComparison (:49-87: the expected 40 bytes), TestNodeIndexKey (:89-184, 6 comparisons),
    TestRoundedNodeIndexKeyFromResourceList (:186-211) against the oracle's byte encoding (oracle_node_index_key_bytes), transcribed by hand;
  * the oracle's struct comparison == bytes.Compare of those encodings on seeded keys (negative quantities included);
  * NodeTypeIterator order on the product (asched_iterate_nodes: CPU build here, HIP library with -m gpu) == ascending byte keys of the nodes' allocatable at that
    priority, on seeded NodeDbs with running jobs (ties broken by the node index, as the trailing key field does).
"""
import ctypes as C
import os

import numpy as np
import pytest

from armada_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def enc(oracle_lib):
    lib = C.CDLL(oracle_lib.path)
    lib.oracle_round_quantity_to_resolution.restype = C.c_int64
    lib.oracle_round_quantity_to_resolution.argtypes = [C.c_int64, C.c_int64]
    lib.oracle_node_index_key_bytes.restype = C.c_int32
    lib.oracle_node_index_key_bytes.argtypes = [C.c_uint64, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_uint8)]
    lib.oracle_node_index_key_compare.restype = C.c_int32
    lib.oracle_node_index_key_compare.argtypes = [C.c_uint64, C.POINTER(C.c_int64), C.c_uint64, C.c_uint64, C.POINTER(C.c_int64), C.c_uint64, C.c_int32]
    return lib


def key_bytes(enc, type_id, resources, resolution=None, node_index=0) -> bytes:
    k = len(resources)
    r = (C.c_int64 * max(k, 1))(*resources)
    res = (C.c_int64 * max(k, 1))(*resolution) if resolution is not None else None
    out = (C.c_uint8 * (8 * (k + 2)))()
    n = enc.oracle_node_index_key_bytes(type_id, r, k, res, node_index, out)
    return bytes(out[:n])


def cmp(a, b):
    return (a > b) - (a < b)


@pytest.mark.parametrize("q,resolution,expected", [(1024, 1024, 1024), (2001, 1000, 2000), (2999, 1000, 2000), (0, 1000, 0)])
def test_round_quantity_to_resolution(enc, q, resolution, expected):
    assert enc.oracle_round_quantity_to_resolution(q, resolution) == expected


def test_node_index_key_comparison(enc):
    """encoding_test.go:49-87.  The ResourceList factory has already scaled the quantities (cpu 0.999958006 -> 1000 milli, rounded up; memory in bytes)."""
    expected = bytes([0x00] * 8 + [0x80, 0, 0, 0, 0, 0, 0x03, 0xe8] + [0x80, 0, 0, 0x02, 0xc0, 0xbf, 0x0d, 0xe8] + [0x80, 0, 0, 0, 0, 0, 0, 0] + [0x00] * 8)
    assert key_bytes(enc, 0, [1000, 11823681536, 0], [1000, 1000, 1000], 0) == expected     # RoundedNodeIndexKeyFromResourceList
    assert key_bytes(enc, 0, [1000, 11823681000, 0]) == expected                               # NodeIndexKey


@pytest.mark.parametrize("a,b", [((10, []), (10, [])), ((10, []), (11, [])), ((10, [1, 2]), (10, [1, 2])), ((10, [2, 1]), (10, [1, 2])), ((10, [1, 2]), (11, [1, 2])),
                                 ((10, [1, 2]), (10, [-1, 2]))])
def test_node_index_key(enc, a, b):
    """encoding_test.go:89-184: bytes.Compare of the keys == comparing (node type id, then the quantities in order)"""
    want = cmp((a[0], a[1]), (b[0], b[1]))
    assert cmp(key_bytes(enc, a[0], a[1]), key_bytes(enc, b[0], b[1])) == want
    k = len(a[1])
    ra, rb = (C.c_int64 * max(k, 1))(*a[1]), (C.c_int64 * max(k, 1))(*b[1])
    assert enc.oracle_node_index_key_compare(a[0], ra, 0, b[0], rb, 0, k) == want


def test_rounded_node_index_key_from_resource_list(enc):
    """encoding_test.go:186-211: memory 1, cpu 2 (= 2000 milli) under resolutions (1, 2000) and (1, 1500)"""
    assert key_bytes(enc, 0, [1, 2000]) == key_bytes(enc, 0, [1, 2000], [1, 2000])
    assert key_bytes(enc, 0, [1, 1500]) == key_bytes(enc, 0, [1, 2000], [1, 1500])


def test_struct_comparison_is_bytes_compare(enc):
    rng = np.random.default_rng(11)
    for _ in range(4000):
        k = int(rng.integers(0, 5))
        ta, tb = (int(x) for x in rng.integers(0, 3, size=2))
        ia, ib = (int(x) for x in rng.integers(0, 3, size=2))
        span = int(rng.choice([2, 300, 2 ** 40]))
        a = [int(x) for x in rng.integers(-span, span, size=k)]
        b = [int(x) if rng.random() < 0.5 else a[i] for i, x in enumerate(rng.integers(-span, span, size=k))]
        ra, rb = (C.c_int64 * max(k, 1))(*a), (C.c_int64 * max(k, 1))(*b)
        assert enc.oracle_node_index_key_compare(ta, ra, ia, tb, rb, ib, k) == cmp(key_bytes(enc, ta, a, None, ia), key_bytes(enc, tb, b, None, ib))


def _iteration_equals_byte_order(lib, enc, seeds):
    checked = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        wl = W.small_random(n_nodes=int(rng.integers(20, 200)), n_jobs=int(rng.integers(100, 1500)), n_queues=3, seed=seed, occupied=float(rng.choice([0.3, 0.7, 0.95])))
        s = W.load(lib, wl)
        cfg = wl.config
        cols, res = list(cfg.indexed_col), list(cfg.indexed_resolution)
        alloc = s.get_nodes_alloc()                                # [N][P][R] allocatable by priority (asched_get_nodes_alloc)
        prios = list(s.priorities)
        for level, prio in enumerate(prios):
            got = s.iterate_nodes(prio, [0] * len(cols))            # every node of every type, in NodeTypesIterator order
            keys = {n: key_bytes(enc, 0, [int(alloc[n][level][c]) for c in cols], res, n) for n in range(wl.num_nodes)}
            want = sorted(range(wl.num_nodes), key=lambda n: keys[n])
            assert got == want, (seed, prio)
            checked += len(got)
        s.close()
    return checked


def test_product_iteration_order_is_byte_key_order(hostsim_lib, enc):
    assert _iteration_equals_byte_order(hostsim_lib, enc, range(4200, 4212)) > 2000


@pytest.mark.gpu
def test_product_iteration_order_is_byte_key_order_gpu(hip_lib, enc):
    assert _iteration_equals_byte_order(hip_lib, enc, range(4200, 4208)) > 1000
