"""bsk_store_* (FileStore / StoreFASTXN behind the C ABI, /root/reference/bigseqkit-lib/helper.go:378-460): the single file
holds the parts in the order of their numbers whatever the order they arrive in; the directory holds one file per part.
Host-side logic only -- runs without a GPU."""
import ctypes as C
import os

from bigseqkit_amd._lib import lib


def _open(path, merge):
    s = C.c_void_p()
    assert lib.bsk_store_open(str(path).encode(), merge, C.byref(s)) == 0
    return s


def test_single_file_keeps_partition_order_for_any_arrival_order(tmp_path):
    parts = [b">a\nAC\n", b"", b">b\nGT\n>c\nTT\n", b">d\nA\n"]
    for order in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 3, 0, 2]):
        p = tmp_path / ("m%s.fa" % "".join(map(str, order)))
        s = _open(p, 1)
        for k in order:
            buf = C.create_string_buffer(parts[k], len(parts[k])) if parts[k] else None
            assert lib.bsk_store_put_host(s, k, buf, len(parts[k])) == 0
            # the file already holds every part whose predecessors have all arrived
            done = 0
            seen = set(order[:order.index(k) + 1])
            while done in seen:
                done += 1
            assert os.path.getsize(p) == sum(len(parts[j]) for j in range(done))
        total = C.c_uint64()
        assert lib.bsk_store_close(s, C.byref(total)) == 0
        assert p.read_bytes() == b"".join(parts) and total.value == len(b"".join(parts))


def test_directory_of_parts(tmp_path):
    d = tmp_path / "out"
    s = _open(d, 0)
    assert lib.bsk_store_put_host(s, 2, C.create_string_buffer(b"two", 3), 3) == 0
    assert lib.bsk_store_put_host(s, 0, C.create_string_buffer(b"zero", 4), 4) == 0
    total = C.c_uint64()
    assert lib.bsk_store_close(s, C.byref(total)) == 0
    assert sorted(os.listdir(d)) == ["part00000", "part00002"]
    assert (d / "part00000").read_bytes() == b"zero" and (d / "part00002").read_bytes() == b"two" and total.value == 7


def test_a_part_cannot_be_written_twice(tmp_path):
    s = _open(tmp_path / "x", 1)
    assert lib.bsk_store_put_host(s, 0, C.create_string_buffer(b"a", 1), 1) == 0
    assert lib.bsk_store_put_host(s, 0, C.create_string_buffer(b"b", 1), 1) != 0
    assert b"already completed" in lib.bsk_store_error(s)
    assert lib.bsk_store_close(s, None) == 0
